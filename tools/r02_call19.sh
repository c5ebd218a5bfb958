#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02s
mkdir -p $O
cd $R
timeout 120 tools/exp/flag_handoff.bin > $O/flag_handoff.txt 2>&1
cat $O/flag_handoff.txt
timeout 900 python -m pytest tests/test_gpu_parity_uint8.py tests/test_gpu_baseline_batches.py tests/test_plugin_dropin.py -q -x -k "maxpool or yolo or route" > $O/pytest_convpool.txt 2>&1
tail -15 $O/pytest_convpool.txt
timeout 300 python tools/profile_layers.py yolov3_tiny 1 20 uint8 > $O/layers_yolov3_tiny_uint8_b1.txt 2>&1
cat $O/layers_yolov3_tiny_uint8_b1.txt
