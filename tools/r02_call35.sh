#!/bin/bash
# uint8 MFMA conv: staging of the next stage in the MFMAs' shadow -- uint8 parity first, then the tables
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02aj
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity_uint8.py tests/test_gpu_baseline_batches.py tests/test_gpu_edge_cases.py -m gpu -q -x 2>&1 | tail -4 | tee $O/pytest_uint8.txt
timeout 300 python tools/profile_layers.py yolov3_tiny 8 10 uint8 > $O/layers_yolov3_tiny_uint8_b8.txt 2>&1; cat $O/layers_yolov3_tiny_uint8_b8.txt
timeout 300 python tools/profile_layers.py mssd 16 10 uint8 > $O/layers_mssd_uint8_b16.txt 2>&1; tail -1 $O/layers_mssd_uint8_b16.txt
