#!/bin/bash
# uint8: dword loads in the depthwise kernel, division-free rounding in both uint8 quantisers -- whole suite, then tables
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02ai
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu_all.txt 2>&1
tail -4 $O/pytest_gpu_all.txt
timeout 300 python tools/profile_layers.py mssd 16 10 uint8 > $O/layers_mssd_uint8_b16.txt 2>&1; grep -E "conv0|dw|sum of" $O/layers_mssd_uint8_b16.txt | head -16
timeout 300 python tools/profile_layers.py yolov3_tiny 8 10 uint8 > $O/layers_yolov3_tiny_uint8_b8.txt 2>&1; head -4 $O/layers_yolov3_tiny_uint8_b8.txt; tail -1 $O/layers_yolov3_tiny_uint8_b8.txt
timeout 300 python bench.py --model mssd --dtype uint8 --batch 16 --steps 50 --no-cpu-baseline > $O/bench_mssd_uint8_b16.json 2>/dev/null; tail -1 $O/bench_mssd_uint8_b16.json | cut -c1-200
timeout 300 python bench.py --model yolov3_tiny --dtype uint8 --batch 8 --steps 50 --no-cpu-baseline > $O/bench_yolov3_tiny_uint8_b8.json 2>/dev/null; tail -1 $O/bench_yolov3_tiny_uint8_b8.json | cut -c1-200
