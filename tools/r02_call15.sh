#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02m
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_gemm_family.py -q -x > $O/pytest_family.txt 2>&1
tail -5 $O/pytest_family.txt
timeout 600 python tools/profile_layers.py resnet50 32 5 int8 > $O/layers_resnet50_int8_b32.txt 2>&1
cat $O/layers_resnet50_int8_b32.txt
timeout 600 python -m pytest tests/test_gpu_baseline_batches.py -q -x -k "resnet50 or mobilenet" > $O/pytest_batches.txt 2>&1
tail -3 $O/pytest_batches.txt
