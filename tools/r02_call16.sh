#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02p
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_gemm_family.py tests/test_gpu_parity.py tests/test_gpu_baseline_batches.py -q -x -k "elt or resnet" > $O/pytest_elt.txt 2>&1
tail -3 $O/pytest_elt.txt
timeout 300 python tools/profile_layers.py resnet50 32 20 int8 > $O/layers_resnet50_int8_b32.txt 2>&1
grep -E "branch2c|branch1 |sum of" $O/layers_resnet50_int8_b32.txt
