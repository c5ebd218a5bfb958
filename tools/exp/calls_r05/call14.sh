#!/bin/bash
# round 5, call 14: two 16-channel slices per pwdw block + accumulators from the bias in the GEMM family / pw_stream: parity, then A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_call14
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_pwdw.py tests/test_gpu_parity.py tests/test_gpu_gemm_family.py tests/test_gpu_pgemm.py tests/test_gpu_baseline_batches.py -m gpu -x -q 2>&1 | tail -5 | tee $O/pytest_subset.txt
PRE=$R/tools/exp/ab/libtengine_amd_r05_pre_window.so
for cfg in "mobilenet_v1 64 int8 100" "mobilenet_v1 1 int8 1000" "resnet50 32 int8 100"; do
  set -- $cfg
  AB_LAYERS=1 timeout 900 python tools/exp/ab_lib.py $1 $2 $3 $4 2 before=$PRE now=product > $O/ab_slices_layers_$1_b$2.txt 2>&1
  grep -v "^Tengine" $O/ab_slices_layers_$1_b$2.txt | cut -c1-160
done
