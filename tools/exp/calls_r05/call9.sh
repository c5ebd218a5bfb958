#!/bin/bash
# round 5, call 9: the one-binade requantisation (result byte and hand-over flag read off the bit pattern for windows that start at
# 128.25) with a run-time branch inside requant4 -- parity first, then A/B against the build before it
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_call9
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_gemm_family.py tests/test_gpu_pwdw.py tests/test_gpu_dwpw.py tests/test_gpu_pgemm.py -m gpu -x -q 2>&1 | tail -5 | tee $O/pytest_subset.txt
PRE=$R/tools/exp/ab/libtengine_amd_r05_pre_window.so
for cfg in "mobilenet_v1 64 int8 100" "resnet50 32 int8 100" "mobilenet_v1 1 int8 2000"; do
  set -- $cfg
  timeout 600 python tools/exp/ab_lib.py $1 $2 $3 $4 3 before=$PRE window=product > $O/ab_window_$1_b$2.txt 2>&1
  grep -v "^Tengine" $O/ab_window_$1_b$2.txt
done
