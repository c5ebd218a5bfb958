#!/bin/bash
# round 5, call 24: the residual tail fed with the conv's result as a float (no pack / unpack between the two requantisations), pair-wise
# multiply-adds: parity of everything that carries a residual tail, then ResNet-50 b32 against the evidence build
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_call24
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_gemm_family.py tests/test_gpu_pgemm.py tests/test_gpu_parity.py tests/test_gpu_baseline_batches.py tests/test_gpu_glue_int8.py -m gpu -x -q 2>&1 | tail -3
EV=$R/tools/exp/ab/libtengine_amd_r05_evidence.so
AB_LAYERS=1 timeout 900 python tools/exp/ab_lib.py resnet50 32 int8 100 3 evidence=$EV float_carry=product > $O/ab_float_carry_resnet50_b32.txt 2>&1
grep -v "^Tengine" $O/ab_float_carry_resnet50_b32.txt | grep "^==\|^  [a-z]\|elt\|branch1 " | cut -c1-160
