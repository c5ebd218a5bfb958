#!/bin/bash
# round 5, call 26: the first-layer pair (PROD 1) with the next tile's patch gather in flight under this tile's MFMAs (the second operand buffer the
# pointwise producers already have): parity of the first-layer cases, then batch 1 / batch 64 against the evidence build
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_call26
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_pwdw.py tests/test_gpu_baseline_batches.py -m gpu -x -q 2>&1 | tail -3
EV=$R/tools/exp/ab/libtengine_amd_r05_final_evidence.so
AB_LAYERS=1 timeout 900 python tools/exp/ab_lib.py mobilenet_v1 1 int8 2000 3 evidence=$EV pingpong=product > $O/ab_firstdw_pingpong_mobilenet_v1_b1.txt 2>&1
grep -v "^Tengine" $O/ab_firstdw_pingpong_mobilenet_v1_b1.txt | head -5 | cut -c1-160
AB_LAYERS=1 timeout 900 python tools/exp/ab_lib.py mobilenet_v1 64 int8 100 3 evidence=$EV pingpong=product > $O/ab_firstdw_pingpong_mobilenet_v1_b64.txt 2>&1
grep -v "^Tengine" $O/ab_firstdw_pingpong_mobilenet_v1_b64.txt | head -5 | cut -c1-160
