#!/bin/bash
# round 5, call 23: the depthwise tail's pairing with its consumer (dwpw) tried before the pwdw pairing: MobileNet-v1 b64 / b1 against the evidence build
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_call23
mkdir -p $O
cd $R
export TMPDIR=/tmp
EV=$R/tools/exp/ab/libtengine_amd_r05_evidence.so
AB_LAYERS=1 timeout 900 python tools/exp/ab_lib.py mobilenet_v1 64 int8 100 3 evidence=$EV dwpw_first=product > $O/ab_dwpw_first_mobilenet_v1_b64.txt 2>&1
grep -v "^Tengine" $O/ab_dwpw_first_mobilenet_v1_b64.txt | cut -c1-160
timeout 900 python tools/exp/ab_lib.py mobilenet_v1 1 int8 2000 3 evidence=$EV dwpw_first=product > $O/ab_dwpw_first_mobilenet_v1_b1.txt 2>&1
grep -v "^Tengine" $O/ab_dwpw_first_mobilenet_v1_b1.txt | cut -c1-160
timeout 600 python -m pytest tests/test_gpu_dwpw.py tests/test_gpu_pwdw.py tests/test_gpu_baseline_batches.py -m gpu -x -q 2>&1 | tail -3
