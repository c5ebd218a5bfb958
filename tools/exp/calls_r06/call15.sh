#!/bin/bash
# round 6, call 15: dwpw step 2 -- taps addressed as scalar row pointer + per-lane offset (no vector address arithmetic per load), and the two
# waves of a SIMD running the stage's two phases in opposite orders (DWPW_FLIP): anatomy of both, dwpw tests, three builds on MobileNet-v1 b64
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_call15
mkdir -p $O
cd $R
export TMPDIR=/tmp
for f in 0 1; do echo "== DWPW_FLIP=$f"; timeout 120 tools/exp/dwpw_anatomy_flip$f.bin; done 2>&1 | tee $O/dwpw_anatomy.txt
timeout 900 python -m pytest tests/test_gpu_dwpw.py tests/test_gpu_baseline_batches.py -m gpu -x -q --tb=short 2>&1 | grep -v "^Tengine" | tail -15 > $O/pytest_dwpw.txt; tail -3 $O/pytest_dwpw.txt
AB_LAYERS=1 timeout 1500 python tools/exp/ab_lib.py mobilenet_v1 64 int8 200 3 base=tools/exp/ab/base.so noflip=tools/exp/ab/noflip.so 2>&1 | grep -v "^Tengine" | tee $O/ab_mobilenet_b64_noflip.txt | head -4
AB_LAYERS=1 timeout 1500 python tools/exp/ab_lib.py mobilenet_v1 64 int8 200 3 noflip=tools/exp/ab/noflip.so flip=product 2>&1 | grep -v "^Tengine" | tee $O/ab_mobilenet_b64_flip.txt
