#!/bin/bash
# round 6, call 13: dwpw.hip with its per-channel constants in LDS, unconditional loads and run-ahead A fragments (the ISA of the
# old form waited vmcnt(0) in every stage and 16 times in its epilogue): dwpw tests, then old build (tools/exp/ab/base.so) against the
# tree's on MobileNet-v1 int8 b64, interleaved in one box, with the isolated launches side by side
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_call13
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_dwpw.py tests/test_gpu_baseline_batches.py -m gpu -x -q --tb=short 2>&1 | grep -v "^Tengine" | tail -15 > $O/pytest_dwpw.txt; tail -3 $O/pytest_dwpw.txt
AB_LAYERS=1 timeout 1200 python tools/exp/ab_lib.py mobilenet_v1 64 int8 200 3 base=tools/exp/ab/base.so new=product 2>&1 | grep -v "^Tengine" | tee $O/ab_mobilenet_b64.txt
