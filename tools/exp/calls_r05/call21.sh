#!/bin/bash
# round 5, call 21: batch 1, launch by launch: the build of call 12 against the product
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_call21
mkdir -p $O
cd $R
export TMPDIR=/tmp
AB_LAYERS=1 timeout 900 python tools/exp/ab_lib.py mobilenet_v1 1 int8 2000 3 call12=$R/tools/exp/ab/libtengine_amd_r05_window_only.so now=product > $O/ab_b1_call12_vs_now.txt 2>&1
grep -v "^Tengine" $O/ab_b1_call12_vs_now.txt | cut -c1-160
cat /tmp/ab_lib_*/now_plan.txt 2>/dev/null | head -5
