#!/bin/bash
# round 5, call 18: batch 1: which of {one-binade requantisation, accumulators from the bias} costs the pwdw launches their 0.5 %?
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_call18
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python tools/exp/ab_lib.py mobilenet_v1 1 int8 2000 3 before=$R/tools/exp/ab/libtengine_amd_r05_pre_window.so nowin=$R/tools/exp/ab/libtengine_amd_pwdw_nowin.so now=product > $O/ab_b1_window_or_bias.txt 2>&1
grep -v "^Tengine" $O/ab_b1_window_or_bias.txt | cut -c1-160
