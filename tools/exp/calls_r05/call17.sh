#!/bin/bash
# round 5, call 17: the slice instances in a code object of their own: batch 1 again (three fresh processes per build), b64 for the record
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_call17
mkdir -p $O
cd $R
export TMPDIR=/tmp
PRE=$R/tools/exp/ab/libtengine_amd_r05_pre_window.so
AB_LAYERS=1 timeout 900 python tools/exp/ab_lib.py mobilenet_v1 1 int8 2000 3 before=$PRE now=product > $O/ab_final_layers_mobilenet_v1_b1.txt 2>&1
grep -v "^Tengine" $O/ab_final_layers_mobilenet_v1_b1.txt | cut -c1-160
AB_LAYERS=1 timeout 900 python tools/exp/ab_lib.py mobilenet_v1 64 int8 100 2 before=$PRE now=product > $O/ab_final_layers_mobilenet_v1_b64.txt 2>&1
grep -v "^Tengine" $O/ab_final_layers_mobilenet_v1_b64.txt | cut -c1-160 | head -4
