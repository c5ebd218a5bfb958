#!/bin/bash
# round 5, call 15: batch 1 with the two-slice blocks offered to big grids only (three fresh processes per build)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_call15
mkdir -p $O
cd $R
export TMPDIR=/tmp
PRE=$R/tools/exp/ab/libtengine_amd_r05_pre_window.so
AB_LAYERS=1 timeout 900 python tools/exp/ab_lib.py mobilenet_v1 1 int8 2000 3 before=$PRE now=product > $O/ab_slices_layers_mobilenet_v1_b1.txt 2>&1
grep -v "^Tengine" $O/ab_slices_layers_mobilenet_v1_b1.txt | cut -c1-160
