#!/bin/bash
# round 5, call 19: batch 1 on ONE box: the build before the requantisation work, the build of call 12 (one-binade + bias start, before the
# slice work) and the product (slices; PwDwArgs::sl moved behind the fields the one-slice kernels read)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_call19
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python tools/exp/ab_lib.py mobilenet_v1 1 int8 2000 3 before=$R/tools/exp/ab/libtengine_amd_r05_pre_window.so call12=$R/tools/exp/ab/libtengine_amd_r05_window_only.so now=product > $O/ab_b1_three_builds.txt 2>&1
grep -v "^Tengine" $O/ab_b1_three_builds.txt | cut -c1-160
