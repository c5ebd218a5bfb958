#!/bin/bash
# round 6, call 45: the plugin after handing its split to the library: host to host through the unmodified reference library, TAMD_SPLIT_BATCH=0
# against the default, interleaved in one box (the A/B of call 29 again)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_call45
mkdir -p $O
cd $R
export TMPDIR=/tmp
for cfg in "resnet50 32 50 3" "mobilenet_v1 64 50 3" "mobilenet_v1 16 100 3"; do
  timeout 400 python tools/exp/plugin_split_ab.py $cfg 2>&1 | grep -v "^Tengine" | tail -1
done | tee $O/plugin_split_ab_library_pair.txt
