#!/bin/bash
# round 6, call 31: from which batch does the plugin's two-half-batch form pay?  forced split against one graph at batch 2, 4, 8
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_call31
mkdir -p $O
cd $R
export TMPDIR=/tmp
for cfg in "mobilenet_v1 2 200 3 2" "mobilenet_v1 4 200 3 2" "mobilenet_v1 8 200 3 2" "resnet50 2 100 3 2" "resnet50 4 100 3 2" "resnet50 8 100 3 2" "resnet50 16 100 3"; do
  timeout 900 python tools/exp/plugin_split_ab.py $cfg 2>&1 | grep -v "^Tengine" | tail -2
done | tee $O/plugin_split_threshold.txt
