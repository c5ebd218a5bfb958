#!/bin/bash
# round 5, call 11: where the one-binade / bias-start build wins and loses, launch by launch
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_call11
mkdir -p $O
cd $R
export TMPDIR=/tmp
PRE=$R/tools/exp/ab/libtengine_amd_r05_pre_window.so
for cfg in "resnet50 32 int8 100" "mobilenet_v1 64 int8 100" "mobilenet_v1 1 int8 1000"; do
  set -- $cfg
  AB_LAYERS=1 timeout 600 python tools/exp/ab_lib.py $1 $2 $3 $4 2 before=$PRE window=product > $O/ab_window_layers_$1_b$2.txt 2>&1
  grep -v "^Tengine" $O/ab_window_layers_$1_b$2.txt | cut -c1-160
done
