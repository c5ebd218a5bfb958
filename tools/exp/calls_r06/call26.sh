#!/bin/bash
# round 6, call 26: a per-GPU batch as two concurrent half-batch graphs, each on its own HSA queue (direct dispatch) / HIP stream (replay)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_call26
mkdir -p $O
cd $R
export TMPDIR=/tmp
for d in 1 0; do
  for cfg in "resnet50 32 int8 100" "mobilenet_v1 64 int8 200" "yolov3_tiny 8 uint8 50" "mssd 16 uint8 50"; do
    timeout 600 python tools/exp/split_batch_direct.py $cfg $d 2>&1 | grep -v "^Tengine" | tail -3
  done
done | tee $O/split_batch_direct.txt
