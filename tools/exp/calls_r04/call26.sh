#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_call26
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_u8_patch.py -q -m gpu --tb=short -p no:cacheprovider > $O/pytest.txt 2>&1
grep -E "passed|failed|error" $O/pytest.txt | tail -3
grep -E "^FAILED|^ERROR|differ|^E  " $O/pytest.txt | head -30
TAMD_DEBUG=1 timeout 300 python tools/profile_layers.py yolov3_tiny 8 20 uint8 2> $O/debug.txt | grep -v "^Tengine" > $O/layers_yolov3_tiny_uint8_b8.txt
awk '{printf "%-24s %-34s %8s\n", $1,$2,$3}' $O/layers_yolov3_tiny_uint8_b8.txt | head -8
grep -E "conv1:|conv2:|conv0:" $O/debug.txt | head -20
