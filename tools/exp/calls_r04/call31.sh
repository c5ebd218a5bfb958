#!/bin/bash
# round 4, call 31: 2-D pixel tiles in the uint8 patch kernel -- parity, YOLOv3-tiny b8 A/B and layer table
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_call31
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_u8_patch.py -q -m gpu --tb=short -p no:cacheprovider > $O/pytest.txt 2>&1
grep -E "passed|failed|error" $O/pytest.txt | tail -3
grep -E "^FAILED|^ERROR|differ|^E  " $O/pytest.txt | head -30
timeout 600 python tools/exp/ab_step.py yolov3_tiny 8 uint8 30 3 "one_d_tiles=TAMD_U8_PATCH_2D=0" "two_d_tiles" 2>&1 | grep -v "^Tengine" | tee $O/ab_2d_yolo_b8.txt
TAMD_DEBUG=1 timeout 300 python tools/profile_layers.py yolov3_tiny 8 20 uint8 2> $O/debug.txt | grep -v "^Tengine" > $O/layers_yolov3_tiny_uint8_b8.txt
awk '{printf "%-24s %-38s %8s\n", $1,$2,$3}' $O/layers_yolov3_tiny_uint8_b8.txt | head -8
grep -E "conv1:|conv2:|conv3:" $O/debug.txt | head -30
