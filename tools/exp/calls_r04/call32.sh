#!/bin/bash
# round 4, call 32: conv_u8_c3 (wave-level shallow 3x3) -- parity, YOLOv3-tiny b8 A/B and layer table
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_call32
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_u8_patch.py -q -m gpu --tb=short -p no:cacheprovider -k "shallow_3x3" > $O/pytest.txt 2>&1
grep -E "passed|failed|error" $O/pytest.txt | tail -3
grep -E "^FAILED|^ERROR|differ|^E  " $O/pytest.txt | head -30
timeout 600 python tools/exp/ab_step.py yolov3_tiny 8 uint8 30 3 "without_c3=TAMD_U8_C3=0" "with_c3" 2>&1 | grep -v "^Tengine" | tee $O/ab_c3_yolo_b8.txt
TAMD_DEBUG=1 timeout 300 python tools/profile_layers.py yolov3_tiny 8 20 uint8 2> $O/debug.txt | grep -v "^Tengine" > $O/layers_yolov3_tiny_uint8_b8.txt
awk '{printf "%-24s %-38s %8s\n", $1,$2,$3}' $O/layers_yolov3_tiny_uint8_b8.txt | head -8
grep -E "conv_u8_c3" $O/debug.txt | head -10
