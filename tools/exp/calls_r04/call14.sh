#!/bin/bash
# round 4, call 14: lane-level chains for the small uint8 layers -- parity, A/B on mssd b16 inside one box, layer table
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_call14
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_u8_lanes.py -q -m gpu --tb=short -p no:cacheprovider > $O/pytest_lanes.txt 2>&1
grep -E "passed|failed|error" $O/pytest_lanes.txt | tail -3
grep -E "^FAILED|^ERROR|differ|^E  " $O/pytest_lanes.txt | head -30
timeout 600 python tools/exp/ab_step.py mssd 16 uint8 30 3 "gemm_family_for_small_layers=TAMD_U8_LANES=0" "lane_chains" 2>&1 | grep -v "^Tengine" | tee $O/ab_lanes_mssd_b16.txt
timeout 300 python tools/profile_layers.py mssd 16 20 uint8 2>&1 | grep -v "^Tengine" > $O/layers_mssd_uint8_b16.txt
awk '{printf "%-24s %-30s %8s\n", $1,$2,$3}' $O/layers_mssd_uint8_b16.txt | tail -26
timeout 600 python tools/exp/ab_step.py mssd 16 uint8 30 3 "int_gemm_family_for_small_layers=TAMD_U8_LANES=0,TAMD_U8_INT=1" "int_lane_chains=TAMD_U8_INT=1" 2>&1 | grep -v "^Tengine" | tee $O/ab_lanes_mssd_b16_int.txt
timeout 600 python tools/exp/ab_step.py yolov3_tiny 8 uint8 30 3 "gemm_family_for_small_layers=TAMD_U8_LANES=0" "lane_chains" 2>&1 | grep -v "^Tengine" | tee $O/ab_lanes_yolo_b8.txt
