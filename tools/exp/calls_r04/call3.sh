#!/bin/bash
# round 4, GPU call 3: the integer uint8 path after the restructure (2-D tiles, channel groups per chunk, 8-deep weight ring)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_call3
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_u8_int.py -q -m gpu -s --tb=short -p no:cacheprovider > $O/pytest_u8_int.txt 2>&1
grep -E "passed|failed|error" $O/pytest_u8_int.txt | tail -3
grep -E "^FAILED|^ERROR|differ|^E  " $O/pytest_u8_int.txt | head -40
grep -E "TEACHER|END TO END" $O/pytest_u8_int.txt | head -40
for cfg in "yolov3_tiny 8 uint8 30" "mssd 16 uint8 30"; do
  set -- $cfg
  timeout 600 python tools/exp/ab_step.py $1 $2 $3 $4 3 "byte_exact" "integer=TAMD_U8_INT=1" "integer_cg1=TAMD_U8_INT=1,TAMD_U8I_CG=1" 2>&1 | grep -v "^Tengine" | tee -a $O/ab_u8_int.txt
done
for m in "yolov3_tiny 8" "mssd 16"; do
  set -- $m
  TAMD_U8_INT=1 timeout 300 python tools/profile_layers.py $1 $2 20 uint8 2>&1 | grep -v "^Tengine" > $O/layers_${1}_uint8_b${2}_int.txt
  tail -1 $O/layers_${1}_uint8_b${2}_int.txt
done
sort -k3 -n -r $O/layers_yolov3_tiny_uint8_b8_int.txt | head -14
sort -k3 -n -r $O/layers_mssd_uint8_b16_int.txt | head -24
