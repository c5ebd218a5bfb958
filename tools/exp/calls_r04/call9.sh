#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_call9
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_u8_int.py -q -m gpu --tb=short -p no:cacheprovider > $O/pytest.txt 2>&1
grep -E "passed|failed|error" $O/pytest.txt | tail -3
grep -E "^FAILED|^ERROR|differ|^E  " $O/pytest.txt | head -30
for m in "yolov3_tiny 8" "mssd 16"; do
  set -- $m
  for ab in 0 3; do
    TAMD_U8I_ABLATE=$ab TAMD_U8_INT=1 timeout 300 python tools/profile_layers.py $1 $2 20 uint8 2>&1 | grep -v "^Tengine" > $O/layers_${1}_b${2}_int_ablate$ab.txt
    echo "$1 ablate=$ab: $(tail -1 $O/layers_${1}_b${2}_int_ablate$ab.txt)"
  done
done
paste <(awk '{print $1, $2, $3}' $O/layers_mssd_b16_int_ablate0.txt) <(awk '{print $3}' $O/layers_mssd_b16_int_ablate3.txt) | sort -k3 -n -r | head -26
paste <(awk '{print $1, $2, $3}' $O/layers_yolov3_tiny_b8_int_ablate0.txt) <(awk '{print $3}' $O/layers_yolov3_tiny_b8_int_ablate3.txt) | sort -k3 -n -r | head -14
timeout 600 python tools/exp/ab_step.py yolov3_tiny 8 uint8 30 3 "byte_exact" "integer=TAMD_U8_INT=1" 2>&1 | grep -v "^Tengine" | tee -a $O/ab_u8_int.txt
timeout 600 python tools/exp/ab_step.py mssd 16 uint8 30 3 "byte_exact" "integer=TAMD_U8_INT=1" 2>&1 | grep -v "^Tengine" | tee -a $O/ab_u8_int.txt
