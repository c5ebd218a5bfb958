#!/bin/bash
# round 4, GPU call 5: integer path with the first-layer kernel; anatomy of the short-K layers by ablation
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_call5
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_u8_int.py tests/test_gpu_direct.py -q -m gpu -s --tb=short -p no:cacheprovider > $O/pytest.txt 2>&1
grep -E "passed|failed|error" $O/pytest.txt | tail -3
grep -E "^FAILED|^ERROR|differ|^E  " $O/pytest.txt | head -30
grep -E "TEACHER|END TO END" $O/pytest.txt | head
for m in "yolov3_tiny 8" "mssd 16"; do
  set -- $m
  for ab in 0 1 2 3 4 7; do
    TAMD_U8I_ABLATE=$ab TAMD_U8_INT=1 timeout 300 python tools/profile_layers.py $1 $2 20 uint8 2>&1 | grep -v "^Tengine" > $O/layers_${1}_b${2}_int_ablate$ab.txt
    echo "$1 ablate=$ab: $(tail -1 $O/layers_${1}_b${2}_int_ablate$ab.txt)"
  done
done
paste <(awk '{print $1, $2, $3}' $O/layers_mssd_b16_int_ablate0.txt) <(awk '{print $3}' $O/layers_mssd_b16_int_ablate1.txt) <(awk '{print $3}' $O/layers_mssd_b16_int_ablate2.txt) <(awk '{print $3}' $O/layers_mssd_b16_int_ablate3.txt) <(awk '{print $3}' $O/layers_mssd_b16_int_ablate4.txt) <(awk '{print $3}' $O/layers_mssd_b16_int_ablate7.txt) | grep u8i | sort -k3 -n -r | head -14
paste <(awk '{print $1, $2, $3}' $O/layers_yolov3_tiny_b8_int_ablate0.txt) <(awk '{print $3}' $O/layers_yolov3_tiny_b8_int_ablate1.txt) <(awk '{print $3}' $O/layers_yolov3_tiny_b8_int_ablate2.txt) <(awk '{print $3}' $O/layers_yolov3_tiny_b8_int_ablate3.txt) <(awk '{print $3}' $O/layers_yolov3_tiny_b8_int_ablate4.txt) <(awk '{print $3}' $O/layers_yolov3_tiny_b8_int_ablate7.txt) | sort -k3 -n -r | head -14
