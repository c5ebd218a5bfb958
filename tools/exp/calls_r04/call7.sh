#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_call7
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_u8_int.py -q -m gpu -s --tb=short -p no:cacheprovider > $O/pytest.txt 2>&1
grep -E "passed|failed|error" $O/pytest.txt | tail -3
grep -E "^FAILED|^ERROR|differ|^E  " $O/pytest.txt | head -30
grep -E "TEACHER|END TO END" $O/pytest.txt | head -8
for m in "yolov3_tiny 8" "mssd 16"; do
  set -- $m
  TAMD_U8_INT=1 timeout 300 python tools/profile_layers.py $1 $2 20 uint8 2>&1 | grep -v "^Tengine" > $O/layers_${1}_b${2}_int.txt
  echo "$1: $(tail -1 $O/layers_${1}_b${2}_int.txt)"
  sort -k3 -n -r $O/layers_${1}_b${2}_int.txt | head -12
done
