#!/bin/bash
# round 4, call 20: depthwise forms as shipped -- parity by name, MobileNet b64 A/B against round 3's form inside one box
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_call20
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_batches.py tests/test_gpu_pwdw.py -q -m gpu --tb=short -p no:cacheprovider -k "not uint8 and not yolo and not mssd" > $O/pytest.txt 2>&1
grep -E "passed|failed|error" $O/pytest.txt | tail -3
grep -E "^FAILED|^ERROR|differ|^E  " $O/pytest.txt | head -30
timeout 600 python tools/exp/ab_step.py mobilenet_v1 64 int8 50 5 "round3_form=TAMD_DW_FORM=21" "narrow_tall" 2>&1 | grep -v "^Tengine" | tee $O/ab_dw_mobilenet_b64.txt
timeout 300 python tools/profile_layers.py mobilenet_v1 64 20 int8 2>&1 | grep -v "^Tengine" > $O/layers_mobilenet_v1_int8_b64.txt
awk '{printf "%-28s %-30s %8s\n", $1,$2,$3}' $O/layers_mobilenet_v1_int8_b64.txt
