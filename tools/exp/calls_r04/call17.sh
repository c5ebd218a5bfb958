#!/bin/bash
# round 4, call 17: int8 depthwise kernel v2 (unconditional loads, two output rows per lane) -- parity + MobileNet b64 table
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_call17
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_batches.py tests/test_gpu_edge_cases.py -q -m gpu --tb=short -p no:cacheprovider -k "not uint8 and not yolo and not mssd" > $O/pytest.txt 2>&1
grep -E "passed|failed|error" $O/pytest.txt | tail -3
grep -E "^FAILED|^ERROR|differ|^E  " $O/pytest.txt | head -30
timeout 300 python tools/profile_layers.py mobilenet_v1 64 20 int8 2>&1 | grep -v "^Tengine" > $O/layers_mobilenet_v1_int8_b64.txt
awk '{printf "%-28s %-30s %8s\n", $1,$2,$3}' $O/layers_mobilenet_v1_int8_b64.txt
