#!/bin/bash
# round 4, call 22: uint8 depthwise with the wave-level requantisation -- parity + mssd layer table
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_call22
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity_uint8.py tests/test_gpu_baseline_batches.py -q -m gpu --tb=short -p no:cacheprovider -k "uint8 or mssd or yolo or u8" > $O/pytest.txt 2>&1
grep -E "passed|failed|error" $O/pytest.txt | tail -3
grep -E "^FAILED|^ERROR|differ|^E  " $O/pytest.txt | head -30
timeout 300 python tools/profile_layers.py mssd 16 20 uint8 2>&1 | grep -v "^Tengine" > $O/layers_mssd_uint8_b16.txt
grep -E "/dw |sum of" $O/layers_mssd_uint8_b16.txt | awk '{printf "%-24s %-30s %8s\n", $1,$2,$3}'
