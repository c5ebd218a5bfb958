#!/bin/bash
# round 4, call 35: dwpw (depthwise -> pointwise in one launch) -- parity, MobileNet b64 A/B and table
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_call35
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_dwpw.py -q -m gpu --tb=short -p no:cacheprovider > $O/pytest.txt 2>&1
grep -E "passed|failed|error" $O/pytest.txt | tail -3
grep -E "^FAILED|^ERROR|differ|^E  " $O/pytest.txt | head -20
timeout 600 python tools/exp/ab_step.py mobilenet_v1 64 int8 50 3 "two_launches=TAMD_FUSE_DWPW=0" "dwpw" 2>&1 | grep -v "^Tengine" | tee $O/ab_dwpw_mobilenet_b64.txt
TAMD_DEBUG=1 timeout 300 python tools/profile_layers.py mobilenet_v1 64 20 int8 2> $O/debug.txt | grep -v "^Tengine" > $O/layers_mobilenet_v1_int8_b64.txt
awk '{printf "%-30s %-30s %8s\n", $1,$2,$3}' $O/layers_mobilenet_v1_int8_b64.txt | sed -n 8,24p
grep "dwpw" $O/debug.txt | head -8
