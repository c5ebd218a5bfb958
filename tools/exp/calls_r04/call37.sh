#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_call37
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_dwpw.py -q -m gpu --tb=short -p no:cacheprovider > $O/pytest.txt 2>&1
grep -E "passed|failed|error" $O/pytest.txt | tail -3
grep -E "^FAILED|^ERROR|differ|^E  " $O/pytest.txt | head -10
TAMD_DEBUG=1 timeout 300 python tools/profile_layers.py mobilenet_v1 64 20 int8 2> $O/debug.txt | grep -v "^Tengine" | grep -E "dwpw|sum of"
grep "dwpw" $O/debug.txt | head -4
