#!/bin/bash
# round 6, call 37: split tests after the timestamp warm-up fix (the warm-up pass completes before the stamped passes are written), the
# direct-path tests next to it
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_call37
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_split_batch.py tests/test_gpu_direct.py -m gpu -q --tb=short -s 2>&1 | grep -v "^Tengine" > $O/pytest_split_direct.txt; grep "direct path\|passed\|failed" $O/pytest_split_direct.txt | tail; grep -B2 -A12 "^E " $O/pytest_split_direct.txt | head -60 | cut -c1-300
for cfg in "mobilenet_v1 1 int8 300" "mobilenet_v1 64 int8 100"; do
  set -- $cfg
  python tools/direct_timestamps.py $1 $2 $3 $4 2>&1 | grep -v "^Tengine" | tail -2
done
