#!/bin/bash
# round 4, call 30: the whole GPU suite on the final code, then the smoke entry
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_call30
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider > $O/pytest_gpu_all.txt 2>&1
grep -E "passed|failed|error" $O/pytest_gpu_all.txt | tail -3
grep -E "^FAILED|^ERROR|differ|^E  " $O/pytest_gpu_all.txt | head -30
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
