#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_call38
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_dwpw.py tests/test_gpu_baseline_batches.py tests/test_gpu_pwdw.py -q -m gpu --tb=short -p no:cacheprovider -k "not uint8 and not yolo and not mssd and not resnet" > $O/pytest.txt 2>&1
grep -E "passed|failed|error" $O/pytest.txt | tail -3
grep -E "^FAILED|^ERROR|differ|^E  " $O/pytest.txt | head -10
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
