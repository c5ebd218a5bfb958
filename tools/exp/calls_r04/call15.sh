#!/bin/bash
# round 4, call 15: uint8 suites after the lanes kernel
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_call15
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity_uint8.py tests/test_gpu_u8_patch.py tests/test_gpu_u8_int.py tests/test_gpu_u8_lanes.py tests/test_gpu_baseline_batches.py tests/test_gpu_stem.py tests/test_plugin_dropin.py -q -m gpu --tb=short -p no:cacheprovider > $O/pytest.txt 2>&1
grep -E "passed|failed|error" $O/pytest.txt | tail -3
grep -E "^FAILED|^ERROR|differ|^E  " $O/pytest.txt | head -30
