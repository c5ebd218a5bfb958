#!/bin/bash
# round 5, call 7: the whole GPU suite (TAMD_PIN value pointers fixed: a second tamd_pin() call used to overwrite the first one's
# value), then the device fuzzers over the pinned forms
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_call7
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider > $O/pytest_gpu_all.txt 2>&1
grep -E "passed|failed|error" $O/pytest_gpu_all.txt | tail -3
grep -E "^FAILED|^ERROR" $O/pytest_gpu_all.txt | head -30
timeout 200 python tools/fuzz_device.py --dtype uint8 --seconds 60 --seed 5 > $O/fuzz_device_uint8.txt 2>&1; tail -3 $O/fuzz_device_uint8.txt
timeout 200 python tools/fuzz_device.py --dtype int8 --seconds 60 --seed 5 > $O/fuzz_device_int8.txt 2>&1; tail -3 $O/fuzz_device_int8.txt
