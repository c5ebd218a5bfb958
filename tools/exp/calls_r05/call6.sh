#!/bin/bash
# round 5, call 6: uint8 byte-weight fragments with the dequantisation one MFMA step ahead: A/B against the library before the change,
# uint8 parity suites, the environment-switch tests
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_call6
mkdir -p $O
cd $R
export TMPDIR=/tmp
OLD=$R/tools/exp/ab/libtengine_amd_before_u8_byte_weights.so
timeout 600 python tools/exp/ab_lib.py yolov3_tiny 8 uint8 100 3 before=$OLD after=product > $O/ab_u8_byte_weights_yolov3_tiny_b8.txt 2>&1
cat $O/ab_u8_byte_weights_yolov3_tiny_b8.txt | grep -v "^Tengine"
timeout 600 python tools/exp/ab_lib.py mssd 16 uint8 100 3 before=$OLD after=product > $O/ab_u8_byte_weights_mssd_b16.txt 2>&1
cat $O/ab_u8_byte_weights_mssd_b16.txt | grep -v "^Tengine"
timeout 1200 python -m pytest tests/test_gpu_env_switches.py tests/test_gpu_u8_patch.py tests/test_gpu_u8_lanes.py tests/test_gpu_parity_uint8.py tests/test_gpu_baseline_batches.py -q -m gpu --tb=short -p no:cacheprovider > $O/pytest_u8.txt 2>&1
grep -E "passed|failed|error" $O/pytest_u8.txt | tail -3
grep -E "^FAILED|^ERROR|^E  " $O/pytest_u8.txt | head -20
