#!/bin/bash
# round 5, call 5: the whole GPU suite after the environment pruning + uint8 byte-weight fragments; A/B of the library before / after
# the byte-weight change on the two uint8 configs (fresh processes, interleaved)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_call5
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider -x > $O/pytest_gpu_all.txt 2>&1
grep -E "passed|failed|error" $O/pytest_gpu_all.txt | tail -3
grep -E "^FAILED|^ERROR|^E  " $O/pytest_gpu_all.txt | head -20
OLD=$R/tools/exp/ab/libtengine_amd_before_u8_byte_weights.so
timeout 600 python tools/exp/ab_lib.py yolov3_tiny 8 uint8 100 4 before=$OLD after=product > $O/ab_u8_byte_weights_yolov3_tiny_b8.txt 2>&1
cat $O/ab_u8_byte_weights_yolov3_tiny_b8.txt | grep -v "^Tengine"
timeout 600 python tools/exp/ab_lib.py mssd 16 uint8 100 4 before=$OLD after=product > $O/ab_u8_byte_weights_mssd_b16.txt 2>&1
cat $O/ab_u8_byte_weights_mssd_b16.txt | grep -v "^Tengine"
