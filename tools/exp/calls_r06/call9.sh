#!/bin/bash
# round 6, call 9: the directly dispatched pass under the HSA runtime's dispatch profiling (new), all five configs; its test; the ABI tests
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_call9
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_direct.py tests/test_abi.py -m gpu -q --tb=short -s 2>&1 | grep -v "^Tengine" | grep "direct path\|passed\|failed\|Error" | tail -8
for cfg in "mobilenet_v1 1 int8 500" "mobilenet_v1 64 int8 100" "resnet50 32 int8 100" "yolov3_tiny 8 uint8 50" "mssd 16 uint8 50"; do
  set -- $cfg
  python tools/direct_timestamps.py $1 $2 $3 $4 2>&1 | grep -v "^Tengine" > $O/direct_path_timestamps_$1_$3_b$2.txt
  head -1 $O/direct_path_timestamps_$1_$3_b$2.txt; tail -2 $O/direct_path_timestamps_$1_$3_b$2.txt
done
cat $O/direct_path_timestamps_mobilenet_v1_int8_b1.txt
