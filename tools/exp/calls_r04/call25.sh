#!/bin/bash
# round 4, call 25: byte tables for the fused ReLU / pool nodes + wave-level hand-over in every byte-exact epilogue
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_call25
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity_uint8.py tests/test_gpu_u8_patch.py tests/test_gpu_u8_lanes.py -q -m gpu --tb=short -p no:cacheprovider -x > $O/pytest.txt 2>&1
grep -E "passed|failed|error" $O/pytest.txt | tail -3
grep -E "^FAILED|^ERROR|differ|^E  " $O/pytest.txt | head -30
timeout 300 python tools/profile_layers.py yolov3_tiny 8 20 uint8 2>&1 | grep -v "^Tengine" > $O/layers_yolov3_tiny_uint8_b8.txt
awk '{printf "%-24s %-34s %8s\n", $1,$2,$3}' $O/layers_yolov3_tiny_uint8_b8.txt
timeout 300 python tools/profile_layers.py mssd 16 20 uint8 2>&1 | grep -v "^Tengine" > $O/layers_mssd_uint8_b16.txt
tail -1 $O/layers_mssd_uint8_b16.txt
