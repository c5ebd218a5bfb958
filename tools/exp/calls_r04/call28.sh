#!/bin/bash
# round 4, call 28: per-pixel first layer with four channels per iteration and a phased epilogue
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_call28
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity_uint8.py tests/test_gpu_baseline_batches.py -q -m gpu --tb=short -p no:cacheprovider -k "first_layer or yolo or mssd or pool" > $O/pytest.txt 2>&1
grep -E "passed|failed|error" $O/pytest.txt | tail -3
grep -E "^FAILED|^ERROR|differ|^E  " $O/pytest.txt | head -30
timeout 300 python tools/profile_layers.py yolov3_tiny 8 20 uint8 2>&1 | grep -v "^Tengine" | awk 'NR<=3 || /sum of/'
timeout 300 python tools/profile_layers.py mssd 16 20 uint8 2>&1 | grep -v "^Tengine" | awk 'NR<=3 || /sum of/'
