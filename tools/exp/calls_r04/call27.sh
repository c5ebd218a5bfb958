#!/bin/bash
# round 4, call 27: uint8 first layers with the main pixels on the matrix cores -- parity, A/B on YOLOv3-tiny b8 and mssd b16
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_call27
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity_uint8.py tests/test_gpu_baseline_batches.py -q -m gpu --tb=short -p no:cacheprovider -k "first_layer or yolo or mssd or pool" > $O/pytest.txt 2>&1
grep -E "passed|failed|error" $O/pytest.txt | tail -3
grep -E "^FAILED|^ERROR|differ|^E  " $O/pytest.txt | head -30
timeout 600 python tools/exp/ab_step.py yolov3_tiny 8 uint8 30 3 "per_pixel_first_layer=TAMD_U8_RGB_MFMA=0" "mfma_first_layer" 2>&1 | grep -v "^Tengine" | tee $O/ab_rgb_mfma_yolo_b8.txt
timeout 600 python tools/exp/ab_step.py mssd 16 uint8 30 3 "per_pixel_first_layer=TAMD_U8_RGB_MFMA=0" "mfma_first_layer" 2>&1 | grep -v "^Tengine" | tee $O/ab_rgb_mfma_mssd_b16.txt
timeout 300 python tools/profile_layers.py yolov3_tiny 8 20 uint8 2>&1 | grep -v "^Tengine" | head -3
timeout 300 python tools/profile_layers.py mssd 16 20 uint8 2>&1 | grep -v "^Tengine" | head -3
