#!/bin/bash
# round 4, call 13: all-tail layers (maps of < 8 pixels) on conv_u8_patch_tail -- parity, A/B on mssd b16 inside one box
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_call13
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity_uint8.py tests/test_gpu_u8_patch.py -q -m gpu --tb=short -p no:cacheprovider -x > $O/pytest_u8.txt 2>&1
grep -E "passed|failed|error" $O/pytest_u8.txt | tail -3
grep -E "^FAILED|^ERROR|differ|^E  " $O/pytest_u8.txt | head -30
timeout 600 python tools/exp/ab_step.py mssd 16 uint8 30 3 "gemm_family_for_tiny_maps=TAMD_U8_PATCH_TINY=0" "patch_tail" 2>&1 | grep -v "^Tengine" | tee $O/ab_tiny_mssd_b16.txt
timeout 300 python tools/profile_layers.py mssd 16 20 uint8 2>&1 | grep -v "^Tengine" > $O/layers_mssd_uint8_b16.txt
awk '{printf "%-24s %-30s %8s\n", $1,$2,$3}' $O/layers_mssd_uint8_b16.txt | tail -26
timeout 600 python -m pytest tests/test_gpu_baseline_batches.py -q -m gpu -k "mssd" --tb=short -p no:cacheprovider 2>&1 | tail -3
