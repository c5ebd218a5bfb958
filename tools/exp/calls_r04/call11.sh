#!/bin/bash
# round 4, call 11: fused stem v2 (batched staging loads, packed pooling) -- parity, A/B on ResNet-50 b32 inside one box
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_call11
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_stem.py -q -m gpu --tb=short -p no:cacheprovider > $O/pytest_stem.txt 2>&1
grep -E "passed|failed|error" $O/pytest_stem.txt | tail -3
grep -E "^FAILED|^ERROR|differ|^E  " $O/pytest_stem.txt | head -30
timeout 600 python tools/exp/ab_step.py resnet50 32 int8 30 5 "two_launches=TAMD_FIRST_POOL=0" "fused_stem" 2>&1 | grep -v "^Tengine" | tee $O/ab_stem_resnet50_b32.txt
timeout 300 python tools/profile_layers.py resnet50 32 20 int8 2>&1 | grep -v "^Tengine" > $O/layers_resnet50_int8_b32.txt
head -4 $O/layers_resnet50_int8_b32.txt; tail -1 $O/layers_resnet50_int8_b32.txt
