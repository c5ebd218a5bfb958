#!/bin/bash
# round 5, call 2 (call 1 again: its anatomy run faulted on a harness flag clash, its A/B compared the same code twice): conv_pgemm_w.hip (eight-wave blocks, table-driven set-up, zero areas, bias in the accumulators, ks2 through an LDS tile):
# parity of every new member, block anatomy on the four ResNet-50 3x3 shapes (old and new kernels, phase-skew flags), and the
# ResNet-50 b32 step old vs new as an interleaved A/B with separate plan files
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_call2
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_pgemm.py -q -x --tb=short -p no:cacheprovider > $O/pytest_pgemm.txt 2>&1
tail -15 $O/pytest_pgemm.txt
timeout 300 tools/exp/pgemm_anatomy.bin 32 3x3 > $O/pgemm_anatomy_3x3.txt 2>&1
cut -c1-260 $O/pgemm_anatomy_3x3.txt
timeout 600 python tools/exp/ab_step.py resnet50 32 int8 40 7 old=TAMD_PGEMM_W=0,TAMD_PLAN_CACHE=/tmp/plan_old.txt new=TAMD_PLAN_CACHE=/tmp/plan_new.txt > $O/ab_pgemm_w_resnet50_b32.txt 2>&1
grep -v "^Tengine" $O/ab_pgemm_w_resnet50_b32.txt | tail -5
cp /tmp/plan_new.txt $O/plan_new.txt; cp /tmp/plan_old.txt $O/plan_old.txt
TAMD_PLAN_CACHE=/tmp/plan_new.txt timeout 300 python tools/profile_layers.py resnet50 32 20 int8 2>&1 | grep -v "^Tengine" > $O/layers_resnet50_int8_b32_new.txt
grep "3x3\|branch2b\|sum of" $O/layers_resnet50_int8_b32_new.txt
timeout 300 python -m pytest tests/test_gpu_direct.py -q -x --tb=short -p no:cacheprovider -k "zero_copy or same_bytes" 2>&1 | tail -5
