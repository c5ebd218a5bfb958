#!/bin/bash
# round 6, call 14: (a) tools/exp/dwpw_anatomy.bin: stage stamps + ablations of the new dwpw; (b) igemm small tiles with multipliers and
# residual operand requested at kernel top: old build against the tree's on ResNet-50 int8 b32 (isolated launches side by side);
# (c) the whole GPU suite on the tree's build
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_call14
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 120 tools/exp/dwpw_anatomy.bin 2>&1 | tee $O/dwpw_anatomy.txt
AB_LAYERS=1 timeout 1500 python tools/exp/ab_lib.py resnet50 32 int8 100 3 base=tools/exp/ab/base.so new=product 2>&1 | grep -v "^Tengine" | tee $O/ab_resnet50_b32.txt
timeout 1500 python -m pytest tests -m gpu -q --tb=short -x 2>&1 | grep -v "^Tengine" | tail -15 > $O/pytest_gpu_all.txt; tail -3 $O/pytest_gpu_all.txt
