#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_call41
mkdir -p $O
cd $R
export TMPDIR=/tmp
python -c "from tengine_amd import plans; print(plans.seed('/tmp/ab_plan.txt', 'resnet50', 'int8', 32))"
TAMD_PLAN_CACHE=/tmp/ab_plan.txt timeout 45 python tools/exp/ab_step.py resnet50 32 int8 40 7 blocks2048 blocks1024=TAMD_PW_STREAM_BLOCKS=1024 blocks512=TAMD_PW_STREAM_BLOCKS=512 blocks256=TAMD_PW_STREAM_BLOCKS=256 blocks4096=TAMD_PW_STREAM_BLOCKS=4096 > $O/ab_pw_stream_blocks_resnet50_b32.txt 2>&1
cat $O/ab_pw_stream_blocks_resnet50_b32.txt | tail -8
