#!/bin/bash
# round 5, call 8: (1) are the ResNet-50 1x1 + eltwise layers VALU-bound?  SQ activity counters per kernel on the b32 plan;
# (2) the uint8 patch kernel with a 4-super-step fragment ring for every configuration (RA = 4: 216 -> ~150 VGPRs) against the product
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_call8
mkdir -p $O
cd /tmp
export TMPDIR=/tmp
export TAMD_PLAN_CACHE=$O/plan_resnet50_int8_b32.txt
python $R/tools/run_model.py resnet50 32 2 int8 > /dev/null 2>&1
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc1 -- python $R/tools/run_model.py resnet50 32 2 int8 > $O/pmc1.log 2>&1
python $R/tools/pmc_summary.py $O/pmc_sq_activity_resnet50_int8_b32.csv $O/pmc1 > /dev/null
rm -rf $O/pmc1
cut -c1-230 $O/pmc_sq_activity_resnet50_int8_b32.csv | grep -v "rocclr\|copy_bytes"
unset TAMD_PLAN_CACHE
cd $R
timeout 600 python tools/exp/ab_lib.py yolov3_tiny 8 uint8 100 3 product=product ra4=$R/tools/exp/ab/libtengine_amd_u8_patch_ra4.so > $O/ab_u8_patch_ra4_yolov3_tiny_b8.txt 2>&1
grep -v "^Tengine" $O/ab_u8_patch_ra4_yolov3_tiny_b8.txt
timeout 600 python tools/exp/ab_lib.py mssd 16 uint8 100 3 product=product ra4=$R/tools/exp/ab/libtengine_amd_u8_patch_ra4.so > $O/ab_u8_patch_ra4_mssd_b16.txt 2>&1
grep -v "^Tengine" $O/ab_u8_patch_ra4_mssd_b16.txt
