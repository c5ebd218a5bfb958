#!/bin/bash
# round 5, call 13: is MobileNet-v1 b64 VALU-bound launch by launch?  SQ activity counters per kernel
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_call13
mkdir -p $O
cd /tmp
export TMPDIR=/tmp
export TAMD_PLAN_CACHE=$O/plan_mobilenet_v1_int8_b64.txt
python $R/tools/run_model.py mobilenet_v1 64 2 int8 > /dev/null 2>&1
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc1 -- python $R/tools/run_model.py mobilenet_v1 64 2 int8 > $O/pmc1.log 2>&1
python $R/tools/pmc_summary.py $O/pmc_sq_activity_mobilenet_v1_int8_b64.csv $O/pmc1 > /dev/null
timeout 400 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_MFMA --output-format csv -d $O/pmc2 -- python $R/tools/run_model.py mobilenet_v1 64 2 int8 > $O/pmc2.log 2>&1
python $R/tools/pmc_summary.py $O/pmc_sq_mix_mobilenet_v1_int8_b64.csv $O/pmc2 > /dev/null
rm -rf $O/pmc1 $O/pmc2
cut -c1-230 $O/pmc_sq_activity_mobilenet_v1_int8_b64.csv | grep -v "rocclr\|copy_bytes"
cut -c1-230 $O/pmc_sq_mix_mobilenet_v1_int8_b64.csv | grep -v "rocclr\|copy_bytes"
cd $R; TAMD_PLAN_CACHE=$O/plan_mobilenet_v1_int8_b64.txt python tools/profile_layers.py mobilenet_v1 64 20 int8 2>&1 | grep -v "^Tengine" > $O/layers_mobilenet_v1_int8_b64.txt; cat $O/layers_mobilenet_v1_int8_b64.txt | cut -c1-120
