#!/bin/bash
# round 5, call 25: the MobileNet-v1 b64 bench line again, now that its PMC traffic summary (profiles/r05_traffic_mobilenet_v1_int8_b64.json) is the
# one of the final plan (the evidence pass of the final code wrote its summaries under the tag r05b, which bench.py's newest-round glob does
# not see); SQ activity counters of the final b64 kernels
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_call25
mkdir -p $O
cd $R
export TMPDIR=/tmp
python bench.py --model mobilenet_v1 --batch 64 --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_mobilenet_v1_int8_b64.json 2> $O/bench_mb64.err
tail -c 1500 $O/bench_mobilenet_v1_int8_b64.json | cut -c1-400
cd /tmp
cp $R/tengine_amd/plans/mobilenet_v1_int8_b64.txt $O/pmc_plan.txt; export TAMD_PLAN_CACHE=$O/pmc_plan.txt
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc1 -- python $R/tools/run_model.py mobilenet_v1 64 2 int8 > $O/pmc1.log 2>&1
python $R/tools/pmc_summary.py $O/pmc_sq_activity_mobilenet_v1_int8_b64_final.csv $O/pmc1 > /dev/null
rm -rf $O/pmc1 $O/pmc_plan.txt
cut -c1-200 $O/pmc_sq_activity_mobilenet_v1_int8_b64_final.csv | grep -v "rocclr\|copy_bytes"
