#!/bin/bash
# Runs ON THE GPU BOX (gpurun): the judged evidence of round 6.  usage: tools/collect_evidence_r06.sh [tag]
#  0. plans: tools/make_plans.py -> tengine_amd/plans/ (copied back through gpurun_out/<tag>/plans/; each file now records every
#     candidate plan's step time), every later pass runs on them
#  1. PMC passes FIRST (separate runs, counters only): FETCH_SIZE / WRITE_SIZE traffic + MFMA busy per kernel, all five configs; the
#     VALU / wait counters (tools/pmc_model.sh) for the two batched int8 configs
#  2. bench lines: the driver's invocation (headline + `configs`: all five BASELINE configs in one process), a long headline, and one
#     line per side config (its own CPU-baseline-free run)
#  3. rocprofv3 --kernel-trace --stats of EVERY config's bench run (round 5 had this for batch 1 only)
#  4. per-launch tables (HIP events) of the configs; ResNet-50 inside a pass (kernel trace) against its isolated table
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O $O/plans
cd /tmp; export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o $R/tools/exp/hbm_calib.bin $R/tools/exp/hbm_calib.hip 2>&1 | tail -2
# ---- 0. plans
python $R/tools/make_plans.py $O/plans 2>&1 | grep -v "^Tengine" | tee $O/plans.txt
mkdir -p $R/tengine_amd/plans; cp $O/plans/*.txt $R/tengine_amd/plans/
plan_of() { echo $R/tengine_amd/plans/$1_$3_b$2.txt; }
# ---- 1. PMC
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/calib_$c -- $R/tools/exp/hbm_calib.bin > $O/calib_$c.log 2>&1
done
for cfg in "mobilenet_v1 1 int8" "resnet50 32 int8" "mobilenet_v1 64 int8" "yolov3_tiny 8 uint8" "mssd 16 uint8"; do
  set -- $cfg
  grep -v "^# " $(plan_of $1 $2 $3) > $O/pmc_plan.txt; export TAMD_PLAN_CACHE=$O/pmc_plan.txt
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $O/m_$c -- python $R/tools/run_model.py $1 $2 5 $3 > $O/m_$c.log 2>&1
  done
  K=$(grep -o "launches_per_run [0-9]*" $O/m_FETCH_SIZE.log | cut -d' ' -f2)
  python $R/tools/traffic_summary.py $O/traffic_$1_$3_b$2.json $O/calib_FETCH_SIZE $O/calib_WRITE_SIZE $O/m_FETCH_SIZE $O/m_WRITE_SIZE $((K * 5)) > $O/traffic_$1_$3_b$2.txt 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $O/m_mfma -- python $R/tools/run_model.py $1 $2 5 $3 > $O/m_mfma.log 2>&1
  python $R/tools/pmc_summary.py $O/pmc_mfma_$1_$3_b$2.csv $O/m_mfma > /dev/null 2>&1
  rm -rf $O/m_FETCH_SIZE $O/m_WRITE_SIZE $O/m_mfma
  unset TAMD_PLAN_CACHE
  cp $O/traffic_$1_$3_b$2.json $R/profiles/${TAG}_traffic_$1_$3_b$2.json
  echo "$cfg: $(wc -l < $O/traffic_$1_$3_b$2.txt) traffic lines, $(wc -l < $O/pmc_mfma_$1_$3_b$2.csv) pmc lines"
done
for cfg in "resnet50 32" "mobilenet_v1 64"; do        # VALU / wait / issue counters (three passes each)
  set -- $cfg
  grep -v "^# " $(plan_of $1 $2 int8) > $O/pmc_plan.txt; export TAMD_PLAN_CACHE=$O/pmc_plan.txt
  bash $R/tools/pmc_model.sh $1 $2 int8 $TAG/sq_$1_b$2 > /dev/null 2>&1
  unset TAMD_PLAN_CACHE
  cp $O/sq_$1_b$2.csv $O/pmc_sq_activity_$1_int8_b$2.csv 2>/dev/null
  rm -rf $O/sq_$1_b$2_1 $O/sq_$1_b$2_2 $O/sq_$1_b$2_3 $O/sq_$1_b$2_*.log $O/sq_$1_b$2.csv
  echo "$cfg: $(wc -l < $O/pmc_sq_activity_$1_int8_b$2.csv) sq-activity lines"
done
rm -f $O/pmc_plan.txt
# ---- 2. bench lines (each seeds its plan file from tengine_amd/plans/)
( time python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_b1_driver_invocation.json 2> $O/bench_b1.err ) 2> $O/bench_b1_driver_invocation_time.txt
python $R/bench.py --steps 2000 --warmup 100 --cpu-seconds 8 --configs none > $O/bench_b1.json 2>> $O/bench_b1.err
python $R/bench.py --model resnet50 --batch 32 --steps 100 --warmup 10 --cpu-seconds 6 > $O/bench_resnet50_int8_b32.json 2> $O/bench_rn.err
python $R/bench.py --model mobilenet_v1 --batch 64 --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_mobilenet_v1_int8_b64.json 2> $O/bench_mb64.err
python $R/bench.py --model yolov3_tiny --dtype uint8 --batch 8 --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_yolov3_tiny_uint8_b8.json 2> $O/bench_yolo.err
python $R/bench.py --model mssd --dtype uint8 --batch 16 --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_mssd_uint8_b16.json 2> $O/bench_mssd.err
python $R/bench.py --model yolov3_tiny --dtype uint8 --u8-integer --batch 8 --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_yolov3_tiny_uint8_int_b8.json 2> $O/bench_yolo_int.err
python $R/bench.py --model mssd --dtype uint8 --u8-integer --batch 16 --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_mssd_uint8_int_b16.json 2> $O/bench_mssd_int.err
for f in $O/bench_*.json; do echo $f; tail -1 $f | python -c "
import sys, json
j = json.loads(sys.stdin.read())
r = j.get('roofline') or {}
print('  value %.0f img/s  %.4f ms/step (%s regions)  h2h %s  pipelined %s  prerun %s ms plan %s golden %s | %s %s frac %.4f avg %.2f us traffic %s mfma_util %.2f%% step_frac %.3f' % (j['value'], j['ms_per_step'], (j.get('timed_regions') or {}).get('repeats'), j.get('host_to_host_images_per_s'), j.get('host_to_host_pipelined_images_per_s'), j.get('prerun_ms'), j.get('shipped_plan'), j.get('golden_match'), r.get('kernel'), r.get('bound'), r.get('frac', 0), r.get('avg_launch_us', 0), r.get('traffic'), r.get('mfma_util_pct', 0), r.get('step_frac', 0)))
for k, c in (j.get('configs') or {}).items():
    rr = c.get('roofline') or {}
    print('    configs.%s: %s' % (k, c.get('error') or '%.4f ms/step %.0f img/s | %s %s frac %.3f step_frac %.3f golden %s' % (c['ms_per_step'], c['images_per_s'], rr.get('kernel'), rr.get('bound'), rr.get('frac', 0), rr.get('step_frac', 0), c['golden_match'])))
"; done
# ---- 3. kernel trace + stats of every config's bench run (hipGraph replay under the tool: rocprofv3 cannot see the direct queue)
for cfg in "mobilenet_v1 1 int8 2000" "mobilenet_v1 64 int8 200" "resnet50 32 int8 200" "yolov3_tiny 8 uint8 100" "mssd 16 uint8 100"; do
  set -- $cfg
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/bench.py --model $1 --dtype $3 --batch $2 --steps $4 --warmup 20 --no-cpu-baseline --configs none --min-seconds 0 > $O/bench_$1_$3_b$2_under_rocprofv3.json 2> $O/trace.err
  find $O/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/rocprofv3_kernel_stats_bench_$1_$3_b$2.csv
  rm -rf $O/trace
  echo "== $cfg"; head -6 $O/rocprofv3_kernel_stats_bench_$1_$3_b$2.csv | cut -c1-160
done
# ---- 4. per-launch tables, and ResNet-50 inside a pass
cd $R
for cfg in "mobilenet_v1 1 50 int8" "mobilenet_v1 64 10 int8" "resnet50 32 10 int8" "yolov3_tiny 8 10 uint8" "mssd 16 10 uint8"; do
  set -- $cfg
  grep -v "^# " $(plan_of $1 $2 $4) > $O/tbl_plan.txt
  TAMD_PLAN_CACHE=$O/tbl_plan.txt python tools/profile_layers.py $1 $2 $3 $4 2>&1 | grep -v "^Tengine" > $O/layers_$1_$4_b$2.txt
done
for f in $O/layers_*.txt; do echo $f; tail -1 $f; done
cd /tmp
grep -v "^# " $(plan_of resnet50 32 int8) > $O/tbl_plan.txt; export TAMD_PLAN_CACHE=$O/tbl_plan.txt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_is -- python $R/tools/replay_model.py resnet50 32 30 int8 > $O/replay.txt 2> $O/trace_is.err
T=$(find $O/trace_is -name "*kernel_trace.csv" | head -1)
N=$(grep -o "launches_per_replay [0-9]*" $O/replay.txt | cut -d' ' -f2)
python $R/tools/trace_gaps.py $T $N 30 > $O/insitu_trace_resnet50_int8_b32.txt 2>&1
unset TAMD_PLAN_CACHE
rm -rf $O/trace_is $O/trace $O/tbl_plan.txt $O/replay.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; find $O -name "*counter_collection.csv" -delete; rm -rf $O/calib_FETCH_SIZE $O/calib_WRITE_SIZE
ls $O | head -80
