#!/bin/bash
# Runs ON THE GPU BOX (gpurun): the judged evidence of round 3.  usage: tools/collect_evidence_r03.sh [tag]
#  1. bench lines: the BASELINE headline (driver invocation and a long one) and the side configs
#  2. rocprofv3 --kernel-trace --stats of the headline bench with a WARM plan cache: the CSV holds the run's own launches only
#  3. per-launch tables (HIP events) of the configs
#  4. PMC passes (separate runs, counters only) for the side configs: FETCH_SIZE / WRITE_SIZE traffic + MFMA busy, per kernel
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
# ---- 1. bench lines
python $R/bench.py --steps 20 --warmup 5 > $O/bench_b1_driver_invocation.json 2> $O/bench_b1.err
python $R/bench.py --steps 2000 --warmup 100 --cpu-seconds 8 > $O/bench_b1.json 2>> $O/bench_b1.err
python $R/bench.py --model resnet50 --batch 32 --steps 100 --warmup 10 --cpu-seconds 6 > $O/bench_resnet50_int8_b32.json 2> $O/bench_rn.err
python $R/bench.py --model mobilenet_v1 --batch 64 --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_mobilenet_v1_int8_b64.json 2> $O/bench_mb64.err
python $R/bench.py --model yolov3_tiny --dtype uint8 --batch 8 --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_yolov3_tiny_uint8_b8.json 2> $O/bench_yolo.err
python $R/bench.py --model mssd --dtype uint8 --batch 16 --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_mssd_uint8_b16.json 2> $O/bench_mssd.err
for f in $O/bench_*.json; do echo $f; tail -1 $f | python -c "
import sys, json
j = json.loads(sys.stdin.read())
r = j.get('roofline') or {}
print('  value %.0f img/s  %.4f ms/step  h2h %s  pipelined %s  prerun %s ms | %s %s frac %.4f avg %.2f us  mfma_util %.2f%%' % (j['value'], j['ms_per_step'], j.get('host_to_host_images_per_s'), j.get('host_to_host_pipelined_images_per_s'), j.get('prerun_ms'), r.get('kernel'), r.get('bound'), r.get('frac', 0), r.get('avg_launch_us', 0), r.get('mfma_util_pct', 0)))
"; done
# ---- 2. kernel trace of the headline bench, autotune launches excluded through the plan cache
export TAMD_PLAN_CACHE=$O/plan_mobilenet_v1_int8_b1.txt
python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline > /dev/null 2>&1          # writes the plan
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/bench.py --steps 2000 --warmup 100 --no-cpu-baseline > $O/bench_b1_under_rocprofv3.json 2> $O/trace.err
find $O/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/rocprofv3_kernel_stats_bench_b1.csv
head -12 $O/rocprofv3_kernel_stats_bench_b1.csv | cut -c1-150
unset TAMD_PLAN_CACHE
# ---- 3. per-launch tables
cd $R
python tools/profile_layers.py mobilenet_v1 1 50 int8  > $O/layers_mobilenet_v1_int8_b1.txt 2>&1
python tools/profile_layers.py mobilenet_v1 64 10 int8 > $O/layers_mobilenet_v1_int8_b64.txt 2>&1
python tools/profile_layers.py resnet50 32 10 int8     > $O/layers_resnet50_int8_b32.txt 2>&1
python tools/profile_layers.py yolov3_tiny 8 10 uint8  > $O/layers_yolov3_tiny_uint8_b8.txt 2>&1
python tools/profile_layers.py mssd 16 10 uint8        > $O/layers_mssd_uint8_b16.txt 2>&1
for f in $O/layers_*.txt; do echo $f; tail -1 $f; done
# ---- 4. PMC: traffic (calibrated) and MFMA busy per kernel, plan cache warm so that only the runs' launches are counted
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/calib_$c -- $R/tools/exp/hbm_calib.bin > $O/calib_$c.log 2>&1
done
for cfg in "mobilenet_v1 1 int8" "resnet50 32 int8" "mobilenet_v1 64 int8" "yolov3_tiny 8 uint8" "mssd 16 uint8"; do
  set -- $cfg
  export TAMD_PLAN_CACHE=$O/plan_$1_$3_b$2.txt
  python $R/tools/run_model.py $1 $2 1 $3 > /dev/null 2>&1        # writes the plan
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $O/m_$c -- python $R/tools/run_model.py $1 $2 5 $3 > $O/m_$c.log 2>&1
  done
  K=$(grep -o "launches_per_run [0-9]*" $O/m_FETCH_SIZE.log | cut -d' ' -f2)
  python $R/tools/traffic_summary.py $O/traffic_$1_$3_b$2.json $O/calib_FETCH_SIZE $O/calib_WRITE_SIZE $O/m_FETCH_SIZE $O/m_WRITE_SIZE $((K * 5)) > $O/traffic_$1_$3_b$2.txt 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $O/m_mfma -- python $R/tools/run_model.py $1 $2 5 $3 > $O/m_mfma.log 2>&1
  python $R/tools/pmc_summary.py $O/pmc_mfma_$1_$3_b$2.csv $O/m_mfma > /dev/null 2>&1
  rm -rf $O/m_FETCH_SIZE $O/m_WRITE_SIZE $O/m_mfma
  unset TAMD_PLAN_CACHE
  echo "$cfg: $(wc -l < $O/traffic_$1_$3_b$2.txt) traffic lines, $(wc -l < $O/pmc_mfma_$1_$3_b$2.csv) pmc lines"
done
# ---- 5. inside a pass: per-position kernel durations of hipGraph replays (uint8 configs), and the patch kernel's ablation anatomy
cd /tmp
for cfg in "yolov3_tiny 8" "mssd 16"; do
  set -- $cfg
  export TAMD_PLAN_CACHE=$O/plan_$1_uint8_b$2.txt
  rm -rf $O/trace_is
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_is -- python $R/tools/replay_model.py $1 $2 30 uint8 > $O/replay_$1.txt 2> $O/trace_is.err
  T=$(find $O/trace_is -name "*kernel_trace.csv" | head -1)
  N=$(grep -o "launches_per_replay [0-9]*" $O/replay_$1.txt | cut -d' ' -f2)
  python $R/tools/trace_gaps.py $T $N 30 > $O/insitu_trace_$1_uint8_b$2.txt 2>&1
  unset TAMD_PLAN_CACHE
  python $R/tools/replay_model.py $1 $2 30 uint8 >> $O/insitu_trace_$1_uint8_b$2.txt 2>&1      # the same replays without the tracer
done
rm -rf $O/trace_is
if ls $R/tools/exp/u8_patch_anatomy_0.bin > /dev/null 2>&1; then
  cd $R/tools/exp
  for m in 0 1 2 4 8 16 3 7 15 31; do timeout 120 ./u8_patch_anatomy_$m.bin; done 2>&1 | sort -s -k1,1 > $O/u8_patch_anatomy_ablation.txt
fi
cd /tmp
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; find $O -name "*counter_collection.csv" -delete; rm -rf $O/trace $O/calib_FETCH_SIZE $O/calib_WRITE_SIZE
ls $O | head -60
