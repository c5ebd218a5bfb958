#!/bin/bash
# Runs ON THE GPU BOX (gpurun): the judged evidence of one round.  usage: tools/collect_profiles.sh <tag>
#  1. bench.py (default: int8 MobileNet-v1 b1) plain, then under rocprofv3 --kernel-trace --stats
#  2. FETCH_SIZE / WRITE_SIZE passes (separate --pmc runs, no tracing) for the calibration copy and for the model; only the
#     trailing dispatches of the model passes (the runs themselves, not the plan-time autotune) enter the traffic figures
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
python $R/bench.py --steps 500 --warmup 50 > $O/bench_b1.json 2> $O/bench_b1.err
tail -1 $O/bench_b1.json | cut -c1-400
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/bench.py --steps 500 --warmup 50 --no-cpu-baseline > $O/bench_b1_traced.json 2> $O/trace.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/calib_$c -- $R/tools/exp/hbm_calib.bin > $O/calib_$c.log 2>&1
  rocprofv3 --pmc $c --output-format csv -d $O/model_$c -- python $R/tools/run_model.py mobilenet_v1 1 10 int8 > $O/model_$c.log 2>&1
done
K=$(grep -o "launches_per_run [0-9]*" $O/model_FETCH_SIZE.log | cut -d' ' -f2)
python $R/tools/traffic_summary.py $O/traffic_mobilenet_v1_int8_b1.json $O/calib_FETCH_SIZE $O/calib_WRITE_SIZE $O/model_FETCH_SIZE $O/model_WRITE_SIZE $((K * 10)) > $O/traffic.txt 2>&1
head -40 $O/traffic.txt
find $O/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
head -14 $O/kernel_stats.csv | cut -c1-170
# keep the merged output small: drop the raw per-dispatch traces
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; find $O -name "*counter_collection.csv" -delete
