#!/bin/bash
# round 6, call 33: the batch as two half-batch device graphs INSIDE one tamd_graph (csrc/graph_pair.hip): its tests, the neighbours it
# touches, the A/B per batch (threshold of the default rule), the driver's bench invocation, PMC traffic of the half-batch launches
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_call33
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_split_batch.py -m gpu -q --tb=short -x 2>&1 | grep -v "^Tengine" | tail -40 > $O/pytest_split.txt; tail -15 $O/pytest_split.txt
timeout 900 python -m pytest tests/test_abi.py tests/test_gpu_direct.py tests/test_gpu_async.py tests/test_plugin_dropin.py tests/test_gpu_plan_cache.py tests/test_bench_host_logic.py tests/test_gpu_bench_dist.py tests/test_gpu_rccl_c.py -m gpu -q --tb=short 2>&1 | grep -v "^Tengine" | tail -15 > $O/pytest_neighbours.txt; tail -6 $O/pytest_neighbours.txt
for cfg in "mobilenet_v1 int8 8" "mobilenet_v1 int8 16" "mobilenet_v1 int8 32" "mobilenet_v1 int8 64" "resnet50 int8 8" "resnet50 int8 16" "resnet50 int8 32" "yolov3_tiny uint8 8" "mssd uint8 16"; do
  timeout 600 python tools/exp/split_ab.py $cfg 100 5 2>&1 | grep -v "^Tengine" | tail -1
done | tee $O/split_ab_one_handle.txt
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_b1_driver_invocation.json 2> $O/bench_b1.err ) 2> $O/bench_time.txt
tail -3 $O/bench_time.txt; tail -5 $O/bench_b1.err
tail -1 $O/bench_b1_driver_invocation.json | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('headline %.0f img/s %.4f ms golden %s' % (j['value'], j['ms_per_step'], j.get('golden_match')))
for k, c in (j.get('configs') or {}).items():
    rr = c.get('roofline') or {}
    o = c.get('one_launch_list') or c.get('two_half_batches') or {}
    print('  %s: %s' % (k, c.get('error') or '%.4f ms/step halves %s golden %s | %s frac %.3f step_frac %.3f traffic %s | other form: %s ms golden %s %s' % (c['ms_per_step'], c.get('halves'), c['golden_match'], rr.get('kernel'), rr.get('frac', 0), rr.get('step_frac', 0), rr.get('traffic'), o.get('ms_per_step'), o.get('golden_match'), o.get('error', ''))))
"
# PMC traffic of the half-batch launch lists (what a pair launches)
cd /tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o $R/tools/exp/hbm_calib.bin $R/tools/exp/hbm_calib.hip 2>&1 | tail -2
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/calib_$c -- $R/tools/exp/hbm_calib.bin > $O/calib_$c.log 2>&1
done
for cfg in "resnet50 16 int8" "mobilenet_v1 32 int8"; do
  set -- $cfg
  grep -v "^# " $R/tengine_amd/plans/$1_$3_b$2.txt > $O/pmc_plan.txt; export TAMD_PLAN_CACHE=$O/pmc_plan.txt
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $O/m_$c -- python $R/tools/run_model.py $1 $2 5 $3 > $O/m_$c.log 2>&1
  done
  K=$(grep -o "launches_per_run [0-9]*" $O/m_FETCH_SIZE.log | cut -d' ' -f2)
  python $R/tools/traffic_summary.py $O/traffic_$1_$3_b$2.json $O/calib_FETCH_SIZE $O/calib_WRITE_SIZE $O/m_FETCH_SIZE $O/m_WRITE_SIZE $((K * 5)) > $O/traffic_$1_$3_b$2.txt 2>&1
  rm -rf $O/m_FETCH_SIZE $O/m_WRITE_SIZE
  unset TAMD_PLAN_CACHE
  echo "$cfg: $(wc -l < $O/traffic_$1_$3_b$2.txt) traffic lines"
done
rm -rf $O/calib_FETCH_SIZE $O/calib_WRITE_SIZE $O/pmc_plan.txt
find $O -name "*.db" -delete
ls $O
