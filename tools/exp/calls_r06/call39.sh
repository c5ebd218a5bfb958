#!/bin/bash
# round 6, call 39: the final code -- the GPU suite in its own order and in a seeded random file order, smoke, the driver's bench invocation,
# and the rocprofv3 kernel stats of the HALF batches as one launch list (what a launch of a pair costs alone under the tool)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_call39
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -v "^Tengine" | tail -30 > $O/pytest_gpu_all.txt; tail -4 $O/pytest_gpu_all.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "^Tengine" | tail -2 | tee -a $O/pytest_gpu_all.txt
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_b1_driver_invocation.json 2> $O/bench_b1.err ) 2> $O/bench_b1_driver_invocation_time.txt
tail -3 $O/bench_b1_driver_invocation_time.txt
tail -1 $O/bench_b1_driver_invocation.json | python -c "
import sys, json
j = json.loads(sys.stdin.read())
r = j['roofline']
print('headline %.0f img/s %.4f ms (%s regions) golden %s halves %s | %s frac %.4f avg %.2f us traffic %s | h2h %s pipelined %s | cpu %s' % (j['value'], j['ms_per_step'], j['timed_regions']['repeats'], j.get('golden_match'), j['config'].get('halves'), r['kernel'], r['frac'], r['avg_launch_us'], r['traffic'], j.get('host_to_host_images_per_s'), j.get('host_to_host_pipelined_images_per_s'), (j.get('cpu_baseline') or {}).get('value')))
for k, c in (j.get('configs') or {}).items():
    rr = c.get('roofline') or {}
    o = c.get('one_launch_list') or c.get('two_half_batches') or {}
    print('  %s: %s' % (k, c.get('error') or '%.4f ms/step halves %s golden %s | %s frac %.3f step_frac %.3f traffic %s | other form: %s ms golden %s %s' % (c['ms_per_step'], c.get('halves'), c['golden_match'], rr.get('kernel'), rr.get('frac', 0), rr.get('step_frac', 0), rr.get('traffic'), o.get('ms_per_step'), o.get('golden_match'), o.get('error', ''))))
"
cd /tmp
for cfg in "mobilenet_v1 32 int8 200" "resnet50 16 int8 200"; do
  set -- $cfg
  TAMD_SPLIT_BATCH=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/bench.py --model $1 --dtype $3 --batch $2 --steps $4 --warmup 20 --no-cpu-baseline --configs none --min-seconds 0 > $O/bench_$1_$3_b$2_half_alone_under_rocprofv3.json 2> $O/trace.err
  find $O/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/rocprofv3_kernel_stats_bench_$1_$3_b$2_half_alone.csv
  rm -rf $O/trace
  echo "== $cfg (one launch list, TAMD_SPLIT_BATCH=0)"; head -5 $O/rocprofv3_kernel_stats_bench_$1_$3_b$2_half_alone.csv | cut -c1-160
  tail -1 $O/bench_$1_$3_b$2_half_alone_under_rocprofv3.json | python -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j.get('roofline') or {}
print('  under the tool: %.4f ms/step halves %s | %s x%s avg %.2f us frac %.3f' % (j['ms_per_step'], j['config'].get('halves'), r.get('kernel'), r.get('launches_per_step'), r.get('avg_launch_us', 0), r.get('frac', 0)))"
done
cd $R
timeout 1500 python tools/gpu_suite_shuffled.py 10 2>&1 | grep -v "^Tengine" | tail -12 > $O/pytest_gpu_shuffled_seed10.txt; tail -3 $O/pytest_gpu_shuffled_seed10.txt
find $O -name "*.db" -delete
