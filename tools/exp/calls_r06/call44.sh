#!/bin/bash
# round 6, call 44: the final code -- the GPU suite in its own order and in a seeded random file order (seed 11), smoke, the driver's bench invocation
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_call44
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
print('headline %.0f img/s %.4f ms (%s regions) golden %s | %s frac %.4f avg %.2f us traffic %s | h2h %s pipelined %s | cpu %s' % (j['value'], j['ms_per_step'], j['timed_regions']['repeats'], j.get('golden_match'), r['kernel'], r['frac'], r['avg_launch_us'], r['traffic'], j.get('host_to_host_images_per_s'), j.get('host_to_host_pipelined_images_per_s'), (j.get('cpu_baseline') or {}).get('value')))
for k, c in (j.get('configs') or {}).items():
    o = c.get('one_launch_list') or c.get('two_half_batches') or {}
    print('  %s: %s' % (k, c.get('error') or '%.4f ms/step halves %s golden %s | other form: %s ms golden %s' % (c['ms_per_step'], c.get('halves'), c['golden_match'], o.get('ms_per_step'), o.get('golden_match'))))
"
timeout 1500 python tools/gpu_suite_shuffled.py 11 2>&1 | grep -v "^Tengine" | tail -12 > $O/pytest_gpu_shuffled_seed11.txt; tail -3 $O/pytest_gpu_shuffled_seed11.txt
