#!/bin/bash
# round 6, call 34: the whole GPU suite (with tests/test_gpu_split_batch.py), smoke, the driver's bench invocation -- on the code with the
# two-half-batch form inside one tamd_graph
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_call34
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -v "^Tengine" | tail -30 > $O/pytest_gpu_all.txt; tail -8 $O/pytest_gpu_all.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "^Tengine" | tail -3 | tee $O/smoke.txt
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_b1_driver_invocation.json 2> $O/bench_b1.err ) 2> $O/bench_time.txt
tail -3 $O/bench_time.txt
tail -1 $O/bench_b1_driver_invocation.json | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('headline %.0f img/s %.4f ms golden %s halves %s' % (j['value'], j['ms_per_step'], j.get('golden_match'), j['config'].get('halves')))
for k, c in (j.get('configs') or {}).items():
    rr = c.get('roofline') or {}
    o = c.get('one_launch_list') or c.get('two_half_batches') or {}
    print('  %s: %s' % (k, c.get('error') or '%.4f ms/step halves %s golden %s | %s frac %.3f step_frac %.3f traffic %s | other form: %s ms golden %s %s' % (c['ms_per_step'], c.get('halves'), c['golden_match'], rr.get('kernel'), rr.get('frac', 0), rr.get('step_frac', 0), rr.get('traffic'), o.get('ms_per_step'), o.get('golden_match'), o.get('error', ''))))
"
