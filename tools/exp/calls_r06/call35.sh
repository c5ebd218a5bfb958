#!/bin/bash
# round 6, call 35: evidence of the two configurations whose graph took the two-half-batch form (MobileNet-v1 b64, ResNet-50 b32): their own
# bench lines, rocprofv3 --kernel-trace --stats of the same commands, the directly dispatched passes under the HSA runtime's dispatch stamps
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_call35
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
python $R/bench.py --model resnet50 --batch 32 --steps 100 --warmup 10 --cpu-seconds 6 > $O/bench_resnet50_int8_b32.json 2> $O/bench_rn.err
python $R/bench.py --model mobilenet_v1 --batch 64 --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_mobilenet_v1_int8_b64.json 2> $O/bench_mb64.err
for f in $O/bench_*.json; do echo $f; tail -1 $f | python -c "
import sys, json
j = json.loads(sys.stdin.read())
r = j.get('roofline') or {}
print('  value %.0f img/s  %.4f ms/step (%s regions) halves %s h2h %s  pipelined %s  plan %s golden %s | %s %s frac %.4f avg %.2f us traffic %s alg %s mfma_util %.2f%% step_frac %.3f stamps %s' % (j['value'], j['ms_per_step'], (j.get('timed_regions') or {}).get('repeats'), j['config'].get('halves'), j.get('host_to_host_images_per_s'), j.get('host_to_host_pipelined_images_per_s'), j.get('shipped_plan'), j.get('golden_match'), r.get('kernel'), r.get('bound'), r.get('frac', 0), r.get('avg_launch_us', 0), r.get('traffic'), r.get('algorithmic_bytes_per_launch'), r.get('mfma_util_pct', 0), r.get('step_frac', 0), (r.get('hsa_dispatch_stamps') or {}).get('avg_launch_us')))
"; done
for cfg in "mobilenet_v1 64 int8 200" "resnet50 32 int8 200"; do
  set -- $cfg
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/bench.py --model $1 --dtype $3 --batch $2 --steps $4 --warmup 20 --no-cpu-baseline --configs none --min-seconds 0 > $O/bench_$1_$3_b$2_under_rocprofv3.json 2> $O/trace.err
  find $O/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/rocprofv3_kernel_stats_bench_$1_$3_b$2.csv
  rm -rf $O/trace
  echo "== $cfg"; head -6 $O/rocprofv3_kernel_stats_bench_$1_$3_b$2.csv | cut -c1-160
  tail -1 $O/bench_$1_$3_b$2_under_rocprofv3.json | python -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j.get('roofline') or {}
print('  under the tool: %.4f ms/step halves %s | %s avg %.2f us' % (j['ms_per_step'], j['config'].get('halves'), r.get('kernel'), r.get('avg_launch_us', 0)))"
done
cd $R
for cfg in "mobilenet_v1 64 int8 100" "resnet50 32 int8 100"; do
  set -- $cfg
  python tools/direct_timestamps.py $1 $2 $3 $4 2>&1 | grep -v "^Tengine" > $O/direct_path_timestamps_$1_$3_b$2.txt
  head -1 $O/direct_path_timestamps_$1_$3_b$2.txt; tail -2 $O/direct_path_timestamps_$1_$3_b$2.txt
done
# host to host through the handle: blocking tamd_graph_run per batch, both forms (tools/exp/split_ab.py times the resident loop; this is run())
python - <<'PY' 2>&1 | grep -v "^Tengine" | tee $O/split_h2h_one_handle.txt
import os, sys, time, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
import numpy as np
from tengine_amd import capi, models, plans, tm2
for model, batch in (("mobilenet_v1", 64), ("resnet50", 32)):
    plan = os.path.join(tempfile.gettempdir(), "h2h_%s.txt" % model)
    plans.seed(plan, model, "int8", batch)
    os.environ["TAMD_PLAN_CACHE"] = plan
    g = models.build(model, "int8", batch)
    b = tm2.write_tm2(g)
    x = models.synth_input(g, 1000, tm2.DT_INT8)
    grs = [capi.Graph(b, batch=batch, direct_dispatch=True, split_batch=m) for m in (1, 2)]
    best = [1e9, 1e9]
    outs = []
    for gr in grs:
        gr.set_input(x)
        for _ in range(5):
            gr.run_noreturn()
    for _ in range(5):
        for k, gr in enumerate(grs):
            t0 = time.perf_counter()
            for _ in range(50):
                gr.run_noreturn()
            best[k] = min(best[k], (time.perf_counter() - t0) / 50)
    same = all(np.array_equal(a, c) for a, c in zip(grs[0].run(), grs[1].run()))
    print("%-14s int8 batch %3d, blocking tamd_graph_run() host to host, us per run (min of 5 x 50, interleaved): one launch list %8.1f | two half-batch graphs behind one handle %8.1f (%+5.1f %%) | outputs %s"
          % (model, batch, 1e6 * best[0], 1e6 * best[1], 100.0 * (best[0] / best[1] - 1.0), "identical" if same else "DIFFER"))
    for gr in grs:
        gr.close()
PY
find $O -name "*.db" -delete
ls $O
