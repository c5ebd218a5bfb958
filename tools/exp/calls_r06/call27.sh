#!/bin/bash
# round 6, call 27: plans of the four half-batch configurations (bench.py's `two_half_batches` side measurement), then the driver's bench
# invocation with it
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_call27
mkdir -p $O/plans
cd $R
export TMPDIR=/tmp
python tools/make_plans.py $O/plans half 2>&1 | grep -v "^Tengine" | tee $O/plans_half.txt
cp $O/plans/*.txt tengine_amd/plans/
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_b1_driver_invocation.json 2> $O/bench.err ) 2> $O/bench_time.txt
tail -5 $O/bench.err; cat $O/bench_time.txt
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r06_call27/bench_b1_driver_invocation.json").read().strip().splitlines()[-1])
print("value %.0f ms %.4f" % (j["value"], j["ms_per_step"]))
for k, c in j["configs"].items():
    h = c.get("two_half_batches") or {}
    print(k, c.get("error") or "%.4f ms golden %s | halves: %s" % (c["ms_per_step"], c["golden_match"], h.get("error") or ("%.4f ms (%+.1f %%) golden %s plan %s" % (h["ms_per_step"], 100 * (c["ms_per_step"] / h["ms_per_step"] - 1), h["golden_match"], h["shipped_plan"]) if h else None)))
PY
