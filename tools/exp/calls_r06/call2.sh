#!/bin/bash
# round 6, call 2: the driver's bench invocation with the new `configs` object and repeated regions; the GPU suite with its files in a
# seeded random order (the fixed zero-copy test in whatever position the shuffle gives it); round 5 call 27's subset again
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_call2
mkdir -p $O
cd $R
export TMPDIR=/tmp
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_invocation.json 2> $O/bench.err ) 2> $O/bench_time.txt
tail -3 $O/bench_time.txt; tail -c 300 $O/bench.err
python - <<'PY'
import json, os
j = json.loads(open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/r06_call2/bench_driver_invocation.json")).read().strip().splitlines()[-1])
print("headline %.0f img/s %.4f ms (min %.4f max %.4f, %d regions) golden_match %s" % (j["value"], j["ms_per_step"], j["timed_regions"]["ms_per_step_min"], j["timed_regions"]["ms_per_step_max"], j["timed_regions"]["repeats"], j["golden_match"]))
for k, c in (j.get("configs") or {}).items():
    if "error" in c:
        print(k, "ERROR", c["error"]); continue
    r = c["roofline"]
    print("%s: %.4f ms/step %.0f img/s | %s %s frac %.3f step_frac %.3f traffic %s | golden_match %s prerun %.0f ms plan %s" % (k, c["ms_per_step"], c["images_per_s"], r["kernel"], r["bound"], r["frac"], r["step_frac"], r["traffic"], c["golden_match"], c["prerun_ms"], c["shipped_plan"]))
PY
timeout 900 python -m pytest tests/test_gpu_pwdw.py tests/test_gpu_plan_cache.py tests/test_gpu_baseline_batches.py tests/test_gpu_direct.py -m gpu -x -q --tb=short 2>&1 | grep -v "^Tengine" > $O/subset.txt; tail -2 $O/subset.txt
timeout 1500 python tools/gpu_suite_shuffled.py 1 2>&1 | grep -v "^Tengine" | tail -60 > $O/pytest_gpu_shuffled_seed1.txt; head -1 $O/pytest_gpu_shuffled_seed1.txt | cut -c1-300; tail -3 $O/pytest_gpu_shuffled_seed1.txt
