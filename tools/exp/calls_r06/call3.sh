#!/bin/bash
# round 6, call 3: the int8 head plumbing (new), the split units (graph*.hip, u8_*.hip) under the GPU suite in a second shuffled order
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_call3
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_int8_heads.py tests/test_plugin_dropin.py tests/test_gpu_edge_cases.py tests/test_gpu_softmax_i8.py tests/test_gpu_glue_int8.py -m gpu -x -q --tb=short 2>&1 | grep -v "^Tengine" > $O/heads.txt; tail -30 $O/heads.txt
timeout 1500 python tools/gpu_suite_shuffled.py 2 2>&1 | grep -v "^Tengine" | tail -60 > $O/pytest_gpu_shuffled_seed2.txt; head -1 $O/pytest_gpu_shuffled_seed2.txt | cut -c1-300; tail -5 $O/pytest_gpu_shuffled_seed2.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json, os
j = json.loads(open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/r06_call3/bench.json")).read().strip().splitlines()[-1])
print("headline %.0f img/s %.4f ms golden %s" % (j["value"], j["ms_per_step"], j["golden_match"]))
for k, c in (j.get("configs") or {}).items():
    print(k, c.get("error") or ("%.4f ms golden %s" % (c["ms_per_step"], c["golden_match"])))
PY
