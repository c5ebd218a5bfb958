#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02af
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_direct.py tests/test_gpu_async.py -q -x 2>&1 | tail -5 | tee $O/pytest_direct.txt
timeout 300 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline > $O/bench_b1_direct.json 2> $O/bench_b1_direct.err
python - <<PY
import json
for f in ["bench_b1_direct.json"]:
    l=json.loads(open("$O/"+f).read().strip().splitlines()[-1]); print(f, round(l["ms_per_step"]*1e3,2), "us/step", round(l["value"]), "img/s ; host_to_host:", json.dumps(l["host_to_host"])[:600])
PY
timeout 300 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --direct 0 > $O/bench_b1_graph.json 2> $O/bench_b1_graph.err
python - <<PY
import json
for f in ["bench_b1_graph.json"]:
    l=json.loads(open("$O/"+f).read().strip().splitlines()[-1]); print(f, round(l["ms_per_step"]*1e3,2), "us/step", round(l["value"]), "img/s ; host_to_host:", json.dumps(l["host_to_host"])[:600])
PY
TAMD_DIRECT_DISPATCH=1 timeout 300 python tools/tm_benchmark.py > $O/tm_benchmark_direct.txt 2>&1; tail -8 $O/tm_benchmark_direct.txt
