#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02ab
mkdir -p $O
cd $R
for v in "1 1" "1 0" "0 1"; do
  set -- $v
  TAMD_DIRECT_FENCE=$2 timeout 300 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --direct $1 > $O/bench_direct$1_fence$2.json 2> $O/bench_direct$1_fence$2.err
  python - <<PY
import json
l=json.loads(open("$O/bench_direct$1_fence$2.json").read().strip().splitlines()[-1])
print("direct=$1 fence=$2", round(l["ms_per_step"]*1e3,2), "us/step", round(l["value"]), "img/s", l["config"]["workload"][-70:], "checksum", l["output_checksum"])
PY
done
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --model resnet50 --batch 32 > $O/bench_rn50_direct.json 2>&1; tail -1 $O/bench_rn50_direct.json | cut -c1-200
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --model resnet50 --batch 32 --direct 0 > $O/bench_rn50_graph.json 2>&1; tail -1 $O/bench_rn50_graph.json | cut -c1-200
timeout 300 python bench.py --steps 500 --warmup 50 --no-cpu-baseline --streams 4 > $O/bench_b1_4streams_direct.json 2>&1; tail -1 $O/bench_b1_4streams_direct.json | cut -c1-200
timeout 300 python bench.py --steps 500 --warmup 50 --no-cpu-baseline --streams 4 --direct 0 > $O/bench_b1_4streams_graph.json 2>&1; tail -1 $O/bench_b1_4streams_graph.json | cut -c1-200
