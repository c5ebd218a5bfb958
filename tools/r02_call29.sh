#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02ae
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_direct.py tests/test_gpu_pwdw.py -q -x -s 2>&1 | tail -8 | tee $O/pytest_direct.txt
timeout 300 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline > $O/bench_b1_direct_coh.json 2> $O/bench_b1_direct_coh.err; tail -1 $O/bench_b1_direct_coh.json | cut -c1-330
TAMD_DIRECT_COHERENT=0 timeout 300 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline > $O/bench_b1_direct_cohkernels_agentfences.json 2>/dev/null; tail -1 $O/bench_b1_direct_cohkernels_agentfences.json | cut -c1-230
timeout 300 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --direct 0 > $O/bench_b1_graph.json 2>/dev/null; tail -1 $O/bench_b1_graph.json | cut -c1-230
timeout 300 python tools/profile_layers.py mobilenet_v1 1 50 int8 > $O/layers_mobilenet_v1_int8_b1.txt 2>&1; tail -3 $O/layers_mobilenet_v1_int8_b1.txt
