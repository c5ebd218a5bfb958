#!/bin/bash
# final state of the round: whole GPU suite + the bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02ak
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu_all.txt 2>&1
tail -3 $O/pytest_gpu_all.txt
timeout 300 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline > $O/bench_b1.json 2> $O/bench_b1.err; tail -1 $O/bench_b1.json | cut -c1-260
