#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02ac
mkdir -p $O
cd $R
timeout 500 python -m pytest tests/test_gpu_direct.py tests/test_gpu_bench_dist.py -q -x -s 2>&1 | tail -8 | tee $O/pytest_direct.txt
timeout 300 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline > $O/bench_b1_direct.json 2> $O/bench_b1_direct.err; tail -1 $O/bench_b1_direct.json | cut -c1-330
timeout 300 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --direct 0 > $O/bench_b1_graph.json 2> $O/bench_b1_graph.err; tail -1 $O/bench_b1_graph.json | cut -c1-330
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof_direct -- python $R/bench.py --steps 300 --warmup 30 --no-cpu-baseline > $O/rocprof_direct.log 2>&1
find $O/rocprof_direct -name "*kernel_stats.csv" | head -2
f=$(find $O/rocprof_direct -name "*kernel_stats.csv" | head -1); head -12 "$f" | cut -c1-200; cp "$f" $O/rocprofv3_kernel_stats_bench_b1_direct.csv; rm -rf $O/rocprof_direct
