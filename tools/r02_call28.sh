#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02ad
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
env > $O/env_plain.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof_direct -- bash -c "env | grep -i -E 'rocp|hsa_tools|preload' > $O/env_under_rocprofv3.txt; python $R/bench.py --steps 300 --warmup 30 --no-cpu-baseline" > $O/rocprof_direct.log 2>&1
echo rc=$?
cat $O/env_under_rocprofv3.txt | cut -c1-200
tail -1 $O/rocprof_direct.log | cut -c1-400
f=$(find $O/rocprof_direct -name "*kernel_stats.csv" | head -1); head -12 "$f" | cut -c1-200; cp "$f" $O/rocprofv3_kernel_stats_bench_b1.csv; rm -rf $O/rocprof_direct
