#!/bin/bash
# round 6, call 32: plugin tests with the split threshold at batch 8
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_call32
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_plugin_dropin.py tests/test_tm_benchmark_hip.py tests/test_reference_benchmark_files.py -m gpu -q --tb=short 2>&1 | grep -v "^Tengine" | tail -30 > $O/pytest_plugin.txt; tail -6 $O/pytest_plugin.txt
