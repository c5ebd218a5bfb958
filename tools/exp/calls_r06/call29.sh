#!/bin/bash
# round 6, call 29: the plugin's two-half-batch form: the drop-in tests (incl. the new ones), the reference's own tm_benchmark / model-file
# tests, and its host-to-host A/B through the unmodified reference library
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_call29
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_plugin_dropin.py tests/test_tm_benchmark_hip.py tests/test_reference_benchmark_files.py -m gpu -q --tb=short 2>&1 | grep -v "^Tengine" | tail -30 > $O/pytest_plugin.txt; tail -12 $O/pytest_plugin.txt
for cfg in "resnet50 32 50 3" "mobilenet_v1 64 50 3" "mobilenet_v1 16 100 3"; do
  timeout 900 python tools/exp/plugin_split_ab.py $cfg 2>&1 | grep -v "^Tengine" | tail -3
done | tee $O/plugin_split_ab.txt
