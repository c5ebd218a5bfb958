#!/bin/bash
# round 6, call 46: the plugin without its own two-object paths (the library's pair is the only one): its drop-in tests, the reference's tm_benchmark
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r06_call46
timeout 600 python -m pytest tests/test_plugin_dropin.py tests/test_tm_benchmark_hip.py tests/test_reference_benchmark_files.py -m gpu -q --tb=short 2>&1 | grep -v "^Tengine" | tail -6 | tee $R/gpurun_out/r06_call46/pytest_plugin.txt
timeout 200 python tools/exp/plugin_split_ab.py mobilenet_v1 64 50 3 2>&1 | grep -v "^Tengine" | tail -1
