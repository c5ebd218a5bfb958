#!/bin/bash
# round 6, call 43: after tamd_graph.formula_batch (the halves follow the WHOLE batch's depthwise formula) and the plugin delegating its split to
# the library: the pair fuzz again, the split / plugin tests
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_call43
mkdir -p $O
cd $R
export TMPDIR=/tmp
for cfg in "int8 80 1" "int8 60 2" "uint8 40 1"; do
  set -- $cfg
  timeout 400 python tools/fuzz_split.py --dtype $1 --seconds $2 --seed $3 2>&1 | grep -v "^Tengine" | tail -6
done | tee $O/fuzz_split_device.txt
timeout 900 python -m pytest tests/test_gpu_split_batch.py tests/test_plugin_dropin.py tests/test_tm_benchmark_hip.py -m gpu -q --tb=short 2>&1 | grep -v "^Tengine" | tail -15 | cut -c1-300
