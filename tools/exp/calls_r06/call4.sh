#!/bin/bash
# round 6, call 4: the int8 head tests after the Flatten-of-a-map fix, the plugin tests with their new CPU-only tail; the int8 Winograd
# F(2,3) measurement (row W) against the product's direct kernel on the same layer, same box
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_call4
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_int8_heads.py tests/test_plugin_dropin.py tests/test_gpu_glue_int8.py tests/test_gpu_edge_cases.py -m gpu -q --tb=short 2>&1 | grep -v "^Tengine" > $O/heads.txt; tail -15 $O/heads.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/winograd_i8_anatomy.bin tools/exp/winograd_i8_anatomy.hip 2>&1 | tail -3
{
  /tmp/winograd_i8_anatomy.bin 32 28
  echo "---- the product on the same layer, same box (3x3 / s1 / p1, 128 -> 128, 28 x 28, batch 32; plan-time race as in a model) ----"
  python - <<'PY' 2>&1 | grep -v "^Tengine"
import sys
sys.path.insert(0, "tests")
from helpers import conv_graph
from tengine_amd import capi, tm2
g, x = conv_graph(1, 32, 128, 28, 28, 128, 3, 1, 1, 1, 0, True, 1)
gr = capi.Graph(tm2.write_tm2(g))
gr.set_input(x)
gr.run()
for rep in range(3):
    for q in gr.profile(20):
        print("%-12s %-44s %8.2f us  (%.0f TOP/s)" % (q["node"], q["kernel"], q["ms"] * 1e3, 2e-9 * q["macs"] / (q["ms"] * 1e-3) / 1e3))
gr.close()
PY
  echo "---- smaller case (N 2, 13 x 13: odd map, edge tiles) ----"
  /tmp/winograd_i8_anatomy.bin 2 13
} 2>&1 | tee $O/winograd_i8_res3.txt
