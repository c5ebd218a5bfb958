#!/bin/bash
# round 6, call 38: direct_timestamps drops the passes in front of the first fully stamped one (short launch lists); split + direct tests; what
# the stamps of a short request look like (TAMD_DEBUG)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_call38
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_split_batch.py tests/test_gpu_direct.py -m gpu -q --tb=short -s 2>&1 | grep -v "^Tengine" > $O/pytest_split_direct.txt; grep "direct path\|passed\|failed" $O/pytest_split_direct.txt | tail -4; grep -B2 -A12 "^E " $O/pytest_split_direct.txt | head -40 | cut -c1-300
TAMD_DEBUG=1 python - <<'PY' 2>&1 | grep "stamp\|rows\|Error\|error" | head -20
import os, sys
sys.path.insert(0, os.getcwd())
from tengine_amd import capi, models, tm2
g = models.build("mobilenet_v1", "int8", 4)
gr = capi.Graph(tm2.write_tm2(g), direct_dispatch=True, split_batch=2)
gr.set_input(models.synth_input(g, 1, tm2.DT_INT8)); gr.upload(); gr.sync()
gr.time_launches(5)
for p in (5, 5, 30):
    rows = gr.direct_timestamps(p)
    print("rows", p, len(rows), "%.2f" % sum(r[1] for r in rows))
PY
