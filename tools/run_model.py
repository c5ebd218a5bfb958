#!/usr/bin/env python3
"""Run a config model a few times eagerly (no hipGraph) -- the target command for rocprofv3 counter passes.
usage: run_model.py [model] [batch] [iters] [int8|uint8]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tengine_amd import capi, models, tm2  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "mobilenet_v1"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dtype = sys.argv[4] if len(sys.argv) > 4 else "int8"
g = models.build(name, dtype, batch, device_only=True)
gr = capi.Graph(tm2.write_tm2(g), batch=batch, use_hip_graph=False)
gr.set_input(models.synth_input(g, 3, {"uint8": tm2.DT_UINT8, "fp32": tm2.DT_FP32}.get(dtype, tm2.DT_INT8)))
print("launches_per_run %d iters %d" % (gr.kernel_num() + 2, iters))     # + upload / download launches of tamd_graph_run
for _ in range(iters):
    gr.run()
gr.close()
