#!/usr/bin/env python3
"""Run a config model a few times eagerly (no hipGraph) -- the target command for rocprofv3 counter passes.
usage: run_model.py [model] [batch] [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tengine_amd import capi, models, tm2  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "mobilenet_v1"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
g = models.build(name, "int8", batch, device_only=True)
gr = capi.Graph(tm2.write_tm2(g), batch=batch, use_hip_graph=False)
gr.set_input(models.synth_input(g, 3))
for _ in range(iters):
    gr.run()
gr.close()
