#!/usr/bin/env python3
"""hipGraph replays of a config model and nothing else after prerun -- the target command for
`rocprofv3 --kernel-trace` when the per-dispatch timestamps are analysed (tools/trace_gaps.py).
usage: replay_model.py [model] [batch] [replays] [int8|uint8]   (prints the number of launches per replay)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tengine_amd import capi, models, tm2  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "mobilenet_v1"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1
replays = int(sys.argv[3]) if len(sys.argv) > 3 else 100
dtype = sys.argv[4] if len(sys.argv) > 4 else "int8"
eager = len(sys.argv) > 5 and sys.argv[5] == "eager"      # launch the kernels one by one instead of replaying the hipGraph
g = models.build(name, dtype, batch, device_only=(name != "mobilenet_v1"))
gr = capi.Graph(tm2.write_tm2(g), batch=batch, use_hip_graph=not eager)
gr.set_input(models.synth_input(g, 3, {"uint8": tm2.DT_UINT8, "fp32": tm2.DT_FP32}.get(dtype, tm2.DT_INT8)))
gr.upload()
gr.sync()
ms = gr.time_launches(replays)
print("launches_per_replay %d replays %d us_per_replay %.2f %s" % (gr.kernel_num(), replays, 1e3 * ms / replays, "eager" if eager else "hipGraph"))
gr.close()
