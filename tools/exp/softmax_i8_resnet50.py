"""ResNet-50 int8 batch 32 WITH its Softmax (the benchmark graph): launch count, softmax_i8's own time, the pass with and
without it -- on one box (round 4, after the evidence pass)."""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tengine_amd import capi, models, plans, tm2

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
tmp = tempfile.mkdtemp()
for full in (False, True):
    plan = os.path.join(tmp, "plan_%d.txt" % full)
    plans.seed(plan, "resnet50", "int8", batch)
    os.environ["TAMD_PLAN_CACHE"] = plan
    g = models.build("resnet50", "int8", batch, device_only=not full)
    gr = capi.Graph(tm2.write_tm2(g), direct_dispatch=True)
    gr.set_input(models.synth_input(g, 5))
    out = gr.run()[0]
    for _ in range(3):
        gr.time_launches(20)
    ms = min(gr.time_launches(50) for _ in range(5)) / 50.0
    prof = gr.profile(20)
    print("resnet50 int8 b%d %s: %d launches, %.1f us / pass ; tail: %s" % (
        batch, "with prob (softmax_i8)" if full else "logits only", len(prof), 1e3 * ms,
        ", ".join("%s %.2f us" % (k["kernel"], 1e3 * k["ms"]) for k in prof[-3:])))
    if full:
        print("prob: rows sum to %s (int8 steps of %.5f), argmax %s" % (
            out.reshape(batch, -1).astype(int).sum(axis=1)[:4].tolist(), g.tensors[g.nodes[g.output_nodes[0]].outputs[0]].scales[0],
            out.reshape(batch, -1).argmax(axis=1)[:4].tolist()))
    gr.close()
