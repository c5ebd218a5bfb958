#!/usr/bin/env python3
"""A/B of environment variants on the device-resident step of one config, interleaved on ONE box (boxes of the pool differ by
20 %): every variant is a fresh graph pre-run under its own environment with the SAME plan file (the first prerun measures, the
others take its choices), the step is timed as the bench times it (direct AQL passes, host clock around submit .. complete).

usage: ab_step.py model batch dtype iters rounds  NAME=ENV1=V1,ENV2=V2  NAME=...      (NAME alone: no extra environment)"""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tengine_amd import capi, models, tm2  # noqa: E402

name, batch, dtype, iters, rounds = sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
variants = []
for spec in sys.argv[6:]:
    nm, _, envs = spec.partition("=")
    variants.append((nm, dict(e.split("=", 1) for e in envs.split(",") if e)))
g = models.build(name, dtype, batch, device_only=True)
tmb = tm2.write_tm2(g)
x = models.synth_input(g, 3, {"uint8": tm2.DT_UINT8, "fp32": tm2.DT_FP32}.get(dtype, tm2.DT_INT8))
os.environ.setdefault("TAMD_PLAN_CACHE", os.path.join(tempfile.gettempdir(), "ab_plan_%s_%s_b%d.txt" % (name, dtype, batch)))
graphs, ref = [], None
for nm, env in variants:
    for k, v in env.items():
        os.environ[k] = v
    gr = capi.Graph(tmb, batch=batch, direct_dispatch=True)
    for k in env:
        del os.environ[k]
    gr.set_input(x)
    out = [o.copy() for o in gr.run()]
    if ref is None:
        ref = out
    same = all((a == b).all() for a, b in zip(ref, out))
    gr.upload()
    gr.sync()
    gr.time_launches(max(3, iters // 10))
    graphs.append((nm, gr, same, gr.kernel_num(), gr.direct_packets(), gr.prerun_ms()))
res = {nm: [] for nm, *_ in graphs}
for r in range(rounds):
    for nm, gr, *_ in graphs:
        res[nm].append(1e3 * gr.time_launches(iters) / iters)
print("== %s %s b%d: us per step, %d rounds of %d steps, interleaved" % (name, dtype, batch, rounds, iters))
for nm, gr, same, kn, pk, pms in graphs:
    v = sorted(res[nm])
    print("  %-28s min %9.2f  median %9.2f  max %9.2f | launches %d packets %d prerun %.0f ms | same bytes as first variant: %s"
          % (nm, v[0], v[len(v) // 2], v[-1], kn, pk, pms, same))
    gr.close()
