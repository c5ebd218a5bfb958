#!/usr/bin/env python3
"""The blocking host-to-host run (what tm_benchmark times) under the variants of its list, interleaved on one box; each graph
prints where the host side of a run spends its time when it is destroyed (TAMD_H2H_TRACE=1).
usage: h2h_variants.py [model] [batch] [runs] [rounds]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402

from tengine_amd import capi, models, tm2  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "mobilenet_v1"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1
runs = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 5
os.environ["TAMD_H2H_TRACE"] = "1"
g = models.build(name, "int8", batch, device_only=(name != "mobilenet_v1"))
tmb = tm2.write_tm2(g)
x = models.synth_input(g, 3)
VARIANTS = [("round-3 list (download launch, barrier packet)", {"TAMD_IO_ZERO_COPY": "0", "TAMD_DIRECT_CLOSE_ON_LAST": "0"}),
            ("outputs to pinned host + close on last (default)", {}),
            ("+ first launch reads the pinned input (no upload)", {"TAMD_IO_ZERO_COPY_IN": "1"}),
            ("hipGraph", None)]
if os.environ.get("H2H_ONLY"):
    VARIANTS = [v for i, v in enumerate(VARIANTS) if str(i) in os.environ["H2H_ONLY"].split(",")]
graphs, ref = [], None
for nm, env in VARIANTS:
    for k, v in (env or {}).items():
        os.environ[k] = v
    gr = capi.Graph(tmb, batch=batch, direct_dispatch=env is not None)
    for k in (env or {}):
        del os.environ[k]
    gr.set_input(x)
    out = gr.run()[0].copy()
    ref = out if ref is None else ref
    assert np.array_equal(ref, out), nm
    for _ in range(50):
        gr.run_noreturn()
    graphs.append((nm, gr))
res = {nm: [] for nm, _ in graphs}
pipe = {nm: [] for nm, _ in graphs}
for r in range(rounds):
    for nm, gr in graphs:
        ts = []
        for _ in range(runs):
            t0 = time.perf_counter()
            gr.run_noreturn()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        res[nm].append(1e6 * ts[len(ts) // 2])
        outs2 = [gr.output_like(), gr.output_like()]
        gr.run_async(outs2[0])
        tp = time.perf_counter()
        for k in range(runs):
            gr.run_async(outs2[(k + 1) & 1])
            gr.wait()
        gr.wait()
        pipe[nm].append(1e6 * (time.perf_counter() - tp) / runs)
        gr.bind_default_outputs()
print("== %s int8 b%d blocking tamd_graph_run: median us per run (%d runs) per round, and two-in-flight us per run" % (name, batch, runs))
for nm, gr in graphs:
    print("  %-50s blocking %s | pipelined %s | packets %d" % (nm, " ".join("%.1f" % v for v in res[nm]), " ".join("%.1f" % v for v in pipe[nm]), gr.direct_packets()))
    sys.stdout.flush()
    print("  ^ host-side anatomy of this variant:", file=sys.stderr)
    gr.close()
