#!/usr/bin/env python3
"""How many parts?  The same batch as 1, 2, 4 (and 8) graphs of batch / parts images each, every graph one launch list on its own HSA queue
(split_batch = 1), submitted side by side: device-resident launch() x steps + sync(), interleaved in one process, outputs compared with the
one-graph form.  Decides whether csrc/graph_pair.hip should know more than halves.   usage: split_parts.py <model> <batch> [steps [regions]]"""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tengine_amd import capi, models, tm2  # noqa: E402


def main():
    model, batch = sys.argv[1], int(sys.argv[2])
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    regions = int(sys.argv[4]) if len(sys.argv) > 4 else 5
    os.environ["TAMD_PLAN_CACHE"] = os.path.join(tempfile.gettempdir(), "split_parts_%d.txt" % os.getpid())
    x = None
    forms = {}
    for parts in (1, 2, 4, 8):
        if batch % parts or batch // parts < 2:
            continue
        g = models.build(model, "int8", batch // parts)
        b = tm2.write_tm2(g)
        if x is None:
            x = models.synth_input(g, 1000, tm2.DT_INT8)
        grs = []
        for k in range(parts):
            gr = capi.Graph(b, batch=batch // parts, direct_dispatch=True, split_batch=1)
            n = batch // parts
            gr.set_input(np.ascontiguousarray(x[k * n:(k + 1) * n])); gr.upload(); gr.sync()
            grs.append(gr)
        for _ in range(10):
            for gr in grs:
                gr.launch()
        for gr in grs:
            gr.sync()
        forms[parts] = grs
    best = {p: 1e9 for p in forms}
    for _ in range(regions):
        for p, grs in forms.items():
            t0 = time.perf_counter()
            for _ in range(steps):
                for gr in grs:
                    gr.launch()
            for gr in grs:
                gr.sync()
            best[p] = min(best[p], (time.perf_counter() - t0) / steps)
    outs = {p: np.concatenate([gr.download()[0].reshape(batch // p, -1) for gr in grs]) for p, grs in forms.items()}
    print("%-14s int8 batch %3d, device-resident, us per step (min of %d x %d, interleaved; plans timed in this process): %s | outputs %s" % (
        model, batch, regions, steps, " | ".join("%d part%s %8.1f (%+5.1f %%)" % (p, "s" if p > 1 else " ", 1e6 * best[p], 100.0 * (best[1] / best[p] - 1.0)) for p in forms),
        "identical" if all(np.array_equal(outs[1], o) for o in outs.values()) else "DIFFER"))
    for grs in forms.values():
        for gr in grs:
            gr.close()
    if os.path.exists(os.environ["TAMD_PLAN_CACHE"]):
        os.remove(os.environ["TAMD_PLAN_CACHE"])


if __name__ == "__main__":
    main()
