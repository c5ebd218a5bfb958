#!/usr/bin/env python3
"""One tamd_graph, two forms: the batch as ONE launch list (tamd_options.split_batch = 1) against TWO half-batch device graphs side by side
behind the same handle (= 2), device-resident launch() x steps + sync(), regions interleaved in one process on one box, outputs compared.
Decides the default rule's batch threshold (csrc/graph_pair.hip).   usage: split_ab.py <model> <dtype> <batch> [steps [regions]]"""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tengine_amd import capi, models, plans, tm2  # noqa: E402


def main():
    model, dtype, batch = sys.argv[1], sys.argv[2], int(sys.argv[3])
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 100
    regions = int(sys.argv[5]) if len(sys.argv) > 5 else 5
    plan = os.path.join(tempfile.gettempdir(), "split_ab_%d.txt" % os.getpid())
    shipped = plans.seed(plan, model, dtype, batch)
    os.environ["TAMD_PLAN_CACHE"] = plan
    g = models.build(model, dtype, batch)
    b = tm2.write_tm2(g)
    x = models.synth_input(g, 1000, tm2.DT_UINT8 if dtype == "uint8" else tm2.DT_INT8)
    grs = [capi.Graph(b, batch=batch, direct_dispatch=True, split_batch=m) for m in (1, 2)]
    assert grs[0].halves() == 0
    if grs[1].halves() != 2:
        print("%-14s %-5s batch %3d: the graph cannot be halved" % (model, dtype, batch))
        return
    for gr in grs:
        gr.set_input(x); gr.upload(); gr.sync()
        for _ in range(10):
            gr.launch()
        gr.sync()
    best = [1e9, 1e9]
    for _ in range(regions):
        for k, gr in enumerate(grs):
            t0 = time.perf_counter()
            for _ in range(steps):
                gr.launch()
            gr.sync()
            best[k] = min(best[k], (time.perf_counter() - t0) / steps)
    same = all(np.array_equal(a, c) for a, c in zip(grs[0].download(), grs[1].download()))
    print("%-14s %-5s batch %3d, device-resident, us per step (min of %d x %d, interleaved): one launch list %8.1f | two half-batch graphs behind one handle %8.1f (%+5.1f %%) | outputs %s | plan %s"
          % (model, dtype, batch, regions, steps, 1e6 * best[0], 1e6 * best[1], 100.0 * (best[0] / best[1] - 1.0), "identical" if same else "DIFFER", shipped))
    for gr in grs:
        gr.close()
    if os.path.exists(plan):
        os.remove(plan)


if __name__ == "__main__":
    main()
