#!/usr/bin/env python3
"""A per-GPU batch as TWO concurrent half-batch graphs, each dispatched directly on its own HSA queue (round 4 measured this with hipGraph
replays on two HIP streams: +6-10 %; the single graph has since moved to direct dispatch).  Device-resident steps, outputs compared with
the one-graph run of the same images.
usage: split_batch_direct.py model batch dtype iters [direct=1]"""
import hashlib
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
from tengine_amd import capi, models, tm2  # noqa: E402


def main():
    name, batch, dtype, iters = sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
    direct = (int(sys.argv[5]) if len(sys.argv) > 5 else 1) == 1
    dt = {"uint8": tm2.DT_UINT8}.get(dtype, tm2.DT_INT8)

    def build(b):
        g = models.build(name, dtype, b, device_only=True)
        return g, capi.Graph(tm2.write_tm2(g), batch=b, direct_dispatch=direct)

    g1, whole = build(batch)
    x = models.synth_input(g1, 3, dt)
    whole.set_input(x)
    ref = [o.copy() for o in whole.run()]
    whole.upload(); whole.sync()

    def time_steps(graphs, n):
        for gr in graphs:
            gr.launch()
        for gr in graphs:
            gr.sync()
        t0 = time.perf_counter()
        for k in range(n):
            for gr in graphs:
                gr.launch()
            if k % 8 == 7:                   # bursts of eight passes per queue (the AQL ring is finite)
                for gr in graphs:
                    gr.sync()
        for gr in graphs:
            gr.sync()
        return (time.perf_counter() - t0) / n * 1e6

    t_whole = min(time_steps([whole], iters) for _ in range(3))
    halves = []
    outs = []
    for h in range(2):
        gh, gr = build(batch // 2)
        xh = np.ascontiguousarray(x[h * (batch // 2):(h + 1) * (batch // 2)])
        gr.set_input(xh)
        outs.append([o.copy() for o in gr.run()])
        gr.upload(); gr.sync()
        halves.append(gr)
    same = all(np.array_equal(np.concatenate([outs[0][i].reshape(batch // 2, -1), outs[1][i].reshape(batch // 2, -1)]), ref[i].reshape(batch, -1)) for i in range(len(ref)))
    t_half_seq = min(time_steps([halves[0]], iters) for _ in range(3))
    t_split = min(time_steps(halves, iters) for _ in range(3))
    print("%-14s %-6s batch %3d %s: one graph %8.1f us/step (%7.0f img/s) | one half alone %8.1f us | two halves concurrently %8.1f us/step (%7.0f img/s, %+5.1f %%) | outputs %s"
          % (name, dtype, batch, "direct dispatch" if direct else "hipGraph replay", t_whole, batch / t_whole * 1e6, t_half_seq, t_split, batch / t_split * 1e6,
             100 * (t_whole / t_split - 1), "identical" if same else "DIFFER"))


if __name__ == "__main__":
    main()
