#!/usr/bin/env python3
"""Latency experiments at batch 1: graph replay vs eager launches vs S concurrent graph instances (streams)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tengine_amd import capi, models, tm2  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
g = models.build("mobilenet_v1", "int8", batch)
b = tm2.write_tm2(g)
x = models.synth_input(g, 3)


def mk(graph=True):
    gr = capi.Graph(b, batch=batch, use_hip_graph=graph)
    gr.set_input(x)
    gr.run()
    return gr


def timeit(grs, iters):
    for gr in grs:
        gr.launch()
    for gr in grs:
        gr.sync()
    t0 = time.perf_counter()
    for i in range(iters):
        grs[i % len(grs)].launch()
    for gr in grs:
        gr.sync()
    return (time.perf_counter() - t0) / iters * 1e6


for mode in (True, False):
    gr = mk(mode)
    print("graph=%s single stream: %.1f us/step (host-timed), %.1f us/step (events)" % (mode, timeit([gr], 300), gr.time_launches(300) / 300 * 1e3))
    gr.close()
for s in (2, 4, 8, 16):
    grs = [mk(True) for _ in range(s)]
    us = timeit(grs, 600)
    print("streams=%d: %.1f us/step -> %.0f img/s" % (s, us, batch * 1e6 / us))
    for gr in grs:
        gr.close()
