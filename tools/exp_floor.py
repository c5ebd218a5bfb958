#!/usr/bin/env python3
"""Measures the per-launch floor of a dependent kernel chain on this GPU: a graph of N tiny 1x1 convolutions
(16 channels, 1 pixel) replayed as one hipGraph -- the work is nil, what remains is the kernel boundary."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tengine_amd import capi, tm2  # noqa: E402
from tengine_amd.tm2 import DT_INT8, DT_INT32, Graph  # noqa: E402


def chain(n_layers, c=16, hw=1):
    g = Graph(name="chain")
    x = g.add_input("data", [1, c, hw, hw], DT_INT8, [0.05], [0])
    rng = np.random.default_rng(0)
    for i in range(n_layers):
        w = g.add_const("w%d" % i, rng.integers(-5, 6, size=(c, c, 1, 1)).astype(np.int8), DT_INT8, [0.01] * c, [0] * c)
        y = g.add_tensor("y%d" % i, [1, c, hw, hw], DT_INT8, tm2.TT_VAR, None, [0.05], [0])
        ni = g.add_node("c%d" % i, "Convolution", [x, w], [y], kernel_h=1, kernel_w=1, stride_h=1, stride_w=1,
                        dilation_h=1, dilation_w=1, input_channel=c, output_channel=c, group=1, activation=0,
                        pad_h0=0, pad_w0=0, pad_h1=0, pad_w1=0)
        x = y
    g.output_nodes = [ni]
    return g


for n in (1, 8, 29, 58):
    g = chain(n)
    gr = capi.Graph(tm2.write_tm2(g))
    gr.set_input(np.ones((1, 16, 1, 1), np.int8))
    gr.run()
    us = gr.time_launches(300) / 300 * 1e3
    print("chain of %2d trivial launches: %.1f us/replay -> %.2f us per launch" % (n, us, us / n))
    gr.close()
