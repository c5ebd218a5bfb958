#!/usr/bin/env python3
"""int8 depthwise 3x3 launch forms (TAMD_PIN dw_form = <fragments per row><output rows per lane>) on MobileNet-shaped layers at
large batch: us per isolated launch (HIP events).  usage: dw_forms.py [forms ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import conv_graph  # noqa: E402
from tengine_amd import capi, tm2  # noqa: E402

forms = sys.argv[1:] or ["11", "12", "21", "22"]
SHAPES = [(16, 64, 112, 1), (16, 64, 112, 2), (32, 128, 56, 1), (32, 128, 56, 2), (64, 256, 28, 1), (64, 256, 28, 2), (64, 512, 14, 1), (64, 512, 14, 2),
          (64, 1024, 7, 1), (8, 512, 14, 1), (1, 512, 14, 1)]
print("%-26s" % "n x c @ hw, stride" + "".join("%10s" % f for f in forms))
for n, c, hw, s in SHAPES:
    g, x = conv_graph(200 + c + hw + s, n, c, hw, hw, c, 3, s, 1, group=c, act=0)
    b = tm2.write_tm2(g)
    cells = []
    for f in forms:
        os.environ["TAMD_PIN"] = "dw_form=" + f
        gr = capi.Graph(b)
        gr.set_input(x)
        gr.run()
        k = [q for q in gr.profile(30) if q["macs"] > 0][-1]
        cells.append("%10.2f" % (k["ms"] * 1e3))
        gr.close()
    del os.environ["TAMD_PIN"]
    print("%-26s" % ("%d x %d @ %d, s%d" % (n, c, hw, s)) + "".join(cells))
