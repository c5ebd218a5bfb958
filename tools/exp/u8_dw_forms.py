#!/usr/bin/env python3
"""uint8 depthwise 3x3 block heights (TAMD_PIN u8_dw_th = output rows per thread) on SSD-shaped layers: us per isolated launch."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import u8_conv_graph  # noqa: E402
from tengine_amd import capi, tm2  # noqa: E402

forms = sys.argv[1:] or ["1", "2", "4"]
SHAPES = [(16, 32, 150, 1), (16, 64, 150, 2), (16, 128, 75, 1), (16, 128, 75, 2), (16, 256, 38, 1), (16, 256, 38, 2), (16, 512, 19, 1), (16, 512, 19, 2),
          (16, 1024, 10, 1), (2, 512, 19, 1), (1, 32, 150, 1)]
print("%-26s" % "n x c @ hw, stride" + "".join("%10s" % f for f in forms))
for n, c, hw, s in SHAPES:
    g, x = u8_conv_graph(200 + c + hw + s, n, c, hw, hw, c, 3, s, 1, c, 0)
    b = tm2.write_tm2(g)
    cells = []
    for f in forms:
        os.environ["TAMD_PIN"] = "u8_dw_th=" + f          # 0: the launcher's own choice
        gr = capi.Graph(b)
        gr.set_input(x)
        gr.run()
        k = [q for q in gr.profile(30) if q["macs"] > 0][-1]
        cells.append("%10.2f" % (k["ms"] * 1e3))
        gr.close()
    del os.environ["TAMD_PIN"]
    print("%-26s" % ("%d x %d @ %d, s%d" % (n, c, hw, s)) + "".join(cells))
