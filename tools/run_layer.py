#!/usr/bin/env python3
"""One int8 conv layer, one pinned member of the GEMM family, a few eager launches -- the thing to put under
`rocprofv3 --pmc` (separate passes, no tracing) when the question is what ONE kernel on ONE shape is waiting for.
usage: run_layer.py cin hw cout k batch member [iters] [eltwise: 0|1]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import conv_graph, eltwise_relu_graph  # noqa: E402
from tengine_amd import capi, tm2  # noqa: E402


def main():
    cin, hw, cout, k, batch = (int(v) for v in sys.argv[1:6])
    member = sys.argv[6]
    iters = int(sys.argv[7]) if len(sys.argv) > 7 else 5
    elt = len(sys.argv) > 8 and sys.argv[8] == "1"
    os.environ["TAMD_AUTOTUNE"] = "0"
    os.environ["TAMD_FORCE_GEMM"] = member
    if elt:
        g, x = eltwise_relu_graph(1, batch, cin, hw, hw, True, tm2.ELT_SUM)
    else:
        g, x = conv_graph(1, batch, cin, hw, hw, cout, k, 1, k // 2, 1, 0, True, 1)
    gr = capi.Graph(tm2.write_tm2(g), use_hip_graph=False)
    gr.set_input(x)
    for _ in range(iters):
        gr.run()
    prof = gr.profile(10)
    for q in prof:
        print("%-28s %-40s %8.2f us" % (q["node"], q["kernel"], q["ms"] * 1e3))
    gr.close()


if __name__ == "__main__":
    main()
