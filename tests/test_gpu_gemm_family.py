"""Every member of the int8 GEMM kernel family must produce the same bytes on every shape it accepts: the planner's
autotune picks among them by speed alone.  TAMD_FORCE_GEMM pins one member at prerun."""
import os

import numpy as np
import pytest

from helpers import conv_graph, fc_graph
from oracle import oracle
from tengine_amd import capi, tm2

pytestmark = pytest.mark.gpu

SHAPES = [
    # n, cin, h, w, cout, k, s, p, group, act, bias, dil
    (1, 32, 112, 112, 64, 1, 1, 0, 1, 0, True, 1),
    (1, 128, 56, 56, 128, 1, 1, 0, 1, 0, True, 1),
    (1, 256, 28, 28, 256, 1, 1, 0, 1, 0, True, 1),
    (1, 512, 14, 14, 512, 1, 1, 0, 1, 0, True, 1),
    (1, 512, 7, 7, 1024, 1, 1, 0, 1, 0, True, 1),
    (1, 1024, 1, 1, 1000, 1, 1, 0, 1, -1, True, 1),
    (2, 64, 56, 56, 64, 3, 1, 1, 1, 0, True, 1),
    (4, 128, 28, 28, 128, 3, 1, 1, 1, 0, True, 1),
    (1, 256, 14, 14, 512, 1, 2, 0, 1, -1, True, 1),
    (3, 48, 9, 7, 40, 3, 2, 1, 1, 6, True, 1),
    (1, 20, 10, 10, 36, 1, 1, 0, 1, 0, True, 1),
    (2, 32, 12, 12, 16, 3, 1, 2, 1, 0, True, 2),
    (9, 80, 33, 31, 96, 3, 1, 2, 1, 0, True, 2),
    (1, 512, 7, 7, 512, 3, 1, 1, 1, 0, True, 1),
]
MEMBERS = ["igemm0", "igemm1", "igemm2", "igemm3", "igemm4", "igemm5", "igemm6", "igemm7", "igemm8", "igemm9", "gemm_direct", "pw_stream",
           "conv_igemm2"]


@pytest.fixture(scope="module")
def cases():
    out = []
    for c in SHAPES:
        n, cin, h, w, cout, k, s, p, group, act, bias, dil = c
        g, x = conv_graph(500 + cin + cout + h, n, cin, h, w, cout, k, s, p, group, act, bias, dil)
        out.append((c, tm2.write_tm2(g), x, oracle.run_graph(g, x)[0]))
    g, x = fc_graph(5, 32, (2048,), 1000)
    out.append((("fc", 32, 2048, 1000), tm2.write_tm2(g), x, oracle.run_graph(g, x)[0]))
    return out


@pytest.mark.parametrize("member", MEMBERS)
def test_family_member_is_exact_on_every_shape(member, cases):
    os.environ["TAMD_FORCE_GEMM"] = member
    try:
        used = 0
        for c, b, x, want in cases:
            gr = capi.Graph(b)
            gr.set_input(x)
            got = gr.run()[0].reshape(want.shape)
            name = gr.profile(1)[-1]["kernel"]
            gr.close()
            used += member.replace("igemm", "x") in name or member in name
            assert np.array_equal(got, want), "%s (%s) wrong on %s" % (member, name, c)
    finally:
        del os.environ["TAMD_FORCE_GEMM"]
