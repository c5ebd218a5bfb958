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
    (3, 64, 27, 31, 200, 1, 1, 0, 1, -1, True, 1),      # pointwise, pixels not a multiple of 32, channels not a multiple of 128
    (2, 96, 40, 40, 256, 1, 1, 0, 1, 6, False, 1),      # two 128-channel groups, relu6 window, no bias
]
MEMBERS = ["igemm0", "igemm1", "igemm2", "igemm3", "igemm4", "igemm5", "igemm6", "igemm7", "igemm8", "igemm9", "igemm10", "igemm11",
           "igemm12", "igemm13", "igemm14", "igemm15", "gemm_direct", "pw_stream", "pw_rows",
           "conv_igemm2", "pw_small", "conv_pgemm_i8<128x64", "conv_pgemm_i8<128x128", "conv_pgemm_i8<64x64", "conv_pgemm_i8<64x128"]


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
        if member == "pw_small":      # the 1x1 shapes with <= 4096 pixels must really have run it
            assert used >= 5, used
        if member == "pw_rows":       # 1x1, cin <= 128, >= 2048 pixels
            assert used >= 4, used
        if member.startswith("conv_pgemm"):      # every 1x1 shape with >= 32 channels, every k x k shape with cin % 64 == 0
            assert used >= (8 if "<64x64" in member else 5), used
    finally:
        del os.environ["TAMD_FORCE_GEMM"]


@pytest.mark.parametrize("member", ["pw_stream", "pw_rows", "igemm0", "igemm2", "igemm10", "igemm14", "conv_igemm2", "conv_pgemm_i8<128x64",
                                    "conv_pgemm_i8<64x64"])
@pytest.mark.parametrize("etype,own_relu_scale", [(tm2.ELT_SUM, False), (tm2.ELT_SUB, True)])
def test_fused_eltwise_tail_is_exact_in_every_member(member, etype, own_relu_scale):
    """conv -> eltwise -> ReLU folded into the epilogue of each family member that offers it (ResNet block tails);
    4 x 64 x 28 x 28 makes the streaming pointwise kernel eligible too."""
    from helpers import eltwise_relu_graph
    g, x = eltwise_relu_graph(23, 4, 64, 28, 28, True, etype)
    if own_relu_scale:
        r = [t for t in g.tensors if t.name == "relu"][0]
        r.scales = [float(np.float32(r.scales[0] * 0.83))]
    want = oracle.run_graph(g, x)[0]
    os.environ["TAMD_FORCE_GEMM"] = member
    try:
        gr = capi.Graph(tm2.write_tm2(g))
    finally:
        del os.environ["TAMD_FORCE_GEMM"]
    gr.set_input(x)
    got = gr.run()[0].reshape(want.shape)
    names = [k["kernel"] for k in gr.profile(1)]
    gr.close()
    assert np.array_equal(got, want), names
    assert any("+eltwise" in k for k in names), names
    assert len(np.unique(want)) > 3


@pytest.mark.parametrize("member", ["pw_stream", "pw_rows", "igemm0", "igemm10", "conv_pgemm"])
@pytest.mark.parametrize("variant", ["sum_relu_fold", "sum_norelu_fold", "sum_relu_scales_too_wide", "sum_relu_fold_disabled"])
def test_residual_tail_two_fma_form_and_its_fallbacks(member, variant, monkeypatch):
    """the SUM (+ scale-keeping ReLU) tail runs as two fused multiply-adds per value when (s_conv + s_res) / s_out <= 2
    (epilogue.h: elt_sum4_fold, planner fold in graph_plan.hip); wider scale ratios and TAMD_PIN elt_fold=0 take the general tail.
    Same bytes either way, all against the oracle."""
    from helpers import eltwise_relu_graph
    g, x = eltwise_relu_graph(41, 6, 64, 28, 28, variant != "sum_norelu_fold", tm2.ELT_SUM)
    if variant == "sum_relu_scales_too_wide":
        sa = [t for t in g.tensors if t.name == "out"][0].scales[0]
        for t in g.tensors:
            if t.name in ("sum", "relu"):
                t.scales = [float(np.float32(sa * 0.9))]       # (1 + 1.37) / 0.9 > 2
    if variant == "sum_relu_fold_disabled":
        monkeypatch.setenv("TAMD_PIN", "elt_fold=0")
    want = oracle.run_graph(g, x)[0]
    monkeypatch.setenv("TAMD_FORCE_GEMM", member)
    gr = capi.Graph(tm2.write_tm2(g))
    gr.set_input(x)
    got = gr.run()[0].reshape(want.shape)
    names = [k["kernel"] for k in gr.profile(1)]
    gr.close()
    assert np.array_equal(got, want), names
    assert any("+eltwise" in k and (member if member == "conv_pgemm" else member.replace("igemm", "conv_igemm")[:8]) in k for k in names), names
    assert len(np.unique(want)) > 20
