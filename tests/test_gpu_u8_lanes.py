"""GPU parity of the lane-level chains for SMALL uint8 layers (u8_conv_patch.hip: conv_u8_lanes_k = conv_u8_patch_lane_main +
conv_u8_patch_tail): the 5x5 .. 1x1 ends of an SSD pyramid, where a GEMM launch is all set-up around a handful of live MFMA
columns.  Every output is one lane's fmaf chain in the reference's order -- the single chain over k for pixels j < (OH*OW)&~7,
the four k%4 chains + combine for the tail pixels (conv_kernel_x86.c:322-960) -- so the bytes must equal the oracle's (pinned
to the real reference by tests/test_uint8_oracle.py) and the GEMM member's.  TAMD_PIN="u8_patch=1,u8_patch_cfg=4" pins it."""
import os

import numpy as np
import pytest

from helpers import u8_conv_graph, u8_conv_pool_graph
from oracle import oracle
from tengine_amd import capi, models, tm2

pytestmark = pytest.mark.gpu


def run_with(g, x, pins):
    """pins: TAMD_PIN keys (csrc/env.h) -- u8_patch, u8_patch_cfg, u8_lanes"""
    from helpers import pinned
    with pinned(**pins):
        gr = capi.Graph(tm2.write_tm2(g))
        gr.set_input(x)
        out = [o.copy() for o in gr.run()]
        kernels = [k["kernel"] for k in gr.profile(1)]
        gr.close()
    return out, kernels


CASES = [
    # n, cin, h, w, cout, k, s, p, group, act, bias, dil
    (16, 128, 3, 3, 256, 3, 2, 1, 1, 0, True, 1),       # mssd conv16_2: 2x2 out, every pixel a tail pixel, K = 1152
    (16, 64, 2, 2, 128, 3, 2, 1, 1, 0, True, 1),        # mssd conv17_2: 1x1 out
    (16, 256, 2, 2, 64, 1, 1, 0, 1, 0, True, 1),        # mssd conv17_1: 1x1 conv on a 2x2 map
    (16, 128, 1, 1, 24, 1, 1, 0, 1, -1, True, 1),       # conv17_2_mbox_loc: 1x1 map, 24 couts (a tile and a half)
    (16, 128, 1, 1, 126, 1, 1, 0, 1, -1, True, 1),      # conv17_2_mbox_conf: 126 couts (cout % 4 == 2: the last rows combine differently)
    (16, 256, 5, 5, 128, 3, 2, 1, 1, 0, True, 1),       # mssd conv15_2: 3x3 out = 8 main pixels + 1 tail pixel per image
    (16, 512, 5, 5, 128, 1, 1, 0, 1, 0, True, 1),       # mssd conv15_1: 25 px = 24 main + 1 tail, K = 512
    (3, 256, 10, 10, 24, 1, 1, 0, 1, -1, True, 1),      # conv13_mbox_loc class: 100 px = 96 + 4
    (2, 32, 4, 7, 63, 3, 1, 1, 1, 6, True, 1),          # 28 px = 24 + 4, cout 63, relu6, K = 288
    (5, 16, 3, 3, 70, 1, 1, 0, 1, 6, False, 1),         # K = 16: one super-step; batch 5 (40 main pixels = 10 groups), no bias
    (1, 48, 6, 5, 20, 3, 1, 1, 1, 1, True, 1),          # 30 px = 24 + 6, relu1, one image
    (2, 16, 6, 6, 32, 3, 1, 2, 1, 0, True, 2),          # dilation 2, pad 2: 36 px = 32 + 4
    (1, 16, 1, 1, 16, 1, 1, 0, 1, -1, True, 1),         # the smallest layer there is: one pixel, one tile, one super-step
]


@pytest.mark.parametrize("case", CASES, ids=[str(c) for c in CASES])
def test_lane_chains_equal_oracle_and_gemm_member(case):
    n, cin, h, w, cout, k, s, p, group, act, bias, dil = case
    g, x = u8_conv_graph(311 + cin + cout + h, n, cin, h, w, cout, k, s, p, group, act, bias, dil)
    want = oracle.run_graph(g, x)
    got, kernels = run_with(g, x, {"u8_patch": "1", "u8_patch_cfg": "4"})
    assert any("conv_u8_lanes" in kn for kn in kernels), kernels
    ref, kernels0 = run_with(g, x, {"u8_patch": "0"})
    assert not any("conv_u8_lanes" in kn or "conv_u8_patch" in kn for kn in kernels0), kernels0
    for wv, a, b in zip(want, got, ref):
        a = a.reshape(wv.shape)
        bad = np.count_nonzero(a != wv)
        assert bad == 0, "%d / %d bytes differ from the oracle (max |d| %d)" % (bad, wv.size, np.abs(a.astype(int) - wv.astype(int)).max())
        assert np.array_equal(a, b.reshape(wv.shape))
        assert len(np.unique(wv)) > 3


@pytest.mark.parametrize("dims,kw", [((4, 32, 5, 5, 48, 3, 1), dict(slope=0.1)), ((3, 64, 6, 4, 20, 1, 0), dict(slope=0.0)), ((2, 16, 2, 2, 16, 3, 1), dict(slope=0.1, pool_k=2, pool_s=1))],
                         ids=["3x3 leaky 5x5", "1x1 relu 6x4", "3x3 leaky 2x2 (all tail pixels)"])
def test_lane_chains_with_a_fused_relu_node(dims, kw):
    """conv -> (leaky) ReLU node folded into the conv launch (the byte table of u8_epilogue.h) -> a pool that does NOT fuse
    (3x3 / stride 1), so the conv runs on the lanes kernel with relu.on"""
    kw = dict(dict(pool_k=3, pool_s=1), **kw)
    g, x = u8_conv_pool_graph(77 + dims[1] + dims[2], *dims, **kw)
    want = oracle.run_graph(g, x)
    got, kernels = run_with(g, x, {"u8_patch": "1", "u8_patch_cfg": "4"})
    assert any(kn.startswith("conv_u8_lanes") and "+relu" in kn for kn in kernels), kernels
    for wv, a in zip(want, got):
        assert np.array_equal(a.reshape(wv.shape), wv), "%d bytes differ" % np.count_nonzero(a.reshape(wv.shape) != wv)


def test_default_plan_of_the_ssd_tail_uses_lane_chains():
    """BASELINE configs[4] stand-in at batch 2: the pyramid's small layers take the lanes kernel by default, bytes unchanged"""
    g = models.build("mssd", "uint8", 2)
    x = models.synth_input(g, 5, tm2.DT_UINT8)
    want = oracle.run_graph(g, x)
    got, kernels = run_with(g, x, {})
    assert sum("conv_u8_lanes" in kn for kn in kernels) >= 6, kernels
    for wv, a in zip(want, got):
        assert np.array_equal(a.reshape(wv.shape), wv), "%d bytes differ" % np.count_nonzero(a.reshape(wv.shape) != wv)
    off, kernels0 = run_with(g, x, {"u8_lanes": "0"})
    assert not any("conv_u8_lanes" in kn for kn in kernels0), kernels0
    for a, b in zip(got, off):
        assert np.array_equal(a, b)


def test_large_layers_do_not_take_lane_chains():
    g, x = u8_conv_graph(5, 8, 64, 38, 38, 256, 3, 1, 1)          # 8 x 1444 px x 16 cout tiles: far beyond the wave bound
    got, kernels = run_with(g, x, {"u8_patch": "1", "u8_patch_cfg": "4"})
    assert not any("conv_u8_lanes" in kn for kn in kernels), kernels
    want = oracle.run_graph(g, x)
    assert np.array_equal(got[0].reshape(want[0].shape), want[0])
