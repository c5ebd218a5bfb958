"""conv_pgemm.hip (lean-loop implicit GEMM: fragment-ordered weights by LDS-DMA, k x k activations as an LDS-resident input
patch) against the oracle on the geometry its patch addressing has to get right: pixel tiles that straddle rows and IMAGES,
halo rows of neighbouring images, strides, dilation, asymmetric maps, pads larger than the halo, ragged M / cout, the 64-channel
chunk walk with two patch buffers, tails of K, and the fused eltwise epilogue.  Every tile variant is pinned by name."""
import os

import numpy as np
import pytest

from helpers import conv_graph, eltwise_relu_graph
from oracle import oracle
from tengine_amd import capi, tm2

pytestmark = pytest.mark.gpu

VARIANTS = ["conv_pgemm_i8<128x64", "conv_pgemm_i8<128x128", "conv_pgemm_i8<64x64", "conv_pgemm_i8<64x128"]
# conv_pgemm_w.hip (round 5), the 3x3 kernels: table-driven set-up on four waves (w4t) or eight (w8: 128 x 128 tiles), and the forms
# with ONE barrier per filter row (b3: a 9-slot ring, the waves drift up to two stages apart).  The forms that lost on every
# ResNet-50 shape -- two K groups adding into one LDS tile (ks2w), eight waves with copy roles over 64 couts, the 2 x 4 wave grid --
# are built for tools/exp/pgemm_anatomy.hip only; their parity runs of round 5 are profiles/r05_pytest_pgemm_all_forms.txt.
WGRID = ["conv_pgemm_i8<128x128,3x3,w8>", "conv_pgemm_i8<128x128,3x3,w8b3>",
         "conv_pgemm_i8<128x64,3x3,w4t>", "conv_pgemm_i8<64x64,3x3,w4t>", "conv_pgemm_i8<128x64,3x3,w4b3>", "conv_pgemm_i8<64x64,3x3,w4b3>"]

# n, cin, h, w, cout, k, s, p, group, act, bias, dil
PATCH_CASES = [
    (5, 64, 7, 7, 96, 3, 1, 1, 1, 0, True, 1),        # 49-pixel images: a 128-pixel tile spans three of them (+ their halo rows)
    (3, 128, 14, 14, 130, 3, 1, 1, 1, 0, True, 1),    # two chunks of 64 channels (both patch buffers), cout ragged
    (2, 192, 10, 13, 64, 3, 1, 1, 1, 6, True, 1),     # three chunks, relu6 window, odd map
    (4, 64, 15, 15, 72, 3, 2, 1, 1, 0, True, 1),      # stride 2
    (2, 64, 20, 20, 40, 3, 1, 2, 1, 0, True, 2),      # dilation 2, pad 2
    (2, 64, 12, 9, 48, 3, 1, 0, 1, -1, False, 1),     # no pad, no bias, no activation
    (1, 64, 9, 9, 32, 5, 1, 2, 1, 0, True, 1),        # 5 x 5
    (2, 64, 11, 11, 80, 3, 1, 3, 1, 0, True, 1),      # pad larger than the halo: whole border ring of zeros
    (1, 256, 56, 56, 64, 3, 1, 1, 1, 0, True, 1),     # wide rows: > 256 patch pixels (two pieces per wave)
    (7, 64, 5, 5, 64, 3, 1, 1, 1, 0, True, 1),        # 25-pixel images, M = 175 (ragged last tile)
    (9, 512, 7, 7, 96, 3, 1, 1, 1, 0, True, 1),       # eight chunks (four per wave group with ks2), tiles over three images
    (3, 128, 17, 17, 256, 3, 2, 1, 1, 0, True, 1),    # stride 2, two cout tiles of 128
]
ROW_CASES = [
    (3, 64, 27, 31, 200, 1, 1, 0, 1, -1, True, 1),    # M = 2511 (ragged), cout ragged
    (2, 24, 10, 10, 36, 1, 1, 0, 1, 0, True, 1),      # K tail: 32 padded channels in one 64-deep stage
    (2, 100, 9, 9, 64, 1, 1, 0, 1, 0, True, 1),       # K tail in the second stage (ckp = 112)
    (2, 256, 14, 14, 512, 1, 2, 0, 1, -1, True, 1),   # stride-2 pointwise (ResNet branch1)
    (1, 1024, 14, 14, 256, 1, 1, 0, 1, 0, True, 1),   # deep K: 16 stages through a 4-slot ring
    (32, 64, 3, 3, 64, 1, 1, 0, 1, 0, True, 1),
]


def _run(member, case, seed):
    n, cin, h, w, cout, k, s, p, group, act, bias, dil = case
    g, x = conv_graph(seed, n, cin, h, w, cout, k, s, p, group, act, bias, dil)
    want = oracle.run_graph(g, x)[0]
    os.environ["TAMD_FORCE_GEMM"] = member
    os.environ["TAMD_AUTOTUNE"] = "0"
    if member in VARIANTS:       # the generic patch kernel is offered a 3x3 layer only where the dedicated kernels are switched off
        os.environ["TAMD_PGEMM_W"] = "0"
    try:
        gr = capi.Graph(tm2.write_tm2(g))
    finally:
        del os.environ["TAMD_FORCE_GEMM"]
        del os.environ["TAMD_AUTOTUNE"]
        os.environ.pop("TAMD_PGEMM_W", None)
    gr.set_input(x)
    got = gr.run()[0].reshape(want.shape)
    name = gr.profile(1)[-1]["kernel"]
    gr.close()
    return want, got, name


@pytest.mark.parametrize("member", VARIANTS + WGRID)
@pytest.mark.parametrize("ci", range(len(PATCH_CASES)))
def test_patch_kernel_is_exact(member, ci):
    want, got, name = _run(member, PATCH_CASES[ci], 900 + ci)
    if member in name:       # a variant that does not apply to the shape (e.g. 128-pixel tiles on 64 pixels) falls back
        assert ("patch" in name) if member in VARIANTS else ("3x3" in name), name
    assert np.array_equal(got, want), (name, PATCH_CASES[ci], int((got != want).sum()))
    assert len(np.unique(want)) > 5


@pytest.mark.parametrize("member", VARIANTS)
@pytest.mark.parametrize("ci", range(len(ROW_CASES)))
def test_pointwise_kernel_is_exact(member, ci):
    want, got, name = _run(member, ROW_CASES[ci], 950 + ci)
    if member in name:
        assert "rows" in name, name
    assert np.array_equal(got, want), (name, ROW_CASES[ci], int((got != want).sum()))


def test_variants_really_run():
    """the pinned names must be the kernels that ran on the shapes they accept"""
    for member in VARIANTS:
        for case in (PATCH_CASES[1], ROW_CASES[0]):
            _, _, name = _run(member, case, 1)
            assert member in name, (member, name)
    for member in (VARIANTS[0], VARIANTS[2]):       # dilated 3x3 with 40 couts: the 64-channel-wide tiles only
        _, _, name = _run(member, PATCH_CASES[4], 1)
        assert member in name, (member, name)
    for member in WGRID:
        # 14 x 14 x 128 -> 130 (two chunks, ragged cout), 7 x 7 x 512 (eight chunks, tiles over three images)
        for case in (PATCH_CASES[1], PATCH_CASES[10]):
            _, _, name = _run(member, case, 1)
            assert member in name, (member, name)
    for member in [m for m in WGRID if "<64x" in m]:       # stride 2 on 17 x 17 with 256 couts: the patch of a 128-pixel tile is > 512 units, 64-pixel tiles take it
        _, _, name = _run(member, PATCH_CASES[11], 1)
        assert member in name, (member, name)
    for member in [m for m in WGRID if "x64," in m]:       # the 64-cout tiles also take the narrow layers: dilation 2, pad > halo, 25-pixel images, wide rows
        for case in (PATCH_CASES[4], PATCH_CASES[7], PATCH_CASES[9], PATCH_CASES[8]):
            _, _, name = _run(member, case, 1)
            assert member in name, (member, name)


@pytest.mark.parametrize("member", WGRID)
@pytest.mark.parametrize("relu,etype", [(True, tm2.ELT_SUM), (False, tm2.ELT_SUM), (True, tm2.ELT_MAX)])
def test_wave_grid_kernels_fused_residual_tail(member, relu, etype):
    """3 x 3 convolution + eltwise (+ ReLU) in the wave-grid kernels' epilogues: from registers (w8) and from the LDS partial-sum tile (ks2w)"""
    g, x = eltwise_relu_graph(78, 5, 128, 14, 14, relu, etype, k=3)
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
    assert any("+eltwise" in k and member in k for k in names), names


@pytest.mark.parametrize("member", ["conv_pgemm_i8<128x64", "conv_pgemm_i8<64x64"])
def test_pgemm_fused_residual_tail(member):
    g, x = eltwise_relu_graph(77, 5, 64, 14, 14, True, tm2.ELT_SUM)
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
    assert any("+eltwise" in k and member in k for k in names), names
