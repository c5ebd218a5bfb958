"""GPU parity of the uint8 patch convolution (u8_conv_patch.hip: conv_u8_patch_k) -- the 3x3 / 1x1 member that dequantises the
input patch of a pixel tile once into LDS and walks the reference's k order (channel-major, tap-minor; conv_kernel_x86.c im2col)
through the fp32 MFMA chain.  It is an autotune candidate next to conv_u8_mfma_*; here TAMD_PIN u8_patch=1 pins it wherever it
applies, and the bytes must equal the oracle's (pinned to the real reference by tests/test_uint8_oracle.py) and the GEMM member's."""
import numpy as np
import pytest

from helpers import u8_conv_graph, u8_conv_pool_graph
from oracle import oracle
from tengine_amd import capi, models, tm2

pytestmark = pytest.mark.gpu


def run_with(g, x, patch, batch=None, **pins):
    """patch: the TAMD_PIN value of u8_patch (csrc/env.h): "1" pins conv_u8_patch wherever it applies, "0" keeps it out"""
    from helpers import pinned
    with pinned(u8_patch=patch, **pins):
        gr = capi.Graph(tm2.write_tm2(g))
        gr.set_input(x)
        out = [o.copy() for o in gr.run()]
        kernels = [k["kernel"] for k in gr.profile(1)]
        gr.close()
    return out, kernels


CASES = [
    # n, cin, h, w, cout, k, s, p, group, act, bias, dil
    (1, 16, 40, 40, 32, 3, 1, 1, 1, -1, True, 1),       # one 32-row cout tile, 1600 px = 25 pixel tiles
    (1, 64, 20, 20, 128, 3, 1, 1, 1, 0, True, 1),       # 400 px: ragged last pixel tile (16 px)
    (2, 128, 13, 13, 255, 1, 1, 0, 1, -1, True, 1),     # 1x1, 169 px: 1 tail pixel (GEMM kernel's tail launch), cout 255
    (1, 256, 13, 13, 512, 3, 1, 1, 1, 6, True, 1),      # K = 2304, relu6 (YOLO conv5 class)
    (2, 32, 26, 26, 70, 3, 2, 1, 1, 0, False, 1),       # stride 2, no bias, cout % 4 == 2, 169 px
    (3, 8, 9, 11, 13, 3, 1, 1, 1, 1, True, 1),          # small odd map: 99 px -> 96 main + 3 tail, K = 72
    (1, 32, 12, 12, 16, 3, 1, 2, 1, 0, True, 2),        # dilation 2, pad 2
    (1, 512, 7, 7, 64, 3, 1, 1, 1, 0, True, 1),         # K = 4608, 49 px: 48 main
    (2, 64, 19, 19, 128, 1, 1, 0, 1, 0, True, 1),       # mssd pointwise class: 361 px, 1x1
    (1, 96, 10, 10, 40, 1, 1, 0, 1, -1, True, 1),       # 1x1, C = 96 = 6 super-steps (a chunk and a half), cout 40
    (1, 32, 19, 19, 48, 1, 2, 0, 1, 0, True, 1),        # 1x1 stride 2
    (2, 48, 13, 13, 100, 1, 1, 0, 1, 6, True, 1),       # 1x1, C = 48 = three 16-channel super-steps (one partial chunk), 1 tail pixel
    (1, 4, 24, 24, 24, 3, 1, 1, 1, -1, True, 1),        # C = 4: one super-step in all (first-layer kernel also applies; patch pinned)
    (1, 20, 16, 16, 32, 3, 1, 0, 1, 0, True, 1),        # pad 0, C = 20: five super-steps, odd chunk count
    (1, 16, 8, 30, 32, 3, 1, 1, 1, 0, True, 1),         # wide rows
]


@pytest.mark.parametrize("case", CASES, ids=[str(c) for c in CASES])
def test_patch_conv_bytes_equal_oracle_and_gemm_member(case):
    n, cin, h, w, cout, k, s, p, group, act, bias, dil = case
    g, x = u8_conv_graph(57 + cin + cout, n, cin, h, w, cout, k, s, p, group, act, bias, dil)
    want = oracle.run_graph(g, x)
    got, kernels = run_with(g, x, "1")
    assert any("conv_u8_patch" in kn for kn in kernels), kernels
    ref, kernels0 = run_with(g, x, "0")
    assert not any("conv_u8_patch" in kn for kn in kernels0), kernels0
    for wv, a, b in zip(want, got, ref):
        a = a.reshape(wv.shape)
        bad = np.count_nonzero(a != wv)
        assert bad == 0, "%d / %d bytes differ from the oracle (max |d| %d)" % (bad, wv.size, np.abs(a.astype(int) - wv.astype(int)).max())
        assert np.array_equal(a, b.reshape(wv.shape))
        assert len(np.unique(wv)) > 3


POOL_CASES = [
    ((1, 16, 16, 16, 32, 3, 1), dict()),                     # conv -> leaky -> 2x2/2 pool fused in the patch kernel's epilogue
    ((2, 32, 12, 20, 70, 3, 1), dict(slope=0.0)),            # 240 px, cout 70
    ((1, 32, 16, 16, 32, 1, 0), dict(relu=False)),           # 1x1 conv -> pool
    ((1, 16, 16, 16, 32, 3, 1), dict(second_reader=True)),   # unpooled tensor stored too
    ((1, 16, 52, 52, 32, 3, 1), dict()),                     # YOLO conv2 class: 2704 px, window-major tiles span 4 rows
]


@pytest.mark.parametrize("dims,kw", POOL_CASES, ids=[str(c[0]) + str(sorted(c[1].items())) for c in POOL_CASES])
def test_patch_conv_with_fused_relu_and_pool(dims, kw):
    g, x = u8_conv_pool_graph(190 + dims[1] + dims[2], *dims, **kw)
    want = oracle.run_graph(g, x)
    got, kernels = run_with(g, x, "1")
    assert any("conv_u8_patch" in kn and "+maxpool" in kn for kn in kernels), kernels
    for wv, a in zip(want, got):
        assert np.array_equal(a.reshape(wv.shape), wv), "%d bytes differ" % np.count_nonzero(a.reshape(wv.shape) != wv)


@pytest.mark.parametrize("name,batch,res", [("yolov3_tiny", 2, 416), ("mssd", 2, 300), ("yolov3_tiny", 1, 160)])
def test_models_with_the_patch_member_pinned(name, batch, res):
    g = models.build(name, "uint8", batch, res=res)
    x = models.synth_input(g, 5, tm2.DT_UINT8)
    want = oracle.run_graph(g, x)
    got, kernels = run_with(g, x, "1")
    assert sum("conv_u8_patch" in kn for kn in kernels) >= 5, kernels
    for wv, a in zip(want, got):
        assert np.array_equal(a.reshape(wv.shape), wv), "%d bytes differ" % np.count_nonzero(a.reshape(wv.shape) != wv)


# (the 2-D pixel-tile forms of the patch kernel, configurations 5 .. 8, lost on the wide maps they were built for --
#  profiles/r04_experiment_u8_patch_2d_tiles.txt -- and are compiled into tools/exp builds only since round 5)


# n, cin, h, w, cout, stride, pad, act, pool
C3_CASES = [
    (1, 16, 32, 48, 32, 1, 1, 0, False),          # YOLOv3-tiny conv1 class: C = 16 (four super-steps), two cout tiles
    (2, 32, 24, 40, 64, 1, 1, 6, False),          # conv2 class: C = 32 (eight super-steps), two cout groups, relu6, batch 2
    (3, 16, 21, 23, 24, 1, 1, -1, False),         # 483 px: 480 main + 3 tail pixels (patch-tail blocks), cout 24 (a tile and a half)
    (1, 16, 33, 47, 70, 2, 1, 0, False),          # stride 2, 17 x 24 = 408 outputs, cout 70: three cout groups (waves % 3 == 0), ragged rows
    (1, 32, 9, 9, 40, 1, 0, 0, False),            # no padding, 7 x 7 = 49 px: 48 main (three tiles) + 1 tail
    (1, 16, 16, 48, 32, 1, 1, -1, True),          # fused leaky ReLU + 2x2 pool (window-major tiles, byte tables)
    (2, 32, 32, 104, 64, 1, 1, -1, True),         # conv2 under the pool, batch 2
]


@pytest.mark.parametrize("case", C3_CASES, ids=[str(c) for c in C3_CASES])
def test_shallow_3x3_wave_kernel(case, monkeypatch):
    """conv_u8_c3 (weights resident in registers, 3x3 gather straight from the NCHW input) pinned with TAMD_PIN u8_c3=1: the oracle's bytes
    and the GEMM member's"""
    n, cin, h, w, cout, s, p, act, pool = case
    if pool:
        g, x = u8_conv_pool_graph(230 + cin + w, n, cin, h, w, cout, 3, p)
    else:
        g, x = u8_conv_graph(91 + cin + cout + w, n, cin, h, w, cout, 3, s, p, 1, act, True, 1)
    want = oracle.run_graph(g, x)
    got, kernels = run_with(g, x, "0", u8_c3=1)
    assert any(kn.startswith("conv_u8_c3") for kn in kernels), kernels
    ref, kernels0 = run_with(g, x, "0", u8_c3=0)
    assert not any(kn.startswith("conv_u8_c3") for kn in kernels0), kernels0
    for wv, a, b in zip(want, got, ref):
        a = a.reshape(wv.shape)
        assert np.array_equal(a, wv), "%d / %d bytes differ from the oracle" % (np.count_nonzero(a != wv), wv.size)
        assert np.array_equal(a, b.reshape(wv.shape))


PW_CASES = [
    # n, cin, h, w, cout, act, bias      (1x1, stride 1, pad 0; K = cin in {32, 64})
    (2, 32, 20, 20, 64, 0, True),        # K = 32, 400 px: 25 column tiles, no tail
    (1, 32, 13, 13, 13, -1, True),       # 169 px: 168 main (half a tile at the end) + 1 tail pixel; cout 13 (one partial row tile)
    (3, 64, 10, 10, 100, 6, True),       # K = 64, 100 px: 96 main + 4 tail; relu6; cout 100 (two 64-row wave tiles, the second ragged)
    (1, 64, 38, 38, 128, 0, False),      # MobileNet-SSD conv2 class, no bias
    (2, 64, 19, 19, 255, 1, True),       # cout 255 (four wave tiles, the last ragged), 361 px: 360 main + 1 tail, relu1 clamp
    (1, 32, 8, 8, 32, -1, True),         # 64 px: four tiles in all
]


@pytest.mark.parametrize("case", PW_CASES, ids=[str(c) for c in PW_CASES])
def test_shallow_pointwise_kernel_bytes_equal_oracle(case, monkeypatch):
    """conv_u8_pw: a wave keeps the weights of its output rows for the whole K in registers and reads B straight from the NCHW
    input -- TAMD_PIN u8_pw=1 pins it where it applies; bytes == oracle == the other members"""
    n, cin, h, w, cout, act, bias = case
    g, x = u8_conv_graph(77 + cin + cout, n, cin, h, w, cout, 1, 1, 0, 1, act, bias, 1)
    want = oracle.run_graph(g, x)
    got, kernels = run_with(g, x, "0", u8_pw=1)
    assert any(kn.startswith("conv_u8_pw<") for kn in kernels), kernels
    ref, kernels0 = run_with(g, x, "0", u8_pw=0)
    assert not any("conv_u8_pw" in kn for kn in kernels0), kernels0
    for wv, a, b in zip(want, got, ref):
        a = a.reshape(wv.shape)
        bad = np.count_nonzero(a != wv)
        assert bad == 0, "%d / %d bytes differ from the oracle (max |d| %d)" % (bad, wv.size, np.abs(a.astype(int) - wv.astype(int)).max())
        assert np.array_equal(a, b.reshape(wv.shape))
        assert len(np.unique(wv)) > 3


def test_mssd_with_the_shallow_pointwise_kernel_pinned():
    g = models.build("mssd", "uint8", 2)
    x = models.synth_input(g, 6, tm2.DT_UINT8)
    want = oracle.run_graph(g, x)
    got, kernels = run_with(g, x, "1", u8_pw=1)
    assert sum(kn.startswith("conv_u8_pw<") for kn in kernels) >= 2, kernels
    for wv, a in zip(want, got):
        assert np.array_equal(a.reshape(wv.shape), wv), "%d bytes differ" % np.count_nonzero(a.reshape(wv.shape) != wv)
