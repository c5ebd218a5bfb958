"""GPU parity tests proper: the HIP backend, driven through the C ABI (include/tengine_amd.h) with the
SAME tmfile bytes the reference loads, against the CPU oracle (oracle/tg_oracle.c, itself pinned
bit-exactly to the real reference in tests/test_oracle_vs_reference.py) and, where the prebuilt
reference library travelled to the box (oracle/_ref), against the real reference as well.
Bar: bit-exact (int8)."""
import os

import numpy as np
import pytest

from helpers import conv_graph, eltwise_relu_graph, fc_graph, pool_graph
from oracle import oracle
from tengine_amd import capi, models, tm2

pytestmark = pytest.mark.gpu


def run_hip(g, x, batch=None):
    gr = capi.Graph(tm2.write_tm2(g), batch=batch)
    gr.set_input(x)
    out = gr.run()
    gr.close()
    return out


def check(g, x, tag=""):
    want = oracle.run_graph(g, x)
    got = run_hip(g, x)
    assert len(want) == len(got)
    for w, o in zip(want, got):
        o = o.reshape(w.shape)
        bad = np.count_nonzero(w != o)
        assert bad == 0, "%s: %d / %d bytes differ (max |d| %d)" % (
            tag, bad, w.size, np.abs(w.astype(int) - o.astype(int)).max())
        assert np.count_nonzero(w) > 0


# every MobileNet-v1 pointwise shape (SURVEY §8d layer list), batch 1
MBV1_PW = [(32, 64, 112), (64, 128, 56), (128, 128, 56), (128, 256, 28), (256, 256, 28), (256, 512, 14),
           (512, 512, 14), (512, 1024, 7), (1024, 1024, 7), (1024, 1000, 1)]


@pytest.mark.parametrize("cin,cout,hw", MBV1_PW)
def test_pointwise_mobilenet_shapes(cin, cout, hw):
    g, x = conv_graph(100 + cin + hw, 1, cin, hw, hw, cout, 1, act=0 if cout != 1000 else -1)
    check(g, x, "pw %d->%d@%d" % (cin, cout, hw))


MBV1_DW = [(32, 112, 1), (64, 112, 2), (128, 56, 1), (128, 56, 2), (256, 28, 1), (256, 28, 2), (512, 14, 1),
           (512, 14, 2), (1024, 7, 1)]


@pytest.mark.parametrize("c,hw,s", MBV1_DW)
def test_depthwise_mobilenet_shapes(c, hw, s):
    g, x = conv_graph(200 + c + hw + s, 1, c, hw, hw, c, 3, s, 1, group=c, act=0)
    check(g, x, "dw %d@%d s%d" % (c, hw, s))


CONV_CASES = [
    # n, cin, h, w, cout, k, s, p, group, act, bias, dil
    (1, 3, 224, 224, 32, 3, 2, 1, 1, 0, True, 1),     # MobileNet conv1 (direct from NCHW)
    (2, 3, 64, 64, 64, 7, 2, 3, 1, 0, True, 1),       # ResNet stem, batch 2
    (1, 64, 56, 56, 64, 3, 1, 1, 1, 0, True, 1),      # ResNet 3x3
    (2, 128, 28, 28, 128, 3, 1, 1, 1, 0, False, 1),   # batch 2, no bias
    (1, 256, 14, 14, 512, 1, 2, 0, 1, -1, True, 1),   # ResNet 1x1 stride 2 projection
    (1, 16, 13, 13, 24, 3, 1, 1, 1, 6, True, 1),      # odd sizes, relu6, cout % 16 != 0
    (3, 48, 9, 7, 40, 3, 2, 1, 1, 1, True, 1),        # act code 1 on the hcl path == relu6
    (2, 32, 12, 12, 16, 3, 1, 2, 1, 0, True, 2),      # dilation 2
    (1, 20, 10, 10, 36, 1, 1, 0, 1, 0, True, 1),      # cin % 16 != 0 (padded K)
    (1, 512, 7, 7, 512, 3, 1, 1, 1, 0, True, 1),      # K = 4608
    (2, 32, 10, 10, 32, 3, 1, 1, 32, 0, True, 1),     # dw batch 2 -> reference uses the naive-ref epilogue
    (2, 32, 10, 10, 32, 3, 1, 1, 32, 1, True, 1),     # .. with relu1 clamp
    (1, 16, 9, 9, 32, 3, 1, 1, 4, 0, True, 1),        # grouped conv -> direct fallback
    (1, 24, 9, 9, 24, 5, 1, 2, 24, 6, True, 1),       # dw 5x5 -> direct fallback
    (4, 64, 1, 1, 10, 1, 1, 0, 1, -1, True, 1),       # 1x1 map, tiny cout
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[str(c) for c in CONV_CASES])
def test_conv_cases(case):
    n, cin, h, w, cout, k, s, p, group, act, bias, dil = case
    g, x = conv_graph(7 + cin + cout, n, cin, h, w, cout, k, s, p, group, act, bias, dil)
    check(g, x, str(case))


BIG_CASES = [   # large enough for the LDS-DMA 3-stage kernel (conv_igemm2.hip)
    (8, 64, 56, 56, 64, 3, 1, 1, 1, 0, True, 1),      # ResNet res2 3x3, 128x64 tile
    (4, 128, 56, 56, 256, 1, 1, 0, 1, -1, True, 1),   # wide 1x1, 128x128 tile, no activation
    (32, 128, 28, 28, 128, 3, 1, 1, 1, 0, True, 1),   # K = 1152
    (24, 64, 40, 40, 128, 3, 2, 1, 1, 6, False, 1),   # stride 2, relu6, no bias, ragged M tail
    (9, 80, 33, 31, 96, 3, 1, 2, 1, 0, True, 2),      # dilation 2, cin/cout not multiples of 32/128
]


@pytest.mark.parametrize("case", BIG_CASES, ids=[str(c) for c in BIG_CASES])
def test_conv_big_cases(case):
    n, cin, h, w, cout, k, s, p, group, act, bias, dil = case
    g, x = conv_graph(17 + cin + cout, n, cin, h, w, cout, k, s, p, group, act, bias, dil)
    check(g, x, str(case))


def test_extreme_values_saturate_like_reference():
    """all-127 inputs x all-(-127) weights: accumulators at their extremes, outputs clamp to -127."""
    g, x = conv_graph(1, 1, 64, 8, 8, 64, 3, 1, 1, act=-1)
    g.tensors[[i for i, t in enumerate(g.tensors) if t.name == "w"][0]].data[:] = -127
    x[:] = 127
    check(g, x, "extreme")


@pytest.mark.parametrize("case", [(1, (64,), 10), (3, (32, 2, 2), 17), (32, (2048,), 1000)])
def test_fc(case):
    n, hd, nout = case
    g, x = fc_graph(5, n, hd, nout)
    check(g, x, "fc")


@pytest.mark.parametrize("case", [(1, 1024, 7, 7, 1, 7, 1, 0, 1, 1), (2, 64, 112, 112, 0, 3, 2, 0, 0, 1),
                                  (1, 16, 15, 15, 0, 3, 2, 1, 0, 0), (1, 24, 12, 12, 1, 3, 2, 1, 0, 0),
                                  (1, 24, 12, 12, 1, 3, 2, 1, 0, 1), (1, 16, 13, 13, 0, 2, 2, 0, 0, 0)])
def test_pool(case):
    n, c, h, w, alg, k, s, p, glob, caffe = case
    g, x = pool_graph(3, n, c, h, w, alg, k, s, p, glob, caffe)
    check(g, x, "pool")


@pytest.mark.parametrize("with_relu", [False, True])
def test_eltwise_relu(with_relu):
    g, x = eltwise_relu_graph(9, 2, 64, 14, 14, with_relu)
    check(g, x, "eltwise")


@pytest.mark.parametrize("etype", [tm2.ELT_SUM, tm2.ELT_SUB, tm2.ELT_MAX, tm2.ELT_PROD])
def test_eltwise_relu_with_its_own_scale(etype):
    """ReLU output scale != eltwise output scale: the general two-rounding tail (not the max(y,0) shortcut), and the
    non-commutative SUB with the conv on either side; fused into the later conv's epilogue."""
    g, x = eltwise_relu_graph(19, 3, 32, 9, 9, True, etype)
    r = [t for t in g.tensors if t.name == "relu"][0]
    r.scales = [float(np.float32(r.scales[0] * 0.83))]
    check(g, x, "eltwise etype %d" % etype)


def test_mobilenet_v1_int8_batch1_bit_exact():
    g = models.build("mobilenet_v1", "int8", 1)
    x = models.synth_input(g, 7)
    want = oracle.run_graph(g, x, keep_all=True)
    out_t = g.nodes[g.output_nodes[0]].outputs[0]
    golden = os.path.join(os.path.dirname(__file__), "golden", "mobilenet_v1_int8_seed7.npy")
    # layer by layer with every node in its own launch (pinpoints a regression), then the default plan
    # (pointwise + depthwise pairs fused, pwdw.hip): same bytes
    for fuse in ("0", None):
        if fuse is not None:
            os.environ["TAMD_FUSE_PWDW"] = fuse
        try:
            # layer by layer: every tensor keeps its own buffer (the default shares memory between disjoint lifetimes)
            gr = capi.Graph(tm2.write_tm2(g), keep_tensors=(fuse == "0"))
        finally:
            os.environ.pop("TAMD_FUSE_PWDW", None)
        gr.set_input(x)
        got = gr.run()[0]
        if fuse == "0":
            for n in g.nodes:
                if n.op in ("Const", "InputOp"):
                    continue
                t = n.outputs[0]
                dev = gr.read_tensor(t)
                assert np.array_equal(dev.reshape(want[t].shape), want[t]), "layer %s differs" % n.name
        assert np.array_equal(got.reshape(want[out_t].shape), want[out_t])
        if os.path.exists(golden):     # produced by the REAL reference (tests/golden/make_golden.py)
            assert np.array_equal(got.ravel(), np.load(golden).ravel())
        # replaying the captured hipGraph is idempotent
        again = gr.run()[0]
        assert np.array_equal(again, got)
        gr.close()


def test_mobilenet_v1_int8_batch4_matches_per_image_oracle():
    """batch > 1: the reference's depthwise selection switches to the naive-ref epilogue;
    the backend must follow it."""
    g = models.build("mobilenet_v1", "int8", 4)
    x = models.synth_input(g, 11)
    want = oracle.run_graph(g, x)[0]
    got = run_hip(g, x)[0]
    assert np.array_equal(got.reshape(want.shape), want)


def test_real_reference_side_by_side_if_present():
    """When oracle/_ref travelled to the box, compare against the real reference CPU backend too."""
    from oracle import ref_capi
    if not ref_capi.available():
        pytest.skip("prebuilt reference library not present")
    g, x = conv_graph(77, 1, 64, 28, 28, 128, 3, 1, 1)
    b = tm2.write_tm2(g)
    want = ref_capi.run_model(b, x, ref_capi.MODE_INT8, 2)[0]
    got = run_hip(g, x)[0]
    assert np.array_equal(want, got.reshape(want.shape))


def test_resnet50_int8_batch2_bit_exact():
    """BASELINE configs[2] topology (ResNet-50 int8: 3x3 implicit GEMM up to K=4608, 7x7 stem, 1x1 stride 2,
    eltwise+relu fusion, max pool, global avg pool, FC) at a batch the oracle finishes in seconds."""
    g = models.build("resnet50", "int8", 2, device_only=True)
    x = models.synth_input(g, 5)
    want = oracle.run_graph(g, x, keep_all=True)
    gr = capi.Graph(tm2.write_tm2(g))
    gr.set_input(x)
    got = gr.run()[0]
    out_t = g.nodes[g.output_nodes[0]].outputs[0]
    if not np.array_equal(got.reshape(want[out_t].shape), want[out_t]):
        gr.close()
        gr = capi.Graph(tm2.write_tm2(g), keep_tensors=True)      # again with one buffer per tensor, to read the layers back
        gr.set_input(x)
        gr.run()
        for n in g.nodes:                      # pinpoint the first differing layer
            if n.op in ("Const", "InputOp"):
                continue
            t = n.outputs[0]
            if t in want and not (n.op == "Eltwise"):      # eltwise outputs are fused into the following relu
                try:
                    dev = gr.read_tensor(t)
                except capi.TamdError:                     # conv results folded into an eltwise epilogue never reach memory
                    continue
                assert np.array_equal(dev.reshape(want[t].shape), want[t]), "layer %s differs" % n.name
        raise AssertionError("output differs")
    gr.close()


FIRST_LAYER_CASES = [
    # n, cin, h, w, cout, k, s, p, act  -- first layer straight from the NCHW graph input (conv_first.hip)
    (2, 3, 37, 41, 32, 7, 2, 3, 0),      # 7x7 s2 p3 (ResNet stem), odd sizes: left/right/top/bottom borders, 2 images
    (3, 3, 20, 10, 16, 3, 1, 1, 0),      # 3x3 s1 p1, width 10: the 4-byte rows of the last pixels run past the row end
    (1, 1, 9, 9, 8, 3, 2, 1, -1),        # one input channel
    (2, 4, 12, 14, 40, 5, 1, 2, 6),      # 4 channels, 5x5 (rows of 8), relu6, cout 40 (two cout tiles, ragged)
    (1, 3, 16, 16, 64, 7, 1, 3, 0),      # 7x7 s1: three border columns on either side
    (2, 3, 8, 8, 24, 3, 2, 0, 0),        # no padding
    (1, 4, 7, 9, 16, 7, 2, 3, 0),        # 4 x 7 rows of 8 = 224 k > 192: generic gather kernel
    (1, 3, 12, 12, 16, 3, 1, 2, 0, 2),   # dilation 2: generic gather kernel
]


@pytest.mark.parametrize("rows", [True, False])
@pytest.mark.parametrize("case", FIRST_LAYER_CASES, ids=[str(c) for c in FIRST_LAYER_CASES])
def test_first_layer_from_nchw(case, rows):
    """both first-layer kernels (row-granular unaligned loads / generic byte gather, TAMD_PIN first_rows=0) on shapes that
    hit every border case; the graph input is read in NCHW, images are adjacent in memory."""
    n, cin, h, w, cout, k, s, p, act = case[:9]
    dil = case[9] if len(case) > 9 else 1
    g, x = conv_graph(900 + h + w + cout, n, cin, h, w, cout, k, s, p, 1, act, True, dil)
    x[:] = np.random.default_rng(7).integers(-127, 128, size=x.shape)       # dense non-zero borders
    want = oracle.run_graph(g, x)[0]
    if not rows:
        os.environ["TAMD_PIN"] = "first_rows=0"
    try:
        gr = capi.Graph(tm2.write_tm2(g))
    finally:
        os.environ.pop("TAMD_PIN", None)
    gr.set_input(x)
    got = gr.run()[0].reshape(want.shape)
    names = [q["kernel"] for q in gr.profile(1)]
    gr.close()
    assert names[0].startswith("conv_first"), names
    assert np.array_equal(got, want), "%d bytes differ" % np.count_nonzero(got != want)
    assert np.count_nonzero(want) > 0
