"""GPU parity, uint8: the HIP backend through the C ABI against the CPU oracle, which
tests/test_uint8_oracle.py pins BIT-EXACTLY to the real reference (including the summation order of its AVX
sgemm).  Bar: byte-identical -- the device performs the reference's fp32 operations in the reference's order."""
import numpy as np
import pytest

import os

from helpers import (PRIORBOX_CASES, axis_concat_graph, priorbox_graph, u8_conv_graph, u8_conv_pool_graph, u8_fc_graph, u8_pool_graph, u8_route_graph,
                     u8_ssd_head_graph, u8_unary_graph)
from oracle import oracle
from tengine_amd import capi, models, tm2

pytestmark = pytest.mark.gpu


def run_hip(g, x):
    gr = capi.Graph(tm2.write_tm2(g))
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
        assert o.dtype == np.uint8
        bad = np.count_nonzero(w != o)
        assert bad == 0, "%s: %d / %d bytes differ (max |d| %d)" % (
            tag, bad, w.size, np.abs(w.astype(int) - o.astype(int)).max())
        assert len(np.unique(w)) > 3, "degenerate case"


U8_CONV = [
    # n, cin, h, w, cout, k, s, p, group, act, bias, dil
    (1, 3, 64, 64, 16, 3, 1, 1, 1, -1, True, 1),       # YOLO conv0 shape class (cout 16 -> 256x16 tiles)
    (1, 16, 40, 40, 32, 3, 1, 1, 1, -1, True, 1),      # 128x32 tiles
    (1, 64, 20, 20, 128, 3, 1, 1, 1, 0, True, 1),
    (2, 128, 13, 13, 255, 1, 1, 0, 1, -1, True, 1),    # 169 px: 1 tail pixel; cout 255: 3 rows outside the 8/4 blocks
    (1, 256, 13, 13, 512, 3, 1, 1, 1, 6, True, 1),     # K = 2304, relu6
    (1, 30, 26, 26, 70, 3, 2, 1, 1, 0, False, 1),      # stride 2, no bias, K % 4 == 2, cout % 4 == 2, 169 px
    (3, 7, 9, 11, 13, 3, 1, 1, 1, 1, True, 1),         # odd everything, K = 63 (K % 4 == 3), act code 1 == relu6
    (1, 32, 12, 12, 16, 3, 1, 2, 1, 0, True, 2),       # dilation 2
    (1, 64, 1, 1, 10, 1, 1, 0, 1, -1, True, 1),        # 1x1 map: every pixel is a "tail" pixel
    (1, 512, 7, 7, 64, 3, 1, 1, 1, 0, True, 1),        # K = 4608, 49 px
    (1, 32, 12, 12, 32, 3, 1, 1, 32, 0, True, 1),      # depthwise -> naive-ref formula
    (2, 24, 13, 13, 24, 3, 2, 1, 24, 6, True, 1),      # depthwise stride 2, relu6, batch 2
    (1, 16, 9, 9, 32, 3, 1, 1, 4, 1, True, 1),         # grouped, relu1 clamp
    (2, 40, 19, 19, 40, 3, 1, 1, 40, 0, True, 1),      # dw, width 19: last quad of a row is ragged (mssd conv7..11/dw)
    (1, 48, 38, 38, 48, 3, 2, 1, 48, 0, True, 1),      # dw stride 2, 38 -> 19 (mssd conv6/dw)
    (3, 16, 10, 10, 16, 3, 2, 1, 16, -1, False, 1),    # dw stride 2, 10 -> 5, no bias, no activation
    (1, 16, 5, 5, 16, 3, 2, 1, 16, 0, True, 1),        # dw, OW = 3 < 4: single-output kernel
    (1, 16, 12, 12, 16, 3, 1, 0, 16, 0, True, 1),      # dw, pad 0: single-output kernel
    (1, 16, 12, 12, 16, 3, 1, 2, 16, 1, True, 2),      # dw, dilation 2, relu1
    (1, 8, 11, 11, 8, 5, 1, 2, 8, 0, True, 1),         # dw 5x5: generic direct kernel
]


@pytest.mark.parametrize("case", U8_CONV, ids=[str(c) for c in U8_CONV])
def test_conv_u8(case):
    n, cin, h, w, cout, k, s, p, group, act, bias, dil = case
    g, x = u8_conv_graph(31 + cin + cout, n, cin, h, w, cout, k, s, p, group, act, bias, dil)
    check(g, x, str(case))


@pytest.mark.parametrize("zps", [(0, 0, 0), (255, 255, 255), (0, 255, 128), (255, 0, 7)])
def test_conv_u8_extreme_zero_points(zps):
    g, x = u8_conv_graph(5, 1, 32, 10, 10, 48, 3, 1, 1, act=-1, in_zp=zps[0], w_zp=zps[1], out_zp=zps[2])
    want = oracle.run_graph(g, x)[0]
    got = run_hip(g, x)[0].reshape(want.shape)
    assert np.array_equal(want, got)


@pytest.mark.parametrize("case", [(3, (64,), 10), (2, (32, 4, 4), 100), (4, (2048,), 1000)])
def test_fc_u8(case):
    n, hd, nout = case
    g, x = u8_fc_graph(8, n, hd, nout)
    check(g, x, "fc")


@pytest.mark.parametrize("case", [(1, 16, 14, 14, 0, 2, 2, 0, 0, 0), (1, 16, 13, 13, 0, 2, 1, -1, 0, 0),
                                  (2, 8, 12, 12, 1, 3, 2, 1, 0, 0), (2, 8, 12, 12, 1, 3, 2, 1, 0, 1),
                                  (2, 64, 7, 7, 1, 7, 1, 0, 1, 0), (1, 24, 15, 15, 0, 3, 2, 0, 0, 1)])
def test_pool_u8(case):
    n, c, h, w, alg, k, s, p, glob, caffe = case
    g, x = u8_pool_graph(9, n, c, h, w, alg, k, s, p, glob, caffe)
    check(g, x, "pool")


@pytest.mark.parametrize("slope", [0.0, 0.1])
def test_relu_u8(slope):
    g, x = u8_unary_graph(13, "ReLU", [2, 8, 19, 9], negative_slope=slope)
    check(g, x, "relu")


def test_upsample_u8():
    g, x = u8_unary_graph(15, "Upsample", [2, 8, 5, 7], [2, 8, 10, 14], scale=2)
    check(g, x, "upsample")


def test_route_u8():
    g, x = u8_route_graph(16, 2, 8, 6, 6)
    check(g, x, "route")


@pytest.mark.parametrize("group,slope", [(1, 0.1), (1, 0.0), (24, 0.1)])
def test_conv_relu_fused_u8(group, slope):
    """conv -> (leaky) ReLU folded into one launch: the ReLU node works on the conv's own uint8 bytes in registers."""
    g, x = u8_conv_graph(77, 2, 24, 13, 13, 24 if group > 1 else 40, 3, 1, 1, group=group, act=-1)
    c = g.nodes[-1].outputs[0]
    r = g.add_tensor("lk", list(g.tensors[c].dims), tm2.DT_UINT8, tm2.TT_VAR, None, [g.tensors[c].scales[0] * 0.6], [31])
    ni = g.add_node("lk", "ReLU", [c], [r], negative_slope=slope)
    g.output_nodes = [ni]
    want = oracle.run_graph(g, x)[0]
    gr = capi.Graph(tm2.write_tm2(g))
    gr.set_input(x)
    got = gr.run()[0].reshape(want.shape)
    assert gr.kernel_num() == 1, "conv and ReLU should be one launch"
    gr.close()
    assert np.array_equal(want, got)


@pytest.mark.parametrize("batch", [1, 3])
def test_concat_by_offset_u8(batch):
    """two convs whose outputs carry the concat's own (scale, zp): the planner lets them write into the concat
    buffer directly (the reference's per-element rescale is then the identity) -- result must not change."""
    g, x = u8_conv_graph(21, batch, 16, 11, 11, 24, 1, act=0)
    g.output_nodes = []
    xin = g.nodes[g.input_nodes[0]].outputs[0]
    a = g.nodes[-1].outputs[0]
    rng = np.random.default_rng(4)
    q = (g.tensors[a].scales[0], g.tensors[a].zero_points[0])
    w2 = g.add_const("w2", rng.integers(0, 256, size=(40, 16, 3, 3)).astype(np.uint8), tm2.DT_UINT8, [0.004], [121])
    b = g.add_tensor("out2", [batch, 40, 11, 11], tm2.DT_UINT8, tm2.TT_VAR, None, [q[0]], [q[1]])
    g.add_node("conv2", "Convolution", [xin, w2], [b], kernel_h=3, kernel_w=3, stride_h=1, stride_w=1, dilation_h=1,
               dilation_w=1, input_channel=16, output_channel=40, group=1, activation=0, pad_h0=1, pad_w0=1, pad_h1=1, pad_w1=1)
    c = g.add_tensor("cat", [batch, 64, 11, 11], tm2.DT_UINT8, tm2.TT_VAR, None, [q[0]], [q[1]])
    ni = g.add_node("cat", "Concat", [a, b], [c], axis=1)
    r = g.add_tensor("lk", [batch, 64, 11, 11], tm2.DT_UINT8, tm2.TT_VAR, None, [q[0] * 0.7], [9])
    ni = g.add_node("lk", "ReLU", [c], [r], negative_slope=0.1)
    g.output_nodes = [ni]
    want = oracle.run_graph(g, x, keep_all=True)
    gr = capi.Graph(tm2.write_tm2(g))
    gr.set_input(x)
    got = gr.run()[0]
    assert not [k for k in gr.profile(1) if "concat" in k["kernel"]], "the concat launches should be gone"
    for t in (a, b, c, r):
        assert np.array_equal(gr.read_tensor(t).reshape(want[t].shape), want[t])
    assert np.array_equal(got.reshape(want[r].shape), want[r])
    gr.close()


def test_fused_away_uint8_tensors_are_refused_by_read_tensor():
    """conv -> leaky ReLU in one launch (and conv -> ReLU -> 2x2 max-pool): the conv's own bytes (and the unpooled map) never reach
    memory -- read_tensor says so instead of returning the zeros of an unwritten buffer (graph_u8 fused_away)"""
    from helpers import u8_conv_graph
    g, x = u8_conv_graph(77, 1, 16, 16, 16, 24, 3, 1, 1, act=-1)
    a = g.nodes[-1].outputs[0]
    q = (g.tensors[a].scales[0], g.tensors[a].zero_points[0])
    r = g.add_tensor("lk", [1, 24, 16, 16], tm2.DT_UINT8, tm2.TT_VAR, None, [q[0] * 0.8], [7])
    g.add_node("lk", "ReLU", [a], [r], negative_slope=0.1)
    pq = g.add_tensor("pool", [1, 24, 8, 8], tm2.DT_UINT8, tm2.TT_VAR, None, [q[0] * 0.8], [7])
    ni = g.add_node("pool", "Pooling", [r], [pq], alg=0, kernel_h=2, kernel_w=2, stride_h=2, stride_w=2, **{"global": 0}, caffe_flavor=0,
                    pad_h0=0, pad_w0=0, pad_h1=0, pad_w1=0)
    g.output_nodes = [ni]
    want = oracle.run_graph(g, x, keep_all=True)
    gr = capi.Graph(tm2.write_tm2(g))
    gr.set_input(x)
    got = gr.run()[0]
    assert np.array_equal(got.reshape(want[pq].shape), want[pq])
    names = [k["kernel"] for k in gr.profile(1)]
    assert len(names) == 1, names                    # conv + leaky ReLU + max-pool: one launch
    for t in (a, r):
        with pytest.raises(capi.TamdError):
            gr.read_tensor(t)
    assert np.array_equal(gr.read_tensor(pq).reshape(want[pq].shape), want[pq])
    gr.close()


def test_yolov3_tiny_uint8_bit_exact():
    """BASELINE configs[3] class: YOLOv3-tiny uint8 (13 convs up to K = 4608, leaky ReLU, max pools incl. the
    stride-1 'same' pool, upsample, concat with per-input rescale), whole graph on the device, layer by layer."""
    import os
    g = models.build("yolov3_tiny", "uint8", 1, res=160)
    x = models.synth_input(g, 3, tm2.DT_UINT8)
    want = oracle.run_graph(g, x, keep_all=True)
    os.environ["TAMD_FUSE_RELU"] = "0"        # every node in its own launch, so that every tensor exists
    try:
        gr = capi.Graph(tm2.write_tm2(g))
    finally:
        del os.environ["TAMD_FUSE_RELU"]
    gr.set_input(x)
    outs = gr.run()
    for n in g.nodes:
        if n.op in ("Const", "InputOp"):
            continue
        t = n.outputs[0]
        dev = gr.read_tensor(t)
        assert np.array_equal(dev.reshape(want[t].shape), want[t]), "layer %s differs" % n.name
    for ni, o in zip(g.output_nodes, outs):
        w = want[g.nodes[ni].outputs[0]]
        assert np.array_equal(o.reshape(w.shape), w)
    gr.close()


def test_yolov3_tiny_uint8_416_matches_golden_of_real_reference():
    import os
    golden = os.path.join(os.path.dirname(__file__), "golden", "yolov3_tiny_uint8_416_seed3.npz")
    if not os.path.exists(golden):
        pytest.skip("golden file not generated")
    g = models.build("yolov3_tiny", "uint8", 1)
    x = models.synth_input(g, 3, tm2.DT_UINT8)
    outs = run_hip(g, x)
    ref = np.load(golden)
    for i, o in enumerate(outs):
        assert np.array_equal(o.ravel(), ref["out%d" % i].ravel())


@pytest.mark.parametrize("name,dev_only,batch", [("mobilenet_v1", False, 1), ("resnet50", True, 1), ("mobilenet_v1", False, 3)])
def test_uint8_classifiers_bit_exact(name, dev_only, batch):
    """MobileNet-v1 uint8 (depthwise -> conv_ref order, 1x1-map classifier) and ResNet-50 uint8 (eltwise, pools, fc)."""
    import os
    g = models.build(name, "uint8", batch, device_only=dev_only)
    x = models.synth_input(g, 5, tm2.DT_UINT8)
    got = run_hip(g, x)[0]
    if batch == 1:
        golden = np.load(os.path.join(os.path.dirname(__file__), "golden", "%s_uint8_seed5.npy" % name))
        assert np.array_equal(got.ravel(), golden.ravel())      # bytes of the REAL reference
    else:
        want = oracle.run_graph(g, x)[0]
        assert np.array_equal(got.reshape(want.shape), want)


@pytest.mark.parametrize("fuse", [True, False])
@pytest.mark.parametrize("same_q", [False, True])
def test_ssd_head_plumbing_u8(fuse, same_q):
    """conv -> Permute(0,2,3,1) -> Flatten -> Concat(axis 1): by default the concat reads the conv result in permuted
    order itself (one launch per head); TAMD_FUSE_PERMUTE=0 keeps the permute launches. Same bytes either way."""
    import os
    g, x = u8_ssd_head_graph(31, 3, 24, 10, 6, same_q=same_q)
    want = oracle.run_graph(g, x)[0]
    if not fuse:
        os.environ["TAMD_FUSE_PERMUTE"] = "0"
    try:
        gr = capi.Graph(tm2.write_tm2(g))
    finally:
        os.environ.pop("TAMD_FUSE_PERMUTE", None)
    gr.set_input(x)
    got = gr.run()[0]
    kernels = [k["kernel"] for k in gr.profile(1)]
    gr.close()
    assert np.array_equal(got.reshape(want.shape), want)
    assert len(np.unique(want)) > 3
    assert ("permute_u8" in kernels) == (not fuse), kernels
    assert any(k.startswith("permute_concat_u8<x") for k in kernels) == fuse, kernels      # one launch for both heads


def test_concat_inputs_in_one_launch_or_one_each(monkeypatch):
    """all copied inputs of a concat node go through ONE launch (six per concat in MobileNet-SSD); TAMD_FUSE_CONCAT=0: one each"""
    g, x = u8_ssd_head_graph(37, 2, 24, 10, 6)
    want = oracle.run_graph(g, x)[0]
    for mode, launches in (("1", 1), ("0", 2)):
        monkeypatch.setenv("TAMD_FUSE_CONCAT", mode)
        gr = capi.Graph(tm2.write_tm2(g))
        gr.set_input(x)
        got = gr.run()[0]
        kernels = [k["kernel"] for k in gr.profile(1)]
        gr.close()
        assert np.array_equal(got.reshape(want.shape), want)
        assert sum(k.startswith("permute_concat_u8") for k in kernels) == launches, kernels


def test_permute_standalone_u8():
    check(*u8_ssd_head_graph(32, 2, 8, 5, 7, standalone_permute=True), tag="permute")


@pytest.mark.parametrize("batch", [1, 2])
def test_mssd_uint8_300_bit_exact(batch):
    """BASELINE configs[4] stand-in (MobileNet-SSD 300x300 uint8, mssd_benchmark.tmfile topology): 47 convs incl. 13
    depthwise (conv_ref order) and the 12 Permute -> Flatten -> Concat heads, whole graph on the device."""
    import os
    g = models.build("mssd", "uint8", batch)
    x = models.synth_input(g, 5, tm2.DT_UINT8)
    outs = run_hip(g, x)
    if batch == 1:
        ref = np.load(os.path.join(os.path.dirname(__file__), "golden", "mssd_uint8_300_seed5.npz"))
        for i, o in enumerate(outs):
            assert np.array_equal(o.ravel(), ref["out%d" % i].ravel())      # bytes of the REAL reference
    else:
        want = oracle.run_graph(g, x)
        for w, o in zip(want, outs):
            assert np.array_equal(o.reshape(w.shape), w)


RGB_CASES = [
    # n, cin, h, w, cout, s, p, act
    (2, 3, 80, 80, 16, 1, 1, -1),      # YOLOv3-tiny conv0 class
    (2, 3, 101, 99, 32, 2, 1, 0),      # MobileNet / SSD conv0 class: stride 2, 50*50 = 2500 px -> 4 tail pixels per image
    (4, 3, 51, 53, 18, 1, 1, 6),       # 2703 px: 7 tail pixels; cout 18: rows 16, 17 sit outside the 8-/4-row blocks
    (3, 4, 60, 60, 13, 1, 0, 0),       # 4 channels (K = 36), no padding, 58*58 = 3364 px: 4 tail pixels
    (4, 1, 150, 150, 24, 2, 1, -1),    # one channel (K = 9: K % 4 == 1 remainder chain on the tail pixels)
]


@pytest.mark.parametrize("case", RGB_CASES, ids=[str(c) for c in RGB_CASES])
@pytest.mark.parametrize("leaky", [False, True])
def test_first_layer_u8_valu_kernel(case, leaky):
    """the per-pixel first-layer kernel conv_u8_rgb3x3 pinned with TAMD_PIN u8_rgb3x3=1 (C = 1 | 3 | 4): main and tail pixels, blocked and
    unblocked rows, fused leaky ReLU -- must equal the oracle (and therefore the GEMM family) byte for byte.  (Its matrix-core form lost
    the race, profiles/r04_experiment_u8_first_layer_mfma.txt, and is built for tools/exp only since round 5.)"""
    import os
    n, cin, h, w, cout, s, p, act = case
    g, x = u8_conv_graph(600 + h + cout, n, cin, h, w, cout, 3, s, p, 1, act, True, 1)
    if leaky:
        c = g.nodes[-1].outputs[0]
        t = g.tensors[c]
        r = g.add_tensor("lk", list(t.dims), tm2.DT_UINT8, tm2.TT_VAR, None, [float(np.float32(t.scales[0] * 0.8))], [31])
        ni = g.add_node("lk", "ReLU", [c], [r], negative_slope=0.1)
        g.output_nodes = [ni]
    want = oracle.run_graph(g, x)[0]
    os.environ["TAMD_PIN"] = "u8_rgb3x3=1"
    try:
        gr = capi.Graph(tm2.write_tm2(g))
    finally:
        del os.environ["TAMD_PIN"]
    gr.set_input(x)
    got = gr.run()[0].reshape(want.shape)
    names = [k["kernel"] for k in gr.profile(1)]
    gr.close()
    assert names[0] == "conv_u8_rgb3x3" or names[0].startswith("conv_u8_rgb3x3+"), names
    assert np.array_equal(got, want), "%d bytes differ" % np.count_nonzero(got != want)
    assert len(np.unique(want)) > 3


@pytest.mark.parametrize("dims,axis", [((3, 7, 5), 1), ((2, 50, 21), 2), ((2, 21, 6, 5), 1), ((4, 33), 1), ((1, 5, 4, 3), 3)])
def test_softmax_u8(dims, axis):
    """softmax_kernel_ref_uint8.c over any axis: dequantise, exp in fp64 rounded to fp32, sequential fp32 sum, requantise"""
    g, x = u8_unary_graph(70 + axis + len(dims), "Softmax", dims, axis=axis)
    t = g.tensors[g.nodes[-1].outputs[0]]
    t.scales, t.zps = [1.0 / 255.0], [0]                  # the quantiser's choice for a probability tensor
    check(g, x, "softmax %s axis %d" % (dims, axis))


@pytest.mark.parametrize("batch", [1, 3])
def test_mssd_uint8_quantised_tail_on_device(batch):
    """mbox_conf -> Reshape(0,-1,21) -> Softmax(axis 2) -> Flatten stays on the device (Reshape / Flatten are
    views of the dense NCHW tensor, softmax_u8 is one launch); outputs = mbox_loc and the class probabilities"""
    g = models.build("mssd", "uint8", batch, tail=True)
    x = models.synth_input(g, 5, tm2.DT_UINT8)
    want = oracle.run_graph(g, x)
    gr = capi.Graph(tm2.write_tm2(g))
    gr.set_input(x)
    outs = gr.run()
    kernels = [k["kernel"] for k in gr.profile(1)]
    gr.close()
    assert "softmax_u8" in kernels, kernels
    for w, o in zip(want, outs):
        assert np.array_equal(o.reshape(w.shape), w), np.count_nonzero(o.reshape(w.shape) != w)


GOLDEN_PRIORBOX = os.path.join(os.path.dirname(__file__), "golden", "priorbox_cases.npz")


@pytest.mark.parametrize("case", sorted(PRIORBOX_CASES))
def test_priorbox_uint8_is_a_prerun_constant(case):
    """PriorBox depends on shapes only -- evaluated once at prerun (graph_infer.hip priorbox_eval), its Concat runs
    once at prerun too; a run launches nothing for them.  Bytes == oracle == the real reference's (golden fixture)."""
    g, x = priorbox_graph(dtype=tm2.DT_UINT8, **PRIORBOX_CASES[case])
    want = oracle.run_graph(g, x)[0]
    gr = capi.Graph(tm2.write_tm2(g))
    gr.set_input(x)
    got = gr.run()[0].reshape(want.shape)
    again = gr.run()[0].reshape(want.shape)          # the constant survives runs
    kernels = [k["kernel"] for k in gr.profile(1)]
    gr.close()
    assert np.array_equal(got, want) and np.array_equal(again, want), np.count_nonzero(got != want)
    assert np.array_equal(got, np.load(GOLDEN_PRIORBOX)["%s_uint8" % case])
    assert not any("concat" in k for k in kernels), kernels


@pytest.mark.parametrize("dims,axis", [([1, 5, 6, 7], 2), ([2, 3, 4, 9], 3), ([2, 8, 5], -1), ([3, 6, 4, 4], 1), ([1, 2, 700, 1], 2)])
def test_concat_any_axis_uint8(dims, axis):
    g, x = axis_concat_graph(21 + axis, tm2.DT_UINT8, dims, axis)
    check(g, x, "concat %s axis %d" % (dims, axis))


def test_mssd_uint8_ends_at_detection_output_inputs():
    """the whole uint8 MobileNet-SSD as ONE device graph up to detection_output's three inputs (mbox_loc, softmaxed mbox_conf,
    mbox_priorbox); the priors add no launch to a run"""
    g = models.build("mssd", "uint8", 1, tail=True, priorbox=True)
    x = models.synth_input(g, 5, tm2.DT_UINT8)
    want = oracle.run_graph(g, x)
    gr = capi.Graph(tm2.write_tm2(g))
    gr.set_input(x)
    outs = gr.run()
    n_launch = gr.kernel_num()
    gr.close()
    g0 = models.build("mssd", "uint8", 1, tail=True)
    gr0 = capi.Graph(tm2.write_tm2(g0))
    n0 = gr0.kernel_num()
    gr0.close()
    assert len(outs) == 3 and n_launch == n0, (n_launch, n0)
    for w, o in zip(want, outs):
        assert np.array_equal(o.reshape(w.shape), w), np.count_nonzero(o.reshape(w.shape) != w)
    assert np.array_equal(outs[2].reshape(want[2].shape), np.load(GOLDEN_PRIORBOX)["mssd_mbox_priorbox_uint8"])


def test_priorbox_batch_2_is_refused():
    """priorbox_ref.c fills image 0 only; the backend refuses instead of inventing the rest"""
    g = models.build("mssd", "uint8", 2, tail=True, priorbox=True)
    with pytest.raises(RuntimeError, match="batch 1"):
        capi.Graph(tm2.write_tm2(g))


CONV_POOL = [
    # n, cin, h, w, cout, k, p, kwargs                      -> fused?
    ((1, 3, 32, 32, 16, 3, 1), dict(), True),                                   # first layer (per-pixel kernel), leaky, YOLO conv0 class
    ((2, 3, 24, 40, 16, 3, 1), dict(same_q=True), True),                        # batch 2, non-square, the quantiser's shared parameters
    ((1, 16, 16, 16, 32, 3, 1), dict(), True),                                  # MFMA GEMM kernel, 256 px
    ((2, 32, 12, 20, 70, 3, 1), dict(slope=0.0), True),                         # plain ReLU, cout % 16 != 0, 240 px (not a tile multiple)
    ((1, 16, 16, 16, 32, 1, 0), dict(relu=False), True),                        # conv -> pool directly (1x1 conv)
    ((1, 16, 16, 16, 32, 3, 1), dict(second_reader=True), True),                # the unpooled tensor is read again: stored too
    ((1, 16, 26, 26, 32, 3, 1), dict(), False),                                 # 676 px: 4 reference tail pixels -> separate pool_u8
    ((1, 16, 13, 13, 32, 3, 1), dict(pool_s=1), False),                         # YOLO maxpool5 (stride 1): separate
    ((1, 16, 16, 18, 32, 3, 1), dict(pool_k=3), False),                         # 3x3 windows: separate
]


@pytest.mark.parametrize("dims,kw,fused", CONV_POOL, ids=[str(c[0]) + str(sorted(c[1].items())) for c in CONV_POOL])
def test_conv_relu_maxpool_fused_in_the_conv_epilogue(dims, kw, fused, monkeypatch):
    """SURVEY 8 f1 (remainder): conv (-> leaky ReLU) -> 2x2 stride-2 max-pool as ONE launch; bytes == oracle (pinned to the
    reference's pooling_kernel_ref_uint8.c), == the unfused launches; geometries the fused kernel does not cover keep pool_u8"""
    g, x = u8_conv_pool_graph(90 + dims[1] + dims[2], *dims, **kw)
    want = oracle.run_graph(g, x)
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("TAMD_FUSE_POOL", mode)
        gr = capi.Graph(tm2.write_tm2(g))
        gr.set_input(x)
        outs[mode] = [o.copy() for o in gr.run()]
        kernels = [k["kernel"] for k in gr.profile(1)]
        gr.close()
        if mode == "1":
            assert any("+maxpool" in k for k in kernels) == fused, kernels
            assert ("pool_u8" in kernels) == (not fused), kernels
        else:
            assert "pool_u8" in kernels and not any("+maxpool" in k for k in kernels), kernels
    for w, a, b in zip(want, outs["1"], outs["0"]):
        assert np.array_equal(a.reshape(w.shape), w), "%d bytes differ" % np.count_nonzero(a.reshape(w.shape) != w)
        assert np.array_equal(a, b)
        assert len(np.unique(w)) > 3


def test_yolov3_tiny_fuses_its_first_four_pools():
    g = models.build("yolov3_tiny", "uint8", 1)
    gr = capi.Graph(tm2.write_tm2(g))
    kernels = [k["kernel"] for k in gr.profile(1)]
    gr.close()
    assert sum("+maxpool" in k for k in kernels) == 4 and kernels.count("pool_u8") == 2, kernels
