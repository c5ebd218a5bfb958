"""uint8: pins the CPU oracle's uint8 restatement BIT-EXACTLY to the real reference
(oracle/_ref) -- op by op, including the position-dependent summation order of the reference's AVX sgemm --
and to the committed golden vectors the real reference produced for YOLOv3-tiny uint8 416x416."""
import copy
import os

import numpy as np
import pytest

from helpers import (u8_conv_graph, u8_fc_graph, u8_pool_graph, u8_route_graph, u8_ssd_head_graph, u8_unary_graph)
from oracle import oracle, ref_capi
from tengine_amd import models, tm2

needs_ref = pytest.mark.skipif(not ref_capi.available(), reason="reference library not built (oracle/build_ref.py)")


def both(g, x):
    want = ref_capi.run_model(tm2.write_tm2(g), x, ref_capi.MODE_UINT8, 2)
    got = oracle.run_graph(g, x)
    assert len(want) == len(got)
    for w, o in zip(want, got):
        assert np.array_equal(w, o.reshape(w.shape)), "%d bytes differ" % np.count_nonzero(w != o.reshape(w.shape))
        assert len(np.unique(w)) > 3


CASES = [
    (1, 32, 14, 14, 48, 3, 1, 1, 1, 0, True, 1),
    (2, 64, 9, 9, 40, 1, 1, 0, 1, -1, True, 1),
    (1, 16, 12, 12, 24, 3, 2, 1, 1, 6, True, 1),
    (2, 128, 13, 13, 255, 1, 1, 0, 1, -1, True, 1),    # tail pixel + rows outside the 8/4 row blocks
    (1, 30, 26, 26, 70, 3, 2, 1, 1, 0, False, 1),      # K % 4 == 2
    (3, 7, 9, 11, 13, 3, 1, 1, 1, 1, True, 1),         # K % 4 == 3, everything ragged
    (1, 32, 12, 12, 16, 3, 1, 2, 1, 0, True, 2),
    (1, 64, 1, 1, 10, 1, 1, 0, 1, -1, True, 1),
    (1, 512, 7, 7, 64, 3, 1, 1, 1, 0, True, 1),        # K = 4608
    (1, 32, 12, 12, 32, 3, 1, 1, 32, 0, True, 1),      # depthwise -> conv_ref
    (2, 24, 13, 13, 24, 3, 2, 1, 24, 6, True, 1),
    (1, 16, 9, 9, 32, 3, 1, 1, 4, 1, True, 1),
]


@needs_ref
@pytest.mark.parametrize("case", CASES, ids=[str(c) for c in CASES])
def test_conv_uint8_oracle_is_the_reference(case):
    n, cin, h, w, cout, k, s, p, group, act, bias, dil = case
    both(*u8_conv_graph(31 + cin + cout, n, cin, h, w, cout, k, s, p, group, act, bias, dil))


@needs_ref
@pytest.mark.parametrize("zps", [(0, 0, 0), (255, 255, 255), (0, 255, 128), (255, 0, 7)])
def test_conv_uint8_extreme_zero_points(zps):
    g, x = u8_conv_graph(5, 1, 32, 10, 10, 48, 3, 1, 1, act=-1, in_zp=zps[0], w_zp=zps[1], out_zp=zps[2])
    want = ref_capi.run_model(tm2.write_tm2(g), x, ref_capi.MODE_UINT8, 2)[0]
    assert np.array_equal(want, oracle.run_graph(g, x)[0].reshape(want.shape))


@needs_ref
def test_glue_ops_uint8_oracle_is_the_reference():
    both(*u8_fc_graph(7, 3, (64,), 10))
    both(*u8_fc_graph(8, 2, (32, 4, 4), 100))
    both(*u8_pool_graph(9, 1, 16, 14, 14, 0, 2, 2))
    both(*u8_pool_graph(10, 1, 16, 13, 13, 0, 2, 1, -1))
    both(*u8_pool_graph(11, 2, 8, 12, 12, 1, 3, 2, 1))
    both(*u8_pool_graph(11, 2, 8, 12, 12, 1, 3, 2, 1, caffe=1))
    both(*u8_pool_graph(12, 2, 64, 7, 7, 1, 7, 1, 0, glob=1))
    both(*u8_unary_graph(13, "ReLU", [2, 8, 9, 9], negative_slope=0.0))
    both(*u8_unary_graph(14, "ReLU", [2, 8, 9, 9], negative_slope=0.1))
    both(*u8_unary_graph(15, "Upsample", [1, 8, 5, 5], [1, 8, 10, 10], scale=2))
    both(*u8_route_graph(16, 1, 8, 6, 6))
    both(*u8_ssd_head_graph(17, 2, 16, 6, 6))                       # permute -> flatten -> concat with rescale
    both(*u8_ssd_head_graph(18, 1, 8, 5, 7, same_q=True))
    both(*u8_ssd_head_graph(19, 2, 8, 4, 4, standalone_permute=True))


@needs_ref
def test_yolov3_tiny_uint8_layer_by_layer_against_reference():
    """every node of YOLOv3-tiny uint8 (res 160) on exactly the inputs the reference saw (teacher forcing)."""
    g = models.quantize_uint8(models.yolov3_tiny_fp32(1, 160))
    x = models.synth_input(g, 3, tm2.DT_UINT8)
    teacher = {}
    for ni, n in enumerate(g.nodes):
        if n.op in ("Const", "InputOp"):
            continue
        g2 = copy.copy(g)
        g2.nodes, g2.output_nodes = g.nodes[:ni + 1], [ni]
        teacher[n.outputs[0]] = ref_capi.run_model(tm2.write_tm2(g2), x, ref_capi.MODE_UINT8, 4)[0]
    rep = []
    oracle.run_graph(g, x, teacher=teacher, report=rep)
    assert len(rep) == 35          # the benchmark file's 35 compute nodes (two Dropout heads and the single-input route included)
    assert all(r[2] == 0 for r in rep), [r for r in rep if r[2]]


def test_yolov3_tiny_uint8_416_oracle_matches_golden_of_real_reference():
    """runs without the reference: the fixture was produced by the real reference (tests/golden/make_golden.py)."""
    golden = np.load(os.path.join(os.path.dirname(__file__), "golden", "yolov3_tiny_uint8_416_seed3.npz"))
    g = models.build("yolov3_tiny", "uint8", 1)
    x = models.synth_input(g, 3, tm2.DT_UINT8)
    outs = oracle.run_graph(g, x)
    for i, o in enumerate(outs):
        assert np.array_equal(o.ravel(), golden["out%d" % i].ravel())


def test_uint8_quantizer_restatement():
    gf = models.yolov3_tiny_fp32(1, 64)
    g = models.quantize_uint8(gf)
    for t in g.tensors:
        if t.dtype == tm2.DT_UINT8:
            assert len(t.scales) == 1 and 0 <= t.zero_points[0] <= 255 and t.scales[0] > 0
    # relu(slope 0)/max-pool outputs hand their parameters to a single-consumer producer; leaky relu does not
    names = {t.name: t for t in g.tensors}
    assert names["maxpool0/0"].scales == names["leaky0/0"].scales
    assert names["leaky0/0"].scales != names["conv0/0"].scales
    # round trip through the tmfile keeps zero points
    g2 = tm2.read_tm2(tm2.write_tm2(g))
    assert [t.zero_points for t in g2.tensors if t.dtype == tm2.DT_UINT8] == \
           [t.zero_points for t in g.tensors if t.dtype == tm2.DT_UINT8]


@pytest.mark.parametrize("name,dev_only", [("mobilenet_v1", False), ("resnet50", True)])
def test_uint8_classifiers_oracle_matches_golden_of_real_reference(name, dev_only):
    """MobileNet-v1 uint8 (13 depthwise convs on the conv_ref formula, classifier = 1x1 conv on a 1x1 map, i.e. only
    'tail' pixels) and ResNet-50 uint8 (eltwise, max/avg pool, fc): oracle == bytes of the real reference."""
    golden = np.load(os.path.join(os.path.dirname(__file__), "golden", "%s_uint8_seed5.npy" % name))
    g = models.build(name, "uint8", 1, device_only=dev_only)
    x = models.synth_input(g, 5, tm2.DT_UINT8)
    out = oracle.run_graph(g, x)[0]
    assert np.array_equal(out.ravel(), golden.ravel())


def test_mssd_uint8_300_oracle_matches_golden_of_real_reference():
    """MobileNet-SSD 300x300 uint8: 47 convs (13 depthwise on the conv_ref formula) and the Permute -> Flatten ->
    Concat head plumbing; this graph is what exposed the unfused bias add of the reference's uint8 GEMM epilogue."""
    golden = np.load(os.path.join(os.path.dirname(__file__), "golden", "mssd_uint8_300_seed5.npz"))
    g = models.build("mssd", "uint8", 1)
    x = models.synth_input(g, 5, tm2.DT_UINT8)
    outs = oracle.run_graph(g, x)
    assert [o.shape for o in outs] == [(1, 1917 * 4), (1, 1917 * 21)]
    for i, o in enumerate(outs):
        assert np.array_equal(o.ravel(), golden["out%d" % i].ravel())


@pytest.mark.parametrize("tag", ["uint8", "fp32"])
def test_priorbox_oracle_matches_golden_of_real_reference(tag):
    """PriorBox (priorbox_ref.c) outputs of the real reference, committed as tests/golden/priorbox_cases.npz
    (make_golden.py priorbox): the oracle reproduces every byte / every float exactly"""
    from helpers import PRIORBOX_CASES, priorbox_graph
    golden = np.load(os.path.join(os.path.dirname(__file__), "golden", "priorbox_cases.npz"))
    for case, kw in sorted(PRIORBOX_CASES.items()):
        g, x = priorbox_graph(dtype=tm2.DT_UINT8 if tag == "uint8" else tm2.DT_FP32, **kw)
        out = oracle.run_graph(g, x)[0]
        assert np.array_equal(out, golden["%s_%s" % (case, tag)]), case
    if tag == "uint8":
        g = models.build("mssd", "uint8", 1, tail=True, priorbox=True)
        outs = oracle.run_graph(g, models.synth_input(g, 5, tm2.DT_UINT8))
        assert np.array_equal(outs[2], golden["mssd_mbox_priorbox_uint8"])


@needs_ref
def test_softmax_and_reshape_uint8_oracle_is_the_reference():
    """the quantised part of the SSD tail (next row): Softmax (C `exp` on a double, sequential fp32 sum)
    and Reshape, single ops and behind the whole MobileNet-SSD graph."""
    both(*u8_unary_graph(41, "Softmax", [2, 21, 5, 7], axis=1))
    both(*u8_unary_graph(42, "Softmax", [3, 40, 21], axis=2))
    both(*u8_unary_graph(43, "Softmax", [4, 100], axis=1))
    g = models.build("mssd", "uint8", 2, tail=True)
    assert [n.op for n in g.nodes[-3:]] == ["Reshape", "Softmax", "Flatten"]
    both(g, models.synth_input(g, 5, tm2.DT_UINT8))
