"""Pins the oracle restatement (oracle/tg_oracle.c) against the REAL reference library built from the
unmodified sources (oracle/_ref, see oracle/build_ref.py): bit-exact for int8, on seeded random layers
covering every kernel-selection branch of the reference and on whole models."""
import numpy as np
import pytest

from helpers import (PRIORBOX_CASES, axis_concat_graph, conv_graph, eltwise_relu_graph, fc_graph, i8_concat_graph, i8_unary_graph,
                     pool_graph, priorbox_graph)
from oracle import oracle
from tengine_amd import models, tm2

CONV_CASES = [
    # (n, cin, h, w, cout, k, s, p, group, act, bias, dil)   -> reference kernel
    (1, 32, 14, 14, 64, 1, 1, 0, 1, 0, True, 1),      # hcl 1x1
    (1, 64, 9, 11, 48, 1, 1, 0, 1, -1, False, 1),     # hcl 1x1 no bias / no act
    (2, 16, 8, 8, 32, 1, 2, 0, 1, 6, True, 1),        # hcl 1x1 stride 2, relu6, batch 2
    (1, 3, 32, 32, 32, 3, 2, 1, 1, 0, True, 1),       # first layer 3x3 s2
    (1, 16, 13, 13, 24, 3, 1, 1, 1, 0, True, 1),      # hcl 3x3
    (2, 8, 12, 12, 16, 3, 1, 2, 1, 1, True, 2),       # dilation 2, act code 1 (treated as relu6 by hcl)
    (1, 3, 40, 40, 16, 7, 2, 3, 1, 0, True, 1),       # 7x7 s2 (ResNet stem)
    (1, 32, 16, 16, 32, 3, 1, 1, 32, 0, True, 1),     # dw s1 batch 1 -> dw_hcl
    (1, 48, 15, 15, 48, 3, 2, 1, 48, 0, True, 1),     # dw s2 batch 1 -> dw_hcl
    (2, 32, 10, 10, 32, 3, 1, 1, 32, 0, True, 1),     # dw batch 2 -> ref_conv_int8 (other epilogue)
    (1, 32, 10, 10, 32, 3, 1, 1, 32, 1, True, 1),     # dw act=1: hcl treats as relu6
    (2, 32, 10, 10, 32, 3, 1, 1, 32, 1, True, 1),     # dw batch 2 act=1: ref clamps [-1,1]
    (1, 16, 9, 9, 32, 3, 1, 1, 4, 0, True, 1),        # grouped (non-dw) -> ref
    (1, 24, 9, 9, 24, 5, 1, 2, 24, 6, True, 1),       # dw 5x5 -> ref, relu6
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[str(c) for c in CONV_CASES])
def test_conv_int8_oracle_equals_reference(ref, case):
    n, cin, h, w, cout, k, s, p, group, act, bias, dil = case
    for seed in (11, 12):
        g, x = conv_graph(seed, n, cin, h, w, cout, k, s, p, group, act, bias, dil)
        want = ref.run_model(tm2.write_tm2(g), x, ref.MODE_INT8, 2)[0]
        got = oracle.run_graph(g, x)[0]
        assert want.shape == got.shape
        assert np.array_equal(want, got), "mismatches: %d / %d" % (np.count_nonzero(want != got), want.size)
        assert np.count_nonzero(got) > got.size // 8      # non-degenerate


@pytest.mark.parametrize("case", [(1, (64,), 10), (3, (32, 2, 2), 17), (2, (2048,), 100)])
def test_fc_int8_oracle_equals_reference(ref, case):
    n, hd, nout = case
    g, x = fc_graph(5, n, hd, nout)
    want = ref.run_model(tm2.write_tm2(g), x, ref.MODE_INT8, 1)[0]
    got = oracle.run_graph(g, x)[0]
    assert np.array_equal(want.reshape(got.shape), got)


POOL_CASES = [
    # n, c, h, w, alg, k, s, p, glob, caffe
    (1, 32, 7, 7, 1, 7, 1, 0, 1, 1),      # global avg (MobileNet / ResNet tail)
    (2, 16, 14, 14, 0, 3, 2, 0, 0, 1),    # caffe max 3x3 s2 (ResNet pool1 / SqueezeNet)
    (1, 16, 15, 15, 0, 3, 2, 1, 0, 0),    # padded max
    (1, 8, 12, 12, 1, 3, 2, 1, 0, 0),     # padded avg, pool_size counts in-image taps
    (1, 8, 12, 12, 1, 3, 2, 1, 0, 1),     # caffe avg, pool_size counts the padded window
    (1, 8, 13, 13, 0, 2, 2, 0, 0, 0),     # yolo maxpool
]


@pytest.mark.parametrize("case", POOL_CASES, ids=[str(c) for c in POOL_CASES])
def test_pool_int8_oracle_equals_reference(ref, case):
    n, c, h, w, alg, k, s, p, glob, caffe = case
    g, x = pool_graph(3, n, c, h, w, alg, k, s, p, glob, caffe)
    want = ref.run_model(tm2.write_tm2(g), x, ref.MODE_INT8, 1)[0]
    got = oracle.run_graph(g, x)[0]
    assert want.shape == got.shape and np.array_equal(want, got)


@pytest.mark.parametrize("with_relu", [False, True])
def test_eltwise_relu_int8_oracle_equals_reference(ref, with_relu):
    g, x = eltwise_relu_graph(9, 2, 32, 6, 6, with_relu)
    want = ref.run_model(tm2.write_tm2(g), x, ref.MODE_INT8, 1)[0]
    got = oracle.run_graph(g, x)[0]
    assert np.array_equal(want, got)


@pytest.mark.parametrize("axis,shrink", [(1, False), (1, True), (2, True), (3, True), (0, True)])
def test_concat_int8_oracle_equals_reference(ref, axis, shrink):
    """int8 concat with per-input rescale (concat_kernel_ref_int8.c), incl. the reference's lower clamp, which writes
    +127 for values below -127 at every one of its ten sites: restated as it is (`shrink` exercises it)."""
    g, x = i8_concat_graph(30 + axis, 2, 8, 5, 6, axis, shrink)
    want = ref.run_model(tm2.write_tm2(g), x, ref.MODE_INT8, 1)[0]
    got = oracle.run_graph(g, x)[0]
    assert np.array_equal(want, got.reshape(want.shape))
    if shrink:
        assert (want == 127).sum() > 100 and want.min() >= -127      # negative inputs came out as +127


def test_mobilenet_v1_int8_whole_model(ref):
    g = models.build("mobilenet_v1", "int8", 1)
    x = models.synth_input(g, 7)
    b = tm2.write_tm2(g)
    want = ref.run_model(b, x, ref.MODE_INT8, 4)[0]
    got = oracle.run_graph(g, x)[0]
    assert np.array_equal(want, got)
    assert np.abs(got.astype(int)).max() > 60
    # thread-count independence of the reference int8 path (SURVEY §8c determinism)
    assert np.array_equal(want, ref.run_model(b, x, ref.MODE_INT8, 1)[0])


def test_resnet50_int8_whole_model(ref):
    """53 conv (1x1, 1x1 s2, 3x3, 7x7 s2) + 16 eltwise + 16 relu + max/avg pool + fc: oracle == real reference on the logits
    (the graph with its Softmax: test_resnet50_int8_whole_model_with_its_softmax)."""
    g = models.build("resnet50", "int8", 1, device_only=True)
    x = models.synth_input(g, 5)
    want = ref.run_model(tm2.write_tm2(g), x, ref.MODE_INT8, 8)[0]
    got = oracle.run_graph(g, x)[0]
    assert np.array_equal(want.reshape(got.shape), got)
    assert np.abs(got.astype(int)).max() > 40


SOFTMAX_I8_CASES = [([2, 21, 5, 7], 1, None), ([3, 40, 21], 2, None), ([4, 1000], 1, 2e-4), ([2, 1000, 1, 1], 1, 1e-4),
                    ([2, 6, 9], 1, None), ([1, 7, 3, 5], 3, None), ([5, 64], -1, 0.02)]


@pytest.mark.parametrize("dims,axis,out_scale", SOFTMAX_I8_CASES, ids=[str(c) for c in SOFTMAX_I8_CASES])
def test_softmax_int8_oracle_equals_reference(ref, dims, axis, out_scale):
    """softmax_kernel_ref_int8.c: dequantise, C `exp` on a double rounded to float, sequential fp32 sum over the
    axis, requantise with clamp +-127 -- any axis, any rank"""
    for seed in (51, 52):
        g, x = i8_unary_graph(seed, "Softmax", dims, out_scale=out_scale, axis=axis)
        want = ref.run_model(tm2.write_tm2(g), x, ref.MODE_INT8, 1)[0]
        got = oracle.run_graph(g, x)[0]
        assert np.array_equal(np.asarray(want).reshape(got.shape), got)
        assert len(np.unique(got)) >= 3


def test_resnet50_int8_whole_model_with_its_softmax(ref):
    """the benchmark graph as the reference runs it (.. fc, 1 softmax): oracle == real reference"""
    g = models.build("resnet50", "int8", 1)
    assert g.nodes[-1].op == "Softmax"
    x = models.synth_input(g, 6)
    want = ref.run_model(tm2.write_tm2(g), x, ref.MODE_INT8, 8)[0]
    got = oracle.run_graph(g, x)[0]
    assert np.array_equal(want.reshape(got.shape), got)
    assert got.max() > 0


@pytest.mark.parametrize("dtype", [tm2.DT_UINT8, tm2.DT_FP32], ids=["uint8", "fp32"])
@pytest.mark.parametrize("case", sorted(PRIORBOX_CASES))
def test_priorbox_oracle_equals_reference(ref, case, dtype):
    """priorbox_ref.c:53-213 -- int truncation of the sizes, double sqrt, flipped priors, clip, explicit step /
    image size, the truncating uint8 quantisation; fp32 compared exactly too (same operations in the same order)"""
    g, x = priorbox_graph(dtype=dtype, **PRIORBOX_CASES[case])
    want = ref.run_model(tm2.write_tm2(g), x, ref.MODE_UINT8 if dtype == tm2.DT_UINT8 else ref.MODE_FP32, 2)
    got = oracle.run_graph(g, x)
    assert len(want) == len(got) == 1
    assert np.array_equal(np.asarray(want[0]).reshape(got[0].shape), got[0])
    assert len(np.unique(got[0])) > 20


@pytest.mark.parametrize("dtype", [tm2.DT_UINT8, tm2.DT_FP32], ids=["uint8", "fp32"])
@pytest.mark.parametrize("dims,axis", [([1, 5, 6, 7], 2), ([2, 3, 4, 9], 3), ([2, 8, 5], -1), ([3, 6, 4, 4], 1)])
def test_concat_any_axis_oracle_equals_reference(ref, dims, axis, dtype):
    g, x = axis_concat_graph(21 + axis, dtype, dims, axis)
    want = ref.run_model(tm2.write_tm2(g), x, ref.MODE_UINT8 if dtype == tm2.DT_UINT8 else ref.MODE_FP32, 2)[0]
    got = oracle.run_graph(g, x)[0]
    assert np.array_equal(np.asarray(want).reshape(got.shape), got)


def test_mssd_full_tail_oracle_equals_reference(ref):
    """the whole uint8 MobileNet-SSD up to detection_output's three inputs: mbox_loc, softmaxed mbox_conf, mbox_priorbox"""
    g = models.build("mssd", "uint8", 1, tail=True, priorbox=True)
    x = models.synth_input(g, 5, tm2.DT_UINT8)
    want = ref.run_model(tm2.write_tm2(g), x, ref.MODE_UINT8, 8)
    got = oracle.run_graph(g, x)
    assert [tuple(o.shape) for o in got] == [(1, 7668), (1, 40257), (1, 2, 7668, 1)]
    for w, o in zip(want, got):
        assert np.array_equal(np.asarray(w).reshape(o.shape), o)
