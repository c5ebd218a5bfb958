"""int8 Softmax on the device (the tail of the ResNet-50 benchmark graph, appendix C): softmax_i8_kernel through
the C ABI against the oracle restatement of softmax_kernel_ref_int8.c (pinned to the real reference in
tests/test_oracle_vs_reference.py::test_softmax_int8_oracle_equals_reference), and against the reference itself where its
prebuilt library travelled to the box.  Bar: bit-exact."""
import numpy as np
import pytest

from helpers import conv_graph, fc_graph, i8_unary_graph
from oracle import oracle
from tengine_amd import capi, models, tm2

pytestmark = pytest.mark.gpu

CASES = [
    # dims, axis, out_scale
    ([2, 21, 5, 7], 1, None),          # a class axis in front of a map: 70 positions of 21 channels (cs 32: padding bytes stay zero)
    ([4, 1000], 1, 2e-4),              # ResNet-50's prob, 2-D (a scale that spreads 1000-way probabilities over the bytes)
    ([2, 1000, 1, 1], 1, 1e-4),        # .. as a 1x1 map (a classifier that ends in a 1x1 convolution)
    ([3, 2500], -1, 5e-5),             # more than 2048 channels: one wave per block, 10 KB of exponentials
    ([1, 3, 4, 4], 1, None),           # fewer channels than lanes
    ([33, 10], 1, None),               # positions not a multiple of the block's four waves
    ([1, 64, 9, 9], -3, 0.01),         # exactly one element per lane; negative axis
    ([5, 130, 2, 3], 1, 1e-3),         # 2 and a fraction elements per lane
]


def run(g, x):
    gr = capi.Graph(tm2.write_tm2(g))
    gr.set_input(x)
    out = gr.run()
    kernels = [k["kernel"] for k in gr.profile(1)]
    gr.close()
    return out, kernels


@pytest.mark.parametrize("dims,axis,out_scale", CASES, ids=[str(c[0]) for c in CASES])
def test_softmax_i8_equals_the_oracle(dims, axis, out_scale):
    for seed in (61, 62):
        g, x = i8_unary_graph(seed, "Softmax", dims, out_scale=out_scale, axis=axis)
        want = oracle.run_graph(g, x)[0]
        got, kernels = run(g, x)
        assert "softmax_i8" in kernels, kernels
        got = got[0].reshape(want.shape)
        assert np.array_equal(want, got), "%d / %d bytes differ" % (np.count_nonzero(want != got), want.size)
        assert len(np.unique(want)) >= 3


def test_softmax_i8_saturated_and_flat_rows():
    """rows the generators never draw: every logit equal (1 / C everywhere), one logit far above the rest (1.0 saturates to 127 at a
    scale below 1 / 127, the others underflow to exact zeros), the most negative byte"""
    g, _ = i8_unary_graph(7, "Softmax", [4, 40], out_scale=0.005, axis=1)
    x = np.zeros((4, 40), np.int8)
    x[0, :] = 17
    x[1, :] = -127
    x[1, 5] = 127
    x[2, :] = -128
    x[2, 39] = -127
    x[3, :] = np.arange(40) * 6 - 120
    want = oracle.run_graph(g, x)[0]
    got, _ = run(g, x)
    assert np.array_equal(want, got[0].reshape(want.shape))
    assert want[1, 5] == 127 and want[1, 0] == 0 and len(set(want[0].tolist())) == 1


def test_fc_then_softmax_keeps_the_padded_hand_over():
    """fc1000 -> prob as in ResNet-50: the fc output lives in a 1008-byte-per-row buffer, the graph output is dense"""
    g, x = fc_graph(9, 3, (64,), 1000)
    y = g.nodes[-1].outputs[0]
    o = g.add_tensor("prob", [3, 1000], tm2.DT_INT8, tm2.TT_VAR, None, [1.0 / 127.0], [0])
    ni = g.add_node("prob", "Softmax", [y], [o], axis=1)
    g.output_nodes = [ni]
    want = oracle.run_graph(g, x)[0]
    got, kernels = run(g, x)
    assert kernels[-1] == "softmax_i8" and "nhwc_to_nchw" not in kernels, kernels
    assert np.array_equal(want, got[0].reshape(want.shape))
    assert want.max() > 0


def test_conv_then_softmax_over_a_map():
    """a segmentation-style tail: softmax over the channels at every pixel of a convolution's output"""
    g, x = conv_graph(13, 2, 32, 9, 11, 24, 1, act=-1)
    y = g.nodes[-1].outputs[0]
    o = g.add_tensor("prob", list(g.tensors[y].dims), tm2.DT_INT8, tm2.TT_VAR, None, [0.9 / 127.0], [0])
    ni = g.add_node("prob", "Softmax", [y], [o], axis=1)
    g.output_nodes = [ni]
    want = oracle.run_graph(g, x)[0]
    got, kernels = run(g, x)
    assert "softmax_i8" in kernels
    assert np.array_equal(want, got[0].reshape(want.shape))


def test_resnet50_int8_with_its_softmax_bit_exact():
    """BASELINE configs[2]'s graph as the reference's benchmark runs it -- .. pool5, fc1000, prob -- in one device graph; against
    the real reference where it is on the box, the oracle otherwise"""
    from oracle import ref_capi
    g = models.build("resnet50", "int8", 2)
    assert g.nodes[-1].op == "Softmax"
    x = models.synth_input(g, 6)
    b = tm2.write_tm2(g)
    want = ref_capi.run_model(b, x, ref_capi.MODE_INT8, 8)[0] if ref_capi.available() else oracle.run_graph(g, x)[0]
    got, kernels = run(g, x)
    assert kernels[-1] == "softmax_i8", kernels[-3:]
    assert np.array_equal(np.asarray(want).reshape(got[0].shape), got[0])
    assert got[0].max() > 0
