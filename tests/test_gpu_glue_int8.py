"""int8 glue the planner used to refuse: concat inputs that cannot be written in place are
copied with the reference's re-scaling arithmetic (concat_kernel_ref_int8.c:70-80, incl. its +127 lower clamp), Flatten
of an H x W map keeps the NCHW element order through a following FC or a graph output.  Bit-exact vs the oracle (pinned
to the real reference for these ops in tests/test_oracle_vs_reference.py)."""
import numpy as np
import pytest

from helpers import _scales, conv_graph, i8_concat_graph
from oracle import oracle
from tengine_amd import capi, tm2
from tengine_amd.tm2 import DT_INT8, DT_INT32

pytestmark = pytest.mark.gpu


def run(g, x):
    gr = capi.Graph(tm2.write_tm2(g))
    gr.set_input(x)
    out = gr.run()
    names = [k["kernel"] for k in gr.profile(1)]
    gr.close()
    return out, names


def check(g, x):
    want = oracle.run_graph(g, x)
    got, names = run(g, x)
    for w, o in zip(want, got):
        assert np.array_equal(o.reshape(w.shape), w), (names, np.count_nonzero(o.reshape(w.shape) != w))
        assert len(np.unique(w)) > 3
    return names


@pytest.mark.parametrize("shrink", [False, True])
@pytest.mark.parametrize("shape", [(2, 8, 5, 6), (1, 24, 7, 3), (3, 16, 4, 4)])
def test_rescaling_concat(shape, shrink):
    """ReLU / leaky-ReLU results with their own scales into one concat: both inputs are re-scaled copies; `shrink` drives
    values past -127, where the reference writes +127"""
    g, x = i8_concat_graph(40 + shape[1], *shape, axis=1, shrink=shrink)
    names = check(g, x)
    assert names.count("concat_copy_i8") == 2, names


def _two_branch_concat(seed, n, cin, h, w, ca, cb, same_scale_a=True, twice=False):
    """data -> convA (ca channels) ; data -> convB (cb channels) ; concat -> 1x1 conv.  convA carries the concat's scale
    (written in place when ca % 16 == 0), convB its own (copied with re-scaling)."""
    rng = np.random.default_rng(seed)
    g, x = conv_graph(seed, n, cin, h, w, ca, 3, 1, 1)
    g.output_nodes = []
    xin = g.nodes[g.input_nodes[0]].outputs[0]
    a = g.nodes[-1].outputs[0]
    sa = g.tensors[a].scales[0]
    wq = rng.integers(-127, 128, size=(cb, cin, 1, 1)).astype(np.int8)
    wb = g.add_const("wb", wq, DT_INT8, _scales(rng, cb), [0] * cb)
    b = g.add_tensor("b_out", [n, cb, h, w], DT_INT8, tm2.TT_VAR, None, [float(np.float32(sa * 0.37))], [0])
    g.add_node("convB", "Convolution", [xin, wb], [b], kernel_h=1, kernel_w=1, stride_h=1, stride_w=1, dilation_h=1, dilation_w=1,
               input_channel=cin, output_channel=cb, group=1, activation=0, pad_h0=0, pad_w0=0, pad_h1=0, pad_w1=0)
    ins = [a, b, a] if twice else [a, b]
    ctot = sum(g.tensors[i].dims[1] for i in ins)
    cs = sa if same_scale_a else float(np.float32(sa * 1.21))
    c = g.add_tensor("cat", [n, ctot, h, w], DT_INT8, tm2.TT_VAR, None, [cs], [0])
    g.add_node("cat", "Concat", ins, [c], axis=1)
    wq2 = rng.integers(-127, 128, size=(20, ctot, 1, 1)).astype(np.int8)
    ws2 = _scales(rng, 20)
    w2 = g.add_const("w2", wq2, DT_INT8, ws2, [0] * 20)
    o = g.add_tensor("out", [n, 20, h, w], DT_INT8, tm2.TT_VAR, None, [float(np.float32(cs * np.mean(ws2) * 73.0 * np.sqrt(ctot) * 73.0 / 60.0))], [0])
    ni = g.add_node("head", "Convolution", [c, w2], [o], kernel_h=1, kernel_w=1, stride_h=1, stride_w=1, dilation_h=1, dilation_w=1,
                    input_channel=ctot, output_channel=20, group=1, activation=-1, pad_h0=0, pad_w0=0, pad_h1=0, pad_w1=0)
    g.output_nodes = [ni]
    return g, x


def test_concat_mixes_in_place_and_copied_inputs():
    g, x = _two_branch_concat(51, 2, 16, 9, 9, 32, 24)          # convA: view at offset 0; convB: 24 channels, own scale -> copy
    names = check(g, x)
    assert names.count("concat_copy_i8") == 1, names


def test_concat_unaligned_offsets_and_ragged_channels():
    g, x = _two_branch_concat(52, 1, 16, 6, 7, 20, 12)          # 20 + 12 channels: nothing is 16-aligned -> two copies
    names = check(g, x)
    assert names.count("concat_copy_i8") == 2, names


def test_concat_all_scales_differ():
    g, x = _two_branch_concat(53, 2, 32, 5, 5, 16, 16, same_scale_a=False)
    names = check(g, x)
    assert names.count("concat_copy_i8") == 2, names


def test_concat_same_tensor_twice():
    g, x = _two_branch_concat(54, 1, 16, 6, 6, 16, 16, twice=True)
    names = check(g, x)
    assert names.count("concat_copy_i8") == 3, names              # a read twice: neither position is written in place


@pytest.mark.parametrize("as_output", [False, True])
def test_flatten_of_a_map(as_output):
    """conv -> Flatten -> FC: the FC weight indexes [c][h][w] (NCHW flatten order) while the device tensor is NHWC"""
    rng = np.random.default_rng(61)
    g, x = conv_graph(61, 3, 8, 5, 4, 12, 3, 1, 1)
    a = g.nodes[-1].outputs[0]
    sa = g.tensors[a].scales[0]
    f = g.add_tensor("flat", [3, 12 * 5 * 4], DT_INT8, tm2.TT_VAR, None, [sa], [0])
    ni = g.add_node("flatten", "Flatten", [a], [f], axis=1, end_axis=3)
    if not as_output:
        hidden, nout = 12 * 5 * 4, 10
        wq = rng.integers(-127, 128, size=(nout, hidden)).astype(np.int8)
        ws = _scales(rng, nout)
        wt = g.add_const("wfc", wq, DT_INT8, ws, [0] * nout)
        bt = g.add_const("bfc", rng.integers(-2000, 2000, size=(nout,)).astype(np.int32), DT_INT32, [1.0], [0])
        o = g.add_tensor("fc_out", [3, nout], DT_INT8, tm2.TT_VAR, None, [float(np.float32(sa * np.mean(ws) * 73.0 * np.sqrt(hidden) * 73.0 / 60.0))], [0])
        ni = g.add_node("fc", "FullyConnected", [f, wt, bt], [o], num_output=nout)
    g.output_nodes = [ni]
    check(g, x)
