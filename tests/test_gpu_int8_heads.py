"""Round 6 (the int8 remainder; VERDICT r5 missing #4): the SSD head plumbing of an INT8 graph on the device --
Permute(0,2,3,1) (permute_ref.c:305-343), Flatten (flatten_ref.c:74-80), Reshape (reshape_ref.c:76-90), Concat on any axis of dense
tensors (concat_kernel_ref_int8.c), Softmax over any axis (softmax_kernel_ref_int8.c:41-117), PriorBox (priorbox_ref.c:195-210).
Device against the oracle (tests/test_int8_heads_oracle.py pins the oracle to the real reference on the same cases), bit-exact."""
import numpy as np
import pytest

from helpers import I8_HEAD_CASES, PRIORBOX_CASES, i8_head_graph, priorbox_graph, single_input_concat_graph
from oracle import oracle
from tengine_amd import capi, models, tm2

pytestmark = pytest.mark.gpu


def run_device(g, x, **kw):
    gr = capi.Graph(tm2.write_tm2(g), **kw)
    gr.set_input(x)
    out = gr.run()
    kernels = [k["kernel"] for k in gr.profile(1)]
    out2 = gr.run()                                   # a second run after the profile pass: same bytes
    gr.close()
    for a, b in zip(out, out2):
        assert np.array_equal(a, b)
    return out, kernels


@pytest.mark.parametrize("direct", [False, True], ids=["hipgraph", "direct"])
@pytest.mark.parametrize("case", sorted(I8_HEAD_CASES))
def test_int8_head_cases_bit_exact(case, direct):
    g, x = i8_head_graph(**I8_HEAD_CASES[case])
    want = oracle.run_graph(g, x)
    got, kernels = run_device(g, x, direct_dispatch=direct)
    assert len(want) == len(got)
    for w, o in zip(want, got):
        assert np.array_equal(np.asarray(w).ravel(), o.ravel()), (case, kernels)
    tail = I8_HEAD_CASES[case]["tail"]
    if tail == "concat":
        # the Permute / Flatten nodes launch nothing: the Concat reads the convolutions' NHWC buffers itself, <= 8 inputs per launch
        assert not any(k in ("permute_i8", "reshape_i8") for k in kernels), kernels
        assert kernels.count("permute_concat_i8") == (len(I8_HEAD_CASES[case]["couts"]) + 7) // 8, kernels
    elif tail == "permute":
        assert kernels[-1] == "permute_i8", kernels
    elif tail == "flatcat":
        assert kernels[-1] == "flatcat_i8", kernels
    else:
        assert "softmax_i8" in kernels, kernels


@pytest.mark.parametrize("batch", [1, 2])
@pytest.mark.parametrize("kw", [dict(), dict(tail=True)], ids=["heads", "tail"])
def test_int8_mobilenet_ssd_is_one_device_graph(kw, batch):
    """int8 MobileNet-SSD (the reference's mssd graph quantised like the int8 classifiers): 47 convolutions, 12 Permute -> Flatten ->
    2 Concat heads, with `tail` the Reshape -> Softmax(axis 2) -> Flatten on mbox_conf -- every node a device launch or a view"""
    g = models.build("mssd", "int8", batch, **kw)
    x = models.synth_input(g, 5, tm2.DT_INT8)
    want = oracle.run_graph(g, x)
    got, kernels = run_device(g, x, direct_dispatch=True)
    for w, o in zip(want, got):
        assert np.array_equal(np.asarray(w).ravel(), o.ravel())
        assert len(np.unique(o)) > 20
    assert kernels.count("permute_concat_i8") == 2, kernels      # six heads per Concat, one launch each
    if kw:
        assert "softmax_i8" in kernels


def test_int8_mobilenet_ssd_with_priors():
    """+ six PriorBox nodes and their Concat(axis 2): evaluated and copied ONCE at prerun (no launch in a run), quantised as
    priorbox_ref.c:195-210; batch 1 only (the reference fills image 0 only)"""
    g = models.build("mssd", "int8", 1, tail=True, priorbox=True)
    x = models.synth_input(g, 5, tm2.DT_INT8)
    want = oracle.run_graph(g, x)
    got, kernels = run_device(g, x, direct_dispatch=True)
    assert len(got) == 3
    for w, o in zip(want, got):
        assert np.array_equal(np.asarray(w).ravel(), o.ravel())
    assert kernels.count("permute_concat_i8") == 2 and "flatcat_i8" not in kernels, kernels     # the priors' Concat ran at prerun


def test_dense_tensor_feeding_a_convolution_is_refused():
    """a tensor in dense (permuted) order can only be re-read (Flatten / Reshape / Concat / Softmax): the planner says so instead of
    mis-reading it as NHWC"""
    from tengine_amd.tm2 import DT_INT8, DT_INT32
    g, x = i8_head_graph(**I8_HEAD_CASES["standalone_permute"])
    pm = g.nodes[g.output_nodes[0]].outputs[0]
    d = g.tensors[pm].dims
    rng = np.random.default_rng(0)
    w = g.add_const("w_bad", rng.integers(-127, 128, size=(8, d[1], 1, 1)).astype(np.int8), DT_INT8, [0.01] * 8, [0] * 8)
    b = g.add_const("b_bad", np.zeros(8, np.int32), DT_INT32, [1.0], [0])
    y = g.add_tensor("bad", [d[0], 8, d[2], d[3]], DT_INT8, tm2.TT_VAR, None, [0.05], [0])
    g.output_nodes = [g.add_node("bad", "Convolution", [pm, w, b], [y], kernel_h=1, kernel_w=1, stride_h=1, stride_w=1, dilation_h=1, dilation_w=1,
                                 input_channel=d[1], output_channel=8, group=1, activation=-1, pad_h0=0, pad_w0=0, pad_h1=0, pad_w1=0)]
    with pytest.raises(capi.TamdError, match="dense"):
        capi.Graph(tm2.write_tm2(g))


@pytest.mark.parametrize("case", sorted(PRIORBOX_CASES))
def test_int8_priorbox_cases(case):
    """PriorBox depends on shapes only: evaluated once at prerun with the reference's arithmetic types (graph_infer.hip priorbox_eval),
    quantised as priorbox_ref.c:195-210 (round, clamp +-127); its Concat(axis 2) runs once at prerun too -- no launch in a run"""
    g, x = priorbox_graph(dtype=tm2.DT_INT8, **PRIORBOX_CASES[case])
    want = oracle.run_graph(g, x)
    got, kernels = run_device(g, x)
    for w, o in zip(want, got):
        assert np.array_equal(np.asarray(w).ravel(), o.ravel()), case
    assert "flatcat_i8" not in kernels, kernels


@pytest.mark.parametrize("dtype", ["int8", "uint8"])
def test_a_single_input_concat_is_a_byte_copy_on_the_device(dtype):
    """concat_kernel_ref_int8.c:47-57 / concat_kernel_ref_uint8.c:47-58: a lone input is copied as it is, not rescaled (the uint8 planner
    rescaled it until round 6; tests/test_int8_heads_oracle.py pins the oracle's side of this to the real reference)"""
    dt = tm2.DT_UINT8 if dtype == "uint8" else tm2.DT_INT8
    g, x = single_input_concat_graph(3, dt)
    want = oracle.run_graph(g, x)[0]
    got, _ = run_device(g, x)
    assert np.array_equal(np.asarray(want).ravel(), got[0].ravel())
