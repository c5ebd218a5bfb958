"""Round 6 (the int8 remainder): Permute / Flatten / Reshape / Concat / Softmax / PriorBox of an int8 graph.  CPU half:
the oracle restatement against the REAL reference on the cases the GPU tests use (tests/test_gpu_int8_heads.py checks the device against
the oracle on the same cases), and int8 MobileNet-SSD with its tail and priors."""
import numpy as np
import pytest

from helpers import I8_HEAD_CASES, PRIORBOX_CASES, i8_head_graph, priorbox_graph, single_input_concat_graph
from oracle import oracle, ref_capi
from tengine_amd import models, tm2

needs_ref = pytest.mark.skipif(not ref_capi.available(), reason="oracle/_ref/libtengine-lite.so not built")


@needs_ref
@pytest.mark.parametrize("case", sorted(I8_HEAD_CASES))
def test_oracle_equals_reference_on_the_int8_head_cases(case):
    g, x = i8_head_graph(**I8_HEAD_CASES[case])
    want = ref_capi.run_model(tm2.write_tm2(g), x, ref_capi.MODE_INT8, 2)
    got = oracle.run_graph(g, x)
    assert len(want) == len(got)
    for w, o in zip(want, got):
        assert np.array_equal(np.asarray(w).ravel(), np.asarray(o).ravel()), case
        assert len(np.unique(w)) > 3, "degenerate case: %s" % case


@needs_ref
@pytest.mark.parametrize("kw", [dict(), dict(tail=True), dict(tail=True, priorbox=True)], ids=["heads", "tail", "tail_priorbox"])
def test_oracle_equals_reference_on_int8_mobilenet_ssd(kw):
    g = models.build("mssd", "int8", 1, **kw)
    x = models.synth_input(g, 5, tm2.DT_INT8)
    want = ref_capi.run_model(tm2.write_tm2(g), x, ref_capi.MODE_INT8, 4)
    got = oracle.run_graph(g, x)
    for w, o in zip(want, got):
        assert np.array_equal(np.asarray(w).ravel(), np.asarray(o).ravel())


@needs_ref
@pytest.mark.parametrize("case", sorted(PRIORBOX_CASES))
def test_oracle_equals_reference_on_int8_priorbox(case):
    """priorbox_ref.c:195-210: the int8 quantisation ROUNDS (the uint8 one truncates), clamp +-127; the Concat(axis 2) rescales"""
    g, x = priorbox_graph(dtype=tm2.DT_INT8, **PRIORBOX_CASES[case])
    want = ref_capi.run_model(tm2.write_tm2(g), x, ref_capi.MODE_INT8, 2)
    got = oracle.run_graph(g, x)
    for w, o in zip(want, got):
        assert np.array_equal(np.asarray(w).ravel(), np.asarray(o).ravel()), case
        assert len(np.unique(w)) > 5


@needs_ref
@pytest.mark.parametrize("dtype", ["int8", "uint8"])
def test_a_single_input_concat_is_a_byte_copy_in_the_reference(dtype):
    """round 6 (tools/fuzz_heads.py --ref found it): a Concat of ONE tensor copies the bytes, it does not rescale them, whatever the two
    tensors' quantisation says -- the oracle restated the rescale until then (and the uint8 planner launched it)"""
    dt = tm2.DT_UINT8 if dtype == "uint8" else tm2.DT_INT8
    g, x = single_input_concat_graph(3, dt)
    want = ref_capi.run_model(tm2.write_tm2(g), x, ref_capi.MODE_UINT8 if dtype == "uint8" else ref_capi.MODE_INT8, 1)[0]
    got = oracle.run_graph(g, x)[0]
    assert np.array_equal(np.asarray(want).ravel(), np.asarray(got).ravel())
    relu_only = oracle.run_graph(g, x, keep_all=True)
    r = [i for i, t in enumerate(g.tensors) if t.name == "r"][0]
    assert np.array_equal(np.asarray(want).ravel(), np.asarray(relu_only[r]).ravel())       # .. i.e. the ReLU's bytes, untouched
