"""Known-answer tests restated from the reference's own op tests (tests/golden/kat_reference_tests.json):
the oracle, the real reference (when built) and -- on the GPU box -- the HIP backend must all reproduce
the reference test's expected values within 1.5 output quantisation steps (int8 operand noise), and agree with each other bit for bit."""
import json
import os

import numpy as np
import pytest

from oracle import oracle
from tengine_amd import tm2
from tengine_amd.tm2 import DT_INT8, DT_INT32, Graph

CASES = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat_reference_tests.json")))["cases"]


def build(case):
    g = Graph(name=case["name"])
    x = g.add_input("input_node", case["input_dims"], DT_INT8, [case["input_scale"]], [0])
    wd = case["weight_dims"]
    w = g.add_const("weight", np.array(case["weight_i8"], np.int8).reshape(wd), DT_INT8, case["weight_scales"], [0] * wd[0])
    b = g.add_const("bias", np.array(case["bias_i32"], np.int32), DT_INT32, [1.0], [0])
    p = dict(case["conv"])
    n, c, h, wdt = case["input_dims"]
    oh = (h - p["kernel_h"] + p["pad_h0"] + p["pad_h1"]) // p["stride_h"] + 1
    ow = (wdt - p["kernel_w"] + p["pad_w0"] + p["pad_w1"]) // p["stride_w"] + 1
    y = g.add_tensor("conv", [n, wd[0], oh, ow], DT_INT8, tm2.TT_VAR, None, [case["output_scale"]], [0])
    ni = g.add_node("conv", "Convolution", [x, w, b], [y], input_channel=c, output_channel=wd[0], group=case["group"], **p)
    g.output_nodes = [ni]
    return g, np.array(case["input_i8"], np.int8).reshape(case["input_dims"])


def _check(case, out_i8):
    got = out_i8.astype(np.float32).ravel() * np.float32(case["output_scale"])
    want = np.array(case["reference_out_fp32"], np.float32)
    # the expected values are the fp32 results; int8 operands carry input/weight quantisation noise on top
    # of the output step, and the REAL reference CPU backend lands at the same distance (max 0.283 on these
    # vectors) -- so the KAT pins semantics at 1.5 output steps, bit-exactness is asserted separately
    assert np.abs(got - want).max() <= 1.5 * case["output_scale"], (got, want)


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_oracle_reproduces_reference_kat(case):
    g, x = build(case)
    _check(case, oracle.run_graph(g, x)[0])


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_real_reference_reproduces_its_own_kat(ref, case):
    g, x = build(case)
    out = ref.run_model(tm2.write_tm2(g), x, ref.MODE_INT8, 1)[0]
    _check(case, out)
    assert np.array_equal(out, oracle.run_graph(g, x)[0])


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_hip_reproduces_reference_kat(case):
    from tengine_amd import capi
    g, x = build(case)
    gr = capi.Graph(tm2.write_tm2(g))
    gr.set_input(x)
    out = gr.run()[0]
    gr.close()
    _check(case, out)
    assert np.array_equal(out.ravel(), oracle.run_graph(g, x)[0].ravel())
