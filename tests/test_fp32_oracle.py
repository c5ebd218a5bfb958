"""fp32 (parity bar 1e-4): the oracle's fp32 restatement (double accumulation) against the real
reference's fp32 kernels (im2col + sgemm, Winograd F(4,3) for the eligible 3x3 layers, direct depthwise)."""
import numpy as np
import pytest

from oracle import oracle, ref_capi
from tengine_amd import models, tm2

TOL = dict(rtol=1e-4, atol=1e-4)
needs_ref = pytest.mark.skipif(not ref_capi.available(), reason="reference library not built (oracle/build_ref.py)")


@needs_ref
@pytest.mark.parametrize("name", ["squeezenet_v1.1", "mobilenet_v1"])
def test_fp32_models_oracle_matches_reference(name):
    g = models.build(name, "fp32", 1)
    x = models.synth_input(g, 5, tm2.DT_FP32)
    want = ref_capi.run_model(tm2.write_tm2(g), x, ref_capi.MODE_FP32, 4)
    got = oracle.run_graph(g, x)
    for w, o in zip(want, got):
        assert np.allclose(w, o.reshape(w.shape), **TOL)
        assert np.abs(w).max() > 1e-3
