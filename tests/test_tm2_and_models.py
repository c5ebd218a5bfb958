import os

import numpy as np
import pytest

from tengine_amd import models, tm2


def test_tm2_roundtrip_small():
    from helpers import conv_graph
    g, _ = conv_graph(1, 1, 16, 9, 9, 32, 3, 1, 1)
    b = tm2.write_tm2(g)
    g2 = tm2.read_tm2(b)
    assert [n.op for n in g2.nodes] == [n.op for n in g.nodes]
    for a, c in zip(g.tensors, g2.tensors):
        assert a.dims == c.dims and a.dtype == c.dtype and a.ttype == c.ttype and a.name == c.name
        if a.data is not None:
            assert np.array_equal(a.data, c.data)
        if a.scales is not None:
            assert np.array_equal(np.float32(a.scales), np.float32(c.scales))
    conv = [n for n in g2.nodes if n.op == "Convolution"][0]
    assert conv.params["kernel_h"] == 3 and conv.params["pad_w1"] == 1 and conv.params["activation"] == 0


@pytest.mark.parametrize("name,ref_file", [("mobilenet_v1", "mobilenet"), ("resnet50", "resnet50"),
                                           ("squeezenet_v1.1", "squeezenet_v1.1")])
def test_topology_matches_reference_benchmark_model(name, ref_file):
    """Our synthetic topologies are node-for-node the graphs the reference benchmarks."""
    path = "/root/reference/benchmark/models/%s_benchmark.tmfile" % ref_file
    if not os.path.exists(path):
        pytest.skip("reference tree not present")
    gr = tm2.read_tm2(open(path, "rb").read())
    gm = models.BUILDERS[name]()
    ops_r = [(n.op, n.params) for n in gr.nodes if n.op not in ("Const", "InputOp")]
    ops_m = [(n.op, n.params) for n in gm.nodes if n.op not in ("Const", "InputOp")]
    assert [o for o, _ in ops_r] == [o for o, _ in ops_m]
    for (op, pr), (_, pm) in zip(ops_r, ops_m):
        if op == "Convolution":
            for k in ("kernel_h", "kernel_w", "stride_h", "stride_w", "group", "activation", "pad_h0", "pad_w0",
                      "pad_h1", "pad_w1", "output_channel", "dilation_h"):
                assert pr[k] == pm[k], (k, pr, pm)
        if op == "Pooling":
            for k in ("alg", "kernel_h", "stride_h", "global", "caffe_flavor", "pad_h0"):
                assert pr[k] == pm[k], (k, pr, pm)
    wr = [t.dims for t in gr.tensors if t.ttype == tm2.TT_CONST]
    wm = [t.dims for t in gm.tensors if t.ttype == tm2.TT_CONST]
    assert sorted(wr) == sorted(wm)          # tensor order in the file differs, the multiset must not


def test_quantizer_conventions():
    gf = models.mobilenet_v1_fp32()
    g = models.quantize_int8(gf)
    for n in g.nodes:
        if n.op != "Convolution":
            continue
        w, b = g.tensors[n.inputs[1]], g.tensors[n.inputs[2]]
        assert w.dtype == tm2.DT_INT8 and len(w.scales) == w.dims[0]       # per-out-channel symmetric
        assert np.abs(w.data).max() <= 127 and all(z == 0 for z in w.zero_points)
        assert b.dtype == tm2.DT_INT32
        x = g.tensors[n.inputs[0]]
        wf = gf.tensors[n.inputs[1]].data.reshape(w.dims[0], -1)
        np.testing.assert_allclose(np.float32(w.scales), np.abs(wf).max(1) / np.float32(127), rtol=1e-6)
        bf = gf.tensors[n.inputs[2]].data
        np.testing.assert_array_equal(
            b.data, np.round(bf / (np.float32(x.scales[0]) * np.float32(w.scales))).astype(np.int32))
