"""Activation buffers shared between tensors of disjoint lifetimes (graph_plan.hip plan_i8(): tamd_options.keep_tensors = 0, the default)
and device memory carved out of a few large arenas (dev_alloc): same bytes as one buffer per tensor, intermediate tensors are
refused by read_tensor unless the graph was pre-run with keep_tensors."""
import numpy as np
import pytest

from oracle import oracle
from tengine_amd import capi, models, tm2

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,batch,kw", [("resnet50", 2, dict(device_only=True)), ("mobilenet_v1", 3, dict()), ("squeezenet_v1.1", 2, dict(device_only=True))])
def test_shared_buffers_give_the_same_bytes(name, batch, kw, monkeypatch):
    """residual blocks (a tensor read again three nodes later), fused pointwise+depthwise pairs (a tail that runs at its
    producer's position), concat views (fire modules: two producers write one shared buffer)"""
    g = models.build(name, "int8", batch, **kw)
    x = models.synth_input(g, 31)
    want = oracle.run_graph(g, x)[0]
    outs = []
    for keep, env in ((False, {}), (True, {}), (False, {"TAMD_PIN": "arena=0"}), (False, {"TAMD_AUTOTUNE": "0"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        for direct in (False, True):
            gr = capi.Graph(tm2.write_tm2(g), keep_tensors=keep, direct_dispatch=direct)
            gr.set_input(x)
            a = gr.run()[0].copy()
            gr.upload()
            for _ in range(3):
                gr.launch()
            gr.sync()
            b = gr.download()[0].copy()
            gr.close()
            assert np.array_equal(a, b)
            outs.append(a)
        for k in env:
            monkeypatch.delenv(k)
    for o in outs:
        assert np.array_equal(o.reshape(want.shape), want)


def test_read_tensor_of_a_shared_buffer_is_refused_and_keep_tensors_reads_it():
    g = models.build("resnet50", "int8", 1, device_only=True)
    x = models.synth_input(g, 8)
    want = oracle.run_graph(g, x, keep_all=True)
    t = [n.outputs[0] for n in g.nodes if n.name == "res3a_branch2a"][0]
    gr = capi.Graph(tm2.write_tm2(g))
    gr.set_input(x)
    gr.run()
    with pytest.raises(capi.TamdError, match="shares its device memory"):
        gr.read_tensor(t)
    gr.close()
    gr = capi.Graph(tm2.write_tm2(g), keep_tensors=True)
    gr.set_input(x)
    gr.run()
    assert np.array_equal(gr.read_tensor(t).reshape(want[t].shape), want[t])
    gr.close()
