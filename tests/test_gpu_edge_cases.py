"""Edge cases of the hot path through the C ABI: ragged / degenerate shapes, extreme sizes, error behaviour, repeated
and concurrent execution.  Same bars as the parity tests (bytes for int8 / uint8)."""
import threading

import numpy as np
import pytest

from helpers import conv_graph, fc_graph, pool_graph, u8_conv_graph
from oracle import oracle
from tengine_amd import capi, models, tm2

pytestmark = pytest.mark.gpu


def run_hip(g, x):
    gr = capi.Graph(tm2.write_tm2(g))
    gr.set_input(x)
    out = gr.run()
    gr.close()
    return out


RAGGED_I8 = [
    # n, cin, h, w, cout, k, s, p, group, act, bias, dil
    (5, 1, 9, 9, 1, 3, 1, 1, 1, 0, True, 1),          # single channel in and out
    (1, 3, 1, 1, 7, 1, 1, 0, 1, -1, True, 1),         # 1x1 image
    (2, 8, 3, 3, 8, 5, 1, 2, 1, 0, True, 1),          # kernel larger than the image (all taps partially outside)
    (1, 16, 17, 5, 24, 3, 3, 1, 1, 6, False, 1),      # stride 3, H != W
    (7, 33, 6, 6, 65, 1, 1, 0, 1, 0, True, 1),        # channels just over the 16/64 granules, batch 7
    (1, 2048, 2, 2, 2048, 1, 1, 0, 1, 0, True, 1),    # K = cout = 2048 on a 2x2 map
    (1, 64, 1, 1, 4096, 1, 1, 0, 1, -1, True, 1),     # very wide FC-like layer
    (1, 512, 7, 7, 32, 7, 1, 3, 1, 0, True, 1),       # K = 25088 (7x7x512)
    (3, 12, 8, 8, 12, 3, 1, 1, 3, 0, True, 1),        # groups of 4 channels
]


@pytest.mark.parametrize("case", RAGGED_I8, ids=[str(c) for c in RAGGED_I8])
def test_ragged_int8_conv(case):
    n, cin, h, w, cout, k, s, p, group, act, bias, dil = case
    g, x = conv_graph(1000 + cin + cout + h, n, cin, h, w, cout, k, s, p, group, act, bias, dil)
    want = oracle.run_graph(g, x)[0]
    got = run_hip(g, x)[0].reshape(want.shape)
    assert np.array_equal(want, got)


RAGGED_U8 = [
    (5, 1, 9, 9, 1, 3, 1, 1, 1, 0, True, 1),
    (1, 3, 1, 1, 7, 1, 1, 0, 1, -1, True, 1),         # one pixel: only the four-chain "tail" order
    (2, 8, 3, 3, 8, 5, 1, 2, 1, 0, True, 1),
    (1, 16, 17, 5, 24, 3, 3, 1, 1, 6, False, 1),
    (7, 33, 6, 6, 65, 1, 1, 0, 1, 0, True, 1),
    (1, 640, 3, 5, 9, 3, 1, 1, 1, 0, True, 1),        # K = 5760, 15 pixels (8 main + 7 tail), 9 rows (8-block + 1 single)
    (1, 512, 7, 7, 34, 7, 1, 3, 1, 0, True, 1),       # K = 25088: 100 KB tap table in LDS (opt-in above 64 KB)
    (1, 3600, 3, 3, 8, 3, 1, 1, 1, 0, True, 1),       # K = 32400: tap table too big for LDS -> the LDS-DMA kernel
    (3, 12, 8, 8, 12, 3, 1, 1, 3, 0, True, 1),
]


@pytest.mark.parametrize("case", RAGGED_U8, ids=[str(c) for c in RAGGED_U8])
def test_ragged_uint8_conv(case):
    n, cin, h, w, cout, k, s, p, group, act, bias, dil = case
    g, x = u8_conv_graph(2000 + cin + cout + h, n, cin, h, w, cout, k, s, p, group, act, bias, dil)
    want = oracle.run_graph(g, x)[0]
    got = run_hip(g, x)[0].reshape(want.shape)
    assert np.array_equal(want, got)


def test_saturating_uint8_conv():
    """all-255 inputs and weights with zero points 0: accumulators at their largest, outputs clamp to 255."""
    g, x = u8_conv_graph(3, 1, 64, 6, 6, 32, 3, 1, 1, act=-1, in_zp=0, w_zp=0, out_zp=0)
    g.tensors[[i for i, t in enumerate(g.tensors) if t.name == "w"][0]].data[:] = 255
    x[:] = 255
    want = oracle.run_graph(g, x)[0]
    got = run_hip(g, x)[0].reshape(want.shape)
    assert np.array_equal(want, got) and want.max() == 255


def test_wrong_input_size_is_an_error():
    g, x = conv_graph(1, 1, 16, 8, 8, 16, 1)
    gr = capi.Graph(tm2.write_tm2(g))
    with pytest.raises(RuntimeError):
        gr.set_input(x[:, :8])
    gr.close()


def test_unsupported_op_fails_loudly_at_prerun():
    """an int8 Softmax over the batch axis has no device kernel (until round 5 this test used a spatial axis; those run since round 6):
    prerun must fail with a message, never run something else."""
    g, x = conv_graph(5, 1, 32, 6, 6, 10, 1, act=-1)
    y = g.nodes[-1].outputs[0]
    o = g.add_tensor("prob", list(g.tensors[y].dims), tm2.DT_INT8, tm2.TT_VAR, None, [1.0 / 127.0], [0])
    ni = g.add_node("softmax", "Softmax", [y], [o], axis=0)
    g.output_nodes = [ni]
    with pytest.raises(RuntimeError) as e:
        capi.Graph(tm2.write_tm2(g))
    assert "not supported" in str(e.value) or "unsupported" in str(e.value)


def test_rerun_with_new_input_and_interleaved_graphs():
    g, x1 = conv_graph(11, 2, 32, 10, 10, 48, 3, 1, 1)
    x2 = np.random.default_rng(99).integers(-127, 128, size=x1.shape).astype(np.int8)
    w1, w2 = oracle.run_graph(g, x1)[0], oracle.run_graph(g, x2)[0]
    b = tm2.write_tm2(g)
    ga, gb = capi.Graph(b), capi.Graph(b)
    for _ in range(3):
        ga.set_input(x1); gb.set_input(x2)
        ga.upload(); gb.upload(); ga.launch(); gb.launch()
        oa, ob = ga.download()[0], gb.download()[0]
        assert np.array_equal(oa.reshape(w1.shape), w1) and np.array_equal(ob.reshape(w2.shape), w2)
        ga.set_input(x2)
        assert np.array_equal(ga.run()[0].reshape(w2.shape), w2)
    ga.close(); gb.close()


def test_two_host_threads_two_graphs():
    """one graph per host thread (Tengine itself is single-threaded per graph): results stay exact."""
    g = models.build("mobilenet_v1", "int8", 1)
    b = tm2.write_tm2(g)
    xs = [models.synth_input(g, 100 + i) for i in range(2)]
    wants = [oracle.run_graph(g, x)[0] for x in xs]
    errs = []

    def work(i):
        try:
            gr = capi.Graph(b)
            for _ in range(5):
                gr.set_input(xs[i])
                got = gr.run()[0]
                if not np.array_equal(got.reshape(wants[i].shape), wants[i]):
                    errs.append("thread %d mismatch" % i)
            gr.close()
        except Exception as e:      # noqa
            errs.append(repr(e))

    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs


def test_large_batch_pool_and_fc():
    g, x = pool_graph(3, 33, 24, 9, 9, 1, 3, 2, 1, 0, 0)
    want = oracle.run_graph(g, x)[0]
    assert np.array_equal(run_hip(g, x)[0].reshape(want.shape), want)
    g, x = fc_graph(5, 65, (48,), 33)
    want = oracle.run_graph(g, x)[0]
    assert np.array_equal(run_hip(g, x)[0].reshape(want.shape), want)
