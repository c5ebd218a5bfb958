"""tamd_graph_run_async / tamd_graph_wait (interface.async_run / async_wait, device.h:60-63): two runs in
flight, results delivered in submission order to the buffers named at submission, byte-identical to blocking runs."""
import numpy as np
import pytest

from oracle import oracle
from tengine_amd import capi, models, tm2

pytestmark = pytest.mark.gpu


def test_pipelined_runs_equal_blocking_runs():
    g = models.build("mobilenet_v1", "int8", 1)
    tmb = tm2.write_tm2(g)
    xs = [models.synth_input(g, 100 + i) for i in range(7)]
    gr = capi.Graph(tmb)
    want = []
    for x in xs:
        gr.set_input(x)
        want.append(gr.run()[0])
    assert np.array_equal(want[0].reshape(-1), oracle.run_graph(g, xs[0])[0].reshape(-1))
    assert len({w.tobytes() for w in want}) > 1                  # the inputs really produce different outputs
    outs = [gr.output_like() for _ in xs]
    pending = 0
    for i, x in enumerate(xs):
        if pending == 2:
            gr.wait()
            pending -= 1
        gr.set_input(x)
        gr.run_async(outs[i])
        pending += 1
    while pending:
        gr.wait()
        pending -= 1
    for i in range(len(xs)):
        assert np.array_equal(outs[i][0], want[i]), "run %d" % i
    gr.close()


def test_async_protocol_errors():
    g = models.build("mobilenet_v1", "int8", 1)
    gr = capi.Graph(tm2.write_tm2(g))
    gr.set_input(models.synth_input(g, 1))
    with pytest.raises(capi.TamdError, match="no run in flight"):
        gr.wait()
    gr.run_async()
    gr.run_async()
    with pytest.raises(capi.TamdError, match="already in flight"):
        gr.run_async()
    with pytest.raises(capi.TamdError, match="in flight"):
        gr.run()
    gr.wait()
    gr.wait()
    gr.run()
    gr.close()


@pytest.mark.parametrize("direct", [False, True])
def test_destroy_and_sync_with_uncollected_runs(direct):
    """asynchronous runs on the direct path are closed at submit time; destroying (or syncing) the graph while
    they are still executing must wait for them -- the queue, the kernel arguments and the pinned buffers outlive the packets."""
    g = models.build("mobilenet_v1", "int8", 1)
    tmb = tm2.write_tm2(g)
    x = models.synth_input(g, 11)
    ref = capi.Graph(tmb)
    ref.set_input(x)
    want = ref.run()[0].copy()
    ref.close()
    for _ in range(20):                       # many short-lived graphs: a use-after-free would fault or corrupt sooner or later
        gr = capi.Graph(tmb, direct_dispatch=direct)
        gr.set_input(x)
        gr.run_async()
        gr.run_async()
        gr.close()                            # nothing collected
    gr = capi.Graph(tmb, direct_dispatch=direct)
    gr.set_input(x)
    out = gr.output_like()
    gr.run_async(out)
    gr.sync()                                 # device work of the uncollected run is complete after this ...
    gr.wait()                                 # ... and wait() only delivers it
    assert np.array_equal(out[0], want)
    gr.bind_default_outputs()
    assert np.array_equal(gr.run()[0], want)
    gr.close()


def test_blocking_run_variants_agree(monkeypatch):
    """the host-to-host list with outputs stored straight into the pinned host buffer and the burst closed by its last packet
    (defaults) == the round-3 form (download launch, barrier packet) == the hipGraph form, for a 1x1-map output (MobileNet) and
    for outputs that pass through a layout launch (a conv graph whose output is a map)."""
    from helpers import conv_graph
    cases = [models.build("mobilenet_v1", "int8", 1)]
    xs = [models.synth_input(cases[0], 5)]
    g2, x2 = conv_graph(21, 2, 32, 14, 14, 48, 3, 1, 1)
    cases.append(g2)
    xs.append(x2)
    for g, x in zip(cases, xs):
        tmb = tm2.write_tm2(g)
        outs = []
        for env in ({}, {"TAMD_IO_ZERO_COPY": "0"}, {"TAMD_PIN": "direct_close_on_last=0"}, {"TAMD_IO_ZERO_COPY": "0", "TAMD_PIN": "direct_close_on_last=0"}, None):
            for k, v in (env or {}).items():
                monkeypatch.setenv(k, v)
            gr = capi.Graph(tmb, direct_dispatch=env is not None)
            for k in (env or {}):
                monkeypatch.delenv(k)
            gr.set_input(x)
            a = gr.run()[0].copy()
            b = gr.run()[0].copy()
            o2 = gr.output_like()
            gr.run_async(o2)
            gr.wait()
            gr.close()
            assert np.array_equal(a, b) and np.array_equal(a, o2[0])
            outs.append(a)
        for o in outs[1:]:
            assert np.array_equal(outs[0], o)
        assert np.array_equal(outs[0].reshape(-1), oracle.run_graph(g, x)[0].reshape(-1))


def test_two_threads_on_one_graph_are_refused_not_raced():
    """include/tengine_amd.h: one graph = one thread at a time.  Enforced since round 5 (graph_internal.h: OneThread): while a thread is
    inside a call, a second thread's call on the SAME graph fails with an error -- it does not interleave with the first one's
    launch list.  Two threads hammer run() on one graph: every call either succeeds with the right bytes or fails with that
    error; afterwards the graph is as usable as before.  (Calls from different threads one AFTER the other are fine: the last
    part runs the graph from a third thread.)"""
    import threading
    g = models.build("mobilenet_v1", "int8", 1)
    x = models.synth_input(g, 21)
    gr = capi.Graph(tm2.write_tm2(g), direct_dispatch=True)
    gr.set_input(x)
    want = [o.copy() for o in gr.run()]
    ok, refused, other = [0], [0], []
    lock = threading.Lock()

    def hammer():
        mine = 0
        for _ in range(200000):              # a refused call returns at once: attempts are cheap, successes are what is counted
            if mine >= 60:
                break
            try:
                out = gr.run()
                good = all(np.array_equal(a, b) for a, b in zip(want, out))
                mine += 1
                with lock:
                    ok[0] += 1
                    if not good:
                        other.append("a run that was let through returned other bytes")
            except capi.TamdError as e:
                with lock:
                    if "one thread at a time" in str(e):
                        refused[0] += 1
                    else:
                        other.append(str(e))

    ts = [threading.Thread(target=hammer) for _ in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not other, other[:3]
    assert ok[0] == 120, (ok[0], refused[0])          # both threads got their 60 runs through, between the other's calls
    res = []
    t = threading.Thread(target=lambda: res.append([o.copy() for o in gr.run()]))
    t.start()
    t.join()
    assert all(np.array_equal(a, b) for a, b in zip(want, res[0]))
    gr.close()
