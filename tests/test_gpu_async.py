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
