"""Direct dispatch (tamd_options.direct_dispatch, csrc/direct.cc): tamd_graph_launch replays the recorded launch list as AQL
packets on the graph's own HSA queue.  Same kernels, same arguments, same bytes as the hipGraph replay -- checked against the
oracle and against the hipGraph path for an int8, a uint8 and an fp32 model (single-struct and multi-scalar argument lists,
dynamic LDS, 1-D and 3-D grids), mixed with the stream-ordered entry points, and timed."""
import numpy as np
import pytest

from oracle import oracle
from tengine_amd import capi, models, tm2
from tengine_amd.tm2 import DT_FP32, DT_INT8, DT_UINT8

pytestmark = pytest.mark.gpu
NP = {"int8": DT_INT8, "uint8": DT_UINT8, "fp32": DT_FP32}


def _resident(gr, x, launches):
    gr.set_input(x)
    gr.upload()
    for _ in range(launches):
        gr.launch()
    gr.sync()
    return gr.download()


@pytest.mark.parametrize("name,dtype,batch", [("mobilenet_v1", "int8", 1), ("mobilenet_v1", "int8", 8), ("resnet50", "int8", 2),
                                              ("yolov3_tiny", "uint8", 1), ("mssd", "uint8", 2), ("squeezenet_v1.1", "fp32", 1)])
def test_direct_dispatch_same_bytes_as_graph_replay(name, dtype, batch):
    g = models.build(name, dtype, batch, device_only=(name != "mobilenet_v1"))      # logits-only graphs (a softmaxed output is mostly zeros)
    x = models.synth_input(g, 77, NP[dtype])
    b = tm2.write_tm2(g)
    ref = capi.Graph(b)
    want = _resident(ref, x, 1)
    ref.close()
    gr = capi.Graph(b, direct_dispatch=True)
    assert gr.direct_packets() >= gr.kernel_num() > 0, "direct dispatch did not take effect"
    # every packet's hidden arguments sit where the kernel's own code-object metadata says (csrc/codeobj_meta.h), not at assumed offsets
    assert gr.direct_meta_packets() == gr.direct_packets(), (gr.direct_meta_packets(), gr.direct_packets())
    got = _resident(gr, x, 3)
    for w, o in zip(want, got):
        assert np.array_equal(w, o)
    # the stream-ordered entry points still work on the same graph, before and after more direct passes
    gr.set_input(x)
    got2 = gr.run()
    for w, o in zip(want, got2):
        assert np.array_equal(w, o)
    x2 = models.synth_input(g, 78, NP[dtype])
    got3 = _resident(gr, x2, 2)
    gr.close()
    ref = capi.Graph(b)
    want3 = _resident(ref, x2, 1)
    ref.close()
    for w, o in zip(want3, got3):
        assert np.array_equal(w, o)
    assert any(not np.array_equal(a, c) for a, c in zip(want, want3))


def test_direct_dispatch_against_the_oracle_and_the_clock():
    g = models.build("mobilenet_v1", "int8", 1)
    x = models.synth_input(g, 5, DT_INT8)
    want = oracle.run_graph(g, x)
    b = tm2.write_tm2(g)
    gr = capi.Graph(b, direct_dispatch=True)
    got = _resident(gr, x, 2)
    for w, o in zip(want, got):
        assert np.array_equal(o.reshape(w.shape), w)
    gr.time_launches(50)
    t_direct = min(gr.time_launches(300) for _ in range(3)) / 300
    gr.close()
    ref = capi.Graph(b)
    _resident(ref, x, 1)
    ref.time_launches(50)
    t_graph = min(ref.time_launches(300) for _ in range(3)) / 300
    ref.close()
    print("mobilenet_v1 int8 b1: direct %.1f us/step, hipGraph %.1f us/step" % (t_direct * 1e3, t_graph * 1e3))
    assert t_direct < t_graph * 1.02, (t_direct, t_graph)


def test_direct_dispatch_env_switch(monkeypatch):
    g = models.build("mobilenet_v1", "int8", 1)
    b = tm2.write_tm2(g)
    monkeypatch.setenv("TAMD_DIRECT_DISPATCH", "1")
    gr = capi.Graph(b)
    assert gr.direct_packets() > 0
    gr.close()
    monkeypatch.setenv("TAMD_DIRECT_DISPATCH", "0")
    gr = capi.Graph(b, direct_dispatch=True)
    assert gr.direct_packets() == 0
    gr.close()


@pytest.mark.parametrize("batch", [1, 4])
def test_direct_dispatch_coherent_launches_never_serve_stale_tensors(batch):
    """Under direct dispatch the pointwise+depthwise family runs its COHERENT instances (agent-scope loads, write-through
    stores) and their packets carry no fences.  Every pass rewrites the same tensors at the same addresses, so a stale L1 / L2
    line would show as the PREVIOUS input's result: alternate two inputs over many bursts of several passes."""
    g = models.build("mobilenet_v1", "int8", batch)
    b = tm2.write_tm2(g)
    xs = [models.synth_input(g, 300 + i, DT_INT8) for i in range(3)]
    ref = capi.Graph(b)
    wants = [_resident(ref, x, 1) for x in xs]
    ref.close()
    assert not np.array_equal(wants[0][0], wants[1][0])
    gr = capi.Graph(b, direct_dispatch=True)
    assert gr.direct_packets() > 0
    for it in range(90):
        k = (it * 7 + it // 3) % 3
        got = _resident(gr, xs[k], 1 + it % 4)
        assert np.array_equal(got[0], wants[k][0]), "burst %d (input %d)" % (it, k)
    got = _resident(gr, xs[1], 300)          # one long burst
    assert np.array_equal(got[0], wants[1][0])
    gr.close()


def test_direct_dispatch_coherent_kernels_with_fences_kept(monkeypatch):
    monkeypatch.setenv("TAMD_DIRECT_COHERENT", "0")      # the same coherent instances behind agent-scope fences
    g = models.build("mobilenet_v1", "int8", 1)
    x = models.synth_input(g, 9, DT_INT8)
    want = oracle.run_graph(g, x)
    gr = capi.Graph(tm2.write_tm2(g), direct_dispatch=True)
    got = _resident(gr, x, 5)
    gr.close()
    for w, o in zip(want, got):
        assert np.array_equal(o.reshape(w.shape), w)


def test_direct_overlap_of_independent_launches_keeps_the_bytes(monkeypatch):
    """TAMD_DIRECT_OVERLAP=1: launches that touch nothing their predecessors touch (the SSD head convolutions, the concat copies)
    go out without the AQL barrier bit; the results must be the ordered pass's, run after run"""
    import numpy as np
    from tengine_amd import capi, models, tm2
    g = models.build("mssd", "uint8", 2)
    x = models.synth_input(g, 11, tm2.DT_UINT8)
    outs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("TAMD_DIRECT_OVERLAP", mode)
        gr = capi.Graph(tm2.write_tm2(g), direct_dispatch=True)
        gr.set_input(x)
        first = [o.copy() for o in gr.run()]
        for _ in range(20):
            again = gr.run()
            for a, b in zip(first, again):
                assert np.array_equal(a, b)
        outs[mode] = first
        assert gr.direct_packets() > 0
        gr.close()
    for a, b in zip(outs["0"], outs["1"]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("name,dtype,batch", [("mobilenet_v1", "int8", 1), ("mobilenet_v1", "int8", 4), ("yolov3_tiny", "uint8", 1)])
def test_back_to_back_passes_overlap_at_the_seam_and_keep_the_bytes(name, dtype, batch, monkeypatch):
    """a burst of passes on the direct path: the first launch of pass k+1 carries no barrier bit when it touches nothing the last
    launch of pass k touches (TAMD_DIRECT_WRAP=1; off by default, it buys nothing on this stack); hundreds of queued passes, a changed input in between, and the
    results are those of the ordered list"""
    g = models.build(name, dtype, batch, device_only=(name != "mobilenet_v1"))
    b = tm2.write_tm2(g)
    xs = [models.synth_input(g, s, NP[dtype]) for s in (31, 32)]
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("TAMD_DIRECT_WRAP", mode)
        gr = capi.Graph(b, direct_dispatch=True)
        assert gr.direct_packets() > 0
        outs = []
        for x in xs:
            outs.append([o.copy() for o in _resident(gr, x, 300)])            # 300 passes queued behind each other, one wait
        res[mode] = outs
        gr.close()
    for a, c in zip(res["0"], res["1"]):
        for u, v in zip(a, c):
            assert np.array_equal(u, v)
    assert any(not np.array_equal(u, v) for u, v in zip(res["1"][0], res["1"][1]))


@pytest.mark.parametrize("name,dtype,batch", [("mobilenet_v1", "int8", 1), ("yolov3_tiny", "uint8", 1)])
def test_device_copy_of_the_outputs_after_a_zero_copy_run(name, dtype, batch):
    """tamd_graph_run on the direct path stores the outputs straight into the pinned host buffers (no download launch): the
    device-side copy -- tamd_graph_output_device (the RCCL gather reads it), tamd_graph_download_outputs -- must still be THIS
    run's bytes, not the previous pass's (it used to be stale)."""
    import ctypes as C
    g = models.build(name, dtype, batch, device_only=(name != "mobilenet_v1"))
    b = tm2.write_tm2(g)
    gr = capi.Graph(b, direct_dispatch=True)
    x1, x2 = models.synth_input(g, 31, NP[dtype]), models.synth_input(g, 32, NP[dtype])
    first = _resident(gr, x1, 1)                  # the staging buffers now hold x1's outputs
    gr.set_input(x2)
    got = [o.copy() for o in gr.run()]            # host-to-host, zero-copy outputs
    assert any(not np.array_equal(a, c) for a, c in zip(first, got))
    again = gr.download()                         # no pass in between: must be x2's outputs
    for a, c in zip(got, again):
        assert np.array_equal(a, c)
    hip = C.CDLL("libamdhip64.so")
    for i, want in enumerate(got):
        p, n = gr.output_device(i)
        assert n == want.nbytes
        host = np.empty_like(want)
        assert hip.hipMemcpy(C.c_void_p(host.ctypes.data), C.c_void_p(p), C.c_size_t(n), 2) == 0      # hipMemcpyDeviceToHost
        assert np.array_equal(host, want), "output %d: the device copy is not the last run's" % i
    # the asynchronous pair: the newest run decides
    gr.set_input(x1)
    gr.run_async()
    gr.set_input(x2)
    gr.run_async()
    gr.wait()
    gr.wait()
    for a, c in zip(got, gr.download()):
        assert np.array_equal(a, c)
    gr.close()
