"""A batched graph as TWO half-batch device graphs behind one tamd_graph (tamd_options.split_batch, csrc/graph_pair.hip).

The reference runs a batch as a loop over images inside every operator (conv_kernel_x86.c:2241-2263, pooling / eltwise / fc the same),
so the two halves' bytes ARE the one graph's bytes: every entry point of the C ABI must behave as for one graph -- same descriptions,
same outputs bit for bit (against the one-launch-list form AND the oracle), host buffers used as two contiguous halves."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import oracle
from tengine_amd import capi, models, tm2
import helpers
from test_gpu_direct import hip_runtime_of_the_library

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DT = {"int8": tm2.DT_INT8, "uint8": tm2.DT_UINT8}


def _both_forms(g, direct, **kw):
    b = tm2.write_tm2(g)
    one = capi.Graph(b, direct_dispatch=direct, split_batch=1, **kw)
    two = capi.Graph(b, direct_dispatch=direct, split_batch=2, **kw)
    assert one.halves() == 0 and two.halves() == 2
    return one, two


@pytest.mark.parametrize("direct", [False, True])
# (mobilenet_v1 batch 2: halves of ONE image -- the reference's depthwise formula depends on batch == 1, conv_dw_hcl_x86.c:508-543, and the
#  halves must follow the whole batch's: tamd_graph.formula_batch, found by tools/fuzz_split.py)
@pytest.mark.parametrize("name,dtype,batch", [("mobilenet_v1", "int8", 2), ("mobilenet_v1", "int8", 4), ("resnet50", "int8", 2), ("mobilenet_v1", "int8", 16)])
def test_two_halves_give_the_one_graphs_bytes_through_every_run_path(name, dtype, batch, direct):
    g = models.build(name, dtype, batch, device_only=(name != "mobilenet_v1"))      # logits-only (a softmaxed output is mostly zeros)
    x = models.synth_input(g, 31, DT[dtype])
    want = oracle.run_graph(g, x) if (name == "mobilenet_v1" and batch <= 4) else None      # (the others: against the one-list form, itself pinned elsewhere)
    one, two = _both_forms(g, direct)
    assert two.input_desc() == one.input_desc() == (list(x.shape), tm2.DT_INT8)
    for gr in (one, two):
        gr.set_input(x)
    ref = one.run()
    got = two.run()                                            # blocking host-to-host run
    for a, c in zip(ref, got):
        assert a.shape == c.shape and a.shape[0] == batch and np.array_equal(a, c)
    if want is not None:
        for w, c in zip(want, got):
            assert np.array_equal(c.reshape(w.shape), w)
    # the images of the two halves really differ (a pair that ran one half twice would show here)
    assert not np.array_equal(got[0][:batch // 2], got[0][batch // 2:])
    # resident path: upload, launches, sync, download
    x2 = models.synth_input(g, 32, DT[dtype])
    one.set_input(x2); two.set_input(x2)
    ref2 = one.run()
    two.upload()
    for _ in range(3):
        two.launch()
    two.sync()
    for a, c in zip(ref2, two.download()):
        assert np.array_equal(a, c)
    # the device copy of an output: one contiguous buffer of the whole batch
    p, n = two.output_device(0)
    back = np.zeros(ref2[0].size, ref2[0].dtype)
    assert n == back.nbytes and hip_runtime_of_the_library().hipMemcpy(back.ctypes.data, p, n, 2) == 0
    assert np.array_equal(back.reshape(ref2[0].shape), ref2[0])
    # asynchronous runs: two in flight, each delivered to the buffers named at submission
    outs = [two.output_like(), two.output_like()]
    two.set_input(x); two.run_async(outs[0])
    two.set_input(x2); two.run_async(outs[1])
    with pytest.raises(capi.TamdError, match="already in flight"):
        two.run_async()
    with pytest.raises(capi.TamdError, match="in flight"):
        two.run()
    two.wait(); two.wait()
    with pytest.raises(capi.TamdError, match="no run in flight"):
        two.wait()
    for a, c in zip(ref, outs[0]):
        assert np.array_equal(a, c)
    for a, c in zip(ref2, outs[1]):
        assert np.array_equal(a, c)
    # launch lists, packets, timing entry points
    # (the two halves plan on their own: without a plan file their timing races may choose different members of a kernel family --
    #  the same bytes either way -- so only the sums are asserted)
    assert two.kernel_num() >= 2
    prof = two.profile(3)
    assert len(prof) == two.kernel_num() and all(k["ms"] > 0 for k in prof)
    assert abs(sum(k["macs"] for k in prof) - sum(k["macs"] for k in one.profile(1))) < 1e-6 * sum(k["macs"] for k in prof)
    assert two.time_launches(5) > 0
    if direct:
        assert two.direct_packets() >= 2
    else:
        assert two.direct_packets() == 0
    # ... and the passes above left the graph usable
    two.bind_default_outputs()
    two.set_input(x)
    for a, c in zip(ref, two.run()):
        assert np.array_equal(a, c)
    one.close(); two.close()


_STAMPS = r"""
import os, sys
sys.path.insert(0, %r)
from tengine_amd import capi, models, tm2
g = models.build("mobilenet_v1", "int8", 4)
gr = capi.Graph(tm2.write_tm2(g), direct_dispatch=True, split_batch=2)
assert gr.halves() == 2
gr.set_input(models.synth_input(g, 1, tm2.DT_INT8))
gr.upload()
gr.sync()
step_us = min(1e3 * gr.time_launches(100) / 100 for _ in range(3))
for passes in (5, 30):
    rows = gr.direct_timestamps(passes)
    assert len(rows) == gr.direct_packets() and all(d > 0.2 for _, d, _ in rows), rows
alone = sum(d + gp for _, d, gp in rows)
print("STAMPS %%d packets, the two lists alone %%.2f us; host clock of the overlapped step %%.2f us" %% (len(rows), alone, step_us))
assert step_us < alone
gr.close()
"""


def test_direct_timestamps_of_a_pair_list_both_halves():
    """tamd_graph_direct_timestamps of a pair: the first half's packets stamped alone, then the second's (per-packet durations of the path
    the timed loop runs; the overlap of the two lists is what the host's clock sees: the step is shorter than the two lists one after the
    other).  In a process of its own, as tools/direct_timestamps.py and bench.py's per-configuration runs use it: in a process that has
    created and destroyed hundreds of HSA queues (the whole GPU suite in one pytest process, GPU call 39) a pair's stamped passes came back
    WITHOUT stamps on some packets -- the entry point then returns an error that names them, never numbers."""
    e = dict(os.environ)
    e.pop("TAMD_SPLIT_BATCH", None)
    e.pop("TAMD_DIRECT_DISPATCH", None)
    r = subprocess.run([sys.executable, "-c", _STAMPS % ROOT], capture_output=True, text=True, env=e, timeout=600)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("STAMPS")]
    assert r.returncode == 0 and line, (r.stdout[-800:], r.stderr[-2000:])
    print(line[0])


def test_read_tensor_of_a_pair_returns_the_whole_batch():
    g, x = helpers.pwdw_graph(5, 4, 16, 12, 12, 32)
    one, two = _both_forms(g, True, keep_tensors=True)
    for gr in (one, two):
        gr.set_input(x)
        gr.run()
    n = capi.lib().tamd_graph_tensor_num(two._h)
    assert n == capi.lib().tamd_graph_tensor_num(one._h)
    seen = 0
    for i in range(n):
        try:
            a = one.read_tensor(i)
        except capi.TamdError:
            with pytest.raises(capi.TamdError):
                two.read_tensor(i)                             # fused away in the one form: fused away in the halves
            continue
        assert np.array_equal(a, two.read_tensor(i)), i
        seen += 1
    assert seen >= 3
    one.close(); two.close()


def test_uint8_conv_as_a_pair_is_byte_exact():
    g, x = helpers.u8_conv_graph(9, 6, 16, 14, 14, 32, 3, p=1)
    want = oracle.run_graph(g, x)
    one, two = _both_forms(g, True)
    for gr in (one, two):
        gr.set_input(x)
    a, c = one.run(), two.run()
    assert np.array_equal(a[0], c[0]) and np.array_equal(c[0].reshape(want[0].shape), want[0])
    one.close(); two.close()


def test_graphs_that_cannot_be_halved_stay_one_launch_list():
    """an odd batch; operators outside the batch-wise independent list (Flatten / Reshape / Permute / PriorBox heads): split_batch = 2 is
    a wish, not an error"""
    g = models.build("mobilenet_v1", "int8", 3)
    gr = capi.Graph(tm2.write_tm2(g), direct_dispatch=True, split_batch=2)
    assert gr.halves() == 0
    gr.close()
    g, x = helpers.i8_head_graph(3, 2, 16, 5, 5)
    gr = capi.Graph(tm2.write_tm2(g), direct_dispatch=True, split_batch=2)
    assert gr.halves() == 0
    gr.set_input(x)
    for w, o in zip(oracle.run_graph(g, x), gr.run()):
        assert np.array_equal(o.reshape(w.shape), w)
    gr.close()


_RULE = r"""
import os, sys
sys.path.insert(0, %r)
sys.path.insert(0, %r + '/tests')
from tengine_amd import capi, models, tm2
import helpers
import numpy as np
def halves(g, **kw):
    gr = capi.Graph(tm2.write_tm2(g), **kw)
    h = gr.halves()
    gr.close()
    return h
i8 = lambda n: helpers.conv_graph(1, n, 16, 10, 10, 32, 3, p=1)[0]
u8 = lambda n: helpers.u8_conv_graph(1, n, 16, 10, 10, 32, 3, p=1)[0]
out = [halves(i8(16), direct_dispatch=True), halves(i8(8), direct_dispatch=True), halves(i8(16), direct_dispatch=False),
       halves(u8(16), direct_dispatch=True), halves(i8(16), direct_dispatch=True, split_batch=1), halves(i8(2), direct_dispatch=False, split_batch=2)]
print("HALVES", *out)
"""


@pytest.mark.parametrize("env,want", [(None, "2 0 0 0 0 2"), ("0", "0 0 0 0 0 0"), ("2", "2 2 2 2 0 2"), ("1", "2 0 0 0 0 0")])
def test_the_default_rule_and_the_switch(env, want):
    """default: int8 graphs with direct dispatch from batch 16 on; split_batch = 1 is final (a caller that splits by itself); TAMD_SPLIT_BATCH
    overrides the rest (0 never, 1 the default rule, 2 wherever possible) -- read at prerun, in a process of its own"""
    e = dict(os.environ)
    e.pop("TAMD_SPLIT_BATCH", None)
    e.pop("TAMD_DIRECT_DISPATCH", None)
    if env is not None:
        e["TAMD_SPLIT_BATCH"] = env
    r = subprocess.run([sys.executable, "-c", _RULE % (ROOT, ROOT)], capture_output=True, text=True, env=e, timeout=600)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("HALVES")]
    assert line, (r.stdout[-800:], r.stderr[-2000:])
    assert line[0] == "HALVES " + want
