"""Host tools that run without a GPU: the quantised-tmfile writer and its check against the real
reference; the GPU-only tools must refuse to run loudly instead of falling back to anything."""
import os
import subprocess
import sys

import pytest

from oracle import ref_capi
from tengine_amd import models, tm2

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not ref_capi.available(), reason="reference library not built (oracle/build_ref.py)")
def test_save_graph_writes_a_tmfile_the_real_reference_runs(tmp_path):
    out = str(tmp_path / "mobilenet_int8.tmfile")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "save_graph.py"), "--model", "mobilenet_v1", "--dtype", "int8",
                        "-o", out, "--check"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-400:]
    assert "-> equal" in r.stdout and "DIFFERENT" not in r.stdout, r.stdout
    g = tm2.read_tm2(open(out, "rb").read())
    assert sum(1 for n in g.nodes if n.op == "Convolution") == 28


def test_mssd_topology_and_tmfile_round_trip():
    g = models.build("mssd", "uint8", 2)
    assert sum(1 for n in g.nodes if n.op == "Convolution") == 47          # mssd_benchmark.tmfile: 47 convs
    assert sum(1 for n in g.nodes if n.op == "Permute") == 12 and sum(1 for n in g.nodes if n.op == "Flatten") == 12
    outs = [g.tensors[g.nodes[i].outputs[0]].dims for i in g.output_nodes]
    assert outs == [[2, 1917 * 4], [2, 1917 * 21]]
    g2 = tm2.read_tm2(tm2.write_tm2(g))
    assert [n.op for n in g2.nodes] == [n.op for n in g.nodes]
    assert [n.params.get("order") for n in g2.nodes if n.op == "Permute"] == [[0, 2, 3, 1]] * 12


def test_gpu_only_tool_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "tm_benchmark.py"), "-r", "1", "-s", "1"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "no HIP device" in r.stderr or "no CPU fallback" in r.stderr, r.stderr[-400:]


def test_pair_fuzz_graphs_have_an_even_batch_and_the_oracle_runs_them():
    """tools/fuzz_split.py (the two-half-batch form's device campaign): its generator only hands out graphs the library can halve by batch --
    an even batch >= 2 on the graph input -- and the oracle, its checker, evaluates every one of them (both dtypes)"""
    import importlib

    import numpy as np

    from oracle import oracle
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    fs = importlib.import_module("fuzz_split")
    for dtype in ("int8", "uint8"):
        rng = np.random.default_rng(11)
        kinds = set()
        for _ in range(25):
            g, x = fs.even_batch_graph(rng, dtype)
            assert x.shape[0] >= 2 and x.shape[0] % 2 == 0
            outs = oracle.run_graph(g, x)
            assert outs and outs[0].shape[0] == x.shape[0]
            assert len(tm2.write_tm2(g)) > 100
            kinds.add(g.name)
        assert len(kinds) >= 2, kinds
