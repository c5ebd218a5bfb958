"""BASELINE configs[0] "via tm_benchmark", literally: the reference's own benchmark/tm_benchmark.cc, unmodified, linked
with the unmodified reference objects and our device compiled in-tree (oracle/build_ref.py build_tm_benchmark ==
INTEGRATION.md route B).  `-d HIP` selects the device exactly as `-d CUDA` would select the reference's CUDA backend
(tm_benchmark.cc:191-221)."""
import os
import re
import subprocess

import pytest

from tengine_amd import capi, models, tm2

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "tm_benchmark_hip")


def _run(tmp_path, device, model="mobilenet_v1", dtype="int8", loops=3):
    if not os.path.exists(EXE):
        pytest.skip("oracle/_ref/tm_benchmark_hip not built (needs /root/reference once)")
    g = models.build(model, dtype, 1)
    f = tmp_path / ("%s_%s.tmfile" % (model, dtype))
    f.write_bytes(tm2.write_tm2(g))
    shape = ",".join(str(d) for d in g.tensors[g.nodes[g.input_nodes[0]].outputs[0]].dims)
    code = {"fp32": "0", "int8": "2", "uint8": "3"}[dtype]
    r = subprocess.run([EXE, "-r", str(loops), "-t", "2", "-d", device, "-m", str(f), "-i", shape, "-f", code], capture_output=True,
                       text=True, timeout=300, cwd=str(tmp_path))
    return r


def test_tm_benchmark_binary_runs_on_the_cpu_device(tmp_path):
    # fp32: tm_benchmark hard-codes opt.precision = FP32 (tm_benchmark.cc:224), which the CPU device needs to match the model
    # (SURVEY Appendix D); the HIP device goes by the tensors' own data types
    r = _run(tmp_path, "CPU", "squeezenet_v1.1", "fp32")
    assert r.returncode == 0, r.stderr[-500:]
    assert re.search(r"min =\s+[\d.]+ ms", r.stderr + r.stdout), (r.stdout, r.stderr)


def test_tm_benchmark_hip_without_gpu_fails_loudly(tmp_path):
    if capi.device_count() > 0:
        pytest.skip("GPU present")
    r = _run(tmp_path, "HIP")
    assert "min =" not in (r.stderr + r.stdout)
    assert "failed" in (r.stderr + r.stdout).lower()


@pytest.mark.gpu
@pytest.mark.parametrize("model,dtype", [("mobilenet_v1", "int8"), ("squeezenet_v1.1", "fp32"), ("yolov3_tiny", "uint8")])
def test_unmodified_tm_benchmark_on_device_hip(tmp_path, model, dtype):
    r = _run(tmp_path, "HIP", model, dtype, loops=20)
    out = r.stderr + r.stdout
    assert r.returncode == 0, out[-800:]
    m = re.search(r"min =\s+([\d.]+) ms", out)
    assert m, out[-800:]
    assert float(m.group(1)) < 50.0           # milliseconds per blocking run_graph: the device really ran it
    assert "device:   HIP" in out
