"""INTEGRATION.md section F, compiled (tengine_amd/harness/rccl_gather.cpp): the multi-GPU path from C -- ncclBroadcast of the tmfile bytes,
tamd_graph_load_tm2 on what arrived, static image shards, passes without a collective, one ncclAllGather per graph output -- run
with world size 1 on the GPU box and checked against the Python binding on the same seeded images (FNV-1a of every output in
global image order).  N > 1 needs N GPUs; the sharding / ordering logic for that is covered with gloo on CPU
(tests/test_distributed_cpu.py)."""
import os
import subprocess

import numpy as np
import pytest

from tengine_amd import capi, models, tm2

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tengine_amd", "lib", "rccl_gather.bin")


def _fnv1a(b, h=1469598103934665603):
    for v in bytes(b):
        h = ((h ^ v) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def _images(total, per_image):
    out = np.empty((total, per_image), np.uint8)
    for i in range(total):
        lcg = (0x5EED0000 + i) & 0xFFFFFFFF
        row = np.empty(per_image, np.uint8)
        for k in range(per_image):
            lcg = (lcg * 1664525 + 1013904223) & 0xFFFFFFFF
            row[k] = lcg >> 24
        out[i] = row
    return out


@pytest.mark.parametrize("model,dtype,total", [("mobilenet_v1", "int8", 2), ("yolov3_tiny", "uint8", 1)])
def test_c_harness_world1_matches_python_binding(tmp_path, model, dtype, total):
    if not os.path.exists(BIN):          # product code: built by __graft_entry__.build(); built here when a box arrives without it
        from tengine_amd import build as tb
        tb.build_harness()
    # (a small YOLO map keeps the pure-Python LCG and the CPU calibration pass short)
    g = models.build(model, dtype, total, device_only=(model != "mobilenet_v1"), **({"res": 64} if model == "yolov3_tiny" else {}))
    b = tm2.write_tm2(g)
    tmf = tmp_path / "m.tmfile"
    tmf.write_bytes(b)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([BIN, "0", "1", str(tmp_path / "id"), str(tmf), str(total), "3"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:] + r.stdout[-500:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("output ")]
    assert "ok world 1" in r.stdout and lines
    shape = g.tensors[g.nodes[g.input_nodes[0]].outputs[0]].dims
    per_image = int(np.prod(shape[1:]))
    x = _images(total, per_image).reshape([total] + list(shape[1:])).view(np.int8 if dtype == "int8" else np.uint8)
    gr = capi.Graph(b, batch=total)
    gr.set_input(x)
    outs = gr.run()
    gr.close()
    assert len(lines) == len(outs)
    for line, o in zip(lines, outs):
        f = line.split()
        assert int(f[3]) == o.nbytes // total
        assert int(f[7], 16) == _fnv1a(np.ascontiguousarray(o).tobytes()), line
