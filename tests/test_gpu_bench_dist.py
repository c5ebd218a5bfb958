"""The N-rank code path of bench.py on ONE GPU (world_size 1 through RCCL, `--force-dist`): process-group set-up on the nccl
(= RCCL) backend, tmfile broadcast, the output all_gather -- final (default: no per-step collective) and per step.  The
sharding / ordering logic for N > 1 is covered on CPU with gloo (tests/test_distributed_cpu.py); this catches API-level
breakage of the device-side calls before the driver's multi-GPU tier runs them."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*argv):
    env = dict(os.environ)
    env.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29541", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-1500:]
    lines = r.stdout.strip().splitlines()
    assert lines and lines[-1].startswith("{"), "the JSON line must be the last line of stdout:\n" + r.stdout[-600:]
    return json.loads(lines[-1])


@pytest.mark.parametrize("gather", ["final", "every"])
def test_bench_rccl_path_on_one_gpu(gather):
    line = _bench("--force-dist", "--steps", "60", "--warmup", "10", "--no-cpu-baseline", "--gather", gather)
    assert line["n_gpus"] == 1 and line["value"] > 1000 and line["steps"] == 60
    assert ("per step" in line["config"]["collectives"]) == (gather == "every")
    plain = _bench("--steps", "60", "--warmup", "10", "--no-cpu-baseline")
    assert plain["output_checksum"] == line["output_checksum"]
    if gather == "final":          # no per-step collective: the N-rank loop costs what the N = 1 loop costs
        assert line["ms_per_step"] < 1.5 * plain["ms_per_step"] + 0.02


def test_bench_default_reports_both_gather_modes_and_the_same_path_at_n1():
    """N > 1 default: two timed regions -- `value` = one overlapped all_gather per step, `gather_final` = no per-step
    collective; N = 1 carries the hipGraph-replay figure too, the dispatch path of the per-step region (like for like curves)"""
    line = _bench("--force-dist", "--steps", "60", "--warmup", "10", "--no-cpu-baseline")
    assert "per step" in line["config"]["collectives"] and line["value"] > 1000
    assert line["gather_final"]["value"] > 1000 and line["gather_final"]["ms_per_step"] > 0
    plain = _bench("--steps", "60", "--warmup", "10", "--no-cpu-baseline")
    assert plain["hipgraph_replay"]["value"] > 1000 and plain["output_checksum"] == line["output_checksum"]
    assert plain["host_to_host_images_per_s"] > 1000 and plain["host_to_host"]["runs"] >= 200 and plain["prerun_ms"] > 0


def test_bench_two_stream_yolo_gathers_both_heads():
    line = _bench("--force-dist", "--steps", "12", "--warmup", "2", "--no-cpu-baseline", "--model", "yolov3_tiny", "--dtype", "uint8",
                  "--batch", "2", "--streams", "2", "--gather", "every")
    assert "2 output(s)" in line["config"]["collectives"] and line["value"] > 100
