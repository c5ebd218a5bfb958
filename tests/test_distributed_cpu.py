"""N>1 path on CPU: world_size-2 `gloo` run of the multi-GPU harness logic -- tmfile
broadcast, static image sharding, output all_gather.  No GPU here, so the per-rank "device" is the CPU
oracle (tests may use it); what is under test is the sharding / collective logic bench.py uses with RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tengine_amd import dist as tdist
from tengine_amd import tm2


def test_shard_range_covers_batch_exactly():
    for total in (1, 7, 8, 64, 128, 130):
        for world in (1, 2, 3, 8):
            spans = [tdist.shard_range(total, world, r) for r in range(world)]
            assert sum(c for _, c in spans) == total
            pos = 0
            for s, c in spans:
                assert s == pos
                pos += c
            counts = [c for _, c in spans]
            assert max(counts) - min(counts) <= 1 and counts == sorted(counts, reverse=True)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from helpers import conv_graph
    from oracle import oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g_full, x_full = conv_graph(21, total, 16, 10, 10, 24, 3, 1, 1)
        tm_bytes = tm2.write_tm2(g_full) if rank == 0 else None
        got = tdist.broadcast_tmfile(tm_bytes, dist, "cpu")
        g = tm2.read_tm2(got)                       # every rank rebuilds the model from the broadcast bytes
        start, count = tdist.shard_range(total, world, rank)
        counts = [tdist.shard_range(total, world, r)[1] for r in range(world)]
        for t in g.tensors:                         # == tamd_graph_set_batch(count)
            if t.ttype != tm2.TT_CONST and t.dims:
                t.dims = [count] + list(t.dims[1:])
        y = oracle.run_graph(g, x_full[start:start + count])[0]
        allout = tdist.all_gather_outputs(torch.from_numpy(y), dist, counts)
        if rank == 0:
            want = oracle.run_graph(g_full, x_full)[0]
            q.put((np.array_equal(allout.numpy(), want), len(got)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [4, 5])
def test_gloo_world2_broadcast_shard_gather(total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    ok, nbytes = q.get(timeout=10)
    assert ok and nbytes > 1000


# ---- bench.py's own N-rank entry: `python bench.py --gpus N` must start N ranks or fail ------------
import json  # noqa: E402
import subprocess  # noqa: E402
import sys  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*argv, env=None):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), capture_output=True, text=True,
                          env=e, timeout=300)


def test_bench_gpus2_dry_run_spawns_two_ranks_and_gathers_every_output():
    """plain `python bench.py --gpus 2 ...`: re-exec under torch.distributed.run, gloo plumbing on CPU -- tmfile
    broadcast, ragged shards of --global-batch 5, ONE gather of both YOLOv3-tiny heads, global image order"""
    r = _bench("--gpus", "2", "--dry-run", "--model", "yolov3_tiny", "--dtype", "uint8", "--global-batch", "5", "--steps", "2")
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["dry_run"] is True and line["value"] is None
    assert line["config"]["shards"] == [3, 2] and line["config"]["outputs"] == 2
    assert line["config"]["gather_bytes_per_image"] == 255 * 13 * 13 + 255 * 26 * 26      # 215 475 B / image


@pytest.mark.parametrize("model,total,per_image", [("yolov3_tiny", 64, 255 * 13 * 13 + 255 * 26 * 26), ("mssd", 128, None)])
def test_bench_gpus8_dry_run_of_the_baseline_multi_gpu_configs(model, total, per_image):
    """BASELINE configs[3] / [4] at the node's size: `python bench.py --gpus 8 --global-batch 64|128` -- eight ranks (gloo on CPU),
    tmfile broadcast + integrity check on every rank, 8 / 16 images per rank, ONE gather of every output in global image order"""
    r = _bench("--gpus", "8", "--dry-run", "--model", model, "--dtype", "uint8", "--global-batch", str(total), "--steps", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 8 and line["dry_run"] is True and line["value"] is None
    assert line["config"]["shards"] == [total // 8] * 8 and line["config"]["global_batch"] == total
    assert line["config"]["outputs"] >= 2
    if per_image:
        assert line["config"]["gather_bytes_per_image"] == per_image
    _check_multi_gpu_keys(line, 8)


def _check_multi_gpu_keys(line, n):
    """what an N > 1 line must carry so that a scaling curve can be read like for like -- the judged `value`
    (per-step gather), `gather_final`, the same-dispatch-path figure without a collective, the name of the N = 1 key the curve is
    read against, the stall counter of the bounded gather -- with a consistent GPU count / parallelism tag"""
    for key in ("value", "gather_final", "no_collective", "scaling_baseline_key", "gather_stalls", "scaling"):
        assert key in line, key
    assert line["scaling_baseline_key"] == "value"          # round 5: the per-step-gather region dispatches directly, like the N = 1 `value`
    assert line["n_gpus"] == n and line["config"]["parallelism"] == "dp%d" % n and sum(line["config"]["shards"]) == line["config"]["global_batch"]


def test_bench_gpus2_dry_run_of_the_headline_config_carries_the_multi_gpu_keys():
    """BASELINE configs[1] (int8 MobileNet-v1, batch 1 per GPU) at N = 2: weak scaling, one image per rank"""
    r = _bench("--gpus", "2", "--dry-run", "--steps", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["scaling"] == "weak" and line["config"]["shards"] == [1, 1]
    _check_multi_gpu_keys(line, 2)


def test_bench_gpus2_without_two_devices_fails_loudly():
    r = _bench("--gpus", "2", "--steps", "1", "--warmup", "0", env={"HIP_VISIBLE_DEVICES": "", "ROCR_VISIBLE_DEVICES": ""})
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")], "no JSON line may be printed for the wrong GPU count"


def test_bench_refuses_world_size_mismatch():
    r = _bench("--gpus", "1", "--dry-run", env={"WORLD_SIZE": "2", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
