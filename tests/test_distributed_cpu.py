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
