"""Multi-GPU harness pieces.  The reference has no distributed code at all (its CUDA
backend pins device 0, source/device/cuda/cuda_executor.hpp:41); the hot path shards naturally:
independent images, read-only weights.  One process per GPU (`torch.distributed`, backend "nccl" ==
RCCL over xGMI on ROCm; "gloo" on CPU for the tests):

  * once:      rank 0 reads / synthesises the tmfile -> broadcast of the raw bytes -> every rank loads
               them with the native loader (tamd_graph_load_tm2 == `create_graph(ctx,"tengine:m",buf,size)`)
  * per batch: static contiguous sharding of the batch (no exchange), outputs returned with one
               all_gather (1000 B/image for classification) that the caller overlaps with the next batch.
No all-reduce anywhere, so the per-link ring bound of xGMI is irrelevant here (messages are KB..MB).
"""
import hashlib


def shard_range(total, world, rank):
    """Contiguous B/G images per rank, remainder to the low ranks: returns (start, count)."""
    base, rem = divmod(total, world)
    count = base + (1 if rank < rem else 0)
    start = rank * base + min(rank, rem)
    return start, count


def broadcast_tmfile(tm_bytes, dist, device, src=0):
    """Broadcast model bytes from `src`; returns the bytes on every rank. Works for nccl and gloo."""
    import torch
    rank = dist.get_rank()
    n = torch.tensor([len(tm_bytes) if rank == src else 0], dtype=torch.int64, device=device)
    dist.broadcast(n, src)
    buf = torch.empty(int(n.item()), dtype=torch.uint8, device=device)
    if rank == src:
        buf.copy_(torch.frombuffer(bytearray(tm_bytes), dtype=torch.uint8))
    dist.broadcast(buf, src)
    out = bytes(buf.cpu().numpy().tobytes())
    # integrity: every rank must hold the same model
    digest = torch.tensor(list(hashlib.sha256(out).digest()[:8]), dtype=torch.int64, device=device)
    ref = digest.clone()
    dist.broadcast(ref, src)
    if not torch.equal(ref, digest):
        raise RuntimeError("tmfile broadcast corrupted on rank %d" % rank)
    return out


def all_gather_outputs(local, dist, counts=None):
    """all_gather of per-rank output blocks (uint8/int8 tensors of equal per-image size).
    `counts` = images per rank when the shards are ragged (pads to the max, trims after)."""
    import torch
    world = dist.get_world_size()
    if counts is None:
        out = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out.view(-1), local.contiguous().view(-1))
        return out.reshape((-1,) + tuple(local.shape[1:]))
    mx = max(counts)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)
