#!/usr/bin/env python3
"""bench.py -- images/sec of int8 MobileNet-v1 224x224 on N MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W      (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" = one forward pass of the hot path over one batch (default batch 1 = BASELINE configs[1]) of
synthetic int8 input that is already resident in HBM.  Weights: the seeded synthetic int8 model written
as a real tmfile; at N>1 rank 0 RCCL-broadcasts the tmfile bytes once (north_star), every rank loads them
with the native loader, the images are sharded (each rank owns its own batch, weak scaling) and the
outputs are all-gathered over RCCL every step, double-buffered so the gather of step k overlaps step k+1.

Timing: W untimed warm-up steps, then exactly K steps bracketed by barrier + torch.cuda.synchronize() on
both sides, MAX over ranks.  value = N * batch * K / t.

Extra objects on the JSON line:
  roofline      dominant kernel family, algorithmic bytes / average launch duration measured with HIP
                events on the launch stream in this same process (tamd_graph_profile), vs the HBM peak
  cpu_baseline  the REAL reference CPU backend (oracle/_ref, built from the unmodified sources) timed on
                this host's cores on a bounded sample (rank 0, N=1 only); falls back to the C oracle port
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
MFMA_I8_PEAK_TOPS = 5000.0   # dense int8 MFMA peak (2x bf16 2.5 PF)
MFMA_F32_PEAK_TOPS = 157.3   # fp32 MFMA peak (the uint8 configs are fp32-simulated: u8_kernels.hip)


class _CAI:
    """Expose a raw device pointer to torch through __cuda_array_interface__ (zero copy)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=1, help="images per GPU per step (BASELINE configs[1]: 1)")
    ap.add_argument("--model", default="mobilenet_v1")
    ap.add_argument("--dtype", default="int8", choices=["int8", "uint8"],
                    help="int8 = BASELINE metric; uint8 = the fp32-simulated configs (yolov3_tiny ...), side lines only")
    ap.add_argument("--streams", type=int, default=1, help="concurrent batch-1 graph instances (1 = sequential, tm_benchmark semantics)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="run the multi-rank code path with world_size 1 (test)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()

    import numpy as np
    import torch

    from tengine_amd import capi, models, tm2

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:  # noqa
        raise SystemExit("--gpus %d but WORLD_SIZE %d" % (args.gpus, world))
    use_dist = world > 1 or args.force_dist   # --force-dist exercises the RCCL path on a single GPU (testing)
    dist = None
    torch.cuda.set_device(local_rank)
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    # ---- model: rank 0 synthesises the int8 tmfile, RCCL broadcast of the raw bytes ----------------
    if rank == 0:
        g = models.build(args.model, args.dtype, args.batch, device_only=(args.model != "mobilenet_v1"))
        tm_bytes = tm2.write_tm2(g)
    if use_dist:
        from tengine_amd import dist as tdist
        # RCCL over xGMI, once, outside the timed loop (tengine_amd/dist.py; gloo-tested on CPU)
        tm_bytes = tdist.broadcast_tmfile(tm_bytes if rank == 0 else None, dist, "cuda")
    g = tm2.read_tm2(tm_bytes)

    # S graph instances on S HIP streams: independent batch-1 requests in flight concurrently (serving mode).
    # Default S=1 == tm_benchmark's semantics (one blocking run_graph after the other).
    S = max(1, args.streams)
    grs = [capi.Graph(tm_bytes, batch=args.batch, gpu_index=local_rank) for _ in range(S)]
    gr = grs[0]
    u8 = args.dtype == "uint8"
    x = models.synth_input(g, 1000 + rank, tm2.DT_UINT8 if u8 else tm2.DT_INT8)   # each rank owns its own shard of images
    for q in grs:
        q.set_input(x)
        q.upload()                               # inputs resident in HBM before the timed region
        q.sync()

    exts = [torch.cuda.ExternalStream(q.stream(), device=torch.device("cuda", local_rank)) for q in grs]
    views = []
    for q in grs:
        out_ptr, out_bytes = q.output_device(0)
        views.append(torch.as_tensor(_CAI(out_ptr, out_bytes), device="cuda"))
    slots, gathered, works = None, None, {}
    if use_dist:
        slots = [[torch.empty(out_bytes, dtype=torch.uint8, device="cuda") for _ in range(2)] for _ in range(S)]
        gathered = [[torch.empty(out_bytes * world, dtype=torch.uint8, device="cuda") for _ in range(2)] for _ in range(S)]

    def step(k):
        i = k % S
        grs[i].launch()
        if use_dist:
            s = (k // S) & 1
            with torch.cuda.stream(exts[i]):
                if works.get((i, s)) is not None:
                    works[(i, s)].wait()         # slot free again (gather issued two rounds ago is done)
                slots[i][s].copy_(views[i], non_blocking=True)
                works[(i, s)] = dist.all_gather_into_tensor(gathered[i][s], slots[i][s], async_op=True)

    def drain():
        if use_dist:
            for (i, s), w in works.items():
                if w is not None:
                    with torch.cuda.stream(exts[i]):
                        w.wait()
        for q in grs:
            q.sync()
        torch.cuda.synchronize()

    for k in range(args.warmup):
        step(k)
    drain()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    drain()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([el], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())

    # ---- roofline of the dominant kernel (HIP events on the launch stream, same process) -----------
    roofline = None
    if rank == 0:
        prof = gr.profile(20)
        fam = {}
        for k in prof:
            f = fam.setdefault(k["kernel"], {"ms": 0.0, "bytes": 0.0, "macs": 0.0, "launches": 0})
            f["ms"] += k["ms"]; f["bytes"] += k["bytes"]; f["macs"] += k["macs"]; f["launches"] += 1
        dom = max(fam, key=lambda n: fam[n]["ms"])
        d = fam[dom]
        t_hbm = d["bytes"] / (HBM_PEAK_GBS * 1e9)
        mfma_peak = MFMA_F32_PEAK_TOPS if u8 else MFMA_I8_PEAK_TOPS
        t_mfma = 2.0 * d["macs"] / (mfma_peak * 1e12)
        if t_hbm >= t_mfma:
            ach = d["bytes"] / (d["ms"] * 1e-3) / 1e9
            roofline = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS}
        else:
            ach = 2.0 * d["macs"] / (d["ms"] * 1e-3) / 1e12
            roofline = {"bound": "mfma", "achieved": ach, "peak": mfma_peak, "unit": "TOP/s", "frac": ach / mfma_peak}
        roofline.update({"kernel": dom, "launches_per_step": d["launches"], "avg_launch_us": 1e3 * d["ms"] / d["launches"],
                         "traffic": pmc_traffic(args.model, args.dtype, args.batch, dom),
                         "algorithmic_bytes_per_launch": d["bytes"] / d["launches"],
                         # whole-step matrix-core utilisation (BASELINE metric's second half): all MACs of the step
                         # over the summed kernel time, against the dense MFMA peak of the compute dtype
                         "mfma_util_pct": 100.0 * 2.0 * sum(f["macs"] for f in fam.values())
                         / (max(sum(f["ms"] for f in fam.values()), 1e-12) * 1e-3) / (mfma_peak * 1e12),
                         "kernel_time_share": d["ms"] / max(sum(f["ms"] for f in fam.values()), 1e-12),
                         "sum_kernel_ms_per_step": sum(f["ms"] for f in fam.values())})

    # ---- CPU baseline: the real reference backend on this host's cores (rank 0, N=1 only) ----------
    cpu = None
    if rank == 0 and (world == 1 and not args.force_dist) and not args.no_cpu_baseline:
        cpu = cpu_baseline(tm_bytes, g, x, args.batch, args.cpu_seconds, u8)

    out = gr.download()[0]
    for q in grs:
        q.close()
    if rank == 0:
        value = world * args.batch * args.steps / el
        line = {
            "metric": "images/sec int8 MobileNet-v1 224x224" if (args.model, args.dtype) == ("mobilenet_v1", "int8")
            else "images/sec %s %s" % (args.dtype, args.model), "value": value, "unit": "images/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32 (uint8 simulated in fp32, as the reference)" if u8 else "int8",
            "data": "synthetic",
            "config": {"workload": "%s %s batch=%d per GPU%s, weights = seeded synthetic tmfile, input resident in HBM, "
                                   "hipGraph replay, %d stream(s)" % (args.model, args.dtype, args.batch,
                                                                      " (BASELINE configs[1])" if (args.model, args.dtype, args.batch) == ("mobilenet_v1", "int8", 1) else "", S),
                       "streams": S,
                       "global_batch": world * args.batch, "parallelism": "dp%d" % world,
                       "collectives": "rccl broadcast(tmfile) once + all_gather(outputs) per step" if use_dist else "none"},
            "roofline": roofline, "cpu_baseline": cpu,
            "output_checksum": int(np.asarray(out, dtype=np.int64).sum()),
        }
        if cpu:
            line["speedup_vs_cpu_reference"] = value / cpu["value"] if cpu["value"] else None
        print(json.dumps(line))
    if use_dist:
        dist.destroy_process_group()


def pmc_traffic(model, dtype, batch, family):
    """HBM bytes per launch of the dominant kernel family, from the committed rocprofv3 PMC summary of this workload
    (profiles/r01_traffic_<model>_<dtype>_b<batch>.json: separate --pmc FETCH_SIZE / WRITE_SIZE passes, scaled by the
    same-session streaming-copy calibration -- tools/collect_profiles.sh, tools/traffic_summary.py); None when this
    workload has no PMC pass (counters cannot be collected from inside the timed process)."""
    path = os.path.join(ROOT, "profiles", "r01_traffic_%s_%s_b%d.json" % (model, dtype, batch))
    if not os.path.exists(path):
        return None
    ks = json.load(open(path))["kernels"]
    tot, n = 0.0, 0
    for name, v in ks.items():
        if family in name and v["hbm_read_bytes_per_launch"] is not None:
            tot += (v["hbm_read_bytes_per_launch"] + (v["hbm_write_bytes_per_launch"] or 0.0)) * v["launches"]
            n += v["launches"]
    return tot / n if n else None


def cpu_baseline(tm_bytes, g, x, batch, budget_s, u8=False):
    """Reference `source/device/cpu` path through create_graph/prerun/run_graph on the host cores."""
    import numpy as np
    threads = os.cpu_count() or 1
    try:
        from oracle import ref_capi
        if not ref_capi.available():
            raise FileNotFoundError
        rg = ref_capi.RefGraph(tm_bytes, ref_capi.MODE_UINT8 if u8 else ref_capi.MODE_INT8, threads)
        rg.set_input(x)
        rg.run()                                   # warm-up (weight packing, pool alloc)
        ts, t_end = [], time.perf_counter() + budget_s
        while time.perf_counter() < t_end or len(ts) < 3:
            t0 = time.perf_counter()
            rg.run()
            ts.append(time.perf_counter() - t0)
        rg.close()
        return {"value": batch / min(ts), "unit": "images/s", "cores": threads, "kind": "reference",
                "sample": "%d timed run_graph() calls of the same tmfile/input (batch %d), min %.1f ms, mean %.1f ms, "
                          "reference CPU backend built -O3 -mfma -fopenmp from the unmodified sources"
                          % (len(ts), batch, 1e3 * min(ts), 1e3 * float(np.mean(ts)))}
    except (FileNotFoundError, OSError):
        from oracle import oracle
        oracle.run_graph(g, x)
        ts, t_end = [], time.perf_counter() + budget_s
        while time.perf_counter() < t_end or len(ts) < 3:
            t0 = time.perf_counter()
            oracle.run_graph(g, x)
            ts.append(time.perf_counter() - t0)
        return {"value": batch / min(ts), "unit": "images/s", "cores": threads, "kind": "port",
                "sample": "%d timed oracle passes (batch %d), min %.1f ms" % (len(ts), batch, 1e3 * min(ts))}


if __name__ == "__main__":
    main()
