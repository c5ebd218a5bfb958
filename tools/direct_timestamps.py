#!/usr/bin/env python3
"""The directly dispatched pass -- the path bench.py's timed loop runs -- under the HSA runtime's own dispatch profiling
(tamd_graph_direct_timestamps: hsa_amd_profiling_get_dispatch_time per packet; csrc/direct.cc).  rocprofv3 cannot see this path (its
queue interceptor does not survive packets it did not see HIP write, so a traced process falls back to hipGraph replay); these are the
same start / end stamps a kernel trace reports, read without the tool.  Prints, per packet: kernel, mean duration, mean gap to the next
packet's start; and the sums against the step's own host clock.

usage: direct_timestamps.py <model> <batch> <int8|uint8> [passes]        (runs ON THE GPU BOX; TAMD_PLAN_CACHE honoured)"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tengine_amd import capi, models, plans, tm2  # noqa: E402


def short(sym):
    m = re.match(r"(?:void )?(?:tamd::)?([A-Za-z0-9_]+)(<[^(]*>)?", sym)
    return (m.group(1) + (m.group(2) or "")) if m else sym[:60]


def main():
    name, batch, dtype = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    passes = int(sys.argv[4]) if len(sys.argv) > 4 else 200
    if "TAMD_PLAN_CACHE" not in os.environ:
        import tempfile
        p = os.path.join(tempfile.gettempdir(), "tamd_plan_ts_%d.txt" % os.getpid())
        plans.seed(p, name, dtype, batch)
        os.environ["TAMD_PLAN_CACHE"] = p
    g = models.build(name, dtype, batch)
    gr = capi.Graph(tm2.write_tm2(g), batch=batch, direct_dispatch=True)
    gr.set_input(models.synth_input(g, 1000, tm2.DT_UINT8 if dtype == "uint8" else tm2.DT_INT8))
    gr.upload()
    gr.sync()
    gr.time_launches(50)
    step_us = min(1e3 * gr.time_launches(passes) / passes for _ in range(3))
    rows = gr.direct_timestamps(passes)
    step_us2 = min(1e3 * gr.time_launches(passes) / passes for _ in range(3))
    halves = gr.halves()
    gr.close()
    if halves:
        print("(the graph is two device graphs of batch %d side by side on their own queues, tamd_options.split_batch: the first half of the rows is the first "
              "half's launch list, stamped alone, then the second's -- the host clock is the two lists OVERLAPPED, so the sums below exceed it)" % (batch // 2))
    print("%s %s batch %d: %d packets per pass, %d passes back to back; host clock %.2f us per step before, %.2f after the stamped passes" % (name, dtype, batch, len(rows), passes, step_us, step_us2))
    print("%-4s %-58s %10s %14s" % ("#", "kernel", "us", "gap to next us"))
    for i, (sym, d, gp) in enumerate(rows):
        print("%-4d %-58s %10.2f %14.2f" % (i, short(sym)[:58], d, gp))
    sd, sg = sum(r[1] for r in rows), sum(r[2] for r in rows)
    print("sum of durations %.2f us + sum of gaps %.2f us = %.2f us per pass (device stamps; every packet carries a completion signal here, the timed loop's carry none)" % (sd, sg, sd + sg))
    print("mean duration %.2f us, mean gap %.2f us; host clock of the unstamped pass: %.2f us" % (sd / len(rows), sg / len(rows), step_us))


if __name__ == "__main__":
    main()
