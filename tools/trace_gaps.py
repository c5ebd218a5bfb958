#!/usr/bin/env python3
"""Body vs gap of every launch of a replayed hipGraph, from rocprofv3 --kernel-trace timestamps.

usage: trace_gaps.py <kernel_trace.csv> <launches_per_replay> <replays>

Takes the LAST launches_per_replay x replays dispatches of the trace (the replays; everything before is prerun:
autotune, warm-up, capture), folds them by position in the replay and prints per position the average body
(End - Start of the dispatch) and the average gap to the previous dispatch's End (for position 0: to the last
launch of the previous replay)."""
import csv
import sys


def main():
    path, per, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    rows = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    rows = rows[-per * reps:]
    if len(rows) < per * reps:
        raise SystemExit("trace has only %d dispatches" % len(rows))
    body = [0.0] * per
    gap = [0.0] * per
    ngap = [0] * per
    name = [""] * per
    for i, (s, e, n) in enumerate(rows):
        p = i % per
        body[p] += e - s
        name[p] = n
        if i > 0:
            gap[p] += s - rows[i - 1][1]
            ngap[p] += 1
    print("%-4s %-64s %10s %10s" % ("pos", "kernel", "body us", "gap us"))
    tb = tg = 0.0
    for p in range(per):
        b = body[p] / reps / 1e3
        g_ = gap[p] / max(ngap[p], 1) / 1e3
        tb += b
        tg += g_
        print("%-4d %-64s %10.2f %10.2f" % (p, name[p].split("(")[0][:64], b, g_))
    wall = (rows[-1][1] - rows[0][0]) / reps / 1e3
    print("per replay: sum of bodies %.1f us, sum of gaps %.1f us, first start -> last end %.1f us" % (tb, tg, wall))


if __name__ == "__main__":
    main()
