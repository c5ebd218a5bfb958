#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (`--kernel-trace` output of ROCm 7.2) as the per-kernel
stats table `--stats` would print: calls, total/avg/min/max ns, share.  Usage: rocprof_db_stats.py X.db [out.csv]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                      "max(grid_x), max(workgroup_x), max(vgpr_count), max(accum_vgpr_count), max(lds_size) "
                      "from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    lines = ["name,calls,total_ns,avg_ns,min_ns,max_ns,pct,grid_x,wg_x,vgpr,agpr,lds"]
    for r in rows:
        lines.append('"%s",%d,%d,%.1f,%d,%d,%.2f,%d,%d,%d,%d,%d' % (r[0], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot,
                                                                  r[6], r[7], r[8], r[9], r[10]))
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out)
    sys.stdout.write(out)


if __name__ == "__main__":
    main()
