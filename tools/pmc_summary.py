#!/usr/bin/env python3
"""Merge rocprofv3 --pmc CSV passes (counter_collection.csv) into a per-kernel table: mean counter value
per dispatch, grouped by kernel name.  usage: pmc_summary.py out.csv pass1_dir [pass2_dir ...]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    acc = defaultdict(lambda: defaultdict(list))
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            per_dispatch = defaultdict(float)
            names = {}
            for r in csv.DictReader(open(f)):
                key = (r["Dispatch_Id"], r["Counter_Name"])
                per_dispatch[key] += float(r["Counter_Value"])
                names[r["Dispatch_Id"]] = r["Kernel_Name"]
            for (disp, cname), v in per_dispatch.items():
                acc[names[disp]][cname].append(v)
    counters = sorted({c for k in acc.values() for c in k})
    with open(out, "w") as fh:
        fh.write("kernel,dispatches," + ",".join(counters) + "\n")
        for k in sorted(acc):
            n = max(len(v) for v in acc[k].values())
            fh.write('"%s",%d,' % (k, n) + ",".join("%.1f" % (sum(acc[k][c]) / len(acc[k][c])) if acc[k][c] else "" for c in counters) + "\n")
    print(open(out).read())


if __name__ == "__main__":
    main()
