#!/usr/bin/env python3
"""HBM traffic per launch and per kernel from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, calibrated on the
streaming copy of tools/exp/hbm_calib.hip measured in the same session (MI355X_MICROARCH.md, HBM section: on gfx950
FETCH_SIZE under-reports wide coalesced reads by 2x; WRITE_SIZE is uncalibrated -> both are scaled by
known bytes / reported value of calib_copy_k).
usage: traffic_summary.py out.json calib_fetch_dir calib_write_dir model_fetch_dir model_write_dir"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def per_kernel(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = defaultdict(float)
        names = {}
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            per[r["Dispatch_Id"]] += float(r["Counter_Value"])
            names[r["Dispatch_Id"]] = r["Kernel_Name"]
        for disp, v in per.items():
            acc[names[disp]].append(v)
    return acc


def main():
    out, cf, cw, mf, mw = sys.argv[1:6]
    known = float(1024 << 20)
    cal = {}
    for counter, d in (("FETCH_SIZE", cf), ("WRITE_SIZE", cw)):
        v = [x for k, vals in per_kernel(d, counter).items() if "calib_copy_k" in k for x in vals]
        cal[counter] = {"reported_per_launch": sum(v) / len(v), "known_bytes": known, "bytes_per_unit": known / (sum(v) / len(v))}
    res = {"calibration": cal, "kernels": {}}
    fetch, write = per_kernel(mf, "FETCH_SIZE"), per_kernel(mw, "WRITE_SIZE")
    for k in sorted(set(fetch) | set(write)):
        fv, wv = fetch.get(k, []), write.get(k, [])
        res["kernels"][k] = {
            "launches": max(len(fv), len(wv)),
            "hbm_read_bytes_per_launch": (sum(fv) / len(fv)) * cal["FETCH_SIZE"]["bytes_per_unit"] if fv else None,
            "hbm_write_bytes_per_launch": (sum(wv) / len(wv)) * cal["WRITE_SIZE"]["bytes_per_unit"] if wv else None,
        }
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps(res["calibration"], indent=1))
    for k, v in res["kernels"].items():
        print("%-90s %5d  read %12.0f  write %12.0f" % (k[:90], v["launches"], v["hbm_read_bytes_per_launch"] or 0, v["hbm_write_bytes_per_launch"] or 0))


if __name__ == "__main__":
    main()
