#!/usr/bin/env python3
"""HBM traffic per launch and per kernel from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, calibrated on the
streaming copy of tools/exp/hbm_calib.hip measured in the same session (MI355X_MICROARCH.md, HBM section: on gfx950
FETCH_SIZE under-reports wide coalesced reads by 2x; WRITE_SIZE is uncalibrated -> both are scaled by
known bytes / reported value of calib_copy_k).
usage: traffic_summary.py out.json calib_fetch_dir calib_write_dir model_fetch_dir model_write_dir [tail]
`tail` = number of trailing dispatches of the model passes to keep (launches per run x runs): everything before them is
prerun -- plan-time autotune launches on other shapes, warm-up, layout kernels -- and would pollute the per-kernel means
(VERDICT r1 weak #5)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def per_kernel(d, counter, tail=0):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = defaultdict(float)
        names = {}
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            per[int(r["Dispatch_Id"])] += float(r["Counter_Value"])
            names[int(r["Dispatch_Id"])] = r["Kernel_Name"]
        disps = sorted(per)
        if tail:
            disps = disps[-tail:]
        for disp in disps:
            acc[names[disp]].append(per[disp])
    return acc


def main():
    out, cf, cw, mf, mw = sys.argv[1:6]
    tail = int(sys.argv[6]) if len(sys.argv) > 6 else 0
    known = float(1024 << 20)
    cal = {}
    for counter, d in (("FETCH_SIZE", cf), ("WRITE_SIZE", cw)):
        v = [x for k, vals in per_kernel(d, counter).items() if "calib_copy_k" in k for x in vals]
        cal[counter] = {"reported_per_launch": sum(v) / len(v), "known_bytes": known, "bytes_per_unit": known / (sum(v) / len(v))}
    res = {"calibration": cal, "tail_dispatches_kept": tail, "kernels": {}}
    fetch, write = per_kernel(mf, "FETCH_SIZE", tail), per_kernel(mw, "WRITE_SIZE", tail)
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
