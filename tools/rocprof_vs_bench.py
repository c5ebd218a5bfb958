#!/usr/bin/env python3
"""Cross-check of a bench line's `roofline` against the rocprofv3 --kernel-trace --stats summary of the same command (VERDICT r5 item 2:
"`frac` for each is reproducible from the CSV"): the dominant family's launches per step, its average launch duration from the CSV
(calls-weighted over the family's template instances, plan-time / self-check launches excluded by taking steps x launches_per_step of
each instance's calls), and the fraction of the roofline recomputed from the CSV figure.

usage: rocprof_vs_bench.py <round tag, e.g. r06>      (reads profiles/<tag>_bench_<config>.json + <tag>_rocprofv3_kernel_stats_bench_<config>.csv)"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib.util  # noqa: E402

spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)

CONFIGS = [("mobilenet_v1", "int8", 1, "b1"), ("mobilenet_v1", "int8", 64, None), ("resnet50", "int8", 32, None), ("yolov3_tiny", "uint8", 8, None), ("mssd", "uint8", 16, None)]
ALIAS = {"conv_u8_mfma": "conv_u8_gemm_k", "conv_u8_patch": "conv_u8_patch_k", "conv_pgemm_i8": "conv_pgemm", "conv_igemm_i8": "conv_igemm", "conv_u8i": "conv_u8i_k"}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
    P = os.path.join(ROOT, "profiles")
    print("| config | family | bound | launches / step | µs / launch, HIP events (bench line) | µs / launch, rocprofv3 CSV | frac (bench line) | frac from the CSV | step µs (bench) |")
    print("|---|---|---|---|---|---|---|---|---|")
    for model, dtype, batch, short in CONFIGS:
        bj = os.path.join(P, "%s_bench_%s.json" % (tag, short)) if short else os.path.join(P, "%s_bench_%s_%s_b%d.json" % (tag, model, dtype, batch))
        cs = os.path.join(P, "%s_rocprofv3_kernel_stats_bench_%s_%s_b%d.csv" % (tag, model, dtype, batch))
        if not (os.path.exists(bj) and os.path.exists(cs)):
            continue
        j = json.loads(open(bj).read().strip().splitlines()[-1])
        r = j["roofline"]
        fam = r["kernel"]
        rows = list(csv.DictReader(open(cs)))
        tot_ns, calls = 0.0, 0
        for row in rows:
            name = row["Name"]
            if "tamd::" not in name or "copy_bytes" in name or "nchw_to_nhwc" in name:
                continue
            m = bench.re.search(r"pwdw_i8(?:_coh)?_kernel<\s*\d+,\s*(\d+),\s*\w+,\s*(\d+)", name)
            if m:
                mode, prod = int(m.group(1)), int(m.group(2))
                f = "firstdw_i8" if prod == 1 else "pwpool_i8" if mode == 0 else "pw_small_i8" if mode == 4 else "pwdw_i8"
                member = f == fam
            else:
                member = ALIAS.get(fam, fam) in name
            if member:
                tot_ns += float(row["TotalDurationNs"])
                calls += int(row["Calls"])
        avg_us = 1e-3 * tot_ns / max(calls, 1)
        unit = r["algorithmic_bytes_per_launch"] / 1e9 / bench.HBM_PEAK_GBS if r["bound"] == "hbm" else \
            2.0 * r["algorithmic_macs_per_launch"] / 1e12 / r["peak"]
        frac_csv = unit / (avg_us * 1e-6)
        print("| %s %s b%d | `%s` | %s | %d | %.2f | %.2f | %.3f | %.3f | %.1f |" % (
            model, dtype, batch, fam, r["bound"], r["launches_per_step"], r["avg_launch_us"], avg_us, r["frac"], frac_csv, 1e3 * j["ms_per_step"]))


if __name__ == "__main__":
    main()
