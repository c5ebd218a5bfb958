#!/usr/bin/env python3
"""The per-config table of profiles/README.md from the committed evidence of a tag: algorithmic bytes per step (sum of the
per-launch figures of the layer table == what bench.py's roofline uses), HBM bytes per step from the PMC passes (FETCH_SIZE /
WRITE_SIZE, calibrated: tools/traffic_summary.py), their ratio, and the bench line's dominant kernel.
usage: evidence_table.py [tag=r03]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
CFGS = [("mobilenet_v1", "int8", 1, "b1"), ("mobilenet_v1", "int8", 64, "mobilenet_v1_int8_b64"), ("resnet50", "int8", 32, "resnet50_int8_b32"),
        ("yolov3_tiny", "uint8", 8, "yolov3_tiny_uint8_b8"), ("mssd", "uint8", 16, "mssd_uint8_b16")]
print("| config | ms / step | img/s | algorithmic MB / step | counter MB / step (read + write) | ratio | dominant kernel | launches | µs / launch | bound | frac | MFMA busy |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|")
for model, dt, b, bench in CFGS:
    name = "%s_%s_b%d" % (model, dt, b)
    try:
        j = json.loads(open(os.path.join(P, "%s_bench_%s.json" % (tag, bench))).read().strip().splitlines()[-1])
    except OSError:
        continue
    alg = 0.0
    for line in open(os.path.join(P, "%s_layers_%s.txt" % (tag, name))):
        f = line.split()
        if len(f) >= 7 and f[0] not in ("node", "sum"):
            try:
                alg += float(f[-3]) * 1e3          # KB column
            except ValueError:
                pass
    cnt = None
    tp = os.path.join(P, "%s_traffic_%s.json" % (tag, name))
    if os.path.exists(tp):
        ks = json.load(open(tp))["kernels"]
        runs = 5.0
        cnt = sum(((v["hbm_read_bytes_per_launch"] or 0) + (v["hbm_write_bytes_per_launch"] or 0)) * v["launches"] for v in ks.values()) / runs
    r = j["roofline"]
    print("| %s %s b%d | %.4f | %.0f | %.2f | %s | %s | `%s` | %d | %.2f | %s | %.3f | %.1f %% |" % (
        model, dt, b, j["ms_per_step"], j["value"], alg / 1e6, "%.2f" % (cnt / 1e6) if cnt else "–", "%.2f" % (cnt / alg) if cnt and alg else "–",
        r["kernel"], r["launches_per_step"], r["avg_launch_us"], r["bound"], r["frac"], r.get("mfma_util_pct") or 0.0))
