#!/usr/bin/env python3
"""Writes tengine_amd/plans/<model>_<dtype>_b<batch>.txt for the BASELINE configurations (runs ON THE GPU BOX): one prerun per
configuration under a fresh TAMD_PLAN_CACHE, i.e. exactly what the plan-time autotune chooses there, plus the step time the
plan gives (direct dispatch, device-resident) so the file can be judged against the evidence tables.

usage: make_plans.py [out_dir]      (default tengine_amd/plans; TAMD_U8_INT plans get the dtype suffix _int)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tengine_amd import capi, models, plans, tm2  # noqa: E402

CONFIGS = [("mobilenet_v1", "int8", 1, False), ("mobilenet_v1", "int8", 64, False), ("resnet50", "int8", 32, False),
           ("yolov3_tiny", "uint8", 8, False), ("mssd", "uint8", 16, False), ("yolov3_tiny", "uint8", 8, True), ("mssd", "uint8", 16, True)]


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else plans.PLAN_DIR
    os.makedirs(out, exist_ok=True)
    for name, dtype, batch, integer in CONFIGS:
        path = os.path.join(out, "%s_%s%s_b%d.txt" % (name, dtype, "_int" if integer else "", batch))
        if os.path.exists(path):
            os.remove(path)
        os.environ["TAMD_PLAN_CACHE"] = path
        if integer:
            os.environ["TAMD_U8_INT"] = "1"
        try:
            g = models.build(name, dtype, batch, device_only=(name != "mobilenet_v1"))
            gr = capi.Graph(tm2.write_tm2(g), batch=batch, direct_dispatch=True)
            gr.set_input(models.synth_input(g, 3, tm2.DT_UINT8 if dtype == "uint8" else tm2.DT_INT8))
            gr.run()
            gr.upload()
            gr.sync()
            gr.time_launches(20)
            us = 1e3 * gr.time_launches(100) / 100
            n = gr.kernel_num()
            gr.close()
        finally:
            os.environ.pop("TAMD_U8_INT", None)
        lines = sum(1 for _ in open(path)) - 1 if os.path.exists(path) else 0
        print("%-14s %-6s b%-3d %s: %d launches, %.1f us/step, %d cached choices -> %s" % (name, dtype, batch, "integer" if integer else "       ", n, us, lines,
                                                                                          os.path.relpath(path, ROOT)))


if __name__ == "__main__":
    main()
