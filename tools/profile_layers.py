#!/usr/bin/env python3
"""Per-launch timing table of a config model on the device (HIP events around every launch, the
TG_DEBUG_TIME analogue of source/device/cpu/cpu_dump.c:607-697).  usage: profile_layers.py [model] [batch] [iters] [int8|uint8]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tengine_amd import capi, models, tm2  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "mobilenet_v1"
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    dtype = sys.argv[4] if len(sys.argv) > 4 else "int8"
    g = models.build(name, dtype, batch, device_only=True)
    gr = capi.Graph(tm2.write_tm2(g), batch=batch)
    gr.set_input(models.synth_input(g, 3, {"uint8": tm2.DT_UINT8, "fp32": tm2.DT_FP32}.get(dtype, tm2.DT_INT8)))
    gr.run()
    prof = gr.profile(iters)
    tot = sum(k["ms"] for k in prof)
    print("%-28s %-28s %9s %10s %9s %9s %8s" % ("node", "kernel", "us", "MMAC", "KB", "GB/s", "TOP/s"))
    for k in prof:
        us = k["ms"] * 1e3
        print("%-28s %-28s %9.2f %10.2f %9.1f %9.1f %8.2f" % (k["node"][:28], k["kernel"][:28], us, k["macs"] / 1e6,
                                                          k["bytes"] / 1e3, k["bytes"] / (us * 1e-6) / 1e9 if us else 0,
                                                          2 * k["macs"] / (us * 1e-6) / 1e12 if us else 0))
    print("sum of launches: %.1f us ; graph replay: %.1f us/step" % (tot * 1e3, gr.time_launches(200) / 200 * 1e3))
    gr.close()


if __name__ == "__main__":
    main()
