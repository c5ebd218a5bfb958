#!/usr/bin/env python3
"""The one-FMA requantisation (tengine_amd/csrc/epilogue.h) against the reference chain for EVERY accumulator value of every
(layer, channel) of the synthetic BASELINE int8 models -- CPU only.  For each conv / FC node the three reference formulas are
folded as the planner folds them (graph.hip: fold_requant): A1 x86 hcl (what batch 1 runs), A2 naive ref (what the reference's
depthwise switches to at batch > 1) for the convolutions, A5 for FC; tests/csrc/exhaustive_requant.c then walks the whole
accumulator range of each record.
    python tools/exhaustive_requant.py [mobilenet_v1 resnet50 ...]"""
import os
import subprocess
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tengine_amd import models  # noqa: E402

FMAX = np.float32(3.4028235e38)


def records(name):
    g = models.build(name, "int8", 1, device_only=(name != "mobilenet_v1"))
    out = []
    for nd in g.nodes:
        if nd.op not in ("Convolution", "FullyConnected"):
            continue
        x, w, y = g.tensors[nd.inputs[0]], g.tensors[nd.inputs[1]], g.tensors[nd.outputs[0]]
        in_s, out_s = np.float32(x.scales[0]), np.float32(y.scales[0])
        ws = np.asarray(w.scales, dtype=np.float32)
        if nd.op == "FullyConnected":
            m2 = ((in_s * ws).astype(np.float32) / out_s).astype(np.float32)          # A5: fl(fl(in_s * w_s) / out_s), out_scale 1, no clamp
            out += [(1.0, v, -FMAX, FMAX, 1.0) for v in m2]
            continue
        act = int(nd.params.get("activation", -1))
        lo, hi = (-FMAX, FMAX)
        if act == 0:
            lo = 0.0
        if act > 0:
            lo, hi = 0.0, 6.0
        out += [(in_s, v, lo, hi, out_s) for v in ws]                                  # A1: m1 = in_scale, m2 = w_scale[c]
        lo, hi = (-FMAX, FMAX)
        if act == 1:
            lo, hi = -1.0, 1.0
        elif act >= 0:
            lo = 0.0
            if act == 6:
                hi = 6.0
        out += [(1.0, v, lo, hi, out_s) for v in (in_s * ws).astype(np.float32)]       # A2: m1 = 1, m2 = fl(in_scale * w_scale[c])
    return np.asarray(out, dtype=np.float32)


def main():
    names = sys.argv[1:] or ["mobilenet_v1", "resnet50"]
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "csrc", "exhaustive_requant.c")
    with tempfile.TemporaryDirectory() as td:
        exe = os.path.join(td, "exh")
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", src, "-o", exe, "-lm"])
        rc = 0
        for name in names:
            rec = records(name)
            path = os.path.join(td, name + ".bin")
            rec.tofile(path)
            r = subprocess.run([exe, path], capture_output=True, text=True)
            print("%-14s %s" % (name, r.stdout.strip()))
            rc |= r.returncode
    sys.exit(rc)


if __name__ == "__main__":
    main()
