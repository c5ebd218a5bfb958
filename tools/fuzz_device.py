"""Randomised device-vs-oracle campaign (runs ON THE GPU BOX; the oracle is the checker, nothing under test uses it).

Same random single-op graphs as tools/fuzz_oracle.py, run through the C ABI on the GPU with a randomly pinned member
of the kernel family (TAMD_FORCE_GEMM / TAMD_U8_CFG / TAMD_U8_PATCH / TAMD_U8_RGB3X3 / TAMD_FIRST_ROWS) and compared byte for byte with
oracle/tg_oracle.c (itself pinned to the real reference by fuzz_oracle.py).

    python tools/fuzz_device.py --dtype uint8 --seconds 50 --seed 1"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from fuzz_oracle import random_graph  # noqa: E402
from oracle import oracle  # noqa: E402
from tengine_amd import capi, tm2  # noqa: E402

I8_MEMBERS = ["", "", "igemm0", "igemm1", "igemm2", "igemm3", "igemm4", "igemm5", "igemm6", "igemm7", "igemm8", "igemm9",
              "gemm_direct", "pw_stream", "conv_igemm2"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="uint8", choices=["int8", "uint8"])
    ap.add_argument("--seconds", type=float, default=50.0)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    os.environ["TAMD_AUTOTUNE"] = "0"          # the pinned member decides, not the clock
    t0, graphs, tot, bad, kernels = time.time(), 0, 0, 0, {}
    while time.time() - t0 < a.seconds:
        patch = a.dtype == "uint8" and rng.random() < 0.6
        g, x = random_graph(rng, a.dtype, int(rng.choice([4, 4, 32])) if patch else 1, device=True)
        for k in ("TAMD_FORCE_GEMM", "TAMD_U8_CFG", "TAMD_U8_RGB3X3", "TAMD_FIRST_ROWS", "TAMD_U8_PATCH", "TAMD_U8_PATCH_CFG", "TAMD_U8_PATCH_2D",
                  "TAMD_U8_C3", "TAMD_U8_DW_TH", "TAMD_U8_RGB_MFMA", "TAMD_DW_FORM"):
            os.environ.pop(k, None)
        if a.dtype == "int8":
            m = I8_MEMBERS[int(rng.integers(len(I8_MEMBERS)))]
            if m:
                os.environ["TAMD_FORCE_GEMM"] = m
            if rng.random() < 0.3:
                os.environ["TAMD_FIRST_ROWS"] = "0"
            if rng.random() < 0.7:                 # round 4: the depthwise launch forms (fragments per row x output rows per lane)
                os.environ["TAMD_DW_FORM"] = str(rng.choice(["11", "12", "14", "21", "22"]))
        else:
            if rng.random() < 0.8:
                os.environ["TAMD_U8_CFG"] = str(int(rng.integers(8)))
            if patch:                              # the patch convolution wherever it applies, a random tile configuration first
                os.environ["TAMD_U8_PATCH"] = "1"
                os.environ["TAMD_U8_PATCH_CFG"] = str(int(rng.integers(9)))      # round 4: 4 = lane-level chains, 5 .. 8 = 2-D pixel tiles
                os.environ["TAMD_U8_PATCH_2D"] = "1"
            # round 4: the wave-level shallow 3x3 kernel, the depthwise block heights, the first layer's main pixels on the matrix cores
            if rng.random() < 0.5:
                os.environ["TAMD_U8_C3"] = str(int(rng.integers(2)))
            if rng.random() < 0.7:
                os.environ["TAMD_U8_DW_TH"] = str(rng.choice(["1", "2", "4"]))
            if rng.random() < 0.5:
                os.environ["TAMD_U8_RGB_MFMA"] = "1" 
        want = oracle.run_graph(g, x)
        try:
            gr = capi.Graph(tm2.write_tm2(g))
        except Exception as e:          # a pinned shape the planner refuses (e.g. LDS budget): not a parity event
            print("prerun refused:", str(e)[:120], flush=True)
            continue
        gr.set_input(x)
        got = gr.run()
        for q in gr.profile(1):
            kernels[q["kernel"]] = kernels.get(q["kernel"], 0) + 1
        gr.close()
        graphs += 1
        for w, o in zip(want, got):
            d = int(np.count_nonzero(w != o.reshape(w.shape)))
            tot += w.size
            bad += d
            if d:
                print("MISMATCH", g.name, [t.dims for t in g.tensors[:3]], g.nodes[-1].params,
                      {k: v for k, v in os.environ.items() if k.startswith("TAMD_")}, d, "of", w.size, flush=True)
    print("%s: %d graphs, %d outputs, %d mismatches" % (a.dtype, graphs, tot, bad))
    print("kernels exercised:", dict(sorted(kernels.items())))


if __name__ == "__main__":
    main()
