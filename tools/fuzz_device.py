"""Randomised device-vs-oracle campaign (runs ON THE GPU BOX; the oracle is the checker, nothing under test uses it).

Same random single-op graphs as tools/fuzz_oracle.py, run through the C ABI on the GPU with a randomly pinned member
of the kernel family (TAMD_FORCE_GEMM and the TAMD_PIN keys u8_cfg / u8_patch / u8_patch_cfg / u8_c3 / u8_dw_th / first_rows / dw_form) and compared byte for byte with
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
              "gemm_direct", "pw_stream", "conv_igemm2", "conv_pgemm_i8<128x64,3x3,w4t>", "conv_pgemm_i8<64x64,3x3,w4t>", "conv_pgemm_i8<128x64,3x3,w4b3>",
              "conv_pgemm_i8<64x64,3x3,w4b3>", "conv_pgemm_i8<128x128,3x3,w8>", "conv_pgemm_i8<128x128,3x3,w8b3>", "conv_pgemm_i8<128x64", "conv_pgemm_i8<64x64"]


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
        os.environ.pop("TAMD_FORCE_GEMM", None)
        pins = {}                                  # TAMD_PIN keys (tengine_amd/csrc/env.h): each names a live plan-time candidate
        if a.dtype == "int8":
            m = I8_MEMBERS[int(rng.integers(len(I8_MEMBERS)))]
            if m:
                os.environ["TAMD_FORCE_GEMM"] = m
            if rng.random() < 0.3:
                pins["first_rows"] = "0"
            if rng.random() < 0.7:                 # the depthwise launch forms (fragments per row x output rows per lane)
                pins["dw_form"] = str(rng.choice(["11", "12", "14", "21", "22"]))
        else:
            if rng.random() < 0.8:
                pins["u8_cfg"] = str(int(rng.integers(8)))
            if patch:                              # the patch convolution wherever it applies, a random tile configuration first
                pins["u8_patch"] = "1"
                pins["u8_patch_cfg"] = str(int(rng.integers(5)))      # 4 = lane-level chains
            # the wave-level shallow 3x3 kernel, the depthwise block heights
            if rng.random() < 0.5:
                pins["u8_c3"] = str(int(rng.integers(2)))
            if rng.random() < 0.7:
                pins["u8_dw_th"] = str(rng.choice(["1", "2", "4"]))
        if pins:
            os.environ["TAMD_PIN"] = ",".join("%s=%s" % kv for kv in pins.items())
        else:
            os.environ.pop("TAMD_PIN", None)
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
