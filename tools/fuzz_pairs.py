"""Randomised device-vs-oracle campaign for the FUSED PAIRS (runs ON THE GPU BOX; the oracle is the checker, nothing under test uses it):
tools/fuzz_device.py draws single-operator graphs, so the two-node launches -- depthwise 3x3 -> pointwise (dwpw.hip) and pointwise ->
depthwise 3x3 / global pooling (pwdw.hip) -- only met the fixed shapes of tests/test_gpu_dwpw.py and tests/test_gpu_pwdw.py.  Here: random
batch, map, channel counts (ragged stages, 1 .. 8 wave slices, up to 2048 depthwise channels), paddings, activations, with and without
bias, the fusion FORCED (TAMD_FUSE_DWPW=2 / TAMD_FUSE_PWDW=2) and, for pwdw, a random tile configuration pinned through TAMD_PIN; every
graph is also run as two launches on the device.  Byte for byte against oracle/tg_oracle.c.

    python tools/fuzz_pairs.py --seconds 100 --seed 1"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import dwpw_graph, pwdw_graph  # noqa: E402
from oracle import oracle  # noqa: E402
from tengine_amd import capi, tm2  # noqa: E402


def run(g, x, env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        gr = capi.Graph(tm2.write_tm2(g))
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    gr.set_input(x)
    out = [o.copy() for o in gr.run()]
    names = [k["kernel"] for k in gr.profile(1)]
    gr.close()
    return out, names


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=100.0)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    os.environ["TAMD_AUTOTUNE"] = "0"
    t0, graphs, tot, bad, kernels, fused = time.time(), 0, 0, 0, {}, 0
    while time.time() - t0 < a.seconds:
        seed = int(rng.integers(1 << 30))
        if rng.random() < 0.5:
            # depthwise -> pointwise: maps up to 16 wide, cout in whole 64-channel slices up to 512
            c = int(rng.choice([4, 20, 32, 64, 100, 128, 192, 256, 384, 512, 640, 1024, 2048]))
            h, w = int(rng.integers(2, 17)), int(rng.integers(2, 17))
            n = int(rng.integers(1, max(2, min(9, 200000 // (c * h * w) + 1))))
            cout = 64 * int(rng.integers(1, 9))
            p = int(rng.choice([1, 1, 1, 0])) if min(h, w) >= 3 else 1
            g, x = dwpw_graph(seed, n, c, h, w, cout, p, int(rng.choice([0, 0, 6, -1])), int(rng.choice([0, 0, 6, -1])), bool(rng.random() < 0.8))
            env, key = {"TAMD_FUSE_DWPW": "2"}, "dwpw"
            env0 = {"TAMD_FUSE_DWPW": "0"}
        else:
            # pointwise -> depthwise 3x3 (stride 1 / 2) or global pooling, a random tile configuration
            cin = int(rng.choice([8, 16, 24, 32, 64, 96, 128, 256, 512]))
            c = 16 * int(rng.integers(1, 17))
            h, w = int(rng.integers(3, 30)), int(rng.integers(3, 30))
            n = int(rng.integers(1, 5))
            tail = "dw" if rng.random() < 0.8 else "pool"
            s = int(rng.choice([1, 1, 2]))
            g, x = pwdw_graph(seed, n, cin, h, w, c, s=s, p=1, act_pw=int(rng.choice([0, 6, -1])), act_dw=int(rng.choice([0, 6, -1])), tail=tail,
                              pool_alg=int(rng.choice([0, 1])), bias=bool(rng.random() < 0.8))
            env, key = {"TAMD_FUSE_PWDW": "2"}, "pwdw"
            if tail == "dw" and rng.random() < 0.8:
                sl = int(rng.choice([1, 1, 2, 4]))
                env["TAMD_PIN"] = "pwdw_cfg=%dx%dx%d%s" % (int(rng.choice([1, 2, 4, 7, 8, 14])), int(rng.choice([4, 7, 8, 14, 16, 28])), int(rng.choice([256, 512])),
                                                          "" if sl == 1 else "x%d" % sl)
            env0 = {"TAMD_FUSE_PWDW": "0"}
        x[:] = rng.integers(-127, 128, size=x.shape)
        want = oracle.run_graph(g, x)
        try:
            got, names = run(g, x, env)
            two, names2 = run(g, x, env0)
        except Exception as e:
            print("prerun refused:", str(e)[:160], flush=True)
            continue
        graphs += 1
        fused += len(names) == 1
        for k in names:
            kernels[k] = kernels.get(k, 0) + 1
        for wv, o, o2 in zip(want, got, two):
            d = int(np.count_nonzero(wv != o.reshape(wv.shape))) + int(np.count_nonzero(wv != o2.reshape(wv.shape)))
            tot += 2 * wv.size
            bad += d
            if d:
                print("MISMATCH", key, [t.dims for t in g.tensors[:3]], env, names, names2, d, "of", 2 * wv.size, flush=True)
    print("pairs: %d graphs (%d as ONE launch), %d output bytes (fused + two launches), %d mismatches (%d s, seed %d)" % (graphs, fused, tot, bad, a.seconds, a.seed))
    print("kernels exercised:", dict(sorted(kernels.items())))


if __name__ == "__main__":
    main()
