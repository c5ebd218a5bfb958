"""Randomised campaign for the int8 head plumbing of round 6 (Permute / Flatten / Reshape / Concat / Softmax on dense tensors):
random shapes, head counts, channel counts (padded and unpadded), tails and softmax axes through tests/helpers.py: i8_head_graph.

    python tools/fuzz_heads.py --seconds 60 --seed 1            device (C ABI, GPU box) against the oracle
    python tools/fuzz_heads.py --seconds 60 --seed 1 --ref      oracle against the REAL reference (CPU, where oracle/_ref exists)"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import i8_head_graph  # noqa: E402
from oracle import oracle          # noqa: E402
from tengine_amd import tm2        # noqa: E402


def random_case(rng):
    tail = str(rng.choice(["concat", "concat", "concat", "permute", "flatcat", "reshape", "softmax4"]))
    n = int(rng.integers(1, 4))
    cin = int(rng.choice([3, 8, 16, 24, 40]))
    h, w = int(rng.integers(1, 9)), int(rng.integers(1, 9))
    heads = int(rng.integers(1, 12)) if tail in ("concat", "flatcat") else 1      # (no single-channel convolutions: the reference's int8 path segfaults on them)
    couts = tuple(int(rng.choice([2, 3, 4, 12, 16, 21, 24, 32, 63, 126])) for _ in range(heads))
    kw = dict(seed=int(rng.integers(1 << 30)), n=n, cin=cin, h=h, w=w, couts=couts, tail=tail, same_q=bool(rng.random() < 0.3))
    if tail == "reshape":
        kw["softmax_axis"] = int(rng.choice([1, 2]))
    if tail == "softmax4":
        kw["softmax_axis"] = int(rng.choice([1, 2, 3]))
    return kw


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--ref", action="store_true")
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    if a.ref:
        from oracle import ref_capi
    else:
        from tengine_amd import capi
    t0, graphs, tot, bad, tails = time.time(), 0, 0, 0, {}
    while time.time() - t0 < a.seconds:
        kw = random_case(rng)
        g, x = i8_head_graph(**kw)
        want = oracle.run_graph(g, x)
        if a.ref:
            got = ref_capi.run_model(tm2.write_tm2(g), x, ref_capi.MODE_INT8, 1)
        else:
            gr = capi.Graph(tm2.write_tm2(g), direct_dispatch=bool(rng.random() < 0.5))
            gr.set_input(x)
            got = gr.run()
            gr.close()
        graphs += 1
        tails[kw["tail"]] = tails.get(kw["tail"], 0) + 1
        for w_, o in zip(want, got):
            tot += int(np.asarray(w_).size)
            if not np.array_equal(np.asarray(w_).ravel(), np.asarray(o).ravel()):
                bad += 1
                print("MISMATCH", kw)
    print("fuzz_heads %s: %d graphs, %d outputs bytes, %d mismatching outputs, tails %s (%.0f s, seed %d)"
          % ("oracle vs the real reference" if a.ref else "device vs oracle", graphs, tot, bad, dict(sorted(tails.items())), time.time() - t0, a.seed))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
