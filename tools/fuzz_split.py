"""Randomised campaign for the two-half-batch form (tamd_options.split_batch = 2, csrc/graph_pair.hip), runs ON THE GPU BOX: random graphs
of an EVEN batch -- the single-op graphs of tools/fuzz_oracle.py plus the fused-pair / stem / eltwise / concat graphs of tests/helpers.py --
compiled as two device graphs of half the batch behind one handle, run through a randomly chosen path (blocking run, asynchronous pair,
upload + launches + download; hipGraph replay or direct dispatch) and compared byte for byte with oracle/tg_oracle.c (itself pinned to the
real reference).  The oracle is the checker; nothing under test uses it.

    python tools/fuzz_split.py --dtype int8 --seconds 90 --seed 1"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import helpers as H  # noqa: E402
from fuzz_oracle import random_graph  # noqa: E402
from oracle import oracle  # noqa: E402
from tengine_amd import capi, tm2  # noqa: E402


def even_batch_graph(rng, dtype):
    u8 = dtype == "uint8"
    seed = int(rng.integers(1 << 30))
    n = int(rng.choice([2, 2, 4, 6, 8]))
    kind = int(rng.integers(0, 10))
    if kind < 4 or (u8 and kind < 7):                         # a single-op graph of the oracle fuzz, batch made even by redrawing
        for _ in range(50):
            g, x = random_graph(rng, dtype, 1, device=True)
            if x.shape[0] >= 2 and x.shape[0] % 2 == 0:
                return g, x
    if u8:
        c = int(rng.integers(3, 48))
        return H.u8_conv_graph(seed, n, c, int(rng.integers(5, 30)), int(rng.integers(5, 30)), int(rng.integers(2, 64)), int(rng.choice([1, 3])), 1, 1)
    if kind == 4:
        return H.pwdw_graph(seed, n, int(rng.choice([16, 32, 48, 64])), int(rng.integers(6, 30)), int(rng.integers(6, 30)), int(rng.choice([16, 32, 64, 128])),
                            int(rng.choice([1, 2])), 1)
    if kind == 5:
        return H.dwpw_graph(seed, n, int(rng.choice([16, 32, 64])), int(rng.integers(6, 24)), int(rng.integers(6, 24)), int(rng.choice([16, 32, 64])))
    if kind == 6:
        return H.stem_graph(seed, n, int(rng.integers(20, 60)), int(rng.integers(20, 60)), int(rng.choice([16, 32, 64])))
    if kind == 7:
        return H.eltwise_relu_graph(seed, n, int(rng.choice([16, 32, 64])), int(rng.integers(4, 24)), int(rng.integers(4, 24)), bool(rng.random() < 0.7))
    if kind == 8:
        return H.i8_concat_graph(seed, n, int(rng.choice([8, 16, 24])), int(rng.integers(3, 16)), int(rng.integers(3, 16)), 1)
    return H.conv_graph(seed, n, int(rng.integers(2, 100)), int(rng.integers(5, 30)), int(rng.integers(5, 30)), int(rng.integers(2, 100)), 3, 1, 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="int8", choices=["int8", "uint8"])
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    os.environ["TAMD_AUTOTUNE"] = "0"
    t0, graphs, pairs, tot, bad, paths = time.time(), 0, 0, 0, 0, {}
    while time.time() - t0 < a.seconds:
        g, x = even_batch_graph(rng, a.dtype)
        want = oracle.run_graph(g, x)
        direct = bool(rng.random() < 0.6)
        try:
            gr = capi.Graph(tm2.write_tm2(g), direct_dispatch=direct, split_batch=2)
        except Exception as e:          # a shape the planner refuses: not a parity event
            print("prerun refused:", str(e)[:120], flush=True)
            continue
        gr.set_input(x)
        path = int(rng.integers(3))
        if path == 0:
            got = gr.run()
        elif path == 1:
            outs = [gr.output_like(), gr.output_like()]
            gr.run_async(outs[0]); gr.run_async(outs[1]); gr.wait(); gr.wait()
            got = outs[1] if all(np.array_equal(p, q) for p, q in zip(outs[0], outs[1])) else [np.zeros_like(o) - 1 for o in outs[0]]
        else:
            gr.upload()
            for _ in range(int(rng.integers(1, 4))):
                gr.launch()
            gr.sync()
            got = gr.download()
        key = ("pair" if gr.halves() else "one list") + (", direct" if direct else ", hipGraph") + [", run", ", run_async x2", ", upload/launch/download"][path]
        paths[key] = paths.get(key, 0) + 1
        pairs += 1 if gr.halves() else 0
        gr.close()
        graphs += 1
        for w, o in zip(want, got):
            d = int(np.count_nonzero(w != o.reshape(w.shape)))
            tot += w.size
            bad += d
            if d:
                print("MISMATCH", g.name, x.shape, key, d, "of", w.size, flush=True)
    print("%s: %d graphs (%d of them as two half-batch device graphs), %d output values, %d mismatches" % (a.dtype, graphs, pairs, tot, bad))
    print("paths:", dict(sorted(paths.items())))


if __name__ == "__main__":
    main()
