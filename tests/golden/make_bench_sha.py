"""sha256 of every graph output of the five BASELINE configs AS bench.py RUNS THEM (same builder, same batch per GPU, input seed
1000 = rank 0's), computed by the REAL reference CPU backend (oracle/_ref, built from the unmodified /root/reference sources by
oracle/build_ref.py).  bench.py prints the device's sha256 for each config next to `golden_match`; tests/test_bench_host_logic.py
checks the file's shape.  Run here (where /root/reference exists):  python tests/golden/make_bench_sha.py [threads]
-> tests/golden/bench_outputs_sha256.json"""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import build_ref, ref_capi  # noqa: E402
from tengine_amd import models, tm2     # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
# (model, dtype, batch per GPU): BASELINE configs[1], its batched side line, configs[2], configs[3] / 8 GPUs, configs[4] / 8 GPUs
CONFIGS = [("mobilenet_v1", "int8", 1), ("mobilenet_v1", "int8", 64), ("resnet50", "int8", 32), ("yolov3_tiny", "uint8", 8), ("mssd", "uint8", 16)]
SEED = 1000


def main():
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else (os.cpu_count() or 1)
    build_ref.build()
    path = os.path.join(HERE, "bench_outputs_sha256.json")
    out = json.load(open(path)) if os.path.exists(path) else {}
    for name, dtype, batch in CONFIGS:
        key = "%s_%s_b%d" % (name, dtype, batch)
        if key in out and "--force" not in sys.argv:
            continue
        u8 = dtype == "uint8"
        g = models.build(name, dtype, batch)
        x = models.synth_input(g, SEED, tm2.DT_UINT8 if u8 else tm2.DT_INT8)
        t0 = time.time()
        outs = ref_capi.run_model(tm2.write_tm2(g), x, ref_capi.MODE_UINT8 if u8 else ref_capi.MODE_INT8, threads)
        out[key] = {"seed": SEED, "input_sha256": hashlib.sha256(np.ascontiguousarray(x).tobytes()).hexdigest(),
                    "outputs": [{"shape": list(o.shape), "dtype": str(o.dtype), "sha256": hashlib.sha256(np.ascontiguousarray(o).tobytes()).hexdigest()} for o in outs]}
        print(key, "%.1f s" % (time.time() - t0), out[key]["outputs"], flush=True)
        json.dump(out, open(path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
