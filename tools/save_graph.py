"""Write the quantised tmfile of a synthetic BASELINE model: the host-side restatement of
tools/quantize/quant_save_graph.cpp (tengine_amd/models.py: quantize_int8 :355-613, quantize_uint8 :82-353) plus
the tm2 writer (tengine_amd/tm2.py, tm2_format.h).  The file is what `create_graph(ctx, "tengine", path)` loads --
the reference's own examples / tm_benchmark run it unmodified, on the CPU device or on device "HIP".

    python tools/save_graph.py --model mobilenet_v1 --dtype int8 -o /tmp/mobilenet_int8.tmfile [--check]

--check (only where oracle/_ref is built): loads the file with the REAL reference, runs it on its CPU device with a
seeded input and prints an output checksum next to the oracle's (they must agree)."""
import argparse
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tengine_amd import models, tm2  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="mobilenet_v1", choices=sorted(models.BUILDERS))
    ap.add_argument("--dtype", default="int8", choices=["fp32", "int8", "uint8"])
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--device-only", action="store_true", help="cut the tail ops the HIP device leaves to the CPU (Softmax)")
    ap.add_argument("-o", "--output", required=True)
    ap.add_argument("--check", action="store_true")
    a = ap.parse_args()
    g = models.build(a.model, a.dtype, a.batch, device_only=a.device_only)
    b = tm2.write_tm2(g)
    with open(a.output, "wb") as f:
        f.write(b)
    convs = sum(1 for n in g.nodes if n.op == "Convolution")
    print("%s: %d bytes, %d nodes (%d convolutions), input %s" % (
        a.output, len(b), len(g.nodes), convs, g.tensors[g.nodes[g.input_nodes[0]].outputs[0]].dims))
    if a.check:
        from oracle import oracle, ref_capi
        if not ref_capi.available():
            raise SystemExit("--check needs the reference library (python oracle/build_ref.py)")
        dt = {"fp32": tm2.DT_FP32, "int8": tm2.DT_INT8, "uint8": tm2.DT_UINT8}[a.dtype]
        mode = {"fp32": ref_capi.MODE_FP32, "int8": ref_capi.MODE_INT8, "uint8": ref_capi.MODE_UINT8}[a.dtype]
        x = models.synth_input(g, 5, dt)
        ref = ref_capi.run_model(b, x, mode, os.cpu_count())
        ora = oracle.run_graph(g, x)
        for i, (r, o) in enumerate(zip(ref, ora)):
            cr, co = zlib.crc32(np.ascontiguousarray(r).tobytes()), zlib.crc32(np.ascontiguousarray(o.reshape(r.shape)).tobytes())
            same = np.array_equal(r, o.reshape(r.shape)) if a.dtype != "fp32" else np.allclose(r, o.reshape(r.shape), 1e-4, 1e-4)
            print("output %d %s: reference crc32 %08x, oracle crc32 %08x -> %s" % (i, list(r.shape), cr, co, "equal" if same else "DIFFERENT"))


if __name__ == "__main__":
    main()
