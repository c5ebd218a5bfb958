#!/usr/bin/env python3
"""The plugin's two-half-batch form against its one-graph form, host to host through the UNMODIFIED reference library
(create_graph / prerun_graph / run_graph on device "HIP"): blocking run_graph() calls timed on the host, TAMD_SPLIT_BATCH=0 against the
default, fresh graph each, interleaved; outputs compared.   usage: plugin_split_ab.py model batch runs rounds [2 = force the split below the default threshold]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_capi as ref  # noqa: E402  (the reference library itself is the HOST here: create_graph / run_graph; nothing is checked against it)
from tengine_amd import models, tm2  # noqa: E402

PLUGIN = os.path.join(ROOT, "tengine_amd", "lib", "libtengine_hip_device.so")


class HipOpt(C.Structure):
    _fields_ = [("dev_name", C.c_char_p), ("size", C.c_int), ("gpu_index", C.c_int), ("use_hip_graph", C.c_int), ("profile", C.c_int)]


def main():
    name, batch, runs, rounds = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    force = len(sys.argv) > 5 and sys.argv[5] == "2"
    L = ref.lib()
    assert L.load_tengine_plugin(b"hip", PLUGIN.encode(), b"register_hip_device") == 0
    P = C.CDLL(PLUGIN)
    P.hip_device_split_subgraphs.restype = C.c_int
    g = models.build(name, "int8", batch)
    x = models.synth_input(g, 7)
    b = tm2.write_tm2(g)
    res = {"0": [], "default": []}
    outs = {}
    for r in range(rounds):
        for mode in ("0", "default"):
            if mode == "0":
                os.environ["TAMD_SPLIT_BATCH"] = "0"
            elif force:
                os.environ["TAMD_SPLIT_BATCH"] = "2"
            else:
                os.environ.pop("TAMD_SPLIT_BATCH", None)
            before = P.hip_device_split_subgraphs()
            rg = ref.RefGraph(b, ref.MODE_INT8, 1, device="HIP", dev_opt=HipOpt(b"HIP", C.sizeof(HipOpt), 0, 1, 0))
            rg.set_input(x)
            rg.run()
            split = P.hip_device_split_subgraphs() - before
            for _ in range(5):
                rg.run()
            t0 = time.perf_counter()
            for _ in range(runs):
                rg.run()
            res[mode].append((time.perf_counter() - t0) / runs * 1e6)
            outs[mode] = [o.copy() for o in rg.outputs()]
            rg.close()
            assert (split >= 1) == (mode == "default"), (mode, split)
    same = all(np.array_equal(a, c) for a, c in zip(outs["0"], outs["default"]))
    a, d = min(res["0"]), min(res["default"])
    print("%-14s int8 batch %3d through the plugin, blocking run_graph(), us per run (min of %d x %d): one graph %8.1f | two half-batch graphs %8.1f (%+5.1f %%) | outputs %s"
          % (name, batch, rounds, runs, a, d, 100 * (a / d - 1), "identical" if same else "DIFFER"))


if __name__ == "__main__":
    main()
