"""Generates the committed golden fixtures by running the REAL reference (oracle/_ref, built from the
unmodified /root/reference sources by oracle/build_ref.py) on the seeded synthetic models.
Run here (where /root/reference exists):  python tests/golden/make_golden.py
The fixtures are tiny (model outputs only); the models themselves are re-synthesised from seeds."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import build_ref, ref_capi  # noqa: E402
from tengine_amd import models, tm2     # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def priorbox_fixture():
    """PriorBox outputs of the real reference (priorbox_ref.c) for tests/helpers.py PRIORBOX_CASES, uint8 and fp32, and
    mbox_priorbox of the uint8 MobileNet-SSD: python tests/golden/make_golden.py priorbox"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import PRIORBOX_CASES, priorbox_graph
    out = {}
    for case, kw in sorted(PRIORBOX_CASES.items()):
        for dt, mode, tag in ((tm2.DT_UINT8, ref_capi.MODE_UINT8, "uint8"), (tm2.DT_FP32, ref_capi.MODE_FP32, "fp32")):
            g, x = priorbox_graph(dtype=dt, **kw)
            out["%s_%s" % (case, tag)] = np.asarray(ref_capi.run_model(tm2.write_tm2(g), x, mode, 2)[0])
    g = models.build("mssd", "uint8", 1, tail=True, priorbox=True)
    x = models.synth_input(g, 5, tm2.DT_UINT8)
    out["mssd_mbox_priorbox_uint8"] = np.asarray(ref_capi.run_model(tm2.write_tm2(g), x, ref_capi.MODE_UINT8, os.cpu_count())[2])
    path = os.path.join(HERE, "priorbox_cases.npz")
    np.savez_compressed(path, **out)
    print(path, {k: (v.shape, str(v.dtype)) for k, v in out.items()})


def main():
    build_ref.build()
    if sys.argv[1:] == ["priorbox"]:
        return priorbox_fixture()
    priorbox_fixture()
    for name, batch, seed in [("mobilenet_v1", 1, 7)]:
        g = models.build(name, "int8", batch)
        x = models.synth_input(g, seed)
        out = ref_capi.run_model(tm2.write_tm2(g), x, ref_capi.MODE_INT8, os.cpu_count())[0]
        path = os.path.join(HERE, "%s_int8_seed%d.npy" % (name, seed))
        np.save(path, out)
        print(path, out.shape, out.dtype, "absmax", np.abs(out.astype(int)).max())
    # uint8: YOLOv3-tiny 416x416, both heads
    g = models.build("yolov3_tiny", "uint8", 1)
    x = models.synth_input(g, 3, tm2.DT_UINT8)
    outs = ref_capi.run_model(tm2.write_tm2(g), x, ref_capi.MODE_UINT8, os.cpu_count())
    path = os.path.join(HERE, "yolov3_tiny_uint8_416_seed3.npz")
    np.savez_compressed(path, **{"out%d" % i: o for i, o in enumerate(outs)})
    print(path, [o.shape for o in outs], [len(np.unique(o)) for o in outs])
    # uint8 classification nets: MobileNet-v1 (13 depthwise layers -> conv_ref formula) and ResNet-50 (eltwise, fc)
    for name, dev_only in [("mobilenet_v1", False), ("resnet50", True)]:
        g = models.build(name, "uint8", 1, device_only=dev_only)
        x = models.synth_input(g, 5, tm2.DT_UINT8)
        out = ref_capi.run_model(tm2.write_tm2(g), x, ref_capi.MODE_UINT8, os.cpu_count())[0]
        path = os.path.join(HERE, "%s_uint8_seed5.npy" % name)
        np.save(path, out)
        print(path, out.shape, len(np.unique(out)))
    # MobileNet-SSD 300x300 uint8 (BASELINE configs[4] stand-in): mbox_loc and mbox_conf after the on-graph
    # Permute -> Flatten -> Concat plumbing
    g = models.build("mssd", "uint8", 1)
    x = models.synth_input(g, 5, tm2.DT_UINT8)
    outs = ref_capi.run_model(tm2.write_tm2(g), x, ref_capi.MODE_UINT8, os.cpu_count())
    path = os.path.join(HERE, "mssd_uint8_300_seed5.npz")
    np.savez_compressed(path, **{"out%d" % i: o for i, o in enumerate(outs)})
    print(path, [o.shape for o in outs], [len(np.unique(o)) for o in outs])


if __name__ == "__main__":
    main()
