"""Parity at the BASELINE per-GPU batches: the kernels the batched numbers come from must be the
kernels the tests run.  Autotune and the launch heuristics pick different family members at these sizes (the wide
two-fragment depthwise kernel, deep-K igemm tiles, pw_stream<4,4>, the 64x64 uint8 tiles), so each BASELINE config is
run at its stated per-GPU batch through the C ABI and every output byte is compared with the REAL reference CPU backend
(oracle/_ref, multi-threaded; the C oracle where the prebuilt library did not travel).

  configs[1]  MobileNet-v1 int8 224^2, also at batch 64 (the b64 figure DESIGN.md quotes)
  configs[2]  ResNet-50 int8 224^2 batch 32
  configs[3]  YOLOv3-tiny uint8 416^2, 8 per GPU
  configs[4]  MobileNet-SSD uint8 300^2, 16 per GPU
"""
import os

import numpy as np
import pytest

from helpers import conv_graph
from oracle import oracle, ref_capi
from tengine_amd import capi, models, tm2

pytestmark = pytest.mark.gpu


def reference_outputs(g, tm_bytes, x, mode):
    """bytes of the real reference (conv_hcl / conv_ref / dw selection by its own score()), else the pinned oracle"""
    if ref_capi.available():
        threads = min(os.cpu_count() or 1, 64)
        return ref_capi.run_model(tm_bytes, x, mode, threads), "reference"
    return oracle.run_graph(g, x), "oracle"


def run_and_compare(name, dtype, batch, seed, device_only=False, res=None):
    kw = {"res": res} if res else {}
    g = models.build(name, dtype, batch, device_only=device_only, **kw)
    u8 = dtype == "uint8"
    x = models.synth_input(g, seed, tm2.DT_UINT8 if u8 else tm2.DT_INT8)
    tmb = tm2.write_tm2(g)
    want, kind = reference_outputs(g, tmb, x, ref_capi.MODE_UINT8 if u8 else ref_capi.MODE_INT8)
    gr = capi.Graph(tmb)
    gr.set_input(x)
    got = gr.run()
    kernels = [k["kernel"] for k in gr.profile(1)]
    again = gr.run()
    gr.close()
    assert len(want) == len(got)
    for i, (w, o) in enumerate(zip(want, got)):
        bad = np.count_nonzero(w.ravel() != o.ravel())
        assert bad == 0, "%s %s b%d output %d: %d / %d bytes differ from the %s" % (name, dtype, batch, i, bad, w.size, kind)
        assert len(np.unique(w)) > 3
        assert np.array_equal(o, again[i])            # hipGraph replay is idempotent
    return kernels


def test_mobilenet_v1_int8_batch64():
    kernels = run_and_compare("mobilenet_v1", "int8", 64, 21)
    # the batched depthwise layers run the tall-lane forms (two / four output rows per lane)
    assert any(k.startswith("dwconv3x3_i8<1,1,r") or "dwpw" in k or "pwdw" in k for k in kernels), kernels


def test_resnet50_int8_batch32():
    run_and_compare("resnet50", "int8", 32, 22, device_only=True)


def test_yolov3_tiny_uint8_416_batch8():
    run_and_compare("yolov3_tiny", "uint8", 8, 23)


def test_mssd_uint8_300_batch16():
    run_and_compare("mssd", "uint8", 16, 24)


WIDE_DW = [
    # n, c, hw, stride, pinned form (TAMD_PIN dw_form, None: the launcher's choice), expected kernel
    (16, 64, 112, 1, None, "dwconv3x3_i8<1,1,r4>"),   # large batch, tall map: four output rows per lane
    (32, 64, 112, 2, None, "dwconv3x3_i8<2,1,r2>"),
    (24, 128, 57, 1, None, "dwconv3x3_i8<1,1,r4>"),   # odd width and height: strip tail, last band has one row of four
    (64, 128, 57, 2, None, "dwconv3x3_i8<2,1,r2>"),   # stride 2 on an odd map: last window touches the right border, last band one row
    (64, 512, 14, 1, None, "dwconv3x3_i8<1,1,r2>"),   # MobileNet-v1 conv5_x/dw at batch 64
    (1, 64, 112, 1, None, "dwconv3x3_i8<1,1>"),       # the batch-1 forms
    (1, 64, 112, 2, None, "dwconv3x3_i8<2,1>"),
    (16, 64, 112, 1, "21", "dwconv3x3_i8<1,2>"),      # the two-fragment strips (alignbyte windows across two fragments), pinned
    (32, 64, 112, 2, "22", "dwconv3x3_i8<2,2,r2>"),
    (24, 128, 57, 1, "22", "dwconv3x3_i8<1,2,r2>"),   # odd width: strip tail of the 6-output strips, ragged rows
    (64, 128, 57, 2, "21", "dwconv3x3_i8<2,2>"),
    (3, 32, 9, 1, "14", "dwconv3x3_i8<1,1,r4>"),      # 9 rows in bands of four: the last band holds one
]


@pytest.mark.parametrize("n,c,hw,s,form,kernel", WIDE_DW)
def test_depthwise_variants_by_name(n, c, hw, s, form, kernel, monkeypatch):
    """the launcher's <stride, fragments, rows> choice is asserted by name, so no form (alignbyte windows across two fragments,
    partial last bands) can silently go untested; batch > 1 also switches the reference to its naive-ref epilogue"""
    if form:
        monkeypatch.setenv("TAMD_PIN", "dw_form=" + form)
    g, x = conv_graph(300 + n + c + hw + s, n, c, hw, hw, c, 3, s, 1, group=c, act=0)
    x[:] = np.random.default_rng(n + hw).integers(-127, 128, size=x.shape)
    want = oracle.run_graph(g, x)[0]
    os.environ["TAMD_FUSE_PWDW"] = "0"
    try:
        gr = capi.Graph(tm2.write_tm2(g))
    finally:
        os.environ.pop("TAMD_FUSE_PWDW", None)
    gr.set_input(x)
    got = gr.run()[0].reshape(want.shape)
    names = [k["kernel"] for k in gr.profile(1)]
    gr.close()
    assert names == [kernel], names
    assert np.array_equal(got, want), "%d bytes differ" % np.count_nonzero(got != want)
