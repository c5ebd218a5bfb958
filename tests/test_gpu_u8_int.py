"""The opt-in INTEGER uint8 path (tamd_options.u8_integer / TAMD_U8_INT=1; csrc/u8i_kernels.hip) -- VERDICT r3 item 6.

The reference simulates uint8 in fp32 (conv_kernel_x86.c:68-80, :1703-1794); the default device path repeats that chain byte for
byte on the fp32 MFMA.  The integer path computes the exact int32 sum on the int8 MFMA and requantises once; BASELINE.md section 2 /
SURVEY section 7 step 5 state its bar: within ONE quantisation step of the reference, mismatch fraction reported.  Checked here:
  * every tile configuration against an operation-for-operation numpy model of the integer formula: byte-exact (kernel bugs show
    up as exact mismatches, not as tolerance noise);
  * against the reference's bytes (the pinned C oracle / the real reference): max |d| <= 1, mismatch fraction printed and bounded;
  * the fused tails (leaky ReLU, 2x2 max-pool, concat-by-offset) shared with the byte-exact kernels;
  * YOLOv3-tiny 416^2 batch 8 and MobileNet-SSD 300^2 batch 16 end to end against the REAL reference, with the histogram;
  * the option is OFF by default: the default plan holds no conv_u8i launch."""
import os

import numpy as np
import pytest

from helpers import u8_conv_graph, u8_conv_int_model, u8_conv_pool_graph
from oracle import oracle, ref_capi
from tengine_amd import capi, models, tm2

pytestmark = pytest.mark.gpu


def run_int(g, x, cfg=None, want_kernel="conv_u8i", env=None):
    pins = dict(env or {})                # TAMD_PIN keys (csrc/env.h): u8i_cfg, u8i_tiles, u8i_cg, u8i_pw
    if cfg is not None:
        pins["u8i_cfg"] = str(cfg)
    from helpers import pinned
    with pinned(**pins):
        gr = capi.Graph(tm2.write_tm2(g), u8_integer=True)
    gr.set_input(x)
    out = gr.run()
    kernels = [k["kernel"] for k in gr.profile(1)]
    gr.close()
    if want_kernel:
        assert any(k.startswith(want_kernel) for k in kernels), kernels
    return out, kernels


def hist(want, got):
    d = np.abs(want.astype(int).ravel() - got.astype(int).ravel())
    return d.max(), float((d > 0).mean()), np.bincount(d, minlength=3)[:4]


INT_CONV = [
    # n, cin, h, w, cout, k, s, p, act, bias, dil
    (1, 16, 40, 40, 32, 3, 1, 1, -1, True, 1),       # one 32-channel chunk half full (padded channels meet w' = beta)
    (1, 64, 20, 20, 128, 3, 1, 1, 0, True, 1),
    (2, 128, 13, 13, 255, 1, 1, 0, -1, True, 1),     # 169 px per image: ragged last pixel tile; cout 255: ragged channel tile
    (1, 256, 13, 13, 512, 3, 1, 1, 6, True, 1),      # K = 2304, relu6
    (1, 30, 26, 26, 70, 3, 2, 1, 0, False, 1),       # stride 2, no bias, cin % 32 != 0
    (3, 9, 9, 11, 13, 3, 1, 1, 1, True, 1),          # odd everything, batch 3
    (1, 32, 12, 12, 16, 3, 1, 2, 0, True, 2),        # dilation 2, pad 2
    (1, 512, 7, 7, 64, 3, 1, 1, 0, True, 1),         # 16 chunks, 49 px
    (1, 40, 5, 64, 48, 1, 1, 0, -1, True, 1),        # single-row tiles: the patch is a column range of one row
    (1, 24, 150, 150, 64, 1, 1, 0, 0, True, 1),      # mssd conv1 shape class: wide map, 1x1
    (1, 16, 104, 104, 32, 3, 1, 1, -1, True, 1),     # wide map, 3x3: a linear tile's bounding box is too large -> 2-D tiles
    (1, 40, 70, 210, 24, 3, 1, 1, 0, True, 1),       # 2-D tiles with ragged right / bottom edges
    (2, 48, 10, 10, 24, 5, 1, 2, -1, True, 1),       # 5x5
    (1, 64, 19, 19, 96, 3, 2, 1, 0, True, 1),        # SSD extra-layer class: 3x3 stride 2 on an odd map
    (1, 8, 8, 4, 16, 3, 1, 1, -1, True, 1),          # the narrowest map the dword staging takes (W = 4)
    (2, 64, 1, 1, 100, 1, 1, 0, -1, True, 1),        # pointwise on a 1x1 map: one live pixel per quad, byte-wise stores
    (3, 72, 3, 3, 40, 1, 1, 0, 0, True, 1),          # 9 pixels per image: partial quads, images back to back
    (1, 8, 20, 20, 32, 1, 1, 0, 6, False, 1),        # 8 input channels: three quarters of the K step are padding
    (1, 256, 19, 19, 126, 1, 1, 0, -1, True, 1),     # SSD conf head class: 361 pixels (1 mod 4), cout % 32 != 0
]


@pytest.mark.parametrize("case", INT_CONV, ids=[str(c) for c in INT_CONV])
def test_conv_u8_integer_matches_its_model_and_the_reference_within_one_step(case):
    n, cin, h, w, cout, k, s, p, act, bias, dil = case
    g, x = u8_conv_graph(77 + cin + cout, n, cin, h, w, cout, k, s, p, 1, act, bias, dil)
    model = u8_conv_int_model(g, x)
    ref = oracle.run_graph(g, x)[0]
    seen = set()
    # every tile shape; then 2-D pixel tiles wherever they fit (wide maps get them anyway) and one 32-channel group per chunk
    # for 1x1 layers also the six shapes of the register-only pointwise kernel (ids 6..11), and the general kernel with it switched off
    runs = [(None, None)] + [(c, None) for c in range(6)] + [(c, {"u8i_tiles": "2"}) for c in (0, 2, 3, 5)] + [(1, {"u8i_cg": "2"}), (4, {"u8i_cg": "4"})]
    if k == 1 and s == 1 and p == 0:
        runs += [(c, None) for c in range(6, 12)] + ([(None, {"u8i_pw": "0"}), (0, {"u8i_pw": "0"})] if w >= 4 else [])
    if w < 4:
        runs = [(None, None)] + [(c, None) for c in range(6, 12)]          # maps narrower than a dword: only the pointwise kernel takes them
    for cfg, env in runs:
        (got,), kernels = run_int(g, x, cfg, env=env)
        got = got.reshape(model.shape)
        name = [k for k in kernels if k.startswith("conv_u8i")][0]
        seen.add(name)
        bad = np.count_nonzero(got != model)
        assert bad == 0, "%s cfg %s %s (%s): %d / %d bytes differ from the integer model (max |d| %d)" % (
            case, cfg, env, name, bad, model.size, np.abs(got.astype(int) - model.astype(int)).max())
    mx, frac, h3 = hist(ref, model)
    print("\n  %s: vs reference bytes max |d| = %d, mismatch fraction %.2e, histogram |d| = 0,1,2: %s; tile shapes run: %s"
          % (case, mx, frac, h3[:3], sorted(seen)))
    assert mx <= 1 and frac < 2e-3
    assert len(np.unique(model)) > 3


INT_RGB = [
    # n, cin, h, w, cout, stride, pad, act      first layers: 3x3 on <= 4 channels, the whole K in one 16x16x64 MFMA
    (1, 3, 64, 64, 16, 1, 1, -1),          # YOLOv3-tiny conv0 class
    (2, 3, 50, 70, 32, 2, 1, 0),           # MobileNet-SSD conv0 class: stride 2, ragged windows, two channel tiles
    (1, 4, 33, 35, 24, 1, 1, 6),           # four channels, cout % 16 != 0, relu6
    (1, 1, 20, 20, 8, 1, 0, -1),           # one channel, no padding
    (3, 3, 17, 4, 64, 1, 1, -1),           # the narrowest map, four channel tiles, batch 3
    (1, 3, 130, 130, 16, 1, 1, -1),        # many windows
]


@pytest.mark.parametrize("case", INT_RGB, ids=[str(c) for c in INT_RGB])
def test_conv_u8_integer_first_layer_kernel(case):
    n, cin, h, w, cout, s_, p, act = case
    for zw in (None, 128):                 # beta != 0 (column sums) and beta == 0
        g, x = u8_conv_graph(200 + cin + cout, n, cin, h, w, cout, 3, s_, p, 1, act, True, 1, w_zp=zw)
        model = u8_conv_int_model(g, x)
        (got,), kernels = run_int(g, x, want_kernel="conv_u8i_rgb3x3")
        got = got.reshape(model.shape)
        bad = np.count_nonzero(got != model)
        assert bad == 0, "%s: %d / %d bytes differ from the integer model (max |d| %d)" % (case, bad, model.size, np.abs(got.astype(int) - model.astype(int)).max())
        mx, frac, _ = hist(oracle.run_graph(g, x)[0], model)
        assert mx <= 1 and frac < 2e-3
        assert len(np.unique(model)) > 3


@pytest.mark.parametrize("zps", [(0, 0, 0), (255, 255, 255), (0, 255, 128), (255, 0, 7), (131, 128, 20), (128, 128, 128)])
def test_conv_u8_integer_extreme_zero_points(zps):
    # w_zp == 128: beta == 0, the column sums are skipped; in_zp == 128: alpha == 0, the pad byte is 0
    g, x = u8_conv_graph(5, 1, 32, 10, 10, 48, 3, 1, 1, act=-1, in_zp=zps[0], w_zp=zps[1], out_zp=zps[2])
    model = u8_conv_int_model(g, x)
    (got,), _ = run_int(g, x)
    assert np.array_equal(got.reshape(model.shape), model)
    ref = oracle.run_graph(g, x)[0]
    assert hist(ref, model)[0] <= 1


@pytest.mark.parametrize("case", [
    (1, 16, 16, 16, 32, True, False),      # conv -> leaky -> pool, fused: window-major pixel order inside the tile
    (2, 32, 26, 26, 64, True, False),      # 676 px: tiles that cross window rows
    (1, 16, 208, 208, 32, True, False),    # YOLO conv1 class: single-window-row tiles (column-range patches)
    (1, 24, 12, 12, 40, False, False),     # conv -> pool without the ReLU node
    (2, 3, 48, 40, 16, True, False),       # the first-layer kernel with both tails (YOLOv3-tiny conv0)
    (1, 16, 16, 16, 32, True, True),       # the unpooled tensor has a second reader: stored as well
])
@pytest.mark.parametrize("tiles2d", [False, True])
def test_conv_u8_integer_fused_tails(case, tiles2d):
    n, cin, h, w, cout, relu, second = case
    g, x = u8_conv_pool_graph(3 + cin, n, cin, h, w, cout, relu=relu, second_reader=second)
    want = oracle.run_graph(g, x)
    got, kernels = run_int(g, x, env={"u8i_tiles": "2"} if tiles2d else None)
    if (h * w) % 8 == 0:                       # what the planner asks of a fused pool (the byte-exact kernels' tail-pixel rule)
        assert any(k.startswith("conv_u8i") and "+maxpool" in k for k in kernels), kernels
    for w_, o in zip(want, got):
        mx, frac, h3 = hist(w_, o.reshape(w_.shape))
        print("\n  %s: max |d| = %d, mismatch fraction %.2e" % (case, mx, frac))
        assert mx <= 1 and frac < 2e-3
    # the same graph with the tails as separate launches gives the SAME integer-path bytes (the tails are applied to the bytes)
    old = {k: os.environ.get(k) for k in ("TAMD_FUSE_POOL", "TAMD_FUSE_RELU")}
    os.environ["TAMD_FUSE_POOL"] = "0"; os.environ["TAMD_FUSE_RELU"] = "0"
    try:
        apart, k2 = run_int(g, x)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert not any("+maxpool" in k for k in k2)
    for a_, b_ in zip(got, apart):
        assert np.array_equal(a_, b_)


def test_conv_u8_integer_writes_into_a_concat_view():
    """two convolutions whose outputs carry the concat's quantisation: both write their channels straight into the concat buffer"""
    g, x = u8_conv_graph(11, 2, 32, 13, 13, 48, 3, 1, 1, act=-1)
    conv = g.nodes[-1]
    yt = g.tensors[conv.outputs[0]]
    rng = np.random.default_rng(5)
    w2 = rng.integers(0, 256, size=(24, 32, 1, 1)).astype(np.uint8)
    wt = g.tensors[conv.inputs[1]]
    ins2 = [conv.inputs[0], g.add_const("w2", w2, tm2.DT_UINT8, [wt.scales[0] * 3.0], [121])]
    y2 = g.add_tensor("out2", [2, 24, 13, 13], tm2.DT_UINT8, tm2.TT_VAR, None, list(yt.scales), list(yt.zero_points))
    g.add_node("conv2", "Convolution", ins2, [y2], kernel_h=1, kernel_w=1, stride_h=1, stride_w=1, dilation_h=1, dilation_w=1,
               input_channel=32, output_channel=24, group=1, activation=-1, pad_h0=0, pad_w0=0, pad_h1=0, pad_w1=0)
    cat = g.add_tensor("cat", [2, 72, 13, 13], tm2.DT_UINT8, tm2.TT_VAR, None, list(yt.scales), list(yt.zero_points))
    g.output_nodes = [g.add_node("route", "Concat", [conv.outputs[0], y2], [cat], axis=1)]
    want = oracle.run_graph(g, x)
    got, kernels = run_int(g, x)
    assert sum(k.startswith("conv_u8i") for k in kernels) == 2 and not any("concat" in k for k in kernels), kernels
    for w_, o in zip(want, got):
        mx, frac, _ = hist(w_, o.reshape(w_.shape))
        assert mx <= 1 and frac < 2e-3
        assert len(np.unique(w_)) > 3


def _one_conv_graph(g, node):
    """the single-node graph of conv `node` of model graph `g` (same parameters, weights, quantisation)"""
    g1 = tm2.Graph(name="one_conv")
    t_in = g.tensors[node.inputs[0]]
    ins = [g1.add_input("data", list(t_in.dims), tm2.DT_UINT8, list(t_in.scales), list(t_in.zero_points))]
    for ti in node.inputs[1:]:
        t = g.tensors[ti]
        ins.append(g1.add_const(t.name, t.data, t.dtype, list(t.scales) if t.scales else None, list(t.zero_points) if t.zero_points else None))
    t_out = g.tensors[node.outputs[0]]
    y = g1.add_tensor("out", list(t_out.dims), tm2.DT_UINT8, tm2.TT_VAR, None, list(t_out.scales), list(t_out.zero_points))
    g1.output_nodes = [g1.add_node("conv", "Convolution", ins, [y], **node.params)]
    return g1


def _whole_model(name, batch, seed):
    """(1) TEACHER FORCED, the bar of BASELINE.md section 2: every convolution the integer path takes, fed the reference's own input
    bytes of that layer, lands within one step of the reference's output bytes.  (2) End to end the two arithmetics drift apart
    like any two non-identical implementations of a quantised network do (one differing byte perturbs every output it feeds; it
    is the reference's fp32 rounding noise that is being amplified, not an error of either side): the histogram is reported and
    bounded, the bytes cannot be expected within one step."""
    g = models.build(name, "uint8", batch)
    x = models.synth_input(g, seed, tm2.DT_UINT8)
    tmb = tm2.write_tm2(g)
    if ref_capi.available():
        want, kind = ref_capi.run_model(tmb, x, ref_capi.MODE_UINT8, min(os.cpu_count() or 1, 64)), "reference"
    else:
        want, kind = oracle.run_graph(g, x), "oracle"
    # the byte-exact device graph with every tensor kept and no fused tails = the reference's bytes of every tensor
    old = {k: os.environ.get(k) for k in ("TAMD_FUSE_POOL", "TAMD_FUSE_RELU")}
    os.environ["TAMD_FUSE_POOL"] = "0"; os.environ["TAMD_FUSE_RELU"] = "0"
    try:
        gd = capi.Graph(tmb, keep_tensors=True)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    gd.set_input(x)
    exact = [o.copy() for o in gd.run()]
    kd = [k["kernel"] for k in gd.profile(1)]
    assert not any(k.startswith("conv_u8i") for k in kd), kd          # OFF by default
    for w_, e in zip(want, exact):
        assert np.array_equal(w_.ravel(), e.ravel()), "default path no longer byte-exact"
    # how far does the REFERENCE arithmetic itself move when one input byte in ten thousand changes by one step?  (the yardstick for the
    # end-to-end comparison below: a quantised network amplifies any difference, whichever side it comes from)
    rng = np.random.default_rng(seed)
    xp = x.copy().ravel()
    idx = rng.choice(xp.size, max(1, xp.size // 10000), replace=False)
    xp[idx] = np.where(xp[idx] < 255, xp[idx] + 1, xp[idx] - 1)
    gd.set_input(xp.reshape(x.shape))
    moved = [o.copy() for o in gd.run()]
    for i, (e, m_) in enumerate(zip(exact, moved)):
        dd = np.abs(e.astype(int).ravel() - m_.astype(int).ravel())
        print("\n  %s uint8 b%d output %d, BYTE-EXACT path, 1e-4 of the input bytes moved by one step: max |d| = %d, mean |d| = %.3f steps, %.3f of the bytes differ"
              % (name, batch, i, dd.max(), dd.mean(), (dd > 0).mean()))
    gd.set_input(x)
    gd.run()
    forced, worst_forced, total, mism = 0, 0, 0, 0
    for node in g.nodes:
        if node.op != "Convolution" or node.params.get("group", 1) != 1:
            continue
        t_in = g.tensors[node.inputs[0]]
        if t_in.dims[1] < 8 or t_in.dims[3] < 4:
            continue
        xin = x if t_in.ttype == tm2.TT_INPUT else gd.read_tensor(node.inputs[0])
        ref_out = gd.read_tensor(node.outputs[0])
        g1 = _one_conv_graph(g, node)
        (got,), k1 = run_int(g1, np.ascontiguousarray(xin).reshape(t_in.dims))
        mx, frac, h3 = hist(ref_out, got)
        worst_forced = max(worst_forced, mx)
        total += ref_out.size; mism += int(round(frac * ref_out.size))
        forced += 1
        assert mx <= 1, "%s %s: teacher-forced integer conv differs from the reference by %d steps" % (name, node.name, mx)
    gd.close()
    print("\n  %s uint8 b%d, TEACHER FORCED: %d integer convolutions on the reference's own inputs: max |d| = %d, %d of %d bytes differ (%.2e)"
          % (name, batch, forced, worst_forced, mism, total, mism / max(total, 1)))
    assert forced >= 8
    gr = capi.Graph(tmb, u8_integer=True)
    gr.set_input(x)
    got = [o.copy() for o in gr.run()]
    kernels = [k["kernel"] for k in gr.profile(1)]
    again = gr.run()
    gr.close()
    n_int = sum(k.startswith("conv_u8i") for k in kernels)
    assert n_int >= 8, kernels
    for i, (w_, o) in enumerate(zip(want, got)):
        assert np.array_equal(o, again[i])
        d = np.abs(w_.astype(int).ravel() - o.astype(int).ravel())
        print("  %s uint8 b%d END TO END output %d vs the %s: max |d| = %d, mean |d| = %.3f steps, mismatch fraction %.3f, histogram |d| = 0..4: %s (%d integer convs)"
              % (name, batch, i, kind, d.max(), d.mean(), (d > 0).mean(), np.bincount(d, minlength=5)[:5], n_int))
        assert d.mean() < 1.0 and d.max() <= 8
    return worst_forced


def test_yolov3_tiny_uint8_b8_integer_path_within_one_step():
    assert _whole_model("yolov3_tiny", 8, 23) <= 1


def test_mssd_uint8_b16_integer_path_within_one_step():
    assert _whole_model("mssd", 16, 24) <= 1
