"""pwdw.hip: pointwise conv fused with the depthwise 3x3 / global pooling that consumes it (one launch, the intermediate
tensor stays in LDS).  Bit-exact against the oracle -- whose two-node result is what the reference computes -- for
every MobileNet-v1 pair, for every tile configuration class (interior / clipped edge tiles, odd maps, both strides, pads
0..2, ragged channels, K tails, deep-K chunking, batch > 1 = the reference's naive-ref depthwise epilogue), and against
the same graph run unfused on the device."""
import os

import numpy as np
import pytest

from helpers import pwdw_graph
from oracle import oracle
from tengine_amd import capi, models, tm2

pytestmark = pytest.mark.gpu


def run(g, x, fuse, cfg=None):
    os.environ["TAMD_FUSE_PWDW"] = str(fuse)
    if cfg:
        os.environ["TAMD_PIN"] = "pwdw_cfg=" + cfg.replace(",", "x")        # THxTWxthreads (a TAMD_PIN value holds no comma)
    try:
        gr = capi.Graph(tm2.write_tm2(g))
    finally:
        os.environ.pop("TAMD_FUSE_PWDW", None)
        os.environ.pop("TAMD_PIN", None)
    gr.set_input(x)
    out = gr.run()[0]
    names = [k["kernel"] for k in gr.profile(1)]
    gr.close()
    return out, names


def check(g, x, cfg=None, tag=""):
    want = oracle.run_graph(g, x)[0]
    got, names = run(g, x, 2, cfg)
    assert len(names) == 1 and names[0].split("_i8")[0] in ("pwdw", "pwpool", "firstdw"), names
    got = got.reshape(want.shape)
    bad = np.count_nonzero(got != want)
    assert bad == 0, "%s %s: %d / %d bytes differ (max |d| %d)" % (tag, names[0], bad, want.size, np.abs(got.astype(int) - want.astype(int)).max())
    assert len(np.unique(want)) >= 3          # (the relu1 case clamps to {-1, 0, 1})
    return names[0]


# (cin, C, hw, stride of the depthwise that follows): pointwise_i + depthwise_{i+1} of MobileNet-v1 (SURVEY 8d layer list)
MBV1_PAIRS = [(32, 64, 112, 2), (64, 128, 56, 1), (128, 128, 56, 2), (128, 256, 28, 1), (256, 256, 28, 2), (256, 512, 14, 1),
              (512, 512, 14, 1), (512, 512, 14, 2), (512, 1024, 7, 1)]


@pytest.mark.parametrize("cin,c,hw,s", MBV1_PAIRS)
def test_mobilenet_pairs(cin, c, hw, s):
    g, x = pwdw_graph(400 + cin + c + hw + s, 1, cin, hw, hw, c, s, 1)
    check(g, x, tag="pair %d->%d@%d s%d" % (cin, c, hw, s))


def test_mobilenet_pw_pool_pair():
    g, x = pwdw_graph(431, 1, 1024, 7, 7, 1024, tail="pool")          # conv6/sep + pool6: K = 1024 -> chunked K loop
    assert check(g, x).startswith("pwpool_i8")


CFG_CASES = [
    # n, cin, h, w, C, s, p, act_pw, act_dw, "TH,TW,threads"
    (1, 32, 14, 14, 64, 1, 1, 0, 0, "14,14,256"),     # whole map per block: every border is padding
    (1, 32, 14, 14, 64, 1, 1, 0, 0, "7,14,512"),      # two row tiles: halo rows recomputed
    (1, 32, 14, 14, 64, 1, 1, 0, 0, "4,6,256"),       # ragged tiles in both directions (14 = 4+4+4+2, 6+6+2)
    (1, 32, 14, 14, 64, 1, 1, 0, 0, "1,4,256"),       # single-row tiles
    (1, 32, 14, 14, 64, 2, 1, 0, 0, "7,7,256"),       # stride 2, whole map
    (1, 32, 14, 14, 64, 2, 1, 0, 0, "3,4,512"),       # stride 2, ragged tiles
    (2, 20, 13, 11, 24, 1, 1, 6, 0, "5,7,256"),       # odd map, cin % 16 != 0 (K tail inside a step), C % 16 != 0, relu6 on the pointwise, batch 2
    (3, 20, 13, 11, 24, 2, 1, 0, 6, "3,3,256"),       # stride 2 on an odd map, batch 3 (naive-ref depthwise epilogue), relu6
    (1, 96, 9, 9, 48, 1, 0, 0, -1, "7,7,256"),        # pad 0 (valid depthwise), K = 96: second step half real
    (1, 96, 10, 9, 48, 2, 0, -1, 0, "2,2,256"),       # pad 0 stride 2
    (1, 48, 9, 9, 32, 1, 2, 0, 0, "4,4,256"),         # pad 2: two padding rings
    (2, 640, 7, 7, 48, 1, 1, 0, 0, "7,7,256"),        # K = 640: ten steps -> chunked K loop, weights re-read per tile
    (1, 1024, 7, 7, 32, 2, 1, 0, 0, "4,4,512"),       # K = 1024 stride 2
    (1, 16, 30, 30, 16, 1, 1, 0, 0, "8,28,512"),      # wide tile: 300-pixel region, single 16-deep K step
    (4, 64, 28, 28, 32, 1, 1, 1, 1, "14,14,256"),     # act code 1: relu6 on the hcl pointwise, relu1 clamp on the batch>1 depthwise
]


@pytest.mark.parametrize("case", CFG_CASES, ids=[str(c) for c in CFG_CASES])
def test_tile_configurations(case):
    n, cin, h, w, c, s, p, act_pw, act_dw, cfg = case
    g, x = pwdw_graph(500 + cin + h + c + s + p, n, cin, h, w, c, s, p, act_pw, act_dw)
    name = check(g, x, cfg, str(case))
    th, tw, threads = cfg.split(",")
    assert name == "pwdw_i8<s%d,%sx%s,%s>" % (s, th, tw, threads), name


# two / four 16-channel slices per block ("TH,TW,threads,2" / "..,4": the slices share a tile's address arithmetic, load and loop control; the intermediate
# tensor is slice-major in LDS).  Same cases as above where the channel count allows it, both strides, all three depthwise modes, ragged
# tiles, batch > 1, a K tail inside a step, C % 32 == 0 with C % 16 channels unused by the destination view
CFG2_CASES = [
    (1, 32, 14, 14, 64, 1, 1, 0, 0, "14,14,256,2"),
    (1, 32, 14, 14, 64, 1, 1, 0, 0, "4,6,256,2"),
    (1, 32, 14, 14, 64, 1, 1, 0, 0, "1,4,256,2"),       # small tile: one output per lane (mode 3)
    (1, 32, 14, 14, 64, 2, 1, 0, 0, "7,7,256,2"),
    (1, 32, 14, 14, 64, 2, 1, 0, 0, "3,4,512,2"),
    (2, 20, 13, 11, 32, 1, 1, 6, 0, "5,7,256,2"),       # odd map, cin % 16 != 0, relu6 on the pointwise, batch 2
    (3, 20, 13, 11, 64, 2, 1, 0, 6, "3,3,256,2"),       # stride 2 on an odd map, batch 3 (naive-ref depthwise epilogue)
    (1, 96, 9, 9, 96, 1, 0, 0, -1, "7,7,256,2"),        # pad 0, K = 96 (two steps), three slice pairs, no activation on the depthwise
    (1, 48, 9, 9, 32, 1, 2, 0, 0, "4,4,256,2"),         # pad 2
    (1, 16, 30, 30, 32, 1, 1, 0, 0, "8,28,512,2"),      # wide tile
    (4, 64, 28, 28, 32, 1, 1, 1, 1, "14,14,256,2"),     # act code 1
    (2, 32, 56, 56, 64, 2, 1, 0, 0, "14,28,512,2"),     # the MobileNet conv2_1/sep + conv2_2/dw shape at a batch-64 tile
    # four slices per block (K <= 128: the fragments of all four stay in registers)
    (1, 32, 14, 14, 64, 1, 1, 0, 0, "14,14,256,4"),
    (1, 32, 14, 14, 64, 1, 1, 0, 0, "1,4,256,4"),
    (1, 32, 14, 14, 64, 2, 1, 0, 0, "3,4,512,4"),
    (2, 20, 13, 11, 64, 1, 1, 6, 0, "5,7,256,4"),
    (1, 96, 9, 9, 128, 1, 0, 0, -1, "7,7,256,4"),
    (3, 128, 13, 11, 128, 2, 1, 0, 6, "3,3,256,4"),
    (2, 32, 56, 56, 64, 2, 1, 0, 0, "7,28,512,4"),
]


@pytest.mark.parametrize("case", CFG2_CASES, ids=[str(c) for c in CFG2_CASES])
def test_two_slices_per_block(case):
    n, cin, h, w, c, s, p, act_pw, act_dw, cfg = case
    g, x = pwdw_graph(900 + cin + h + c + s + p, n, cin, h, w, c, s, p, act_pw, act_dw)
    name = check(g, x, cfg, str(case))
    th, tw, threads, sl = cfg.split(",")
    assert name == "pwdw_i8<s%d,%sx%s,%s,c%d>" % (s, th, tw, threads, 16 * int(sl)), name


@pytest.mark.parametrize("alg", [0, 1])
@pytest.mark.parametrize("shape", [(1, 64, 7, 7, 32), (3, 40, 5, 9, 24), (1, 256, 14, 14, 64)])
def test_pointwise_plus_global_pool(alg, shape):
    n, cin, h, w, c = shape
    g, x = pwdw_graph(600 + cin + h + alg, n, cin, h, w, c, tail="pool", pool_alg=alg)
    assert check(g, x).startswith("pwpool_i8")


FIRST_CASES = [
    # n, cin, h, w, C, (k, stride, pad[, dil]) of the first conv, depthwise stride, "TH,TW,threads"
    (1, 3, 224, 224, 32, (3, 2, 1), 1, None),            # MobileNet-v1 conv1 + conv2_1/dw (the planner's own tile choice)
    (1, 3, 224, 224, 32, (3, 2, 1), 1, "7,14,512"),
    (2, 3, 37, 41, 24, (3, 2, 1), 2, "3,4,256"),         # odd sizes, C % 16 != 0, stride-2 tail, batch 2 (naive-ref depthwise epilogue)
    (1, 3, 30, 30, 16, (3, 1, 1), 1, "8,28,512"),        # stride-1 first conv
    (1, 1, 20, 22, 16, (3, 2, 2), 1, "4,4,256"),         # one input channel, pad 2: two columns / rows of padding
    (3, 4, 18, 18, 32, (3, 1, 0), 1, "16,16,256"),       # 4 channels (12 patch rows: third k block), no padding, whole map per block
    (1, 3, 9, 5, 16, (3, 1, 1), 1, "5,6,256"),           # 5-pixel rows: right border inside the 4-byte row loads
    (1, 4, 16, 16, 16, (4, 1, 1), 2, "2,3,256"),         # 4x4 kernel, 4 channels: all 16 patch rows, KW = 4
    (1, 3, 224, 224, 32, (3, 2, 1), 1, "7,14,512,2"),    # two slices per block: the patch gather is shared by the 32 output channels
    (2, 3, 37, 41, 32, (3, 2, 1), 2, "3,4,256,2"),
    (3, 4, 18, 18, 64, (3, 1, 0), 1, "16,16,256,2"),
    (1, 3, 64, 64, 64, (3, 2, 1), 1, "7,14,512,4"),      # four slices
]


@pytest.mark.parametrize("case", FIRST_CASES, ids=[str(c) for c in FIRST_CASES])
def test_first_conv_plus_depthwise(case):
    """the network's first conv, gathered from the NCHW graph input, fused with the depthwise 3x3 behind it"""
    n, cin, h, w, c, first, s, cfg = case
    g, x = pwdw_graph(700 + h + c + s, n, cin, h, w, c, s, 1, 0, 0, first=first)
    x[:] = np.random.default_rng(3).integers(-127, 128, size=x.shape)       # dense borders
    name = check(g, x, cfg, str(case))
    assert name.startswith("firstdw_i8"), name


def test_fused_equals_unfused_on_device_and_intermediate_is_refused():
    g, x = pwdw_graph(77, 1, 128, 28, 28, 128, 1, 1)
    fused, names_f = run(g, x, 2)
    unfused, names_u = run(g, x, 0)
    assert len(names_f) == 1 and len(names_u) == 2, (names_f, names_u)
    assert np.array_equal(fused, unfused)
    # the intermediate tensor only exists in LDS: reading it must fail loudly, not return the memset zeros
    os.environ["TAMD_FUSE_PWDW"] = "2"
    try:
        gr = capi.Graph(tm2.write_tm2(g), keep_tensors=True)
    finally:
        del os.environ["TAMD_FUSE_PWDW"]
    gr.set_input(x)
    gr.run()
    mid = [i for i, t in enumerate(g.tensors) if t.name == "mid"][0]
    with pytest.raises(capi.TamdError, match="fused"):
        gr.read_tensor(mid)
    gr.close()


def test_mobilenet_v1_batch1_default_plan_uses_fused_launches():
    """BASELINE configs[1]: the default plan (autotune decides pair by pair) must fuse, and must still be bit-exact"""
    g = models.build("mobilenet_v1", "int8", 1)
    x = models.synth_input(g, 7)
    want = oracle.run_graph(g, x)[0]
    gr = capi.Graph(tm2.write_tm2(g))
    gr.set_input(x)
    got = gr.run()[0]
    names = [k["kernel"] for k in gr.profile(1)]
    gr.close()
    assert np.array_equal(got.reshape(want.shape), want)
    assert sum(n.split("_i8")[0] in ("pwdw", "pwpool", "firstdw") for n in names) >= 8, names
    assert len(names) <= 17, names
