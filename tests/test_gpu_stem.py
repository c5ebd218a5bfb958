"""conv_first_pool.hip: the ResNet stem (7x7 / stride-2 convolution on the NCHW graph input + MAX pool 3x3 / 2) as ONE launch, the
conv map living in LDS only.  Bit-exact against the oracle's two-node result (what the reference computes: conv_kernel_x86.c
:1826-1889 then pooling_kernel_ref_int8.c:156-166) over interior / clipped tiles, odd maps, ragged channel counts, both
caffe_flavor values and differing pool scales; equal to the same graph run as two launches on the device; and the stem of the
real ResNet-50 (BASELINE configs[2]) takes it by default."""
import os

import numpy as np
import pytest

from helpers import stem_graph
from oracle import oracle
from tengine_amd import capi, models, tm2

pytestmark = pytest.mark.gpu


def run(g, x, fuse, **kw):
    os.environ["TAMD_PIN"] = "first_pool=%d" % fuse
    try:
        gr = capi.Graph(tm2.write_tm2(g), **kw)
    finally:
        os.environ.pop("TAMD_PIN", None)
    gr.set_input(x)
    out = gr.run()[0]
    names = [k["kernel"] for k in gr.profile(1)]
    return out, names, gr


# n, h, w, cout, kw, pad, act, caffe, same_scale
CASES = [
    (1, 224, 224, 64, 7, 3, 0, 1, True),         # ResNet-50 conv1 + pool1: 56 x 56 pooled = 8 x 7 full tiles, last window rows / columns clipped (caffe)
    (2, 224, 224, 64, 7, 3, 0, 1, True),
    (1, 64, 64, 32, 7, 3, 0, 1, True),           # one cout tile; pooled 16 x 16: partial tiles in both directions
    (1, 97, 75, 64, 7, 3, 0, 0, False),          # odd map, floor-mode pooling (no clipped windows), the pool rescales
    (3, 50, 62, 48, 7, 3, -1, 1, False),         # ragged couts (48: half a tile), no activation, batch 3
    (1, 39, 43, 96, 7, 3, 6, 1, True),           # three cout tiles, relu6
    (1, 33, 35, 128, 7, 3, 0, 1, False),         # four cout tiles
    (1, 40, 40, 64, 5, 2, 0, 1, True),           # 7 x 5 kernel: KW < 7 rides in the same 8-byte row pieces; pad 2
    (1, 30, 36, 64, 7, 0, 0, 1, True),           # no conv padding
    (1, 21, 21, 20, 7, 3, 0, 1, True),           # 20 couts: dword-granular pooled rows, tiny map (one partial tile)
]


@pytest.mark.parametrize("case", CASES, ids=[str(c) for c in CASES])
def test_stem_matches_the_oracle(case):
    n, h, w, c, kw, pad, act, caffe, same = case
    g, x = stem_graph(900 + h + w + c, n, h, w, c, kw, pad, act, caffe, same_scale=same)
    x[:] = np.random.default_rng(5).integers(-127, 128, size=x.shape)        # dense borders
    want = oracle.run_graph(g, x)[0]
    got, names, gr = run(g, x, 1)
    gr.close()
    assert names == ["conv_first_pool_i8"], names
    got = got.reshape(want.shape)
    bad = np.count_nonzero(got != want)
    assert bad == 0, "%s: %d / %d bytes differ (max |d| %d)" % (case, bad, want.size, np.abs(got.astype(int) - want.astype(int)).max())
    assert len(np.unique(want)) >= 8


def test_fused_equals_unfused_and_the_conv_map_is_refused():
    g, x = stem_graph(31, 2, 120, 88, 64, tail_conv=True)
    fused, names_f, gr_f = run(g, x, 1, keep_tensors=True)
    unfused, names_u, gr_u = run(g, x, 0)
    gr_u.close()
    assert names_f[0] == "conv_first_pool_i8" and len(names_f) == 2, names_f
    assert names_u[:2] == ["conv_first_i8", "pool_i8"] and len(names_u) == 3, names_u
    assert np.array_equal(fused, unfused)
    mid = [i for i, t in enumerate(g.tensors) if t.name == "mid"][0]
    with pytest.raises(capi.TamdError, match="fused"):
        gr_f.read_tensor(mid)
    # the pooled map is an inner tensor here: read it back and compare it with the oracle's
    pooled = [i for i, t in enumerate(g.tensors) if t.name == "pooled"][0]
    want = oracle.run_graph(g, x, keep_all=True)[pooled]
    got = gr_f.read_tensor(pooled)
    gr_f.close()
    assert np.array_equal(np.asarray(got).reshape(want.shape), want)


@pytest.mark.parametrize("geom", [(3, 1), (2, 2), (3, 2, "avg")])
def test_other_pools_keep_two_launches(geom):
    """anything but MAX 3x3 / 2 falls back to conv_first + pool, still bit-exact"""
    g, x = stem_graph(57, 1, 64, 64, 64, pool_k=geom[0], pool_s=geom[1])
    if len(geom) > 2:
        for n in g.nodes:
            if n.op == "Pooling":
                n.params["alg"] = tm2.POOL_AVG
    want = oracle.run_graph(g, x)[0]
    got, names, gr = run(g, x, 1)
    gr.close()
    assert names == ["conv_first_i8", "pool_i8"], names
    assert np.array_equal(got.reshape(want.shape), want)


def test_resnet50_stem_uses_the_fused_launch():
    g = models.build("resnet50", "int8", 2, device_only=True)
    x = models.synth_input(g, 11)
    want = oracle.run_graph(g, x)[0]
    gr = capi.Graph(tm2.write_tm2(g))
    gr.set_input(x)
    got = gr.run()[0]
    prof = gr.profile(1)
    gr.close()
    assert prof[0]["kernel"] == "conv_first_pool_i8" and prof[0]["node"] == "conv1+pool1", prof[0]
    assert np.array_equal(got.reshape(want.shape), want)
