"""dwpw.hip: depthwise 3x3 (stride 1) fused with the pointwise conv that consumes it (one launch, the depthwise map only in LDS).
Bit-exact against the oracle -- whose two-node result is what the reference computes, including the int8 rounding of the
intermediate tensor and, at batch > 1, the naive-ref epilogue of the depthwise node -- over full and partial row tiles, maps
narrower than a tile row, ragged channel stages, 1 .. 8 output-channel wave slices, paddings 0 / 1, and against the same graph run
as two launches on the device."""
import os

import numpy as np
import pytest

from helpers import dwpw_graph
from oracle import oracle
from tengine_amd import capi, models, tm2

pytestmark = pytest.mark.gpu


def run(g, x, fuse, **kw):
    os.environ["TAMD_FUSE_DWPW"] = str(fuse)
    try:
        gr = capi.Graph(tm2.write_tm2(g), **kw)
    finally:
        os.environ.pop("TAMD_FUSE_DWPW", None)
    gr.set_input(x)
    out = gr.run()[0]
    names = [k["kernel"] for k in gr.profile(1)]
    return out, names, gr


# n, c, h, w, cout, pad, act_dw, act_pw, bias
CASES = [
    (8, 512, 14, 14, 512, 1, 0, 0, True),        # MobileNet-v1 conv5_x/dw + conv5_x/sep (batch 8: 112 rows = 28 whole tiles)
    (3, 64, 9, 11, 64, 1, 0, 0, True),           # 27 rows: the last tile holds three; 11-wide rows: five dead columns; one wave slice
    (2, 128, 16, 16, 128, 1, 6, -1, True),       # 16-wide rows: no dead column; relu6 on the depthwise node, none on the pointwise
    (5, 20, 7, 7, 64, 1, 0, 0, True),            # C = 20: one ragged stage (cw = 32), channels 20 .. 31 are padding
    (1, 256, 14, 14, 320, 1, 0, 0, False),       # five wave slices, no bias, batch 1 (the hcl depthwise epilogue)
    (3, 32, 10, 10, 64, 0, -1, 0, True),         # no padding: 8 x 8 outputs
    (2, 384, 12, 13, 448, 1, 0, 6, True),        # three stages, seven wave slices
    (2, 1024, 7, 7, 128, 1, 0, 0, True),         # round 6: 1024 depthwise channels -- more per-channel constants than two LDS units per thread (the copy's tail loop), eight stages
    (1, 2048, 5, 6, 64, 1, 6, 0, False),         # .. the largest channel count whose constants fit (cw = 2048: 57 KB of LDS), sixteen stages
]


@pytest.mark.parametrize("case", CASES, ids=[str(c) for c in CASES])
def test_dwpw_matches_the_oracle_and_two_launches(case):
    n, c, h, w, cout, pad, act_dw, act_pw, bias = case
    g, x = dwpw_graph(500 + c + h + cout, n, c, h, w, cout, pad, act_dw, act_pw, bias)
    x[:] = np.random.default_rng(9).integers(-127, 128, size=x.shape)          # dense borders
    want = oracle.run_graph(g, x)[0]
    got, names, gr = run(g, x, 2)
    gr.close()
    assert names == ["dwpw_i8"], names
    got = got.reshape(want.shape)
    bad = np.count_nonzero(got != want)
    assert bad == 0, "%s: %d / %d bytes differ (max |d| %d)" % (case, bad, want.size, np.abs(got.astype(int) - want.astype(int)).max())
    assert len(np.unique(want)) >= 5
    two, names2, gr2 = run(g, x, 0)
    gr2.close()
    assert len(names2) == 2 and "dwpw_i8" not in names2, names2
    assert np.array_equal(two.reshape(want.shape), want)


def test_the_depthwise_map_is_refused_when_fused():
    g, x = dwpw_graph(61, 4, 64, 14, 14, 128)
    got, names, gr = run(g, x, 2, keep_tensors=True)
    assert names == ["dwpw_i8"], names
    mid = [i for i, t in enumerate(g.tensors) if t.name == "mid"][0]
    with pytest.raises(capi.TamdError, match="fused"):
        gr.read_tensor(mid)
    gr.close()


def test_unsupported_pairs_keep_two_launches():
    """a 20-wide map, cout not in wave slices, more than 2048 channels: two launches, still bit-exact (TAMD_FUSE_DWPW=2 only forces where it applies)"""
    for args in [(2, 64, 20, 20, 64), (2, 64, 12, 12, 48), (1, 2064, 4, 4, 64)]:      # (.. 2064 channels: their constants do not fit the LDS block)
        g, x = dwpw_graph(71 + args[2], *args)
        want = oracle.run_graph(g, x)[0]
        got, names, gr = run(g, x, 2)
        gr.close()
        assert "dwpw_i8" not in names and len(names) == 2, names
        assert np.array_equal(got.reshape(want.shape), want)
