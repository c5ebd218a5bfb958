"""The transform matrices of tengine_amd/csrc/winograd_f32.hip (F(2x2, 3x3): B^T, G, A^T as written in the kernels and in
graph_f32.hip's weight transform) restated in numpy: Y = A^T [ (G g G^T) . (B^T d B) ] A must be the 3x3 stride-1 convolution.
Pins the constants on the CPU; the device kernels are compared with the oracle in tests/test_gpu_parity_fp32.py."""
import numpy as np

BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64)      # wino_in_f32_k
G = np.array([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=np.float64)         # plan_winograd_f32
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)                                   # wino_out_f32_k


def direct(x, w, pad):
    c, h, ww = x.shape
    m = w.shape[0]
    xp = np.zeros((c, h + 2 * pad, ww + 2 * pad))
    xp[:, pad:pad + h, pad:pad + ww] = x
    oh, ow = h + 2 * pad - 2, ww + 2 * pad - 2
    y = np.zeros((m, oh, ow))
    for ky in range(3):
        for kx in range(3):
            y += np.einsum("mc,chw->mhw", w[:, :, ky, kx], xp[:, ky:ky + oh, kx:kx + ow])
    return y


def winograd(x, w, pad):
    c, h, ww = x.shape
    m = w.shape[0]
    oh, ow = h + 2 * pad - 2, ww + 2 * pad - 2
    th, tw = (oh + 1) // 2, (ow + 1) // 2
    xp = np.zeros((c, 2 * th + 2, 2 * tw + 2))
    xp[:, pad:pad + h, pad:pad + ww] = x                      # out-of-image taps are zeros, ragged last tiles included
    u = np.einsum("ij,mcjk,lk->mcil", G, w, G).astype(np.float32).astype(np.float64)     # rounded to binary32 once, as the planner does
    y = np.zeros((m, 2 * th, 2 * tw))
    for ty in range(th):
        for tx in range(tw):
            d = xp[:, 2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4]
            v = np.einsum("ij,cjk,lk->cil", BT, d, BT)
            mm = np.einsum("mcil,cil->mil", u, v)
            y[:, 2 * ty:2 * ty + 2, 2 * tx:2 * tx + 2] = np.einsum("ij,mjk,lk->mil", AT, mm, AT)
    return y[:, :oh, :ow]


def test_f23_transforms_reproduce_the_convolution():
    rng = np.random.default_rng(3)
    for (c, h, w, m, pad) in [(5, 8, 8, 4, 1), (3, 9, 11, 7, 1), (6, 7, 6, 2, 0), (1, 4, 4, 1, 1)]:
        x = rng.normal(0, 1, size=(c, h, w)).astype(np.float32).astype(np.float64)
        k = rng.normal(0, 0.3, size=(m, c, 3, 3)).astype(np.float32).astype(np.float64)
        want, got = direct(x, k, pad), winograd(x, k, pad)
        assert got.shape == want.shape
        assert np.abs(got - want).max() <= 1e-6 * max(1.0, np.abs(want).max()), (c, h, w, m, pad)
