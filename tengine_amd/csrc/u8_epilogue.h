// Requantisation / fused-tail helpers shared by the uint8 kernels (u8_kernels.hip: byte-exact fp32 chains; u8i_kernels.hip:
// the opt-in integer path).  Every expression is the reference's, cited per function.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "kernels.h"

namespace tamd {

// compile-time loop (register-ring slots and accumulator arrays must be indexed by constants)
template <int I, int N, typename F>
__device__ __forceinline__ void u8_static_for(F&& f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        u8_static_for<I + 1, N>(f);
    }
}

// (int)(round(s / out_scale) + zp), clamp [0,255] -- conv_kernel_x86.c:1783-1788, conv_kernel_ref_uint8.c:177-182,
// fc_ref.c:196-202, eltwise_ref.c:571-578
__device__ __forceinline__ int quant_round_div(float s, float out_scale, int zp)
{
    float r = roundf(__fdiv_rn(s, out_scale));
    r = fminf(fmaxf(r, -65536.f), 65536.f);
    return (int)r + zp;
}
__device__ __forceinline__ uint8_t sat_u8(int v) { return (uint8_t)min(max(v, 0), 255); }

// sat_u8(quant_round_div(s, out_scale, zp)) without the IEEE division on the common path (it was a third of the VALU work of
// a depthwise output).  y = fma(s, fl(1/out_scale), copysign(0.5 + e, s)), e = 2^-13: for |d| < 300, d = fl(s / out_scale),
// |s * inv - d| <= 2 |d| 2^-24 and y rounds within 2^-16, together < 5.1e-5 < e, so trunc(y) = round_half_away(d) unless
// fract(|y|) < 2e -- those ~2.4e-4 of the values take the reference expression.  Beyond |y| = 300 the byte is saturated
// whatever the rounding did (0 <= zp <= 255), so nothing there is handed over.  Only for SATURATING call sites: the pooling
// node has no lower clamp (pooled_byte wraps), it keeps quant_round_div.  tests/csrc/u8_round_check.c replays this on the host.
__device__ __forceinline__ uint8_t quant_round_sat_u8(float s, float out_scale, int zp)
{
    if ((unsigned)zp > 255u) return sat_u8(quant_round_div(s, out_scale, zp));       // uniform; never taken for a uint8 tensor
    const float inv = __fdiv_rn(1.0f, out_scale);                                    // uniform: hoisted out of the pixel loops
    const float y = __fmaf_rn(s, inv, copysignf(0.5f + 0x1p-13f, s));
    const float ay = fabsf(y);
    int r = (int)fminf(fmaxf(y, -65536.f), 65536.f);                                 // truncates
    if (__builtin_amdgcn_fractf(ay) < 0x1p-12f && ay < 300.5f) r = (int)fminf(fmaxf(roundf(__fdiv_rn(s, out_scale)), -65536.f), 65536.f);
    return sat_u8(r + zp);
}

// The same function with the reciprocal handed in (one division per kernel, not per call site) and the rare hand-over to the
// reference expression behind a WAVE-level test: the ~2.4e-4 of the values that need the IEEE division no longer make every lane
// of every wave pay for it (left to the compiler the two-sided `if` is if-converted: the division runs always, predicated).
// Identical results: the same y, the same test, the same fallback.
__device__ __forceinline__ uint8_t quant_round_sat_u8_w(float s, float out_scale, float inv, int zp)
{
    const float y = __fmaf_rn(s, inv, copysignf(0.5f + 0x1p-13f, s));
    const float ay = fabsf(y);
    int r = (int)fminf(fmaxf(y, -65536.f), 65536.f);                                 // truncates
    const bool rare = (__builtin_amdgcn_fractf(ay) < 0x1p-12f && ay < 300.5f) || (unsigned)zp > 255u;
    if (__builtin_amdgcn_ballot_w64(rare) != 0ull) {
        if (rare) r = (int)fminf(fmaxf(roundf(__fdiv_rn(s, out_scale)), -65536.f), 65536.f);
    }
    return sat_u8(r + zp);
}

// Four values at once (one channel's four pixels / four channels of a pixel): four independent chains the scheduler can interleave, ONE
// wave-level test for the group, and a hand-over that simply evaluates the reference expression for all four (it is the definition;
// the fast form equals it wherever the test does not fire).  A group of 256 values takes it with probability ~6 %.
__device__ __forceinline__ void quant_round_sat_u8_w4(const float (&s)[4], float out_scale, float inv, int zp, int (&q)[4])
{
    bool rare = (unsigned)zp > 255u;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float y = __fmaf_rn(s[k], inv, copysignf(0.5f + 0x1p-13f, s[k]));
        const float ay = fabsf(y);
        q[k] = (int)fminf(fmaxf(y, -65536.f), 65536.f);
        rare = rare || (__builtin_amdgcn_fractf(ay) < 0x1p-12f && ay < 300.5f);
    }
    if (__builtin_amdgcn_ballot_w64(rare) != 0ull) {
#pragma unroll
        for (int k = 0; k < 4; k++) q[k] = (int)fminf(fmaxf(roundf(__fdiv_rn(s[k], out_scale)), -65536.f), 65536.f);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) q[k] = min(max(q[k] + zp, 0), 255);
}

// ---- the INTEGER path's requantisation (u8i_kernels.hip) ------------------------------------------------------------------------
// t = the exact int32 value of sum (x - zx)(w - zw) + bias.  The reference computes, on its fp32 sum s ~ t * (in_s * w_s):
// activation clamp, round(fl(s / out_s)) + zp, clamp [0, 255] (conv_kernel_x86.c:1746-1790).  The integer path is not tied to
// that expression's bits (its bar is one quantisation step), so it spends ONE multiply-add per value:
//   y = fma((float)t, M, copysign(0.5, t)),  M = fl(fl(in_s * w_s) / out_s)   (planner, binary32)
//   q = clamp(trunc(y) + zp, lo, hi)          round half away from zero, like the reference's round()
// and the conv's own activation becomes the clamp window (requantisation is monotone): relu -> lo = zp; relu6 -> also
// hi = round(fl(6 / out_s)) + zp.  Against the reference expression on the same real value the result can differ only where
// the value sits within ~1e-6 relative of a rounding boundary: never by more than one step.  tests/helpers.py u8_conv_int_model
// evaluates the same operations (the fma exactly, in binary64) -- the device is byte-exact against it.
struct U8IRq { float m; int zp, lo, hi; };
__device__ __forceinline__ int u8i_requant(int t, const U8IRq r)
{
    const float tf = (float)t;
    float y = __fmaf_rn(tf, r.m, copysignf(0.5f, tf));
    y = fminf(fmaxf(y, -512.f), 512.f);                  // beyond the byte range either way; keeps the conversion + zp inside int
    return min(max((int)y + r.zp, r.lo), r.hi);
}

// round(f / out_scale + zp), clamp -- relu_kernel_ref_uint8.c:83-89, upsample_ref.c:118-125 (zero point INSIDE the round)
__device__ __forceinline__ uint8_t quant_round_in_exact(float f, U8Q q)
{
    float r = roundf(__fdiv_rn(f, q.scale) + (float)q.zp);
    r = fminf(fmaxf(r, -65536.f), 65536.f);
    return sat_u8((int)r);
}
// The same without the division on the common path (the fused ReLU of every uint8 conv output goes through here).  The
// reference rounds x = fl(fl(f / s) + zp); y = fma(f, fl(1/s), zp) is within 2|d| 2^-24 + 2 * 2^-16 < 6.6e-5 of it for |x| < 300
// (the quotient's rounding, the sum's, the fma's), the half is added with one more rounding (2^-16): 8.1e-5 < e = 2^-13, so
// trunc(y + copysign(0.5 + e, y)) is round_half_away(x) unless its fraction is below 2e -- then the reference expression
// decides; beyond 300 the byte is saturated either way.  tests/csrc/u8_round_check.c replays it.
__device__ __forceinline__ uint8_t quant_round_in(float f, U8Q q)
{
    if ((unsigned)q.zp > 255u) return quant_round_in_exact(f, q);                     // uniform; never taken for a uint8 tensor
    const float inv = __fdiv_rn(1.0f, q.scale);
    const float y = __fmaf_rn(f, inv, (float)q.zp);
    const float y2 = y + copysignf(0.5f + 0x1p-13f, y);
    const float ay = fabsf(y2);
    if (__builtin_amdgcn_fractf(ay) < 0x1p-12f && ay < 300.5f) return quant_round_in_exact(f, q);
    return sat_u8((int)fminf(fmaxf(y2, -65536.f), 65536.f));
}
__device__ __forceinline__ float dequant(uint8_t u, float zp, float scale) { return ((float)u - zp) * scale; }

// ReLU / leaky ReLU node applied to a conv's own uint8 result: relu_kernel_ref_uint8.c:48-95 on that byte
__device__ __forceinline__ uint8_t fused_relu(uint8_t q, float scale, int zp, const U8Relu& r)
{
    float f = dequant(q, (float)zp, scale);
    if (f < 0.f) f = (r.slope == 0.f) ? 0.f : f * r.slope;
    return quant_round_in(f, r.out);
}

// The fused ReLU node and the fused pool node are functions of ONE byte (fused_relu of the conv's own byte, pooled_byte of the window
// maximum): tabulated once per block -- tail[t] = fused_relu(t) (identity without the node), tail[256 + t] = pooled_byte(t) -- and
// looked up per output: the same bytes as evaluating them per output (they ARE those evaluations), without two more IEEE
// divisions per value in every epilogue.  The integer path has done this since it exists (u8i_kernels.hip); round 4 brought the
// byte-exact kernels over together with the wave-level hand-over test of quant_round_sat_u8_w.  Callers put a barrier between
// this and the first look-up.
__device__ __forceinline__ void u8_tail_tables(uint8_t* tail, int tid, int nthreads, const U8Relu& relu, float out_scale, int out_zp, const U8PoolFuse& pool);

// Output pixel j of a conv launch -> (oy, ox).  Row-major normally; with a fused 2x2 max-pool (U8PoolFuse) window-major, so
// that lanes 4w..4w+3 of a pixel column group hold the window w = (py, px): j = 4 * (py * OW/2 + px) + 2 * dy + dx.
// The reference's main / tail split of a pixel (j < (OH*OW)&~7) is a property of its ROW-MAJOR index; the planner only
// fuses when OH*OW % 8 == 0, where every pixel is a main pixel whatever the order.
__device__ __forceinline__ void conv_pixel(const U8ConvArgs& a, int j, int* oy, int* ox)
{
    if (a.pk_tw > 0) {
        // 2-D tiles (conv_u8_patch, wide maps): tile t = j / (8 * pk_tw) in row-major tile order, inside it row-major -- or, under a
        // fused pool, window-major over the tile's (4 x pk_tw / 2) windows, so the four pixels of a window still sit in four
        // neighbouring lanes.  Only used where OH % 8 == 0 and OW % pk_tw == 0 (every tile is whole)
        const int tw = a.pk_tw, bn = 8 * tw, t = j / bn, l = j - t * bn, tx_n = a.OW / tw, ty = t / tx_n, tx = t - ty * tx_n;
        if (a.pool.on) {
            const int hw = tw >> 1, w = l >> 2, wy = w / hw, wx = w - wy * hw;
            *oy = ty * 8 + 2 * wy + ((l >> 1) & 1); *ox = tx * tw + 2 * wx + (l & 1);
        } else {
            const int r = l / tw;
            *oy = ty * 8 + r; *ox = tx * tw + (l - r * tw);
        }
        return;
    }
    if (a.pool.on) {
        const int half = a.OW >> 1, w = j >> 2, py = w / half, px = w - py * half;
        *oy = 2 * py + ((j >> 1) & 1); *ox = 2 * px + (j & 1);
    } else {
        *oy = j / a.OW; *ox = j - *oy * a.OW;
    }
}

// max over the four lanes of a quad (DPP quad_perm [1,0,3,2] then [2,3,0,1])
__device__ __forceinline__ int quad_max(int v)
{
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false));
    return max(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false));
}

// the pool node on the window maximum m: pooling_kernel_ref_uint8.c:91-200 dequantises every byte, takes the fp32 max (the
// dequantisation is monotone: that is the dequantised max byte), then round(f / out_scale) + out_zp with ONLY the upper clamp
__device__ __forceinline__ uint8_t pooled_byte(int m, const U8PoolFuse& p)
{
    const float f = ((float)(m - p.in.zp)) * p.in.scale;
    const int od = quant_round_div(f, p.out.scale, p.out.zp);
    return (uint8_t)(od > 255 ? 255 : od);
}

// The epilogue of FOUR accumulator values of one lane -- output channels co .. co+3 of one pixel (the rows of a 16x16 MFMA tile a lane
// holds) -- in phases: bias / activation, requantisation with ONE wave-level hand-over test, the fused ReLU table, the stores, the
// window maxima (the four pixels of a 2x2 window sit in four neighbouring lanes), the pool table, the pooled stores.  Value by value
// every output waited for two dependent LDS look-ups and carried its own ballot.  s[] = the finished fp32 sums (chain order is the
// caller's business); opix / ppool: the pixel's offset in its output plane / its window's offset in the pooled plane.
__device__ __forceinline__ void u8_finish4(const U8ConvArgs& a, const float (&s)[4], int co, int n, int OHW, int opix, int ppool, bool quad_lead,
                                           float rq_inv, const uint8_t* tail)
{
    float sv[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
        sv[e] = s[e];
        // the compiled reference hoists (float)bias * bias_scale out of its pixel loop and ADDS the rounded product
        // (conv_kernel_x86.c:1733-1743: vmulss, then vaddps -- not an fma)
        if (a.bias) sv[e] = sv[e] + (float)a.bias[min(co + e, a.cout - 1)] * a.bias_scale;
        if (a.act == 0) sv[e] = sv[e] < 0.f ? 0.f : sv[e];
        if (a.act > 0) { sv[e] = sv[e] < 0.f ? 0.f : sv[e]; sv[e] = sv[e] > 6.f ? 6.f : sv[e]; }
    }
    int q[4];
    quant_round_sat_u8_w4(sv, a.out_scale, rq_inv, a.out_zp, q);
    if (a.relu.on) {
#pragma unroll
        for (int e = 0; e < 4; e++) q[e] = tail[q[e]];
    }
    if (!a.pool.on || a.pool.write_full) {
#pragma unroll
        for (int e = 0; e < 4; e++)
            if (co + e < a.cout) a.y[(size_t)n * a.out_img + (size_t)(a.out_c0 + co + e) * OHW + opix] = (uint8_t)q[e];
    }
    if (a.pool.on) {                                 // the launch's pixel limit and co are uniform over a quad of lanes: all four pixels of the window are here
        int pb[4];
#pragma unroll
        for (int e = 0; e < 4; e++) pb[e] = tail[256 + quad_max(q[e])];
        if (quad_lead) {
#pragma unroll
            for (int e = 0; e < 4; e++)
                if (co + e < a.cout) a.pool.y[(size_t)n * a.pool.out_img + (size_t)(a.pool.out_c0 + co + e) * (OHW >> 2) + ppool] = (uint8_t)pb[e];
        }
    }
}

__device__ __forceinline__ void u8_tail_tables(uint8_t* tail, int tid, int nthreads, const U8Relu& relu, float out_scale, int out_zp, const U8PoolFuse& pool)
{
    if (!relu.on && !pool.on) return;
    for (int t = tid; t < 256; t += nthreads) {
        tail[t] = relu.on ? fused_relu((uint8_t)t, out_scale, out_zp, relu) : (uint8_t)t;
        if (pool.on) tail[256 + t] = pooled_byte(t, pool);
    }
}

}  // namespace tamd
