// Bit-exact requantising epilogues shared by every int8 kernel (the parity spec).
// Every float operation below is a single correctly-rounded binary32 op in the reference's written
// order; the translation unit is compiled with -ffp-contract=off and without fast-math, rounding is
// C round() == half away from zero.
//
// The reference has three conv/fc formulas (A1 x86 "hcl", A2 naive ref, A5 fc).  The planner folds them
// into ONE branch-free device formula by choosing (m1, m2[c], lo, hi, out_scale):
//
//     f = clamp( ((float)(acc+bias) * m1) * m2[c], lo, hi ) ;  q = sat127( round( f / out_scale ) )
//
//   A1  conv_kernel_x86.c:1826-1889 == conv_dw_hcl_x86.c:197-261,373-436
//       f=(float)acc*in_scale*w_scale[c]                       -> m1=in_scale, m2[c]=w_scale[c]
//       act==0: relu -> [0,+inf) ; act>0 (ANY positive code): [0,6] ; act<0: none
//   A2  conv_kernel_ref_int8.c:72-78,137-167
//       f=(float)acc*(in_scale*w_scale[c])                      -> m1=1 (exact), m2[c]=fl(in_scale*w_scale[c])
//       act==1: [-1,1] ; act==6: [0,6] ; other act>=0: [0,+inf) ; act<0: none
//   A5  fc_ref.c:224-225,252-257   q=roundf((float)acc * r[c]), r[c]=fl(fl(in_scale*w_scale[c])/out_scale)
//                                                               -> m1=1, m2[c]=r[c], no clamp, out_scale=1 (x/1 exact)
// (multiplying by 1.0f and dividing by 1.0f are exact, max/min against +-FLT_MAX are identities.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "kernels.h"

namespace tamd {

__device__ __forceinline__ int sat127(int v) { return v > 127 ? 127 : (v < -127 ? -127 : v); }

// round-half-away-from-zero of a binary32 value, then int conversion + clamp [-127,127]
__device__ __forceinline__ int round_sat(float q) { return sat127((int)roundf(q)); }

// the reference expression itself: it runs for ~1e-4 of the values only
__device__ __forceinline__ int exact_round_div_sat(float f, float s) { return round_sat(__fdiv_rn(f, s)); }

// sat127(round(f / s)) -- the reference's `(int)round(f / out_scale)` + clamp -- WITHOUT the ~20-instruction
// IEEE division on the common path, and still exact.  d = fl(f/s) is what the reference rounds (half away
// from zero).  With inv = fl(1/s), c = 0.5 + eps (eps = 2^-14) and ONE fused op
//     y = fma(f, inv, copysign(c, f))
// we have, for |f/s| < 128,  | (|y| - c) - |d| | <= |f/s| * 2^-23 + 2^-17 < eps/2, hence
// |y| in (|d| + 0.5 + eps/2, |d| + 0.5 + 3eps/2): an integer can lie between |d|+0.5 and |y| only if
// fract(|y|) < 3eps/2.  So whenever fract(|y|) >= 2*eps,  trunc(y) == sign(d) * floor(|d| + 0.5), the reference
// result; the remaining ~1.2e-4 of the values take the exact division (wave-uniformly skipped when no lane
// needs it).  tests/csrc/fast_requant_check.c replays this on the host against the reference expression.
#define TAMD_RQ_EPS 0x1p-14f

// general form (f unbounded): used by pooling / eltwise / relu
__device__ __forceinline__ int round_div_sat(float f, float s, float inv)
{
    const float y = __fmaf_rn(f, inv, copysignf(0.5f + TAMD_RQ_EPS, f));
    int q = sat127((int)y);
    const float ay = fabsf(y);
    if (__builtin_amdgcn_fractf(ay) < 2.f * TAMD_RQ_EPS && ay < 129.f) q = exact_round_div_sat(f, s);
    return q;
}

struct Rq {            // per-launch requantisation constants (see the header comment)
    float m1, lo, hi, out_scale, inv_out;
};

// lo/hi additionally fold the +-127 saturation: any f beyond +-127.49*out_scale rounds to +-127 either way,
// so clamping f there keeps |y| < 128 and the int result needs no further clamp.
__device__ __forceinline__ Rq make_rq(float m1, float lo, float hi, float out_scale)
{
    Rq r;
    const float lim = __fmul_rn(127.49f, out_scale);
    r.m1 = m1; r.out_scale = out_scale;
    r.lo = fmaxf(lo, -lim);
    r.hi = fminf(hi, lim);
    r.inv_out = __fdiv_rn(1.0f, out_scale);
    return r;
}

__device__ __forceinline__ float rq_value(int acc, float m2, const Rq& r)
{
    float f = __fmul_rn(__fmul_rn((float)acc, r.m1), m2);
    return __builtin_amdgcn_fmed3f(f, r.lo, r.hi);          // activation clamp + saturation in one op
}

// fast path on a pre-clamped f: 4 VALU (bfi, fma, fract, cvt) + the risky compare
__device__ __forceinline__ int rq_round_fast(float f, const Rq& r, bool& risky)
{
    const float y = __fmaf_rn(f, r.inv_out, copysignf(0.5f + TAMD_RQ_EPS, f));
    risky = __builtin_amdgcn_fractf(fabsf(y)) < 2.f * TAMD_RQ_EPS;
    return (int)y;
}

// acc already includes the int32 bias (the reference adds bias in int32 before converting)
__device__ __forceinline__ int requant1(int acc, float m2, const Rq& r)
{
    const float f = rq_value(acc, m2, r);
    bool risky;
    int q = rq_round_fast(f, r, risky);
    if (risky) q = exact_round_div_sat(f, r.out_scale);
    return q;
}

__device__ __forceinline__ unsigned pack4(int a, int b, int c, int d)
{
    return (unsigned)(a & 0xff) | ((unsigned)(b & 0xff) << 8) | ((unsigned)(c & 0xff) << 16) | ((unsigned)(d & 0xff) << 24);
}

// the rare path of requant4 (~1e-4 of the values): the reference expression itself for the flagged slots.  Kept OUT of line:
// inlined, its ~60 instructions at each of the 16+ call sites of a GEMM epilogue push the epilogue past the unroller's size
// budget, the accumulator arrays then get dynamic indices and land in scratch memory (seen: 320 B of scratch in every
// 128x128 tile variant of conv_igemm.hip)
__device__ __attribute__((noinline)) static unsigned requant4_exact(float f0, float f1, float f2, float f3, unsigned packed, int flags, float out_scale)
{
    int q[4] = {(int)(signed char)(packed & 0xff), (int)(signed char)((packed >> 8) & 0xff), (int)(signed char)((packed >> 16) & 0xff),
                (int)(signed char)(packed >> 24)};
    if (flags & 1) q[0] = exact_round_div_sat(f0, out_scale);
    if (flags & 2) q[1] = exact_round_div_sat(f1, out_scale);
    if (flags & 4) q[2] = exact_round_div_sat(f2, out_scale);
    if (flags & 8) q[3] = exact_round_div_sat(f3, out_scale);
    return pack4(q[0], q[1], q[2], q[3]);
}

// four consecutive channels -> one packed dword; the exact path is taken once for the group
__device__ __forceinline__ unsigned requant4(int a0, int a1, int a2, int a3, const float4& m2, const Rq& r)
{
    const float f0 = rq_value(a0, m2.x, r), f1 = rq_value(a1, m2.y, r), f2 = rq_value(a2, m2.z, r), f3 = rq_value(a3, m2.w, r);
    bool k0, k1, k2, k3;
    const int q0 = rq_round_fast(f0, r, k0);
    const int q1 = rq_round_fast(f1, r, k1);
    const int q2 = rq_round_fast(f2, r, k2);
    const int q3 = rq_round_fast(f3, r, k3);
    unsigned p = pack4(q0, q1, q2, q3);
    if (k0 | k1 | k2 | k3) p = requant4_exact(f0, f1, f2, f3, p, (int)k0 | ((int)k1 << 1) | ((int)k2 << 2) | ((int)k3 << 3), r.out_scale);
    return p;
}

// 32x32 MFMA C/D layout puts channels 8g + 4hi + {0..3} of one pixel in packed dword p[g] of lane
// (pixel, hi).  After this exchange lanes 0-31 hold channels [0,16) of the 32-channel tile in p[0..3] and
// lanes 32-63 hold [16,32): one 16-B store per lane instead of four 4-B stores into four 64-B segments.
// v_permlane32_swap: lanes 32-63 of vdst <-> lanes 0-31 of src.
__device__ __forceinline__ void half_wave_regroup(unsigned (&p)[4])
{
    auto sw = [](unsigned& vdst, unsigned& src) {
        auto r = __builtin_amdgcn_permlane32_swap(vdst, src, false, false);
        vdst = r[0];
        src = r[1];
    };
    sw(p[0], p[1]);      // lower: p0=0-3   p1=4-7   | upper: p0=8-11  p1=12-15
    sw(p[2], p[3]);      // lower: p2=16-19 p3=20-23 | upper: p2=24-27 p3=28-31
    sw(p[0], p[2]);      // lower: p2=8-11           | upper: p0=16-19
    sw(p[1], p[3]);      // lower: p3=12-15          | upper: p1=20-23
}

// ---- conv + eltwise (+ ReLU) fused in the conv epilogue --------------------------------------------
// The conv result is still rounded to int8 exactly as the stand-alone conv would store it (q_c); the eltwise node
// (eltwise_ref.c:589-640,833-837: f = op(q_a*s_a, q_b*s_b), y = sat(round(f/out_s))) and the optional ReLU node
// (relu_kernel_ref_int8.c:40-94 on y) are then applied to q_c and the residual byte in registers -- the same float
// operations on the same int8 values, so the bytes cannot differ from the three-launch sequence.
// (struct EltFuse lives in kernels.h next to ConvArgs)
__device__ __forceinline__ int sx8(unsigned v, int b) { return (int)(v << (24 - 8 * b)) >> 24; }

__device__ __forceinline__ unsigned fuse_elt4(unsigned pc, unsigned pr, const EltFuse& e, float inv_out, float inv_relu)
{
    int q[4];
#pragma unroll
    for (int b = 0; b < 4; b++) {
        const float fc = __fmul_rn((float)sx8(pc, b), e.s_conv), fr = __fmul_rn((float)sx8(pr, b), e.s_res);
        const float fa = e.conv_is_first ? fc : fr, fb = e.conv_is_first ? fr : fc;
        float f;
        switch (e.type) {
        case 0: f = __fmul_rn(fa, fb); break;
        case 2: f = __fadd_rn(fa, fb); break;
        case 4: f = __fsub_rn(fa, fb); break;
        default: f = fa > fb ? fa : fb; break;
        }
        int y = round_div_sat(f, e.out_scale, inv_out);
        if (e.relu == 2) {
            // ReLU whose output scale IS the eltwise output scale (the quantiser shares them): for an integer |y| <= 127,
            // fl(fl(y*s)/s) is within 2^-22 relative of y, so round() gives y back exactly and the node is max(y, 0)
            y = y < 0 ? 0 : y;
        } else if (e.relu) {
            float f2 = __fmul_rn((float)y, e.out_scale);
            f2 = f2 < 0.f ? 0.f : f2;
            y = round_div_sat(f2, e.relu_out_scale, inv_relu);
        }
        q[b] = y;
    }
    return pack4(q[0], q[1], q[2], q[3]);
}

// 16 channels at once for the GEMM epilogues that hold them after half_wave_regroup.  Out of line (one call per 32x32 tile and
// lane instead of sixteen inlined copies of the tail: the epilogue stays small enough for the unroller, see requant4_exact),
// with the uniform choices -- eltwise type, operand order, ReLU flavour -- hoisted out of the per-value work.
template <int TYPE, bool CONV_FIRST, int RELU>
__device__ __forceinline__ unsigned fuse_elt4_t(unsigned pc, unsigned pr, const EltFuse& e, float inv_out, float inv_relu)
{
    int q[4];
#pragma unroll
    for (int b = 0; b < 4; b++) {
        const float fc = __fmul_rn((float)sx8(pc, b), e.s_conv), fr = __fmul_rn((float)sx8(pr, b), e.s_res);
        const float fa = CONV_FIRST ? fc : fr, fb = CONV_FIRST ? fr : fc;
        const float f = TYPE == 0 ? __fmul_rn(fa, fb) : TYPE == 2 ? __fadd_rn(fa, fb) : TYPE == 4 ? __fsub_rn(fa, fb) : (fa > fb ? fa : fb);
        int y = round_div_sat(f, e.out_scale, inv_out);
        if (RELU == 2) y = y < 0 ? 0 : y;                     // see fuse_elt4
        else if (RELU == 1) {
            float f2 = __fmul_rn((float)y, e.out_scale);
            f2 = f2 < 0.f ? 0.f : f2;
            y = round_div_sat(f2, e.relu_out_scale, inv_relu);
        }
        q[b] = y;
    }
    return pack4(q[0], q[1], q[2], q[3]);
}

template <int TYPE, bool CONV_FIRST, int RELU>
__device__ __forceinline__ void fuse_elt16_t(unsigned (&p)[4], const uint4& r, const EltFuse& e, float inv_out, float inv_relu)
{
    p[0] = fuse_elt4_t<TYPE, CONV_FIRST, RELU>(p[0], r.x, e, inv_out, inv_relu);
    p[1] = fuse_elt4_t<TYPE, CONV_FIRST, RELU>(p[1], r.y, e, inv_out, inv_relu);
    p[2] = fuse_elt4_t<TYPE, CONV_FIRST, RELU>(p[2], r.z, e, inv_out, inv_relu);
    p[3] = fuse_elt4_t<TYPE, CONV_FIRST, RELU>(p[3], r.w, e, inv_out, inv_relu);
}

// The ResNet residual tail -- SUM, optionally followed by a ReLU that keeps the eltwise output scale -- on 16 channels with
// packed fp32 (v_pk_mul / v_pk_add / v_pk_fma process two values per instruction, IEEE per element like their scalar
// forms) and the same division-free rounding as requant4:
//   f = fl(fl(qc*s_conv) + fl(qr*s_res))   (eltwise_ref.c:589-640)   ->  clamp to [-lim | 0, lim], lim = fl(127.49*s)
//   y = fma(f, fl(1/s), copysign(0.5+eps, f)) ; trunc(y) is the reference's sat127(round(f/s)) unless fract(|y|) < 2 eps,
//   then the exact division decides (round_div_sat's argument; clamping f first changes nothing: beyond +-lim the reference
//   saturates to +-127 too, and below 0 a following ReLU maps every result to 0, which is what round(0/s) gives).
typedef float v2f_t __attribute__((ext_vector_type(2)));

template <bool RELU>
__device__ __forceinline__ void elt_sum16(unsigned (&p)[4], const uint4& r, const EltFuse& e, float inv_out)
{
    const float lim = __fmul_rn(127.49f, e.out_scale);
    const float lo = RELU ? 0.f : -lim;
    const unsigned rr[4] = {r.x, r.y, r.z, r.w};
    const v2f_t sc = {e.s_conv, e.s_conv}, sr = {e.s_res, e.s_res}, inv2 = {inv_out, inv_out};
#pragma unroll
    for (int d = 0; d < 4; d++) {
        float f[4], fr[4];
        int q[4];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const v2f_t c = {(float)sx8(p[d], 2 * h), (float)sx8(p[d], 2 * h + 1)};
            const v2f_t v = {(float)sx8(rr[d], 2 * h), (float)sx8(rr[d], 2 * h + 1)};
            v2f_t s = c * sc + v * sr;                      // -ffp-contract=off: both products rounded, then the sum
            s.x = __builtin_amdgcn_fmed3f(s.x, lo, lim);
            s.y = __builtin_amdgcn_fmed3f(s.y, lo, lim);
            const v2f_t half = {RELU ? 0.5f + TAMD_RQ_EPS : copysignf(0.5f + TAMD_RQ_EPS, s.x),
                                RELU ? 0.5f + TAMD_RQ_EPS : copysignf(0.5f + TAMD_RQ_EPS, s.y)};
            const v2f_t y = __builtin_elementwise_fma(s, inv2, half);
            f[2 * h] = s.x; f[2 * h + 1] = s.y;
            q[2 * h] = (int)y.x; q[2 * h + 1] = (int)y.y;
            fr[2 * h] = __builtin_amdgcn_fractf(fabsf(y.x)); fr[2 * h + 1] = __builtin_amdgcn_fractf(fabsf(y.y));
        }
        // one branch per four values (a wave takes it for ~3% of them), as requant4 does
        if (fminf(fminf(fr[0], fr[1]), fminf(fr[2], fr[3])) < 2.f * TAMD_RQ_EPS) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int qe = exact_round_div_sat(f[k], e.out_scale);
                q[k] = fr[k] < 2.f * TAMD_RQ_EPS ? qe : q[k];
            }
        }
        p[d] = pack4(q[0], q[1], q[2], q[3]);
    }
}

// everything by VALUE: a reference into the kernel's argument block would force the whole block into scratch memory
__device__ __attribute__((noinline)) static uint4 fuse_elt16(uint4 pv, uint4 r, EltFuse e, float inv_out, float inv_relu)
{
    unsigned p[4] = {pv.x, pv.y, pv.z, pv.w};
    // the ResNet case first: sum + ReLU whose output scale is the eltwise output scale (sum is commutative)
    if (e.type == 2 && e.relu == 2) elt_sum16<true>(p, r, e, inv_out);
    else if (e.type == 2 && e.relu) fuse_elt16_t<2, true, 1>(p, r, e, inv_out, inv_relu);
    else if (e.type == 2) elt_sum16<false>(p, r, e, inv_out);
    else {
        p[0] = fuse_elt4(p[0], r.x, e, inv_out, inv_relu);
        p[1] = fuse_elt4(p[1], r.y, e, inv_out, inv_relu);
        p[2] = fuse_elt4(p[2], r.z, e, inv_out, inv_relu);
        p[3] = fuse_elt4(p[3], r.w, e, inv_out, inv_relu);
    }
    return make_uint4(p[0], p[1], p[2], p[3]);
}

// compile-time loop: indices that MUST be constants (accumulator arrays: a dynamic index sends the array to scratch memory)
// stay constants whether or not the optimizer chooses to unroll
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

}  // namespace tamd
