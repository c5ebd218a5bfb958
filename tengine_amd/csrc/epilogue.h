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

// ---- conv / FC requantisation: ONE fused multiply-add per value ------------------------------------------------------------
// The reference chain (header comment) is  f = clamp(fl(fl(a * m1) * m2[c]), lo, hi),  q = sat127(round_half_away(fl(f / s)))
// with a = (float)(acc + bias).  The planner folds M[c] = RN32(double(m1) * double(m2[c]) / double(s)) and the kernels evaluate
//
//     y  = fma(a, M[c], 128.5 + e)                 e = 2^-14; one rounding, the result biased into (0, 256)
//     yc = med3(y, ylo, yhi)                       ylo = 128 + q(lo) + 0.25, yhi = 128 + q(hi) + 0.75, q(.) = the reference's
//                                                  sat127(round(fl(. / s))) evaluated by the planner on the clamp bounds
//     q + 128 = trunc(yc)                          unless fract(yc) < 2e: then the reference chain itself decides (~6e-5 of values)
//
// Exactness.  Let d = fl(f / s) be the value the reference rounds (f unclamped) and |d| < 128.6.  d carries three roundings of
// the real number D = a * m1 * m2 / s, a * M one (of M) and y one more (|y| < 512, half an ulp = 2^-16):
//     |y - (d + 128.5 + e)| <= |D| * 4.001 * 2^-24 + 2^-16 < 4.6e-5 < e,      hence     0 < y - (d + 128.5) < 2e.
// An integer lies between d + 128.5 and y only if fract(y) < 2e, and a tie (d + 128.5 integral) puts y within 2e above that
// integer as well; so for every value that is not handed over, trunc(y) = floor(d + 128.5) = round_half_away(d) + 128 for both
// signs (they differ only at ties).  The clamp: R(x) = sat127(round(fl(x / s))) is monotone, so R(clamp(f, lo, hi)) =
// clamp(R(f), R(lo), R(hi)) -- which clamping y to [128 + R(lo) + 0.25, 128 + R(hi) + 0.75] implements (values beyond the
// window, including |d| >= 128.6 where the bound above no longer holds, are on the far side of the window by > 0.2); the
// window's ends have fract 0.25 / 0.75, so a clamped value is never handed over.  M is folded only when every factor is a
// normal number of moderate size (host_rq in graph_plan.hip); otherwise thr = 2 sends every value down the reference chain.
// tests/csrc/fold_requant_check.c replays fast path and chain on the host (identical IEEE operations) over random layers and
// boundary-hugging accumulators.
#define TAMD_RQ_E 0x1p-14f

typedef RqArgs Rq;

__device__ __forceinline__ float rq_biased(int acc, float mf, const Rq& r)
{
    return __builtin_amdgcn_fmed3f(__fmaf_rn((float)acc, mf, 128.5f + TAMD_RQ_E), r.ylo, r.yhi);
}

// the reference chain for one accumulator (acc already includes the int32 bias: the reference adds it before converting)
__device__ __forceinline__ int rq_chain(int acc, float m2, const Rq& r)
{
    const float f = __fmul_rn(__fmul_rn((float)acc, r.m1), m2);
    return exact_round_div_sat(__builtin_amdgcn_fmed3f(f, r.lo, r.hi), r.out_scale);
}

// c: channel of this value (index into Rq::m2 for the hand-over path)
__device__ __forceinline__ int requant1(int acc, float mf, int c, const Rq& r)
{
    const float y = rq_biased(acc, mf, r);
    int q = (int)y - 128;
    if (__builtin_amdgcn_fractf(y) < r.thr) q = rq_chain(acc, r.m2[c], r);
    return q;
}

__device__ __forceinline__ unsigned pack4(int a, int b, int c, int d)
{
    return (unsigned)(a & 0xff) | ((unsigned)(b & 0xff) << 8) | ((unsigned)(c & 0xff) << 16) | ((unsigned)(d & 0xff) << 24);
}

// the hand-over path of requant4 (a wave takes it for ~1.5 % of its groups): the reference chain for the flagged slots.  Kept
// OUT of line: inlined, its ~60 instructions at each of the 16+ call sites of a GEMM epilogue push the epilogue past the
// unroller's size budget, the accumulator arrays then get dynamic indices and land in scratch memory.  Everything by value.
__device__ __attribute__((noinline)) static unsigned requant4_chain(int a0, int a1, int a2, int a3, float4 mf, unsigned packed, const float* m2c,
                                                                   float m1, float lo, float hi, float out_scale, float ylo, float yhi, float thr)
{
    Rq r;
    r.m1 = m1; r.lo = lo; r.hi = hi; r.out_scale = out_scale; r.ylo = ylo; r.yhi = yhi; r.thr = thr; r.m2 = m2c;
    const int a[4] = {a0, a1, a2, a3};
    const float m[4] = {mf.x, mf.y, mf.z, mf.w};
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (__builtin_amdgcn_fractf(rq_biased(a[k], m[k], r)) < thr)
            packed = (packed & ~(0xffu << (8 * k))) | ((unsigned)(rq_chain(a[k], m2c[k], r) & 0xff) << (8 * k));
    return packed;
}

// The window of a node with a fused ReLU / ReLU6 (and of the residual tail with its ReLU) starts at 128.25: every clamped value lies in
// [128, 256), ONE binade, so its bit pattern is 0x43000000 | q << 16 | fract * 2^16 -- the result byte is byte 2 of the pattern and
// the hand-over test reads the low half.  Same y, same yc, same decisions as the general form (trunc(yc) - 128 == byte 2,
// fract(yc) < thr <=> low half < thr * 2^16; tests/csrc/fold_requant_check.c asserts both identities on every value it draws); what
// goes is the float -> int conversion, the shifts and the fract: 19 instead of 26 instructions per four values, 6.53 against
// 5.26 T values/s with the SIMDs full (tools/exp/requant_rates.hip, profiles/r05_requant_rates.txt).  A COMPILE-TIME choice
// (template parameter WIN; the kernels branch once, at their top or around their epilogue, on the node's constants): with both forms
// inlined behind a run-time test at every call site the kernels grew by 10-24 registers and the step got 12 % SLOWER
// (profiles/r05_ab_window_runtime_branch_*.txt).
// (`thr_of_test` = the threshold the caller's integer test stands for: 2^-13 <-> 8, the residual tail's 2^-12 <-> 16)
// The whole window must lie inside the binade: ylo >= 128 AND yhi < 256 (ylo <= yhi), both CHECKED here -- host_rq and the eltwise fold
// clamp yhi to <= 255.75 today, but a window that ever reached 256 (or a NaN bound: every comparison below is false for it) would make
// byte 2 / the low half mean something else, and the WIN instances must then not be selected.
__host__ __device__ __forceinline__ bool rq_window_is_one_binade(float ylo, float yhi, float thr, float thr_of_test)
{
    return ylo >= 128.f && yhi < 256.f && ylo <= yhi && thr == thr_of_test;
}
__host__ __device__ __forceinline__ bool rq_win(const RqArgs& r) { return rq_window_is_one_binade(r.ylo, r.yhi, r.thr, 0x1p-13f); }
__host__ __device__ __forceinline__ bool elt_win(const EltFuse& e) { return rq_window_is_one_binade(e.ylo, e.yhi, e.thr, 0x1p-12f); }

__device__ __forceinline__ unsigned rq_pack_byte2(float y0, float y1, float y2, float y3)
{
    const unsigned b0 = __builtin_bit_cast(unsigned, y0), b1 = __builtin_bit_cast(unsigned, y1);
    const unsigned b2 = __builtin_bit_cast(unsigned, y2), b3 = __builtin_bit_cast(unsigned, y3);
    // v_perm_b32: selector 0-3 = bytes of the second operand, 4-7 = bytes of the first, 0x0c = 0x00
    return __builtin_amdgcn_perm(b1, b0, 0x0c0c0602u) | __builtin_amdgcn_perm(b3, b2, 0x06020c0cu);
}

// min over the low halves of four patterns (v_min3_u16 / v_min_u16 read the low 16 bits; the upper half of the result is masked off)
__device__ __forceinline__ unsigned rq_min_low_half(float y0, float y1, float y2, float y3)
{
    unsigned m;
    asm("v_min3_u16 %0, %1, %2, %3" : "=v"(m) : "v"(y0), "v"(y1), "v"(y2));
    asm("v_min_u16 %0, %1, %2" : "=v"(m) : "v"(m), "v"(y3));
    return m & 0xffffu;
}

// four consecutive channels c .. c+3 -> one packed dword of int8; mf = M[c .. c+3].  WIN 1: the caller has checked rq_win(r)
template <int WIN = 0>
__device__ __forceinline__ unsigned requant4(int a0, int a1, int a2, int a3, const float4& mf, int c, const Rq& r)
{
    const float y0 = rq_biased(a0, mf.x, r), y1 = rq_biased(a1, mf.y, r), y2 = rq_biased(a2, mf.z, r), y3 = rq_biased(a3, mf.w, r);
    if constexpr (WIN) {
        unsigned p = rq_pack_byte2(y0, y1, y2, y3);
        if (rq_min_low_half(y0, y1, y2, y3) < 8u) p = requant4_chain(a0, a1, a2, a3, mf, p, r.m2 + c, r.m1, r.lo, r.hi, r.out_scale, r.ylo, r.yhi, r.thr);
        return p;
    }
    // fract >= 0: the order of the bit patterns is the order of the values (v_min3_u32 instead of IEEE minimum + canonicalisation)
    const unsigned f0 = __builtin_bit_cast(unsigned, __builtin_amdgcn_fractf(y0)), f1 = __builtin_bit_cast(unsigned, __builtin_amdgcn_fractf(y1));
    const unsigned f2 = __builtin_bit_cast(unsigned, __builtin_amdgcn_fractf(y2)), f3 = __builtin_bit_cast(unsigned, __builtin_amdgcn_fractf(y3));
    const unsigned fm = min(min(f0, f1), min(f2, f3));
    // yc in [1.25, 255.75]: the truncated values are bytes, biased by 128
    unsigned p = ((unsigned)y0 | ((unsigned)y1 << 8) | ((unsigned)y2 << 16) | ((unsigned)y3 << 24)) ^ 0x80808080u;
    if (fm < __builtin_bit_cast(unsigned, r.thr)) p = requant4_chain(a0, a1, a2, a3, mf, p, r.m2 + c, r.m1, r.lo, r.hi, r.out_scale, r.ylo, r.yhi, r.thr);
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

// ---- the ResNet residual tail in two fused multiply-adds per value ------------------------------------------------------------
// Reference (eltwise_ref.c:589-640 SUM, then relu_kernel_ref_int8.c:40-94 when the ReLU keeps the scale):
//     f = fl(fl(qc * s_conv) + fl(qr * s_res)),  y = sat127(round_half_away(fl(f / s))),  [y = max(y, 0)]
// Fast path on the BIASED bytes uc = qc + 128, ur = qr + 128 (v_cvt_f32_ubyteN converts a byte of a dword in one instruction):
//     t = fma(uc, Mc, K0),  yb = fma(ur, Mr, t),  Mc = RN32(s_conv / s), Mr = RN32(s_res / s), K0 = RN32(128.5 + e - 128 (Mc + Mr))
//     q + 128 = trunc(med3(yb, ylo, yhi))     unless fract < 2e, then the reference expression decides
// Error against d = fl(f / s) for |d| < 128.6, S = Mc + Mr, u = 2^-24: the reference rounds the two products, the sum and the
// quotient, <= u (127 S + 257.2); the fast path rounds Mc, Mr (<= 255 S u), K0 (|K0| < 256: 2^-17), t and yb (< 512: 2^-16
// each): |yb - (d + 128.5 + e)| <= u (382 S + 257.2) + 2^-17 + 2^-15 <= 9.9e-5 for S <= 2, below e = 2^-13 -- the planner folds
// only then (EltFuse::thr = 0 otherwise and the tail below runs).  Window and hand-over argument as for requant4.
// tests/csrc/fold_requant_check.c replays this path as well.
__device__ __attribute__((noinline)) static unsigned elt_sum4_chain(unsigned uc, unsigned ur, unsigned packed, float mc, float mr, float k0, float ylo,
                                                                   float yhi, float thr, float s_conv, float s_res, float out_scale, int relu)
{
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float fc = (float)((uc >> (8 * k)) & 0xffu), fr = (float)((ur >> (8 * k)) & 0xffu);
        const float y = __builtin_amdgcn_fmed3f(__fmaf_rn(fr, mr, __fmaf_rn(fc, mc, k0)), ylo, yhi);
        if (__builtin_amdgcn_fractf(y) < thr) {
            const float f = __fadd_rn(__fmul_rn(fc - 128.f, s_conv), __fmul_rn(fr - 128.f, s_res));      // u - 128: exact
            int q = exact_round_div_sat(f, out_scale);
            if (relu) q = q < 0 ? 0 : q;
            packed = (packed & ~(0xffu << (8 * k))) | ((unsigned)(q & 0xff) << (8 * k));
        }
    }
    return packed;
}

// pc: four int8 results of the conv, pr: the residual operand's bytes of the same channels -> the eltwise (+ReLU) result
template <int WIN = 0>      // WIN 1: the caller has checked elt_win(e)
__device__ __forceinline__ unsigned elt_sum4_fold(unsigned pc, unsigned pr, const EltFuse& e)
{
    const unsigned uc = pc ^ 0x80808080u, ur = pr ^ 0x80808080u;
    float y[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float fc = (float)((uc >> (8 * k)) & 0xffu), fr = (float)((ur >> (8 * k)) & 0xffu);
        y[k] = __builtin_amdgcn_fmed3f(__fmaf_rn(fr, e.mr, __fmaf_rn(fc, e.mc, e.k0)), e.ylo, e.yhi);
    }
    if constexpr (WIN) {          // a ReLU behind the sum: see requant4 (this tail's e is 2^-13, so the threshold is 2^-12)
        unsigned q = rq_pack_byte2(y[0], y[1], y[2], y[3]);
        if (rq_min_low_half(y[0], y[1], y[2], y[3]) < 16u)
            q = elt_sum4_chain(uc, ur, q, e.mc, e.mr, e.k0, e.ylo, e.yhi, e.thr, e.s_conv, e.s_res, e.out_scale, e.relu);
        return q;
    }
    unsigned fb[4];
#pragma unroll
    for (int k = 0; k < 4; k++) fb[k] = __builtin_bit_cast(unsigned, __builtin_amdgcn_fractf(y[k]));
    unsigned q = ((unsigned)y[0] | ((unsigned)y[1] << 8) | ((unsigned)y[2] << 16) | ((unsigned)y[3] << 24)) ^ 0x80808080u;
    if (min(min(fb[0], fb[1]), min(fb[2], fb[3])) < __builtin_bit_cast(unsigned, e.thr))
        q = elt_sum4_chain(uc, ur, q, e.mc, e.mr, e.k0, e.ylo, e.yhi, e.thr, e.s_conv, e.s_res, e.out_scale, e.relu);
    return q;
}

// p: the conv's int8 results (16 channels of one pixel), r: the residual operand's; result in p
template <int WIN = 0>
__device__ __forceinline__ void elt_sum16_fold(unsigned (&p)[4], const uint4& r, const EltFuse& e)
{
    p[0] = elt_sum4_fold<WIN>(p[0], r.x, e);
    p[1] = elt_sum4_fold<WIN>(p[1], r.y, e);
    p[2] = elt_sum4_fold<WIN>(p[2], r.z, e);
    p[3] = elt_sum4_fold<WIN>(p[3], r.w, e);
}

// the general tail (any eltwise type / ReLU flavour) out of line, for epilogues whose hot path is elt_sum4_fold; scalars only
// (a struct argument travels through scratch memory)
__device__ __attribute__((noinline)) static unsigned fuse_elt4_cold(unsigned pc, unsigned pr, int type, int conv_is_first, float s_conv, float s_res,
                                                                   float out_scale, int relu, float relu_out_scale, float inv_out, float inv_relu)
{
    EltFuse e;
    e.type = type; e.conv_is_first = conv_is_first; e.s_conv = s_conv; e.s_res = s_res; e.out_scale = out_scale; e.relu = relu;
    e.relu_out_scale = relu_out_scale;
    return fuse_elt4(pc, pr, e, inv_out, inv_relu);
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
