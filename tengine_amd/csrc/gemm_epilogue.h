// Fused epilogue of the MFMA GEMM kernels (conv_igemm.hip, conv_igemm2.hip).
#pragma once
#include "epilogue.h"

namespace tamd {

typedef int v16i_t __attribute__((ext_vector_type(16)));

// ---- fused epilogue shared by the 32x32-tile GEMM kernels: +bias, requantise (bit-exact, epilogue.h), pack 4 channels,
// NHWC store.  acc[i][j]: 32x32 tile (cout tile i, pixel tile j) of wave (wm, wn) of the block tile at (m0, n0).
// where the per-channel vectors of the epilogue come from: global memory (a cold round trip at the start of every block's
// epilogue -- these 8 bytes per channel are touched once per layer) or an LDS copy the kernel fetched while it set itself up
struct EpiFromGlobal {
    const int32_t* b; const float* s;
    __device__ __forceinline__ int4 bias4(int c) const { return *reinterpret_cast<const int4*>(b + c); }
    __device__ __forceinline__ float4 scale4(int c) const { return *reinterpret_cast<const float4*>(s + c); }
};
// the same two sources for kernels whose accumulators START at the bias (igemm_acc_from_bias below: integer addition commutes, the
// reference's acc + bias is the same 32-bit sum) -- the epilogue then costs one VALU instruction less per value and, from global
// memory, fetches one vector instead of two
struct EpiScaleFromGlobal {
    const float* s;
    __device__ __forceinline__ int4 bias4(int) const { return make_int4(0, 0, 0, 0); }
    __device__ __forceinline__ float4 scale4(int c) const { return *reinterpret_cast<const float4*>(s + c); }
};
struct EpiFromLdsNoBias {         // multipliers of the block's cout tile from LDS; the bias already sits in the accumulators
    const int8_t* base; int n0, bn;
    __device__ __forceinline__ int4 bias4(int) const { return make_int4(0, 0, 0, 0); }
    __device__ __forceinline__ float4 scale4(int c) const { return *reinterpret_cast<const float4*>(base + (bn + c - n0) * 4); }
};
struct EpiFromLds {               // [BN ints of bias][BN floats of multipliers] of the block's cout tile, at `base`
    const int8_t* base; int n0, bn;
    __device__ __forceinline__ int4 bias4(int c) const { return *reinterpret_cast<const int4*>(base + (c - n0) * 4); }
    __device__ __forceinline__ float4 scale4(int c) const { return *reinterpret_cast<const float4*>(base + (bn + c - n0) * 4); }
};

// accumulators of wave (., wn) of the block tile at cout n0 <- the bias of their channels (C/D layout of the 32x32 MFMA: register e of
// lane (pixel, hi) holds channel 8 (e >> 2) + 4 hi + (e & 3) of its 32-channel tile); `bias` is padded to whole cout tiles
template <int TM, int TN, typename V>
__device__ __forceinline__ void igemm_acc_from_bias(V (&acc)[TN][TM], const int32_t* bias, int n0, int wn, int hi)
{
#pragma unroll
    for (int i = 0; i < TN; i++)
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) {
            const int4 b4 = *reinterpret_cast<const int4*>(bias + n0 + (wn * TN + i) * 32 + 8 * g4 + 4 * hi);
#pragma unroll
            for (int j = 0; j < TM; j++) { acc[i][j][4 * g4 + 0] = b4.x; acc[i][j][4 * g4 + 1] = b4.y; acc[i][j][4 * g4 + 2] = b4.z; acc[i][j][4 * g4 + 3] = b4.w; }
        }
}

// FORM 0: everything (any destination granularity, any fused eltwise tail, the general requantisation).  FORM 1 / 2: the one-binade
// requantisation of epilogue.h for the two common nodes, as SMALL instances (16-channel-granular destination only): 1 = a conv with
// a fused ReLU and no eltwise tail, its own window in the one-binade form; 2 = a conv (general form) + the folded SUM tail whose
// ReLU puts the tail's window there.  igemm_epilogue_src checks the node's constants and picks the instance.
template <int TM, int TN, int FORM, typename Src>
__device__ __forceinline__ void igemm_epilogue_form(const ConvArgs& a, v16i_t (&acc)[TN][TM], int m0, int n0, int wm, int wn, int l31, int hi, const Src src)
{
    // C/D layout of 32x32 MFMA: col = lane&31 (pixel), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (cout)
    const Rq rq = a.rq;
    // 16-B stores (half-wave regroup) whenever the destination is 16-channel granular; dword stores else
    const bool wide = FORM != 0 || (((a.c_limit | a.c_off | a.ldc) & 15) == 0 && (!a.elt.res || ((a.elt.res_ldc | a.elt.res_c_off) & 15) == 0));
    const float inv_elt = (FORM == 0 && a.elt.res) ? __fdiv_rn(1.0f, a.elt.out_scale) : 1.f;
    const float inv_relu = (FORM == 0 && a.elt.res && a.elt.relu) ? __fdiv_rn(1.0f, a.elt.relu_out_scale) : 1.f;
    // every bias / scale vector of the wave's cout tiles is requested before the first requantisation: loaded one group at a
    // time inside the loops they cost a memory round trip each (measured: 1.1 us of epilogue on a 64x64 tile,
    // profiles/r02_igemm_anatomy_*)
    int4 b4s[TN][4];
    float4 s4s[TN][4];
#pragma unroll
    for (int i = 0; i < TN; i++)
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) {
            const int c = n0 + (wn * TN + i) * 32 + 8 * g4 + 4 * hi;
            b4s[i][g4] = src.bias4(c);
            s4s[i][g4] = src.scale4(c);
        }
    static_for<0, TN>([&](auto I) {
        constexpr int i = decltype(I)::value;
        const int cb = n0 + (wn * TN + i) * 32;
        static_for<0, TM>([&](auto J) {
            constexpr int j = decltype(J)::value;
            unsigned p[4];
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) {
                const int4 b4 = b4s[i][g4];
                const float4 s4 = s4s[i][g4];
                p[g4] = requant4<FORM == 1>(acc[i][j][4 * g4 + 0] + b4.x, acc[i][j][4 * g4 + 1] + b4.y, acc[i][j][4 * g4 + 2] + b4.z,
                                            acc[i][j][4 * g4 + 3] + b4.w, s4, cb + 8 * g4 + 4 * hi, rq);
            }
            const int m = m0 + (wm * TM + j) * 32 + l31;
            if (wide) {
                half_wave_regroup(p);
                const int c16 = cb + hi * 16;
                if (m < a.M && c16 < a.c_limit) {
                    if (FORM == 2 || (FORM == 0 && a.elt.res)) {      // eltwise (+ReLU) tail on the 16 channels this lane now holds
                        const uint4 r = *reinterpret_cast<const uint4*>(a.elt.res + (size_t)m * a.elt.res_ldc + a.elt.res_c_off + c16);
                        if (FORM == 2) {
                            elt_sum16_fold<1>(p, r, a.elt);
                        } else if (a.elt.thr > 0.f) {
                            elt_sum16_fold<0>(p, r, a.elt);
                        } else {
                            const uint4 o = fuse_elt16(make_uint4(p[0], p[1], p[2], p[3]), r, a.elt, inv_elt, inv_relu);
                            p[0] = o.x; p[1] = o.y; p[2] = o.z; p[3] = o.w;
                        }
                    }
                    *reinterpret_cast<uint4*>(a.y + (size_t)m * a.ldc + a.c_off + c16) = make_uint4(p[0], p[1], p[2], p[3]);
                }
            } else if constexpr (FORM == 0) {
#pragma unroll
                for (int g4 = 0; g4 < 4; g4++) {
                    const int c0 = cb + 8 * g4 + 4 * hi;
                    if (m < a.M && c0 < a.c_limit) {
                        unsigned v = p[g4];
                        if (a.elt.res)
                            v = fuse_elt4(v, *reinterpret_cast<const unsigned*>(a.elt.res + (size_t)m * a.elt.res_ldc + a.elt.res_c_off + c0),
                                          a.elt, inv_elt, inv_relu);
                        *reinterpret_cast<unsigned*>(a.y + (size_t)m * a.ldc + a.c_off + c0) = v;
                    }
                }
            }
        });
    });
}

// the epilogue is the tail of its kernel: ONE uniform branch on the node's constants picks the instance
template <int TM, int TN, typename Src>
__device__ __forceinline__ void igemm_epilogue_src(const ConvArgs& a, v16i_t (&acc)[TN][TM], int m0, int n0, int wm, int wn, int l31, int hi, const Src src)
{
    const bool wide = ((a.c_limit | a.c_off | a.ldc) & 15) == 0 && (!a.elt.res || ((a.elt.res_ldc | a.elt.res_c_off) & 15) == 0);
    const int form = !wide ? 0 : a.elt.res ? ((a.elt.thr > 0.f && elt_win(a.elt)) ? 2 : 0) : (rq_win(a.rq) ? 1 : 0);
    if (form == 1) igemm_epilogue_form<TM, TN, 1>(a, acc, m0, n0, wm, wn, l31, hi, src);
    else if (form == 2) igemm_epilogue_form<TM, TN, 2>(a, acc, m0, n0, wm, wn, l31, hi, src);
    else igemm_epilogue_form<TM, TN, 0>(a, acc, m0, n0, wm, wn, l31, hi, src);
}

template <int TM, int TN>
__device__ __forceinline__ void igemm_epilogue(const ConvArgs& a, v16i_t (&acc)[TN][TM], int m0, int n0, int wm, int wn, int l31, int hi)
{
    igemm_epilogue_src<TM, TN>(a, acc, m0, n0, wm, wn, l31, hi, EpiScaleFromGlobal{a.wscale});      // (accumulators from igemm_acc_from_bias)
}

}  // namespace tamd
