// The tail pixels of the uint8 patch-family kernels (conv_u8_patch, conv_u8_pw, conv_u8_c3): shared device code of u8_conv_patch.hip and
// u8_conv_small.hip (one file, u8_kernels.hip, until round 5).
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "u8_epilogue.h"

namespace tamd {

// The tail pixels (j >= (OH*OW)&~7) of the patch kernel's layers, on the VALU in the same launch: a lane owns one of the
// reference's four k%4 chains of one output -- lane (row l15, chain r = lane/16) of a wave walks k = r, r+4, r+8, .. with fmaf
// (what the MFMA does inside its step, conv_u8_body's header), reading its weights from the SAME fragment stream a main wave
// fetches (lane (l15, r) of the MFMA A operand holds exactly row l15, k%4 = r) and the dequantised im2col column of its pixel
// from LDS, class-major; the four chains meet in lanes 0..15 and are combined as the reference combines them
// (conv_u8_body: rows inside an 8-/4-row block ((0+(s0+s1))+(s2+s3)), the last cout%4 rows ((s0+s1)+s2)+s3).  K%4 == 0 here
// (the patch kernel takes whole super-steps only), so there is no scalar remainder.  A block = (image, tail pixel, 64 channels).
template <int KHW>
__device__ __forceinline__ void conv_u8_patch_tail(const U8ConvArgs& a, float* xs, int tb, const uint8_t* tail)
{
    const float rq_inv = __fdiv_rn(1.0f, a.out_scale);
    constexpr int NTAPS = KHW * KHW, SS = KHW == 3 ? 9 : 4, G4 = SS / 4, REM = SS - 4 * G4, FRAG = SS * 64, RING = 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, r = lane >> 4;
    const int OHW = a.OH * a.OW, N8 = OHW & ~7, T = OHW - N8, slices = (a.cout + 63) / 64;
    const int slice = tb % slices, nt = tb / slices, t = nt % T, n = nt / T;
    const int opix = N8 + t, oy = opix / a.OW, ox = opix - oy * a.OW;          // no fused pool on a layer with tail pixels
    const int K4 = a.K >> 2, chw = a.H * a.W;
    const uint8_t* xin = a.x + (size_t)n * a.C * chw;
    for (int k = tid; k < a.K; k += 256) {
        const int c = k / NTAPS, tap = k - c * NTAPS, ky = tap / KHW, kx = tap - ky * KHW;
        const int iy = oy * a.SH - a.PH + ky * a.pk_dh, ix = ox * a.SW - a.PW + kx * a.pk_dw;
        float v = 0.f;
        if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) v = dequant(xin[(size_t)c * chw + iy * a.W + ix], a.in_zp, a.in_scale);
        xs[(k & 3) * K4 + (k >> 2)] = v;
    }
    __syncthreads();
    const int tile16 = slice * 4 + wave;
    if (tile16 * 16 >= a.cout) return;
    const int nss = a.K / (4 * SS);
    const float* wb = reinterpret_cast<const float*>(a.wpk) + (size_t)tile16 * nss * FRAG;
    const float* xr = xs + r * K4;
    float4 w4[RING][G4];
    float wr[RING][REM > 0 ? REM : 1];
    auto wload = [&](auto D, int ss) {
        constexpr int d = decltype(D)::value;
        const size_t o = (size_t)(ss < nss ? ss : nss - 1) * FRAG;
#pragma unroll
        for (int v = 0; v < G4; v++) w4[d][v] = *reinterpret_cast<const float4*>(wb + o + v * 256 + lane * 4);
#pragma unroll
        for (int v = 0; v < REM; v++) wr[d][v] = wb[o + G4 * 256 + lane * REM + v];
    };
    float acc = 0.f;
    auto sstep = [&](auto D, int ss) {
        constexpr int d = decltype(D)::value;
        wload(std::integral_constant<int, (d + RING - 1) % RING>{}, ss + RING - 1);
        const float* xp = xr + ss * SS;
#pragma unroll
        for (int v = 0; v < G4; v++) {
            acc = __builtin_fmaf(w4[d][v].x, xp[4 * v], acc);
            acc = __builtin_fmaf(w4[d][v].y, xp[4 * v + 1], acc);
            acc = __builtin_fmaf(w4[d][v].z, xp[4 * v + 2], acc);
            acc = __builtin_fmaf(w4[d][v].w, xp[4 * v + 3], acc);
        }
#pragma unroll
        for (int v = 0; v < REM; v++) acc = __builtin_fmaf(wr[d][v], xp[4 * G4 + v], acc);
    };
    wload(std::integral_constant<int, 0>{}, 0);
    wload(std::integral_constant<int, 1>{}, 1);
    wload(std::integral_constant<int, 2>{}, 2);
    for (int ss = 0; ss < nss; ss += RING) {
        sstep(std::integral_constant<int, 0>{}, ss);
        if (ss + 1 < nss) sstep(std::integral_constant<int, 1>{}, ss + 1);
        if (ss + 2 < nss) sstep(std::integral_constant<int, 2>{}, ss + 2);
        if (ss + 3 < nss) sstep(std::integral_constant<int, 3>{}, ss + 3);
    }
    const float s1 = __shfl(acc, l15 + 16), s2 = __shfl(acc, l15 + 32), s3 = __shfl(acc, l15 + 48);
    const int co = tile16 * 16 + l15;
    if (r != 0 || co >= a.cout) return;
    float s = co < a.m_blocked ? (0.f + (acc + s1)) + (s2 + s3) : ((acc + s1) + s2) + s3;
    if (a.bias) s = s + (float)a.bias[co] * a.bias_scale;
    if (a.act == 0) s = s < 0.f ? 0.f : s;
    if (a.act > 0) { s = s < 0.f ? 0.f : s; s = s > 6.f ? 6.f : s; }
    uint8_t q = quant_round_sat_u8_w(s, a.out_scale, rq_inv, a.out_zp);
    if (a.relu.on) q = tail[q];
    a.y[(size_t)n * a.out_img + (size_t)(a.out_c0 + co) * OHW + opix] = q;
}

}  // namespace tamd
