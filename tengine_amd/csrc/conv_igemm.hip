// int8 implicit-GEMM convolution on MFMA (gfx950): 1x1 and kxk, any stride / pad / dilation, group 1.
//
// Replaces the reference's im2col_int8 + input_pack4_int8 + sgemm_i8 + sgemm_int8 epilogue chain
// (source/device/cpu/op/conv/x86/conv_kernel_x86.c:187-242, :963-1007, :1008-1630, :1796-1893) with
// one launch: no im2col matrix is materialised, the requantising epilogue is fused.
//
// GEMM view:  D[cout][pixel] = sum_k Wp[cout][k] * X[pixel][k],  k = (ky*KW+kx)*ckp + ci.
//   * MFMA A operand = weights (rows -> cout), B operand = activations (cols -> pixel), so each lane's
//     accumulator registers hold 4 *consecutive output channels* of one pixel -> packed dword stores
//     straight into NHWC.
//   * v_mfma_i32_32x32x32_i8 (gfx950 double-K form of 32x32x16): lane l supplies 16 K-contiguous
//     bytes of row/col (l&31) at k-offset (l>>5)*16.  Both operands are K-contiguous in memory
//     (NHWC activations, [cout][k] weights) so operand loads are plain 16-B vectors.
//   * tiles staged through LDS with an 80/144-B padded row (conflict-free ds_read_b128, see
//     MI355X_MICROARCH.md §LDS: 16-lane groups, 64 banks), register-prefetched double buffer,
//     one barrier per K step.
//   * blockIdx -> tile map is XCD-aware: blocks b, b+8, b+16.. run on one XCD (private L2) and are
//     given consecutive cout-tiles of the same pixel-tile, so the activation tile is fetched from
//     HBM once per XCD and re-read from that XCD's L2.
#include <stdlib.h>

#include "epilogue.h"
#include "kernels.h"

namespace tamd {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int BM, int BN, int BK, int WM, int WN, bool IS1X1>
__global__ __launch_bounds__(256) void conv_igemm_i8_kernel(ConvArgs a)
{
    constexpr int LROW = BK + 16;          // padded LDS row (bytes)
    constexpr int G = BK / 16;             // 16-B granules per row
    constexpr int RPP = 256 / G;           // rows covered per pass of the 256 threads
    constexpr int PA = (BN + RPP - 1) / RPP;
    constexpr int PB = (BM + RPP - 1) / RPP;
    constexpr int TM = BM / WM / 32;       // 32x32 tiles per wave along pixels
    constexpr int TN = BN / WN / 32;       // .. along cout
    static_assert(WM * WN == 4, "4 waves per block");
    static_assert(TM >= 1 && TN >= 1, "tile too small");

    extern __shared__ __attribute__((aligned(16))) int8_t smem[];
    __shared__ short2 tap_lut[128];        // tap -> (ky*DH, kx*DW)

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wm = wave % WM, wn = wave / WM;

    // ---- XCD-aware tile mapping -------------------------------------------------------------
    const int tiles_n = (a.cout + BN - 1) / BN;
    const int tiles_m = (a.M + BM - 1) / BM;
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int tile_m = (local / tiles_n) * 8 + xcd;
    const int tile_n = local % tiles_n;
    if (tile_m >= tiles_m) return;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int ntaps = a.KH * a.KW;
    if (!IS1X1) {
        if (t < ntaps && t < 128) tap_lut[t] = make_short2((short)((t / a.KW) * a.DH), (short)((t % a.KW) * a.DW));
        __syncthreads();
    }

    // ---- per-thread loader state ------------------------------------------------------------
    const int q = t % G;                   // granule inside the K tile (fixed for this thread)
    const int r0 = t / G;                  // first row handled
    const int8_t* wptr[PA];
#pragma unroll
    for (int p = 0; p < PA; p++) {
        int row = r0 + p * RPP;
        wptr[p] = a.w + (size_t)(n0 + (row < BN ? row : 0)) * a.kpad + q * 16;
    }
    const int8_t* xbase[PB];
    int iy0[PB], ix0[PB];
    bool rvalid[PB];
#pragma unroll
    for (int p = 0; p < PB; p++) {
        int row = r0 + p * RPP;
        int m = m0 + row;
        rvalid[p] = (row < BM) && (m < a.M);
        int mm = rvalid[p] ? m : 0;
        if (IS1X1) {
            xbase[p] = a.x + (size_t)mm * a.cs_in + q * 16;
            iy0[p] = ix0[p] = 0;
        } else {
            int ohw = a.OH * a.OW;
            int n = mm / ohw, rem = mm - n * ohw;
            int oy = rem / a.OW, ox = rem - oy * a.OW;
            xbase[p] = a.x + (size_t)n * a.H * a.W * a.cs_in;
            iy0[p] = oy * a.SH - a.PH;
            ix0[p] = ox * a.SW - a.PW;
        }
    }
    int tap = (q * 16) / a.ckp;            // general path: position of this thread's granule in K
    int ci = (q * 16) - tap * a.ckp;

    v4i ra[PA], rb[PB];
    auto gload = [&](int kt) {
#pragma unroll
        for (int p = 0; p < PA; p++)
            ra[p] = *reinterpret_cast<const v4i*>(wptr[p] + (size_t)kt * BK);
        if (IS1X1) {
            const bool kvalid = (kt * BK + q * 16) < a.ktot;
#pragma unroll
            for (int p = 0; p < PB; p++) {
                v4i z = {0, 0, 0, 0};
                rb[p] = (rvalid[p] && kvalid) ? *reinterpret_cast<const v4i*>(xbase[p] + (size_t)kt * BK) : z;
            }
        } else {
            const bool tvalid = tap < ntaps;
            short2 d = tap_lut[tvalid ? tap : 0];
#pragma unroll
            for (int p = 0; p < PB; p++) {
                int iy = iy0[p] + d.x, ix = ix0[p] + d.y;
                bool ok = rvalid[p] && tvalid && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
                v4i z = {0, 0, 0, 0};
                rb[p] = ok ? *reinterpret_cast<const v4i*>(xbase[p] + ((size_t)iy * a.W + ix) * a.cs_in + ci) : z;
            }
            ci += BK;
            while (ci >= a.ckp) { ci -= a.ckp; tap++; }
        }
    };
    auto lstore = [&](int buf) {
        int8_t* sA = smem + buf * (BM + BN) * LROW;
        int8_t* sB = sA + BN * LROW;
#pragma unroll
        for (int p = 0; p < PA; p++) {
            int row = r0 + p * RPP;
            if (row < BN) *reinterpret_cast<v4i*>(sA + row * LROW + q * 16) = ra[p];
        }
#pragma unroll
        for (int p = 0; p < PB; p++) {
            int row = r0 + p * RPP;
            if (row < BM) *reinterpret_cast<v4i*>(sB + row * LROW + q * 16) = rb[p];
        }
    };

    v16i acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; i++)
#pragma unroll
        for (int j = 0; j < TM; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0;

    // ceil: a stage may run past kpad -- the activation operand is zero there (k >= ktot / tap >= ntaps) and the
    // weight rows are followed by readable memory (next row, or the planner's 256-byte tail), so it adds exact zeros
    const int nk = (a.kpad + BK - 1) / BK;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; kt++) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload(kt + 1);
        const int8_t* sA = smem + buf * (BM + BN) * LROW;
        const int8_t* sB = sA + BN * LROW;
#pragma unroll
        for (int kk = 0; kk < BK / 32; kk++) {
            v4i af[TN], bf[TM];
#pragma unroll
            for (int i = 0; i < TN; i++)
                af[i] = *reinterpret_cast<const v4i*>(sA + ((wn * TN + i) * 32 + l31) * LROW + kk * 32 + hi * 16);
#pragma unroll
            for (int j = 0; j < TM; j++)
                bf[j] = *reinterpret_cast<const v4i*>(sB + ((wm * TM + j) * 32 + l31) * LROW + kk * 32 + hi * 16);
#pragma unroll
            for (int i = 0; i < TN; i++)
#pragma unroll
                for (int j = 0; j < TM; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }

    // ---- fused epilogue: +bias, requantise (bit-exact, epilogue.h), pack 4 channels, NHWC store ----
    // C/D layout of 32x32 MFMA: col = lane&31 (pixel), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (cout)
    const Rq rq = make_rq(a.m1, a.lo, a.hi, a.out_scale);
    // 16-B stores (half-wave regroup) whenever the destination is 16-channel granular; dword stores else
    const bool wide = ((a.c_limit | a.c_off | a.ldc) & 15) == 0 && (!a.elt.res || ((a.elt.res_ldc | a.elt.res_c_off) & 15) == 0);
    const float inv_elt = a.elt.res ? __fdiv_rn(1.0f, a.elt.out_scale) : 1.f;
    const float inv_relu = (a.elt.res && a.elt.relu) ? __fdiv_rn(1.0f, a.elt.relu_out_scale) : 1.f;
#pragma unroll
    for (int i = 0; i < TN; i++) {
        const int cb = n0 + (wn * TN + i) * 32;
        unsigned pp[TM][4];
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) {
            const int4 b4 = *reinterpret_cast<const int4*>(a.bias + cb + 8 * g4 + 4 * hi);
            const float4 s4 = *reinterpret_cast<const float4*>(a.wscale + cb + 8 * g4 + 4 * hi);
#pragma unroll
            for (int j = 0; j < TM; j++)
                pp[j][g4] = requant4(acc[i][j][4 * g4 + 0] + b4.x, acc[i][j][4 * g4 + 1] + b4.y,
                                     acc[i][j][4 * g4 + 2] + b4.z, acc[i][j][4 * g4 + 3] + b4.w, s4, rq);
        }
#pragma unroll
        for (int j = 0; j < TM; j++) {
            const int m = m0 + (wm * TM + j) * 32 + l31;
            unsigned p[4] = {pp[j][0], pp[j][1], pp[j][2], pp[j][3]};
            if (wide) {
                half_wave_regroup(p);
                const int c16 = cb + hi * 16;
                if (m < a.M && c16 < a.c_limit) {
                    if (a.elt.res) {      // eltwise (+ReLU) tail on the 16 channels this lane now holds
                        const uint4 r = *reinterpret_cast<const uint4*>(a.elt.res + (size_t)m * a.elt.res_ldc + a.elt.res_c_off + c16);
                        p[0] = fuse_elt4(p[0], r.x, a.elt, inv_elt, inv_relu);
                        p[1] = fuse_elt4(p[1], r.y, a.elt, inv_elt, inv_relu);
                        p[2] = fuse_elt4(p[2], r.z, a.elt, inv_elt, inv_relu);
                        p[3] = fuse_elt4(p[3], r.w, a.elt, inv_elt, inv_relu);
                    }
                    *reinterpret_cast<uint4*>(a.y + (size_t)m * a.ldc + a.c_off + c16) = make_uint4(p[0], p[1], p[2], p[3]);
                }
            } else {
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
        }
    }
}

template <int BM, int BN, int BK, int WM, int WN>
static hipError_t launch_cfg(const ConvArgs& a, hipStream_t s, bool is1x1)
{
    const int tiles_n = (a.cout + BN - 1) / BN;
    const int tiles_m = (a.M + BM - 1) / BM;
    const int grid = ((tiles_m + 7) / 8) * 8 * tiles_n;
    const size_t lds = 2 * (size_t)(BM + BN) * (BK + 16);
    if (lds > 64 * 1024) {       // gfx950 has 160 KB of LDS per CU; more than 64 KB per workgroup needs the opt-in
        static bool set1 = false, set0 = false;
        bool& done = is1x1 ? set1 : set0;
        if (!done) {
            if (is1x1) (void)hipFuncSetAttribute((const void*)conv_igemm_i8_kernel<BM, BN, BK, WM, WN, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            else (void)hipFuncSetAttribute((const void*)conv_igemm_i8_kernel<BM, BN, BK, WM, WN, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            done = true;
        }
    }
    if (is1x1)
        hipLaunchKernelGGL((conv_igemm_i8_kernel<BM, BN, BK, WM, WN, true>), dim3(grid), dim3(256), lds, s, a);
    else
        hipLaunchKernelGGL((conv_igemm_i8_kernel<BM, BN, BK, WM, WN, false>), dim3(grid), dim3(256), lds, s, a);
    return hipGetLastError();
}

// weights are padded to a multiple of 128 output channels and kpad to a multiple of 128 by the planner,
// so every tile shape below may be chosen freely.
static int pick_cfg(const ConvArgs& a)
{
    static int forced = -2;
    if (forced == -2) { const char* e = getenv("TAMD_IGEMM_CFG"); forced = e ? atoi(e) : -1; }
    if (forced >= 0 && forced <= 9) return forced;
    if (a.cfg >= 0 && a.cfg <= 9) return a.cfg;            // plan-time autotune result
    auto blocks = [&](int bm, int bn) { return (long)((a.M + bm - 1) / bm) * ((a.cout + bn - 1) / bn); };
    // biggest tile that still gives every CU a block; small problems fall to the small tiles
    if (a.cout <= 32) return a.M > 64 ? 1 : 3;
    if (blocks(128, 128) >= 256 && a.cout >= 96) return 0;
    if (blocks(128, 64) >= 256) return 4;
    if (a.M > 64) return 2;
    return 3;
}

const char* conv_igemm_kernel_name(const ConvArgs& a)
{
    static const char* names[] = {"conv_igemm_i8<128x128x64>", "conv_igemm_i8<128x32x64>", "conv_igemm_i8<64x64x64>",
                                  "conv_igemm_i8<32x128x64>", "conv_igemm_i8<128x64x64>",
                                  // deep-K stages (autotune only): 4x the MFMA work per barrier / per exposed latency, for
                                  // the K >= 512 layers (ResNet 3x3) whose 64-deep stages are shorter than a memory round trip
                                  "conv_igemm_i8<128x128x256>", "conv_igemm_i8<128x64x256>", "conv_igemm_i8<64x64x256>",
                                  "conv_igemm_i8<64x64x128>", "conv_igemm_i8<128x64x128>"};
    return names[pick_cfg(a)];
}
int conv_igemm_num_cfgs() { return 10; }
bool conv_igemm_cfg_ok(const ConvArgs& a, int cfg) { return cfg < 5 || (cfg < 8 ? a.kpad >= 512 : a.kpad >= 256); }

hipError_t launch_conv_igemm(const ConvArgs& a, hipStream_t s)
{
    const bool is1x1 = (a.KH == 1 && a.KW == 1 && a.SH == 1 && a.SW == 1 && a.PH == 0 && a.PW == 0);
    switch (pick_cfg(a)) {
    case 0: return launch_cfg<128, 128, 64, 2, 2>(a, s, is1x1);
    case 1: return launch_cfg<128, 32, 64, 4, 1>(a, s, is1x1);
    case 2: return launch_cfg<64, 64, 64, 2, 2>(a, s, is1x1);
    case 4: return launch_cfg<128, 64, 64, 2, 2>(a, s, is1x1);
    case 5: return launch_cfg<128, 128, 256, 2, 2>(a, s, is1x1);
    case 6: return launch_cfg<128, 64, 256, 2, 2>(a, s, is1x1);
    case 7: return launch_cfg<64, 64, 256, 2, 2>(a, s, is1x1);
    case 8: return launch_cfg<64, 64, 128, 2, 2>(a, s, is1x1);
    case 9: return launch_cfg<128, 64, 128, 2, 2>(a, s, is1x1);
    default: return launch_cfg<32, 128, 64, 1, 4>(a, s, is1x1);
    }
}

}  // namespace tamd
