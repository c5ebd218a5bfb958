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
#include "env.h"

#include <type_traits>

#include "epilogue.h"
#include "gemm_epilogue.h"
#include "kernels.h"
#include "conv_igemm_fast.h"

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
    igemm_acc_from_bias<TM, TN>(acc, a.bias, n0, wn, hi);      // the epilogue adds nothing (gemm_epilogue.h)

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

    igemm_epilogue<TM, TN>(a, acc, m0, n0, wm, wn, l31, hi);
}

// ---- the same GEMM with a software pipeline that really runs ahead (the one-stage register prefetch above
// exposes a memory round trip at every barrier, ~60 % of the wave cycles sat in s_waitcnt).  BK = 64 per stage;
//   * a register RING of D stages of global loads: the loads of stage kt+D-1 are issued while stage kt is multiplied, so
//     D-1 stages of MFMA work cover the load latency (each stage is 4 x 16 B per thread for a 128x128 tile);
//   * LDS double buffer, ONE barrier per stage:  lstore(kt) ; gload(kt+D-1) ; barrier ; MFMAs(kt);
//   * unpadded 64-byte LDS rows with an XOR swizzle of the 16-byte granule, pos = q ^ ((row >> 2) & 3): the 16 lanes of a
//     ds_write_b128 (4 rows x 4 granules) and of a ds_read_b128 (16 rows, one granule) both touch 64 distinct banks
//     (the padded 80-byte rows above collide rows r and r+3 on the stores: 32 % of the LDS cycles).
template <int BM, int BN, int WM, int WN, bool IS1X1, int D>
__global__ __launch_bounds__(256) void conv_igemm_ring_i8_kernel(ConvArgs a)
{
    constexpr int BK = 64, RPP = 64;                   // 256 threads = 64 rows x 4 granules per pass
    constexpr int PA = (BN + RPP - 1) / RPP, PB = (BM + RPP - 1) / RPP;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    static_assert(WM * WN == 4 && TM >= 1 && TN >= 1, "4 waves per block");

    extern __shared__ __attribute__((aligned(16))) int8_t smem[];       // 2 x (BN + BM) rows of 64 B
    __shared__ short2 tap_lut[128];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, hi = lane >> 5;
    const int wm = wave % WM, wn = wave / WM;
    const int tiles_n = (a.cout + BN - 1) / BN, tiles_m = (a.M + BM - 1) / BM;
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int tile_m = (local / tiles_n) * 8 + xcd, tile_n = local % tiles_n;
    if (tile_m >= tiles_m) return;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int ntaps = a.KH * a.KW;
    if (!IS1X1) {
        if (t < ntaps && t < 128) tap_lut[t] = make_short2((short)((t / a.KW) * a.DH), (short)((t % a.KW) * a.DW));
        __syncthreads();
    }

    const int q = t & 3, r0 = t >> 2;
    const int8_t* wptr[PA];
#pragma unroll
    for (int p = 0; p < PA; p++) {
        const int row = r0 + p * RPP;
        wptr[p] = a.w + (size_t)(n0 + (row < BN ? row : 0)) * a.kpad + q * 16;
    }
    const int8_t* xbase[PB];
    int iy0[PB], ix0[PB];
    bool rvalid[PB];
#pragma unroll
    for (int p = 0; p < PB; p++) {
        const int row = r0 + p * RPP, m = m0 + row;
        rvalid[p] = (row < BM) && (m < a.M);
        const int mm = rvalid[p] ? m : 0;
        if (IS1X1) {
            xbase[p] = a.x + (size_t)mm * a.cs_in + q * 16;
            iy0[p] = ix0[p] = 0;
        } else {
            const int ohw = a.OH * a.OW, n = mm / ohw, rem = mm - n * ohw, oy = rem / a.OW, ox = rem - oy * a.OW;
            xbase[p] = a.x + (size_t)n * a.H * a.W * a.cs_in;
            iy0[p] = oy * a.SH - a.PH;
            ix0[p] = ox * a.SW - a.PW;
        }
    }
    int tap = (q * 16) / a.ckp, ci = (q * 16) - tap * a.ckp;

    // Every load of the pipeline is UNCONDITIONAL (out-of-image taps, pixels past M and K past ktot read the planner's zero
    // page instead of being predicated) and every stage executes the same instruction sequence: the compiler's s_waitcnt
    // insertion can then count the loads in flight (vmcnt(N) for the oldest stage only).  With predicated loads or a
    // conditional refill it falls back to vmcnt(0) at every stage -- seen in the ISA of the first version of this kernel --
    // and the ring never runs ahead.
    v4i ra[D][PA], rb[D][PB];
    auto gload = [&](int kt, v4i (&A)[PA], v4i (&B)[PB]) {
#pragma unroll
        for (int p = 0; p < PA; p++) A[p] = *reinterpret_cast<const v4i*>(wptr[p] + (size_t)kt * BK);
        if (IS1X1) {
            const bool kvalid = (kt * BK + q * 16) < a.ktot;
#pragma unroll
            for (int p = 0; p < PB; p++) {
                const int8_t* src = (rvalid[p] && kvalid) ? xbase[p] + (size_t)kt * BK : a.zeros;
                B[p] = *reinterpret_cast<const v4i*>(src);
            }
        } else {
            const bool tvalid = tap < ntaps;
            const short2 d = tap_lut[tvalid ? tap : 0];
#pragma unroll
            for (int p = 0; p < PB; p++) {
                const int iy = iy0[p] + d.x, ix = ix0[p] + d.y;
                const bool ok = rvalid[p] && tvalid && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                const int8_t* src = ok ? xbase[p] + ((size_t)iy * a.W + ix) * a.cs_in + ci : a.zeros;
                B[p] = *reinterpret_cast<const v4i*>(src);
            }
            ci += BK;
            while (ci >= a.ckp) { ci -= a.ckp; tap++; }
        }
    };
    auto lstore = [&](int buf, const v4i (&A)[PA], const v4i (&B)[PB]) {
        int8_t* sA = smem + buf * (BM + BN) * BK;
        int8_t* sB = sA + BN * BK;
#pragma unroll
        for (int p = 0; p < PA; p++) {
            const int row = r0 + p * RPP;
            if (row < BN) *reinterpret_cast<v4i*>(sA + row * BK + ((q ^ ((row >> 2) & 3)) << 4)) = A[p];
        }
#pragma unroll
        for (int p = 0; p < PB; p++) {
            const int row = r0 + p * RPP;
            if (row < BM) *reinterpret_cast<v4i*>(sB + row * BK + ((q ^ ((row >> 2) & 3)) << 4)) = B[p];
        }
    };

    v16i acc[TN][TM];
    igemm_acc_from_bias<TM, TN>(acc, a.bias, n0, wn, hi);      // the epilogue adds nothing (gemm_epilogue.h)

    // the launcher picks D so that the stage count needs little padding; padded stages multiply zeros (B reads the zero page
    // once K is exhausted; the weight rows are followed by readable slack, graph_plan.hip dev_alloc)
    const int nk = ((a.kpad + BK - 1) / BK + D - 1) / D * D;
    gload(0, ra[0], rb[0]);
    if constexpr (D > 2) gload(1, ra[1], rb[1]);
    if constexpr (D > 3) gload(2, ra[2], rb[2]);
    // one pipeline stage with COMPILE-TIME ring slots (a dynamically indexed register array would live in scratch memory)
    auto stage = [&](int kt, auto U) {
        constexpr int u = decltype(U)::value, un = (u + D - 1) % D;
        const int buf = kt & 1;
        lstore(buf, ra[u], rb[u]);                      // stage kt: its loads were issued D-1 stages ago
        gload(kt + D - 1, ra[un], rb[un]);              // runs ahead; past the end it fetches zeros / slack that nobody uses
        __syncthreads();
        const int8_t* sA = smem + buf * (BM + BN) * BK;
        const int8_t* sB = sA + BN * BK;
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
            v4i af[TN], bf[TM];
#pragma unroll
            for (int i = 0; i < TN; i++) {
                const int row = (wn * TN + i) * 32 + l31;
                af[i] = *reinterpret_cast<const v4i*>(sA + row * BK + (((kk * 2 + hi) ^ ((row >> 2) & 3)) << 4));
            }
#pragma unroll
            for (int j = 0; j < TM; j++) {
                const int row = (wm * TM + j) * 32 + l31;
                bf[j] = *reinterpret_cast<const v4i*>(sB + row * BK + (((kk * 2 + hi) ^ ((row >> 2) & 3)) << 4));
            }
#pragma unroll
            for (int i = 0; i < TN; i++)
#pragma unroll
                for (int j = 0; j < TM; j++) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    };
    for (int kt0 = 0; kt0 < nk; kt0 += D) {
        stage(kt0, std::integral_constant<int, 0>{});
        stage(kt0 + 1, std::integral_constant<int, 1>{});
        if constexpr (D > 2) stage(kt0 + 2, std::integral_constant<int, 2>{});
        if constexpr (D > 3) stage(kt0 + 3, std::integral_constant<int, 3>{});
    }
    igemm_epilogue<TM, TN>(a, acc, m0, n0, wm, wn, l31, hi);
}

static bool fast_loader_ok(const ConvArgs& a)
{
    const long xbytes = (long)a.N * a.H * a.W * a.cs_in + (long)(a.PH * a.W + a.PW) * a.cs_in;
    const long wbytes = (long)((a.cout + 255) / 256 * 256) * a.kpad;
    return a.ckp % 64 == 0 && a.KH * a.KW <= 32 && a.M < (1 << 24) && a.OH * a.OW < 65536 && xbytes < (1L << 31) - 4096 && wbytes < (1L << 31) - 4096 && a.ktot == a.KH * a.KW * a.ckp
           && a.kpad == a.ktot;
}

template <int BM, int BN, int WM, int WN, int D>
static hipError_t launch_ring_d(const ConvArgs& a, hipStream_t s, bool is1x1)
{
    const int tiles_n = (a.cout + BN - 1) / BN, tiles_m = (a.M + BM - 1) / BM;
    const int grid = ((tiles_m + 7) / 8) * 8 * tiles_n;
    const size_t lds = 2 * (size_t)(BM + BN) * 64;
    static const bool no_fast = exp_env("TAMD_IGEMM_FAST") && atoi(exp_env("TAMD_IGEMM_FAST")) == 0;      // tests: the generic loader
    if constexpr (BM % 64 == 0 && BN % 64 == 0) {
        if (fast_loader_ok(a) && !no_fast) {
            if (is1x1) hipLaunchKernelGGL((conv_igemm_fast_i8_kernel<BM, BN, WM, WN, true, D>), dim3(grid), dim3(256), lds, s, a);
            else hipLaunchKernelGGL((conv_igemm_fast_i8_kernel<BM, BN, WM, WN, false, D>), dim3(grid), dim3(256), lds, s, a);
            return hipGetLastError();
        }
    }
    if (is1x1) hipLaunchKernelGGL((conv_igemm_ring_i8_kernel<BM, BN, WM, WN, true, D>), dim3(grid), dim3(256), lds, s, a);
    else hipLaunchKernelGGL((conv_igemm_ring_i8_kernel<BM, BN, WM, WN, false, D>), dim3(grid), dim3(256), lds, s, a);
    return hipGetLastError();
}

// ring depth: the deepest of 4 / 3 / 2 that the stage count divides by (3x3 layers: 9 * cin / 64 stages -> 3; deep 1x1: 4),
// else the one with the least padding
static int ring_depth(const ConvArgs& a)
{
    const int nk = (a.kpad + 63) / 64;
    int best = 2, pad = (nk + 1) / 2 * 2 - nk;
    for (int d : {3, 4}) { const int p = (nk + d - 1) / d * d - nk; if (p <= pad) { pad = p; best = d; } }
    return best;
}

template <int BM, int BN, int WM, int WN>
static hipError_t launch_ring(const ConvArgs& a, hipStream_t s, bool is1x1)
{
    switch (ring_depth(a)) {
    case 4: return launch_ring_d<BM, BN, WM, WN, 4>(a, s, is1x1);
    case 3: return launch_ring_d<BM, BN, WM, WN, 3>(a, s, is1x1);
    default: return launch_ring_d<BM, BN, WM, WN, 2>(a, s, is1x1);
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
static constexpr int NCFG = 16;
static int pick_cfg(const ConvArgs& a)
{
    static int forced = -2;
    if (forced == -2) { const char* e = exp_env("TAMD_IGEMM_CFG"); forced = e ? atoi(e) : -1; }
    if (forced >= 0 && forced < NCFG) return forced;
    if (a.cfg >= 0 && a.cfg < NCFG) return a.cfg;            // plan-time autotune result
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
                                  "conv_igemm_i8<64x64x128>", "conv_igemm_i8<128x64x128>",
                                  // software-pipelined variants (register ring of 4 stages, swizzled LDS): autotune only
                                  "conv_igemm_i8<128x128x64,ring>", "conv_igemm_i8<128x64x64,ring>", "conv_igemm_i8<256x64x64,ring>",
                                  "conv_igemm_i8<64x128x64,ring>", "conv_igemm_i8<64x64x64,ring>", "conv_igemm_i8<256x128x64,ring>"};
    return names[pick_cfg(a)];
}
int conv_igemm_num_cfgs() { return NCFG; }
bool conv_igemm_cfg_ok(const ConvArgs& a, int cfg) { return cfg < 5 || cfg >= 10 || (cfg < 8 ? a.kpad >= 512 : a.kpad >= 256); }

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
    case 10: return launch_ring<128, 128, 2, 2>(a, s, is1x1);
    case 11: return launch_ring<128, 64, 2, 2>(a, s, is1x1);     // 128 pixels x 64 couts
    case 12: return launch_ring<256, 64, 4, 1>(a, s, is1x1);
    case 13: return launch_ring<64, 128, 1, 4>(a, s, is1x1);
    case 14: return launch_ring<64, 64, 2, 2>(a, s, is1x1);
    case 15: return launch_ring<256, 128, 2, 2>(a, s, is1x1);    // 128 x 64 per wave: 8 MFMAs per 6 LDS fragment reads
    default: return launch_cfg<32, 128, 64, 1, 4>(a, s, is1x1);
    }
}

}  // namespace tamd
