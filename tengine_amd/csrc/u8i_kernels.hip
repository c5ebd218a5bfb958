// uint8 convolution on the INT8 matrix cores -- the opt-in integer path (tamd_options.u8_integer / TAMD_U8_INT=1).
//
// The reference simulates uint8 in fp32 (conv_kernel_x86.c:68-80 weights -> fp32, :126-185 im2col_uint8, :322-960 sgemm_fp,
// :1703-1794 requantise), so its bytes carry the rounding of an fp32 summation; the byte-exact path (u8_kernels.hip) repeats
// that chain on the fp32 MFMA (157 TFLOP/s peak).  BASELINE.md section 2 / SURVEY section 7 step 5 state the policy for a faster
// uint8 form: results within ONE quantisation step of the reference, mismatch histogram reported.  This file is that form:
//
//   sum_k (x_k - zx)(w_k - zw)        exactly, in int32, on v_mfma_i32_32x32x32_i8 (32x the fp32 MFMA rate)
//   q = clamp(round_half_away((float)(sum + bias) * M) + zp, lo, hi),  M = fl(fl(in_scale * w_scale) / out_scale)
//                                                                         ONE multiply-add instead of K roundings, a division and a
//                                                                         round (u8_epilogue.h: u8i_requant); the conv's own
//                                                                         activation (:1746-1767) is the clamp window
//   then the fused ReLU / leaky ReLU node and the fused 2x2 max-pool node exactly as the byte-exact kernels apply them to the byte
//   (u8_epilogue.h: fused_relu, pooled_byte -- tabulated per block, both are functions of one byte).
//
// Operands: x' = x - 128 and w' = w - 128 are int8 (a byte XOR 0x80).  With alpha = zx - 128, beta = zw - 128:
//   sum_k (x'_k - alpha)(w'_k - beta) = sum x'w'  -  beta * sum_k x'_k  -  alpha * sum_k w'_k  +  Kp * alpha * beta
// `sum x'w'` is the MFMA; `sum_k w'_k` and the constant are folded into a per-channel int32 vector at plan time (cvec); the one
// data-dependent term, the sum of the pixel's im2col column, is accumulated beside the MFMAs with v_dot4_i32_i8 on the B
// fragments the lane holds anyway (skipped when zw == 128).  Out-of-image taps hold x' = alpha (the reference's im2col writes
// 0.0f there = the dequantised zero point) and padded channels hold w' = beta: both contribute exactly 0.
//
// Integer accumulation is order-free, so K is ordered for the hardware: k = (32-channel chunk, ky, kx, channel) -- an MFMA K
// step is 32 channels of one tap.  Layout in HBM stays the reference's dense NCHW bytes on both sides (the glue kernels of the
// uint8 planner are shared with the byte-exact path), so the kernel transposes while it stages:
//   B: the block keeps the input PATCH of its pixel tile (bounding box of every input row / column the tile's taps touch, pad
//      bytes outside the image) for 32 channels in LDS, [16-channel granule][patch pixel][16 B]; a thread loads 4 channels x 4
//      consecutive patch columns as four (unaligned) dwords, fixes the columns that fall outside the image with one v_perm_b32,
//      4x4 byte-transposes (8 v_perm_b32) and writes four dwords.  A tap is then a scalar offset on the lane's ds_read_b128
//      address -- each input byte enters the CU once per block instead of KH*KW times.  Two patch buffers: the next chunk's
//      global loads are issued before the current chunk's MFMAs and stored behind them; one barrier per chunk.
//   A: weights packed at plan time in MFMA fragment order [cout tile][step = (chunk, tap)][32-row fragment][lane][16 B],
//      fetched from global straight into a 3-deep register ring (shared by every pixel tile through the L2s).
//   D: lanes run along pixels = along NCHW rows; 16 output channels per lane and 32x32 tile.
#include <hip/hip_runtime.h>
#include "env.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "dw_common.h"
#include "kernels.h"
#include "u8_epilogue.h"

namespace tamd {

typedef int v4i_q __attribute__((ext_vector_type(4)));
typedef int v16i_q __attribute__((ext_vector_type(16)));

template <int I, int N, typename F>
__device__ __forceinline__ void static_for_u8i(F&& f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for_u8i<I + 1, N>(f);
    }
}

// bounding box of the output pixels j0..jl (conv_pixel order: row-major, or window-major under a fused 2x2 max-pool)
__host__ __device__ inline void u8i_box(int OW, int pool_on, int j0, int jl, int* oy0, int* oy1, int* ox0, int* ox1)
{
    if (pool_on) {
        const int half = OW >> 1, w0 = j0 >> 2, w1 = jl >> 2, py0 = w0 / half, py1 = w1 / half;
        *oy0 = 2 * py0; *oy1 = 2 * py1 + 1;
        if (py0 == py1) { *ox0 = 2 * (w0 - py0 * half); *ox1 = 2 * (w1 - py1 * half) + 1; }
        else { *ox0 = 0; *ox1 = OW - 1; }
    } else {
        *oy0 = j0 / OW; *oy1 = jl / OW;
        if (*oy0 == *oy1) { *ox0 = j0 - *oy0 * OW; *ox1 = jl - *oy1 * OW; }
        else { *ox0 = 0; *ox1 = OW - 1; }
    }
}

template <int WM, int WN, int TM, int TN, int NU>
__global__ __launch_bounds__(256) void conv_u8i_k(const U8ConvArgs a)
{
    static_assert(WM * WN == 4, "four waves");
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const int t = threadIdx.x, lane = t & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const int OHW = a.OH * a.OW, HW = a.H * a.W;
    // pixel tiles: LINEAR (i_tw == 0): BN consecutive pixels of the image in conv_pixel order -- every lane useful on small maps;
    // 2-D (i_tw = tile width 8 | 16): a TH x TW window of the output map, so the patch stays small on maps too wide for a
    // linear tile's bounding box (the launcher picks: conv_u8i_prepare)
    const int TW = a.i_tw, TH = TW ? BN / TW : 0;
    const int tiles_x = TW ? (a.OW + TW - 1) / TW : 0;
    const int tiles = TW ? tiles_x * ((a.OH + TH - 1) / TH) : (OHW + BN - 1) / BN;
    const int n = blockIdx.x / tiles, tile = blockIdx.x - n * tiles;
    const int co0 = blockIdx.y * BM;
    const int j0 = tile * BN, jl = min(j0 + BN, OHW) - 1;                 // linear tiles
    const int oyb = TW ? (tile / tiles_x) * TH : 0, oxb = TW ? (tile - (tile / tiles_x) * tiles_x) * TW : 0;      // 2-D tiles
    const int KH = a.pk_kh, KW = a.pk_kw, DH = a.pk_dh, DW = a.pk_dw, ntaps = KH * KW;
    // local pixel index (0 .. BN-1) -> output coordinates (clamped into the map for lanes past the tile's last pixel) and whether
    // the lane holds a real pixel.  Under a fused 2x2 max-pool both tile forms enumerate window-major (u8_epilogue.h conv_pixel)
    auto pixel_of = [&](int jloc, int* oy, int* ox) -> bool {
        if (TW == 0) {
            const int j = j0 + jloc;
            conv_pixel(a, min(j, jl), oy, ox);
            return j <= jl;
        }
        int dy, dx;
        if (a.pool.on) { const int w = jloc >> 2, hw = TW >> 1, wy = w / hw, wx = w - wy * hw; dy = 2 * wy + ((jloc >> 1) & 1); dx = 2 * wx + (jloc & 1); }
        else { dy = jloc / TW; dx = jloc - dy * TW; }
        const bool live = oyb + dy < a.OH && oxb + dx < a.OW;
        *oy = min(oyb + dy, a.OH - 1); *ox = min(oxb + dx, a.OW - 1);
        return live;
    };

    // ---- the patch: input rows [RY0, RY0 + rows), columns [XA, XA + Wp4) -- out-of-image positions hold the pad byte -----------
    int oy0, oy1, ox0, ox1;
    if (TW == 0) u8i_box(a.OW, a.pool.on, j0, jl, &oy0, &oy1, &ox0, &ox1);
    else { oy0 = oyb; oy1 = min(oyb + TH, a.OH) - 1; ox0 = oxb; ox1 = min(oxb + TW, a.OW) - 1; }
    const int RY0 = oy0 * a.SH - a.PH, XA = ox0 * a.SW - a.PW;
    const int Wp4 = ((ox1 - ox0) * a.SW + (KW - 1) * DW + 1 + 3) & ~3, W4q = Wp4 >> 2;
    const int rows = (oy1 - oy0) * a.SH + (KH - 1) * DH + 1;
    const int P = rows * Wp4;                                // <= NPAD (the launcher checked the worst tile)
    const int NPAD = a.i_npad;                               // pixels per granule plane
    const unsigned padw = (unsigned)(a.i_alpha & 0xff) * 0x01010101u;
    // the fused ReLU node and the fused pool node are functions of ONE byte (u8_epilogue.h: fused_relu of the conv's own byte,
    // pooled_byte of the window maximum): tabulated once per block (thread t: entry t), looked up per output -- the same bytes as
    // evaluating them per output, at a thirtieth of the arithmetic
    uint8_t* lut = smem + 2 * (2 << a.i_cgs) * NPAD * 16;        // [256] fused ReLU, [256] pooled byte
    lut[t] = a.relu.on ? fused_relu((uint8_t)t, a.out_scale, a.out_zp, a.relu) : (uint8_t)t;
    if (a.pool.on) lut[256 + t] = pooled_byte(t, a.pool);
    const U8IRq rq{a.i_m, a.out_zp, a.i_qlo, a.i_qhi};

    // A chunk = CG groups of 32 channels (CG = 1 | 2 | 4: small patches take more channels per barrier; 1x1 layers would otherwise
    // see a barrier per MFMA step).  Staging units: (channel quad cq of the chunk, patch pixel quad q); surplus threads repeat
    // the last unit (same bytes to the same place)
    const int CGS = a.i_cgs, CG = 1 << CGS;                  // log2 / groups per chunk
    const int BUFB = 2 * CG * NPAD * 16;                     // bytes of a patch buffer: 2 * CG granule planes
    const uint8_t* xin = a.x + (size_t)n * a.C * HW;
    int goff[NU], ldst[NU], cq4[NU];
    unsigned sel[NU];
#pragma unroll
    for (int i = 0; i < NU; i++) {
        const int u = min(t + 256 * i, 2 * CG * P - 1);
        const int cq = u & (8 * CG - 1), q = u >> (3 + CGS), r = q / W4q, xq = q - r * W4q;
        const int iy = RY0 + r, ix0 = XA + 4 * xq;
        const bool rowok = (unsigned)iy < (unsigned)a.H;
        const int ixc = min(max(ix0, 0), a.W - 4);
        unsigned sl = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int col = ix0 + j;
            const bool ok = rowok && (unsigned)col < (unsigned)a.W;
            sl |= (unsigned)(ok ? col - ixc : 4) << (8 * j);
        }
        sel[i] = sl;
        goff[i] = rowok ? iy * a.W + ixc : 0;
        cq4[i] = cq * 4;
        ldst[i] = ((cq >> 2) * NPAD + 4 * q) * 16 + (cq & 3) * 4;
    }
    unsigned sv[NU][4];
    auto stage_load = [&](int cc) {
#pragma unroll
        for (int i = 0; i < NU; i++)
#pragma unroll
            for (int c4 = 0; c4 < 4; c4++) {
                const int c = min(cc * (32 * CG) + cq4[i] + c4, a.C - 1);      // padded channels re-read the last one: they meet w' = beta
                if (a.i_dbg & 4) sv[i][c4] = 0x01020304u * (unsigned)(c + 1);   // anatomy runs only (TAMD_U8I_ABLATE): no input loads
                else __builtin_memcpy(&sv[i][c4], xin + (size_t)c * HW + goff[i], 4);
            }
    };
    auto stage_store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NU; i++) {
            unsigned d[4], x[4];
#pragma unroll
            for (int c4 = 0; c4 < 4; c4++) d[c4] = __builtin_amdgcn_perm(padw, sv[i][c4] ^ 0x80808080u, sel[i]);
            transpose4x4(d, x);                                         // d[c] = channel c at 4 pixels -> x[p] = pixel p's 4 channels
#pragma unroll
            for (int p = 0; p < 4; p++) *reinterpret_cast<unsigned*>(smem + buf * BUFB + ldst[i] + 16 * p) = x[p];
        }
    };

    // ---- B fragment addresses: lane (pixel l31 of tile jn, k half hi) ------------------------------------------------------------
    int bfr[TN];
#pragma unroll
    for (int jn = 0; jn < TN; jn++) {
        int oy, ox;
        (void)pixel_of((wn * TN + jn) * 32 + l31, &oy, &ox);              // lanes past the last pixel read a real one; never stored
        const int pp0 = (oy * a.SH - a.PH - RY0) * Wp4 + (ox * a.SW - a.PW - XA);
        bfr[jn] = (hi * NPAD + pp0) * 16;
    }
    const int row_step = (DH * Wp4 - KW * DW) * 16, grp_step = 2 * NPAD * 16;

    // ---- A: fragments of this wave's TM 32-row tiles, one step = (BM / 32) KB; a register ring D steps deep: with one or two
    // waves per SIMD a fragment requested two steps ahead (~200 cycles) is not there yet, eight steps are ----------------------------
    constexpr int D = 8;
    const int ngroups = a.i_nchunks, ns = ngroups * ntaps, nchunks = (ngroups + CG - 1) >> CGS;
    const int8_t* wt = a.iw + ((size_t)blockIdx.y * ns * (BM / 32) + wm * TM) * 1024 + lane * 16;
    // (a running pointer, no clamp: the packed weights carry D + 1 steps of readable slack behind the last one)
    auto load_a = [&](v4i_q (&f)[TM]) {
#pragma unroll
        for (int i = 0; i < TM; i++) f[i] = *reinterpret_cast<const v4i_q*>(wt + i * 1024);
        wt += (BM / 32) * 1024;
    };

    v16i_q acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int jn = 0; jn < TN; jn++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][jn][e] = 0;
    int ssum[TN];
#pragma unroll
    for (int jn = 0; jn < TN; jn++) ssum[jn] = 0;
    const bool need_sum = a.i_beta != 0;

    v4i_q ar[D][TM];
#pragma unroll
    for (int d = 0; d < D; d++) load_a(ar[d]);
    // the epilogue's per-channel constants, requested now (see conv_u8i_pw_k)
    int4 cvr[TM][4];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) cvr[i][g4] = *reinterpret_cast<const int4*>(a.icv + co0 + (wm * TM + i) * 32 + 8 * g4 + 4 * hi);
    stage_load(0);
    stage_store(0);
    __syncthreads();
    if (nchunks > 1) stage_load(1);

    // read position = the step whose B fragments are fetched next (one step ahead of the MFMAs): chunk, group inside it, tap
    int r_chunk = 0, r_grp = 0, r_tap = 0, r_kx = 0, r_off = 0;          // r_off: byte offset inside the patch buffer (group plane + tap)
    int r_ngrp = min(CG, ngroups);                                       // groups of the read chunk
    v4i_q bf[TN], bfn[TN];
#pragma unroll
    for (int jn = 0; jn < TN; jn++) bf[jn] = *reinterpret_cast<const v4i_q*>(smem + bfr[jn]);
    for (int s0 = 0; s0 < ns; s0 += D) {
        static_for_u8i<0, D>([&](auto Dd) {
            constexpr int d = decltype(Dd)::value;
            const int st = s0 + d;
            if (st < ns) {                                               // uniform
                // advance the read position to step st + 1
                bool wrap = false;
                r_tap++; r_kx++; r_off += DW * 16;
                if (r_kx == KW) { r_kx = 0; r_off += row_step; }
                if (r_tap == ntaps) {
                    r_tap = 0; r_kx = 0; r_grp++;
                    r_off = r_grp * grp_step;
                    if (r_grp == r_ngrp) { r_grp = 0; r_off = 0; r_chunk++; r_ngrp = min(CG, ngroups - r_chunk * CG); wrap = true; }
                }
                const bool more = st + 1 < ns;
                if (more && !wrap) {
                    const uint8_t* pb = smem + (r_chunk & 1) * BUFB + r_off;
#pragma unroll
                    for (int jn = 0; jn < TN; jn++) bfn[jn] = *reinterpret_cast<const v4i_q*>(pb + bfr[jn]);
                }
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int jn = 0; jn < TN; jn++) acc[i][jn] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ar[d][i], bf[jn], acc[i][jn], 0, 0, 0);
                if (need_sum) {
#pragma unroll
                    for (int jn = 0; jn < TN; jn++)
#pragma unroll
                        for (int e = 0; e < 4; e++) ssum[jn] = __builtin_amdgcn_sdot4(bf[jn][e], 0x01010101, ssum[jn], false);
                }
                load_a(ar[d]);
                if (more && wrap) {
                    // the next chunk: its bytes were requested when this one started; everyone is past the other buffer's last read
                    // (that chunk ended at the previous barrier)
                    stage_store(r_chunk & 1);
                    __syncthreads();
                    if (r_chunk + 1 < nchunks) stage_load(r_chunk + 1);
                    const uint8_t* pb = smem + (r_chunk & 1) * BUFB;
#pragma unroll
                    for (int jn = 0; jn < TN; jn++) bfn[jn] = *reinterpret_cast<const v4i_q*>(pb + bfr[jn]);
                }
#pragma unroll
                for (int jn = 0; jn < TN; jn++) bf[jn] = bfn[jn];
            }
        });
    }

    // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane & 31 (pixel), row = (e & 3) + 8 * (e >> 2) + 4 * hi (channel) ------
#pragma unroll
    for (int jn = 0; jn < TN; jn++) {
        // the column sum of the pixel: the two k halves of a step sit in lanes l31 and l31 + 32
        const int colsum = ssum[jn] + __shfl_xor(ssum[jn], 32);
        int oy, ox;
        const bool live = pixel_of((wn * TN + jn) * 32 + l31, &oy, &ox);
        const int opix = oy * a.OW + ox, ppix = (oy >> 1) * (a.OW >> 1) + (ox >> 1);
        const int corr = a.i_beta * colsum;
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) {
                const int cb = co0 + (wm * TM + i) * 32 + 8 * g4 + 4 * hi;
                const int4 cv = cvr[i][g4];                                          // (icv is padded to the cout tile)
                const int cvs[4] = {cv.x, cv.y, cv.z, cv.w};
                int qv[4];
                if (a.i_dbg & 2) { for (int e = 0; e < 4; e++) qv[e] = (acc[i][jn][4 * g4 + e] + cvs[e]) & 255; }   // anatomy: no requantisation arithmetic
                else {
#pragma unroll
                    for (int e = 0; e < 4; e++) qv[e] = u8i_requant(acc[i][jn][4 * g4 + e] - corr + cvs[e], rq);
                }
                // table look-ups of the group first (four LDS reads in flight, one wait), then the stores, each under ONE predicate:
                // written value by value the compiler waits for every look-up in front of its own store
                int qb[4], pb[4];
                if (a.relu.on) {
#pragma unroll
                    for (int e = 0; e < 4; e++) qb[e] = lut[qv[e]];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; e++) qb[e] = qv[e];
                }
                const bool full = live && (!a.pool.on || a.pool.write_full) && !(a.i_dbg & 1);
                uint8_t* dst = a.y + (size_t)n * a.out_img + (size_t)(a.out_c0 + cb) * OHW + opix;
#pragma unroll
                for (int e = 0; e < 4; e++)
                    if (full && cb + e < a.cout) dst[(size_t)e * OHW] = (uint8_t)qb[e];
                if (a.pool.on) {                 // OH, OW even under a fused pool: a window's four pixels are four neighbouring lanes, live together
#pragma unroll
                    for (int e = 0; e < 4; e++) pb[e] = lut[256 + quad_max(qb[e])];
                    const bool pst = live && (l31 & 3) == 0 && !(a.i_dbg & 1);
                    uint8_t* pd = a.pool.y + (size_t)n * a.pool.out_img + (size_t)(a.pool.out_c0 + cb) * (OHW >> 2) + ppix;
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        if (pst && cb + e < a.cout) pd[(size_t)e * (OHW >> 2)] = (uint8_t)pb[e];
                }
            }
    }
}

// =================================================================================================================
// First layers (3x3, at most 4 input channels: YOLOv3-tiny / MobileNet / MobileNet-SSD conv0) on the same arithmetic.  The
// 32-channel K step of conv_u8i_k would be nine tenths padding here; instead the whole K = 9 taps x 4 channel slots is ONE
// v_mfma_i32_16x16x64_i8: k = tap * 4 + channel, lane (pixel l15, quarter q) supplies taps 4q .. 4q+3 (taps >= 9 and the 4th
// channel slot are zero bytes on both operands).  The block's 16x16 output window keeps its input patch in LDS pixel-major, 4 bytes
// per pixel {c0, c1, c2, 0} (staged from the three NCHW planes with dword loads + one 4x4 byte transpose), so a lane's 16-byte B
// fragment is four ds_read_b32 at pp0 + tap offset -- no im2col, no per-byte gathers.  Weights: one 16-row fragment per 16 output
// channels, in registers for the whole block.  Epilogue as conv_u8i_k (column sums over the REAL k only: the zero slots add
// nothing to either side of the expansion).
// =================================================================================================================
template <int RT>
__global__ __launch_bounds__(256) void conv_u8i_rgb_k(const U8ConvArgs a)
{
    constexpr int TH = 16, TW = 16;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int t = threadIdx.x, lane = t & 63, l15 = lane & 15, q = lane >> 4, wave = t >> 6;
    const int OHW = a.OH * a.OW, HW = a.H * a.W;
    const int tiles_x = (a.OW + TW - 1) / TW, tiles = tiles_x * ((a.OH + TH - 1) / TH);
    const int n = blockIdx.x / tiles, tile = blockIdx.x - n * tiles;
    const int oyb = (tile / tiles_x) * TH, oxb = (tile - (tile / tiles_x) * tiles_x) * TW;
    const int oy1 = min(oyb + TH, a.OH) - 1, ox1 = min(oxb + TW, a.OW) - 1;
    const int RY0 = oyb * a.SH - a.PH, XA = oxb * a.SW - a.PW;
    const int Wp4 = ((ox1 - oxb) * a.SW + 3 + 3) & ~3, W4q = Wp4 >> 2, rows = (oy1 - oyb) * a.SH + 3;
    const unsigned padw = (unsigned)(a.i_alpha & 0xff) * 0x01010101u;
    const uint8_t* xin = a.x + (size_t)n * a.C * HW;
    uint8_t* lut = smem + a.i_npad * 4;                         // byte tables of the fused ReLU / pool nodes (see conv_u8i_k), behind the patch
    lut[t] = a.relu.on ? fused_relu((uint8_t)t, a.out_scale, a.out_zp, a.relu) : (uint8_t)t;
    if (a.pool.on) lut[256 + t] = pooled_byte(t, a.pool);
    const U8IRq rq{a.i_m, a.out_zp, a.i_qlo, a.i_qhi};

    // ---- weights of this block's output channels: RT fragments of 16 rows, lane (row l15, quarter q) ------------------------------
    v4i_q af[RT];
    int4 cvr[RT];
#pragma unroll
    for (int r = 0; r < RT; r++) {
        af[r] = *reinterpret_cast<const v4i_q*>(a.iw + (size_t)r * 1024 + lane * 16);
        cvr[r] = *reinterpret_cast<const int4*>(a.icv + r * 16 + 4 * q);
    }

    // ---- stage the patch: unit = 4 consecutive patch columns of one row, all channels ------------------------------------------------
    for (int u = t; u < rows * W4q; u += 256) {
        const int r = u / W4q, xq = u - r * W4q;
        const int iy = RY0 + r, ix0 = XA + 4 * xq;
        const bool rowok = (unsigned)iy < (unsigned)a.H;
        const int ixc = min(max(ix0, 0), a.W - 4);
        unsigned sl = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int col = ix0 + j;
            sl |= (unsigned)((rowok && (unsigned)col < (unsigned)a.W) ? col - ixc : 4) << (8 * j);
        }
        const int goff = rowok ? iy * a.W + ixc : 0;
        // every channel slot is loaded (missing ones re-read the last real channel and are zeroed afterwards): no control flow
        // between the loads, so all of a unit's requests are in flight together
        unsigned d[4], x[4], v[3];
#pragma unroll
        for (int c = 0; c < 3; c++) __builtin_memcpy(&v[c], xin + (size_t)min(c, a.C - 1) * HW + goff, 4);
        unsigned v3 = 0u;
        if (a.C > 3) __builtin_memcpy(&v3, xin + (size_t)3 * HW + goff, 4);          // (uniform; the 4-channel case only)
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const unsigned vv = c < 3 ? v[c] : v3;
            d[c] = c < a.C ? __builtin_amdgcn_perm(padw, vv ^ 0x80808080u, sl) : 0u;
        }
        transpose4x4(d, x);
        *reinterpret_cast<uint4*>(smem + (size_t)(r * Wp4 + 4 * xq) * 4) = make_uint4(x[0], x[1], x[2], x[3]);
    }
    __syncthreads();

    // ---- per column tile of 16 pixels: fragment = taps 4q .. 4q+3 of the lane's pixel ------------------------------------------------
    int toff[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { const int tp = 4 * q + i; toff[i] = tp < 9 ? ((tp / 3) * Wp4 + tp % 3) * 4 : -1; }
#pragma unroll
    for (int ct = 0; ct < 4; ct++) {
        const int jloc = wave * 64 + ct * 16 + l15;
        int dy, dx;
        if (a.pool.on) { const int w = jloc >> 2, wy = w >> 3, wx = w & 7; dy = 2 * wy + ((jloc >> 1) & 1); dx = 2 * wx + (jloc & 1); }
        else { dy = jloc >> 4; dx = jloc & 15; }
        const bool live = oyb + dy <= oy1 && oxb + dx <= ox1;
        const int oy = min(oyb + dy, oy1), ox = min(oxb + dx, ox1);
        const int pp0 = ((oy - oyb) * a.SH * Wp4 + (ox - oxb) * a.SW) * 4;
        v4i_q bf;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const unsigned v = *reinterpret_cast<const unsigned*>(smem + pp0 + (toff[i] < 0 ? 0 : toff[i]));
            bf[i] = toff[i] < 0 ? 0 : (int)v;
        }
        int cs = 0;
        if (a.i_beta != 0) {
#pragma unroll
            for (int i = 0; i < 4; i++) cs = __builtin_amdgcn_sdot4(bf[i], 0x01010101, cs, false);
            cs += __shfl_xor(cs, 16);
            cs += __shfl_xor(cs, 32);
        }
        const int corr = a.i_beta * cs;
        const int opix = oy * a.OW + ox, ppix = (oy >> 1) * (a.OW >> 1) + (ox >> 1);
#pragma unroll
        for (int r = 0; r < RT; r++) {
            v4i_q acc = {0, 0, 0, 0};
            acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[r], bf, acc, 0, 0, 0);
            const int cb = r * 16 + 4 * q;                               // D: col = l15 (pixel), rows 4q .. 4q+3 (channels)
            const int4 cv = cvr[r];
            const int cvs[4] = {cv.x, cv.y, cv.z, cv.w};
            int qq[4], qb[4], pb[4];
#pragma unroll
            for (int e = 0; e < 4; e++) qq[e] = u8i_requant(acc[e] - corr + cvs[e], rq);
            if (a.relu.on) {
#pragma unroll
                for (int e = 0; e < 4; e++) qb[e] = lut[qq[e]];
            } else {
#pragma unroll
                for (int e = 0; e < 4; e++) qb[e] = qq[e];
            }
            const bool full = live && (!a.pool.on || a.pool.write_full);
            uint8_t* dst = a.y + (size_t)n * a.out_img + (size_t)(a.out_c0 + cb) * OHW + opix;
#pragma unroll
            for (int e = 0; e < 4; e++)
                if (full && cb + e < a.cout) dst[(size_t)e * OHW] = (uint8_t)qb[e];
            if (a.pool.on) {
#pragma unroll
                for (int e = 0; e < 4; e++) pb[e] = lut[256 + quad_max(qb[e])];
                const bool pst = live && (l15 & 3) == 0;
                uint8_t* pd = a.pool.y + (size_t)n * a.pool.out_img + (size_t)(a.pool.out_c0 + cb) * (OHW >> 2) + ppix;
#pragma unroll
                for (int e = 0; e < 4; e++)
                    if (pst && cb + e < a.cout) pd[(size_t)e * (OHW >> 2)] = (uint8_t)pb[e];
            }
        }
    }
}

bool conv_u8i_rgb_applicable(const U8ConvArgs& a, int KH, int KW, int DH, int DW)
{
    if (a.C < 1 || a.C > 4 || KH != 3 || KW != 3 || DH != 1 || DW != 1 || a.cout > 64 || a.W < 4) return false;
    if (a.pool.on && (((a.OH | a.OW) & 1) || (a.OH * a.OW) % 8 != 0)) return false;
    if (a.SH < 1 || a.SW < 1 || a.SH > 2 || a.SW > 2) return false;      // patch of a 16x16 window: at most 33 x 36 pixels
    return true;
}
void conv_u8i_rgb_prepare(U8ConvArgs& a) { a.i_npad = (15 * a.SH + 3) * ((15 * a.SW + 3 + 3) & ~3); }      // patch pixels of a full window
size_t conv_u8i_rgb_packed_bytes(const U8ConvArgs& a) { return (size_t)((a.cout + 15) / 16) * 1024; }
// fragment r: lane (row l15, quarter q), byte i*4 + c = w'[16r + l15][c][tap 4q + i] (zero outside the real taps / channels);
// cvec[co] = bias - alpha * sum_real w' + 9 * C * alpha * beta
void conv_u8i_rgb_pack(const U8ConvArgs& a, const uint8_t* w, int w_zp, int in_zp, const int32_t* bias, int8_t* out, int32_t* cvec)
{
    const int rt = (a.cout + 15) / 16, alpha = in_zp - 128, beta = w_zp - 128;
    std::fill(out, out + conv_u8i_rgb_packed_bytes(a), (int8_t)0);
    for (int co = 0; co < rt * 16; co++) {
        long w1 = 0;
        if (co < a.cout)
            for (int c = 0; c < a.C; c++)
                for (int tp = 0; tp < 9; tp++) {
                    const int8_t v = (int8_t)(uint8_t)(w[((size_t)co * a.C + c) * 9 + tp] ^ 0x80);
                    out[(size_t)(co / 16) * 1024 + ((tp / 4) * 16 + (co & 15)) * 16 + (tp % 4) * 4 + c] = v;
                    w1 += v;
                }
        cvec[co] = (int32_t)((co < a.cout && bias ? (long)bias[co] : 0L) - (long)alpha * w1 + (long)9 * a.C * alpha * beta);
    }
}
hipError_t launch_conv_u8i_rgb(const U8ConvArgs& a, hipStream_t s)
{
    const int tiles = ((a.OW + 15) / 16) * ((a.OH + 15) / 16);
    const size_t lds = (size_t)a.i_npad * 4 + 512;                 // the patch of a full 16x16 window + the two byte tables
    const dim3 grid(tiles * a.N);
    switch ((a.cout + 15) / 16) {
    case 1: hipLaunchKernelGGL(conv_u8i_rgb_k<1>, grid, dim3(256), lds, s, a); break;
    case 2: hipLaunchKernelGGL(conv_u8i_rgb_k<2>, grid, dim3(256), lds, s, a); break;
    case 3: hipLaunchKernelGGL(conv_u8i_rgb_k<3>, grid, dim3(256), lds, s, a); break;
    default: hipLaunchKernelGGL(conv_u8i_rgb_k<4>, grid, dim3(256), lds, s, a); break;
    }
    return hipGetLastError();
}

// =================================================================================================================
// Pointwise layers (1x1, stride 1, no padding) without LDS and without barriers.
//
// The matrix core wants 16 consecutive K bytes of ONE pixel per lane, NCHW has the channels H*W bytes apart -- conv_u8i_k pays
// for that with a transposing LDS stage.  A 1x1 convolution has no taps, so the transpose can stay in registers: a lane loads
// 16 channels x 4 CONSECUTIVE pixels as 16 dwords (lanes run along the pixel quads: every load instruction of a half-wave
// covers 128 contiguous bytes of one channel row), four 4x4 byte transposes turn them into four 16-byte B fragments, one per
// pixel of the quad, and FOUR MFMAs consume them -- column l31 of MFMA j is pixel 4*l31 + j of the wave's 128-pixel strip.
// Which pixel a matrix column stands for is free, so nothing has to be moved: after the K loop a lane holds, for each of its 16
// output channels per 32-row tile, the results of four consecutive pixels = ONE dword of the NCHW output row, and the
// half-wave's stores are 128 contiguous bytes again.  No shared memory traffic, no barrier, loads one K step ahead; the weights
// stream through the same fragment order (and the same 8-deep register ring) as conv_u8i_k.  Arithmetic and epilogue: identical
// to conv_u8i_k (exact int32 sums, column sums by v_dot4, cvec, the reference's requantisation).
// A quad that hangs over the end of an image plane reads up to 3 bytes of whatever follows (the next channel, the next tensor or
// the allocation's slack -- graph_plan.hip: dev_alloc) and stores byte-wise.
// =================================================================================================================
template <int WP, int WC, int TMC>
__global__ __launch_bounds__(256) void conv_u8i_pw_k(const U8ConvArgs a)
{
    static_assert(WP * WC == 4, "four waves");
    constexpr int BMB = WC * TMC * 32, BNB = WP * 128, D = 4;
    __shared__ uint8_t lut[256];
    const int t = threadIdx.x, lane = t & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wp = wave % WP, wc = wave / WP;
    const int HW = a.H * a.W;                                          // == OH * OW
    const int tiles = (HW + BNB - 1) / BNB;
    const int n = blockIdx.x / tiles, tile = blockIdx.x - n * tiles;
    const int co0 = blockIdx.y * BMB + wc * TMC * 32;
    lut[t] = a.relu.on ? fused_relu((uint8_t)t, a.out_scale, a.out_zp, a.relu) : (uint8_t)t;      // the fused ReLU node as a byte table
    __syncthreads();
    const U8IRq rq{a.i_m, a.out_zp, a.i_qlo, a.i_qhi};
    const int pix0 = tile * BNB + wp * 128 + 4 * l31;                  // first pixel of this lane's quad
    const int nvalid = min(max(HW - pix0, 0), 4);
    const uint8_t* xq = a.x + (size_t)n * a.C * HW + (nvalid ? pix0 : 0);

    const int ns = a.i_nchunks;                                        // K steps of 32 channels
    const int8_t* wt = a.iw + ((size_t)blockIdx.y * ns * (BMB / 32) + wc * TMC) * 1024 + lane * 16;
    v4i_q ar[D][TMC];
    auto load_a = [&](v4i_q (&f)[TMC]) {
#pragma unroll
        for (int i = 0; i < TMC; i++) f[i] = *reinterpret_cast<const v4i_q*>(wt + i * 1024);
        wt += (BMB / 32) * 1024;
    };
#pragma unroll
    for (int d = 0; d < D; d++) load_a(ar[d]);
    // the per-channel constants of the epilogue, requested NOW: fetched where they are used each was a round trip to the L2 in
    // front of every group of stores (the loop cannot hide it: the stores may alias, the compiler keeps the loads behind them)
    int4 cvr[TMC][4];
#pragma unroll
    for (int i = 0; i < TMC; i++)
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) cvr[i][g4] = *reinterpret_cast<const int4*>(a.icv + co0 + i * 32 + 8 * g4 + 4 * hi);
    unsigned xv[2][16];
    auto load_x = [&](unsigned (&v)[16], int st) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int c = min(st * 32 + hi * 16 + i, a.C - 1);         // padded channels re-read the last one: they meet w' = beta
            if (a.i_dbg & 4) v[i] = 0x01020304u * (unsigned)(c + 1);    // anatomy: no input loads
            else __builtin_memcpy(&v[i], xq + (size_t)c * HW, 4);
        }
    };
    load_x(xv[0], 0);

    v16i_q acc[TMC][4];
#pragma unroll
    for (int i = 0; i < TMC; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0;
    int ssum[4] = {0, 0, 0, 0};
    const bool need_sum = a.i_beta != 0;

    for (int s0 = 0; s0 < ns; s0 += D) {
        static_for_u8i<0, D>([&](auto Dd) {
            constexpr int d = decltype(Dd)::value;
            const int st = s0 + d;
            if (st < ns) {
                if (st + 1 < ns) load_x(xv[(d + 1) & 1], st + 1);
                v4i_q F[4];
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const unsigned dd[4] = {xv[d & 1][4 * g] ^ 0x80808080u, xv[d & 1][4 * g + 1] ^ 0x80808080u, xv[d & 1][4 * g + 2] ^ 0x80808080u,
                                            xv[d & 1][4 * g + 3] ^ 0x80808080u};
                    unsigned x[4];
                    transpose4x4(dd, x);                               // x[p] = pixel p's channels 4g .. 4g+3
#pragma unroll
                    for (int p = 0; p < 4; p++) F[p][g] = (int)x[p];
                }
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int i = 0; i < TMC; i++) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ar[d][i], F[j], acc[i][j], 0, 0, 0);
                if (need_sum) {
#pragma unroll
                    for (int j = 0; j < 4; j++)
#pragma unroll
                        for (int g = 0; g < 4; g++) ssum[j] = __builtin_amdgcn_sdot4(F[j][g], 0x01010101, ssum[j], false);
                }
                load_a(ar[d]);
            }
        });
    }

    // ---- epilogue: lane (l31, hi) holds, per 32-row tile i and register e, channel row(e, hi) at pixels pix0 .. pix0+3 -----------------
    int corr[4];
#pragma unroll
    for (int j = 0; j < 4; j++) corr[j] = a.i_beta * (ssum[j] + __shfl_xor(ssum[j], 32));
    const int OHW = HW;
    uint8_t* yb = a.y + (size_t)n * a.out_img + (size_t)a.out_c0 * OHW + pix0;
#pragma unroll
    for (int i = 0; i < TMC; i++)
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) {
            const int cb = co0 + i * 32 + 8 * g4 + 4 * hi;
            const int4 cv = cvr[i][g4];
            const int cvs[4] = {cv.x, cv.y, cv.z, cv.w};
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int co = cb + e;
                int qv[4];
                unsigned pk;
                if (a.i_dbg & 2) { for (int j = 0; j < 4; j++) qv[j] = (acc[i][j][4 * g4 + e] + cvs[e]) & 255; }      // anatomy: no requantisation arithmetic
                else {
#pragma unroll
                    for (int j = 0; j < 4; j++) qv[j] = u8i_requant(acc[i][j][4 * g4 + e] - corr[j] + cvs[e], rq);
                }
                if (a.relu.on) pk = (unsigned)lut[qv[0]] | (unsigned)lut[qv[1]] << 8 | (unsigned)lut[qv[2]] << 16 | (unsigned)lut[qv[3]] << 24;
                else pk = (unsigned)qv[0] | (unsigned)qv[1] << 8 | (unsigned)qv[2] << 16 | (unsigned)qv[3] << 24;
                uint8_t* dst = yb + (size_t)co * OHW;
                const bool st_ok = co < a.cout && !((a.i_dbg & 1) && pk != 0x4d4d4d4du);               // (anatomy: (almost) no stores)
                if (st_ok && nvalid == 4) __builtin_memcpy(dst, &pk, 4);
                if (st_ok && nvalid < 4)
                    for (int j = 0; j < nvalid; j++) dst[j] = (uint8_t)(pk >> (8 * j));
            }
        }
}

static const struct { int wp, wc, tmc; const char* name; } U8IPW_CFGS[] = {
    {4, 1, 2, "conv_u8i_pw_64x512"}, {2, 2, 2, "conv_u8i_pw_128x256"}, {4, 1, 1, "conv_u8i_pw_32x512"}, {2, 2, 1, "conv_u8i_pw_64x256"},
    {1, 4, 2, "conv_u8i_pw_256x128"}, {1, 4, 1, "conv_u8i_pw_128x128"}};
int conv_u8i_pw_num_cfgs() { return 6; }
int conv_u8i_pw_bm(int cfg) { return U8IPW_CFGS[cfg].wc * U8IPW_CFGS[cfg].tmc * 32; }
int conv_u8i_pw_bn(int cfg) { return U8IPW_CFGS[cfg].wp * 128; }
const char* conv_u8i_pw_kernel_name(const U8ConvArgs& a) { return U8IPW_CFGS[a.i_cfg].name; }
// 1x1 / stride 1 / no padding / no fused pool; fills i_cfg, i_nchunks and the filter shape
bool conv_u8i_pw_prepare(U8ConvArgs& a, int cfg, int KH, int KW)
{
    if (cfg < 0 || cfg >= conv_u8i_pw_num_cfgs()) return false;
    if (KH != 1 || KW != 1 || a.SH != 1 || a.SW != 1 || a.PH != 0 || a.PW != 0 || a.pool.on || a.OH != a.H || a.OW != a.W) return false;
    if ((size_t)a.C * a.H * a.W >= (1u << 30)) return false;
    const int bm = conv_u8i_pw_bm(cfg);
    if (bm > 32 && a.cout <= bm / 2) return false;
    a.pk_kh = a.pk_kw = 1; a.pk_dh = a.pk_dw = 1;
    a.i_nchunks = (a.C + 31) / 32;
    a.i_cfg = cfg; a.i_tw = 0; a.i_cgs = 0; a.i_npad = 0;
    a.i_dbg = exp_env("TAMD_U8I_ABLATE") ? atoi(exp_env("TAMD_U8I_ABLATE")) : 0;
    return true;
}
hipError_t launch_conv_u8i_pw(const U8ConvArgs& a, hipStream_t s)
{
    const int bm = conv_u8i_pw_bm(a.i_cfg), bn = conv_u8i_pw_bn(a.i_cfg), HW = a.H * a.W;
    const dim3 grid(((HW + bn - 1) / bn) * a.N, (a.cout + bm - 1) / bm, 1);
    switch (a.i_cfg) {
    case 0: hipLaunchKernelGGL((conv_u8i_pw_k<4, 1, 2>), grid, dim3(256), 0, s, a); break;
    case 1: hipLaunchKernelGGL((conv_u8i_pw_k<2, 2, 2>), grid, dim3(256), 0, s, a); break;
    case 2: hipLaunchKernelGGL((conv_u8i_pw_k<4, 1, 1>), grid, dim3(256), 0, s, a); break;
    case 3: hipLaunchKernelGGL((conv_u8i_pw_k<2, 2, 1>), grid, dim3(256), 0, s, a); break;
    case 4: hipLaunchKernelGGL((conv_u8i_pw_k<1, 4, 2>), grid, dim3(256), 0, s, a); break;
    default: hipLaunchKernelGGL((conv_u8i_pw_k<1, 4, 1>), grid, dim3(256), 0, s, a); break;
    }
    return hipGetLastError();
}

// ---- host side ---------------------------------------------------------------------------------------------------------------
static const struct { int wm, wn, tm, tn; const char* name; } U8I_CFGS[] = {
    {2, 2, 1, 1, "conv_u8i_64x64"}, {2, 2, 2, 1, "conv_u8i_128x64"}, {2, 2, 1, 2, "conv_u8i_64x128"}, {2, 2, 2, 2, "conv_u8i_128x128"},
    {1, 4, 1, 1, "conv_u8i_32x128"}, {1, 4, 1, 2, "conv_u8i_32x256"}};
int conv_u8i_num_cfgs() { return 6; }
int conv_u8i_bm(int cfg) { return U8I_CFGS[cfg].wm * U8I_CFGS[cfg].tm * 32; }
static int u8i_bn(int cfg) { return U8I_CFGS[cfg].wn * U8I_CFGS[cfg].tn * 32; }
const char* conv_u8i_kernel_name(const U8ConvArgs& a) { return U8I_CFGS[a.i_cfg].name; }

// patch pixels of the worst pixel tile of `bn` outputs (rows x pitch rounded to 4)
static int u8i_patch_pixels(const U8ConvArgs& a, int bn)
{
    const int OHW = a.OH * a.OW;
    int worst = 0;
    for (int j0 = 0; j0 < OHW; j0 += bn) {
        const int jl = std::min(j0 + bn, OHW) - 1;
        int oy0, oy1, ox0, ox1;
        u8i_box(a.OW, a.pool.on, j0, jl, &oy0, &oy1, &ox0, &ox1);
        const int wp4 = ((ox1 - ox0) * a.SW + (a.pk_kw - 1) * a.pk_dw + 1 + 3) & ~3;
        const int rows = (oy1 - oy0) * a.SH + (a.pk_kh - 1) * a.pk_dh + 1;
        worst = std::max(worst, rows * wp4);
    }
    return worst;
}

// fills i_cfg / i_npad / i_nchunks and the filter shape; false: this tile configuration cannot take the layer
bool conv_u8i_prepare(U8ConvArgs& a, int cfg, int KH, int KW, int DH, int DW)
{
    if (cfg < 0 || cfg >= conv_u8i_num_cfgs()) return false;
    if (a.W < 4 || a.C < 1 || (size_t)a.C * a.H * a.W >= (1u << 30)) return false;      // the staging loads are dwords inside an image row
    if (a.pool.on && ((a.OH | a.OW) & 1)) return false;
    a.pk_kh = KH; a.pk_kw = KW; a.pk_dh = DH; a.pk_dw = DW;
    const int bm = conv_u8i_bm(cfg), bn = u8i_bn(cfg);
    if (bm > 32 && a.cout <= bm / 2) return false;                                   // a tile half empty: the narrower shape exists
    if (a.pool.on && (a.OH * a.OW) % 8 != 0) return false;                           // what the planner promises the fused pool
    // linear tiles where their worst bounding box fits (at most 4 staging units per thread = 512 patch pixels), else 2-D tiles
    int worst = u8i_patch_pixels(a, bn);
    a.i_tw = 0;
    const char* tm = tamd_pin("u8i_tiles");                                       // tests: 2 = 2-D tiles wherever they fit
    if (worst > 512 || (tm && atoi(tm) == 2)) {
        const int tw = bn >= 128 ? 16 : 8, th = bn / tw;
        const int w2 = ((th - 1) * a.SH + (KH - 1) * DH + 1) * (((tw - 1) * a.SW + (KW - 1) * DW + 1 + 3) & ~3);
        if (w2 > 512) { if (worst > 512) return false; }
        else { worst = w2; a.i_tw = tw; }
    }
    a.i_npad = worst <= 128 ? 128 : worst <= 256 ? 256 : 512;
    a.i_nchunks = (a.C + 31) / 32;                                                   // 32-channel groups
    a.i_dbg = exp_env("TAMD_U8I_ABLATE") ? atoi(exp_env("TAMD_U8I_ABLATE")) : 0;       // anatomy runs (tools/exp): wrong bytes by design
    // groups per chunk (1 | 2 | 4): as many as the staging budget (4 units per thread = 512 patch pixels x groups) and the layer hold
    // (default ONE: measured faster than 2 / 4 on both uint8 configs -- the larger staging burst stalls the MFMAs longer than the
    // saved barriers cost; TAMD_U8I_CG=2|4 asks for more where the budget allows, tests run them)
    int cg_max = 0;
    while (cg_max < 2 && (2 << cg_max) * a.i_npad <= 512 && (2 << cg_max) <= a.i_nchunks) cg_max++;
    a.i_cgs = 0;
    if (const char* cg = tamd_pin("u8i_cg")) a.i_cgs = std::min(cg_max, atoi(cg) >= 4 ? 2 : atoi(cg) >= 2 ? 1 : 0);
    a.i_cfg = cfg;
    return true;
}

// (w ^ 0x80) in fragment order, padded with the weight zero point (w' = beta); cvec[co] = bias - alpha * sum_k w'_k + Kp * alpha * beta
size_t conv_u8i_packed_bytes(const U8ConvArgs& a, int bm)
{
    const int ntile = (a.cout + bm - 1) / bm;
    return ((size_t)ntile * a.i_nchunks * a.pk_kh * a.pk_kw + 10) * bm * 32;         // + readable slack for the ring's last prefetches (D + 1 steps)
}
void conv_u8i_pack(const U8ConvArgs& a, int bm, const uint8_t* w, int w_zp, int in_zp, const int32_t* bias, int8_t* out, int32_t* cvec)
{
    const int ntile = (a.cout + bm - 1) / bm, ntaps = a.pk_kh * a.pk_kw, ns = a.i_nchunks * ntaps;
    const int alpha = in_zp - 128, beta = w_zp - 128;
    const int8_t padb = (int8_t)(uint8_t)(w_zp ^ 0x80);
    std::fill(out, out + conv_u8i_packed_bytes(a, bm), padb);
    for (int co = 0; co < ntile * bm; co++) {
        long w1 = 0;
        for (int s = 0; s < ns; s++) {
            const int cc = s / ntaps, tap = s % ntaps;
            int8_t* frag = out + (((size_t)(co / bm) * ns + s) * (bm / 32) + (co % bm) / 32) * 1024;
            for (int kb = 0; kb < 32; kb++) {
                const int c = cc * 32 + kb;
                int8_t v = padb;
                if (co < a.cout && c < a.C) v = (int8_t)(uint8_t)(w[((size_t)co * a.C + c) * ntaps + tap] ^ 0x80);
                frag[((kb >> 4) * 32 + (co & 31)) * 16 + (kb & 15)] = v;
                w1 += v;
            }
        }
        cvec[co] = (int32_t)((co < a.cout && bias ? (long)bias[co] : 0L) - (long)alpha * w1 + (long)ns * 32 * alpha * beta);
    }
}

hipError_t launch_conv_u8i(const U8ConvArgs& a, hipStream_t s)
{
    const int bm = conv_u8i_bm(a.i_cfg), bn = u8i_bn(a.i_cfg), OHW = a.OH * a.OW;
    const int tiles = a.i_tw ? ((a.OW + a.i_tw - 1) / a.i_tw) * ((a.OH + bn / a.i_tw - 1) / (bn / a.i_tw)) : (OHW + bn - 1) / bn;
    const dim3 grid(tiles * a.N, (a.cout + bm - 1) / bm, 1);
    const size_t lds = (size_t)2 * (2 << a.i_cgs) * a.i_npad * 16 + 512;          // two patch buffers + the two byte tables
    const int units = (a.i_npad << a.i_cgs) / 128;                                  // staging units per thread: 1 | 2 | 4
    auto go = [&](auto kern) {
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
        return hipGetLastError();
    };
#define U8I_NU(WM, WN, TM, TN)                                                      \
    switch (units) {                                                                \
    case 1: return go(conv_u8i_k<WM, WN, TM, TN, 1>);                               \
    case 2: return go(conv_u8i_k<WM, WN, TM, TN, 2>);                               \
    default: return go(conv_u8i_k<WM, WN, TM, TN, 4>);                              \
    }
    switch (a.i_cfg) {
    case 0: U8I_NU(2, 2, 1, 1)
    case 1: U8I_NU(2, 2, 2, 1)
    case 2: U8I_NU(2, 2, 1, 2)
    case 3: U8I_NU(2, 2, 2, 2)
    case 4: U8I_NU(1, 4, 1, 1)
    default: U8I_NU(1, 4, 1, 2)
    }
#undef U8I_NU
}

}  // namespace tamd
