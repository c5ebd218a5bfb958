// uint8 convolution, the special members: pointwise (conv_u8_pw), 3-channel-class 3x3 (conv_u8_c3), RGB first layers (conv_u8_rgb3x3).
// The contract every uint8 kernel follows is stated at the top of u8_kernels.hip.
#include <hip/hip_runtime.h>
#include "env.h"

#include <cstdlib>
#include <type_traits>
#include <algorithm>

#include "kernels.h"
#include "u8_epilogue.h"
#include "u8_patch_tail.h"

namespace tamd {

typedef float v4f __attribute__((ext_vector_type(4)));
// =================================================================================================================
// Shallow pointwise layers of large maps (MobileNet-SSD conv1 / conv2: 1x1, stride 1, K = 32 | 64 on 150^2 / 75^2 maps):
// K is one or two patch chunks, so a conv_u8_patch block is all prologue and epilogue there.  Here a WAVE is the unit: it keeps
// the weight fragments of its 16*TM output channels for the WHOLE K in registers (the conv_u8_patch fragment stream, read once),
// walks 16-pixel column tiles of the batch grid-stride, reads the B operand straight from the NCHW input -- lane (pixel l15,
// k%4 = kq) needs the bytes of channels 4s + kq of its pixel: KS byte loads per tile, requested one tile ahead -- converts them in
// the shadow of the previous MFMAs and issues the chain in ascending k (the reference's order, conv_u8_body's header).  No LDS, no
// barrier, no block-level cooperation.  Tail pixels: conv_u8_patch_tail blocks behind the main grid, as in conv_u8_patch.
// =================================================================================================================
template <int TM, int KS>
__global__ __launch_bounds__(256) void conv_u8_pw_k(const U8ConvArgs a, int main_blocks)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];          // used by the tail blocks only
    __shared__ uint8_t tail[512];                   // fused ReLU node as a byte table (u8_epilogue.h)
    u8_tail_tables(tail, threadIdx.x, 256, a.relu, a.out_scale, a.out_zp, a.pool);
    if ((int)blockIdx.x >= main_blocks) {
        conv_u8_patch_tail<1>(a, smem, blockIdx.x - main_blocks, tail);
        return;
    }
    if (a.relu.on) __syncthreads();                 // the only barrier of a main block: the table before the first look-up (uniform)
    const float rq_inv = __fdiv_rn(1.0f, a.out_scale);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, kq = lane >> 4;
    const int OHW = a.OH * a.OW, N8 = OHW & ~7, PTI = (N8 + 15) / 16, CT = (a.cout + 16 * TM - 1) / (16 * TM);
    const int nwaves = main_blocks * 4, gw = blockIdx.x * 4 + wave;
    const int ct = gw % CT, lanes_of_ct = (nwaves - ct + CT - 1) / CT;        // waves that share this cout tile stride over the pixel tiles
    const int co0 = ct * 16 * TM;
    // ---- the weights of this wave: [tile16][super-step of 4 MFMA steps][lane][float4] (conv_u8_patch_pack, 1x1) -------------------
    constexpr int NSS = KS / 4;
    float4 af[TM][NSS];
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const float* wb = reinterpret_cast<const float*>(a.wpk) + (size_t)(co0 / 16 + i) * NSS * 256;
#pragma unroll
        for (int ss = 0; ss < NSS; ss++) af[i][ss] = *reinterpret_cast<const float4*>(wb + ss * 256 + lane * 4);
    }
    const int total = a.N * PTI;
    auto tile_ptr = [&](int t, int* n, int* pj) __attribute__((always_inline)) {
        *n = t / PTI;
        *pj = (t - *n * PTI) * 16 + l15;
    };
    unsigned bq[2][KS];                                  // raw bytes of the tile in flight / the tile being computed
    auto bload = [&](auto D, int t) __attribute__((always_inline)) {
        constexpr int d = decltype(D)::value;
        int n, pj;
        tile_ptr(t < total ? t : total - 1, &n, &pj);
        const uint8_t* xp = a.x + (size_t)n * a.C * OHW + (size_t)kq * OHW + (pj < N8 ? pj : N8 - 1);
#pragma unroll
        for (int s = 0; s < KS; s++) bq[d][s] = xp[(size_t)(4 * s) * OHW];
    };
    auto compute = [&](auto D, int t) __attribute__((always_inline)) {
        constexpr int d = decltype(D)::value;
        v4f acc[TM];
#pragma unroll
        for (int i = 0; i < TM; i++) acc[i] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; s++) {
            const float bf = dequant((uint8_t)bq[d][s], a.in_zp, a.in_scale);
#pragma unroll
            for (int i = 0; i < TM; i++) {
                const float4 f = af[i][s >> 2];
                const float av = (s & 3) == 0 ? f.x : (s & 3) == 1 ? f.y : (s & 3) == 2 ? f.z : f.w;
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bf, acc[i], 0, 0, 0);
            }
        }
        int n, pj;
        tile_ptr(t, &n, &pj);
        if (pj >= N8) return;
#pragma unroll
        for (int i = 0; i < TM; i++) {
            const int co = co0 + i * 16 + 4 * kq;
            if (co >= a.cout) continue;
            const float s4[4] = {acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
            u8_finish4(a, s4, co, n, OHW, pj, 0, false, rq_inv, tail);       // (no fused pool on this kernel: row-major pixels, opix == pj)
        }
    };
    int t = gw / CT;                                     // this wave's first pixel tile; the next ones follow at a stride of lanes_of_ct
    if (t >= total) return;
    bload(std::integral_constant<int, 0>{}, t);
    for (; t < total; t += 2 * lanes_of_ct) {
        bload(std::integral_constant<int, 1>{}, t + lanes_of_ct);
        compute(std::integral_constant<int, 0>{}, t);
        if (t + lanes_of_ct >= total) break;
        bload(std::integral_constant<int, 0>{}, t + 2 * lanes_of_ct);
        compute(std::integral_constant<int, 1>{}, t + lanes_of_ct);
    }
}

// shallow pointwise layers: 1x1, stride 1, no padding, K in {32, 64} (at K = 128 the patch kernel and the staging GEMM are faster), no fused pool; the patch fields (pk_kh = 1, wpk in the
// 1x1 fragment order) must be prepared (conv_u8_patch_prepare with any configuration)
bool conv_u8_pw_applicable(const U8ConvArgs& a, int KH, int KW)
{
    const char* env = tamd_pin("u8_pw");
    if (env && atoi(env) == 0) return false;
    return KH == 1 && KW == 1 && a.SH == 1 && a.SW == 1 && a.PH == 0 && a.PW == 0 && a.H == a.OH && a.W == a.OW && !a.pool.on
           && (a.K == 32 || a.K == 64) && (a.OH * a.OW & ~7) >= 16 && (size_t)a.C * a.H * a.W < (1u << 31);
}

const char* conv_u8_pw_kernel_name(const U8ConvArgs& a) { return a.K == 32 ? "conv_u8_pw<k32>" : "conv_u8_pw<k64>"; }

hipError_t launch_conv_u8_pw(const U8ConvArgs& a, hipStream_t s)
{
    const int OHW = a.OH * a.OW, N8 = OHW & ~7, PTI = (N8 + 15) / 16;
    const int tm = 4, CT = (a.cout + 16 * tm - 1) / (16 * tm);
    const long items = (long)a.N * PTI * CT;             // (pixel tile, cout tile) pairs: four per block, at most ~8 blocks per CU
    const int main_blocks = (int)std::min<long>((items + 3) / 4, 2048);
    const int tail_blocks = (OHW - N8) * a.N * ((a.cout + 63) / 64);
    const size_t lds = tail_blocks ? (size_t)a.K * 4 : 0;
    const dim3 grid(main_blocks + tail_blocks, 1, 1);
    if (a.K == 32) hipLaunchKernelGGL((conv_u8_pw_k<4, 8>), grid, dim3(256), lds, s, a, main_blocks);
    else hipLaunchKernelGGL((conv_u8_pw_k<4, 16>), grid, dim3(256), lds, s, a, main_blocks);
    return hipGetLastError();
}

// =================================================================================================================
// Shallow 3x3 layers of large maps (YOLOv3-tiny conv1 / conv2: 16 -> 32 @ 208^2, 32 -> 64 @ 104^2; K = 144 | 288): K is one or two
// patch chunks for 32 .. 64 output channels, so a conv_u8_patch block -- and a staging-GEMM block -- is all prologue and epilogue
// there (88 / 66 us for 1.6 GMAC each; profiles/r04_experiment_u8_patch_2d_tiles.txt).  conv_u8_pw's shape with a 3x3 gather (round 4):
// a WAVE keeps the weight fragments of its 16 * TM output channels for the WHOLE K in registers (the patch kernel's fragment stream,
// read once: 9 floats per 16-row tile and super-step of 4 channels), walks 16-pixel tiles of the batch (window-major under a fused
// pool, as everywhere), and gathers the B operand straight from the NCHW input: lane (pixel l15, kq) needs tap k = 4 s + kq of its
// pixel for the nine steps of a super-step -- the (dy, dx, channel-in-group) of those nine k are the same in every super-step, so
// nine per-lane offsets + a plane stride name them all; out-of-image taps enter as 0.0f; the bytes of super-step ss + 1 are requested
// while ss multiplies.  Chain order: accumulator tile i receives its k in ascending steps of four (conv_u8_body's header) -- the
// reference's single chain of a main pixel.  Tail pixels: conv_u8_patch_tail blocks behind the main grid.  No LDS operands, no barrier
// in the loop.
// =================================================================================================================
template <int TM, int NSS>
__global__ __launch_bounds__(256) void conv_u8_c3_k(const U8ConvArgs a, int main_blocks)
{
    constexpr int SS = 9, FRAG = SS * 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];          // used by the tail blocks only
    __shared__ uint8_t tail[512];                   // fused ReLU / pool nodes as byte tables (u8_epilogue.h)
    u8_tail_tables(tail, threadIdx.x, 256, a.relu, a.out_scale, a.out_zp, a.pool);
    if ((int)blockIdx.x >= main_blocks) {
        conv_u8_patch_tail<3>(a, smem, blockIdx.x - main_blocks, tail);
        return;
    }
    __syncthreads();                                // the only barrier of a main block: the tables before the first look-up
    const float rq_inv = __fdiv_rn(1.0f, a.out_scale);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, kq = lane >> 4;
    const int OHW = a.OH * a.OW, N8 = OHW & ~7, PTI = (N8 + 15) / 16, CT = (a.cout + 16 * TM - 1) / (16 * TM);
    const int nwaves = main_blocks * 4, gw = blockIdx.x * 4 + wave;
    const int ct = gw % CT, stride = nwaves / CT;    // (main_blocks * 4 is a multiple of CT: the launcher rounds it)
    const int co0 = ct * 16 * TM;
    // ---- the weights of this wave: [tile16][super-step][2 float4 groups][lane] + [lane] (conv_u8_patch_pack, 3x3) ------------------
    float4 af4[TM][NSS][2];
    float afr[TM][NSS];
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const float* wb = reinterpret_cast<const float*>(a.wpk) + (size_t)(co0 / 16 + i) * NSS * FRAG;
#pragma unroll
        for (int ss = 0; ss < NSS; ss++) {
            af4[i][ss][0] = *reinterpret_cast<const float4*>(wb + ss * FRAG + lane * 4);
            af4[i][ss][1] = *reinterpret_cast<const float4*>(wb + ss * FRAG + 256 + lane * 4);
            afr[i][ss] = wb[ss * FRAG + 512 + lane];
        }
    }
    // ---- the nine taps k = 4 s + kq of a super-step: channel-in-group, dy, dx --------------------------------------------------
    const int chw = a.H * a.W;
    int toff[SS], tdy[SS], tdx[SS];
#pragma unroll
    for (int s = 0; s < SS; s++) {
        const int kl = 4 * s + kq, cl = kl / 9, tap = kl - 9 * cl, dy = tap / 3, dx = tap - 3 * dy;
        toff[s] = cl * chw + dy * a.W + dx;
        tdy[s] = dy; tdx[s] = dx;
    }
    const int total = a.N * PTI;
    struct TileIn { const uint8_t* img; int base; unsigned okm; int n, oy, ox, pj; };
    auto locate = [&](int t, TileIn& ti) __attribute__((always_inline)) {
        const int tc = t < total ? t : total - 1;
        ti.n = tc / PTI;
        ti.pj = (tc - ti.n * PTI) * 16 + l15;
        conv_pixel(a, ti.pj < N8 ? ti.pj : N8 - 1, &ti.oy, &ti.ox);
        const int iy0 = ti.oy * a.SH - a.PH, ix0 = ti.ox * a.SW - a.PW;
        ti.okm = 0;
#pragma unroll
        for (int s = 0; s < SS; s++)
            ti.okm |= (((unsigned)(iy0 + tdy[s]) < (unsigned)a.H) & ((unsigned)(ix0 + tdx[s]) < (unsigned)a.W)) ? 1u << s : 0u;
        ti.img = a.x + (size_t)ti.n * a.C * chw;
        ti.base = iy0 * a.W + ix0;                       // (negative at the top-left border: only ever added to an in-image tap)
    };
    unsigned raw[2][SS];
    auto bload = [&](auto D, const TileIn& ti, int ss) __attribute__((always_inline)) {
        constexpr int d = decltype(D)::value;
        // unconditional loads (an out-of-image tap reads the image's first byte and is replaced by 0.0f at conversion): written as
        // `ok ? load : 0` every load sits in its own divergent branch
        const int o = ti.base + 4 * ss * chw;
#pragma unroll
        for (int s = 0; s < SS; s++) raw[d][s] = ti.img[(ti.okm >> s & 1u) ? o + toff[s] : 0];
    };
    TileIn cur, nxt;
    int t = gw / CT;                                     // this wave's first pixel tile; the next ones follow at `stride`
    if (t >= total) return;
    locate(t, cur);
    bload(std::integral_constant<int, 0>{}, cur, 0);
    for (; t < total; t += stride) {
        locate(t + stride, nxt);
        v4f acc[TM];
#pragma unroll
        for (int i = 0; i < TM; i++) acc[i] = v4f{0.f, 0.f, 0.f, 0.f};
        u8_static_for<0, NSS>([&](auto SSI) {
            constexpr int ss = decltype(SSI)::value, d = ss & 1;
            // request the next super-step's bytes (or the next tile's first) before this one's are converted
            if constexpr (ss + 1 < NSS) bload(std::integral_constant<int, d ^ 1>{}, cur, ss + 1);
            else bload(std::integral_constant<int, d ^ 1>{}, nxt, 0);
#pragma unroll
            for (int s = 0; s < SS; s++) {
                const float bv = (cur.okm >> s & 1u) ? dequant((uint8_t)raw[d][s], a.in_zp, a.in_scale) : 0.f;
#pragma unroll
                for (int i = 0; i < TM; i++) {
                    const float av = s == 8 ? afr[i][ss] : (s & 3) == 0 ? af4[i][ss][s >> 2].x : (s & 3) == 1 ? af4[i][ss][s >> 2].y : (s & 3) == 2 ? af4[i][ss][s >> 2].z : af4[i][ss][s >> 2].w;
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i], 0, 0, 0);
                }
            }
        });
        static_assert((NSS & 1) == 0, "the tile in flight lands in register set 0 again");
        if (cur.pj < N8) {
            const int opix = cur.oy * a.OW + cur.ox;
#pragma unroll
            for (int i = 0; i < TM; i++) {
                const int co = co0 + i * 16 + 4 * kq;
                if (co >= a.cout) continue;
                const float s4[4] = {acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
                u8_finish4(a, s4, co, cur.n, OHW, opix, (cur.oy >> 1) * (a.OW >> 1) + (cur.ox >> 1), (l15 & 3) == 0, rq_inv, tail);
            }
        }
        cur = nxt;
    }
}

// 3x3, stride 1 | 2, dilation 1, group 1, C = 16 | 32 (4 | 8 super-steps, both even), at least one 16-pixel tile of main pixels.
// TAMD_U8_C3=0: never; =1: wherever it applies (tests); default: plan-time race against the other members
bool conv_u8_c3_applicable(const U8ConvArgs& a, int KH, int KW, int DH, int DW)
{
    const char* env = tamd_pin("u8_c3");
    if (env && atoi(env) == 0) return false;
    return KH == 3 && KW == 3 && DH == 1 && DW == 1 && (a.C == 16 || a.C == 32) && a.K == 9 * a.C && ((a.OH * a.OW) & ~7) >= 16
           && (size_t)a.C * a.H * a.W < (1u << 31) && (size_t)a.K * 4 <= 150 * 1024;
}

const char* conv_u8_c3_kernel_name(const U8ConvArgs& a) { return a.C == 16 ? "conv_u8_c3<c16>" : "conv_u8_c3<c32>"; }

hipError_t launch_conv_u8_c3(const U8ConvArgs& a, hipStream_t s)
{
    const int OHW = a.OH * a.OW, N8 = OHW & ~7, PTI = (N8 + 15) / 16;
    const int tm = 2, CT = (a.cout + 16 * tm - 1) / (16 * tm);
    const long items = (long)a.N * PTI * CT;             // (pixel tile, cout group) pairs; ~8 tiles per wave where the layer has them
    long blocks = std::min<long>(std::max<long>((items + 31) / 32, std::min<long>(1024, (items + 3) / 4)), 4096);
    while ((blocks * 4) % CT) blocks++;                  // waves a multiple of CT: a wave keeps ONE cout group for all its tiles
    const int main_blocks = (int)blocks;
    const int tail_blocks = (OHW - N8) * a.N * ((a.cout + 63) / 64);
    const size_t lds = tail_blocks ? (size_t)a.K * 4 : 0;
    const dim3 grid(main_blocks + tail_blocks, 1, 1);
    if (a.C == 16) hipLaunchKernelGGL((conv_u8_c3_k<2, 4>), grid, dim3(256), lds, s, a, main_blocks);
    else hipLaunchKernelGGL((conv_u8_c3_k<2, 8>), grid, dim3(256), lds, s, a, main_blocks);
    return hipGetLastError();
}

// =================================================================================================================
// First layers (3x3, <= 4 input channels, dilation 1: YOLOv3-tiny conv0, MobileNet / SSD conv0): K = 9*C is one
// MFMA stage at most, so the GEMM kernel above is all set-up and epilogue there.  Here a thread owns one pixel, keeps
// its K dequantised taps in registers and walks the output channels: weights are LDS broadcasts, each
// output is its own chain in the reference's order -- the single chain for pixels j < (OH*OW)&~7, the four k%4 chains
// + combine + K%4 remainder for the tail pixels (same rules as conv_u8_body) -- followed by the same epilogue.
// Stores run along pixels (NCHW rows).  Bound: VALU (27..36 fma + the exact requantisation per output).
// =================================================================================================================
template <int C>
__global__ __launch_bounds__(256) void conv_u8_rgb3x3_k(const U8ConvArgs a)
{
    constexpr int K = 9 * C, K4 = K & ~3, LD = (K + 3) & ~3;
    // the dequantised weights live in LDS: every lane reads the same address (broadcast), and unlike global loads they
    // are not ordered against the byte stores of the channel loop (which the compiler must assume may alias them)
    extern __shared__ float wl[];
    for (int i = threadIdx.x; i < a.cout * LD; i += 256) wl[i] = a.wf[i];
    __shared__ uint8_t tail[512];                   // fused ReLU / pool nodes as byte tables (u8_epilogue.h)
    u8_tail_tables(tail, threadIdx.x, 256, a.relu, a.out_scale, a.out_zp, a.pool);
    __syncthreads();
    const float rq_inv = __fdiv_rn(1.0f, a.out_scale);
    const int OHW = a.OH * a.OW;
    const int pj = blockIdx.x * 256 + threadIdx.x, n = blockIdx.y;
    if (pj >= OHW) return;           // OHW % 4 == 0 with a fused pool: a quad of lanes leaves together
    int oy, ox;
    conv_pixel(a, pj, &oy, &ox);
    const int iy0 = oy * a.SH - a.PH, ix0 = ox * a.SW - a.PW;
    const uint8_t* xin = a.x + (size_t)n * C * a.H * a.W;
    unsigned u[K];
    unsigned long long okm = 0;
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int c = k / 9, ky = (k % 9) / 3, kx = k % 3;
        const int iy = iy0 + ky, ix = ix0 + kx;
        const bool ok = ((unsigned)iy < (unsigned)a.H) & ((unsigned)ix < (unsigned)a.W);
        u[k] = xin[ok ? (c * a.H + iy) * a.W + ix : 0];
        okm |= ok ? 1ull << k : 0ull;
    }
    float xf[K];
#pragma unroll
    for (int k = 0; k < K; k++) xf[k] = (okm >> k & 1ull) ? dequant((uint8_t)u[k], a.in_zp, a.in_scale) : 0.f;   // im2col zero
    const bool tail_px = pj >= (OHW & ~7);
    uint8_t* yo = a.y + (size_t)n * a.out_img + (size_t)a.out_c0 * OHW + oy * a.OW + ox;
    uint8_t* yp = a.pool.on ? a.pool.y + (size_t)n * a.pool.out_img + (size_t)a.pool.out_c0 * (OHW >> 2) + (pj >> 2) : nullptr;
    // four output channels per iteration, the epilogue in phases: requantise (one wave-level hand-over test for the four), the ReLU
    // table, the window maxima, the pool table, the stores -- value by value every output waited for two dependent LDS look-ups
    for (int co0 = 0; co0 < a.cout; co0 += 4) {
        float sv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int co = min(co0 + u, a.cout - 1);          // (a ragged last group repeats the last row: never stored)
            float w[LD];
#pragma unroll
            for (int k = 0; k < LD; k += 4) {
                const float4 f = *reinterpret_cast<const float4*>(wl + co * LD + k);
                w[k] = f.x; w[k + 1] = f.y; w[k + 2] = f.z; w[k + 3] = f.w;
            }
            float s = 0.f;
            if (!tail_px) {
#pragma unroll
                for (int k = 0; k < K; k++) s = __builtin_fmaf(xf[k], w[k], s);
            } else {
                float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
                for (int k = 0; k < K4; k += 4) {
                    s0 = __builtin_fmaf(w[k], xf[k], s0);
                    s1 = __builtin_fmaf(w[k + 1], xf[k + 1], s1);
                    s2 = __builtin_fmaf(w[k + 2], xf[k + 2], s2);
                    s3 = __builtin_fmaf(w[k + 3], xf[k + 3], s3);
                }
                if (co < a.m_blocked) s = (0.f + (s0 + s1)) + (s2 + s3);
                else s = ((s0 + s1) + s2) + s3;
#pragma unroll
                for (int k = K4; k < K; k++) s = __builtin_fmaf(w[k], xf[k], s);
            }
            if (a.bias) s = s + (float)a.bias[co] * a.bias_scale;
            if (a.act == 0) s = s < 0.f ? 0.f : s;
            if (a.act > 0) { s = s < 0.f ? 0.f : s; s = s > 6.f ? 6.f : s; }
            sv[u] = s;
        }
        int q4[4];
        quant_round_sat_u8_w4(sv, a.out_scale, rq_inv, a.out_zp, q4);
        if (a.relu.on) {
#pragma unroll
            for (int u = 0; u < 4; u++) q4[u] = tail[q4[u]];
        }
        if (!a.pool.on || a.pool.write_full) {
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (co0 + u < a.cout) yo[(size_t)(co0 + u) * OHW] = (uint8_t)q4[u];
        }
        if (a.pool.on) {
            int pb[4];
#pragma unroll
            for (int u = 0; u < 4; u++) pb[u] = tail[256 + quad_max(q4[u])];
            if ((threadIdx.x & 3) == 0) {
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (co0 + u < a.cout) yp[(size_t)(co0 + u) * (OHW >> 2)] = (uint8_t)pb[u];
            }
        }
    }
}

// The same layers with the MAIN pixels (j < (OH*OW)&~7) on the matrix cores (round 4).  The per-pixel kernel above is bound by its
// VALU work -- 9 C fused multiply-adds per output, 16 .. 64 outputs per pixel: YOLOv3-tiny conv0 at batch 8 was 62 us of arithmetic --
// while the single chain of a main pixel is exactly what v_mfma_f32_16x16x4f32 accumulates (conv_u8_body's header): K = 9 C is
// (9 C + 3) / 4 MFMA steps for 16 channels x 16 pixels at once.  A wave keeps the dequantised weight rows of all its channel tiles
// in registers (A: lane (row l15, kq) holds k = 4 s + kq), walks 16-pixel tiles of one image (window-major under a fused pool, as
// everywhere), gathers the B operand straight from the NCHW input -- lane (pixel l15, kq) needs the (9 C + 3) / 4 taps k = 4 s + kq of
// its pixel: byte loads at offsets tabulated once per lane, out-of-image taps as 0.0f -- and issues the steps in ascending k.  The
// padded k (27 -> 28) carries a zero weight and a zero tap: fma(0, 0, s) == s.  Tail pixels: one extra block per image runs the
// per-pixel code (their four k%4 chains are lane-level arithmetic anyway).
template <int C>
__device__ __forceinline__ void conv_u8_rgb3x3_pixel(const U8ConvArgs& a, const float* wl, const uint8_t* tail, float rq_inv, int n, int pj)
{
    constexpr int K = 9 * C, K4 = K & ~3, LD = (K + 3) & ~3;
    const int OHW = a.OH * a.OW;
    int oy, ox;
    conv_pixel(a, pj, &oy, &ox);
    const int iy0 = oy * a.SH - a.PH, ix0 = ox * a.SW - a.PW;
    const uint8_t* xin = a.x + (size_t)n * C * a.H * a.W;
    float xf[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int c = k / 9, ky = (k % 9) / 3, kx = k % 3;
        const int iy = iy0 + ky, ix = ix0 + kx;
        const bool ok = ((unsigned)iy < (unsigned)a.H) & ((unsigned)ix < (unsigned)a.W);
        const unsigned u = xin[ok ? (c * a.H + iy) * a.W + ix : 0];
        xf[k] = ok ? dequant((uint8_t)u, a.in_zp, a.in_scale) : 0.f;
    }
    uint8_t* yo = a.y + (size_t)n * a.out_img + (size_t)a.out_c0 * OHW + oy * a.OW + ox;
    for (int co = 0; co < a.cout; co++) {
        const float* w = wl + co * LD;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s;
#pragma unroll
        for (int k = 0; k < K4; k += 4) {
            s0 = __builtin_fmaf(w[k], xf[k], s0);
            s1 = __builtin_fmaf(w[k + 1], xf[k + 1], s1);
            s2 = __builtin_fmaf(w[k + 2], xf[k + 2], s2);
            s3 = __builtin_fmaf(w[k + 3], xf[k + 3], s3);
        }
        if (co < a.m_blocked) s = (0.f + (s0 + s1)) + (s2 + s3);
        else s = ((s0 + s1) + s2) + s3;
#pragma unroll
        for (int k = K4; k < K; k++) s = __builtin_fmaf(w[k], xf[k], s);
        if (a.bias) s = s + (float)a.bias[co] * a.bias_scale;
        if (a.act == 0) s = s < 0.f ? 0.f : s;
        if (a.act > 0) { s = s < 0.f ? 0.f : s; s = s > 6.f ? 6.f : s; }
        uint8_t q = quant_round_sat_u8_w(s, a.out_scale, rq_inv, a.out_zp);
        if (a.relu.on) q = tail[q];
        yo[(size_t)co * OHW] = q;                     // (a layer with tail pixels has no fused pool: OH*OW % 8 != 0)
    }
}

#ifdef TAMD_EXPERIMENTS      // the first layer on the matrix cores: lost to the per-pixel kernel (profiles/r04_experiment_u8_first_layer_mfma.txt)
template <int C, int TM>
__global__ __launch_bounds__(256) void conv_u8_rgb3x3_mfma_k(const U8ConvArgs a, int main_x)
{
    constexpr int K = 9 * C, KS = (K + 3) / 4, LD = (K + 3) & ~3;
    extern __shared__ float wl[];                   // tail blocks only: the weight rows
    __shared__ uint8_t tail[512];                   // fused ReLU / pool nodes as byte tables (u8_epilogue.h)
    const int tid = threadIdx.x, n = blockIdx.y;
    u8_tail_tables(tail, tid, 256, a.relu, a.out_scale, a.out_zp, a.pool);
    const float rq_inv = __fdiv_rn(1.0f, a.out_scale);
    const int OHW = a.OH * a.OW, N8 = OHW & ~7;
    if ((int)blockIdx.x >= main_x) {                // the tail pixels of image n
        for (int i = tid; i < a.cout * LD; i += 256) wl[i] = a.wf[i];
        __syncthreads();
        if (N8 + tid < OHW) conv_u8_rgb3x3_pixel<C>(a, wl, tail, rq_inv, n, N8 + tid);
        return;
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kq = lane >> 4;
    // A: this lane's weights of every step, rows past cout repeat the last one (never stored)
    float af[TM][KS];
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const int row = min(i * 16 + l15, a.cout - 1);
#pragma unroll
        for (int s = 0; s < KS; s++) af[i][s] = a.wf[(size_t)row * LD + 4 * s + kq];          // (k >= K: the zero padding of the rows)
    }
    float bf4[TM][4];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int e = 0; e < 4; e++) bf4[i][e] = a.bias ? (float)a.bias[min(i * 16 + 4 * kq + e, a.cout - 1)] * a.bias_scale : 0.f;
    // B: the taps k = 4 s + kq of a pixel: plane offset + in-window offset, and the (dy, dx) the border test needs
    int toff[KS], tdy[KS], tdx[KS];
#pragma unroll
    for (int s = 0; s < KS; s++) {
        const int k = 4 * s + kq, c = k / 9, r = k - 9 * c, dy = r / 3, dx = r - 3 * dy;
        toff[s] = c * a.H * a.W + dy * a.W + dx;
        tdy[s] = k < K ? dy : (1 << 20);            // the padded k: a row no image has
        tdx[s] = dx;
    }
    const uint8_t* xin = a.x + (size_t)n * C * a.H * a.W;
    // (N8 % 16 == 8: the last tile has eight live pixels)
    // the taps of a tile are requested one tile AHEAD (two register sets): without that every tile paid a whole memory round trip
    // between its address arithmetic and its first MFMA (the first version of this kernel lost to the per-pixel one: 83 vs 62 us)
    struct TileIn { unsigned raw[KS]; unsigned okm; int oy, ox, pj; };
    auto fetch = [&](int t, TileIn& ti) {
        ti.pj = t * 16 + l15;
        const bool live = ti.pj < N8;
        conv_pixel(a, live ? ti.pj : N8 - 1, &ti.oy, &ti.ox);
        const int iy0 = ti.oy * a.SH - a.PH, ix0 = ti.ox * a.SW - a.PW, base = iy0 * a.W + ix0;
        ti.okm = 0;
#pragma unroll
        for (int s = 0; s < KS; s++) {
            const bool ok = ((unsigned)(iy0 + tdy[s]) < (unsigned)a.H) & ((unsigned)(ix0 + tdx[s]) < (unsigned)a.W);
            ti.raw[s] = xin[ok ? base + toff[s] : 0];
            ti.okm |= ok ? 1u << s : 0u;
        }
    };
    auto compute = [&](const TileIn& ti) {
        const bool live = ti.pj < N8;
        v4f acc[TM];
#pragma unroll
        for (int i = 0; i < TM; i++) acc[i] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; s++) {
            const float bv = (ti.okm >> s & 1u) ? dequant((uint8_t)ti.raw[s], a.in_zp, a.in_scale) : 0.f;
#pragma unroll
            for (int i = 0; i < TM; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][s], bv, acc[i], 0, 0, 0);
        }
        // D[row = 4 kq + e][col = l15]: this lane holds channels i * 16 + 4 kq + e of its pixel.  The epilogue runs in PHASES over all
        // 4 TM values -- requantise, table look-ups, window maxima, table look-ups, stores -- so that the LDS round trips of the
        // byte tables overlap (value by value they were two dependent LDS latencies per output)
        const int opix = ti.oy * a.OW + ti.ox;
        int qv[TM][4];
#pragma unroll
        for (int i = 0; i < TM; i++) {
            float sv[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                sv[e] = acc[i][e];
                if (a.bias) sv[e] = sv[e] + bf4[i][e];
                if (a.act == 0) sv[e] = sv[e] < 0.f ? 0.f : sv[e];
                if (a.act > 0) { sv[e] = sv[e] < 0.f ? 0.f : sv[e]; sv[e] = sv[e] > 6.f ? 6.f : sv[e]; }
            }
            quant_round_sat_u8_w4(sv, a.out_scale, rq_inv, a.out_zp, qv[i]);
        }
        if (a.relu.on) {
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int e = 0; e < 4; e++) qv[i][e] = tail[qv[i][e]];
        }
        if (!a.pool.on || a.pool.write_full) {
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int co = i * 16 + 4 * kq + e;
                    if (live && co < a.cout) a.y[(size_t)n * a.out_img + (size_t)(a.out_c0 + co) * OHW + opix] = (uint8_t)qv[i][e];
                }
        }
        if (a.pool.on) {                             // N8 == OHW under a fused pool (OHW % 8 == 0): every lane of a quad is live
            int pb[TM][4];
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int e = 0; e < 4; e++) pb[i][e] = tail[256 + quad_max(qv[i][e])];
            if ((l15 & 3) == 0 && live) {
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int co = i * 16 + 4 * kq + e;
                        if (co < a.cout) a.pool.y[(size_t)n * a.pool.out_img + (size_t)(a.pool.out_c0 + co) * (OHW >> 2) + (ti.pj >> 2)] = (uint8_t)pb[i][e];
                    }
            }
        }
    };
    const int stride = main_x * 4;
    int t = blockIdx.x * 4 + wave;
    if (t * 16 >= N8) return;
    TileIn ta, tb;
    fetch(t, ta);
    for (; t * 16 < N8; t += 2 * stride) {
        const bool more = (t + stride) * 16 < N8;
        if (more) fetch(t + stride, tb);
        compute(ta);
        if (!more) break;
        if ((t + 2 * stride) * 16 < N8) fetch(t + 2 * stride, ta);
        compute(tb);
    }
}

#endif

bool conv_u8_rgb3x3_applicable(int cin, int kh, int kw, int dh, int dw, int group)
{
    return group == 1 && kh == 3 && kw == 3 && dh == 1 && dw == 1 && (cin == 1 || cin == 3 || cin == 4);
}

// the MFMA form of a first layer: cout <= 64 (four channel tiles of weights per lane), at least one 16-pixel tile of main pixels.
// OFF by default -- TAMD_U8_RGB_MFMA=1 enables it (tests, A/B runs; read per launch): measured inside one box it does not beat the
// per-pixel kernel (YOLOv3-tiny b8 conv0 64.7 vs 62.2 us isolated, the step +12 us; mssd b16 conv0 41 vs 35 us,
// profiles/r04_experiment_u8_first_layer_mfma.txt): both forms issue the same number of byte gathers and byte stores per pixel, and
// with 16 .. 32 outputs per pixel the requantisation, not the 27 multiply-adds, is most of the arithmetic.
#ifdef TAMD_EXPERIMENTS
static bool u8_rgb_mfma(const U8ConvArgs& a)
{
    const char* e = exp_env("TAMD_U8_RGB_MFMA");
    return e && atoi(e) == 1 && a.cout <= 64 && ((a.OH * a.OW) & ~7) >= 16 && (a.C == 3 || a.C == 4);
}
#else
static bool u8_rgb_mfma(const U8ConvArgs&) { return false; }
#endif
const char* conv_u8_rgb3x3_kernel_name(const U8ConvArgs& a) { return u8_rgb_mfma(a) ? "conv_u8_rgb3x3_mfma" : "conv_u8_rgb3x3"; }

#ifdef TAMD_EXPERIMENTS
template <int C>
static hipError_t launch_rgb_mfma(const U8ConvArgs& a, hipStream_t s)
{
    const int OHW = a.OH * a.OW, N8 = OHW & ~7;
    const int tiles = (N8 + 15) / 16, tail_x = OHW != N8 ? 1 : 0;
    // eight tiles per wave where the layer has them (the kernel requests a tile's taps while it multiplies the previous one),
    // fewer when that would leave CUs without a block
    int per_wave = 8;
    while (per_wave > 1 && (long)((tiles + 4 * per_wave - 1) / (4 * per_wave)) * a.N < 1024) per_wave >>= 1;
    const int main_x = std::min((tiles + 4 * per_wave - 1) / (4 * per_wave), 2048);
    const dim3 grid(main_x + tail_x, a.N);
    const size_t lds = tail_x ? (size_t)a.cout * ((9 * C + 3) & ~3) * 4 : 0;
    switch ((a.cout + 15) / 16) {
    case 1: hipLaunchKernelGGL((conv_u8_rgb3x3_mfma_k<C, 1>), grid, dim3(256), lds, s, a, main_x); break;
    case 2: hipLaunchKernelGGL((conv_u8_rgb3x3_mfma_k<C, 2>), grid, dim3(256), lds, s, a, main_x); break;
    case 3: hipLaunchKernelGGL((conv_u8_rgb3x3_mfma_k<C, 3>), grid, dim3(256), lds, s, a, main_x); break;
    default: hipLaunchKernelGGL((conv_u8_rgb3x3_mfma_k<C, 4>), grid, dim3(256), lds, s, a, main_x); break;
    }
    return hipGetLastError();
}
#endif

hipError_t launch_conv_u8_rgb3x3(const U8ConvArgs& a, hipStream_t s)
{
#ifdef TAMD_EXPERIMENTS
    if (u8_rgb_mfma(a)) return a.C == 3 ? launch_rgb_mfma<3>(a, s) : launch_rgb_mfma<4>(a, s);
#endif
    dim3 grid((a.OH * a.OW + 255) / 256, a.N);
    switch (a.C) {
    case 1: hipLaunchKernelGGL(conv_u8_rgb3x3_k<1>, grid, dim3(256), (size_t)a.cout * 12 * 4, s, a); break;
    case 3: hipLaunchKernelGGL(conv_u8_rgb3x3_k<3>, grid, dim3(256), (size_t)a.cout * 28 * 4, s, a); break;
    case 4: hipLaunchKernelGGL(conv_u8_rgb3x3_k<4>, grid, dim3(256), (size_t)a.cout * 36 * 4, s, a); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}


}  // namespace tamd
