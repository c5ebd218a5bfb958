// uint8 convolution, the PATCH kernel (conv_u8_patch: dequantised input patch in LDS, weight fragments straight into registers) and the
// lane-level chains for small layers (conv_u8_lanes).  The contract every uint8 kernel follows is stated at the top of u8_kernels.hip.
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
// The same GEMM for the MAIN pixels with both operands free of per-element staging (round 3).
//
// conv_u8_body gathers and dequantises every im2col element and every weight -- a 3x3 layer touches each input byte nine times,
// ~14 VALU instructions per touch (tap decode, bounds, byte load, (x - zp) * scale), ~150 per thread per 32 MFMAs -- and a wave
// issues one instruction per 8-10 cycles: the kernels sat at 21-30 % of the fp32 MFMA rate on their staging.  Here
//   B (input):  the block keeps the input patch of its pixel tile (3x3: every input row the tile touches, halo rows and columns
//     included as real 0.0f -- the reference's padding taps; 1x1: the tile's own pixels) for a chunk of channels in LDS, ALREADY
//     dequantised, [channel][patch pixel] floats with a compile-time plane stride: each input byte is converted once per block
//     and chunk, and the B value of MFMA step s for lane (pixel l15, k%4 = kq) is ONE ds_read_b32 at a per-lane address computed
//     once in the prologue plus an immediate -- k = 4s + kq inside a super-step of 4*SS k decomposes into (channel, ky, kx) the
//     same way in every super-step (4*SS is a multiple of KH*KW), so the SS addresses per pixel tile are loop invariants;
//   A (weights): dequantised ONCE at plan time on the host -- ((float)w - zp) * scale in fp32 is the same IEEE value wherever it
//     is computed -- and stored in the MFMA A-fragment order, [16-row tile][super-step][float4 group][lane]: a wave fetches its
//     fragments three to seven super-steps AHEAD straight from global memory (shared by every pixel tile) into a register ring,
//     no LDS, no conversion, no barrier.
// A super-step is 36 k (4 channels x 9 taps, 9 MFMA steps) for 3x3 and 16 k (16 channels, 4 MFMA steps) for 1x1; a patch chunk
// is 4 super-steps; one barrier per chunk; the B reads run two MFMA steps ahead; the refresh of the other patch buffer (convert,
// store, re-request) is spread over the chunk's MFMA steps.  Summation order: unchanged -- accumulator tile (i, j) receives its
// k in ascending steps of 4, which v_mfma_f32_16x16x4f32 adds as four fused multiply-adds in ascending k (conv_u8_body's
// header) -- so the bytes are the reference's.  Tail pixels (OH*OW % 8): conv_u8_patch_tail, extra blocks of the same launch.
// Blocks are numbered so that the eight XCDs split the cout tiles between them (each L2 holds its own slice of the weights).
// =================================================================================================================
// (the tail pixels, conv_u8_patch_tail: u8_patch_tail.h -- shared with u8_conv_small.hip)

// The MAIN pixels (j < (OH*OW)&~7) of SMALL layers the same way: a lane owns one output and walks its single chain k = 0 .. K-1 with
// fmaf -- what the MFMA main tiles compute (conv_u8_body's header: ascending steps of 4, four fused multiply-adds in ascending k
// inside a step) -- reading the weights of its row from the same fragment stream (the four k%4 lanes' float4 groups of a
// super-step, in k order) and the dequantised im2col column of its pixel from LDS.  A block = (four main pixels of the BATCH --
// the flat index v = image * N8 + j, so a 3x3 map's eight main pixels do not leave half-empty tiles behind --, 64 channels); lane
// (row l15, r) of wave w: channel 64 * slice + 16 * w + l15 of pixel 4 * group + r.  Why: the SSD pyramid's tail layers (5x5 ..
// 1x1 maps, 16 .. 400 pixels per batch) gave the GEMM kernels 16-pixel MFMA tiles with 1 .. 8 live columns and a K loop whose
// every 32-k stage re-gathers and re-dequantises its operands for them: 8 .. 26 us per launch for a few MMAC
// (profiles/r03_layers_mssd_uint8_b16.txt); a lane-level chain is K steps of ~1.6 instructions.
template <int KHW>
__device__ __forceinline__ void conv_u8_patch_lane_main(const U8ConvArgs& a, float* xs, int mb, const uint8_t* tail)
{
    const float rq_inv = __fdiv_rn(1.0f, a.out_scale);
    constexpr int NTAPS = KHW * KHW, SS = KHW == 3 ? 9 : 4, G4 = SS / 4, REM = SS - 4 * G4, FRAG = SS * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, r = lane >> 4;
    const int OHW = a.OH * a.OW, N8 = OHW & ~7, total = a.N * N8, slices = (a.cout + 63) / 64;
    const int slice = mb % slices, grp = mb / slices;
    const int chw = a.H * a.W;
    const int KP = a.K + 4;                                // column pitch: the four columns' float4 reads fall into different banks
    // ---- the four im2col columns, dequantised, natural k order: xs[p][KP].  Thread t stages pixel t & 3, k = t / 4 + 64 i; six byte
    // loads are in flight before the first is converted (a loop of load -> convert -> store pairs is one memory round trip each)
    {
        const int p = tid & 3, v = grp * 4 + p;
        const bool pv = v < total;
        const int n = pv ? v / N8 : 0, j = pv ? v - n * N8 : 0, oy = j / a.OW, ox = j - oy * a.OW;      // (no fused pool on this path: row-major pixels)
        const uint8_t* xin = a.x + (size_t)n * a.C * chw;
        const int iy0 = oy * a.SH - a.PH, ix0 = ox * a.SW - a.PW;
        float* xcol = xs + p * KP;
        for (int k0 = tid >> 2; k0 < a.K; k0 += 64 * 6) {
            unsigned raw[6];
            bool ok[6];
#pragma unroll
            for (int u = 0; u < 6; u++) {
                const int k = k0 + 64 * u;
                const int c = k / NTAPS, tap = k - c * NTAPS, ky = tap / KHW, kx = tap - ky * KHW;
                const int iy = iy0 + ky * a.pk_dh, ix = ix0 + kx * a.pk_dw;
                ok[u] = pv && k < a.K && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                raw[u] = xin[ok[u] ? (size_t)c * chw + iy * a.W + ix : 0];
            }
#pragma unroll
            for (int u = 0; u < 6; u++) {
                const int k = k0 + 64 * u;
                if (k < a.K) xcol[k] = ok[u] ? dequant((uint8_t)raw[u], a.in_zp, a.in_scale) : 0.f;
            }
        }
    }
    __syncthreads();
    const int tile16 = slice * 4 + wave;
    if (tile16 * 16 >= a.cout) return;
    const int nss = a.K / (4 * SS);
    const float* wb = reinterpret_cast<const float*>(a.wpk) + (size_t)tile16 * nss * FRAG;
    const float* xr = xs + r * KP;
    // fragment ring: RING super-steps of weights in registers, fetched RING - 1 ahead (a lone wave per SIMD hides nothing by
    // occupancy: with one super-step of cover -- 36 FMAs -- every step waited ~500 cycles for its weights)
    constexpr int RING = KHW == 3 ? 4 : 8;
    float4 w4[RING][G4][4];
    float wr[RING][4][REM > 0 ? REM : 1];
    auto wload = [&](auto D, int ss) {
        constexpr int d = decltype(D)::value;
        const float* wo = wb + (size_t)(ss < nss ? ss : nss - 1) * FRAG;
#pragma unroll
        for (int v = 0; v < G4; v++)
#pragma unroll
            for (int kq = 0; kq < 4; kq++) w4[d][v][kq] = *reinterpret_cast<const float4*>(wo + v * 256 + (kq * 16 + l15) * 4);
#pragma unroll
        for (int kq = 0; kq < 4; kq++)
#pragma unroll
            for (int e = 0; e < REM; e++) wr[d][kq][e] = wo[G4 * 256 + (kq * 16 + l15) * REM + e];
    };
    float acc = 0.f;
    auto sstep = [&](auto D, int ss) {
        constexpr int d = decltype(D)::value;
        wload(std::integral_constant<int, (d + RING - 1) % RING>{}, ss + RING - 1);
        const float* xp = xr + ss * 4 * SS;
#pragma unroll
        for (int v = 0; v < G4; v++) {
            // MFMA steps s = 4v .. 4v+3 of the super-step, k = 4 s + kq inside it: lane kq's float4 holds its value for each of them
            const float4 x0 = *reinterpret_cast<const float4*>(xp + 16 * v), x1 = *reinterpret_cast<const float4*>(xp + 16 * v + 4);
            const float4 x2 = *reinterpret_cast<const float4*>(xp + 16 * v + 8), x3 = *reinterpret_cast<const float4*>(xp + 16 * v + 12);
            acc = __builtin_fmaf(w4[d][v][0].x, x0.x, acc); acc = __builtin_fmaf(w4[d][v][1].x, x0.y, acc);
            acc = __builtin_fmaf(w4[d][v][2].x, x0.z, acc); acc = __builtin_fmaf(w4[d][v][3].x, x0.w, acc);
            acc = __builtin_fmaf(w4[d][v][0].y, x1.x, acc); acc = __builtin_fmaf(w4[d][v][1].y, x1.y, acc);
            acc = __builtin_fmaf(w4[d][v][2].y, x1.z, acc); acc = __builtin_fmaf(w4[d][v][3].y, x1.w, acc);
            acc = __builtin_fmaf(w4[d][v][0].z, x2.x, acc); acc = __builtin_fmaf(w4[d][v][1].z, x2.y, acc);
            acc = __builtin_fmaf(w4[d][v][2].z, x2.z, acc); acc = __builtin_fmaf(w4[d][v][3].z, x2.w, acc);
            acc = __builtin_fmaf(w4[d][v][0].w, x3.x, acc); acc = __builtin_fmaf(w4[d][v][1].w, x3.y, acc);
            acc = __builtin_fmaf(w4[d][v][2].w, x3.z, acc); acc = __builtin_fmaf(w4[d][v][3].w, x3.w, acc);
        }
#pragma unroll
        for (int e = 0; e < REM; e++) {
            const float4 x = *reinterpret_cast<const float4*>(xp + 16 * G4 + 4 * e);
            acc = __builtin_fmaf(wr[d][0][e], x.x, acc); acc = __builtin_fmaf(wr[d][1][e], x.y, acc);
            acc = __builtin_fmaf(wr[d][2][e], x.z, acc); acc = __builtin_fmaf(wr[d][3][e], x.w, acc);
        }
    };
    u8_static_for<0, RING - 1>([&](auto D) { wload(D, decltype(D)::value); });
    for (int ss = 0; ss < nss; ss += RING)
        u8_static_for<0, RING>([&](auto D) {
            constexpr int d = decltype(D)::value;
            if (ss + d < nss) sstep(D, ss + d);
        });
    const int co = tile16 * 16 + l15, v = grp * 4 + r;
    if (v >= total || co >= a.cout) return;
    const int n = v / N8, opix = v - n * N8;
    float s = acc;
    if (a.bias) s = s + (float)a.bias[co] * a.bias_scale;
    if (a.act == 0) s = s < 0.f ? 0.f : s;
    if (a.act > 0) { s = s < 0.f ? 0.f : s; s = s > 6.f ? 6.f : s; }
    uint8_t q = quant_round_sat_u8_w(s, a.out_scale, rq_inv, a.out_zp);
    if (a.relu.on) q = tail[q];
    a.y[(size_t)n * a.out_img + (size_t)(a.out_c0 + co) * OHW + opix] = q;
}

// a whole (small) layer as lane-level chains: main-pixel blocks first, then the tail pixels' (launch_conv_u8_patch, configuration 4)
template <int KHW>
__global__ __launch_bounds__(256) void conv_u8_lanes_k(const U8ConvArgs a, int main_blocks)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ uint8_t tail[512];                   // fused ReLU node as a byte table (both callees put a barrier behind their staging)
    u8_tail_tables(tail, threadIdx.x, 256, a.relu, a.out_scale, a.out_zp, a.pool);
    if ((int)blockIdx.x < main_blocks) conv_u8_patch_lane_main<KHW>(a, smem, blockIdx.x, tail);
    else conv_u8_patch_tail<KHW>(a, smem, blockIdx.x - main_blocks, tail);
}

#ifndef TAMD_U8P_ABLATE
#define TAMD_U8P_ABLATE 0          // tools/exp/u8_patch_anatomy.hip: 1 no MFMA, 2 no B reads, 4 no fragment fetch, 8 no patch refresh, 16 no stores
#endif
template <int WM, int WN, int TM, int TN, int KHW, int NPSTR>
__global__ __launch_bounds__(256) void conv_u8_patch_k(const U8ConvArgs a)
{
    constexpr int ABL = TAMD_U8P_ABLATE;
    static_assert(WM * WN == 4, "four waves");
    constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
    constexpr int NTAPS = KHW * KHW;
    constexpr int SS = KHW == 3 ? 9 : 4;                 // MFMA steps per super-step: 36 k = 4 channels x 9 taps | 16 k = 16 channels
    constexpr int CSS = 4 * SS / NTAPS;                  // channels per super-step
    constexpr int CPC = 4;                               // super-steps per patch chunk == fragment register slots
    constexpr int CC = CPC * CSS;                        // channels per patch chunk (16 | 64)
    constexpr int NP = KHW == 3 ? NPSTR : BN;            // floats per channel plane of the patch
    constexpr int PG = NP >= 256 ? 1 : 256 / NP;         // thread groups along the chunk's channels
    constexpr int NPS = NP >= 256 ? NP / 256 : 1;        // patch pixels per thread
    constexpr int CPT = CC / PG;                         // channels per thread
    constexpr int G4 = SS / 4, REM = SS - 4 * G4;        // float4 groups / single floats of a lane's fragment per super-step
    extern __shared__ __attribute__((aligned(16))) float smem[];          // patch [2][CC][NP]
    __shared__ uint8_t tail[512];                   // fused ReLU / pool nodes as byte tables (u8_epilogue.h); the chunk loop's barriers
    u8_tail_tables(tail, threadIdx.x, 256, a.relu, a.out_scale, a.out_zp, a.pool);      // (>= 1) stand between this and the epilogue
    const float rq_inv = __fdiv_rn(1.0f, a.out_scale);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM, l15 = lane & 15, kq = lane >> 4;
    const int OHW = a.OH * a.OW, N8 = OHW & ~7;
    const int tiles = (N8 + BN - 1) / BN, PT = tiles * a.N, CT = (a.cout + BM - 1) / BM;
    if ((int)blockIdx.x >= PT * CT) {                    // the blocks behind the main grid: tail pixels
        conv_u8_patch_tail<KHW>(a, smem, blockIdx.x - PT * CT, tail);
        return;
    }
    int pt, ct;
    if ((CT & 7) == 0) { const int lin = blockIdx.x, idx = lin >> 3; ct = (lin & 7) + 8 * (idx / PT); pt = idx % PT; }
    else { pt = blockIdx.x % PT; ct = blockIdx.x / PT; }
    const int n = pt / tiles, tile = pt - n * tiles, co0 = ct * BM;
    const int jbase = tile * BN, jlimit = N8;
    const int Wp = a.pk_wp;                              // patch row pitch: the map's width + halo, or (2-D tiles) the tile's
    const int DH = a.pk_dh, DW = a.pk_dw;

    // ---- patch geometry of this pixel tile ---------------------------------------------------------------------------------
    int oy_a, ox_a, oy_b, ox_b;
    conv_pixel(a, jbase, &oy_a, &ox_a);
    conv_pixel(a, (jbase + BN < jlimit ? jbase + BN : jlimit) - 1, &oy_b, &ox_b);
    const uint8_t* xin = a.x + (size_t)n * a.C * a.H * a.W;
    const int chw = a.H * a.W;
    // first input column the patch holds / first output column of the tile: the whole row (1-D runs of pixels) or the tile's own (2-D)
    const int oxt = a.pk_tw > 0 ? ox_a : 0, px0 = oxt * a.SW - a.PW;
    const int pg = PG > 1 ? tid / NP : 0, ppix = PG > 1 ? tid % NP : tid;
    int soff[NPS];                                       // this thread's patch pixels: offset inside a channel plane, -1: a zero
#pragma unroll
    for (int q = 0; q < NPS; q++) {
        const int pp = ppix + 256 * q;
        if (KHW == 3) {
            const int NPX = ((oy_b - oy_a) * a.SH + (KHW - 1) * DH + 1) * Wp;
            const int prow = pp / Wp, pcol = pp - prow * Wp;
            const int iy = oy_a * a.SH - a.PH + prow, ix = pcol + px0;
            soff[q] = (pp < NPX && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) ? iy * a.W + ix : -1;
        } else {
            const int pj = jbase + pp;
            int oy, ox;
            conv_pixel(a, pj < jlimit ? pj : jlimit - 1, &oy, &ox);
            const int iy = oy * a.SH - a.PH, ix = ox * a.SW - a.PW;
            soff[q] = (pj < jlimit && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) ? iy * a.W + ix : -1;
        }
    }
    float pmask[NPS];                                    // clamp bound of the patch pixel: +inf inside the image, 0 outside (branch-free zero)
#pragma unroll
    for (int q = 0; q < NPS; q++) pmask[q] = soff[q] >= 0 ? __builtin_inff() : 0.f;
    const int nss = a.K / (4 * SS), nchunk = (a.C + CC - 1) / CC;
    unsigned pregs[NPS][CPT];                            // raw bytes of the chunk in flight (one register each: no wait until they are used)
    auto pload = [&](int c) {
        const int c0 = (c < nchunk ? c : nchunk - 1) * CC + pg * CPT;        // past the end: a harmless repeat
#pragma unroll
        for (int q = 0; q < NPS; q++)
#pragma unroll
            for (int cl = 0; cl < CPT; cl++) {
                const int ch = c0 + cl < a.C ? c0 + cl : a.C - 1;
                pregs[q][cl] = xin[(size_t)ch * chw + (soff[q] >= 0 ? soff[q] : 0)];
            }
    };
    auto pstore = [&](int buf) {
#pragma unroll
        for (int q = 0; q < NPS; q++)
#pragma unroll
            for (int cl = 0; cl < CPT; cl++)
            {
                const float v = dequant((uint8_t)pregs[q][cl], a.in_zp, a.in_scale);
                smem[(buf * CC + pg * CPT + cl) * NP + ppix + 256 * q] = __builtin_amdgcn_fmed3f(v, pmask[q], -pmask[q]);      // inside: v, outside: 0.f
            }
    };

    // the refresh of the OTHER patch buffer, spread over the chunk's MFMA steps: element e of part u is converted and stored, and
    // its register immediately re-requested for the chunk after (every byte flies for one whole chunk)
    constexpr int EPP = NPS * CPT / CPC, EPS = (EPP + SS - 1) / SS;      // elements per super-step / per MFMA step
    auto refresh = [&](auto BUF, auto U, auto S, int c) {
        constexpr int buf = decltype(BUF)::value, u = decltype(U)::value, st = decltype(S)::value;
        if (ABL & 8) return;
#pragma unroll
        for (int e = st * EPS; e < (st + 1) * EPS && e < EPP; e++) {
            const int idx = u * EPP + e, q = idx / CPT, cl = idx % CPT;
            const float v = dequant((uint8_t)pregs[q][cl], a.in_zp, a.in_scale);
            smem[((buf ^ 1) * CC + pg * CPT + cl) * NP + ppix + 256 * q] = __builtin_amdgcn_fmed3f(v, pmask[q], -pmask[q]);
            const int c0 = (c + 2 < nchunk ? c + 2 : nchunk - 1) * CC + pg * CPT;
            const int ch = c0 + cl < a.C ? c0 + cl : a.C - 1;
            pregs[q][cl] = xin[(size_t)ch * chw + (soff[q] >= 0 ? soff[q] : 0)];
        }
    };

    // ---- weights: this wave's TM fragment streams, [tile16][super-step][G4 x (64 lanes x float4)][64 lanes x REM floats] ---------
    constexpr int FRAG = SS * 64;                        // floats per (16-row tile, super-step)
    const float* wbase[TM];
#pragma unroll
    for (int i = 0; i < TM; i++) wbase[i] = reinterpret_cast<const float*>(a.wpk) + (size_t)(co0 / 16 + wm * TM + i) * nss * FRAG;
    // fragment registers, a ring of RA super-steps (the loaded tuples are used where they land): three super-steps (~1.5 us of MFMA)
    // ahead.  A ring of 2 * CPC for the narrow configurations (seven ahead, round 4) costs them 16-32 registers and is the slower
    // one in a whole pass: YOLOv3-tiny b8 777.7 -> 729.9 us with this ring, MobileNet-SSD b16 unchanged
    // (profiles/r05_ab_u8_patch_ra4_*.txt)
    constexpr int RA = CPC;
    float4 af4[RA][TM][G4];
    float afr[RA][TM][REM > 0 ? REM : 1];
    auto aload = [&](auto D, int ss) {
        constexpr int d = decltype(D)::value;
        const size_t o = (size_t)(ss < nss ? ss : nss - 1) * FRAG;
        if (ABL & 4) { if (ss >= RA - 1) return; }
#pragma unroll
        for (int i = 0; i < TM; i++) {
#pragma unroll
            for (int v = 0; v < G4; v++) af4[d][i][v] = *reinterpret_cast<const float4*>(wbase[i] + o + v * 256 + lane * 4);
#pragma unroll
            for (int v = 0; v < REM; v++) afr[d][i][v] = wbase[i][o + G4 * 256 + lane * REM + v];
        }
    };
    auto afrag = [&](auto D, int i, auto S) -> float {
        constexpr int d = decltype(D)::value, s = decltype(S)::value;
        if constexpr (s >= 4 * G4) return afr[d][i][s - 4 * G4];
        else if constexpr ((s & 3) == 0) return af4[d][i][s >> 2].x;
        else if constexpr ((s & 3) == 1) return af4[d][i][s >> 2].y;
        else if constexpr ((s & 3) == 2) return af4[d][i][s >> 2].z;
        else return af4[d][i][s >> 2].w;
    };

    // ---- per-lane B addresses (floats, relative to the first channel plane of the super-step) ---------------------------------
    int baddr[TN][SS];
#pragma unroll
    for (int j = 0; j < TN; j++) {
        const int pl = (wn * TN + j) * 16 + l15;
        int pp0 = pl;
        if (KHW == 3) {
            int pj = jbase + pl;
            pj = pj < jlimit ? pj : jlimit - 1;
            int oy, ox;
            conv_pixel(a, pj, &oy, &ox);
            pp0 = ((oy - oy_a) * a.SH) * Wp + (ox - oxt) * a.SW;
        }
#pragma unroll
        for (int s = 0; s < SS; s++) {
            const int kl = 4 * s + kq, cl = kl / NTAPS, tap = kl - cl * NTAPS, ky = tap / KHW, kx = tap - ky * KHW;
            baddr[j][s] = cl * NP + pp0 + ky * DH * Wp + kx * DW;
        }
    }

    v4f acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: chunk 0 in patch buffer 0, chunk 1's bytes in flight, the fragments of super-steps 0..2 in slots 0..2 ---------
    aload(std::integral_constant<int, 0>{}, 0);
    aload(std::integral_constant<int, 1>{}, 1);
    aload(std::integral_constant<int, 2>{}, 2);
    pload(0);
    pstore(0);
    pload(1);
    __syncthreads();
    // one super-step: patch buffer BUF, super-step U of the chunk == fragment slot -- all compile time, so every LDS address is
    // a loop-invariant register plus an immediate and no register array is indexed dynamically.  Global latency here is 1-2 us
    // under load and a super-step is ~0.5 us of MFMA: the fragments are requested RA - 1 super-steps ahead.
    float bfr[3][TN];                                    // B values, read TWO MFMA steps ahead (an LDS read takes longer than a step's MFMAs)
    auto bread = [&](auto BUF, auto U, auto S) {         // step S of super-step U (S may run past SS into the chunk's next super-step)
        constexpr int buf = decltype(BUF)::value, sl = decltype(S)::value, u = decltype(U)::value + sl / SS, s = sl % SS;
        constexpr int slot = (decltype(U)::value * SS + sl) % 3;
        if constexpr (u < CPC) {
            const float* pb = smem + (buf * CC + u * CSS) * NP;
            if (ABL & 2) {
#pragma unroll
                for (int j = 0; j < TN; j++) bfr[slot][j] = __builtin_bit_cast(float, baddr[j][s]);
                return;
            }
#pragma unroll
            for (int j = 0; j < TN; j++) bfr[slot][j] = pb[baddr[j][s]];
        }
    };
    auto superstep = [&](auto BUF, auto U, int ss, int c) {
        constexpr int u = decltype(U)::value, slot = (decltype(BUF)::value * CPC + u) % RA;
        aload(std::integral_constant<int, (slot + RA - 1) % RA>{}, ss + RA - 1);
        if constexpr (u == 0) { bread(BUF, U, std::integral_constant<int, 0>{}); bread(BUF, U, std::integral_constant<int, 1>{}); }
        auto step = [&](auto S) {
            constexpr int s = decltype(S)::value, bslot = (u * SS + s) % 3;
            bread(BUF, U, std::integral_constant<int, s + 2>{});
            __builtin_amdgcn_sched_barrier(0);           // the reads for step s + 2 go out BEFORE the MFMAs of step s
            refresh(BUF, U, S, c);                       // (a few VALU + one LDS write + one byte load in the MFMAs' shadow)
#pragma unroll
            for (int i = 0; i < TM; i++) {
                const float av = afrag(std::integral_constant<int, slot>{}, i, S);
#pragma unroll
                for (int j = 0; j < TN; j++) {
                    if (ABL & 1) acc[i][j][0] += av * bfr[bslot][j];
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bfr[bslot][j], acc[i][j], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{});
        step(std::integral_constant<int, 3>{});
        if constexpr (SS > 4) {
            step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{}); step(std::integral_constant<int, 6>{});
            step(std::integral_constant<int, 7>{}); step(std::integral_constant<int, 8>{});
        }
    };
    auto chunk = [&](auto BUF, int c) {
        const int ss0 = c * CPC;
        superstep(BUF, std::integral_constant<int, 0>{}, ss0, c);
        if (ss0 + 1 < nss) superstep(BUF, std::integral_constant<int, 1>{}, ss0 + 1, c);
        if (ss0 + 2 < nss) superstep(BUF, std::integral_constant<int, 2>{}, ss0 + 2, c);
        if (ss0 + 3 < nss) superstep(BUF, std::integral_constant<int, 3>{}, ss0 + 3, c);
        __syncthreads();                                 // the other buffer is complete, this one is free
    };
    for (int c = 0; c < nchunk; c += 2) {
        chunk(std::integral_constant<int, 0>{}, c);
        if (c + 1 < nchunk) chunk(std::integral_constant<int, 1>{}, c + 1);
    }

    // ---- epilogue (conv_u8_body's, main pixels): D[row = 4*kq + e][col = l15] of each 16x16 tile ------------------------------
    if (ABL & 16) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (t == 1234.5f) a.y[tid] = 1;
        return;
    }
#pragma unroll
    for (int j = 0; j < TN; j++) {
        const int pj = jbase + (wn * TN + j) * 16 + l15;
        if (pj >= jlimit) continue;
        int oy, ox;
        conv_pixel(a, pj, &oy, &ox);
        const int opix = oy * a.OW + ox;
#pragma unroll
        for (int i = 0; i < TM; i++) {
            const int co = co0 + (wm * TM + i) * 16 + 4 * kq;
            if (co >= a.cout) continue;
            const float s4[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            u8_finish4(a, s4, co, n, OHW, opix, (oy >> 1) * (a.OW >> 1) + (ox >> 1), (l15 & 3) == 0, rq_inv, tail);
        }
    }
}

// tile configurations: waves along cout x waves along pixels, 16x16 tiles per wave along cout x along pixels
static const struct { int wm, wn, tm, tn; const char* n3; const char* n1; } U8P_CFGS[] = {
    {2, 2, 2, 2, "conv_u8_patch_64x64<3x3>", "conv_u8_patch_64x64<1x1>"},
    {4, 1, 2, 4, "conv_u8_patch_128x64<3x3>", "conv_u8_patch_128x64<1x1>"},
    {2, 2, 2, 4, "conv_u8_patch_64x128<3x3>", "conv_u8_patch_64x128<1x1>"},
    {1, 4, 2, 1, "conv_u8_patch_32x64<3x3>", "conv_u8_patch_32x64<1x1>"}};
static constexpr int U8P_LANES = 4;                 // configuration 4: no MFMA tiles at all, every output a lane-level chain (conv_u8_lanes_k)
// configurations 5 .. 8 (round 4): tile shapes 0 .. 3 with 2-D pixel tiles (8 rows x BN/8 columns) -- wide maps, where a run of 64
// consecutive pixels drags 3-6 whole input rows per channel chunk into LDS (YOLOv3-tiny conv1 / conv2: 208- and 104-wide).
// EXPERIMENT BUILDS ONLY (-DTAMD_EXPERIMENTS, then TAMD_U8_PATCH_2D=1): byte-exact in round 4's suite, but measured it only wins conv2
// in isolation (61.6 vs 66.1 us) and not inside the pass (851.3 vs 856.6 us per step), and loses conv1 (cout 32, one chunk of K: a block
// is all prologue and epilogue, 110-246 vs 83 us) -- profiles/r04_experiment_u8_patch_2d_tiles.txt.  The product offers 0 .. 4.
static constexpr int U8P_2D = 5;
#ifdef TAMD_EXPERIMENTS
int conv_u8_patch_num_cfgs() { return 9; }
#else
int conv_u8_patch_num_cfgs() { return 5; }
#endif
int conv_u8_patch_lanes_cfg() { return U8P_LANES; }
static int u8p_base(int cfg) { return cfg >= U8P_2D ? cfg - U8P_2D : cfg; }
int conv_u8_patch_bm(int cfg) { return cfg == U8P_LANES ? 64 : U8P_CFGS[u8p_base(cfg)].wm * U8P_CFGS[u8p_base(cfg)].tm * 16; }
static int u8p_bn(int cfg) { return cfg == U8P_LANES ? 4 : U8P_CFGS[u8p_base(cfg)].wn * U8P_CFGS[u8p_base(cfg)].tn * 16; }
int conv_u8_patch_ss(const U8ConvArgs& a) { return (a.pk_kh == 3 && a.pk_kw == 3) ? 9 : (a.pk_kh == 1 && a.pk_kw == 1) ? 4 : 0; }
const char* conv_u8_patch_kernel_name(const U8ConvArgs& a)
{
    static const char* n2d[4] = {"conv_u8_patch_64x64<3x3,2d>", "conv_u8_patch_128x64<3x3,2d>", "conv_u8_patch_64x128<3x3,2d>", "conv_u8_patch_32x64<3x3,2d>"};
    if (a.pk_cfg == U8P_LANES) return a.pk_kh == 3 ? "conv_u8_lanes<3x3>" : "conv_u8_lanes<1x1>";
    if (a.pk_cfg >= U8P_2D) return n2d[a.pk_cfg - U8P_2D];
    return a.pk_kh == 3 ? U8P_CFGS[a.pk_cfg].n3 : U8P_CFGS[a.pk_cfg].n1;
}

static size_t u8p_lds(const U8ConvArgs& a)
{
    const int cc = a.pk_kh == 3 ? 16 : 64;
    return (size_t)(2 * cc * a.pk_npad) * 4;
}

// fills the patch fields of `a` for tile configuration cfg; false: this convolution does not go through the patch kernel
bool conv_u8_patch_prepare(U8ConvArgs& a, int cfg, int KH, int KW, int DH, int DW)
{
    const char* env = tamd_pin("u8_patch");
    const bool off = env && atoi(env) == 0;
    a.pk_cfg = -1;
    a.pk_tw = 0;
    a.pk_kh = KH; a.pk_kw = KW; a.pk_dh = DH; a.pk_dw = DW;
    const int ss = conv_u8_patch_ss(a);
    const int OHW = a.OH * a.OW, N8 = OHW & ~7, bn = u8p_bn(cfg);
    if (off || !ss || a.K % (4 * ss) != 0 || a.C % (KH == 3 ? 4 : 16) != 0 || a.PH < 0 || a.PW < 0) return false;
    if (cfg == U8P_LANES) {
        // lane-level chains for the whole layer: bounded to small layers (a chain is K dependent steps; 4096 waves are 4 per SIMD),
        // no fused pool (the pool's window-major pixel order belongs to the MFMA tiles).  TAMD_U8_LANES=0: never
        const char* le = tamd_pin("u8_lanes");                // (read at every prerun: tests and A/B runs flip it inside one process)
        const bool lanes_ok = !(le && atoi(le) == 0);
        const long waves = ((long)a.N * N8 + 3) / 4 * ((a.cout + 15) / 16) + (long)a.N * (OHW - N8) * ((a.cout + 15) / 16);
        if (!lanes_ok || a.pool.on || waves > 4096 || (size_t)a.K * 16 > 150 * 1024 || (size_t)a.C * a.H * a.W >= (1u << 31)) return false;
        a.pk_wp = 0; a.pk_npad = 4; a.pk_cfg = cfg;
        return true;
    }
    if (N8 == 0) return false;                                   // no main pixel: nothing for the MFMA tiles (the lanes configuration takes these)
    a.pk_tw = 0;
    if (cfg >= U8P_2D) {
#ifndef TAMD_EXPERIMENTS
        return false;
#endif
        // 2-D tiles: 3x3 only (a 1x1 patch is the tile's own pixels either way), whole tiles only, no tail pixels (the reference's
        // main / tail split is a property of the ROW-MAJOR pixel index: with OH*OW % 8 == 0 every pixel is a main pixel in any order)
        const char* e2 = exp_env("TAMD_U8_PATCH_2D");
        const int tw = bn / 8;
        if (!(e2 && atoi(e2) == 1) || KH != 3 || OHW != N8 || a.OH % 8 != 0 || a.OW % tw != 0 || (a.pool.on && (tw & 1))) return false;
        if ((size_t)a.C * a.H * a.W >= (1u << 31)) return false;
        a.pk_wp = (tw - 1) * a.SW + (KW - 1) * DW + 1;
        const int worst2 = (7 * a.SH + (KH - 1) * DH + 1) * a.pk_wp;
        if (worst2 > 512) return false;
        a.pk_tw = tw;
        a.pk_npad = worst2 <= 256 ? 256 : 512;
        a.pk_cfg = cfg;
        return true;
    }
    if ((size_t)a.C * a.H * a.W >= (1u << 31) || (size_t)a.K * 4 > 150 * 1024) return false;      // (the tail blocks keep an im2col column in LDS)
    if (KH == 1) { a.pk_wp = 0; a.pk_npad = bn; a.pk_cfg = cfg; return true; }      // the patch is the tile's own pixels
    a.pk_wp = (a.OW - 1) * a.SW + (KW - 1) * DW + 1;
    if (a.pk_wp < a.W + a.PW) a.pk_wp = a.W + a.PW;             // every column a tap can name: [-PW, max(W, last tap) )
    // rows the worst pixel tile touches (window-major enumeration under a fused pool: two output rows per window row)
    int worst = 0;
    for (int j0 = 0; j0 < N8; j0 += bn) {
        const int j1 = std::min(j0 + bn, N8) - 1;
        int oy0, oy1;
        if (a.pool.on) { const int half = a.OW >> 1; oy0 = 2 * ((j0 >> 2) / half); oy1 = 2 * ((j1 >> 2) / half) + 1; }
        else { oy0 = j0 / a.OW; oy1 = j1 / a.OW; }
        worst = std::max(worst, ((oy1 - oy0) * a.SH + (KH - 1) * DH + 1) * a.pk_wp);
    }
    if (worst > 512) return false;
    a.pk_npad = worst <= 256 ? 256 : 512;                        // the two plane strides the kernel is compiled for
    a.pk_cfg = cfg;
    return true;
}

// the dequantised weights in A-fragment order: the same for every tile configuration
size_t conv_u8_patch_packed_bytes(const U8ConvArgs& a)
{
    return (size_t)((a.cout + 15) / 16 + 8) * a.K * 16 * 4;        // + 8 tiles: the last block's waves may fetch rows past cout
}

void conv_u8_patch_pack(const U8ConvArgs& a, const uint8_t* w, uint8_t w_zp, float w_scale, float* out)
{
    const int ss = conv_u8_patch_ss(a), nss = a.K / (4 * ss), g4 = ss / 4, rem = ss - 4 * g4, frag = ss * 64;
    const size_t total = conv_u8_patch_packed_bytes(a) / 4;
    for (size_t i = 0; i < total; i++) out[i] = 0.f;
    for (int co = 0; co < a.cout; co++)
        for (int k = 0; k < a.K; k++) {
            const int st = k / (4 * ss), kl = k % (4 * ss), s = kl >> 2, kq = kl & 3, lane = kq * 16 + (co & 15);
            const size_t base = ((size_t)(co / 16) * nss + st) * frag;
            const size_t at = s < 4 * g4 ? base + (size_t)(s / 4) * 256 + lane * 4 + (s & 3) : base + (size_t)g4 * 256 + lane * rem + (s - 4 * g4);
            out[at] = ((float)w[(size_t)co * a.K + k] - (float)w_zp) * w_scale;      // conv_u8_body's dequant(), computed once
        }
}

hipError_t launch_conv_u8_patch(const U8ConvArgs& a, hipStream_t s)
{
    const int OHW = a.OH * a.OW, N8 = OHW & ~7;
    if (a.pk_cfg == U8P_LANES) {
        const int slices = (a.cout + 63) / 64;
        const int main_blocks = (a.N * N8 + 3) / 4 * slices, tail_blocks = (OHW - N8) * a.N * slices;
        const size_t lds = main_blocks ? (size_t)(a.K + 4) * 16 : (size_t)a.K * 4;
        auto go = [&](auto kern) {
            if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(kern, dim3(main_blocks + tail_blocks), dim3(256), lds, s, a, main_blocks);
            return hipGetLastError();
        };
        return a.pk_kh == 3 ? go(conv_u8_lanes_k<3>) : go(conv_u8_lanes_k<1>);
    }
    const int bm = conv_u8_patch_bm(a.pk_cfg), bn = u8p_bn(a.pk_cfg);
    const int main_blocks = ((N8 + bn - 1) / bn) * a.N * ((a.cout + bm - 1) / bm);
    const int tail_blocks = (OHW - N8) * a.N * ((a.cout + 63) / 64);        // conv_u8_patch_tail: (image, tail pixel, 64 channels)
    const dim3 grid(main_blocks + tail_blocks, 1, 1);
    const size_t lds = std::max(u8p_lds(a), tail_blocks ? (size_t)a.K * 4 : (size_t)0);
    auto go = [&](auto kern) {
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
        return hipGetLastError();
    };
    hipError_t e = hipErrorInvalidValue;
#define U8P_GO(WM, WN, TM, TN)                                                                                                   \
    e = a.pk_kh == 1 ? go(conv_u8_patch_k<WM, WN, TM, TN, 1, 0>)                                                                  \
                     : a.pk_npad == 256 ? go(conv_u8_patch_k<WM, WN, TM, TN, 3, 256>) : go(conv_u8_patch_k<WM, WN, TM, TN, 3, 512>)
    switch (u8p_base(a.pk_cfg)) {
    case 0: U8P_GO(2, 2, 2, 2); break;
    case 1: U8P_GO(4, 1, 2, 4); break;
    case 2: U8P_GO(2, 2, 2, 4); break;
    case 3: U8P_GO(1, 4, 2, 1); break;
    default: break;
    }
#undef U8P_GO
    return e;
}


}  // namespace tamd
