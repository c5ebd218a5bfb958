// uint8 (per-tensor asymmetric) kernels.
//
// The reference does not compute uint8 in integers: it dequantises both operands to fp32, runs its fp32 code and
// requantises the result.  The bytes it produces therefore depend on the exact fp32
// operation sequence -- for convolution on the summation ORDER of its 8x8-tiled AVX sgemm.  To be byte-identical
// this file performs the same IEEE binary32 operations in the same order per output element:
//   * every `a*b + c` the reference's compiler contracts (-O3 -mfma, default -ffp-contract=fast) is one
//     __builtin_fmaf here; nothing else is fused (this TU is built -ffp-contract=off);
//   * divisions are correctly rounded (__fdiv_rn), round() is round-half-away (roundf).
//   * where the reference's compiler did NOT fuse (a loop-invariant product it hoisted, e.g. the conv bias term) the
//     product is rounded first -- read off the reference object's disassembly, not guessed from the C text.
// A sequential chain per output element cannot split K; the parallelism is across output elements, and the fp32
// MFMA instructions happen to accumulate in exactly that sequential fused order (see below).
// Activations stay in the reference's dense NCHW order (lanes along pixels read consecutive bytes).
#include <hip/hip_runtime.h>
#include "env.h"

#include <cstdlib>
#include <type_traits>
#include <algorithm>

#include "kernels.h"
#include "u8_epilogue.h"

namespace tamd {

// =================================================================================================================
// group == 1 convolution: conv/x86/conv_kernel_x86.c:68-80 (weights -> fp32), :126-185 (im2col_uint8, k = (c,ky,kx),
// 0.0f at out-of-image taps), :322-960 sgemm_fp, :1703-1794 bias / activation / requantise.
// Per image the GEMM is [cout] x [OH*OW] x [K]; an element's summation order depends on its place in the
// reference's tiling:
//   pixel j <  (OH*OW)&~7 : one fused chain over k = 0..K-1                       -> "main" blocks
//   pixel j >= (OH*OW)&~7 : four fused chains over k = r (mod 4), k < K&~3, combined
//                           ((0+(s0+s1))+(s2+s3)) for rows in an 8-/4-row block, ((s0+s1)+s2)+s3 for the last
//                           cout%4 rows, then the fused chain over the K%4 tail    -> "tail" blocks (same launch)
//
// The chains run on the MATRIX cores: v_mfma_f32_16x16x4f32 accumulates D = C + a0*b0 + a1*b1 + a2*b2 + a3*b3 as
// four IEEE fused multiply-adds in ascending k -- measured bit for bit against fmaf() chains of 4608 steps,
// profiles/r01_mfma_f32_is_sequential_fma_chain.txt (tools/exp/mfma_f32_exact.hip) -- so issuing the MFMAs of
// one accumulator tile in ascending k IS the reference's chain.  A = weights (rows = channels), B = dequantised
// im2col columns (cols = pixels): D lanes run along pixels, i.e. along the NCHW output rows.
//   * K is staged 32 at a time through LDS (double buffered, one barrier per stage) from a 3-deep REGISTER ring of
//     global loads (raw bytes + packed weights), so three stages of HBM/L2 latency are always in flight per block;
//   * inside a stage the 32 k are stored class-major (k%4, then k/4; rows padded to 36 floats: conflict-free
//     b128): main tiles feed MFMA i with (class kq = lane/16, position i) = k0+4i+kq; tail tiles feed chain r
//     with (class r, position 4j+kq) = k0+r+4(4j+kq): same data, same MFMA count, four accumulators;
//   * the k -> (c,ky,kx) tap table (one packed dword per k) lives in LDS for the whole kernel.
// Padded k rows carry w = 0 and an out-of-image tap: fma(0, 0, s) == s.
// =================================================================================================================
typedef float v4f __attribute__((ext_vector_type(4)));

template <int WM, int WN, int TM, int TN, int KC, bool TAIL>
__device__ __forceinline__ void conv_u8_body(const U8ConvArgs& a, float* __restrict__ ws, float* __restrict__ xs,
                                             const unsigned* __restrict__ lut, int n, int jbase, int jlimit, int co0, const uint8_t* tail)
{
    const float rq_inv = __fdiv_rn(1.0f, a.out_scale);
    constexpr int BM = WM * TM * 16, BN = WN * TN * 16, LD = KC + 4, NT = WM * WN * 64, D = 3;
    constexpr int NPOS = KC / 4;                         // slots per class (k%4) in a row == quads (float4) per row
    constexpr int QPC = NPOS / 4;                        // quads per class
    constexpr int XQ = BN * NPOS / NT;                   // x quads (4 k of one class, one pixel) per thread per stage
    constexpr int WQ = (BM * NPOS + NT - 1) / NT;        // weight quads (float4) per thread per stage
    static_assert(KC == 32 || KC == 64, "stage depth");
    constexpr int NCH = TAIL ? 4 : 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM, l15 = lane & 15, kq = lane >> 4;
    const int K4 = a.K & ~3;

    // staging role: pixel column sp of the x tile, quads su*XQ .. su*XQ+XQ-1 (quad qd: class qd/2, positions 4*(qd%2)..+3)
    const int sp = tid % BN, su = tid / BN;
    const int sj = jbase + sp;
    const bool svalid = sj < jlimit;
    int soy = 0, sox = 0;
    if (svalid) conv_pixel(a, sj, &soy, &sox);
    const int iy0 = soy * a.SH - a.PH, ix0 = sox * a.SW - a.PW;
    const uint8_t* xin = a.x + (size_t)n * a.C * a.H * a.W;
    const int pbase = iy0 * a.W + ix0;

    unsigned xr[D][XQ][4];       // raw bytes, one register each: packing here would make the loads wait at once
    unsigned xok[D];             // bit (4*i + e): element e of quad i is inside the image
    int k0s[D];                  // first k of the stage held in the slot
    const unsigned* wtile = reinterpret_cast<const unsigned*>(a.wq) + (size_t)(co0 / BM) * (a.Kpad / KC) * (BM * NPOS);
    unsigned wr[D][WQ];          // 4 raw weight bytes of one quad
    auto gload = [&](int d, int k0) {
        xok[d] = 0;
        k0s[d] = k0;
#pragma unroll
        for (int i = 0; i < XQ; i++) {
            const int qd = su * XQ + i, c = qd / QPC, pos0 = (qd % QPC) * 4;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const unsigned t = lut[k0 + c + 4 * (pos0 + e)];            // off | dx << 24 | dy << 28
                const int iy = iy0 + (int)(t >> 28), ix = ix0 + (int)((t >> 24) & 15);
                const bool ok = svalid & ((unsigned)iy < (unsigned)a.H) & ((unsigned)ix < (unsigned)a.W);
                xr[d][i][e] = xin[ok ? pbase + (int)(t & 0xffffffu) : 0];
                xok[d] |= ok ? 1u << (4 * i + e) : 0u;
            }
        }
#pragma unroll
        for (int i = 0; i < WQ; i++) {
            const int idx = tid + NT * i;
            // the block's weight tile of a stage is BM*KC contiguous bytes ([cout tile][stage][row][KC slots])
            if (BM * NPOS % NT == 0 || idx < BM * NPOS) wr[d][i] = wtile[(size_t)(k0 / KC) * (BM * NPOS) + idx];
        }
    };
    auto sstore = [&](int d, int buf) {
#pragma unroll
        for (int i = 0; i < XQ; i++) {
            const int qd = su * XQ + i;
            float4 v;
            v.x = (xok[d] >> (4 * i + 0) & 1u) ? dequant((uint8_t)xr[d][i][0], a.in_zp, a.in_scale) : 0.f;
            v.y = (xok[d] >> (4 * i + 1) & 1u) ? dequant((uint8_t)xr[d][i][1], a.in_zp, a.in_scale) : 0.f;
            v.z = (xok[d] >> (4 * i + 2) & 1u) ? dequant((uint8_t)xr[d][i][2], a.in_zp, a.in_scale) : 0.f;
            v.w = (xok[d] >> (4 * i + 3) & 1u) ? dequant((uint8_t)xr[d][i][3], a.in_zp, a.in_scale) : 0.f;
            *reinterpret_cast<float4*>(xs + (buf * BN + sp) * LD + qd * 4) = v;
        }
#pragma unroll
        for (int i = 0; i < WQ; i++) {
            const int idx = tid + NT * i, row = idx / NPOS, qd = idx % NPOS;
            if (BM * NPOS % NT == 0 || idx < BM * NPOS) {
                // conv_kernel_x86.c:68-80: w_fp32 = ((float)w - (float)zp) * scale
                float4 w;
                w.x = dequant((uint8_t)wr[d][i], a.w_zp, a.w_scale);
                w.y = dequant((uint8_t)(wr[d][i] >> 8), a.w_zp, a.w_scale);
                w.z = dequant((uint8_t)(wr[d][i] >> 16), a.w_zp, a.w_scale);
                w.w = dequant((uint8_t)(wr[d][i] >> 24), a.w_zp, a.w_scale);
                if (TAIL) {                                  // the K%4 remainder is chained after the combine
                    const int kb = k0s[d] + qd / QPC + 16 * (qd % QPC);
                    if (kb >= K4) w.x = 0.f;
                    if (kb + 4 >= K4) w.y = 0.f;
                    if (kb + 8 >= K4) w.z = 0.f;
                    if (kb + 12 >= K4) w.w = 0.f;
                }
                *reinterpret_cast<float4*>(ws + (buf * BM + row) * LD + qd * 4) = w;
            }
        }
    };

    v4f acc[NCH][TM][TN];
#pragma unroll
    for (int r = 0; r < NCH; r++)
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++) acc[r][i][j] = v4f{0.f, 0.f, 0.f, 0.f};

    const int nchunk = a.Kpad / KC;
#pragma unroll
    for (int d = 0; d < D; d++)
        if (d < nchunk) gload(d, d * KC);
    // one K stage: ring slot d -> LDS buffer ch&1, refill the slot with stage ch+D, barrier, MFMAs
    auto stage = [&](int ch, int d, bool refill) {
        const int cur = ch & 1;
        __builtin_amdgcn_sched_barrier(0);                    // keep the scheduler from hoisting younger stages' unpacking
        sstore(d, cur);                                       // (and with it their s_waitcnt) above this stage's MFMAs
        if (refill) gload(d, (ch + D) * KC);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        const float* wsb = ws + cur * BM * LD;
        const float* xsb = xs + cur * BN * LD;
        if constexpr (!TAIL) {
            float af[TM][NPOS], bf[TN][NPOS];
#pragma unroll
            for (int i = 0; i < TM; i++) {
                const float* q = wsb + ((wm * TM + i) * 16 + l15) * LD + kq * NPOS;
#pragma unroll
                for (int v = 0; v < QPC; v++) {
                    const float4 f = *reinterpret_cast<const float4*>(q + 4 * v);
                    af[i][4 * v] = f.x; af[i][4 * v + 1] = f.y; af[i][4 * v + 2] = f.z; af[i][4 * v + 3] = f.w;
                }
            }
#pragma unroll
            for (int j = 0; j < TN; j++) {
                const float* q = xsb + ((wn * TN + j) * 16 + l15) * LD + kq * NPOS;
#pragma unroll
                for (int v = 0; v < QPC; v++) {
                    const float4 f = *reinterpret_cast<const float4*>(q + 4 * v);
                    bf[j][4 * v] = f.x; bf[j][4 * v + 1] = f.y; bf[j][4 * v + 2] = f.z; bf[j][4 * v + 3] = f.w;
                }
            }
#pragma unroll
            for (int s = 0; s < NPOS; s++)
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int j = 0; j < TN; j++)
                        acc[0][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][s], bf[j][s], acc[0][i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int s = 0; s < QPC; s++)
#pragma unroll
                    for (int i = 0; i < TM; i++)
#pragma unroll
                        for (int j = 0; j < TN; j++) {
                            if (jbase + (wn * TN + j) * 16 >= jlimit) continue;      // no tail pixel in this 16-pixel column
                            const float av = wsb[((wm * TM + i) * 16 + l15) * LD + r * NPOS + 4 * s + kq];
                            const float bv = xsb[((wn * TN + j) * 16 + l15) * LD + r * NPOS + 4 * s + kq];
                            acc[r][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[r][i][j], 0, 0, 0);
                        }
        }
    };
    int ch = 0;
    // steady state: every refill is in range, the body is branch-free so the load counters stay exact
    for (; ch + 2 * D <= nchunk; ch += D) {
#pragma unroll
        for (int d = 0; d < D; d++) stage(ch + d, d, true);
    }
    for (; ch < nchunk; ch += D) {                            // drain: at most 2*D-1 stages
#pragma unroll
        for (int d = 0; d < D; d++)
            if (ch + d < nchunk) stage(ch + d, d, ch + d + D < nchunk);
    }

    // ---- epilogue: D[row = 4*kq + e][col = l15] of each 16x16 tile ------------------------------------------
    const int OHW = a.OH * a.OW;
#pragma unroll
    for (int j = 0; j < TN; j++) {
        const int pj = jbase + (wn * TN + j) * 16 + l15;
        if (pj >= jlimit) continue;
        int oy, ox;
        conv_pixel(a, pj, &oy, &ox);
        const int opix = oy * a.OW + ox;
#pragma unroll
        for (int i = 0; i < TM; i++) {
            const int cb = co0 + (wm * TM + i) * 16 + 4 * kq;
            if (cb >= a.cout) continue;
            float s4[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int co = min(cb + e, a.cout - 1);          // (rows past cout repeat the last one: never stored)
                float s;
                if constexpr (TAIL) {
                    const float s0 = acc[0][i][j][e], s1 = acc[1][i][j][e], s2 = acc[2][i][j][e], s3 = acc[3][i][j][e];
                    if (co < a.m_blocked) s = (0.f + (s0 + s1)) + (s2 + s3);
                    else s = ((s0 + s1) + s2) + s3;
                    for (int k = K4; k < a.K; k++) {
                        const unsigned t = lut[k];
                        const int iy = oy * a.SH - a.PH + (int)(t >> 28), ix = ox * a.SW - a.PW + (int)((t >> 24) & 15);
                        float v = 0.f;
                        if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
                            v = dequant(xin[(oy * a.SH - a.PH) * a.W + ox * a.SW - a.PW + (int)(t & 0xffffffu)], a.in_zp, a.in_scale);
                        // packed position of k inside its 32-chunk: class k%4, position (k%32)/4
                        const int kl = k % KC;
                        const uint8_t wb = a.wq[((size_t)(co0 / BM) * (a.Kpad / KC) + k / KC) * (BM * KC) + (co - co0) * KC + (kl & 3) * NPOS + (kl >> 2)];
                        s = __builtin_fmaf(dequant(wb, a.w_zp, a.w_scale), v, s);
                    }
                } else
                    s = acc[0][i][j][e];
                s4[e] = s;
            }
            u8_finish4(a, s4, cb, n, OHW, opix, (oy >> 1) * (a.OW >> 1) + (ox >> 1), (l15 & 3) == 0, rq_inv, tail);
        }
    }
}

template <int WM, int WN, int TM, int TN, int KC>
__global__ __launch_bounds__(WM * WN * 64) void conv_u8_gemm_k(const U8ConvArgs a)
{
    constexpr int BM = WM * TM * 16, BN = WN * TN * 16, LD = KC + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* ws = smem;                               // [2][BM][LD]
    float* xs = smem + 2 * BM * LD;                 // [2][BN][LD]
    unsigned* lut = reinterpret_cast<unsigned*>(smem + 2 * (BM + BN) * LD);   // [Kpad]
    for (int k = threadIdx.x; k < a.Kpad; k += WM * WN * 64) lut[k] = a.klut[k];
    __shared__ uint8_t tail[512];                   // fused ReLU / pool nodes as byte tables (u8_epilogue.h)
    u8_tail_tables(tail, threadIdx.x, WM * WN * 64, a.relu, a.out_scale, a.out_zp, a.pool);
    __syncthreads();
    const int OHW = a.OH * a.OW, N8 = OHW & ~7;
    const int tiles = (N8 + BN - 1) / BN, tpi = tiles + (OHW != N8);
    // x = (image, pixel tile): blocks that stream the same weight tile are neighbours in launch order (L2 reuse)
    const int n = blockIdx.x / tpi, tile = blockIdx.x - n * tpi, co0 = blockIdx.y * BM;
    if (tile < tiles) conv_u8_body<WM, WN, TM, TN, KC, false>(a, ws, xs, lut, n, tile * BN, N8, co0, tail);
    else conv_u8_body<WM, WN, TM, TN, KC, true>(a, ws, xs, lut, n, N8, OHW, co0, tail);
}

// configurations: block tile (channels x pixels) and K stage depth.  conv_u8_gemm_pick is the geometry heuristic (the
// largest tile that still gives every CU two blocks); the planner's autotune times all of them and stores the index in
// a.cfg.  The 64-deep stages halve the barriers / exposed load latencies per K at twice the LDS.
static const struct { int bm, bn, kc; const char* name; } U8_CFGS[] = {
    {16, 64, 32, "conv_u8_mfma_16x64"}, {32, 32, 32, "conv_u8_mfma_32x32"}, {64, 64, 32, "conv_u8_mfma_64x64"},
    {32, 64, 32, "conv_u8_mfma_32x64"}, {16, 16, 32, "conv_u8_mfma_16x16"},
    {64, 64, 64, "conv_u8_mfma_64x64k64"}, {32, 64, 64, "conv_u8_mfma_32x64k64"}, {32, 32, 64, "conv_u8_mfma_32x32k64"}};

int conv_u8_gemm_num_cfgs() { return 8; }
int conv_u8_gemm_pick(const U8ConvArgs& a)
{
    const char* e = tamd_pin("u8_cfg");                 // tests / fuzzing: pin one tile shape (read at every prerun)
    if (e && *e) return atoi(e) % 8;
    const int OHW = a.OH * a.OW, N8 = OHW & ~7;
    auto blocks = [&](int bm, int bn) { return (long)((N8 + bn - 1) / bn + (OHW & 7 ? 1 : 0)) * ((a.cout + bm - 1) / bm) * a.N; };
    if (a.cout <= 16) return 0;
    if (a.cout <= 32) return 3;
    if (blocks(64, 64) >= 512) return 2;
    if (blocks(32, 64) >= 384) return 3;
    if (blocks(32, 32) >= 512) return 1;
    return 4;                                          // one wave per block: the most blocks (latency-bound layers)
}
size_t conv_u8_gemm_lds(const U8ConvArgs& a)
{
    return (size_t)(2 * (U8_CFGS[a.cfg].bm + U8_CFGS[a.cfg].bn) * (U8_CFGS[a.cfg].kc + 4) + a.Kpad) * 4;
}
const char* conv_u8_gemm_kernel_name(const U8ConvArgs& a) { return U8_CFGS[a.cfg].name; }
int conv_u8_gemm_bm(int cfg) { return U8_CFGS[cfg].bm; }
int conv_u8_gemm_kc(int cfg) { return U8_CFGS[cfg].kc; }

hipError_t launch_conv_u8_gemm(const U8ConvArgs& a, hipStream_t s)
{
    const int OHW = a.OH * a.OW, N8 = OHW & ~7, ntail = OHW - N8;
    const int bm = U8_CFGS[a.cfg].bm, bn = U8_CFGS[a.cfg].bn;
    const dim3 grid(((N8 + bn - 1) / bn + (ntail ? 1 : 0)) * a.N, (a.cout + bm - 1) / bm, 1);
    const size_t lds = conv_u8_gemm_lds(a);
    auto go = [&](auto kern, int threads) {
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, grid, dim3(threads), lds, s, a);
        return hipGetLastError();
    };
    switch (a.cfg) {
    case 0: return go(conv_u8_gemm_k<1, 4, 1, 1, 32>, 256);
    case 2: return go(conv_u8_gemm_k<2, 2, 2, 2, 32>, 256);
    case 3: return go(conv_u8_gemm_k<2, 2, 1, 2, 32>, 256);
    case 4: return go(conv_u8_gemm_k<1, 1, 1, 1, 32>, 64);
    case 5: return go(conv_u8_gemm_k<2, 2, 2, 2, 64>, 256);
    case 6: return go(conv_u8_gemm_k<2, 2, 1, 2, 64>, 256);
    case 7: return go(conv_u8_gemm_k<2, 2, 1, 1, 64>, 256);
    default: return go(conv_u8_gemm_k<2, 2, 1, 1, 32>, 256);
    }
}

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

// =================================================================================================================
// grouped / depthwise convolution: conv/conv_kernel_ref_uint8.c:42-195 -- one fused chain in (kc, ky, kx) order over
// the in-image taps, + bias_fp32 = ((float)b * in_s) * k_s as a separate add, naive-ref activation, requantise.
// One thread per output element, lanes along the output row (consecutive input bytes for stride 1).
// =================================================================================================================
__global__ __launch_bounds__(256) void conv_u8_direct_k(const U8DirectArgs a)
{
    const int OHW = a.OH * a.OW;
    const int pj = blockIdx.x * 256 + threadIdx.x;
    const int oc = blockIdx.y, n = blockIdx.z;
    if (pj >= OHW) return;
    const int cin_g = a.C / a.group, cout_g = a.cout / a.group, g = oc / cout_g;
    const int oy = pj / a.OW, ox = pj - oy * a.OW;
    const float* wk = a.wf + (size_t)oc * cin_g * a.KH * a.KW;
    float total = 0.f;
    for (int kc = 0; kc < cin_g; kc++) {
        const uint8_t* xc = a.x + ((size_t)n * a.C + (size_t)g * cin_g + kc) * a.H * a.W;
        for (int ky = 0; ky < a.KH; ky++) {
            const int iy = oy * a.SH - a.PH + ky * a.DH;
            if ((unsigned)iy >= (unsigned)a.H) continue;
            for (int kx = 0; kx < a.KW; kx++) {
                const int ix = ox * a.SW - a.PW + kx * a.DW;
                if ((unsigned)ix >= (unsigned)a.W) continue;
                total = __builtin_fmaf(dequant(xc[iy * a.W + ix], a.in_zp, a.in_scale), wk[(kc * a.KH + ky) * a.KW + kx], total);
            }
        }
    }
    if (a.bias) {
        float bf = (float)a.bias[oc] * a.in_scale;
        bf = bf * a.w_scale;
        total = total + bf;
    }
    if (a.act >= 0) {
        if (total < 0.f && a.act != 1) total = 0.f;
        if (total > 1.f && a.act == 1) total = 1.f;
        if (total > 6.f && a.act == 6) total = 6.f;
        if (total < -1.f && a.act == 1) total = -1.f;
    }
    uint8_t q = quant_round_sat_u8(total, a.out_scale, a.out_zp);
    if (a.relu.on) q = fused_relu(q, a.out_scale, a.out_zp, a.relu);
    a.y[(size_t)n * a.out_img + (size_t)(a.out_c0 + oc) * OHW + pj] = q;
}

// depthwise 3x3 (every MobileNet dw layer), same arithmetic as conv_u8_direct_k.  Threads run over the flattened
// (channel, pixel) index of one image so 7x7 and 14x14 maps still fill their wavefronts; the nine taps are loaded up
// front from clamped addresses -- one memory round trip instead of nine dependent ones -- and an out-of-image tap
// enters the chain as 0.0f, which leaves `total` unchanged exactly as the reference's `continue` does.
__global__ __launch_bounds__(256) void conv_u8_dw3_k(const U8DirectArgs a)
{
    const int OHW = a.OH * a.OW;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int n = blockIdx.y;
    if (idx >= a.cout * OHW) return;
    const int oc = idx / OHW, pj = idx - oc * OHW;
    const int oy = pj / a.OW, ox = pj - oy * a.OW;
    const uint8_t* xc = a.x + ((size_t)n * a.C + oc) * a.H * a.W;
    const float* wk = a.wf + (size_t)oc * 9;
    unsigned u[9];
    bool ok[9];
    float w[9];
#pragma unroll
    for (int t = 0; t < 9; t++) {
        const int iy = oy * a.SH - a.PH + (t / 3) * a.DH, ix = ox * a.SW - a.PW + (t % 3) * a.DW;
        ok[t] = ((unsigned)iy < (unsigned)a.H) & ((unsigned)ix < (unsigned)a.W);
        u[t] = xc[ok[t] ? iy * a.W + ix : 0];
        w[t] = wk[t];
    }
    float total = 0.f;
#pragma unroll
    for (int t = 0; t < 9; t++) {
        const float xf = ok[t] ? dequant((uint8_t)u[t], a.in_zp, a.in_scale) : 0.f;
        total = __builtin_fmaf(xf, w[t], total);
    }
    if (a.bias) {
        float bf = (float)a.bias[oc] * a.in_scale;
        bf = bf * a.w_scale;
        total = total + bf;
    }
    if (a.act >= 0) {
        if (total < 0.f && a.act != 1) total = 0.f;
        if (total > 1.f && a.act == 1) total = 1.f;
        if (total > 6.f && a.act == 6) total = 6.f;
        if (total < -1.f && a.act == 1) total = -1.f;
    }
    uint8_t q = quant_round_sat_u8_w(total, a.out_scale, __fdiv_rn(1.0f, a.out_scale), a.out_zp);
    if (a.relu.on) q = fused_relu(q, a.out_scale, a.out_zp, a.relu);
    a.y[(size_t)n * a.out_img + (size_t)a.out_c0 * OHW + idx] = q;
}

// The same, a TH x 4 block of outputs per thread (pad 1, dilation 1, stride S): the ((TH-1) S + 3) x (3S+3) input bytes they share
// are loaded, dequantised and masked once -- an input value costs ~5 instructions (extract, convert, subtract, scale, select) and a
// 1 x 4 block needs 4.5 of them per output, a 2 x 4 block 3, a 4 x 4 block 2.25 (stride 1) -- and the index arithmetic is paid once
// per block.  Each output still runs its own nine-step chain in (ky, kx) order.
template <int S, int TH>
__global__ __launch_bounds__(256) void conv_u8_dw3x4_k(const U8DirectArgs a)
{
    constexpr int NC = 3 * S + 3, NR = (TH - 1) * S + 3;
    const int QW = (a.OW + 3) >> 2, BH = (a.OH + TH - 1) / TH, per = BH * QW;
    // threads run over the flattened (image, channel, band, quad) index, so 10x10 and 19x19 maps still fill their wavefronts
    const long gidx = ((long)blockIdx.x + (long)blockIdx.y * 32768) * 256 + threadIdx.x;
    if (gidx >= (long)a.N * a.cout * per) return;
    const int nc = (int)(gidx / per), idx = (int)(gidx - (long)nc * per);
    const int n = nc / a.cout, oc = nc - n * a.cout;
    const int band = idx / QW, xq = idx - band * QW;
    const int oy0 = band * TH, ox0 = xq * 4, iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;
    const uint8_t* xc = a.x + ((size_t)n * a.C + oc) * a.H * a.W;
    const float* wk = a.wf + (size_t)oc * 9;
    // the NC input bytes of a row as unaligned DWORD loads (2 for stride 1, 3 for stride 2) instead of NC byte gathers: the op
    // is bound by the number of load instructions (a wave's byte gather occupies the address unit for 16 cycles whatever it
    // fetches), not by bytes.  The window starts at column max(ix0, 0) (never in front of the buffer) and may run past the row
    // or the tensor (allocations carry slack); columns / rows outside the image are masked below and enter the chain as 0.0f.
    constexpr int ND = (NC + 1 + 3) / 4;                // dwords that cover NC bytes from a start shifted by at most one
    const int sh = ix0 < 0 ? 1 : 0;                     // left border: the window's first column is outside the image
    unsigned d[NR][ND];
#pragma unroll
    for (int r = 0; r < NR; r++) {
        const int iy = iy0 + r;
        const uint8_t* row = xc + ((unsigned)iy < (unsigned)a.H ? iy : 0) * a.W + (ix0 + sh);
#pragma unroll
        for (int k = 0; k < ND; k++) __builtin_memcpy(&d[r][k], row + 4 * k, 4);
    }
    float w[9];
#pragma unroll
    for (int t = 0; t < 9; t++) w[t] = wk[t];
    float bf = 0.f;
    if (a.bias) {
        bf = (float)a.bias[oc] * a.in_scale;
        bf = bf * a.w_scale;
    }
    unsigned colok = 0;
#pragma unroll
    for (int c = 0; c < NC; c++) colok |= ((unsigned)(ix0 + c) < (unsigned)a.W) ? 1u << c : 0u;
    float xf[NR][NC];
#pragma unroll
    for (int r = 0; r < NR; r++) {
        const bool rok = (unsigned)(iy0 + r) < (unsigned)a.H;
        // at the left border the loaded window starts one column late: shift it up by a byte so that byte c is column c again
        // (byte 0 is then a don't-care: that column is masked)
        unsigned e[ND];
#pragma unroll
        for (int k = 0; k < ND; k++) e[k] = sh ? ((d[r][k] << 8) | (k ? d[r][k - 1] >> 24 : 0u)) : d[r][k];
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const unsigned u = (e[c >> 2] >> (8 * (c & 3))) & 0xffu;
            xf[r][c] = (rok && (colok >> c & 1u)) ? dequant((uint8_t)u, a.in_zp, a.in_scale) : 0.f;
        }
    }
    const float inv = __fdiv_rn(1.0f, a.out_scale);
#pragma unroll
    for (int t = 0; t < TH; t++) {
        const int oy = oy0 + t;
        if (TH > 1 && oy >= a.OH) break;
        float tot[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float total = 0.f;
#pragma unroll
            for (int k = 0; k < 9; k++) total = __builtin_fmaf(xf[t * S + k / 3][j * S + k % 3], w[k], total);
            if (a.bias) total = total + bf;
            if (a.act >= 0) {
                if (total < 0.f && a.act != 1) total = 0.f;
                if (total > 1.f && a.act == 1) total = 1.f;
                if (total > 6.f && a.act == 6) total = 6.f;
                if (total < -1.f && a.act == 1) total = -1.f;
            }
            tot[j] = total;
        }
        // one reciprocal per thread and ONE wave-level test for the four values' rare hand-over to the reference expression
        int q4[4];
        quant_round_sat_u8_w4(tot, a.out_scale, inv, a.out_zp, q4);
        if (a.relu.on) {
#pragma unroll
            for (int j = 0; j < 4; j++) q4[j] = fused_relu((uint8_t)q4[j], a.out_scale, a.out_zp, a.relu);
        }
        uint8_t* yo = a.y + (size_t)n * a.out_img + ((size_t)(a.out_c0 + oc) * a.OH + oy) * a.OW + ox0;
        if (ox0 + 3 < a.OW) {                            // the four bytes as one (unaligned) dword store
            const unsigned pk = (unsigned)q4[0] | ((unsigned)q4[1] << 8) | ((unsigned)q4[2] << 16) | ((unsigned)q4[3] << 24);
            __builtin_memcpy(yo, &pk, 4);
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (ox0 + j < a.OW) yo[j] = (uint8_t)q4[j];
        }
    }
}

// output rows per thread of the depthwise block kernel: taller blocks once the launch has blocks to spare (measured,
// profiles/r04_u8_dw_forms.txt).  TAMD_U8_DW_TH=1|2|4 pins it (experiments; read per launch)
static int u8_dw_th(const U8DirectArgs& a)
{
    if (const char* e = tamd_pin("u8_dw_th")) { const int v = atoi(e); if (v == 1 || v == 2 || (v == 4 && a.SH == 1)) return v; }
    if ((long)a.N * a.cout * a.OH < 65536) return 1;                    // small launches keep the most threads
    if (a.SH == 1) return a.OH >= 64 ? 4 : a.OH >= 32 ? 2 : 1;           // 16 x 32 @ 150^2: 28.1 -> 20.7 us; 16 x 128 @ 75^2: 28.2 -> 20.8; 16 x 256 @ 38^2: 16.3 -> 13.5
    return a.OH >= 32 ? 2 : 1;                                            // 16 x 128 @ 75^2 stride 2: 10.7 -> 9.5 us; 19^2 outputs and below: one row
}

hipError_t launch_conv_u8_direct(const U8DirectArgs& a, hipStream_t s)
{
    if (a.group == a.C && a.cout == a.C && a.KH == 3 && a.KW == 3) {
        const bool quad = a.PH == 1 && a.PW == 1 && a.DH == 1 && a.DW == 1 && a.SH == a.SW && (a.SH == 1 || a.SH == 2) && a.OW >= 4;
        if (quad) {
            const int th = u8_dw_th(a);
            const long blocks = ((long)a.N * a.cout * ((a.OH + th - 1) / th) * ((a.OW + 3) / 4) + 255) / 256;
            dim3 grid((unsigned)(blocks < 32768 ? blocks : 32768), (unsigned)((blocks + 32767) / 32768));
            if (a.SH == 1) {
                if (th == 4) hipLaunchKernelGGL((conv_u8_dw3x4_k<1, 4>), grid, dim3(256), 0, s, a);
                else if (th == 2) hipLaunchKernelGGL((conv_u8_dw3x4_k<1, 2>), grid, dim3(256), 0, s, a);
                else hipLaunchKernelGGL((conv_u8_dw3x4_k<1, 1>), grid, dim3(256), 0, s, a);
            } else {
                if (th == 2) hipLaunchKernelGGL((conv_u8_dw3x4_k<2, 2>), grid, dim3(256), 0, s, a);
                else hipLaunchKernelGGL((conv_u8_dw3x4_k<2, 1>), grid, dim3(256), 0, s, a);
            }
            return hipGetLastError();
        }
        dim3 grid((a.cout * a.OH * a.OW + 255) / 256, a.N);
        hipLaunchKernelGGL(conv_u8_dw3_k, grid, dim3(256), 0, s, a);
        return hipGetLastError();
    }
    dim3 grid((a.OH * a.OW + 255) / 256, a.cout, a.N);
    hipLaunchKernelGGL(conv_u8_direct_k, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

// =================================================================================================================
// fully connected: fc/fc_ref.c:121-207 -- data = (float)bias * bias_scale, then one fused chain over the hidden
// axis, requantise.  Lanes = outputs (coalesced rows of the [hidden][nout_pad] fp32 weights), the dequantised
// input row is staged once in LDS.
// =================================================================================================================
__global__ __launch_bounds__(256) void fc_u8_k(const U8FcArgs a)
{
    extern __shared__ float xrow[];
    const int b = blockIdx.y, o = blockIdx.x * 256 + threadIdx.x;
    for (int j = threadIdx.x; j < a.hidden; j += 256) xrow[j] = dequant(a.x[(size_t)b * a.hidden + j], a.in_zp, a.in_scale);
    __syncthreads();
    if (o >= a.nout) return;
    float data = a.bias ? (float)a.bias[o] * a.bias_scale : 0.f;
    const float* w = a.wf + o;
    int j = 0;
    for (; j + 8 <= a.hidden; j += 8) {
        float wv[8];
#pragma unroll
        for (int u = 0; u < 8; u++) wv[u] = w[(size_t)(j + u) * a.nout_pad];
#pragma unroll
        for (int u = 0; u < 8; u++) data = __builtin_fmaf(xrow[j + u], wv[u], data);
    }
    for (; j < a.hidden; j++) data = __builtin_fmaf(xrow[j], w[(size_t)j * a.nout_pad], data);
    a.y[(size_t)b * a.nout + o] = quant_round_sat_u8(data, a.out_scale, a.out_zp);
}

hipError_t launch_fc_u8(const U8FcArgs& a, hipStream_t s)
{
    dim3 grid((a.nout + 255) / 256, a.batch);
    hipLaunchKernelGGL(fc_u8_k, grid, dim3(256), (size_t)a.hidden * sizeof(float), s, a);
    return hipGetLastError();
}

// =================================================================================================================
// pooling: pooling/pooling_kernel_ref_uint8.c:91-200 -- dequantise, fp32 max / sequential sum (rows, then columns)
// divided by pool_size, round(f/out_s) + out_zp with ONLY the upper clamp (:193-196): a negative value wraps
// through the byte store exactly as the reference's does.
// =================================================================================================================
__global__ __launch_bounds__(256) void pool_u8_k(const U8PoolArgs a)
{
    const int OHW = a.OH * a.OW;
    const int pj = blockIdx.x * 256 + threadIdx.x;
    if (pj >= OHW) return;
    const int ch = blockIdx.y, n = blockIdx.z;
    const int py = pj / a.OW, px = pj - py * a.OW;
    const uint8_t* xc = a.x + ((size_t)n * a.C + ch) * a.H * a.W;
    int hs = py * a.SH - a.PH, he = min(hs + a.KH, a.H + a.PH);
    int ws_ = px * a.SW - a.PW, we = min(ws_ + a.KW, a.W + a.PW);
    int pool_size = 1;
    if (a.caffe_flavor) pool_size = (he - hs) * (we - ws_);
    hs = max(hs, 0); ws_ = max(ws_, 0); he = min(he, a.H); we = min(we, a.W);
    if (!a.caffe_flavor) pool_size = (he - hs) * (we - ws_);
    float f;
    if (a.method == 0) {
        f = ((float)((int)xc[hs * a.W + ws_] - a.in.zp)) * a.in.scale;
        for (int iy = hs; iy < he; iy++)
            for (int ix = ws_; ix < we; ix++) {
                const float v = ((float)((int)xc[iy * a.W + ix] - a.in.zp)) * a.in.scale;
                f = f > v ? f : v;
            }
    } else {
        float sum = 0.f;
        for (int iy = hs; iy < he; iy++)
            for (int ix = ws_; ix < we; ix++) sum = sum + ((float)((int)xc[iy * a.W + ix] - a.in.zp)) * a.in.scale;
        f = __fdiv_rn(sum, (float)pool_size);
    }
    const int od = quant_round_div(f, a.out.scale, a.out.zp);
    a.y[((size_t)n * a.C + ch) * OHW + pj] = (uint8_t)(od > 255 ? 255 : od);
}

hipError_t launch_pool_u8(const U8PoolArgs& a, hipStream_t s)
{
    dim3 grid((a.OH * a.OW + 255) / 256, a.C, a.N);
    hipLaunchKernelGGL(pool_u8_k, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

// =================================================================================================================
// byte maps over an NCHW tensor (grid: x = pixels of one input channel image, y = channel, z = image)
//   MODE 0 relu / leaky  relu/relu_kernel_ref_uint8.c:48-95
//   MODE 1 concat slice  concat/concat_kernel_ref_uint8.c:309-352: roundf(fma((float)(u - zp_in), s_in/s_out, zp_out))
//   MODE 2 upsample      upsample/upsample_ref.c:74-130 (nearest, in = out / scale)
// =================================================================================================================
template <int MODE>
__global__ __launch_bounds__(256) void map_u8_k(const U8MapArgs a, float rescale)
{
    const int OW = a.W * a.scale, OHW = a.H * a.scale * OW;
    const int pj = blockIdx.x * 256 + threadIdx.x;
    if (pj >= OHW) return;
    const int ch = blockIdx.y, n = blockIdx.z;
    const uint8_t* xc = a.x + ((size_t)n * a.C + ch) * a.H * a.W;
    uint8_t* yo = a.y + (size_t)n * a.out_img + (size_t)(a.out_c0 + ch) * OHW + pj;
    if (MODE == 0) {
        float f = dequant(xc[pj], (float)a.in.zp, a.in.scale);
        if (f < 0.f) f = (a.slope == 0.f) ? 0.f : f * a.slope;
        *yo = quant_round_in(f, a.out);
    } else if (MODE == 1) {
        float r = roundf(__builtin_fmaf((float)((int)xc[pj] - a.in.zp), rescale, (float)a.out.zp));
        *yo = sat_u8((int)fminf(fmaxf(r, -65536.f), 65536.f));
    } else {
        const int oy = pj / OW, ox = pj - oy * OW;
        *yo = quant_round_in(dequant(xc[(oy / a.scale) * a.W + ox / a.scale], (float)a.in.zp, a.in.scale), a.out);
    }
}

template <int MODE>
static hipError_t launch_map(const U8MapArgs& a, hipStream_t s)
{
    dim3 grid((a.H * a.scale * a.W * a.scale + 255) / 256, a.C, a.N);
    hipLaunchKernelGGL(map_u8_k<MODE>, grid, dim3(256), 0, s, a, a.in.scale / a.out.scale);
    return hipGetLastError();
}
hipError_t launch_relu_u8(const U8MapArgs& a, hipStream_t s) { return launch_map<0>(a, s); }
hipError_t launch_requant_copy_u8(const U8MapArgs& a, hipStream_t s) { return launch_map<1>(a, s); }
hipError_t launch_upsample_u8(const U8MapArgs& a, hipStream_t s) { return launch_map<2>(a, s); }

// SSD head plumbing in one pass: permute/permute_ref.c:203-296 (order 0,2,3,1: out[n][p][c] = in[n][c][p], bytes
// unchanged), flatten/flatten_ref.c:53-77 (copy) and concat_kernel_ref_uint8.c:127-160 (axis 1 of [n][len] tensors,
// roundf(fma(u - zp_in, s_in / s_out, zp_out)) per element) collapse into one indexed copy per concat input.
__global__ __launch_bounds__(256) void flatcat_u8_k(const U8CatArgs a, float rescale)
{
    const int j = blockIdx.x * 256 + threadIdx.x, n = blockIdx.y;
    if (j >= a.in_img) return;
    const int src = a.perm_c ? (j % a.perm_c) * a.perm_p + j / a.perm_c : j;
    const uint8_t u = a.x[(size_t)n * a.in_img + src];
    uint8_t q = u;
    if (!a.identity) {
        float r = roundf(__builtin_fmaf((float)((int)u - a.in.zp), rescale, (float)a.out.zp));
        q = sat_u8((int)fminf(fmaxf(r, -65536.f), 65536.f));
    }
    a.y[(size_t)n * a.out_img + a.out_off + j] = q;
}

// the same for all inputs of a concat node at once: six 5 us launches of a few kilobytes each become one
__global__ __launch_bounds__(256) void flatcat_multi_u8_k(const U8CatMulti m)
{
    const U8CatArgs& a = m.src[blockIdx.z];
    const int j = blockIdx.x * 256 + threadIdx.x, n = blockIdx.y;
    if (j >= a.in_img) return;
    const int src = a.perm_c ? (j % a.perm_c) * a.perm_p + j / a.perm_c : j;
    const uint8_t u = a.x[(size_t)n * a.in_img + src];
    uint8_t q = u;
    if (!a.identity) {
        float r = roundf(__builtin_fmaf((float)((int)u - a.in.zp), m.rescale[blockIdx.z], (float)a.out.zp));
        q = sat_u8((int)fminf(fmaxf(r, -65536.f), 65536.f));
    }
    a.y[(size_t)n * a.out_img + a.out_off + j] = q;
}

hipError_t launch_flatcat_multi_u8(const U8CatMulti& m, hipStream_t s)
{
    int widest = 0;
    for (int i = 0; i < m.count; i++) widest = std::max(widest, m.src[i].in_img);
    hipLaunchKernelGGL(flatcat_multi_u8_k, dim3((widest + 255) / 256, m.src[0].N, m.count), dim3(256), 0, s, m);
    return hipGetLastError();
}

hipError_t launch_flatcat_u8(const U8CatArgs& a, hipStream_t s)
{
    hipLaunchKernelGGL(flatcat_u8_k, dim3((a.in_img + 255) / 256, a.N), dim3(256), 0, s, a, a.in.scale / a.out.scale);
    return hipGetLastError();
}

// eltwise (same-shape operands): eltwise/eltwise_ref.c:311-585
__global__ __launch_bounds__(256) void eltwise_u8_k(const U8EltArgs a)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.count) return;
    const float fa = (float)((int)a.a[i] - a.qa.zp) * a.qa.scale, fb = (float)((int)a.b[i] - a.qb.zp) * a.qb.scale;
    float r;
    switch (a.type) {
    case 0: r = fa * fb; break;
    case 2: r = fa + fb; break;
    case 4: r = fa - fb; break;
    default: r = fa > fb ? fa : fb; break;
    }
    a.y[i] = quant_round_sat_u8(r, a.out.scale, a.out.zp);
}

hipError_t launch_eltwise_u8(const U8EltArgs& a, hipStream_t s)
{
    hipLaunchKernelGGL(eltwise_u8_k, dim3((unsigned)((a.count + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ---- uint8 softmax: softmax/softmax_kernel_ref_uint8.c:40-119 over softmax_kernel_ref.h:35-85 -------------------------------
// dequantise; per (outer, inner) position: max over the axis; o = (float)exp((double)(f - max)) -- the reference calls C `exp`
// on a float argument, i.e. the DOUBLE routine, and rounds the result to float on the store (:67); sum in fp32 in axis order
// (:68); o / sum; u = (int)(round(o / out_scale) + out_zp), clamp [0, 255] (:103-113).  exp runs in fp64 here too (ocml, <= 1
// ulp): its float rounding differs from glibc's correctly rounded exp only when the true value sits within a double ulp of a
// float rounding midpoint (~2^-29 per call), and that float ulp would then have to straddle a uint8 rounding boundary.
// One thread per position: the axis is short (21 classes in MobileNet-SSD), exp is evaluated in both passes (deterministic).
__global__ __launch_bounds__(256) void softmax_u8_kernel(U8SoftmaxArgs a)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)a.outer * a.inner) return;
    const long o = i / a.inner;
    const int l = (int)(i - o * a.inner);
    const uint8_t* x = a.x + (size_t)o * a.on * a.inner + l;
    uint8_t* y = a.y + (size_t)o * a.on * a.inner + l;
    const float zp = (float)a.in.zp;
    auto deq = [&](int j) { return __fmul_rn(__fsub_rn((float)x[(size_t)j * a.inner], zp), a.in.scale); };
    float mx = deq(0);
    for (int j = 1; j < a.on; j++) { const float f = deq(j); if (mx < f) mx = f; }
    float sum = 0.f;
    for (int j = 0; j < a.on; j++) sum = __fadd_rn(sum, (float)exp((double)__fsub_rn(deq(j), mx)));
    for (int j = 0; j < a.on; j++) {
        const float e = (float)exp((double)__fsub_rn(deq(j), mx));
        const float v = __fdiv_rn(e, sum);
        int u = (int)(round((double)__fdiv_rn(v, a.out.scale)) + (double)a.out.zp);
        u = u < 0 ? 0 : (u > 255 ? 255 : u);
        y[(size_t)j * a.inner] = (uint8_t)u;
    }
}

hipError_t launch_softmax_u8(const U8SoftmaxArgs& a, hipStream_t s)
{
    const long total = (long)a.outer * a.inner;
    hipLaunchKernelGGL(softmax_u8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace tamd
