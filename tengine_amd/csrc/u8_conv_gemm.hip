// uint8 convolution, the staging GEMM (conv_u8_mfma: first layers and what the patch kernel does not take).
//
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
// (One file, u8_kernels.hip, until round 5; split by kernel family in round 6: u8_conv_gemm.hip, u8_conv_patch.hip, u8_conv_small.hip,
//  u8_kernels.hip = depthwise / grouped, FC, pooling, the byte maps, concat, eltwise, softmax.)
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


}  // namespace tamd
