// fp32 GEMM-convolution on the matrix cores with the reference's summation order -- asynchronous version.
//
// Arithmetic contract (same as u8_conv_gemm.hip, which holds the register-staged version of this kernel):
// conv/x86/conv_kernel_x86.c:126-185 (im2col, k = (c,ky,kx), 0.0f outside the image), :322-960 sgemm_fp:
//   pixel j <  (OH*OW)&~7 : one fused chain s = fma(x[k], w[k], s), k ascending        -> "main" blocks
//   pixel j >= (OH*OW)&~7 : four fused chains over k = r (mod 4), k < K&~3, combined ((0+(s0+s1))+(s2+s3)) for
//                           rows in an 8-/4-row block, ((s0+s1)+s2)+s3 for the last cout%4 rows, then the fused
//                           chain over the K%4 remainder                               -> "tail" blocks
// v_mfma_f32_16x16x4f32 IS four sequential IEEE fmas in ascending k (profiles/r01_mfma_f32_is_sequential_fma_chain.txt),
// so one accumulator tile fed in ascending k reproduces the chain bit for bit.
//
// What is different here is how the matrix pipe is fed:
//   * both operands are fp32 in memory -- the weights packed once at prerun (uint8 models: dequantised exactly as
//     conv_kernel_x86.c:68-80 does), the activations dequantised by a byte->float pass (dequant_u8_f32_k; fp32
//     models need none) -- so a K stage is a pure copy and travels global -> LDS with LDS-DMA
//     (global_load_lds_dword): no VGPR round trip, no ds_write, and above all NO register results for the
//     compiler's s_waitcnt insertion to serialise: STAGES-1 stages stay in flight behind counted vmcnt waits
//     (the register ring of u8_conv_gemm.hip is drained by hipcc at every loop back-edge);
//   * LDS is k-major: row k holds the 16/32/64 channels (pixels) of the tile, so a DMA instruction's 64 lanes are
//     64 consecutive channels of the packed weights / 64 consecutive pixels of the image (coalesced), and an MFMA
//     operand read is 16 consecutive dwords per k; row groups are spaced 80 dwords so the four k rows of an MFMA
//     hit four different bank quarters (conflict-free ds_read_b32);
//   * out-of-image taps and padded k rows copy from a zero page instead of being predicated.
#include <hip/hip_runtime.h>
#include "env.h"

#include <cstdlib>

#include "kernels.h"

namespace tamd {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) unsigned* cunsp;      // constant address space => scalar loads

#define TAMD_GLDS4(gptr, lptr)                                                                       \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),         \
                                     (__attribute__((address_space(3))) void*)(lptr), 4, 0, 0)

// LDS geometry of one operand region of width WD (16, 32 or 64 dwords per k row), 32 k rows per stage
template <int WD>
struct Region {
    static constexpr int G = 64 / WD;                       // k rows per DMA instruction (64 lanes)
    static constexpr int GS = WD == 16 ? 64 : 80;           // dwords between row groups
    static constexpr int NI = 32 / G;                       // DMA instructions per stage
    static constexpr int DW = NI * GS;                      // dwords per stage
    __device__ static constexpr int row(int r) { return (r / G) * GS + (r % G) * WD; }
};

__device__ __forceinline__ int quant_round_div_i(float s, float out_scale, int zp)
{
    float r = roundf(__fdiv_rn(s, out_scale));
    r = fminf(fmaxf(r, -65536.f), 65536.f);
    return (int)r + zp;
}

template <int WM, int WN, int TM, int TN, int STAGES, bool TAIL>
__device__ __forceinline__ void conv_f32_body(const F32ConvArgs& a, float* smem, int n, int jbase, int jlimit, int co0)
{
    constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
    typedef Region<BM> RW;
    typedef Region<BN> RX;
    constexpr int STAGE_DW = RW::DW + RX::DW;
    constexpr int NIW = RW::NI / 4, NIX = RX::NI / 4, NI = NIW + NIX;      // DMA instructions per wave per stage
    static_assert(RW::NI % 4 == 0 && RX::NI % 4 == 0, "instructions split over 4 waves");
    static_assert((STAGES - 2) * NI <= 63 && STAGES >= 3 && STAGES <= 6, "vmcnt is a 6-bit counter");
    constexpr int NCH = TAIL ? 4 : 1, LA = STAGES - 1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM, l15 = lane & 15, kq = lane >> 4;
    const int K4 = a.K & ~3, nstage = a.Kpad / 32;
    const cunsp lut = (cunsp)(uintptr_t)a.klut;

    // ---- loader state.  X: this lane always copies pixel (lane % BN) of k row (group*G + lane / BN) -------------
    const int xp = lane % BN, xq = lane / BN;
    const int xj = jbase + xp;
    const bool xvalid = xj < jlimit;
    const int xoy = xvalid ? xj / a.OW : 0, xox = xvalid ? xj - xoy * a.OW : 0;
    const int iy0 = xoy * a.SH - a.PH, ix0 = xox * a.SW - a.PW;
    const float* xin = a.x + (size_t)n * a.C * a.H * a.W + (iy0 * a.W + ix0);      // may point before the image: only
    const float* wsrc = a.w + ((size_t)(co0 / BM) * nstage) * (32 * BM) + lane;    // dereferenced for in-image taps

    auto issue = [&](int s) {
        float* base = smem + (s % STAGES) * STAGE_DW;
#pragma unroll
        for (int i = 0; i < NIW; i++) {
            const int g = i * 4 + wave;                                    // row group of the weight region
            TAMD_GLDS4(wsrc + ((size_t)s * RW::NI + g) * 64, base + g * RW::GS);
        }
#pragma unroll
        for (int i = 0; i < NIX; i++) {
            const int g = i * 4 + wave;
            const int kb = s * 32 + g * RX::G;                             // first k row of this group (wave uniform)
            unsigned t = lut[kb];
            if (RX::G >= 2) { const unsigned t1 = lut[kb + 1]; t = xq == 1 ? t1 : t; }
            if (RX::G == 4) { const unsigned t2 = lut[kb + 2], t3 = lut[kb + 3]; t = xq == 2 ? t2 : (xq == 3 ? t3 : t); }
            const int iy = iy0 + (int)(t >> 28), ix = ix0 + (int)((t >> 24) & 15);
            const bool ok = xvalid & ((unsigned)iy < (unsigned)a.H) & ((unsigned)ix < (unsigned)a.W);
            const float* src = ok ? xin + (t & 0xffffffu) : a.zeros;
            TAMD_GLDS4(src, base + RW::DW + g * RX::GS);
        }
    };

    v4f acc[NCH][TM][TN];
#pragma unroll
    for (int r = 0; r < NCH; r++)
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++) acc[r][i][j] = v4f{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int p = 0; p < LA; p++)
        if (p < nstage) issue(p);
    for (int s = 0; s < nstage; s++) {
        // my own copies of stage s have landed (the younger stages stay in flight) ...
        const int rem = nstage - s;
        if (rem >= LA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((LA - 1) * NI) : "memory");
        else if (LA >= 3 && rem == LA - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((LA >= 3 ? LA - 2 : 0) * NI) : "memory");
        else if (LA >= 4 && rem == LA - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((LA >= 4 ? LA - 3 : 0) * NI) : "memory");
        else if (LA >= 5 && rem == LA - 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((LA >= 5 ? LA - 4 : 0) * NI) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // ... and so have everyone else's; every wave is also done reading the slot stage s+LA overwrites
        __builtin_amdgcn_s_barrier();
        if (s + LA < nstage) issue(s + LA);
        const float* wsb = smem + (s % STAGES) * STAGE_DW;
        const float* xsb = wsb + RW::DW;
        if constexpr (!TAIL) {
            // MFMA i8 consumes k rows 4*i8 .. 4*i8+3 (lane group kq supplies row 4*i8+kq): ascending k
            const float* ap = wsb + RW::row(kq) + wm * TM * 16 + l15;
            const float* bp = xsb + RX::row(kq) + wn * TN * 16 + l15;
            // all operand reads of the stage are issued up front (in-order LDS returns, counted lgkmcnt waits):
            // the first MFMAs start as soon as their operands land while the rest are still in flight
            float af[8][TM], bf[8][TN];
#pragma unroll
            for (int i8 = 0; i8 < 8; i8++) {
#pragma unroll
                for (int i = 0; i < TM; i++) af[i8][i] = ap[i8 * RW::row(4) + i * 16];
#pragma unroll
                for (int j = 0; j < TN; j++) bf[i8][j] = bp[i8 * RX::row(4) + j * 16];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i8 = 0; i8 < 8; i8++)
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int j = 0; j < TN; j++)
                        acc[0][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i8][i], bf[i8][j], acc[0][i][j], 0, 0, 0);
        } else {
            // chain r consumes k = r (mod 4) only: MFMA j2 of chain r takes rows r + 16*j2 + 4*kq (ascending inside
            // the chain); the K%4 remainder rows (k >= K4) are skipped here and chained after the combine
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int j2 = 0; j2 < 2; j2++) {
                    const int row = r + 16 * j2 + 4 * kq;
                    const bool live = s * 32 + row < K4;
#pragma unroll
                    for (int i = 0; i < TM; i++)
#pragma unroll
                        for (int j = 0; j < TN; j++) {
                            // a tail tile holds at most 7 pixels: 16-pixel columns without any are skipped (wave uniform)
                            if (jbase + (wn * TN + j) * 16 >= jlimit) continue;
                            float av = wsb[RW::row(row) + (wm * TM + i) * 16 + l15];
                            const float bv = xsb[RX::row(row) + (wn * TN + j) * 16 + l15];
                            av = live ? av : 0.f;
                            acc[r][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[r][i][j], 0, 0, 0);
                        }
                }
        }
    }

    // ---- epilogue: D[row = 4*kq + e][col = l15] of each 16x16 tile ----------------------------------------------
    const int OHW = a.OH * a.OW;
#pragma unroll
    for (int j = 0; j < TN; j++) {
        const int pj = jbase + (wn * TN + j) * 16 + l15;
        if (pj >= jlimit) continue;
        const int oy = pj / a.OW, ox = pj - oy * a.OW;
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int co = co0 + (wm * TM + i) * 16 + 4 * kq + e;
                if (co >= a.cout) continue;
                float sum;
                if constexpr (TAIL) {
                    const float s0 = acc[0][i][j][e], s1 = acc[1][i][j][e], s2 = acc[2][i][j][e], s3 = acc[3][i][j][e];
                    if (co < a.m_blocked) sum = (0.f + (s0 + s1)) + (s2 + s3);
                    else sum = ((s0 + s1) + s2) + s3;
                    const float* xi = a.x + (size_t)n * a.C * a.H * a.W + ((oy * a.SH - a.PH) * a.W + ox * a.SW - a.PW);
                    for (int k = K4; k < a.K; k++) {
                        const unsigned t = a.klut[k];
                        const int iy = oy * a.SH - a.PH + (int)(t >> 28), ix = ox * a.SW - a.PW + (int)((t >> 24) & 15);
                        float v = 0.f;
                        if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) v = xi[t & 0xffffffu];
                        const int r = k & 31;
                        const float wv = a.w[(((size_t)(co0 / BM) * nstage + (k >> 5)) * RW::NI + r / RW::G) * 64 + (r % RW::G) * BM + (co - co0)];
                        sum = __builtin_fmaf(wv, v, sum);
                    }
                } else
                    sum = acc[0][i][j][e];
                if (a.out_f32) {
                    // fp32 model: conv_kernel_x86.c:1632-1701 (sgemm_fp32): + bias, relu / relu6
                    if (a.bias_f32) sum = sum + a.bias_f32[co];
                    if (a.act == 0) sum = sum < 0.f ? 0.f : sum;
                    if (a.act > 0) { sum = sum < 0.f ? 0.f : sum; sum = sum > 6.f ? 6.f : sum; }
                    a.out_f32[(size_t)n * a.out_img + (size_t)(a.out_c0 + co) * OHW + pj] = sum;
                } else {
                    // uint8 model: conv_kernel_x86.c:1703-1794
                    if (a.bias) sum = sum + (float)a.bias[co] * a.bias_scale;      // rounded product, then add (see u8_kernels.hip)
                    if (a.act == 0) sum = sum < 0.f ? 0.f : sum;
                    if (a.act > 0) { sum = sum < 0.f ? 0.f : sum; sum = sum > 6.f ? 6.f : sum; }
                    const int q = quant_round_div_i(sum, a.out_scale, a.out_zp);
                    a.y[(size_t)n * a.out_img + (size_t)(a.out_c0 + co) * OHW + pj] = (uint8_t)min(max(q, 0), 255);
                }
            }
    }
}

template <int WM, int WN, int TM, int TN, int STAGES>
__global__ __launch_bounds__(256) void conv_f32_mfma_k(const F32ConvArgs a)
{
    constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
    static_assert(WM * WN == 4, "4 waves");
    // ONE LDS object, nothing else read from LDS or global memory by vector loads inside the K loop: hipcc drains
    // vmcnt in front of any such read it cannot disambiguate from the in-flight LDS-DMA
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int OHW = a.OH * a.OW, N8 = a.tail_split ? (OHW & ~7) : OHW;
    const int tiles = (N8 + BN - 1) / BN, tpi = tiles + (OHW != N8);
    const int PT = tpi * a.N, CT = (a.cout + BM - 1) / BM;
    // block -> (channel tile, pixel tile).  Workgroups are dealt round-robin to the 8 XCDs (block b -> XCD b%8) and
    // every XCD has its own 4 MB L2: with >= 8 channel tiles, XCD x only ever touches channel tiles = x (mod 8), so
    // its slice of the packed weights stays L2-resident while it sweeps the pixel tiles; with fewer channel tiles the
    // weights are small and the plain order (channel tile fastest) is used.
    int ct, pt;
    if (CT >= 8) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        ct = (idx / PT) * 8 + xcd;
        pt = idx % PT;
        if (ct >= CT) return;
    } else {
        ct = blockIdx.x % CT;
        pt = blockIdx.x / CT;
    }
    const int n = pt / tpi, tile = pt - n * tpi, co0 = ct * BM;
    if (tile < tiles) conv_f32_body<WM, WN, TM, TN, STAGES, false>(a, smem, n, tile * BN, N8, co0);
    else conv_f32_body<WM, WN, TM, TN, STAGES, true>(a, smem, n, N8, OHW, co0);
}

// ---- configurations ---------------------------------------------------------------------------------------------
static const struct { int bm, bn, stages; const char* name; } F32_CFGS[] = {
    {16, 64, 6, "conv_f32_mfma_16x64"}, {32, 32, 6, "conv_f32_mfma_32x32"}, {64, 64, 4, "conv_f32_mfma_64x64"},
    {32, 64, 5, "conv_f32_mfma_32x64"}, {64, 16, 6, "conv_f32_mfma_64x16"}};

static int region_dw(int wd) { return (32 / (64 / wd)) * (wd == 16 ? 64 : 80); }

int conv_f32_mfma_pick(const F32ConvArgs& a)
{
    static const char* e = exp_env("TAMD_F32_CFG");
    if (e && *e) return atoi(e) % 5;
    const int OHW = a.OH * a.OW, N8 = a.tail_split ? (OHW & ~7) : OHW;
    auto blocks = [&](int bm, int bn) { return (long)((N8 + bn - 1) / bn + (OHW != N8 ? 1 : 0)) * ((a.cout + bm - 1) / bm) * a.N; };
    if (OHW <= 16 && a.cout > 32) return 4;            // 1x1 .. 4x4 maps: 64 channels x 16 pixels
    if (a.cout <= 16) return 0;
    if (a.cout <= 32) return 3;
    if (blocks(64, 64) >= 512) return 2;
    if (blocks(32, 64) >= 384) return 3;
    return 1;
}
int conv_f32_mfma_bm(int cfg) { return F32_CFGS[cfg].bm; }
size_t conv_f32_mfma_lds(int cfg) { return (size_t)F32_CFGS[cfg].stages * (region_dw(F32_CFGS[cfg].bm) + region_dw(F32_CFGS[cfg].bn)) * 4; }
const char* conv_f32_mfma_kernel_name(const F32ConvArgs& a) { return F32_CFGS[a.cfg].name; }

hipError_t launch_conv_f32_mfma(const F32ConvArgs& a, hipStream_t s)
{
    const int OHW = a.OH * a.OW, N8 = a.tail_split ? (OHW & ~7) : OHW, ntail = OHW - N8;
    const int bm = F32_CFGS[a.cfg].bm, bn = F32_CFGS[a.cfg].bn;
    const int PT = ((N8 + bn - 1) / bn + (ntail ? 1 : 0)) * a.N, CT = (a.cout + bm - 1) / bm;
    const dim3 grid(CT >= 8 ? 8 * ((CT + 7) / 8) * PT : CT * PT, 1, 1);
    const size_t lds = conv_f32_mfma_lds(a.cfg);
    auto go = [&](auto kern) {
        static bool attr_set = false;           // one flag per instantiation (the lambda body is instantiated per kernel)
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr_set = true;
        }
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
        return hipGetLastError();
    };
    switch (a.cfg) {
    case 0: return go(conv_f32_mfma_k<1, 4, 1, 1, 6>);
    case 2: return go(conv_f32_mfma_k<2, 2, 2, 2, 4>);
    case 3: return go(conv_f32_mfma_k<2, 2, 1, 2, 5>);
    case 4: return go(conv_f32_mfma_k<4, 1, 1, 1, 6>);   // 64 ch x 16 px
    default: return go(conv_f32_mfma_k<2, 2, 1, 1, 6>);
    }
}

// byte -> float pass feeding the DMA kernel: x_fp32 = ((float)u - (float)zp) * scale (conv_kernel_x86.c:166)
__global__ __launch_bounds__(256) void dequant_u8_f32_k(const uint8_t* __restrict__ x, float* __restrict__ y, size_t n4, size_t n,
                                                        float zp, float scale)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) {
        const unsigned u = reinterpret_cast<const unsigned*>(x)[i];
        float4 v;
        v.x = ((float)(u & 255u) - zp) * scale;
        v.y = ((float)((u >> 8) & 255u) - zp) * scale;
        v.z = ((float)((u >> 16) & 255u) - zp) * scale;
        v.w = ((float)(u >> 24) - zp) * scale;
        reinterpret_cast<float4*>(y)[i] = v;
    }
    if (i == 0)
        for (size_t j = n4 * 4; j < n; j++) y[j] = ((float)x[j] - zp) * scale;
}

hipError_t launch_dequant_u8_f32(const uint8_t* x, float* y, size_t n, float zp, float scale, hipStream_t s)
{
    const size_t n4 = n / 4;
    hipLaunchKernelGGL(dequant_u8_f32_k, dim3((unsigned)((n4 + 255) / 256 + (n4 == 0))), dim3(256), 0, s, x, y, n4, n, zp, scale);
    return hipGetLastError();
}

}  // namespace tamd
