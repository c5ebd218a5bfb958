// uint8 (per-tensor asymmetric) kernels.
//
// The reference does not compute uint8 in integers: it dequantises both operands to fp32, runs its fp32 code and
// requantises the result.  The bytes it produces therefore depend on the exact fp32
// operation sequence -- for convolution on the summation ORDER of its 8x8-tiled AVX sgemm.  To be byte-identical
// this file performs the same IEEE binary32 operations in the same order per output element:
//   * every `a*b + c` the reference's compiler contracts (-O3 -mfma, default -ffp-contract=fast) is one
//     __builtin_fmaf here; nothing else is fused (this TU is built -ffp-contract=off);
//   * divisions are correctly rounded (__fdiv_rn), round() is round-half-away (roundf).
// A sequential fp32 chain per output element cannot use MFMA (its internal accumulation order is not the
// reference's) nor split K; the parallelism is across output elements: lanes = pixels x channels, operands staged
// through LDS, register tiles of independent chains per thread.  Bound: fp32 vector FMA issue, not HBM.
// Activations stay in the reference's dense NCHW order (lanes along pixels read consecutive bytes).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "kernels.h"

namespace tamd {

// (int)(round(s / out_scale) + zp), clamp [0,255] -- conv_kernel_x86.c:1783-1788, conv_kernel_ref_uint8.c:177-182,
// fc_ref.c:196-202, eltwise_ref.c:571-578
__device__ __forceinline__ int quant_round_div(float s, float out_scale, int zp)
{
    float r = roundf(__fdiv_rn(s, out_scale));
    r = fminf(fmaxf(r, -65536.f), 65536.f);
    return (int)r + zp;
}
__device__ __forceinline__ uint8_t sat_u8(int v) { return (uint8_t)min(max(v, 0), 255); }

// round(f / out_scale + zp), clamp -- relu_kernel_ref_uint8.c:83-89, upsample_ref.c:118-125 (zero point INSIDE the round)
__device__ __forceinline__ uint8_t quant_round_in(float f, U8Q q)
{
    float r = roundf(__fdiv_rn(f, q.scale) + (float)q.zp);
    r = fminf(fmaxf(r, -65536.f), 65536.f);
    return sat_u8((int)r);
}
__device__ __forceinline__ float dequant(uint8_t u, float zp, float scale) { return ((float)u - zp) * scale; }

// =================================================================================================================
// group == 1 convolution: conv/x86/conv_kernel_x86.c:68-80 (weights -> fp32), :126-185 (im2col_uint8, k = (c,ky,kx),
// 0.0f at out-of-image taps), :322-960 sgemm_fp, :1703-1794 bias / activation / requantise.
// Per image the GEMM is [cout] x [OH*OW] x [K]; an element's summation order depends on its place in the
// reference's tiling (oracle/tg_oracle.c sgemm_fp_element restates it):
//   pixel j <  (OH*OW)&~7 : one fused chain over k = 0..K-1                       -> "main" blocks
//   pixel j >= (OH*OW)&~7 : four fused chains over k = r (mod 4), k < K&~3, combined
//                           ((0+(s0+s1))+(s2+s3)) for rows in an 8-/4-row block, ((s0+s1)+s2)+s3 for the last
//                           cout%4 rows, then the fused chain over the K%4 tail    -> "tail" blocks (same launch)
// Block = 256 threads = TXN x TYN, thread tile TP pixels x TC channels; K staged 16 at a time through LDS
// (double buffered, register prefetch).  Padded k rows carry w = 0 and an out-of-image lut entry: fma(0,0,s) == s.
// =================================================================================================================
template <int TXN, int TYN, int TP, int TC, bool TAIL>
__device__ __forceinline__ void conv_u8_body(const U8ConvArgs& a, float (&xs)[2][16][TXN * TP],
                                             float (&ws)[2][16][TYN * TC], int n, int jbase, int jlimit, int co0)
{
    constexpr int TPX = TXN * TP, TCX = TYN * TC, KC = 16, NT = 256;
    constexpr int XE = KC * TPX / NT, WE = KC * TCX / NT, NCH = TAIL ? 4 : 1;
    const int tid = threadIdx.x, tx = tid % TXN, ty = tid / TXN;
    const int K4 = a.K & ~3;

    // staging role of this thread: one pixel column of the x tile, XE rows of k
    const int sp = tid % TPX, klb = tid / TPX;
    const int sj = jbase + sp;
    const bool svalid = sj < jlimit;
    const int soy = svalid ? sj / a.OW : 0, sox = svalid ? sj - soy * a.OW : 0;
    const int iy0 = soy * a.SH - a.PH, ix0 = sox * a.SW - a.PW;
    const uint8_t* xin = a.x + (size_t)n * a.C * a.H * a.W;
    const int pbase = iy0 * a.W + ix0;

    float xr[XE], wr[WE];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < XE; i++) {
            const int2 e = a.klut[k0 + klb + i * (NT / TPX)];
            const int iy = iy0 + (e.y >> 16), ix = ix0 + (e.y & 0xffff);
            const bool ok = svalid && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            float v = 0.f;
            if (ok) v = dequant(xin[pbase + e.x], a.in_zp, a.in_scale);
            xr[i] = v;
        }
#pragma unroll
        for (int i = 0; i < WE; i++) {
            const int e = tid + NT * i, kl = e / TCX, c = e % TCX, k = k0 + kl;
            float w = a.wf[(size_t)k * a.cout_pad + co0 + c];
            if (TAIL && k >= K4) w = 0.f;      // the K%4 remainder is chained after the combine
            wr[i] = w;
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < XE; i++) xs[buf][klb + i * (NT / TPX)][sp] = xr[i];
#pragma unroll
        for (int i = 0; i < WE; i++) {
            const int e = tid + NT * i;
            ws[buf][e / TCX][e % TCX] = wr[i];
        }
    };

    float acc[NCH][TP][TC];
#pragma unroll
    for (int r = 0; r < NCH; r++)
#pragma unroll
        for (int i = 0; i < TP; i++)
#pragma unroll
            for (int j = 0; j < TC; j++) acc[r][i][j] = 0.f;

    const int nchunk = a.Kpad / KC;
    gload(0);
    sstore(0);
    __syncthreads();
    for (int ch = 0; ch < nchunk; ch++) {
        const int cur = ch & 1;
        if (ch + 1 < nchunk) gload((ch + 1) * KC);
#pragma unroll
        for (int kl = 0; kl < KC; kl++) {
            float xv[TP], wv[TC];
#pragma unroll
            for (int i = 0; i < TP; i++) xv[i] = xs[cur][kl][tx * TP + i];
#pragma unroll
            for (int j = 0; j < TC; j++) wv[j] = ws[cur][kl][ty * TC + j];
#pragma unroll
            for (int i = 0; i < TP; i++)
#pragma unroll
                for (int j = 0; j < TC; j++) {
                    float& s = acc[TAIL ? (kl & 3) : 0][i][j];
                    s = __builtin_fmaf(xv[i], wv[j], s);
                }
        }
        if (ch + 1 < nchunk) sstore(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue ------------------------------------------------------------------------------------------
    const int OHW = a.OH * a.OW;
#pragma unroll
    for (int j = 0; j < TC; j++) {
        const int co = co0 + ty * TC + j;
        if (co >= a.cout) continue;
        float bf = 0.f;
        if (a.bias) bf = (float)a.bias[co];
#pragma unroll
        for (int i = 0; i < TP; i++) {
            const int pj = jbase + tx * TP + i;
            if (pj >= jlimit) continue;
            float s;
            if constexpr (TAIL) {
                const float s0 = acc[0][i][j], s1 = acc[1][i][j], s2 = acc[2][i][j], s3 = acc[3][i][j];
                if (co < a.m_blocked) s = (0.f + (s0 + s1)) + (s2 + s3);
                else s = ((s0 + s1) + s2) + s3;
                const int oy = pj / a.OW, ox = pj - oy * a.OW;
                for (int k = K4; k < a.K; k++) {
                    const int2 e = a.klut[k];
                    const int iy = oy * a.SH - a.PH + (e.y >> 16), ix = ox * a.SW - a.PW + (e.y & 0xffff);
                    float v = 0.f;
                    if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
                        v = dequant(xin[(oy * a.SH - a.PH) * a.W + ox * a.SW - a.PW + e.x], a.in_zp, a.in_scale);
                    s = __builtin_fmaf(a.wf[(size_t)k * a.cout_pad + co], v, s);
                }
            } else
                s = acc[0][i][j];
            if (a.bias) s = __builtin_fmaf(bf, a.bias_scale, s);
            if (a.act == 0) s = s < 0.f ? 0.f : s;
            if (a.act > 0) { s = s < 0.f ? 0.f : s; s = s > 6.f ? 6.f : s; }
            a.y[(size_t)n * a.out_img + (size_t)(a.out_c0 + co) * OHW + pj] = sat_u8(quant_round_div(s, a.out_scale, a.out_zp));
        }
    }
}

template <int TXN, int TYN, int TP, int TC>
__global__ __launch_bounds__(256) void conv_u8_gemm_k(const U8ConvArgs a)
{
    constexpr int TPX = TXN * TP, TCX = TYN * TC;
    static_assert(TXN * TYN == 256, "256 threads");
    __shared__ __attribute__((aligned(16))) float xs[2][16][TPX];
    __shared__ __attribute__((aligned(16))) float ws[2][16][TCX];
    const int OHW = a.OH * a.OW, N8 = OHW & ~7;
    const int tiles = (N8 + TPX - 1) / TPX;
    const int n = blockIdx.z, co0 = blockIdx.y * TCX;
    if ((int)blockIdx.x < tiles) conv_u8_body<TXN, TYN, TP, TC, false>(a, xs, ws, n, blockIdx.x * TPX, N8, co0);
    else conv_u8_body<TXN, TYN, TP, TC, true>(a, xs, ws, n, N8, OHW, co0);
}

// tile choice: enough blocks to occupy 256 CUs first, then the largest register tile
static int u8_cfg(const U8ConvArgs& a)
{
    static const char* e = getenv("TAMD_U8_CFG");
    if (e && *e) return atoi(e);
    const int N8 = (a.OH * a.OW) & ~7;
    auto blocks = [&](int tpx, int tcx) { return (long)((N8 + tpx - 1) / tpx) * ((a.cout + tcx - 1) / tcx) * a.N; };
    if (a.cout <= 16 && blocks(256, 16) >= 256) return 1;
    if (a.cout <= 32 && blocks(128, 32) >= 256) return 2;
    if (blocks(64, 64) >= 384) return 0;
    if (blocks(32, 32) >= 256) return 3;
    return 4;
}

const char* conv_u8_gemm_kernel_name(const U8ConvArgs& a)
{
    static const char* names[] = {"conv_u8_gemm_64x64", "conv_u8_gemm_256x16", "conv_u8_gemm_128x32", "conv_u8_gemm_32x32",
                                  "conv_u8_gemm_16x32"};
    return names[u8_cfg(a)];
}

hipError_t launch_conv_u8_gemm(const U8ConvArgs& a, hipStream_t s)
{
    const int OHW = a.OH * a.OW, N8 = OHW & ~7, ntail = OHW - N8;
    auto go = [&](auto kern, int tpx, int tcx) {
        dim3 grid((N8 + tpx - 1) / tpx + (ntail ? 1 : 0), (a.cout + tcx - 1) / tcx, a.N);
        hipLaunchKernelGGL(kern, grid, dim3(256), 0, s, a);
        return hipGetLastError();
    };
    switch (u8_cfg(a)) {
    case 1: return go(conv_u8_gemm_k<64, 4, 4, 4>, 256, 16);
    case 2: return go(conv_u8_gemm_k<32, 8, 4, 4>, 128, 32);
    case 3: return go(conv_u8_gemm_k<16, 16, 2, 2>, 32, 32);
    case 4: return go(conv_u8_gemm_k<16, 16, 1, 2>, 16, 32);
    default: return go(conv_u8_gemm_k<16, 16, 4, 4>, 64, 64);
    }
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
    a.y[(size_t)n * a.out_img + (size_t)(a.out_c0 + oc) * OHW + pj] = sat_u8(quant_round_div(total, a.out_scale, a.out_zp));
}

hipError_t launch_conv_u8_direct(const U8DirectArgs& a, hipStream_t s)
{
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
    a.y[(size_t)b * a.nout + o] = sat_u8(quant_round_div(data, a.out_scale, a.out_zp));
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
    a.y[i] = sat_u8(quant_round_div(r, a.out.scale, a.out.zp));
}

hipError_t launch_eltwise_u8(const U8EltArgs& a, hipStream_t s)
{
    hipLaunchKernelGGL(eltwise_u8_k, dim3((unsigned)((a.count + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace tamd
