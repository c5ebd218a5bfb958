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
//
// Mapping (CDNA4): lane = output pixel (64 consecutive pixels of one image per block), wave = TC output channels.
//   * the weights of a wave are wave-uniform -> fetched with SCALAR loads (s_load_dwordxN through the constant
//     address space) and fed to v_fma / v_pk_fma as the SGPR operand: no LDS or VGPR traffic for them at all;
//   * the dequantised im2col column of each pixel is staged once per block in LDS (k-major, 16 k per stage,
//     double buffered, conflict-free b32 both ways) and shared by the block's NW waves; the k -> (c,ky,kx) tap
//     table entry of a staging wave is wave-uniform too (scalar load);
//   * TC independent fused chains per lane, NW*TC channels per block; TC/NW are picked per layer so that the launch
//     has enough waves for 1024 SIMDs (a chain of K dependent FMAs is the critical path, K*4 clocks).
// Padded k rows carry w = 0 and an out-of-image lut entry: fma(0, 0, s) == s.
// =================================================================================================================
typedef const __attribute__((address_space(4))) float* cfloatp;     // constant address space => s_load
typedef const __attribute__((address_space(4))) int32_t* cint32p;

template <int NW, int TC, bool TAIL>
__device__ __forceinline__ void conv_u8_body(const U8ConvArgs& a, float (&xs)[2][16][64], int n, int jbase, int jlimit,
                                             int co0)
{
    constexpr int KC = 16, XE = KC / NW, NCH = TAIL ? 4 : 1, IB = TC * KC <= 32 ? KC : 32 / TC;
    static_assert(KC % NW == 0, "NW divides 16");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int co_w = co0 + wv * TC;
    const int K4 = a.K & ~3;
    const cfloatp wc = (cfloatp)(uintptr_t)a.wf;
    const cfloatp wrow = wc + (size_t)(co_w / TC) * a.Kpad * TC;      // [cout_pad/TC][Kpad][TC]
    const cint32p lutc = (cint32p)(uintptr_t)a.klut;      // {offset, dy << 16 | dx} pairs

    const int pj = jbase + lane;
    const bool valid = pj < jlimit;
    const int oy = valid ? pj / a.OW : 0, ox = valid ? pj - oy * a.OW : 0;
    const int iy0 = oy * a.SH - a.PH, ix0 = ox * a.SW - a.PW;
    const uint8_t* xin = a.x + (size_t)n * a.C * a.H * a.W;
    const int pbase = iy0 * a.W + ix0;

    unsigned xr[XE], xok = 0;      // raw bytes of the next stage; bit i of xok: tap i is inside the image
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < XE; i++) {
            const int ex = lutc[2 * (k0 + wv + i * NW)], ey = lutc[2 * (k0 + wv + i * NW) + 1];
            const int iy = iy0 + (ey >> 16), ix = ix0 + (ey & 0xffff);
            const bool ok = valid & ((unsigned)iy < (unsigned)a.H) & ((unsigned)ix < (unsigned)a.W);   // branch-free
            // unconditional load from a safe address: the wait for it can then sink below the FMA block
            xr[i] = xin[ok ? pbase + ex : 0];
            xok = ok ? (xok | (1u << i)) : (xok & ~(1u << i));
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < XE; i++)
            xs[buf][wv + i * NW][lane] = (xok >> i & 1u) ? dequant((uint8_t)xr[i], a.in_zp, a.in_scale) : 0.f;
    };

    float acc[NCH][TC];
#pragma unroll
    for (int r = 0; r < NCH; r++)
#pragma unroll
        for (int j = 0; j < TC; j++) acc[r][j] = 0.f;

    const int nchunk = a.Kpad / KC;
    gload(0);
    sstore(0);
    __syncthreads();
    for (int ch = 0; ch < nchunk; ch++) {
        const int cur = ch & 1, k0 = ch * KC;
        if (ch + 1 < nchunk) gload(k0 + KC);
        // the wave's weights of this stage are one contiguous run of KC*TC floats ([cout/TC][Kpad][TC] packing):
        // scalar loads with immediate offsets, no address arithmetic in the loop
        const cfloatp wk = wrow + (size_t)k0 * TC;
#pragma unroll 1
        for (int h = 0; h < KC; h += IB) {          // IB k per inner block bounds the live SGPRs to IB*TC <= 32
#pragma unroll
            for (int kk = 0; kk < IB; kk++) {
                const int kl = h + kk;
                const float xv = xs[cur][kl][lane];
                const bool live = !TAIL || (k0 + kl) < K4;      // the K%4 remainder is chained after the combine
#pragma unroll
                for (int j = 0; j < TC; j++) {
                    float w = wk[kl * TC + j];
                    if (TAIL) w = __uint_as_float(__float_as_uint(w) & (live ? 0xffffffffu : 0u));     // scalar ALU
                    float& s = acc[TAIL ? (kk & 3) : 0][j];
                    s = __builtin_fmaf(xv, w, s);
                }
            }
        }
        if (ch + 1 < nchunk) sstore(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: this lane's pixel, the wave's TC channels ----------------------------------------------
    if (!valid) return;
    const int OHW = a.OH * a.OW;
#pragma unroll
    for (int j = 0; j < TC; j++) {
        const int co = co_w + j;
        if (co >= a.cout) continue;
        float s;
        if constexpr (TAIL) {
            const float s0 = acc[0][j], s1 = acc[1][j], s2 = acc[2][j], s3 = acc[3][j];
            if (co < a.m_blocked) s = (0.f + (s0 + s1)) + (s2 + s3);
            else s = ((s0 + s1) + s2) + s3;
            for (int k = K4; k < a.K; k++) {
                const int ex = lutc[2 * k], ey = lutc[2 * k + 1];
                const int iy = iy0 + (ey >> 16), ix = ix0 + (ey & 0xffff);
                float v = 0.f;
                if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) v = dequant(xin[pbase + ex], a.in_zp, a.in_scale);
                s = __builtin_fmaf(wrow[(size_t)k * TC + j], v, s);
            }
        } else
            s = acc[0][j];
        if (a.bias) s = __builtin_fmaf((float)((cint32p)(uintptr_t)a.bias)[co], a.bias_scale, s);
        if (a.act == 0) s = s < 0.f ? 0.f : s;
        if (a.act > 0) { s = s < 0.f ? 0.f : s; s = s > 6.f ? 6.f : s; }
        a.y[(size_t)n * a.out_img + (size_t)(a.out_c0 + co) * OHW + pj] = sat_u8(quant_round_div(s, a.out_scale, a.out_zp));
    }
}

template <int NW, int TC>
__global__ __launch_bounds__(NW * 64) void conv_u8_gemm_k(const U8ConvArgs a)
{
    __shared__ float xs[2][16][64];
    const int OHW = a.OH * a.OW, N8 = OHW & ~7;
    const int tiles = (N8 + 63) / 64;
    const int n = blockIdx.z, co0 = blockIdx.y * (NW * TC);
    if ((int)blockIdx.x < tiles) conv_u8_body<NW, TC, false>(a, xs, n, blockIdx.x * 64, N8, co0);
    else conv_u8_body<NW, TC, true>(a, xs, n, N8, OHW, co0);
}

// tile choice: the widest channel tile per wave that still leaves >= ~2 waves for each of the 1024 SIMDs.
// The planner calls this once (geometry only), packs the weights for the chosen TC and stores the index in a.cfg.
static const struct { int nw, tc; const char* name; } U8_CFGS[] = {
    {4, 4, "conv_u8_px_w4x4"}, {8, 4, "conv_u8_px_w8x4"}, {16, 2, "conv_u8_px_w16x2"}, {16, 4, "conv_u8_px_w16x4"},
    {16, 8, "conv_u8_px_w16x8"}};

int conv_u8_gemm_pick(const U8ConvArgs& a)
{
    static const char* e = getenv("TAMD_U8_CFG");
    if (e && *e) return atoi(e) % 5;
    const int N8 = (a.OH * a.OW) & ~7;
    const long ptiles = (long)((N8 + 63) / 64 + ((a.OH * a.OW) & 7 ? 1 : 0)) * a.N;
    auto waves = [&](int nw, int tc) { return ptiles * ((a.cout + nw * tc - 1) / (nw * tc)) * nw; };
    if (a.cout <= 16) return 0;
    if (a.cout <= 32) return 1;
    if (waves(16, 8) >= 2048) return 4;
    if (waves(16, 4) >= 2048) return 3;
    return 2;
}
int conv_u8_gemm_tc(int cfg) { return U8_CFGS[cfg].tc; }
const char* conv_u8_gemm_kernel_name(const U8ConvArgs& a) { return U8_CFGS[a.cfg].name; }

hipError_t launch_conv_u8_gemm(const U8ConvArgs& a, hipStream_t s)
{
    const int OHW = a.OH * a.OW, N8 = OHW & ~7, ntail = OHW - N8;
    const int nw = U8_CFGS[a.cfg].nw, tc = U8_CFGS[a.cfg].tc;
    dim3 grid((N8 + 63) / 64 + (ntail ? 1 : 0), (a.cout + nw * tc - 1) / (nw * tc), a.N);
    switch (a.cfg) {
    case 0: hipLaunchKernelGGL((conv_u8_gemm_k<4, 4>), grid, dim3(256), 0, s, a); break;
    case 1: hipLaunchKernelGGL((conv_u8_gemm_k<8, 4>), grid, dim3(512), 0, s, a); break;
    case 3: hipLaunchKernelGGL((conv_u8_gemm_k<16, 4>), grid, dim3(1024), 0, s, a); break;
    case 4: hipLaunchKernelGGL((conv_u8_gemm_k<16, 8>), grid, dim3(1024), 0, s, a); break;
    default: hipLaunchKernelGGL((conv_u8_gemm_k<16, 2>), grid, dim3(1024), 0, s, a); break;
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
