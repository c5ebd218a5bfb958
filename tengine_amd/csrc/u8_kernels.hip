// uint8 (per-tensor asymmetric) kernels: the contract, and the members that are not group-1 convolutions.
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

typedef float v4f __attribute__((ext_vector_type(4)));
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

