// First-layer int8 convolution straight from the graph's NCHW input (C <= 4: MobileNet conv1 3x3 s2,
// ResNet / SqueezeNet stems 7x7 / 3x3 s2) on MFMA, writing NHWC.
//
// Replaces the reference's im2col_int8 + sgemm_i8 + epilogue for these layers
// (source/device/cpu/op/conv/x86/conv_kernel_x86.c:187-242, :1008-1630, :1826-1889) and removes the
// separate NCHW->NHWC pass a layout-converting backend would need at the subgraph edge.
//
// GEMM view: D[cout][pixel] = sum_k W[cout][k] * P[pixel][k] with k = (c*KH+ky)*KW+kx -- exactly the
// OIHW order of the model's weight tensor, so weight rows are used as stored (zero padded to 32 k).
// The im2col patch P is never materialised: each lane gathers the 16 patch bytes of its MFMA operand
// from the (L1/L2 resident, 150 KB/img) input with a k -> (plane, dy, dx) table in LDS; out-of-image
// taps read as 0 (== the reference's zero padding).  One wave = one 32-pixel tile x all output
// channels, so the gathered operand is reused for every cout tile.  K is tiny (27..196) and the layer
// is bandwidth/latency shaped; the point of MFMA here is to keep the VALU free for the gather.
#include <type_traits>

#include "epilogue.h"
#include "kernels.h"

namespace tamd {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// CT = cout tiles of 32 (cout_pad = 32*CT)
template <int CT>
__global__ __launch_bounds__(256) void conv_first_i8_kernel(FirstArgs a)
{
    __shared__ int lut[256];             // k -> (c << 16) | (dy << 8) | dx ; 0x80000000 marks padding k
    const int t = threadIdx.x;
    const int kreal = a.C * a.KH * a.KW;
    if (t < a.kp) {
        int v = (int)0x80000000;
        if (t < kreal) {
            const int kx = t % a.KW, r = t / a.KW;
            const int ky = r % a.KH, c = r / a.KH;
            v = (c << 16) | ((ky * a.DH) << 8) | (kx * a.DW);
        }
        lut[t] = v;
    }
    __syncthreads();

    const int lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const long M = (long)a.N * a.OH * a.OW;
    const long tile = (long)blockIdx.x * 4 + wave;
    const long m0 = tile * 32;
    if (m0 >= M) return;
    const long m = m0 + l31;
    const bool mvalid = m < M;
    const int ohw = a.OH * a.OW;
    const long mm = mvalid ? m : 0;
    const int n = (int)(mm / ohw);
    const int rem = (int)(mm - (long)n * ohw);
    const int oy = rem / a.OW, ox = rem - oy * a.OW;
    const int iy0 = oy * a.SH - a.PH, ix0 = ox * a.SW - a.PW;
    const int8_t* xn = a.x + (size_t)n * a.C * a.H * a.W;
    const int hw = a.H * a.W;

    v16i acc[CT];
#pragma unroll
    for (int i = 0; i < CT; i++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[i][e] = 0;

    const int nk = a.kp / 32;
    for (int ks = 0; ks < nk; ks++) {
        const int kb = ks * 32 + hi * 16;
        unsigned pk[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const int d = lut[kb + e];
            const int c = (d >> 16) & 0xff, dy = (d >> 8) & 0xff, dx = d & 0xff;
            const int iy = iy0 + dy, ix = ix0 + dx;
            const bool ok = mvalid && d >= 0 && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            int v = 0;
            if (ok) v = (int)xn[(size_t)c * hw + (size_t)iy * a.W + ix];
            pk[e >> 2] |= ((unsigned)(v & 0xff)) << (8 * (e & 3));
        }
        v4i bf = {(int)pk[0], (int)pk[1], (int)pk[2], (int)pk[3]};
#pragma unroll
        for (int i = 0; i < CT; i++) {
            const v4i af = *reinterpret_cast<const v4i*>(a.w + (size_t)(i * 32 + l31) * a.kp + kb);
            acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, bf, acc[i], 0, 0, 0);
        }
    }

    // epilogue (C/D layout: col = lane&31 -> pixel, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) -> cout)
    const Rq rq = a.rq;
    const bool wide = ((a.c_limit | a.c_off | a.ldc) & 15) == 0;
#pragma unroll
    for (int i = 0; i < CT; i++) {
        unsigned p[4];
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) {
            const int c0 = i * 32 + 8 * g4 + 4 * hi;
            const int4 b4 = *reinterpret_cast<const int4*>(a.bias + c0);
            const float4 s4 = *reinterpret_cast<const float4*>(a.wscale + c0);
            p[g4] = requant4(acc[i][4 * g4 + 0] + b4.x, acc[i][4 * g4 + 1] + b4.y, acc[i][4 * g4 + 2] + b4.z,
                             acc[i][4 * g4 + 3] + b4.w, s4, c0, rq);
        }
        if (wide) {
            half_wave_regroup(p);
            const int c16 = i * 32 + hi * 16;
            if (mvalid && c16 < a.c_limit)
                *reinterpret_cast<uint4*>(a.y + (size_t)m * a.ldc + a.c_off + c16) = make_uint4(p[0], p[1], p[2], p[3]);
        } else {
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) {
                const int c0 = i * 32 + 8 * g4 + 4 * hi;
                if (mvalid && c0 < a.c_limit) *reinterpret_cast<unsigned*>(a.y + (size_t)m * a.ldc + a.c_off + c0) = p[g4];
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Row-granular variant (dilation_w == 1, KW <= 8, the MobileNet / ResNet / SqueezeNet stems): k is ordered
// (c, ky, kx) with kx padded to KWP = 4 or 8, so the KWP operand bytes of one (c, ky) row of a pixel's patch are
// KWP *consecutive* input bytes -- one unaligned 4- / 8-byte load (gfx950 global loads are byte-aligned) instead of
// KWP bounds-checked byte gathers with ~15 VALU instructions each.  The left image border shifts zeros in, the right
// border and the kx >= KW padding are masked (the latter meets zero weights anyway); out-of-image rows are zero.
// The last row of the input may be over-read by < KWP bytes: the planner allocates that slack.
// ---------------------------------------------------------------------------------------------------------------
template <int CT, int KWP>
__global__ __launch_bounds__(256) void conv_first_rows_i8_kernel(FirstArgs a)
{
    constexpr int R = 16 / KWP;          // patch rows per 16-byte MFMA operand
    constexpr int NKMAX = 6;             // kp <= 192
    __shared__ int rowoff[64];           // row (c*KH + ky) -> c*H*W + ky*DH*W
    __shared__ int rowdy[64];            //                 -> ky*DH (a value no image row can reach for padding rows)
    const int t = threadIdx.x;
    if (t < 64) {                        // NKMAX*32/KWP <= 48 entries are ever read
        const int nrows = a.C * a.KH;
        const int c = t / a.KH, ky = t - c * a.KH;
        rowoff[t] = t < nrows ? c * a.H * a.W + ky * a.DH * a.W : 0;
        rowdy[t] = t < nrows ? ky * a.DH : (1 << 24);
    }
    __syncthreads();

    const int lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const long M = (long)a.N * a.OH * a.OW;
    const long tile = (long)blockIdx.x * 4 + wave;
    const long m0 = tile * 32;
    if (m0 >= M) return;
    const long m = m0 + l31;
    const bool mvalid = m < M;
    const int ohw = a.OH * a.OW;
    const long mm = mvalid ? m : 0;
    const int n = (int)(mm / ohw);
    const int rem = (int)(mm - (long)n * ohw);
    const int oy = rem / a.OW, ox = rem - oy * a.OW;
    const int iy0 = oy * a.SH - a.PH, ix0 = ox * a.SW - a.PW;
    const int8_t* xn = a.x + (size_t)n * a.C * a.H * a.W;
    // column handling is the same for every row of this pixel's patch
    const int sft = ix0 < 0 ? -ix0 : 0;                    // operand bytes left of the image: shifted in as zeros
    const int xs = ix0 < 0 ? 0 : ix0;                      // first byte that is actually loaded
    const int nvalid = a.W - ix0;                          // operand byte t is inside the image iff sft <= t < nvalid
    const bool colok = mvalid && nvalid > 0 && sft < KWP;
    const int nk = a.kp / 32;
    typedef typename std::conditional<KWP == 8, unsigned long long, unsigned>::type row_t;
    const row_t cmask = nvalid < KWP ? (row_t)(((row_t)1 << (8 * (nvalid > 0 ? nvalid : 0))) - 1) : (row_t)~(row_t)0;

    // phase 1: every input load of the tile is issued before anything is consumed, all of them unconditional (clamped
    // addresses, padding rows of the table point at the image start): one memory round trip per tile instead of one
    // per load.  Steps >= nk read padding rows and are simply not multiplied.
    row_t raw[NKMAX][R];
    unsigned okm = 0;
#pragma unroll
    for (int ks = 0; ks < NKMAX; ks++) {
        const int r0 = (ks * 32 + hi * 16) / KWP;
#pragma unroll
        for (int j = 0; j < R; j++) {
            const int iy = iy0 + rowdy[r0 + j], ro = rowoff[r0 + j] + iy0 * a.W + xs;
            const bool ok = colok && (unsigned)iy < (unsigned)a.H;
            const int8_t* p = xn + (ok ? ro : 0);
            __builtin_memcpy(&raw[ks][j], p, sizeof(row_t));
            okm |= ok ? 1u << (ks * R + j) : 0u;
        }
    }
    // phase 2: borders, then the MFMAs
    v4i bf[NKMAX];
#pragma unroll
    for (int ks = 0; ks < NKMAX; ks++)
#pragma unroll
        for (int j = 0; j < R; j++) {
            row_t v = (okm >> (ks * R + j) & 1u) ? (row_t)(raw[ks][j] << (8 * sft)) & cmask : (row_t)0;
            if constexpr (KWP == 8) {
                bf[ks][2 * j] = (int)(unsigned)v;
                bf[ks][2 * j + 1] = (int)(unsigned)((unsigned long long)v >> 32);
            } else
                bf[ks][j] = (int)v;
        }
    v16i acc[CT];
#pragma unroll
    for (int i = 0; i < CT; i++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[i][e] = 0;
#pragma unroll
    for (int ks = 0; ks < NKMAX; ks++)
        if (ks < nk) {           // weight rows are L1/L2-hot and loaded per step: preloading all of them costs 24*CT VGPRs
            const int kb = ks * 32 + hi * 16;
            v4i af[CT];
#pragma unroll
            for (int i = 0; i < CT; i++) af[i] = *reinterpret_cast<const v4i*>(a.w + (size_t)(i * 32 + l31) * a.kp + kb);
#pragma unroll
            for (int i = 0; i < CT; i++) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[i], bf[ks], acc[i], 0, 0, 0);
        }

    // epilogue (C/D layout: col = lane&31 -> pixel, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) -> cout)
    const Rq rq = a.rq;
    const bool wide = ((a.c_limit | a.c_off | a.ldc) & 15) == 0;
#pragma unroll
    for (int i = 0; i < CT; i++) {
        unsigned p[4];
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) {
            const int c0 = i * 32 + 8 * g4 + 4 * hi;
            const int4 b4 = *reinterpret_cast<const int4*>(a.bias + c0);
            const float4 s4 = *reinterpret_cast<const float4*>(a.wscale + c0);
            p[g4] = requant4(acc[i][4 * g4 + 0] + b4.x, acc[i][4 * g4 + 1] + b4.y, acc[i][4 * g4 + 2] + b4.z,
                             acc[i][4 * g4 + 3] + b4.w, s4, c0, rq);
        }
        if (wide) {
            half_wave_regroup(p);
            const int c16 = i * 32 + hi * 16;
            if (mvalid && c16 < a.c_limit)
                *reinterpret_cast<uint4*>(a.y + (size_t)m * a.ldc + a.c_off + c16) = make_uint4(p[0], p[1], p[2], p[3]);
        } else {
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) {
                const int c0 = i * 32 + 8 * g4 + 4 * hi;
                if (mvalid && c0 < a.c_limit) *reinterpret_cast<unsigned*>(a.y + (size_t)m * a.ldc + a.c_off + c0) = p[g4];
            }
        }
    }
}

// k-row width the row-granular kernel would use for this layer; 0: not applicable (generic gather kernel)
int conv_first_kwp(int C, int KH, int KW, int DW)
{
    if (DW != 1 || KW > 8) return 0;
    const int kwp = KW <= 4 ? 4 : 8;
    return (C * KH * kwp + 31) / 32 * 32 <= 192 ? kwp : 0;
}

template <int KWP>
static hipError_t launch_rows(const FirstArgs& a, dim3 grid, int ct, hipStream_t s)
{
    switch (ct) {
    case 1: hipLaunchKernelGGL((conv_first_rows_i8_kernel<1, KWP>), grid, dim3(256), 0, s, a); break;
    case 2: hipLaunchKernelGGL((conv_first_rows_i8_kernel<2, KWP>), grid, dim3(256), 0, s, a); break;
    case 3: hipLaunchKernelGGL((conv_first_rows_i8_kernel<3, KWP>), grid, dim3(256), 0, s, a); break;
    case 4: hipLaunchKernelGGL((conv_first_rows_i8_kernel<4, KWP>), grid, dim3(256), 0, s, a); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_conv_first(const FirstArgs& a, hipStream_t s)
{
    const long M = (long)a.N * a.OH * a.OW;
    const long tiles = (M + 31) / 32;
    dim3 grid((unsigned)((tiles + 3) / 4));
    const int ct = (a.cout + 31) / 32;
    if (a.kwp == 4) return launch_rows<4>(a, grid, ct, s);
    if (a.kwp == 8) return launch_rows<8>(a, grid, ct, s);
    switch (ct) {
    case 1: hipLaunchKernelGGL(conv_first_i8_kernel<1>, grid, dim3(256), 0, s, a); break;
    case 2: hipLaunchKernelGGL(conv_first_i8_kernel<2>, grid, dim3(256), 0, s, a); break;
    case 3: hipLaunchKernelGGL(conv_first_i8_kernel<3>, grid, dim3(256), 0, s, a); break;
    case 4: hipLaunchKernelGGL(conv_first_i8_kernel<4>, grid, dim3(256), 0, s, a); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace tamd
