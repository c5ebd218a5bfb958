// int8 depthwise 3x3 convolution, stride 1|2, NHWC, fused requantising epilogue.
//
// Replaces convdw3x3s1_int8_sse / convdw3x3s2_int8_sse + pad_int8
// (source/device/cpu/op/conv/x86/conv_dw_hcl_x86.c:42-95, :97-269, :271-445) and, for batch > 1,
// the depthwise case of ref_conv_int8 (conv/conv_kernel_ref_int8.c:42-177) -- the planner picks the
// epilogue formula the reference's score() would.
//
// HBM-bound work (AI ~ 4.5 op/B): no GEMM reshaping.  Lanes run along the channel dimension (NHWC ->
// consecutive lanes read consecutive bytes, every wave-level load is one contiguous segment), each
// lane owns CH channels and a strip of TW output pixels so every input byte fetched is reused for up to
// 3 horizontal taps from registers; the 3 input rows are the only re-read (L1/L2 hits).  No padded
// copy is made: out-of-image taps are predicated to 0 (== the reference's explicit zero pad).
#include "epilogue.h"
#include "kernels.h"

namespace tamd {

template <int NV> struct VecT;
template <> struct VecT<1> { typedef unsigned type; };
template <> struct VecT<2> { typedef uint2 type; };
template <> struct VecT<4> { typedef uint4 type; };

template <int NV> __device__ __forceinline__ void vload(unsigned (&d)[NV], const int8_t* p, bool ok)
{
    typedef typename VecT<NV>::type V;
    V v;
    if (ok) v = *reinterpret_cast<const V*>(p);
    const unsigned* s = reinterpret_cast<const unsigned*>(&v);
#pragma unroll
    for (int i = 0; i < NV; i++) d[i] = ok ? s[i] : 0u;
}

__device__ __forceinline__ int sx(unsigned v, int b) { return (int)(signed char)((v >> (8 * b)) & 0xff); }

// NV = dwords per lane (4*NV channels), TW = output pixels per lane along W, S = stride
template <int NV, int TW, int S>
__global__ __launch_bounds__(256) void dwconv3x3_i8_kernel(DwArgs a)
{
    constexpr int CH = 4 * NV;
    constexpr int COLS = (TW - 1) * S + 3;
    const int cgs = a.cw / CH;                          // channel groups per pixel
    const int strips = (a.OW + TW - 1) / TW;
    const long total = (long)a.N * a.OH * strips * cgs;
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int cg = (int)(idx % cgs); idx /= cgs;
    const int st = (int)(idx % strips); idx /= strips;
    const int oy = (int)(idx % a.OH);
    const int n = (int)(idx / a.OH);
    const int c0 = cg * CH;
    const int ox0 = st * TW;

    unsigned wv[9][NV];
#pragma unroll
    for (int k = 0; k < 9; k++) vload<NV>(wv[k], a.w + (size_t)k * a.cw + c0, true);

    int acc[TW][CH];
#pragma unroll
    for (int p = 0; p < TW; p++)
#pragma unroll
        for (int c = 0; c < CH; c++) acc[p][c] = 0;

    const int8_t* xn = a.x + (size_t)n * a.H * a.W * a.cs_in + c0;
#pragma unroll
    for (int ky = 0; ky < 3; ky++) {
        const int iy = oy * S - a.PH + ky;
        const bool rowok = iy >= 0 && iy < a.H;
        unsigned xv[COLS][NV];
#pragma unroll
        for (int col = 0; col < COLS; col++) {
            const int ix = ox0 * S - a.PW + col;
            const bool ok = rowok && ix >= 0 && ix < a.W;
            vload<NV>(xv[col], xn + ((size_t)(rowok ? iy : 0) * a.W + (ok ? ix : 0)) * a.cs_in, ok);
        }
#pragma unroll
        for (int p = 0; p < TW; p++)
#pragma unroll
            for (int kx = 0; kx < 3; kx++)
#pragma unroll
                for (int d = 0; d < NV; d++)
#pragma unroll
                    for (int b = 0; b < 4; b++)
                        acc[p][4 * d + b] += sx(xv[p * S + kx][d], b) * sx(wv[ky * 3 + kx][d], b);
    }

    int bias[CH];
    float ws[CH];
#pragma unroll
    for (int d = 0; d < NV; d++) {
        const int4 b4 = *reinterpret_cast<const int4*>(a.bias + c0 + 4 * d);
        const float4 s4 = *reinterpret_cast<const float4*>(a.wscale + c0 + 4 * d);
        bias[4 * d] = b4.x; bias[4 * d + 1] = b4.y; bias[4 * d + 2] = b4.z; bias[4 * d + 3] = b4.w;
        ws[4 * d] = s4.x; ws[4 * d + 1] = s4.y; ws[4 * d + 2] = s4.z; ws[4 * d + 3] = s4.w;
    }
#pragma unroll
    for (int p = 0; p < TW; p++) {
        const int ox = ox0 + p;
        if (ox >= a.OW) break;
        unsigned out[NV];
#pragma unroll
        for (int d = 0; d < NV; d++) {
            int q[4];
#pragma unroll
            for (int b = 0; b < 4; b++)
                q[b] = requant(acc[p][4 * d + b] + bias[4 * d + b], a.in_scale, ws[4 * d + b], a.out_scale, a.act, a.mode);
            out[d] = pack4(q[0], q[1], q[2], q[3]);
        }
        int8_t* yp = a.y + (((size_t)n * a.OH + oy) * a.OW + ox) * a.ldc + a.c_off + c0;
        typedef typename VecT<NV>::type V;
        V v;
        unsigned* vs = reinterpret_cast<unsigned*>(&v);
#pragma unroll
        for (int d = 0; d < NV; d++) vs[d] = out[d];
        *reinterpret_cast<V*>(yp) = v;
    }
}

template <int NV, int TW>
static hipError_t launch_dw(const DwArgs& a, hipStream_t s)
{
    const int cgs = a.cw / (4 * NV);
    const int strips = (a.OW + TW - 1) / TW;
    const long total = (long)a.N * a.OH * strips * cgs;
    const int grid = (int)((total + 255) / 256);
    if (a.S == 1)
        hipLaunchKernelGGL((dwconv3x3_i8_kernel<NV, TW, 1>), dim3(grid), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((dwconv3x3_i8_kernel<NV, TW, 2>), dim3(grid), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_dwconv3x3(const DwArgs& a, hipStream_t s)
{
    // enough lanes to fill 256 CUs x 8 waves first; then widen per-lane work for register reuse
    const long px = (long)a.N * a.OH * a.OW;
    // cw is a multiple of 16 by construction, so every vector width divides it
    if (px * (a.cw / 8) / 2 >= 256L * 64) return launch_dw<2, 2>(a, s);
    return launch_dw<1, 1>(a, s);
}

}  // namespace tamd
