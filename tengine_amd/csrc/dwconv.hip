// int8 depthwise 3x3 convolution, stride 1|2, NHWC, fused requantising epilogue.
//
// Replaces convdw3x3s1_int8_sse / convdw3x3s2_int8_sse + pad_int8
// (source/device/cpu/op/conv/x86/conv_dw_hcl_x86.c:42-95, :97-269, :271-445) and, for batch > 1,
// the depthwise case of ref_conv_int8 (conv/conv_kernel_ref_int8.c:42-177) -- the planner picks the
// epilogue formula the reference's score() would.
//
// HBM-bound work (AI ~ 4.5 op/B): no GEMM reshaping, the job is to keep the VALU cost per output byte
// below the memory time.  Lanes run along the channel dimension (NHWC: consecutive lanes read consecutive
// dwords of one pixel -> coalesced), one lane = 4 channels x a strip of output pixels.  Per input row the
// lane loads 4 (or 8) horizontally adjacent pixel dwords {c0..c3} and transposes each 4x4 byte block with
// 8 v_perm_b32 into per-channel fragments {x0..x3}; a 3-tap row of the filter then is ONE v_dot4_i32_i8
// against the packed taps {w0,w1,w2,0} (or {0,w0,w1,w2}, or a v_alignbyte window across two fragments):
// ~2 VALU per output-row instead of 3 sign-extends + 3 multiply-adds.  No padded copy is made:
// out-of-image pixels are loaded as 0 (== the reference's explicit zero pad).
#include "dw_common.h"
#include "epilogue.h"
#include "kernels.h"

namespace tamd {

// S = stride, NF = 4-pixel fragments per row (1 or 2).  outputs per lane: S1 -> 2 / 6, S2 -> 1 / 3
template <int S, int NF>
__global__ __launch_bounds__(256) void dwconv3x3_i8_kernel(DwArgs a)
{
    constexpr int COLS = 4 * NF;
    constexpr int TW = (S == 1) ? (COLS - 2) : (NF == 1 ? 1 : 3);
    const int cgs = a.cw / 4;                             // channel quads per pixel
    const int strips = (a.OW + TW - 1) / TW;
    // blockIdx.y/z = output row (n, oy): scalar divisions only; threads of a row = strips x quads
    const int row = blockIdx.y + blockIdx.z * 32768;
    if (row >= a.N * a.OH) return;
    const int n = row / a.OH, oy = row - n * a.OH;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= strips * cgs) return;
    const int st = idx / cgs, cq = idx - st * cgs;
    const int c0 = cq * 4;
    const int ox0 = st * TW;

    // packed row taps: wrow[r][c] = {w[r][0], w[r][1], w[r][2], 0} of channel c0+c
    unsigned wrow[3][4];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const uint4 v = *reinterpret_cast<const uint4*>(a.w + ((size_t)r * a.cw + c0) * 4);
        wrow[r][0] = v.x; wrow[r][1] = v.y; wrow[r][2] = v.z; wrow[r][3] = v.w;
    }

    int acc[TW][4];
#pragma unroll
    for (int j = 0; j < TW; j++)
#pragma unroll
        for (int c = 0; c < 4; c++) acc[j][c] = 0;

    const int8_t* xn = a.x + (size_t)n * a.H * a.W * a.cs_in + c0;
    const int ixb = ox0 * S - a.PW;
    // issue ALL input loads of the 3 x COLS window first (one memory round trip per lane, not three)
    unsigned raw[3][COLS];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const int iy = oy * S - a.PH + r;
        const bool rowok = iy >= 0 && iy < a.H;
        const int8_t* xr = xn + (size_t)(rowok ? iy : 0) * a.W * a.cs_in;
#pragma unroll
        for (int p = 0; p < COLS; p++) {
            const int ix = ixb + p;
            const bool ok = rowok && ix >= 0 && ix < a.W;
            raw[r][p] = ok ? *reinterpret_cast<const unsigned*>(xr + (ok ? ix : 0) * a.cs_in) : 0u;
        }
    }
#pragma unroll
    for (int r = 0; r < 3; r++) {
        unsigned frag[NF][4];
#pragma unroll
        for (int f = 0; f < NF; f++) {
            const unsigned d[4] = {raw[r][4 * f], raw[r][4 * f + 1], raw[r][4 * f + 2], raw[r][4 * f + 3]};
            transpose4x4(d, frag[f]);
        }
#pragma unroll
        for (int j = 0; j < TW; j++) {
            constexpr int dummy = 0;
            const int sc = j * S;                         // first input column of this output's window
            const int f = sc >> 2, sh = sc & 3;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                unsigned win, wt;
                if (sh == 0) { win = frag[f][c]; wt = wrow[r][c]; }
                else if (sh == 1) { win = frag[f][c]; wt = wrow[r][c] << 8; }
                else { win = __builtin_amdgcn_alignbyte(frag[(f + 1) < NF ? f + 1 : f][c], frag[f][c], sh); wt = wrow[r][c]; }
                acc[j][c] = __builtin_amdgcn_sdot4((int)win, (int)wt, acc[j][c], false);
            }
            (void)dummy;
        }
    }

    const Rq rq = a.rq;
    const int4 b4 = *reinterpret_cast<const int4*>(a.bias + c0);
    const float4 s4 = *reinterpret_cast<const float4*>(a.wscale + c0);
#pragma unroll
    for (int j = 0; j < TW; j++) {
        const int ox = ox0 + j;
        const unsigned p = requant4(acc[j][0] + b4.x, acc[j][1] + b4.y, acc[j][2] + b4.z, acc[j][3] + b4.w, s4, c0, rq);
        if (ox < a.OW)
            *reinterpret_cast<unsigned*>(a.y + (((size_t)n * a.OH + oy) * a.OW + ox) * a.ldc + a.c_off + c0) = p;
    }
}

template <int S, int NF>
static hipError_t launch_dw(const DwArgs& a, hipStream_t s)
{
    constexpr int TW = (S == 1) ? (4 * NF - 2) : (NF == 1 ? 1 : 3);
    const int per_row = ((a.OW + TW - 1) / TW) * (a.cw / 4);
    const int rows = a.N * a.OH;
    // short rows use smaller blocks so that lanes are not wasted on the tail
    const int bs = per_row >= 192 ? 256 : (per_row >= 96 ? 128 : 64);
    dim3 grid((per_row + bs - 1) / bs, rows < 32768 ? rows : 32768, (rows + 32767) / 32768);
    hipLaunchKernelGGL((dwconv3x3_i8_kernel<S, NF>), grid, dim3(bs), 0, s, a);
    return hipGetLastError();
}

// two-fragment strips (6 / 3 outputs per lane) once there is enough work to fill the chip with them
// and the row is long enough not to waste the strip tail; else the short strips
static bool dw_wide(const DwArgs& a)
{
    const long px = (long)a.N * a.OH * a.OW;
    return px * (a.cw / 4) >= 6L * 256 * 256 * 4 && a.OW >= 12;
}

const char* dwconv3x3_kernel_name(const DwArgs& a)
{
    const bool wide = dw_wide(a);
    if (a.S == 1) return wide ? "dwconv3x3_i8<1,2>" : "dwconv3x3_i8<1,1>";
    return wide ? "dwconv3x3_i8<2,2>" : "dwconv3x3_i8<2,1>";
}

hipError_t launch_dwconv3x3(const DwArgs& a, hipStream_t s)
{
    const bool wide = dw_wide(a);
    if (a.S == 1) return wide ? launch_dw<1, 2>(a, s) : launch_dw<1, 1>(a, s);
    return wide ? launch_dw<2, 2>(a, s) : launch_dw<2, 1>(a, s);
}

}  // namespace tamd
