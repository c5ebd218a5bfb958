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
#include "env.h"
#include "epilogue.h"
#include "kernels.h"

namespace tamd {

// S = stride, NF = 4-pixel fragments per row (1 or 2), TH = output rows per lane.  outputs per lane and row: S1 -> 2 / 6, S2 -> 1 / 3.
// Round 4: every load is unconditional (clamped address, the value masked afterwards) -- written as `ok ? load : 0` each of the 24
// loads of the <1,2> kernel sat in its own divergent branch (37 s_cbranch_execz, 147 v_mov, 1089 instructions for six output
// dwords); and a lane takes TH = 2 output rows where the map has them, so the rows between them are loaded and transposed once.
// WIN: the one-binade requantisation of epilogue.h (the node's window starts at 128.25: a fused ReLU / ReLU6), checked once by the kernel
template <int S, int NF, int TH, int WIN>
__device__ __forceinline__ void dwconv3x3_body(const DwArgs& a)
{
    constexpr int COLS = 4 * NF;
    constexpr int TW = (S == 1) ? (COLS - 2) : (NF == 1 ? 1 : 3);
    constexpr int ROWS = (TH - 1) * S + 3;                // input rows behind TH output rows
    const int cgs = a.cw / 4;                             // channel quads per pixel
    const int strips = (a.OW + TW - 1) / TW;
    const int bands = (a.OH + TH - 1) / TH;
    // blockIdx.y/z = (n, band of TH output rows): scalar divisions only; threads of a band = strips x quads
    const int rowg = blockIdx.y + blockIdx.z * 32768;
    if (rowg >= a.N * bands) return;
    const int n = rowg / bands, oy0 = (rowg - n * bands) * TH;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= strips * cgs) return;
    const int st = idx / cgs, cq = idx - st * cgs;
    const int c0 = cq * 4;
    const int ox0 = st * TW;

    const int8_t* xn = a.x + (size_t)n * a.H * a.W * a.cs_in + c0;
    const int ixb = ox0 * S - a.PW;
    // ALL input loads of the ROWS x COLS window first (one memory round trip per lane)
    unsigned raw[ROWS][COLS];
    int pixoff[COLS];
    unsigned colok = 0;
#pragma unroll
    for (int p = 0; p < COLS; p++) {
        const int ix = ixb + p;
        const bool ok = (unsigned)ix < (unsigned)a.W;
        pixoff[p] = (ok ? ix : 0) * a.cs_in;
        colok |= ok ? 1u << p : 0u;
    }
    unsigned rowok = 0;
#pragma unroll
    for (int r = 0; r < ROWS; r++) {
        const int iy = oy0 * S - a.PH + r;
        const bool ok = (unsigned)iy < (unsigned)a.H;
        rowok |= ok ? 1u << r : 0u;
        const int8_t* xr = xn + (size_t)(ok ? iy : 0) * a.W * a.cs_in;
#pragma unroll
        for (int p = 0; p < COLS; p++) raw[r][p] = *reinterpret_cast<const unsigned*>(xr + pixoff[p]);
    }
    // packed row taps: wrow[r][c] = {w[r][0], w[r][1], w[r][2], 0} of channel c0+c
    unsigned wrow[3][4];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const uint4 v = *reinterpret_cast<const uint4*>(a.w + ((size_t)r * a.cw + c0) * 4);
        wrow[r][0] = v.x; wrow[r][1] = v.y; wrow[r][2] = v.z; wrow[r][3] = v.w;
    }
    const int4 b4 = *reinterpret_cast<const int4*>(a.bias + c0);
    const float4 s4 = *reinterpret_cast<const float4*>(a.wscale + c0);
    // out-of-image pixels are zeros (== the reference's explicit zero pad), then per-channel fragments of every input row, once
    unsigned frag[ROWS][NF][4];
#pragma unroll
    for (int r = 0; r < ROWS; r++)
#pragma unroll
        for (int f = 0; f < NF; f++) {
            unsigned d[4];
#pragma unroll
            for (int q = 0; q < 4; q++) d[q] = ((rowok >> r) & (colok >> (4 * f + q)) & 1u) ? raw[r][4 * f + q] : 0u;
            transpose4x4(d, frag[r][f]);
        }
    const Rq rq = a.rq;
#pragma unroll
    for (int t = 0; t < TH; t++) {
        const int oy = oy0 + t;
        if (TH > 1 && oy >= a.OH) break;
        int acc[TW][4];
#pragma unroll
        for (int j = 0; j < TW; j++)
#pragma unroll
            for (int c = 0; c < 4; c++) acc[j][c] = c == 0 ? b4.x : c == 1 ? b4.y : c == 2 ? b4.z : b4.w;      // the bias is where the dot chain starts
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int j = 0; j < TW; j++) {
                const int sc = j * S;                         // first input column of this output's window
                const int f = sc >> 2, sh = sc & 3;
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    unsigned win, wt;
                    if (sh == 0) { win = frag[t * S + r][f][c]; wt = wrow[r][c]; }
                    else if (sh == 1) { win = frag[t * S + r][f][c]; wt = wrow[r][c] << 8; }
                    else { win = __builtin_amdgcn_alignbyte(frag[t * S + r][(f + 1) < NF ? f + 1 : f][c], frag[t * S + r][f][c], sh); wt = wrow[r][c]; }
                    acc[j][c] = __builtin_amdgcn_sdot4((int)win, (int)wt, acc[j][c], false);
                }
            }
        int8_t* yr = a.y + (((size_t)n * a.OH + oy) * a.OW + ox0) * a.ldc + a.c_off + c0;
#pragma unroll
        for (int j = 0; j < TW; j++) {
            const unsigned p = requant4<WIN>(acc[j][0], acc[j][1], acc[j][2], acc[j][3], s4, c0, rq);
            if (ox0 + j < a.OW) *reinterpret_cast<unsigned*>(yr + (size_t)j * a.ldc) = p;
        }
    }
}

template <int S, int NF, int TH>
__global__ __launch_bounds__(256) void dwconv3x3_i8_kernel(DwArgs a)
{
    if (rq_win(a.rq)) dwconv3x3_body<S, NF, TH, 1>(a);
    else dwconv3x3_body<S, NF, TH, 0>(a);
}

template <int S, int NF, int TH>
static hipError_t launch_dw(const DwArgs& a, hipStream_t s)
{
    constexpr int TW = (S == 1) ? (4 * NF - 2) : (NF == 1 ? 1 : 3);
    const int per_row = ((a.OW + TW - 1) / TW) * (a.cw / 4);
    const int rows = a.N * ((a.OH + TH - 1) / TH);
    // short rows use smaller blocks so that lanes are not wasted on the tail
    const int bs = per_row >= 192 ? 256 : (per_row >= 96 ? 128 : 64);
    dim3 grid((per_row + bs - 1) / bs, rows < 32768 ? rows : 32768, (rows + 32767) / 32768);
    hipLaunchKernelGGL((dwconv3x3_i8_kernel<S, NF, TH>), grid, dim3(bs), 0, s, a);
    return hipGetLastError();
}

// The form of a launch: NF = 4-pixel fragments per lane and input row (1: 2 | 1 outputs per row for stride 1 | 2; 2: 6 | 3), TH =
// output rows per lane.  Measured on MobileNet-shaped layers (profiles/r04_dw_forms.txt, isolated launches): the narrow strips
// win at every size -- more, lighter waves overlap their load / compute / store phases, the wide strips' waves all sit in the same
// phase -- and taller lanes (the rows between two outputs loaded and transposed once) pay as soon as the launch has rows to
// spare: 16 x 64 @ 112^2 stride 1: 18.6 us <1,2> (round 3's choice) -> 12.8 us <1,1,r4>; 64 x 512 @ 14^2: 11.4 -> 8.7 us <1,1,r2>.
// Batch-1 layers keep one row per lane (block count).  TAMD_DW_FORM="<nf><th>" pins a form (tests, experiments; read per launch).
static void dw_form(const DwArgs& a, int* nf, int* th)
{
    const long rows = (long)a.N * a.OH;
    *nf = 1;
    *th = rows >= 256 ? 2 : 1;
    if (a.S == 1 && a.OH >= 28 && rows >= 1024) *th = 4;
    if (const char* e = tamd_pin("dw_form")) {
        const int v = atoi(e);
        if (v / 10 >= 1 && v / 10 <= 2 && (v % 10 == 1 || v % 10 == 2 || (v % 10 == 4 && v / 10 == 1))) { *nf = v / 10; *th = v % 10; }
    }
}

const char* dwconv3x3_kernel_name(const DwArgs& a)
{
    static const char* names[2][2][3] = {{{"dwconv3x3_i8<1,1>", "dwconv3x3_i8<1,1,r2>", "dwconv3x3_i8<1,1,r4>"}, {"dwconv3x3_i8<1,2>", "dwconv3x3_i8<1,2,r2>", "?"}},
                                         {{"dwconv3x3_i8<2,1>", "dwconv3x3_i8<2,1,r2>", "dwconv3x3_i8<2,1,r4>"}, {"dwconv3x3_i8<2,2>", "dwconv3x3_i8<2,2,r2>", "?"}}};
    int nf, th;
    dw_form(a, &nf, &th);
    return names[a.S == 1 ? 0 : 1][nf - 1][th == 4 ? 2 : th - 1];
}

hipError_t launch_dwconv3x3(const DwArgs& a, hipStream_t s)
{
    int nf, th;
    dw_form(a, &nf, &th);
    if (a.S == 1) {
        if (nf == 2) return th == 2 ? launch_dw<1, 2, 2>(a, s) : launch_dw<1, 2, 1>(a, s);
        return th == 4 ? launch_dw<1, 1, 4>(a, s) : th == 2 ? launch_dw<1, 1, 2>(a, s) : launch_dw<1, 1, 1>(a, s);
    }
    if (nf == 2) return th == 2 ? launch_dw<2, 2, 2>(a, s) : launch_dw<2, 2, 1>(a, s);
    return th == 4 ? launch_dw<2, 1, 4>(a, s) : th == 2 ? launch_dw<2, 1, 2>(a, s) : launch_dw<2, 1, 1>(a, s);
}

}  // namespace tamd
