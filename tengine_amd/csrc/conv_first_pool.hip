// ResNet-style stem in ONE launch: first-layer int8 convolution from the graph's NCHW input (C = 3, KH x KW <= 7 x 8,
// stride 2) + the 3x3 / stride-2 / unpadded MAX pooling that consumes it, writing only the pooled NHWC map.
//
// Arithmetic = conv_first.hip's row-granular kernel followed by pool_i8_kernel, value for value: the convolution's int8
// result (conv_kernel_x86.c:187-242 im2col, :1008-1630 sgemm_i8, :1826-1889 requantisation) is formed in full -- ReLU and
// saturation included -- before the maximum is taken, and the pooled byte goes through the reference's own rescale
// (pooling_kernel_ref_int8.c:156-166, y = round((float)max_q * (in_scale / out_scale))).  What changes is where the
// intermediate map lives: ResNet-50 at batch 32 wrote 25.7 MB of conv1 output and read it back for pool1 (two launches,
// 36.7 + 17.9 us, profiles/r03_layers_resnet50_int8_b32.txt); here a block keeps the 15 x 17 conv pixels behind its 7 x 8
// pooled pixels in LDS and only the 6.4 MB pooled map reaches memory.
//
//   * the block's INPUT PATCH (35 rows x 39 columns x C planes for 7x7/s2, zero-filled outside the image) is staged into LDS
//     once, twice: copy 0 as it is and copy 1 shifted by two bytes.  A conv pixel's patch row starts at column 2*cx (even), so
//     with the copy chosen by cx & 1 every 8-byte operand piece is two ALIGNED dwords: one ds_read2_b32 with an immediate
//     offset, no border tests, no shifts, no per-load address arithmetic (the global-memory gather of conv_first_rows costs
//     ~15 VALU instructions per piece).
//   * 15 x 17 = 255 conv pixels = 8 MFMA pixel tiles of 32: two per wave, weights (all K steps x cout tiles) resident in
//     registers for both.
//   * requantised conv bytes -> LDS [conv pixel][cout] with a 4-byte pad per pixel (17-dword pitch: the 32 lanes of a store
//     hit 32 different banks), then 56 x cout/4 pooling tasks read their (clipped) 3x3 window from LDS.
#include "epilogue.h"
#include "kernels.h"

namespace tamd {

typedef int v4i_fp __attribute__((ext_vector_type(4)));
typedef int v16i_fp __attribute__((ext_vector_type(16)));

namespace {
constexpr int FP_PH = 7, FP_PW = 8;                    // pooled pixels of a block
constexpr int FP_CH = 2 * FP_PH + 1, FP_CW = 2 * FP_PW + 1;      // conv pixels behind them: 15 x 17 = 255
constexpr int FP_NPIX = FP_CH * FP_CW;
static_assert(FP_NPIX <= 256, "eight MFMA pixel tiles");
}

// CT = cout tiles of 32; KH = conv kernel height (7: the only stem instantiated; KW <= 8 rides in the 8-byte row pieces);
// conv stride 2, pool 3x3 / 2 / pad 0
// RESCALE: the pool's output scale differs from its input scale (else round((float)m * 1) == m and the maxima are stored as they are)
template <int CT, int KH, bool RESCALE>
__global__ __launch_bounds__(256) void conv_first_pool_i8_kernel(FirstPoolArgs a)
{
    constexpr int IR = (FP_CH - 1) * 2 + KH;           // input rows of the patch (35 for KH = 7)
    constexpr int IWB = (FP_CW - 1) * 2 + 8;           // bytes of a patch row that any operand piece can touch (40)
    constexpr int PITCH = (IWB + 3) / 4 * 4;           // 40
    constexpr int PLANE = IR * PITCH;
    constexpr int COPY = 3 * PLANE;                    // one copy of the patch (3 planes)
    constexpr int NK = (3 * KH * 8 + 31) / 32;         // K steps of 32 (6 for KH = 7)
    constexpr int OPITCH = CT * 32 + 4;                // bytes per conv pixel in the result buffer
    extern __shared__ __attribute__((aligned(16))) int8_t smem[];
    int8_t* const patch = smem;                        // [2 copies][3 planes][IR][PITCH]
    int8_t* const cres = smem + 2 * COPY;              // [256 conv pixels][OPITCH]
    int* const sbias = reinterpret_cast<int*>(cres + 256 * OPITCH);
    float* const sscale = reinterpret_cast<float*>(sbias + CT * 32);

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, hi = lane >> 5;
    const int tiles_x = (a.POW + FP_PW - 1) / FP_PW, tiles_y = (a.POH + FP_PH - 1) / FP_PH;
    int b = blockIdx.x;
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y;
    const int n = b / tiles_y;
    const int py0 = ty * FP_PH, px0 = tx * FP_PW;      // pooled origin
    const int cy0 = 2 * py0, cx0 = 2 * px0;            // conv origin (pool stride 2, no pool padding)
    const int iy0 = cy0 * 2 - a.PH, ix0 = cx0 * 2 - a.PW;      // input origin of the patch (conv stride 2)

    // ---- weights: every K step of every cout tile, resident for both pixel tiles of the wave -------------------------------
    v4i_fp af[CT][NK];
#pragma unroll
    for (int i = 0; i < CT; i++)
#pragma unroll
        for (int ks = 0; ks < NK; ks++) af[i][ks] = *reinterpret_cast<const v4i_fp*>(a.w + (size_t)(i * 32 + l31) * a.kp + ks * 32 + hi * 16);
    if (t < CT * 32) { sbias[t] = a.bias[t]; sscale[t] = a.wscale[t]; }

    // ---- stage the input patch, both copies, one aligned 8-byte LDS piece per task -------------------------------------------
    // A copy is IR x (PITCH / 8) = 175 pieces per plane: thread t < 175 owns piece (row, j) of all six (plane, copy) pairs.  Every
    // load is issued before the first LDS store (a loop of load -> store pairs is one memory round trip per iteration: nine of
    // them in a row made the first version of this kernel slower than the convolution it replaces), and every load is
    // unconditional: the address is clamped into the row and the bytes that fall outside the image are shifted out.
    {
        constexpr int QW = PITCH / 8;
        static_assert(PITCH % 8 == 0 && IR * QW <= 256, "one 8-byte piece per thread and (plane, copy)");
        const int row = t / QW, j = t - row * QW;
        const int iy = iy0 + row, iyc = min(max(iy, 0), a.H - 1);
        const int8_t* xr = a.x + ((size_t)n * 3 * a.H + iyc) * a.W;
        const size_t plane = (size_t)a.H * a.W;
        const bool live = t < IR * QW;
        unsigned long long raw[3][2];
        int sh[2];
#pragma unroll
        for (int cp = 0; cp < 2; cp++) {
            const int ix = ix0 + 8 * j + 2 * cp, ixc = min(max(ix, 0), a.W - 8);
            // ix - ixc: < 0 -> the piece starts left of the image (bytes move up, zeros enter below); > 0 -> right of it
            sh[cp] = (iy == iyc) ? ix - ixc : 8;
#pragma unroll
            for (int c = 0; c < 3; c++) __builtin_memcpy(&raw[c][cp], xr + c * plane + ixc, 8);
        }
        if (live) {
#pragma unroll
            for (int cp = 0; cp < 2; cp++) {
                const int d = sh[cp];
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    // branch-free: up to 8 bytes out on either side, as two half shifts (a shift by 64 is not a shift)
                    const int l = min(max(-d, 0), 8), r = min(max(d, 0), 8);
                    unsigned long long v = raw[c][cp];
                    v = ((v << (4 * l)) << (4 * l));
                    v = ((v >> (4 * r)) >> (4 * r));
                    *reinterpret_cast<unsigned long long*>(patch + cp * COPY + c * PLANE + row * PITCH + 8 * j) = v;
                }
            }
        }
    }
    __syncthreads();

    // ---- convolution: two 32-pixel tiles per wave ---------------------------------------------------------------------------
    const Rq rq = a.rq;
    const bool win = rq_win(rq);
#pragma unroll 1
    for (int tt = 0; tt < 2; tt++) {
        const int q = (wave * 2 + tt) * 32 + l31;      // conv pixel of this lane (255 is padding: computed, never read)
        const int qq = q < FP_NPIX ? q : FP_NPIX - 1;
        const int cy = qq / FP_CW, cx = qq - cy * FP_CW;
        // patch row r = c * KH + ky of this pixel starts at byte (2 cy + ky) * PITCH + 2 cx of plane c; copy cx & 1 holds it
        // dword aligned at 2 cx - 2 (cx & 1)
        const int8_t* pb = patch + (cx & 1) * COPY + (2 * cy) * PITCH + 2 * (cx - (cx & 1));
        v4i_fp bf[NK];
#pragma unroll
        for (int ks = 0; ks < NK; ks++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
                // rows 4 ks + 2 hi + j: the two halves of the wave take different rows -> one select per piece
                constexpr int nrows = 3 * KH;
                const int r_lo = 4 * ks + j, r_hi = 4 * ks + 2 + j;
                const int o_lo = r_lo < nrows ? (r_lo / KH) * PLANE + (r_lo % KH) * PITCH : 0;
                const int o_hi = r_hi < nrows ? (r_hi / KH) * PLANE + (r_hi % KH) * PITCH : 0;
                const int8_t* p = pb + (hi ? o_hi : o_lo);
                const uint2 v = make_uint2(*reinterpret_cast<const unsigned*>(p), *reinterpret_cast<const unsigned*>(p + 4));
                bf[ks][2 * j] = (int)v.x;
                bf[ks][2 * j + 1] = (int)v.y;
            }
        v16i_fp acc[CT];                   // starts at the bias (register e of lane (pixel, hi) = channel 8 (e >> 2) + 4 hi + (e & 3) of the tile)
#pragma unroll
        for (int i = 0; i < CT; i++)
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) {
                const int4 b4 = *reinterpret_cast<const int4*>(&sbias[i * 32 + 8 * g4 + 4 * hi]);
                acc[i][4 * g4 + 0] = b4.x; acc[i][4 * g4 + 1] = b4.y; acc[i][4 * g4 + 2] = b4.z; acc[i][4 * g4 + 3] = b4.w;
            }
#pragma unroll
        for (int ks = 0; ks < NK; ks++)
#pragma unroll
            for (int i = 0; i < CT; i++) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[i][ks], bf[ks], acc[i], 0, 0, 0);
        // requantise (C/D layout: col = lane & 31 -> pixel, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) -> cout)
        auto requantise = [&](auto WIN) {
#pragma unroll
            for (int i = 0; i < CT; i++)
#pragma unroll
                for (int g4 = 0; g4 < 4; g4++) {
                    const int c0 = i * 32 + 8 * g4 + 4 * hi;
                    const float4 s4 = *reinterpret_cast<const float4*>(&sscale[c0]);
                    const unsigned p = requant4<decltype(WIN)::value>(acc[i][4 * g4 + 0], acc[i][4 * g4 + 1], acc[i][4 * g4 + 2], acc[i][4 * g4 + 3], s4, c0, rq);
                    *reinterpret_cast<unsigned*>(cres + q * OPITCH + c0) = p ^ 0x80808080u;      // biased: unsigned byte order == signed order
                }
        };
        if (win) requantise(std::integral_constant<int, 1>{});      // the stem's ReLU: the one-binade form of epilogue.h
        else requantise(std::integral_constant<int, 0>{});
    }
    __syncthreads();

    // ---- pooling: (pooled pixel, channel quad) tasks; window rows / columns clipped to the conv map as pool_i8_kernel does ----
    // (a clipped tap reads the window's first pixel again: the maximum does not change).  The conv bytes are biased by 128, so the
    // maximum of four channels is two v_pk_max_u16 on the even / odd bytes instead of four sign extensions + four compares.
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    const float prq = __fdiv_rn(a.pool_in_scale, a.pool_out_scale);
    constexpr int CQ = CT * 8;                                          // channel quads of the padded cout range
    const int cq_n = (a.cout + 3) / 4;
#pragma unroll
    for (int k = 0; k < (FP_PH * FP_PW * CQ + 255) / 256; k++) {
        const int task = t + 256 * k;
        const int cq = task % CQ, pp = task / CQ;
        const int py = pp / FP_PW, px = pp - py * FP_PW;
        if (pp >= FP_PH * FP_PW || cq >= cq_n || py0 + py >= a.POH || px0 + px >= a.POW) continue;
        const int ylim = a.OH - cy0 - 2 * py, xlim = a.OW - cx0 - 2 * px;   // taps dy < ylim, dx < xlim are inside the conv map (>= 1 each)
        const int8_t* wbase = cres + ((2 * py) * FP_CW + 2 * px) * OPITCH + 4 * cq;
        const int xo1 = 1 < xlim ? OPITCH : 0, xo2 = 2 < xlim ? 2 * OPITCH : 0;
        unsigned tap[9];                   // all nine reads first: one LDS round trip per task, not nine
#pragma unroll
        for (int dy = 0; dy < 3; dy++) {
            const int8_t* wr = wbase + (dy < ylim ? dy * FP_CW * OPITCH : 0);
            tap[3 * dy] = *reinterpret_cast<const unsigned*>(wr);
            tap[3 * dy + 1] = *reinterpret_cast<const unsigned*>(wr + xo1);
            tap[3 * dy + 2] = *reinterpret_cast<const unsigned*>(wr + xo2);
        }
        us2 me = __builtin_bit_cast(us2, tap[0] & 0x00ff00ffu), mo = __builtin_bit_cast(us2, (tap[0] >> 8) & 0x00ff00ffu);
#pragma unroll
        for (int i = 1; i < 9; i++) {
            me = __builtin_elementwise_max(me, __builtin_bit_cast(us2, tap[i] & 0x00ff00ffu));
            mo = __builtin_elementwise_max(mo, __builtin_bit_cast(us2, (tap[i] >> 8) & 0x00ff00ffu));
        }
        const unsigned mb = __builtin_bit_cast(unsigned, me) | (__builtin_bit_cast(unsigned, mo) << 8);      // biased maxima of the four channels
        unsigned o = mb ^ 0x80808080u;
        if constexpr (RESCALE) {           // pooling_kernel_ref_int8.c:156-166
            const int m0 = (int)(mb & 0xffu) - 128, m1 = (int)((mb >> 8) & 0xffu) - 128, m2 = (int)((mb >> 16) & 0xffu) - 128, m3 = (int)(mb >> 24) - 128;
            o = pack4(round_sat(__fmul_rn((float)m0, prq)), round_sat(__fmul_rn((float)m1, prq)),
                      round_sat(__fmul_rn((float)m2, prq)), round_sat(__fmul_rn((float)m3, prq)));
        }
        *reinterpret_cast<unsigned*>(a.y + (((size_t)n * a.POH + py0 + py) * a.POW + px0 + px) * a.ldc + a.c_off + 4 * cq) = o;
    }
}

static size_t fp_lds(int ct, int kh)
{
    const int ir = (FP_CH - 1) * 2 + kh, pitch = ((FP_CW - 1) * 2 + 8 + 3) / 4 * 4;
    return 2 * 3 * (size_t)ir * pitch + 256 * (size_t)(ct * 32 + 4) + (size_t)ct * 32 * 8;
}

// the stem shapes this launch covers: conv 7 x KW (KW <= 8), stride 2, dilation 1, C = 3, cout <= 128 from the
// NCHW input, row-granular weight packing with 8-byte rows (FirstArgs.kwp == 8); pool MAX 3x3 / stride 2 / no padding, any
// caffe_flavor (it only changes the divisor of AVG pooling), output channels on dword granularity
bool conv_first_pool_applicable(const FirstArgs& c, const PoolArgs& p)
{
    if (c.kwp != 8 || c.C != 3 || c.KH != 7 || c.KW > 8 || c.SH != 2 || c.SW != 2 || c.DH != 1 || c.DW != 1) return false;
    if (c.cout > 128 || c.kp != (3 * c.KH * 8 + 31) / 32 * 32 || c.W < 8) return false;      // (W >= 8: the staging loads are whole 8-byte pieces)
    if (p.method != 0 || p.KH != 3 || p.KW != 3 || p.SH != 2 || p.SW != 2 || p.PH != 0 || p.PW != 0) return false;
    if (p.H != c.OH || p.W != c.OW || p.C != c.cout || p.N != c.N) return false;
    if ((p.ldc | p.c_off) & 3) return false;
    // every pooled pixel's window must start inside the conv map (true for any geometry pool_geom produces without padding)
    return 2 * (p.OH - 1) < c.OH && 2 * (p.OW - 1) < c.OW;
}

FirstPoolArgs conv_first_pool_args(const FirstArgs& c, const PoolArgs& p)
{
    FirstPoolArgs a{};
    a.x = c.x; a.w = c.w; a.bias = c.bias; a.wscale = c.wscale; a.y = p.y;
    a.N = c.N; a.C = c.C; a.H = c.H; a.W = c.W; a.OH = c.OH; a.OW = c.OW; a.cout = c.cout;
    a.KH = c.KH; a.KW = c.KW; a.PH = c.PH; a.PW = c.PW; a.kp = c.kp; a.rq = c.rq;
    a.POH = p.OH; a.POW = p.OW; a.ldc = p.ldc; a.c_off = p.c_off;
    a.pool_in_scale = p.in_scale; a.pool_out_scale = p.out_scale;
    return a;
}

template <int KH, bool RESCALE>
static hipError_t fp_launch(const FirstPoolArgs& a, hipStream_t s)
{
    const int ct = (a.cout + 31) / 32;
    const int tiles = ((a.POW + FP_PW - 1) / FP_PW) * ((a.POH + FP_PH - 1) / FP_PH);
    const dim3 grid((unsigned)(a.N * tiles));
    const size_t lds = fp_lds(ct, KH);
    switch (ct) {
    case 1: hipLaunchKernelGGL((conv_first_pool_i8_kernel<1, KH, RESCALE>), grid, dim3(256), lds, s, a); break;
    case 2: hipLaunchKernelGGL((conv_first_pool_i8_kernel<2, KH, RESCALE>), grid, dim3(256), lds, s, a); break;
    case 3: hipLaunchKernelGGL((conv_first_pool_i8_kernel<3, KH, RESCALE>), grid, dim3(256), lds, s, a); break;
    case 4: hipLaunchKernelGGL((conv_first_pool_i8_kernel<4, KH, RESCALE>), grid, dim3(256), lds, s, a); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_conv_first_pool(const FirstPoolArgs& a, hipStream_t s)
{
    if (a.KH != 7) return hipErrorInvalidValue;
    // equal scales divide to exactly 1.0f, and round((float)m * 1.0f) == m for every int8 m the convolution can produce
    return a.pool_in_scale == a.pool_out_scale ? fp_launch<7, false>(a, s) : fp_launch<7, true>(a, s);
}

}  // namespace tamd
