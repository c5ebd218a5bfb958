// Row-major streaming int8 pointwise (1x1) convolution on MFMA: shallow K (cin <= 128), many pixels, output-heavy layers
// (ResNet branch2c / branch1 with their residual tails, the early MobileNet / SSD pointwise layers).
//
// Same arithmetic, weight buffer and epilogue as the other members of the GEMM family (reference chain:
// conv_kernel_x86.c:187-242, :963-1007, :1008-1630, :1796-1893; bit-exact epilogue.h).  What differs from pw_stream.hip is
// the ORIENTATION of the MFMA and the schedule:
//   * the activations are the A operand (rows = 32 pixels), the weights the B operand, and column l of cout tile i is output
//     channel 4 l + i of the block's 128-channel group.  A lane therefore owns the four consecutive channels 4 l .. 4 l + 3
//     of every pixel row it holds: its bias / multiplier vectors are loop-invariant registers (no LDS, no per-group reads),
//     requant4 packs the four accumulators of a row into ONE dword without any cross-lane regrouping, and the dword stores
//     of a half-wave cover 128 contiguous bytes of one pixel -- full cache lines, where the 32 x 32-byte pieces of the
//     column-major epilogue leave the write combining to the L2.  The residual operand of a fused eltwise tail is read with
//     the same full-line pattern.
//   * persistent waves with a software pipeline: a wave walks 32-pixel tiles with a grid-sized stride and requests the next
//     tile's activations (and residual rows) before it multiplies and requantises the current one, so every resident wave
//     keeps a tile in flight; the grid is sized to the device (blocks per CU from the occupancy query), not to the layer.
#include "epilogue.h"
#include "env.h"
#include "kernels.h"

namespace tamd {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int S, bool ELT>
__global__ __launch_bounds__(256) void pw_rows_i8_kernel(ConvArgs a)
{
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int c = blockIdx.y * 128 + 4 * l31;          // this lane's four output channels
    const bool cvalid = c < a.c_limit;
    const int cr = cvalid ? c : 0;                     // address-safe channel for the residual loads of idle lanes

    v4i wf[4][S];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int s = 0; s < S; s++)
            wf[i][s] = *reinterpret_cast<const v4i*>(a.w + (size_t)(c + i) * a.kpad + s * 32 + hi * 16);
    const int4 b4 = *reinterpret_cast<const int4*>(a.bias + c);
    const float4 m4 = *reinterpret_cast<const float4*>(a.wscale + c);
    const Rq rq = a.rq;
    const float inv_elt = (ELT && a.elt.thr <= 0.f) ? __fdiv_rn(1.0f, a.elt.out_scale) : 1.f;
    const float inv_relu = (ELT && a.elt.thr <= 0.f && a.elt.relu) ? __fdiv_rn(1.0f, a.elt.relu_out_scale) : 1.f;
    // K bytes past the tensor's channels meet zero weights: any readable bytes will do, so those steps re-read offset 0
    int koff[S];
#pragma unroll
    for (int s = 0; s < S; s++) koff[s] = (s * 32 + hi * 16) < a.ktot ? s * 32 + hi * 16 : 0;

    const int tiles_m = (a.M + 31) >> 5;
    const int stride = gridDim.x * 4;
    int tile = blockIdx.x * 4 + wave;
    if (tile >= tiles_m) return;
    const int mlast = a.M - 1;

    auto load_x = [&](int tl, v4i (&bf)[S]) {
        int m = tl * 32 + l31;
        m = m < mlast ? m : mlast;
        const int8_t* xp = a.x + (size_t)m * a.cs_in;
#pragma unroll
        for (int s = 0; s < S; s++) bf[s] = *reinterpret_cast<const v4i*>(xp + koff[s]);
    };
    auto load_r = [&](int tl, unsigned (&rv)[16]) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            int m = tl * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            m = m < mlast ? m : mlast;
            rv[r] = *reinterpret_cast<const unsigned*>(a.elt.res + (size_t)m * a.elt.res_ldc + a.elt.res_c_off + cr);
        }
    };

    v4i bf[S];
    unsigned rv[ELT ? 16 : 1];
    load_x(tile, bf);
    if (ELT) load_r(tile, reinterpret_cast<unsigned (&)[16]>(rv));
    for (; tile < tiles_m; tile += stride) {
        // the next tile's operands first (the last iteration re-requests its own tile: unconditional loads, counted waits)
        const int nt = tile + stride < tiles_m ? tile + stride : tile;
        v4i bn[S];
        unsigned rn[ELT ? 16 : 1];
        load_x(nt, bn);
        if (ELT) load_r(nt, reinterpret_cast<unsigned (&)[16]>(rn));

        v16i acc[4];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][e] = 0;
#pragma unroll
        for (int s = 0; s < S; s++)
#pragma unroll
            for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(bf[s], wf[i][s], acc[i], 0, 0, 0);

        // C/D layout: column = lane & 31 -> channel c + i of tile i; register r -> pixel row (r & 3) + 8 (r >> 2) + 4 hi
        static_for<0, 16>([&](auto R) {
            constexpr int r = decltype(R)::value;
            const int m = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            unsigned p = requant4(acc[0][r] + b4.x, acc[1][r] + b4.y, acc[2][r] + b4.z, acc[3][r] + b4.w, m4, c, rq);
            if constexpr (ELT) p = a.elt.thr > 0.f ? elt_sum4_fold(p, rv[r], a.elt) : fuse_elt4_cold(p, rv[r], a.elt.type, a.elt.conv_is_first, a.elt.s_conv, a.elt.s_res, a.elt.out_scale, a.elt.relu,
                                                                                                a.elt.relu_out_scale, inv_elt, inv_relu);
            if (m <= mlast && cvalid) *reinterpret_cast<unsigned*>(a.y + (size_t)m * a.ldc + a.c_off + c) = p;
        });
#pragma unroll
        for (int s = 0; s < S; s++) bf[s] = bn[s];
        if (ELT) {
#pragma unroll
            for (int r = 0; r < 16; r++) rv[r] = rn[r];
        }
    }
}

bool pw_rows_applicable(const ConvArgs& a)
{
    const bool is1x1 = (a.KH == 1 && a.KW == 1 && a.SH == 1 && a.SW == 1 && a.PH == 0 && a.PW == 0);
    const int S = a.kpad / 32;
    if (a.elt.res && ((a.elt.res_ldc | a.elt.res_c_off) & 3)) return false;        // dword reads of the residual operand
    return is1x1 && S <= 4 && a.M >= 2048 && ((a.c_limit | a.c_off | a.ldc) & 3) == 0;
}

template <int S, bool ELT>
static hipError_t launch_rows(const ConvArgs& a, hipStream_t s)
{
    // persistent grid: as many blocks as the device holds at once (occupancy query, once per variant)
    static int resident = 0;
    if (resident == 0) {
        int dev = 0, cus = 256, per_cu = 2;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pw_rows_i8_kernel<S, ELT>, 256, 0) != hipSuccess || per_cu < 1) per_cu = 2;
        (void)hipGetLastError();
        const char* e = exp_env("TAMD_PW_ROWS_BPC");             // experiments: blocks per CU
        if (e && atoi(e) > 0) per_cu = atoi(e);
        resident = cus * per_cu;
    }
    const int tiles_m = (a.M + 31) / 32;
    const int groups = (a.cout + 127) / 128;
    int bx = (tiles_m + 3) / 4;
    const int cap = resident / groups > 0 ? resident / groups : 1;
    if (bx > cap) bx = cap;
    hipLaunchKernelGGL((pw_rows_i8_kernel<S, ELT>), dim3(bx, groups), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_pw_rows(const ConvArgs& a, hipStream_t s)
{
    const int S = a.kpad / 32;      // kpad is a multiple of 64 -> 2 or 4
    if (S <= 2) return a.elt.res ? launch_rows<2, true>(a, s) : launch_rows<2, false>(a, s);
    return a.elt.res ? launch_rows<4, true>(a, s) : launch_rows<4, false>(a, s);
}

}  // namespace tamd
