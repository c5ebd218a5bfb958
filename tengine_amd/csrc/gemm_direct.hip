// Latency-shaped int8 1x1 convolution / FC on MFMA for small pixel counts (batch-1 MobileNet tails,
// classifier layers): same arithmetic and weight packing as conv_igemm.hip, different schedule.
//
// At batch 1 the late MobileNet layers are 49..196 pixels: a K-loop with a barrier and an LDS round trip
// per 64-deep step is a serial chain of memory latencies (measured 7-10 us per layer), while the data is a
// few hundred KB that sits in L2/MALL.  Here every wave loads its MFMA operand fragments straight from
// global memory into registers -- NHWC activations and [cout][k] weights are both K-contiguous, so a
// fragment is one 16-B load per lane -- with ALL loads of its K slice in flight before the first MFMA,
// and the K dimension is split over the block's waves (partials reduced through LDS once).
// One memory round trip + a handful of MFMAs + the fused requantising epilogue.
#include <stdlib.h>
#include "env.h"

#include "epilogue.h"
#include "kernels.h"

namespace tamd {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// block = 4 waves = KSPLIT k-slices x NT (=4/KSPLIT) cout tiles of 32; pixel tile 32
template <int KSPLIT>
__global__ __launch_bounds__(256) void gemm_direct_i8_kernel(ConvArgs a)
{
    constexpr int NT = 4 / KSPLIT;
    constexpr int U = 8;                                  // k-steps (of 32) in flight per wave
    __shared__ int red[KSPLIT > 1 ? (KSPLIT - 1) * NT * 16 * 64 : 1];

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int ks = wave % KSPLIT, nt = wave / KSPLIT;
    const int tiles_n = (a.cout + 32 * NT - 1) / (32 * NT);
    const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x % tiles_n;
    const int m0 = tile_m * 32, n0 = (tile_n * NT + nt) * 32;
    const bool nvalid = n0 < a.cout;                      // wave-uniform

    const int S = a.kpad / 32;                            // total k-steps
    const int per = (S + KSPLIT - 1) / KSPLIT;
    const int s_begin = ks * per, s_end = (s_begin + per < S) ? s_begin + per : S;

    const int m = m0 + l31;
    const bool mvalid = m < a.M;
    const int8_t* wp = a.w + (size_t)(nvalid ? n0 + l31 : 0) * a.kpad + hi * 16;
    const int8_t* xp = a.x + (size_t)(mvalid ? m : 0) * a.cs_in + hi * 16;

    v16i acc;
#pragma unroll
    for (int e = 0; e < 16; e++) acc[e] = 0;

    if (nvalid) {
        for (int sb = s_begin; sb < s_end; sb += U) {
            v4i af[U], bf[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int s = sb + u;
                const bool sv = s < s_end;
                const v4i z = {0, 0, 0, 0};
                af[u] = sv ? *reinterpret_cast<const v4i*>(wp + (size_t)s * 32) : z;
                const bool kv = sv && mvalid && (s * 32 + hi * 16) < a.ktot;
                bf[u] = kv ? *reinterpret_cast<const v4i*>(xp + (size_t)s * 32) : z;
            }
#pragma unroll
            for (int u = 0; u < U; u++)
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[u], bf[u], acc, 0, 0, 0);
        }
    }

    if (KSPLIT > 1) {
        if (ks > 0) {
#pragma unroll
            for (int e = 0; e < 16; e++) red[(((ks - 1) * NT + nt) * 16 + e) * 64 + lane] = acc[e];
        }
        __syncthreads();
        if (ks > 0) return;
#pragma unroll
        for (int k2 = 0; k2 < KSPLIT - 1; k2++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[e] += red[((k2 * NT + nt) * 16 + e) * 64 + lane];
    }
    if (!nvalid) return;

    const Rq rq = a.rq;
#pragma unroll
    for (int g4 = 0; g4 < 4; g4++) {
        const int c0 = n0 + 8 * g4 + 4 * hi;
        const int4 b4 = *reinterpret_cast<const int4*>(a.bias + c0);
        const float4 s4 = *reinterpret_cast<const float4*>(a.wscale + c0);
        const unsigned p = requant4(acc[4 * g4 + 0] + b4.x, acc[4 * g4 + 1] + b4.y, acc[4 * g4 + 2] + b4.z,
                                    acc[4 * g4 + 3] + b4.w, s4, c0, rq);
        if (mvalid && c0 < a.c_limit) *reinterpret_cast<unsigned*>(a.y + (size_t)m * a.ldc + a.c_off + c0) = p;
    }
}

static int direct_max_m()
{
    static int v = -1;
    if (v < 0) {
        const char* e = exp_env("TAMD_DIRECT_MAX_M");
        v = e ? atoi(e) : 1024;
    }
    return v;
}

bool gemm_direct_applicable(const ConvArgs& a)
{
    const bool is1x1 = (a.KH == 1 && a.KW == 1 && a.SH == 1 && a.SW == 1 && a.PH == 0 && a.PW == 0);
    return is1x1 && a.M <= direct_max_m();
}

hipError_t launch_gemm_direct(const ConvArgs& a, hipStream_t s)
{
    const int S = a.kpad / 32;
    const int tiles_m = (a.M + 31) / 32;
    if (S >= 8) {
        const int tiles_n = (a.cout + 31) / 32;
        hipLaunchKernelGGL(gemm_direct_i8_kernel<4>, dim3(tiles_m * tiles_n), dim3(256), 0, s, a);
    } else if (S >= 4) {
        const int tiles_n = (a.cout + 63) / 64;
        hipLaunchKernelGGL(gemm_direct_i8_kernel<2>, dim3(tiles_m * tiles_n), dim3(256), 0, s, a);
    } else {
        const int tiles_n = (a.cout + 127) / 128;
        hipLaunchKernelGGL(gemm_direct_i8_kernel<1>, dim3(tiles_m * tiles_n), dim3(256), 0, s, a);
    }
    return hipGetLastError();
}

}  // namespace tamd
