// Streaming int8 pointwise (1x1) convolution on MFMA for shallow K (cin <= 128) and many pixels --
// the early MobileNet / SSD / SqueezeNet pointwise layers, where an LDS-tiled GEMM spends its time on
// per-block set-up (one or two K steps per barrier) instead of on memory.
//
// Same arithmetic and weight packing as conv_igemm.hip (reference chain: conv_kernel_x86.c:187-242,
// :963-1007, :1008-1630, :1796-1893).  Schedule: the whole [cout_tile][K] weight panel of a block lives in
// registers as MFMA A-fragments for the lifetime of the block; every wave then walks 32-pixel tiles of the
// NHWC activation stream -- one 16-B load per lane per 32-deep K step, the 32 x K bytes of a tile are one
// contiguous segment -- multiplies, requantises (bit-exact epilogue.h) and stores.  No LDS staging of
// operands, no barrier in the loop.  The epilogue re-distributes the packed results between the two
// half-waves with v_permlane32_swap so that every lane stores 16 contiguous output channels (one
// dwordx4 instead of four dword stores into 4 different 64-B segments).
#include <stdlib.h>
#include "env.h"

#include "epilogue.h"
#include "kernels.h"

namespace tamd {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// ELT: the eltwise (+ReLU) node that consumes this conv is applied in the epilogue (epilogue.h: fuse_elt4), as in conv_igemm
// WIN: the one-binade requantisation of epilogue.h -- of the conv's own window (ELT false) or of the residual tail's (ELT true; the conv in
// front of an eltwise node keeps the general form); launch_pw checks the node's constants.  A kernel of its own and not a branch at the
// top of one: with both bodies in one kernel <2,4> needs 106 registers instead of 90 / 92 and loses a wave per SIMD.
template <int S, int NT, bool ELT, int WIN>
__global__ __launch_bounds__(256) void pw_stream_i8_kernel(ConvArgs a)
{
    __shared__ __attribute__((aligned(16))) int sbias[NT * 32];
    __shared__ __attribute__((aligned(16))) float sscale[NT * 32];
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int n0 = blockIdx.y * NT * 32;
    if (t < NT * 32) { sbias[t] = a.bias[n0 + t]; sscale[t] = a.wscale[n0 + t]; }

    v4i af[NT][S];
#pragma unroll
    for (int i = 0; i < NT; i++)
#pragma unroll
        for (int s = 0; s < S; s++)
            af[i][s] = *reinterpret_cast<const v4i*>(a.w + (size_t)(n0 + i * 32 + l31) * a.kpad + s * 32 + hi * 16);
    __syncthreads();

    const Rq rq = a.rq;
    const float inv_elt = ELT ? __fdiv_rn(1.0f, a.elt.out_scale) : 1.f;
    const float inv_relu = (ELT && a.elt.relu) ? __fdiv_rn(1.0f, a.elt.relu_out_scale) : 1.f;
    const int tiles_m = (a.M + 31) / 32;
    for (int tile = blockIdx.x * 4 + wave; tile < tiles_m; tile += gridDim.x * 4) {
        const int m = tile * 32 + l31;
        const bool mvalid = m < a.M;
        const int8_t* xp = a.x + (size_t)(mvalid ? m : 0) * a.cs_in + hi * 16;
        v4i bf[S];
#pragma unroll
        for (int s = 0; s < S; s++) {
            const v4i z = {0, 0, 0, 0};
            bf[s] = (mvalid && (s * 32 + hi * 16) < a.ktot) ? *reinterpret_cast<const v4i*>(xp + s * 32) : z;
        }
#pragma unroll
        for (int i = 0; i < NT; i++) {
            v16i acc;                  // starts at the bias: the LDS read lands in the accumulator, the epilogue adds nothing
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) {
                int c = i * 32 + 8 * g4 + 4 * hi;
                // keep the (loop-invariant) bias/scale reads inside the tile loop: hoisted, they cost
                // 32*NT VGPRs and drop the kernel to one wave per SIMD
                asm volatile("" : "+v"(c));
                const int4 b4 = *reinterpret_cast<const int4*>(&sbias[c]);
                acc[4 * g4 + 0] = b4.x; acc[4 * g4 + 1] = b4.y; acc[4 * g4 + 2] = b4.z; acc[4 * g4 + 3] = b4.w;
            }
#pragma unroll
            for (int s = 0; s < S; s++) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[i][s], bf[s], acc, 0, 0, 0);
            unsigned p[4];
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) {
                int c = i * 32 + 8 * g4 + 4 * hi;
                asm volatile("" : "+v"(c));
                const float4 s4 = *reinterpret_cast<const float4*>(&sscale[c]);
                p[g4] = requant4<ELT ? 0 : WIN>(acc[4 * g4 + 0], acc[4 * g4 + 1], acc[4 * g4 + 2], acc[4 * g4 + 3], s4, n0 + c, rq);
            }
            half_wave_regroup(p);
            const int cb = n0 + i * 32 + hi * 16;
            if (mvalid && cb < a.c_limit) {
                if (ELT) {         // residual operand: the same pixel, the 16 channels this lane now holds
                    const uint4 r = *reinterpret_cast<const uint4*>(a.elt.res + (size_t)m * a.elt.res_ldc + a.elt.res_c_off + cb);
                    if (a.elt.thr > 0.f) {
                        elt_sum16_fold<WIN>(p, r, a.elt);
                    } else {
                        const uint4 o = fuse_elt16(make_uint4(p[0], p[1], p[2], p[3]), r, a.elt, inv_elt, inv_relu);
                        p[0] = o.x; p[1] = o.y; p[2] = o.z; p[3] = o.w;
                    }
                }
                *reinterpret_cast<uint4*>(a.y + (size_t)m * a.ldc + a.c_off + cb) = make_uint4(p[0], p[1], p[2], p[3]);
            }
        }
    }
}

// (k steps, cout tiles per block) combinations instantiated; weights padded to 128 couts by the planner
bool pw_stream_applicable(const ConvArgs& a)
{
    const bool is1x1 = (a.KH == 1 && a.KW == 1 && a.SH == 1 && a.SW == 1 && a.PH == 0 && a.PW == 0);
    const int S = a.kpad / 32;
    // 16-B stores need 16-channel granularity of the destination (always true for non-view outputs)
    if (a.elt.res && ((a.elt.res_ldc | a.elt.res_c_off) & 15)) return false;      // 16-B reads of the residual operand
    return is1x1 && S <= 4 && a.M >= 2048 && (a.c_limit % 16 == 0) && (a.c_off % 16 == 0) && (a.ldc % 16 == 0);
}

template <int S, int NT>
static hipError_t launch_pw(const ConvArgs& a, hipStream_t s)
{
    const int tiles_m = (a.M + 31) / 32;
    const int groups = (a.cout + NT * 32 - 1) / (NT * 32);
    int bx = (tiles_m + 3) / 4;
    // blocks of a launch (the waves stride over the pixel tiles beyond it).  TAMD_PW_STREAM_BLOCKS: experiments only -- a copy kernel
    // with this layer's read : write mix ran 15.3 us from 512 blocks against 20.9 us from 2048 (profiles/r02_write_bw_access_shapes.txt)
    const char* be = exp_env("TAMD_PW_STREAM_BLOCKS");
    const int total = be && atoi(be) > 0 ? atoi(be) : 2048;
    const int cap = total / groups > 0 ? total / groups : 1;
    if (bx > cap) bx = cap;
    if (a.elt.res) {
        if (a.elt.thr > 0.f && elt_win(a.elt)) hipLaunchKernelGGL((pw_stream_i8_kernel<S, NT, true, 1>), dim3(bx, groups), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((pw_stream_i8_kernel<S, NT, true, 0>), dim3(bx, groups), dim3(256), 0, s, a);
    } else if (rq_win(a.rq)) hipLaunchKernelGGL((pw_stream_i8_kernel<S, NT, false, 1>), dim3(bx, groups), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((pw_stream_i8_kernel<S, NT, false, 0>), dim3(bx, groups), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_pw_stream(const ConvArgs& a, hipStream_t s)
{
    const int S = a.kpad / 32;
    const int ct = (a.cout + 31) / 32;
    if (S <= 2) {      // kpad is a multiple of 64 -> S is 2 or 4
        if (ct <= 2) return launch_pw<2, 2>(a, s);
        return launch_pw<2, 4>(a, s);
    }
    if (ct <= 2) return launch_pw<4, 2>(a, s);
    return launch_pw<4, 4>(a, s);
}

}  // namespace tamd
