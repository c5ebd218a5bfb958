// Throughput of the requantising epilogue's instruction sequences with the SIMDs FULL of waves (8 per SIMD), not the lone-wave issue
// costs of valu_rates.hip: the general requant4 of epilogue.h against the one-binade forms (result byte = byte 2 of the bit pattern via
// v_perm_b32; hand-over flag from the low halves via v_min3_u16 / v_min_u16) and their two halves separately.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I include -I tengine_amd/csrc -o requant_rates.bin tools/exp/requant_rates.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include "epilogue.h"

using namespace tamd;
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)

// V 0: general form (cvt_u32 + shifts, fract + min3_u32) | 1: one-binade (perm pack, min3_u16 flag) | 2: perm pack + fract flag
//   3: cvt pack + min3_u16 flag | 4: no flag at all (pack by perm) -- the floor
template <int V>
__device__ __forceinline__ unsigned rq4(int a0, int a1, int a2, int a3, const float4& mf, int c, const Rq& r)
{
    const float y0 = rq_biased(a0, mf.x, r), y1 = rq_biased(a1, mf.y, r), y2 = rq_biased(a2, mf.z, r), y3 = rq_biased(a3, mf.w, r);
    unsigned p;
    bool hand;
    if (V == 1 || V == 2 || V == 4) p = rq_pack_byte2(y0, y1, y2, y3);
    else p = ((unsigned)y0 | ((unsigned)y1 << 8) | ((unsigned)y2 << 16) | ((unsigned)y3 << 24)) ^ 0x80808080u;
    if (V == 1 || V == 3) hand = rq_min_low_half(y0, y1, y2, y3) < 8u;
    else if (V == 4) hand = false;
    else {
        const unsigned f0 = __builtin_bit_cast(unsigned, __builtin_amdgcn_fractf(y0)), f1 = __builtin_bit_cast(unsigned, __builtin_amdgcn_fractf(y1));
        const unsigned f2 = __builtin_bit_cast(unsigned, __builtin_amdgcn_fractf(y2)), f3 = __builtin_bit_cast(unsigned, __builtin_amdgcn_fractf(y3));
        hand = min(min(f0, f1), min(f2, f3)) < __builtin_bit_cast(unsigned, r.thr);
    }
    if (hand) p = requant4_chain(a0, a1, a2, a3, mf, p, r.m2 + c, r.m1, r.lo, r.hi, r.out_scale, r.ylo, r.yhi, r.thr);
    return p;
}

template <int V>
__global__ __launch_bounds__(256) void k(unsigned* out, Rq r, float4 mf, int iters, int seed)
{
    int a0 = threadIdx.x * 7 + seed, a1 = a0 + 1111, a2 = a0 + 2222, a3 = a0 + 3333;
    unsigned x = 0;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            x ^= rq4<V>(a0, a1, a2, a3, mf, 0, r);
            a0 += 37; a1 += 41; a2 += 43; a3 += 47;          // 4 VALU of harness per 4 values, in every variant
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x;
}

int main()
{
    unsigned* d;
    const int blocks = 256 * 8;
    CK(hipMalloc(&d, blocks * 256 * 4));
    float* m2;
    CK(hipMalloc(&m2, 64));
    CK(hipMemset(m2, 0, 64));
    Rq r{};
    r.m1 = 1.f; r.lo = 0.f; r.hi = 1e30f; r.out_scale = 1.f; r.ylo = 128.25f; r.yhi = 255.75f; r.thr = 0x1p-13f; r.m2 = m2;
    const float4 mf = {0.00731f, 0.00653f, 0.00597f, 0.00811f};
    const int iters = 2000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[5] = {"general (cvt + shifts, fract + min3_u32)", "one binade (perm, min3_u16)", "perm pack, fract flag", "cvt pack, min3_u16 flag", "perm pack, no flag"};
    for (int rep = 0; rep < 2; rep++)
        for (int v = 0; v < 5; v++) {
            CK(hipEventRecord(e0));
            switch (v) {
            case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, r, mf, iters, 1); break;
            case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, r, mf, iters, 1); break;
            case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d, r, mf, iters, 1); break;
            case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, d, r, mf, iters, 1); break;
            default: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, d, r, mf, iters, 1); break;
            }
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double vals = (double)blocks * 256 * iters * 8 * 4;
            if (rep) printf("%-44s %8.3f ms  %6.2f Gvalues/s  %5.2f SIMD-cycles per value at 2.4 GHz (4 = one VALU instruction)\n", names[v], ms, vals / ms * 1e-6,
                            ms * 1e-3 * 2.4e9 * 1024 * 16 / vals / 4 * 4 / 4);
        }
    return 0;
}
