// Shared by the depthwise kernels (dwconv.hip, pwdw.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace tamd {

// 4x4 byte transpose: d[p] = pixel p's channels {c0,c1,c2,c3}  ->  x[c] = channel c at pixels {p0,p1,p2,p3}
// v_perm_b32 D = bytes of {S0(hi):S1(lo)} picked by the selector (0-3 -> S1, 4-7 -> S0)
__device__ __forceinline__ void transpose4x4(const unsigned (&d)[4], unsigned (&x)[4])
{
    const unsigned t0 = __builtin_amdgcn_perm(d[1], d[0], 0x05010400u);   // {d0.c0, d1.c0, d0.c1, d1.c1}
    const unsigned t1 = __builtin_amdgcn_perm(d[1], d[0], 0x07030602u);   // {d0.c2, d1.c2, d0.c3, d1.c3}
    const unsigned t2 = __builtin_amdgcn_perm(d[3], d[2], 0x05010400u);   // {d2.c0, d3.c0, d2.c1, d3.c1}
    const unsigned t3 = __builtin_amdgcn_perm(d[3], d[2], 0x07030602u);
    x[0] = __builtin_amdgcn_perm(t2, t0, 0x05040100u);                    // {t0.b0, t0.b1, t2.b0, t2.b1}
    x[1] = __builtin_amdgcn_perm(t2, t0, 0x07060302u);
    x[2] = __builtin_amdgcn_perm(t3, t1, 0x05040100u);
    x[3] = __builtin_amdgcn_perm(t3, t1, 0x07060302u);
}

}  // namespace tamd
