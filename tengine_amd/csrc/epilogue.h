// Bit-exact requantising epilogues shared by every int8 kernel (the parity spec).
// Every float operation below is a single correctly-rounded binary32 op in the reference's written
// order; the translation unit is compiled with -ffp-contract=off and without fast-math, division is
// the IEEE sequence (__fdiv_rn), rounding is C round() == half away from zero (roundf).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tamd {

enum RequantMode {
    RQ_CONV_HCL = 0,  // A1: conv/x86/conv_kernel_x86.c:1826-1889 == conv_dw_hcl_x86.c:197-261,373-436
    RQ_CONV_REF = 1,  // A2: conv/conv_kernel_ref_int8.c:72-78,137-167
    RQ_FC = 2,        // A5: fc/fc_ref.c:224-225,252-257
};

__device__ __forceinline__ int sat127(int v) { return v > 127 ? 127 : (v < -127 ? -127 : v); }

// round-half-away-from-zero of a binary32 value, then int conversion + clamp [-127,127]
__device__ __forceinline__ int round_sat(float q) { return sat127((int)roundf(q)); }

// acc already includes the int32 bias (the reference adds bias in int32 before converting)
__device__ __forceinline__ int requant(int acc, float in_scale, float w_scale, float out_scale, int act, int mode)
{
    float f;
    if (mode == RQ_CONV_HCL) {
        f = __fmul_rn((float)acc, in_scale);
        f = __fmul_rn(f, w_scale);
        if (act == 0) f = f < 0.f ? 0.f : f;
        if (act > 0) { f = f < 0.f ? 0.f : f; f = f > 6.f ? 6.f : f; }
        return round_sat(__fdiv_rn(f, out_scale));
    } else if (mode == RQ_CONV_REF) {
        float d = __fmul_rn(in_scale, w_scale);
        f = __fmul_rn((float)acc, d);
        if (act >= 0) {
            if (f < 0.f && act != 1) f = 0.f;
            if (f > 1.f && act == 1) f = 1.f;
            if (f > 6.f && act == 6) f = 6.f;
            if (f < -1.f && act == 1) f = -1.f;
        }
        return round_sat(__fdiv_rn(f, out_scale));
    } else {
        float r = __fdiv_rn(__fmul_rn(in_scale, w_scale), out_scale);
        return round_sat(__fmul_rn((float)acc, r));
    }
}

__device__ __forceinline__ unsigned pack4(int a, int b, int c, int d)
{
    return (unsigned)(a & 0xff) | ((unsigned)(b & 0xff) << 8) | ((unsigned)(c & 0xff) << 16) | ((unsigned)(d & 0xff) << 24);
}

}  // namespace tamd
