// Generic direct int8 convolution: any group / channel count / kernel, input either the graph's NCHW
// tensor (first layer: no separate layout pass) or NHWC.  Used for (a) first layers with a tiny
// input-channel count (MobileNet conv1 3x3 s2 C=3, ResNet conv1 7x7 s2 C=3) where K = KH*KW*3 is
// too ragged for 16-byte MFMA granules, and (b) as the always-correct fallback for grouped /
// odd-channel convolutions (the role conv_ref.c:43 plays on the CPU: score 4000, "can do").
//
// Reference arithmetic: identical exact int32 sum as ref_conv_int8 (conv/conv_kernel_ref_int8.c:86-136)
// and im2col+sgemm_i8 (conv_kernel_x86.c:187-242,1008-1630); epilogue chosen by the planner.
//
// Mapping: blockIdx.y = group of 4 consecutive output channels (uniform per block -> weight bytes
// come through the scalar cache), lanes run along output pixels (ox fastest) so NCHW input reads are
// coalesced; each lane produces one packed dword (4 channels) of the NHWC output.
#include "epilogue.h"
#include "kernels.h"

namespace tamd {

__global__ __launch_bounds__(256) void conv_direct_i8_kernel(DirectArgs a)
{
    const int cout_g = a.cout / a.group, cin_g = a.C / a.group;
    const int co0 = blockIdx.y * 4;                 // first of 4 output channels (may cross a group edge)
    const long M = (long)a.N * a.OH * a.OW;
    const long m = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const int ohw = a.OH * a.OW;
    const int n = (int)(m / ohw);
    const int rem = (int)(m - (long)n * ohw);
    const int oy = rem / a.OW, ox = rem - oy * a.OW;

    int acc[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int co = co0 + j;
        if (co >= a.cout) continue;
        const int g = co / cout_g;
        const int8_t* wk = a.w + (size_t)co * cin_g * a.KH * a.KW;
        int s = 0;
        for (int kc = 0; kc < cin_g; kc++) {
            const int c = g * cin_g + kc;
            for (int ky = 0; ky < a.KH; ky++) {
                const int iy = oy * a.SH - a.PH + ky * a.DH;
                if (iy < 0 || iy >= a.H) continue;
                for (int kx = 0; kx < a.KW; kx++) {
                    const int ix = ox * a.SW - a.PW + kx * a.DW;
                    if (ix < 0 || ix >= a.W) continue;
                    const int8_t xv = a.cs_in == 0 ? a.x[(((size_t)n * a.C + c) * a.H + iy) * a.W + ix]
                                                   : a.x[(((size_t)n * a.H + iy) * a.W + ix) * a.cs_in + c];
                    s += (int)xv * (int)wk[(kc * a.KH + ky) * a.KW + kx];
                }
            }
        }
        acc[j] = s;
    }
    const Rq rq = a.rq;
    int q[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int co = co0 + j;
        if (co < a.cout) {
            const int b = a.bias ? a.bias[co] : 0;
            q[j] = requant1(acc[j] + b, a.wscale[co], co, rq);
        } else {
            q[j] = 0;
        }
    }
    *reinterpret_cast<unsigned*>(a.y + (size_t)m * a.ldc + a.c_off + co0) = pack4(q[0], q[1], q[2], q[3]);
}

hipError_t launch_conv_direct(const DirectArgs& a, hipStream_t s)
{
    const long M = (long)a.N * a.OH * a.OW;
    // channel groups cover the padded channel stride region this conv owns (zeros beyond cout)
    const int cgroups = (a.cout + 3) / 4;
    dim3 grid((unsigned)((M + 255) / 256), (unsigned)cgroups);
    hipLaunchKernelGGL(conv_direct_i8_kernel, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace tamd
