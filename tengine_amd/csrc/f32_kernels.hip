// fp32 glue kernels (SURVEY §8 a10/a12, fp32 models: tolerance 1e-4, order-free).  Dense NCHW fp32 tensors, one
// thread per output element, lanes along the innermost (pixel) axis.  The group==1 convolutions and FC run on the
// matrix cores (conv_f32_mfma.hip); everything here is bandwidth-shaped.
//   conv (grouped / depthwise)  conv/x86/conv_dw_kernel_x86.c, conv_kernel_ref_fp32 semantics: + bias, relu / relu6
//   pooling                     pooling/pooling_kernel_ref_fp32.c (max / avg, caffe_flavor window rule)
//   relu / leaky / relu6, concat slice copy, nearest upsample, eltwise, softmax (softmax/softmax_ref.c)
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace tamd {

__global__ __launch_bounds__(256) void conv_f32_direct_k(const F32DirectArgs a)
{
    const int OHW = a.OH * a.OW;
    const int pj = blockIdx.x * 256 + threadIdx.x;
    const int oc = blockIdx.y, n = blockIdx.z;
    if (pj >= OHW) return;
    const int cin_g = a.C / a.group, cout_g = a.cout / a.group, g = oc / cout_g;
    const int oy = pj / a.OW, ox = pj - oy * a.OW;
    const float* wk = a.w + (size_t)oc * cin_g * a.KH * a.KW;
    float total = 0.f;
    for (int kc = 0; kc < cin_g; kc++) {
        const float* xc = a.x + ((size_t)n * a.C + (size_t)g * cin_g + kc) * a.H * a.W;
        for (int ky = 0; ky < a.KH; ky++) {
            const int iy = oy * a.SH - a.PH + ky * a.DH;
            if ((unsigned)iy >= (unsigned)a.H) continue;
            for (int kx = 0; kx < a.KW; kx++) {
                const int ix = ox * a.SW - a.PW + kx * a.DW;
                if ((unsigned)ix >= (unsigned)a.W) continue;
                total = __builtin_fmaf(xc[iy * a.W + ix], wk[(kc * a.KH + ky) * a.KW + kx], total);
            }
        }
    }
    if (a.bias) total = total + a.bias[oc];
    if (a.act == 0) total = total < 0.f ? 0.f : total;
    if (a.act > 0) { total = total < 0.f ? 0.f : total; total = total > 6.f ? 6.f : total; }
    a.y[(size_t)n * a.out_img + (size_t)(a.out_c0 + oc) * OHW + pj] = total;
}

hipError_t launch_conv_f32_direct(const F32DirectArgs& a, hipStream_t s)
{
    dim3 grid((a.OH * a.OW + 255) / 256, a.cout, a.N);
    hipLaunchKernelGGL(conv_f32_direct_k, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void pool_f32_k(const F32PoolArgs a)
{
    const int OHW = a.OH * a.OW;
    const int pj = blockIdx.x * 256 + threadIdx.x;
    if (pj >= OHW) return;
    const int ch = blockIdx.y, n = blockIdx.z;
    const int py = pj / a.OW, px = pj - py * a.OW;
    const float* xc = a.x + ((size_t)n * a.C + ch) * a.H * a.W;
    int hs = py * a.SH - a.PH, he = min(hs + a.KH, a.H + a.PH);
    int ws_ = px * a.SW - a.PW, we = min(ws_ + a.KW, a.W + a.PW);
    int pool_size = 1;
    if (a.caffe_flavor) pool_size = (he - hs) * (we - ws_);
    hs = max(hs, 0); ws_ = max(ws_, 0); he = min(he, a.H); we = min(we, a.W);
    if (!a.caffe_flavor) pool_size = (he - hs) * (we - ws_);
    float f;
    if (a.method == 0) {
        f = xc[hs * a.W + ws_];
        for (int iy = hs; iy < he; iy++)
            for (int ix = ws_; ix < we; ix++) f = fmaxf(f, xc[iy * a.W + ix]);
    } else {
        float sum = 0.f;
        for (int iy = hs; iy < he; iy++)
            for (int ix = ws_; ix < we; ix++) sum += xc[iy * a.W + ix];
        f = sum / (float)pool_size;
    }
    a.y[((size_t)n * a.C + ch) * OHW + pj] = f;
}

hipError_t launch_pool_f32(const F32PoolArgs& a, hipStream_t s)
{
    dim3 grid((a.OH * a.OW + 255) / 256, a.C, a.N);
    hipLaunchKernelGGL(pool_f32_k, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

// MODE 0 relu / leaky (slope), 1 copy into a channel slice (concat), 2 nearest upsample, 3 relu6
template <int MODE>
__global__ __launch_bounds__(256) void map_f32_k(const F32MapArgs a)
{
    const int OW = a.W * a.scale, OHW = a.H * a.scale * OW;
    const int pj = blockIdx.x * 256 + threadIdx.x;
    if (pj >= OHW) return;
    const int ch = blockIdx.y, n = blockIdx.z;
    const float* xc = a.x + ((size_t)n * a.C + ch) * a.H * a.W;
    float* yo = a.y + (size_t)n * a.out_img + (size_t)(a.out_c0 + ch) * OHW + pj;
    if (MODE == 0) {
        const float f = xc[pj];
        *yo = f < 0.f ? f * a.slope : f;
    } else if (MODE == 1)
        *yo = xc[pj];
    else if (MODE == 2) {
        const int oy = pj / OW, ox = pj - oy * OW;
        *yo = xc[(oy / a.scale) * a.W + ox / a.scale];
    } else
        *yo = fminf(fmaxf(xc[pj], 0.f), 6.f);
}

hipError_t launch_map_f32(const F32MapArgs& a, int mode, hipStream_t s)
{
    dim3 grid((a.H * a.scale * a.W * a.scale + 255) / 256, a.C, a.N);
    switch (mode) {
    case 0: hipLaunchKernelGGL(map_f32_k<0>, grid, dim3(256), 0, s, a); break;
    case 1: hipLaunchKernelGGL(map_f32_k<1>, grid, dim3(256), 0, s, a); break;
    case 2: hipLaunchKernelGGL(map_f32_k<2>, grid, dim3(256), 0, s, a); break;
    default: hipLaunchKernelGGL(map_f32_k<3>, grid, dim3(256), 0, s, a); break;
    }
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void eltwise_f32_k(const float* a, const float* b, float* y, size_t count, int type)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const float fa = a[i], fb = b[i];
    y[i] = type == 0 ? fa * fb : type == 2 ? fa + fb : type == 4 ? fa - fb : fmaxf(fa, fb);
}

hipError_t launch_eltwise_f32(const float* a, const float* b, float* y, size_t count, int type, hipStream_t s)
{
    hipLaunchKernelGGL(eltwise_f32_k, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, a, b, y, count, type);
    return hipGetLastError();
}

// softmax over the channel axis of [N][C][inner]: one thread per (n, inner) position
__global__ __launch_bounds__(64) void softmax_f32_k(const float* x, float* y, int N, int C, int inner)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= N * inner) return;
    const int n = i / inner, p = i - n * inner;
    const float* xc = x + (size_t)n * C * inner + p;
    float* yc = y + (size_t)n * C * inner + p;
    float m = xc[0];
    for (int c = 1; c < C; c++) m = fmaxf(m, xc[(size_t)c * inner]);
    float sum = 0.f;
    for (int c = 0; c < C; c++) {
        const float e = expf(xc[(size_t)c * inner] - m);
        yc[(size_t)c * inner] = e;
        sum += e;
    }
    for (int c = 0; c < C; c++) yc[(size_t)c * inner] = yc[(size_t)c * inner] / sum;
}

hipError_t launch_softmax_f32(const float* x, float* y, int N, int C, int inner, hipStream_t s)
{
    hipLaunchKernelGGL(softmax_f32_k, dim3((N * inner + 63) / 64), dim3(64), 0, s, x, y, N, C, inner);
    return hipGetLastError();
}

}  // namespace tamd
