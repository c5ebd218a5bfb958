// Winograd F(2x2, 3x3) convolution for fp32 models (3x3, stride 1, dilation 1, group 1) -- SURVEY §8 row W.
//
// The reference's CPU backend runs its fp32 3x3 stride-1 convolutions through a Winograd transform
// (conv/x86/wino_conv_kernel_x86.c: F(4x4, 3x3), transforms at :126-700, the batched product at :700-1100): the same
// mathematical result as the direct convolution with 4x (there) / 2.25x (here) fewer multiplications.  The device version
// uses the smaller F(2,3) tiles -- all transform constants are 0, +-1, +-1/2, exact in binary32, and the rounding error
// stays a few ulp (the parity bar for fp32 is 1e-4) -- and puts the multiplications on the matrix cores:
//
//   1. wino_in_f32    V[xi][c][t]  = (B^T d B)[xi]      d = the 4x4 input patch of tile t = (n, ty, tx), zero padded,
//                                                        xi = 4*i + j the position inside the transformed tile
//   2. wino_gemm_f32  M[xi][co][t] = sum_c U[xi][co][c] * V[xi][c][t]     16 independent [cout] x [tiles] x [cin] GEMMs in
//                                                        one launch (grid.z = xi) on v_mfma_f32_32x32x2f32; U = G g G^T is
//                                                        computed on the host at prerun
//   3. wino_out_f32   Y = A^T M A + bias, activation, 2x2 outputs per tile stored to the NCHW output (ragged edges clipped,
//                                                        concat-by-offset placement like every other fp32 kernel)
//
// V and M are padded to multiples of the GEMM tile (channels to 16 / 64, tiles to 64) and the padding is written as zeros,
// so the GEMM has no edge predicates.  Layouts keep the TILE index fastest: transforms and GEMM operand loads are coalesced.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace tamd {

typedef float v16f __attribute__((ext_vector_type(16)));

// ---- 1. input transform -------------------------------------------------------------------------------------------
// B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
__global__ __launch_bounds__(256) void wino_in_f32_k(const F32WinoArgs a)
{
    const int t = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
    if (t >= a.Tpad) return;
    float d[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) d[i][j] = 0.f;
    if (t < a.T && c < a.C) {
        const int tpi = a.TH * a.TW, n = t / tpi, r = t - n * tpi, ty = r / a.TW, tx = r - ty * a.TW;
        const int iy0 = 2 * ty - a.PH, ix0 = 2 * tx - a.PW;
        const float* xc = a.x + ((size_t)n * a.C + c) * a.H * a.W;
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int iy = iy0 + i, ix = ix0 + j;
                if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) d[i][j] = xc[iy * a.W + ix];
            }
    }
    float w[4][4];              // B^T d
#pragma unroll
    for (int j = 0; j < 4; j++) {
        w[0][j] = d[0][j] - d[2][j];
        w[1][j] = d[1][j] + d[2][j];
        w[2][j] = d[2][j] - d[1][j];
        w[3][j] = d[1][j] - d[3][j];
    }
    float* v = a.V + (size_t)c * a.Tpad + t;
    const size_t plane = (size_t)a.Cpad * a.Tpad;
#pragma unroll
    for (int i = 0; i < 4; i++) {       // (B^T d) B
        v[(4 * i + 0) * plane] = w[i][0] - w[i][2];
        v[(4 * i + 1) * plane] = w[i][1] + w[i][2];
        v[(4 * i + 2) * plane] = w[i][2] - w[i][1];
        v[(4 * i + 3) * plane] = w[i][1] - w[i][3];
    }
}

// ---- 2. the 16 GEMMs ---------------------------------------------------------------------------------------------
// block = 64 output channels x 64 tiles of one xi, four waves of 32 x 32; K = cin in steps of 16 through LDS.
// v_mfma_f32_32x32x2f32: A lane l = U[row l%32][k0 + l/32], B lane l = V[k0 + l/32][col l%32],
// D register r of lane l = M[row (r/4)*8 + (l/32)*4 + r%4][col l%32].
__global__ __launch_bounds__(256) void wino_gemm_f32_k(const F32WinoArgs a)
{
    constexpr int KC = 16, LDA = KC + 1, LDB = 64 + 4;
    __shared__ float As[64 * LDA];      // [co][k]
    __shared__ float Bs[KC * LDB];      // [k][t]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave & 1, wn = wave >> 1;
    const int t0 = blockIdx.x * 64, co0 = blockIdx.y * 64, xi = blockIdx.z;
    const float* U = a.U + ((size_t)xi * a.Mpad + co0) * a.Cpad;
    const float* V = a.V + (size_t)xi * a.Cpad * a.Tpad + t0;
    v16f acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    // staging roles: A tile 64 x 16 = 256 float4 along k; B tile 16 x 64 = 256 float4 along t
    const int ar = tid >> 2, ak = (tid & 3) * 4, bk = tid >> 4, bt = (tid & 15) * 4;
    for (int k0 = 0; k0 < a.Cpad; k0 += KC) {
        const float4 av = *reinterpret_cast<const float4*>(U + (size_t)ar * a.Cpad + k0 + ak);
        const float4 bv = *reinterpret_cast<const float4*>(V + (size_t)(k0 + bk) * a.Tpad + bt);
        __syncthreads();                 // the previous step's fragment reads are done
        As[ar * LDA + ak] = av.x; As[ar * LDA + ak + 1] = av.y; As[ar * LDA + ak + 2] = av.z; As[ar * LDA + ak + 3] = av.w;
        *reinterpret_cast<float4*>(Bs + bk * LDB + bt) = bv;
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < KC; kk += 2) {
            const float fa = As[(wm * 32 + (lane & 31)) * LDA + kk + (lane >> 5)];
            const float fb = Bs[(kk + (lane >> 5)) * LDB + wn * 32 + (lane & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc, 0, 0, 0);
        }
    }
    float* M = a.M + ((size_t)xi * a.Mpad + co0 + wm * 32) * a.Tpad + t0 + wn * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; r++) M[(size_t)((r >> 2) * 8 + (lane >> 5) * 4 + (r & 3)) * a.Tpad] = acc[r];
}

// ---- 3. output transform ---------------------------------------------------------------------------------------------
// A^T = [1 1 1 0; 0 1 -1 -1]
__global__ __launch_bounds__(256) void wino_out_f32_k(const F32WinoArgs a)
{
    const int t = blockIdx.x * 256 + threadIdx.x, co = blockIdx.y;
    if (t >= a.T) return;
    const float* m = a.M + (size_t)co * a.Tpad + t;
    const size_t plane = (size_t)a.Mpad * a.Tpad;
    float s[2][4];                      // A^T M
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const float m0 = m[(0 + j) * plane], m1 = m[(4 + j) * plane], m2 = m[(8 + j) * plane], m3 = m[(12 + j) * plane];
        s[0][j] = m0 + m1 + m2;
        s[1][j] = m1 - m2 - m3;
    }
    const float bias = a.bias ? a.bias[co] : 0.f;
    const int tpi = a.TH * a.TW, n = t / tpi, r = t - n * tpi, ty = r / a.TW, tx = r - ty * a.TW;
    float* yo = a.y + (size_t)n * a.out_img + (size_t)(a.out_c0 + co) * a.OH * a.OW;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        float y2[2] = {s[i][0] + s[i][1] + s[i][2], s[i][1] - s[i][2] - s[i][3]};
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int oy = 2 * ty + i, ox = 2 * tx + j;
            float v = y2[j] + bias;
            if (a.act == 0) v = v < 0.f ? 0.f : v;
            if (a.act > 0) { v = v < 0.f ? 0.f : v; v = v > 6.f ? 6.f : v; }       // conv_kernel_x86.c:1666-1690: any positive code clamps to [0, 6]
            if (oy < a.OH && ox < a.OW) yo[oy * a.OW + ox] = v;
        }
    }
}

hipError_t launch_wino_in_f32(const F32WinoArgs& a, hipStream_t s)
{
    hipLaunchKernelGGL(wino_in_f32_k, dim3(a.Tpad / 256 + (a.Tpad % 256 ? 1 : 0), a.Cpad), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_wino_gemm_f32(const F32WinoArgs& a, hipStream_t s)
{
    hipLaunchKernelGGL(wino_gemm_f32_k, dim3(a.Tpad / 64, a.Mpad / 64, 16), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_wino_out_f32(const F32WinoArgs& a, hipStream_t s)
{
    hipLaunchKernelGGL(wino_out_f32_k, dim3((a.T + 255) / 256, a.cout), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace tamd
