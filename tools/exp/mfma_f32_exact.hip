// Experiment: is the fp32 MFMA accumulation a chain of IEEE fused multiply-adds?  Compares, bit for bit,
//   (1) v_mfma_f32_32x32x1f32 issued K times          vs  s = fmaf(a[k], b[k], s), k ascending
//   (2) v_mfma_f32_32x32x2f32 issued K/2 times        vs  the same chain (k = 2i then 2i+1)
//   (3) v_mfma_f32_16x16x4f32 issued K/4 times        vs  the same chain
// build: hipcc --offload-arch=gfx950 -O2 -o mfma_f32_exact mfma_f32_exact.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v32f __attribute__((ext_vector_type(32)));
typedef float v4f __attribute__((ext_vector_type(4)));

// A: [M=64][K] row-major (two 32-row blocks), B: [K][N=64] (two 32-col blocks)
__global__ void k_x1(const float* A, const float* B, float* D, int K)   // 32x32x1, 2 blocks: block b = rows 32b.., cols 32b..
{
    const int lane = threadIdx.x;
    v32f acc;
    for (int i = 0; i < 32; i++) acc[i] = 0.f;
    for (int k = 0; k < K; k++) {
        const float a = A[(size_t)lane * K + k];         // lane -> row (lane%32) of block (lane/32): rows 0..63
        const float b = B[(size_t)k * 64 + lane];        // lane -> col (lane%32) of block (lane/32): cols 0..63
        acc = __builtin_amdgcn_mfma_f32_32x32x1f32(a, b, acc, 0, 0, 0);
    }
    // D layout per block: 16 regs, col = lane%32, row = (r%4) + 8*(r/4) + 4*(lane/32)
    for (int blk = 0; blk < 2; blk++)
        for (int r = 0; r < 16; r++) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31;
            D[((size_t)blk * 32 + row) * 32 + col] = acc[blk * 16 + r];
        }
}

__global__ void k_x2(const float* A, const float* B, float* D, int K)   // 32x32x2, 1 block: rows 0..31, cols 0..31
{
    const int lane = threadIdx.x;
    v16f acc;
    for (int i = 0; i < 16; i++) acc[i] = 0.f;
    for (int k = 0; k < K; k += 2) {
        const float a = A[(size_t)(lane & 31) * K + k + (lane >> 5)];
        const float b = B[(size_t)(k + (lane >> 5)) * 64 + (lane & 31)];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 16; r++) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31;
        D[(size_t)row * 32 + col] = acc[r];
    }
}

__global__ void k_x4(const float* A, const float* B, float* D, int K)   // 16x16x4, 1 block
{
    const int lane = threadIdx.x;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < K; k += 4) {
        const float a = A[(size_t)(lane & 15) * K + k + (lane >> 4)];
        const float b = B[(size_t)(k + (lane >> 4)) * 64 + (lane & 15)];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; r++) D[(size_t)(4 * (lane >> 4) + r) * 16 + (lane & 15)] = acc[r];
}

static float chain(const float* A, const float* B, int row, int col, int K)
{
    float s = 0.f;
    for (int k = 0; k < K; k++) s = fmaf(A[(size_t)row * K + k], B[(size_t)k * 64 + col], s);
    return s;
}
static float chain_unfused(const float* A, const float* B, int row, int col, int K)
{
    volatile float s = 0.f;
    for (int k = 0; k < K; k++) { volatile float p = A[(size_t)row * K + k] * B[(size_t)k * 64 + col]; s = s + p; }
    return s;
}

int main()
{
    const int K = 4608;
    std::vector<float> A(64 * (size_t)K), B((size_t)K * 64);
    int total_bad[3] = {0, 0, 0}, total_unf[3] = {0, 0, 0}, total = 0;
    for (int trial = 0; trial < 6; trial++) {
        srand(1234 + trial);
        for (auto& v : A) {
            if (trial < 3) v = ((float)(rand() % 256) - 128.f) * 0.0123f * (1 + trial);       // dequantised-uint8 like
            else v = ldexpf((float)rand() / RAND_MAX - 0.5f, rand() % 40 - 20);                 // wide exponent range
        }
        for (auto& v : B) {
            if (trial < 3) v = ((float)(rand() % 256) - 131.f) * 0.00217f;
            else v = ldexpf((float)rand() / RAND_MAX - 0.5f, rand() % 40 - 20);
        }
        float *dA, *dB, *dD;
        hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, 64 * 64 * 4);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        std::vector<float> D(64 * 64);
        for (int variant = 0; variant < 3; variant++) {
            if (variant == 0) hipLaunchKernelGGL(k_x1, dim3(1), dim3(64), 0, 0, dA, dB, dD, K);
            if (variant == 1) hipLaunchKernelGGL(k_x2, dim3(1), dim3(64), 0, 0, dA, dB, dD, K);
            if (variant == 2) hipLaunchKernelGGL(k_x4, dim3(1), dim3(64), 0, 0, dA, dB, dD, K);
            hipDeviceSynchronize();
            hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
            int bad = 0, unf = 0, n = 0;
            if (variant == 0)
                for (int blk = 0; blk < 2; blk++)
                    for (int r = 0; r < 32; r++)
                        for (int c = 0; c < 32; c++, n++) {
                            const float got = D[((size_t)blk * 32 + r) * 32 + c];
                            const float want = chain(A.data(), B.data(), blk * 32 + r, blk * 32 + c, K);
                            bad += memcmp(&got, &want, 4) != 0;
                            const float w2 = chain_unfused(A.data(), B.data(), blk * 32 + r, blk * 32 + c, K);
                            unf += memcmp(&got, &w2, 4) != 0;
                        }
            else {
                const int dim = variant == 1 ? 32 : 16;
                for (int r = 0; r < dim; r++)
                    for (int c = 0; c < dim; c++, n++) {
                        const float got = D[(size_t)r * dim + c];
                        const float want = chain(A.data(), B.data(), r, c, K);
                        bad += memcmp(&got, &want, 4) != 0;
                        const float w2 = chain_unfused(A.data(), B.data(), r, c, K);
                        unf += memcmp(&got, &w2, 4) != 0;
                    }
            }
            printf("trial %d variant %s: %d outputs, %d differ from the fused chain, %d differ from mul-then-add\n", trial,
                   variant == 0 ? "32x32x1" : variant == 1 ? "32x32x2" : "16x16x4", n, bad, unf);
            total_bad[variant] += bad; total_unf[variant] += unf;
        }
        total++;
        hipFree(dA); hipFree(dB); hipFree(dD);
    }
    printf("SUMMARY fused-chain mismatches: x1 %d, x2 %d, x4 %d ; unfused mismatches: x1 %d, x2 %d, x4 %d\n", total_bad[0],
           total_bad[1], total_bad[2], total_unf[0], total_unf[1], total_unf[2]);
    return 0;
}
