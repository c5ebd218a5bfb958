// What does a WRITE-heavy layer have to live with?  Streaming stores / loads / mixed on buffers of the layers' sizes (25.7 MB:
// ResNet-50 res2 outputs at batch 32, inside the 256 MB Infinity Cache) and far past it (1 GiB), in the access shapes the int8
// epilogues use: 16 B per lane fully coalesced; 16 B per lane in 32-B pieces of 256-B pixel rows (column-major MFMA epilogue);
// dword per lane in 128-B pieces (row-major epilogue).  Prints GB/s per variant.
// build: hipcc --offload-arch=gfx950 -O3 -o write_bw.bin write_bw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void fill16(uint4* p, size_t n16, unsigned v)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4(v, v, v, v);
}
// column-major epilogue: a wave stores 32 pixel rows x 32 B per instruction (lane l31 -> pixel, hi -> 16-B half), 256-B rows
__global__ void fill_cols(char* p, size_t pixels, unsigned v)
{
    const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t t = wave; t < pixels / 32; t += nw)
        for (int i = 0; i < 8; i++) *reinterpret_cast<uint4*>(p + (t * 32 + l31) * 256 + i * 32 + hi * 16) = make_uint4(v, v, v, v);
}
// row-major epilogue: a wave stores 2 pixel rows x 128 B per instruction (dword per lane), 16 instructions per 32-pixel tile
__global__ void fill_rows(char* p, size_t pixels, unsigned v)
{
    const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t t = wave; t < pixels / 32; t += nw)
        for (int g = 0; g < 2; g++)
            for (int r = 0; r < 16; r++)
                *reinterpret_cast<unsigned*>(p + (t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi) * 256 + g * 128 + l31 * 4) = v;
}
__global__ void read16(const uint4* p, size_t n16, unsigned* out)
{
    unsigned s = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { uint4 v = p[i]; s += v.x ^ v.y ^ v.z ^ v.w; }
    if (s == 0x12345678u) *out = s;
}
// read 1 part, write 4 parts (64 -> 256 channels)
__global__ void expand4(const uint4* src, uint4* dst, size_t n16)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = src[i];
        for (int k = 0; k < 4; k++) dst[i * 4 + k] = v;
    }
}
// residual tail traffic: read 1 + read 4 + write 4
__global__ void expand4_res(const uint4* src, const uint4* res, uint4* dst, size_t n16)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = src[i];
        for (int k = 0; k < 4; k++) { uint4 r = res[i * 4 + k]; r.x ^= v.x; r.y += v.y; dst[i * 4 + k] = r; }
    }
}

template <typename F>
static float time_it(F f, int reps)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); f();
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; i++) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main()
{
    const size_t sizes[] = {25690112ull, 1ull << 30};
    unsigned* flag; CK(hipMalloc(&flag, 4));
    for (size_t bytes : sizes) {
        char *a, *b, *c;
        CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&c, bytes));
        CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes)); CK(hipMemset(c, 3, bytes));
        const size_t n16 = bytes / 16, pixels = bytes / 256;
        const int reps = bytes > (1u << 28) ? 5 : 30;
        printf("---- %.1f MB\n", bytes / 1e6);
        for (int blocks : {512, 2048, 8192}) {
            float t;
            t = time_it([&] { hipLaunchKernelGGL(fill16, dim3(blocks), dim3(256), 0, 0, (uint4*)a, n16, 7u); }, reps);
            printf("blocks %5d  fill 16B/lane coalesced      %7.1f us  %7.1f GB/s written\n", blocks, t * 1e3, bytes / t / 1e6);
            t = time_it([&] { hipLaunchKernelGGL(fill_cols, dim3(blocks), dim3(256), 0, 0, a, pixels, 7u); }, reps);
            printf("blocks %5d  fill column-major pieces     %7.1f us  %7.1f GB/s written\n", blocks, t * 1e3, bytes / t / 1e6);
            t = time_it([&] { hipLaunchKernelGGL(fill_rows, dim3(blocks), dim3(256), 0, 0, a, pixels, 7u); }, reps);
            printf("blocks %5d  fill row-major dwords        %7.1f us  %7.1f GB/s written\n", blocks, t * 1e3, bytes / t / 1e6);
            t = time_it([&] { hipLaunchKernelGGL(read16, dim3(blocks), dim3(256), 0, 0, (const uint4*)a, n16, flag); }, reps);
            printf("blocks %5d  read 16B/lane                %7.1f us  %7.1f GB/s read\n", blocks, t * 1e3, bytes / t / 1e6);
            t = time_it([&] { hipLaunchKernelGGL(expand4, dim3(blocks), dim3(256), 0, 0, (const uint4*)b, (uint4*)a, n16 / 4); }, reps);
            printf("blocks %5d  read 1/4 + write 1           %7.1f us  %7.1f GB/s total\n", blocks, t * 1e3, 1.25 * bytes / t / 1e6);
            t = time_it([&] { hipLaunchKernelGGL(expand4_res, dim3(blocks), dim3(256), 0, 0, (const uint4*)b, (const uint4*)c, (uint4*)a, n16 / 4); }, reps);
            printf("blocks %5d  read 1/4 + read 1 + write 1  %7.1f us  %7.1f GB/s total\n", blocks, t * 1e3, 2.25 * bytes / t / 1e6);
        }
        CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(c));
    }
    return 0;
}
