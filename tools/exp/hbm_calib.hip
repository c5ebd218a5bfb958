// Calibration for rocprofv3 FETCH_SIZE / WRITE_SIZE on this box: a streaming copy of a known byte count
// (16 B per lane, grid-stride), larger than the 256 MiB Infinity Cache, launched 3 times.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void calib_copy_k(const float4* __restrict__ src, float4* __restrict__ dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
int main()
{
    const size_t bytes = 1024ull << 20;       // 1 GiB read + 1 GiB written per launch
    float4 *a, *b;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes);
    hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; it++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(calib_copy_k, dim3(256 * 8), dim3(256), 0, 0, a, b, bytes / 16);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("calib_copy_k: %zu bytes read + %zu bytes written in %.3f ms = %.2f TB/s (read+write)\n", bytes, bytes, ms, 2.0 * bytes / ms / 1e9);
    }
    return 0;
}
