// What does v_cvt_pk_u8_f32 do with fractions, negatives and values past 255?  (candidate for the requantising epilogue's
// float -> byte step: one instruction converts AND inserts the byte.)  Prints input, result byte.
// build: hipcc --offload-arch=gfx950 -O2 -o cvt_pk_u8_probe.bin cvt_pk_u8_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void probe(const float* x, unsigned* y, int n)
{
    const int i = threadIdx.x;
    if (i < n) y[i] = __builtin_amdgcn_cvt_pk_u8_f32(x[i], 1, 0xAABBCCDDu);
}

int main()
{
    const float h[] = {0.f, 0.49f, 0.5f, 0.51f, 0.99f, 1.0f, 1.5f, 2.5f, 3.5f, 127.5f, 128.25f, 128.75f, 254.99f, 255.f, 255.5f, 256.f, 300.f, 1e9f, -0.4f, -0.6f, -1.5f, -300.f};
    const int n = sizeof(h) / sizeof(h[0]);
    float* dx; unsigned* dy; unsigned r[64];
    hipMalloc(&dx, sizeof(h)); hipMalloc(&dy, sizeof(r));
    hipMemcpy(dx, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dx, dy, n);
    hipMemcpy(r, dy, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; i++) printf("%12g -> byte1 %3u   dword %08x\n", h[i], (r[i] >> 8) & 0xff, r[i]);
    return 0;
}
