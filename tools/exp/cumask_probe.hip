// Which CUs does a CU mask select on this device?  For two mask shapes (the first 32 bits; every 8th bit) launch 2048 workgroups on a
// masked stream, each records the XCC id and hardware id it ran on.  Build: hipcc --offload-arch=gfx950 -O2 cumask_probe.hip -o cumask_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>

__global__ void where(unsigned* out)
{
    unsigned xcc = 0, hwid = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hwid; }
    for (int i = 0; i < 2000; i++) asm volatile("s_nop 15");
}

int main()
{
    unsigned* d;
    hipMalloc(&d, 2 * 2048 * sizeof(unsigned));
    const char* names[3] = {"no mask", "bits 0..31", "every 8th bit"};
    for (int m = 0; m < 3; m++) {
        uint32_t mask[8] = {0};
        if (m == 1) mask[0] = 0xffffffffu;
        if (m == 2) for (int i = 0; i < 8; i++) mask[i] = 0x01010101u;
        hipStream_t s;
        if (m == 0) hipStreamCreate(&s);
        else if (hipExtStreamCreateWithCUMask(&s, 8, mask) != hipSuccess) { printf("%s: stream creation failed\n", names[m]); continue; }
        hipMemsetAsync(d, 0xff, 2 * 2048 * sizeof(unsigned), s);
        hipLaunchKernelGGL(where, dim3(2048), dim3(64), 0, s, d);
        hipStreamSynchronize(s);
        std::vector<unsigned> h(2 * 2048);
        hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
        std::map<unsigned, std::map<unsigned, int>> per;      // xcc -> (se, cu) -> count
        for (int b = 0; b < 2048; b++) per[h[2 * b] & 0xf][(h[2 * b + 1] >> 8) & 0xff7]++;       // HW_ID: cu_id bits 8-11, sh 12, se 13-15
        printf("%s:", names[m]);
        for (auto& x : per) printf("  xcc %u: %zu CUs", x.first, x.second.size());
        printf("\n");
        hipStreamDestroy(s);
    }
    return 0;
}
