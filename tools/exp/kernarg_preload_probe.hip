// Does the packet processor of this box preload kernel arguments into SGPRs, and what is the argument segment's round trip worth?
// One tiny kernel whose only work depends on its two pointer arguments, launched as a chain of dependent launches inside a hipGraph;
// build twice: plain, and with -mllvm -amdgpu-kernarg-preload-count=4 (then the compatibility prologue at the entry loads the arguments
// only on firmware WITHOUT the feature; with it the wave starts 256 bytes in, the pointers already in SGPRs).  Prints us per launch.
// build: hipcc --offload-arch=gfx950 -O3 [-mllvm -amdgpu-kernarg-preload-count=4] -o kernarg_preload_probe[_on].bin kernarg_preload_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)

__global__ void step(const int* __restrict__ in, int* __restrict__ out, int pad0, int pad1, int pad2, int pad3)
{
    out[threadIdx.x] = in[threadIdx.x] + 1;
}

int main()
{
    int *a, *b;
    CK(hipMalloc(&a, 4096)); CK(hipMalloc(&b, 4096));
    CK(hipMemset(a, 0, 4096)); CK(hipMemset(b, 0, 4096));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    const int N = 200;
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < N; i++) hipLaunchKernelGGL(step, dim3(1), dim3(64), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, 0, 0, 0, 0);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < 3; r++) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    float best = 1e30f;
    for (int r = 0; r < 10; r++) {
        CK(hipEventRecord(e0, s));
        for (int k = 0; k < 5; k++) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    int h[64];
    CK(hipMemcpy(h, a, 256, hipMemcpyDeviceToHost));
    printf("%.3f us per dependent launch (chain of %d in a hipGraph, best of 10 x 5 replays); a[0] = %d (counts the launches: the kernel ran on its real arguments)\n", best * 1e3f / (5 * N), N, h[0]);
    return 0;
}
