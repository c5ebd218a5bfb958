// How long does a producer -> consumer hand-over between two workgroups of ONE running kernel take on MI355X, compared with
// the dependent-launch boundary (1.25-1.45 us + ramp, profiles/r02_launch_chain2_device_clock_anatomy.txt)?
// Ping-pong between block 0 and block B (B = 1: the neighbouring XCD, workgroups are dealt round-robin over the 8 XCDs;
// B = 8: the same XCD): the producer stores a 4 KB payload, releases (agent scope) and bumps a flag; the consumer spins on the
// flag with acquire loads, reads the payload, answers on a second flag.  Every spin is bounded (no hang on a logic error).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)

__device__ __forceinline__ bool wait_for(const int* flag, int want)
{
    for (int spin = 0; spin < (1 << 22); spin++)
        if (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= want) return true;
    return false;
}

__global__ __launch_bounds__(256) void pingpong(int* flags, int* payload, unsigned long long* out, int partner, int rounds, int words)
{
    const int b = blockIdx.x;
    if (b != 0 && b != partner) return;
    int* f_go = flags;            // written by block 0
    int* f_back = flags + 64;     // written by the partner (own cache line)
    __shared__ int ok;
    if (threadIdx.x == 0) ok = 1;
    __syncthreads();
    long long t0 = 0;
    int acc = 0;
    if (b == 0) t0 = wall_clock64();
    for (int r = 1; r <= rounds; r++) {
        if (b == 0) {
            for (int i = threadIdx.x; i < words; i += 256) payload[i] = r + i;
            __syncthreads();          // all stores issued ...
            if (threadIdx.x == 0) {
                __hip_atomic_store(f_go, r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);      // ... released with the flag
                if (!wait_for(f_back, r)) ok = 0;
            }
            __syncthreads();
            if (!ok) break;
        } else {
            if (threadIdx.x == 0 && !wait_for(f_go, r)) ok = 0;
            __syncthreads();
            if (!ok) break;
            for (int i = threadIdx.x; i < words; i += 256) acc += __hip_atomic_load(payload + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - (r + i);
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_store(f_back, r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (b == 0 && threadIdx.x == 0) { out[0] = wall_clock64() - t0; out[1] = ok; }
    if (b != 0) atomicAdd((int*)(out + 2), acc);       // 0 when every payload word arrived
    if (b != 0 && threadIdx.x == 0) out[3] = ok;
}

int main()
{
    int *flags, *payload; unsigned long long* out;
    CK(hipMalloc(&flags, 1024)); CK(hipMalloc(&payload, 1 << 20)); CK(hipMalloc(&out, 64));
    const int rounds = 2000;
    int rate = 0;
    CK(hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0));      // kHz
    printf("wall clock %d kHz; round trip = 2 hand-overs (payload store + release + flag, acquire + payload load)\n", rate);
    for (int partner : {1, 4, 8, 9, 16}) {
        for (int words : {0, 1024, 16384}) {
            unsigned long long h[4];
            for (int rep = 0; rep < 2; rep++) {
                CK(hipMemset(flags, 0, 1024)); CK(hipMemset(out, 0, 64));
                hipLaunchKernelGGL(pingpong, dim3(partner + 1), dim3(256), 0, 0, flags, payload, out, partner, rounds, words);
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(h, out, 32, hipMemcpyDeviceToHost));
            }
            printf("block 0 <-> block %-2d (%s XCD) payload %6d B: %.3f us per hand-over  ok=%llu/%llu payload_err=%d\n", partner,
                   partner % 8 ? "other" : "same", words * 4, (double)h[0] / rate * 1e3 / rounds / 2, h[1], h[3], (int)h[2]);
        }
    }
    return 0;
}
