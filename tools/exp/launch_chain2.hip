// Device-clock anatomy of a short dependent launch (follow-up of launch_chain.hip).
//
// rocprofv3's per-dispatch Start/End on this stack abut (gap 0.00, "body" = dispatch latency + kernel), so the split
// is measured ON THE DEVICE: every block records s_memrealtime (constant 100 MHz, chip-global) at entry and exit, wave 0
// of every block also at its internal stage boundaries.  For launch i of a hipGraph chain:
//     gap_i   = min(entry_i) - max(exit_{i-1})        the dependent-launch boundary as the shader sees it
//     ramp_i  = max(entry_i) - min(entry_i)           dispatch skew over the grid
//     body_i  = max(exit_i)  - min(entry_i)
// plus the average stage durations inside the gemm_direct-shaped body, an instruction-fetch probe (straight-line
// s_nop padding: does cold code cost time at every launch?) and the shader clock actually running in such a chain
// (s_memtime ticks per s_memrealtime tick).
//
// build: hipcc --offload-arch=gfx950 -O3 -o launch_chain2.bin launch_chain2.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int NSTAMP = 8;
struct Args {
    const int8_t* x; const int8_t* w; const int* bias; const float* scale; int8_t* y;
    unsigned long long* stamps;      // [block][NSTAMP]
    int M, K, cout, ldc;
    float m1, os;
    int variant;
    int pad[32];
};

__device__ __forceinline__ unsigned long long rt() { return wall_clock64(); }
#define STAMP(i) do { if ((threadIdx.x & 63) == 0 && threadIdx.x < 64) a.stamps[(size_t)blockIdx.x * NSTAMP + (i)] = rt(); } while (0)

__global__ __launch_bounds__(256) void k_ld1(Args a)
{
    STAMP(0);
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const v4i v = *reinterpret_cast<const v4i*>(a.x + i * 16);
    *reinterpret_cast<v4i*>(a.y + i * 16) = v + 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(7);
}

template <int N>
__global__ __launch_bounds__(256) void k_nops(Args a)
{
    STAMP(0);
    if (N >= 256) asm volatile(".rept 256\n s_nop 0\n .endr" ::: "memory");
    if (N >= 1024) asm volatile(".rept 768\n s_nop 0\n .endr" ::: "memory");
    if (N >= 4096) asm volatile(".rept 3072\n s_nop 0\n .endr" ::: "memory");
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const v4i v = *reinterpret_cast<const v4i*>(a.x + i * 16);
    *reinterpret_cast<v4i*>(a.y + i * 16) = v + 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(7);
}

__device__ __forceinline__ unsigned rq4(int a0, int a1, int a2, int a3, float4 s, float m1, float inv)
{
    auto one = [&](int a, float sc) {
        float f = (float)a * m1 * sc;
        f = fminf(fmaxf(f, -127.f), 127.f);
        return (int)__fmaf_rn(f, inv, copysignf(0.5f, f)) & 0xff;
    };
    return one(a0, s.x) | (one(a1, s.y) << 8) | (one(a2, s.z) << 16) | (one(a3, s.w) << 24);
}

// gemm_direct shape with stage stamps.  variant bit 0: skip the epilogue arithmetic (store raw), bit 1: no LDS reduce
// (every wave stores its own partial), bit 2: no MFMA, bit 3: coalesced operand rows (lane -> consecutive 16 B)
__global__ __launch_bounds__(256) void k_mimic(Args a)
{
    __shared__ int red[3 * 16 * 64];
    STAMP(0);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, hi = lane >> 5;
    const int tiles_n = a.cout / 32;
    const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x % tiles_n;
    const int m = tile_m * 32 + l31, n0 = tile_n * 32;
    const int per = a.K / 32 / 4, sb = wave * per;
    const int8_t* wp = a.w + (size_t)(n0 + l31) * a.K + hi * 16;
    const int8_t* xp = a.x + (size_t)(m < a.M ? m : 0) * a.K + hi * 16;
    if (a.variant & 8) { wp = a.w + ((size_t)blockIdx.x * 256 + t) * 16; xp = a.x + ((size_t)blockIdx.x * 256 + t) * 16; }
    int4 b4[4]; float4 s4[4];
#pragma unroll
    for (int g = 0; g < 4; g++) {
        b4[g] = *reinterpret_cast<const int4*>(a.bias + n0 + 8 * g + 4 * hi);
        s4[g] = *reinterpret_cast<const float4*>(a.scale + n0 + 8 * g + 4 * hi);
    }
    v16i acc;
#pragma unroll
    for (int e = 0; e < 16; e++) acc[e] = 0;
    v4i af[4], bf[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const size_t off = (a.variant & 8) ? (size_t)u * 65536 : (size_t)(sb + u) * 32;
        af[u] = *reinterpret_cast<const v4i*>(wp + off);
        bf[u] = *reinterpret_cast<const v4i*>(xp + off);
    }
    STAMP(1);                                             // loads issued
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(2);                                             // loads landed
    if (!(a.variant & 4)) {
#pragma unroll
        for (int u = 0; u < 4; u++) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[u], bf[u], acc, 0, 0, 0);
    } else {
#pragma unroll
        for (int u = 0; u < 4; u++) acc[u] += af[u][0] + bf[u][1];
    }
    asm volatile("s_nop 7\n s_nop 7" : "+v"(acc));
    STAMP(3);                                             // MFMA results readable
    if (!(a.variant & 2)) {
        if (wave > 0) {
#pragma unroll
            for (int e = 0; e < 16; e++) red[((wave - 1) * 16 + e) * 64 + lane] = acc[e];
        }
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int k2 = 0; k2 < 3; k2++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[e] += red[(k2 * 16 + e) * 64 + lane];
    }
    asm volatile("" : "+v"(acc));
    STAMP(4);                                             // reduced
    const float inv = 1.0f / a.os;
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const int c0 = n0 + 8 * g + 4 * hi;
        unsigned p;
        if (a.variant & 1) p = (unsigned)(acc[4 * g] + acc[4 * g + 1] + acc[4 * g + 2] + acc[4 * g + 3] + b4[g].x);
        else p = rq4(acc[4 * g] + b4[g].x, acc[4 * g + 1] + b4[g].y, acc[4 * g + 2] + b4[g].z, acc[4 * g + 3] + b4[g].w, s4[g], a.m1, inv);
        if (m < a.M) *reinterpret_cast<unsigned*>(a.y + (size_t)m * a.ldc + c0 + ((a.variant & 2) ? wave * 0 : 0)) = p;
    }
    STAMP(5);                                             // stores issued
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(7);                                             // stores acknowledged
}

__global__ void k_clock(unsigned long long* out, int spin)
{
    const unsigned long long r0 = wall_clock64(), c0 = clock64();
    while ((long long)(clock64() - c0) < spin) {}
    const unsigned long long r1 = wall_clock64(), c1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = r1 - r0; out[1] = c1 - c0; }
}

int main(int argc, char** argv)
{
    const int L = 12, reps = 50;
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int M = 196, K = 512, cout = 512, grid = 112;
    int8_t *xa, *xb, *w; int* bias; float* scale; unsigned long long* stamps;
    CK(hipMalloc(&xa, 1 << 22)); CK(hipMalloc(&xb, 1 << 22)); CK(hipMalloc(&w, 1 << 22));
    CK(hipMalloc(&bias, 4096 * 4)); CK(hipMalloc(&scale, 4096 * 4));
    CK(hipMalloc(&stamps, (size_t)L * 1024 * NSTAMP * 8));
    CK(hipMemset(xa, 1, 1 << 22)); CK(hipMemset(xb, 1, 1 << 22)); CK(hipMemset(w, 1, 1 << 22));
    CK(hipMemset(bias, 0, 4096 * 4));
    std::vector<float> sc(4096, 0.01f);
    CK(hipMemcpy(scale, sc.data(), 4096 * 4, hipMemcpyHostToDevice));

    // shader clock in a chain-like situation: short launch after idle, then back to back
    {
        unsigned long long* d; CK(hipMalloc(&d, 64));
        unsigned long long h[2];
        for (int i = 0; i < 4; i++) {
            hipLaunchKernelGGL(k_clock, dim3(256), dim3(256), 0, st, d, 20000);
            CK(hipStreamSynchronize(st));
            CK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
            printf("clock probe %d: %llu shader ticks in %llu realtime ticks (100 MHz) -> %.0f MHz\n", i, h[1], h[0], 100.0 * h[1] / h[0]);
        }
    }

    struct Row { const char* name; int kind; int variant; };
    const Row rows[] = {
        {"ld1", 0, 0}, {"nops 256 + ld1", 1, 0}, {"nops 1024 + ld1", 2, 0}, {"nops 4096 + ld1", 3, 0},
        {"mimic", 4, 0}, {"mimic, raw store (no requant)", 4, 1}, {"mimic, no LDS reduce", 4, 2}, {"mimic, no MFMA", 4, 4},
        {"mimic, no reduce no requant", 4, 3}, {"mimic, coalesced operand loads", 4, 8}, {"mimic, coalesced, no reduce, raw", 4, 11},
    };
    for (const Row& r : rows) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < L; i++) {
            Args a{};
            a.x = (i & 1) ? xb : xa; a.y = (i & 1) ? xa : xb; a.w = w; a.bias = bias; a.scale = scale;
            a.stamps = stamps + (size_t)i * 1024 * NSTAMP;
            a.M = M; a.K = K; a.cout = cout; a.ldc = K; a.m1 = 0.02f; a.os = 0.7f; a.variant = r.variant;
            switch (r.kind) {
            case 0: hipLaunchKernelGGL(k_ld1, dim3(grid), dim3(256), 0, st, a); break;
            case 1: hipLaunchKernelGGL(k_nops<256>, dim3(grid), dim3(256), 0, st, a); break;
            case 2: hipLaunchKernelGGL(k_nops<1024>, dim3(grid), dim3(256), 0, st, a); break;
            case 3: hipLaunchKernelGGL(k_nops<4096>, dim3(grid), dim3(256), 0, st, a); break;
            default: hipLaunchKernelGGL(k_mimic, dim3(grid), dim3(256), 0, st, a); break;
            }
        }
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 10; i++) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; i++) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        // stamps of the LAST replay
        std::vector<unsigned long long> h((size_t)L * 1024 * NSTAMP);
        CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
        double gap = 0, ramp = 0, body = 0, stage[NSTAMP] = {0};
        unsigned long long prev_exit = 0;
        int ng = 0;
        for (int i = 0; i < L; i++) {
            unsigned long long e_min = ~0ull, e_max = 0, x_max = 0;
            double sacc[NSTAMP] = {0};
            for (int b = 0; b < grid; b++) {
                const unsigned long long* s = &h[((size_t)i * 1024 + b) * NSTAMP];
                e_min = std::min(e_min, s[0]); e_max = std::max(e_max, s[0]); x_max = std::max(x_max, s[7]);
                for (int k = 1; k < NSTAMP; k++) sacc[k] += (s[k] >= s[0]) ? (double)(s[k] - s[0]) : 0.0;
            }
            if (i > 0) { gap += (double)(e_min - prev_exit); ng++; }
            ramp += (double)(e_max - e_min); body += (double)(x_max - e_min);
            for (int k = 1; k < NSTAMP; k++) stage[k] += sacc[k] / grid;
            prev_exit = x_max;
        }
        printf("%-38s %.2f us/launch | gap %.2f  ramp %.2f  body %.2f us", r.name, 1e3 * ms / reps / L, gap / ng / 100.0, ramp / L / 100.0, body / L / 100.0);
        if (r.kind == 4)
            printf(" | wave0 since entry: issued %.2f landed %.2f mfma %.2f reduced %.2f stored %.2f acked %.2f", stage[1] / L / 100.0,
                   stage[2] / L / 100.0, stage[3] / L / 100.0, stage[4] / L / 100.0, stage[5] / L / 100.0, stage[7] / L / 100.0);
        else
            printf(" | exit since entry %.2f", stage[7] / L / 100.0);
        printf("\n");
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
