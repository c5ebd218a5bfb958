// VERDICT r5 item 5 as a measurement: can MobileNet-v1's last launch (fc7: 1024 -> 1000 on the pooled vector, 2.8 us at batch 1) be
// folded into the launch in front of it (conv6/sep + pool6, 64 blocks of one 16-channel slice each)?  fc7 needs ALL 1024 pooled
// values, each block owns 16: the fold is "every block adds its 16-channel partial dot products for the 1000 outputs into a global
// int32 accumulator with atomicAdd (integer sums are order independent: bit-exact), a ticket counter finds the last block, which
// requantises and stores" -- 64 000 device-scope atomics and one all-to-one hand-over on the critical path of a 52 us step.
//
// Two dependent chains of the TAIL ONLY (the pointwise convolution in front is the same in both and left out):
//   (a) two launches: pool stand-in (64 blocks write their 16 pooled bytes) -> fc kernel (125 blocks of 8 outputs, 1 MB of weights)
//   (b) one launch:   64 blocks: pooled bytes -> 1000 partial sums from the block's [1000][16] weight slice (prefetched before anything
//       else, as the real kernel could) -> atomicAdd -> __threadfence -> ticket; the last block reads the sums back, requantises, stores
//       and re-arms the accumulator for the next pass
// timed as back-to-back dependent passes on one stream (the launch boundary of the stream path, ~the direct path's + 0.5 us, is in
// both), results compared.
// build + run: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o fc_fold_probe.bin fc_fold_probe.hip && ./fc_fold_probe.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#define CHECK(e)                                                                                                   \
    do {                                                                                                           \
        hipError_t e_ = (e);                                                                                       \
        if (e_ != hipSuccess) { fprintf(stderr, "%s failed: %s (line %d)\n", #e, hipGetErrorString(e_), __LINE__); exit(1); } \
    } while (0)

constexpr int CIN = 1024, COUT = 1000, SLICES = 64;

__device__ __forceinline__ int8_t requant(int acc, float m) { float y = rintf((float)acc * m); y = fminf(fmaxf(y, -127.f), 127.f); return (int8_t)y; }

// (a1) the stand-in for the end of conv6/sep + pool6: block b publishes its 16 pooled values (derived from the pass counter so every
// pass has different data and nothing can be hoisted)
__global__ __launch_bounds__(512) void pool_standin_kernel(const int8_t* __restrict__ seed, int8_t* __restrict__ pooled, int pass)
{
    if (threadIdx.x < 16) pooled[blockIdx.x * 16 + threadIdx.x] = (int8_t)(seed[blockIdx.x * 16 + threadIdx.x] + pass);
}

// (a2) fc7 as its own launch: 125 blocks x 8 outputs, a wave per output, K = 1024 (w: [1000][1024])
__global__ __launch_bounds__(512) void fc_kernel(const int8_t* __restrict__ pooled, const int8_t* __restrict__ w, const int* __restrict__ bias, int8_t* __restrict__ y, float m)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int o = blockIdx.x * 8 + wave;
    if (o >= COUT) return;
    const uint4 xv = *reinterpret_cast<const uint4*>(pooled + lane * 16);
    const uint4 wv = *reinterpret_cast<const uint4*>(w + (size_t)o * CIN + lane * 16);
    int acc = 0;
    acc = __builtin_amdgcn_sdot4((int)xv.x, (int)wv.x, acc, false); acc = __builtin_amdgcn_sdot4((int)xv.y, (int)wv.y, acc, false);
    acc = __builtin_amdgcn_sdot4((int)xv.z, (int)wv.z, acc, false); acc = __builtin_amdgcn_sdot4((int)xv.w, (int)wv.w, acc, false);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if (lane == 0) y[o] = requant(acc + bias[o], m);
}

// (b) the fold: ws = the same weights in slice order [64 slices][1000][16]
__global__ __launch_bounds__(512) void fold_kernel(const int8_t* __restrict__ seed, const int8_t* __restrict__ ws, const int* __restrict__ bias, int* __restrict__ sums,
                                                   unsigned* __restrict__ ticket, int8_t* __restrict__ y, float m, int pass)
{
    __shared__ int8_t px[16];
    __shared__ unsigned last;
    const int b = blockIdx.x, t = threadIdx.x;
    // the block's weight slice first (independent of everything before it in the real kernel): outputs t and t + 512
    const uint4 w0 = *reinterpret_cast<const uint4*>(ws + ((size_t)b * COUT + t) * 16);
    uint4 w1 = make_uint4(0, 0, 0, 0);
    if (t + 512 < COUT) w1 = *reinterpret_cast<const uint4*>(ws + ((size_t)b * COUT + t + 512) * 16);
    if (t < 16) px[t] = (int8_t)(seed[b * 16 + t] + pass);
    __syncthreads();
    const uint4 xv = *reinterpret_cast<const uint4*>(px);
    int a0 = 0, a1 = 0;
    a0 = __builtin_amdgcn_sdot4((int)xv.x, (int)w0.x, a0, false); a0 = __builtin_amdgcn_sdot4((int)xv.y, (int)w0.y, a0, false);
    a0 = __builtin_amdgcn_sdot4((int)xv.z, (int)w0.z, a0, false); a0 = __builtin_amdgcn_sdot4((int)xv.w, (int)w0.w, a0, false);
    a1 = __builtin_amdgcn_sdot4((int)xv.x, (int)w1.x, a1, false); a1 = __builtin_amdgcn_sdot4((int)xv.y, (int)w1.y, a1, false);
    a1 = __builtin_amdgcn_sdot4((int)xv.z, (int)w1.z, a1, false); a1 = __builtin_amdgcn_sdot4((int)xv.w, (int)w1.w, a1, false);
    atomicAdd(sums + t, a0);
    if (t + 512 < COUT) atomicAdd(sums + t + 512, a1);
    __threadfence();
    __syncthreads();
    if (t == 0) last = atomicAdd(ticket, 1u);
    __syncthreads();
    if (last != SLICES - 1) return;
    __threadfence();
    for (int o = t; o < COUT; o += 512) {
        const int s = __hip_atomic_load(sums + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        y[o] = requant(s + bias[o], m);
        __hip_atomic_store(sums + o, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);         // re-armed for the next pass
    }
    if (t == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

int main()
{
    std::vector<int8_t> hw((size_t)COUT * CIN), hws((size_t)COUT * CIN), hseed(CIN);
    std::vector<int> hb(COUT);
    unsigned lcg = 777u;
    auto rnd = [&]() { lcg = lcg * 1664525u + 1013904223u; return (int)(lcg >> 24) - 128; };
    for (auto& v : hw) v = (int8_t)rnd();
    for (auto& v : hseed) v = (int8_t)(rnd() / 2);
    for (auto& v : hb) v = rnd() * 50;
    for (int s = 0; s < SLICES; s++)
        for (int o = 0; o < COUT; o++)
            for (int k = 0; k < 16; k++) hws[((size_t)s * COUT + o) * 16 + k] = hw[(size_t)o * CIN + s * 16 + k];
    int8_t *dw, *dws, *dseed, *dpooled, *dya, *dyb;
    int *dbias, *dsums;
    unsigned* dticket;
    CHECK(hipMalloc(&dw, hw.size())); CHECK(hipMalloc(&dws, hws.size())); CHECK(hipMalloc(&dseed, CIN)); CHECK(hipMalloc(&dpooled, CIN));
    CHECK(hipMalloc(&dya, 1024)); CHECK(hipMalloc(&dyb, 1024)); CHECK(hipMalloc(&dbias, COUT * 4)); CHECK(hipMalloc(&dsums, 1024 * 4)); CHECK(hipMalloc(&dticket, 64));
    CHECK(hipMemcpy(dw, hw.data(), hw.size(), hipMemcpyHostToDevice)); CHECK(hipMemcpy(dws, hws.data(), hws.size(), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dseed, hseed.data(), CIN, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dbias, hb.data(), COUT * 4, hipMemcpyHostToDevice));
    CHECK(hipMemset(dsums, 0, 1024 * 4)); CHECK(hipMemset(dticket, 0, 64));
    const float m = 1.0f / 2048.f;
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    auto two = [&](int pass) {
        hipLaunchKernelGGL(pool_standin_kernel, dim3(SLICES), dim3(512), 0, st, dseed, dpooled, pass);
        hipLaunchKernelGGL(fc_kernel, dim3(125), dim3(512), 0, st, dpooled, dw, dbias, dya, m);
    };
    auto one = [&](int pass) { hipLaunchKernelGGL(fold_kernel, dim3(SLICES), dim3(512), 0, st, dseed, dws, dbias, dsums, dticket, dyb, m, pass); };
    auto only_pool = [&](int pass) { hipLaunchKernelGGL(pool_standin_kernel, dim3(SLICES), dim3(512), 0, st, dseed, dpooled, pass); };
    // correctness: same bytes on several passes
    size_t bad = 0;
    for (int pass = 0; pass < 5; pass++) {
        two(pass); one(pass);
        CHECK(hipStreamSynchronize(st));
        std::vector<int8_t> ya(COUT), yb(COUT);
        CHECK(hipMemcpy(ya.data(), dya, COUT, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(yb.data(), dyb, COUT, hipMemcpyDeviceToHost));
        for (int o = 0; o < COUT; o++) bad += ya[o] != yb[o];
    }
    printf("fold vs two launches: %zu of %d bytes differ over 5 passes -> %s\n", bad, 5 * COUT, bad ? "DIFFERENT" : "identical");
    auto chain_us = [&](auto&& f, int n) {
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        for (int i = 0; i < 20; i++) f(i);
        CHECK(hipStreamSynchronize(st));
        CHECK(hipEventRecord(e0, st));
        for (int i = 0; i < n; i++) f(i);
        CHECK(hipEventRecord(e1, st));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        return 1e3f * ms / n;
    };
    for (int rep = 0; rep < 3; rep++) {
        const float tp = chain_us(only_pool, 2000), ta = chain_us(two, 2000), tb = chain_us(one, 2000);
        printf("per pass, dependent chain on one stream: stand-in alone %.2f us | (a) stand-in + fc7 launch %.2f us | (b) folded, one launch %.2f us  ->  fc7 as a launch costs %.2f us, as a fold %.2f us\n",
               tp, ta, tb, ta - tp, tb - tp);
    }
    return bad ? 2 : 0;
}
