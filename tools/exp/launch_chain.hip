// Where does a batch-1 launch's time go?  
//
// Chains of L dependent launches captured into ONE hipGraph and replayed, per-launch time = replay time / L.
// Each chain uses a different kernel body, from genuinely empty up to a gemm_direct-shaped body, so the
// differences between rows price the serial stages of a short kernel:
//
//   empty        no arguments, no memory access                     -> the dependent-launch boundary itself
//   args         reads a 256-B kernarg block, lane 0 stores 4 B      -> + kernarg fetch + one store drain
//   ld1          one 16-B load per lane of the PREVIOUS launch's output, 16-B store  -> + one memory round trip
//   ld2          a second, address-dependent 16-B load               -> + another round trip
//   mimic        gemm_direct shape: 16 x 16-B loads in flight, 8 MFMA 32x32x32 i8, LDS K-reduce over 4 waves,
//                bias/scale loads AFTER the reduction, requantising epilogue by wave 0, dword stores
//   mimic_h      same with bias/scale loads hoisted above the main loads
//   mimic_hq     hoisted + every wave requantises and stores a quarter of the tile
//
// build:  hipcc --offload-arch=gfx950 -O3 -o launch_chain launch_chain.hip ; run: ./launch_chain [L] [replays]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

struct Args {                 // ~ConvArgs sized
    const int8_t* x; const int8_t* w; const int* bias; const float* scale; int8_t* y;
    int M, K, cout, ldc;
    float m1, lo, hi, os;
    int pad[40];
};

__global__ void k_empty() {}

__global__ __launch_bounds__(256) void k_args(Args a)
{
    if (threadIdx.x == 0) *reinterpret_cast<int*>(a.y + (size_t)blockIdx.x * 64) = a.M + a.pad[39];
}

__global__ __launch_bounds__(256) void k_ld1(Args a)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const v4i v = *reinterpret_cast<const v4i*>(a.x + i * 16);
    *reinterpret_cast<v4i*>(a.y + i * 16) = v + 1;
}

__global__ __launch_bounds__(256) void k_ld2(Args a)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const v4i v = *reinterpret_cast<const v4i*>(a.x + i * 16);
    const size_t j = ((unsigned)v[0] & 1023u);
    const v4i u = *reinterpret_cast<const v4i*>(a.w + j * 16);
    *reinterpret_cast<v4i*>(a.y + i * 16) = v + u;
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

// MODE 0: bias/scale after the reduction, wave 0 does the whole epilogue; 1: hoisted; 2: hoisted + all waves store
template <int MODE>
__global__ __launch_bounds__(256) void k_mimic(Args a)
{
    __shared__ int red[4 * 16 * 64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, hi = lane >> 5;
    const int tiles_n = a.cout / 32;
    const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x % tiles_n;
    const int m = tile_m * 32 + l31, n0 = tile_n * 32;
    const int S = a.K / 32, per = S / 4, sb = wave * per;
    const int8_t* wp = a.w + (size_t)(n0 + l31) * a.K + hi * 16;
    const int8_t* xp = a.x + (size_t)(m < a.M ? m : 0) * a.K + hi * 16;
    int4 b4[4]; float4 s4[4];
    if (MODE >= 1) {
#pragma unroll
        for (int g = 0; g < 4; g++) {
            b4[g] = *reinterpret_cast<const int4*>(a.bias + n0 + 8 * g + 4 * hi);
            s4[g] = *reinterpret_cast<const float4*>(a.scale + n0 + 8 * g + 4 * hi);
        }
    }
    v16i acc;
#pragma unroll
    for (int e = 0; e < 16; e++) acc[e] = 0;
    v4i af[8], bf[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
        const int s = sb + (u < per ? u : 0);
        af[u] = *reinterpret_cast<const v4i*>(wp + (size_t)s * 32);
        bf[u] = *reinterpret_cast<const v4i*>(xp + (size_t)s * 32);
    }
#pragma unroll
    for (int u = 0; u < 8; u++)
        if (u < per) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[u], bf[u], acc, 0, 0, 0);
    if (MODE <= 1) {
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
        const float inv = 1.0f / a.os;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int c0 = n0 + 8 * g + 4 * hi;
            if (MODE == 0) { b4[g] = *reinterpret_cast<const int4*>(a.bias + c0); s4[g] = *reinterpret_cast<const float4*>(a.scale + c0); }
            const unsigned p = rq4(acc[4 * g] + b4[g].x, acc[4 * g + 1] + b4[g].y, acc[4 * g + 2] + b4[g].z, acc[4 * g + 3] + b4[g].w, s4[g], a.m1, inv);
            if (m < a.M) *reinterpret_cast<unsigned*>(a.y + (size_t)m * a.ldc + c0) = p;
        }
    } else {
        // every wave publishes its partials; wave g then reduces and requantises channel group g
#pragma unroll
        for (int e = 0; e < 16; e++) red[(wave * 16 + e) * 64 + lane] = acc[e];
        __syncthreads();
        const int g = wave;
        int r[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            r[e] = 0;
#pragma unroll
            for (int k2 = 0; k2 < 4; k2++) r[e] += red[(k2 * 16 + 4 * g + e) * 64 + lane];
        }
        const float inv = 1.0f / a.os;
        const int c0 = n0 + 8 * g + 4 * hi;
        const int4 bb = g == 0 ? b4[0] : g == 1 ? b4[1] : g == 2 ? b4[2] : b4[3];
        const float4 ss = g == 0 ? s4[0] : g == 1 ? s4[1] : g == 2 ? s4[2] : s4[3];
        const unsigned p = rq4(r[0] + bb.x, r[1] + bb.y, r[2] + bb.z, r[3] + bb.w, ss, a.m1, inv);
        if (m < a.M) *reinterpret_cast<unsigned*>(a.y + (size_t)m * a.ldc + c0) = p;
    }
}

int main(int argc, char** argv)
{
    const int L = argc > 1 ? atoi(argv[1]) : 30;
    const int reps = argc > 2 ? atoi(argv[2]) : 300;
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int M = 196, K = 512, cout = 512;                       // the 14x14 512->512 MobileNet layers
    int8_t *xa, *xb, *w; int* bias; float* scale;
    CK(hipMalloc(&xa, 1 << 22)); CK(hipMalloc(&xb, 1 << 22)); CK(hipMalloc(&w, 1 << 22));
    CK(hipMalloc(&bias, 4096 * 4)); CK(hipMalloc(&scale, 4096 * 4));
    CK(hipMemset(xa, 1, 1 << 22)); CK(hipMemset(xb, 1, 1 << 22)); CK(hipMemset(w, 1, 1 << 22));
    CK(hipMemset(bias, 0, 4096 * 4));
    std::vector<float> sc(4096, 0.01f);
    CK(hipMemcpy(scale, sc.data(), 4096 * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

    struct Row { const char* name; int kind; int grid; };
    const Row rows[] = {
        {"empty  grid 112", 0, 112}, {"empty  grid 256", 0, 256}, {"empty  grid 1024", 0, 1024},
        {"args   grid 112", 1, 112}, {"args   grid 256", 1, 256},
        {"ld1    grid 112", 2, 112}, {"ld1    grid 256", 2, 256}, {"ld1    grid 1024", 2, 1024},
        {"ld2    grid 112", 3, 112}, {"ld2    grid 256", 3, 256},
        {"mimic     (196x512x512, grid 112)", 4, 112}, {"mimic_h   (bias/scale hoisted)", 5, 112}, {"mimic_hq  (hoisted, all waves store)", 6, 112},
    };
    printf("%-40s %10s %12s\n", "chain of dependent launches", "us/launch", "us/replay");
    for (const Row& r : rows) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < L; i++) {
            Args a{};
            a.x = (i & 1) ? xb : xa; a.y = (i & 1) ? xa : xb; a.w = w; a.bias = bias; a.scale = scale;
            a.M = M; a.K = K; a.cout = cout; a.ldc = K; a.m1 = 0.02f; a.os = 0.7f;
            switch (r.kind) {
            case 0: hipLaunchKernelGGL(k_empty, dim3(r.grid), dim3(256), 0, st); break;
            case 1: hipLaunchKernelGGL(k_args, dim3(r.grid), dim3(256), 0, st, a); break;
            case 2: hipLaunchKernelGGL(k_ld1, dim3(r.grid), dim3(256), 0, st, a); break;
            case 3: hipLaunchKernelGGL(k_ld2, dim3(r.grid), dim3(256), 0, st, a); break;
            case 4: hipLaunchKernelGGL(k_mimic<0>, dim3(r.grid), dim3(256), 0, st, a); break;
            case 5: hipLaunchKernelGGL(k_mimic<1>, dim3(r.grid), dim3(256), 0, st, a); break;
            default: hipLaunchKernelGGL(k_mimic<2>, dim3(r.grid), dim3(256), 0, st, a); break;
            }
        }
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 20; i++) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        float best = 1e30f;
        for (int round = 0; round < 3; round++) {
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < reps; i++) CK(hipGraphLaunch(ge, st));
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("%-40s %10.3f %12.2f\n", r.name, 1e3 * best / reps / L, 1e3 * best / reps);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
