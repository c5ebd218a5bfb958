// Issue cost of the instructions the short batch-1 kernels are made of, one wave per SIMD (the situation of a
// latency-bound launch): cycles per instruction for a DEPENDENT chain and for FOUR independent chains, measured with
// s_memtime around 1024 instances.  build: hipcc --offload-arch=gfx950 -O3 -o valu_rates.bin valu_rates.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define REP256(x) ".rept 256\n" x "\n.endr\n"
#define REP64(x) ".rept 64\n" x "\n.endr\n"

// dependent: one chain of 1024; independent: 4 chains x 256
#define BENCH(ID, DEP, IND)                                                                                       \
    __global__ void k_dep_##ID(unsigned long long* out, int seed)                                                 \
    {                                                                                                             \
        int a = threadIdx.x + seed, b = seed * 3 + 1, c = 5, d = 7;                                               \
        float fa = (float)a * 0.001f, fb = 1.0001f, fc = 0.5f, fd = 0.25f;                                        \
        const unsigned long long t0 = clock64();                                                                  \
        asm volatile(".rept 1024\n" DEP "\n.endr\n" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd) : : "s20", "s21", "s22", "s23", "vcc", "scc"); \
        const unsigned long long t1 = clock64();                                                                  \
        if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;                                                \
        if (a + b + c + d == 0x7fffffff && fa + fb + fc + fd == 1.f) out[1] = 1;                                  \
    }                                                                                                             \
    __global__ void k_ind_##ID(unsigned long long* out, int seed)                                                 \
    {                                                                                                             \
        int a = threadIdx.x + seed, b = seed * 3 + 1, c = 5, d = 7;                                               \
        float fa = (float)a * 0.001f, fb = 1.0001f, fc = 0.5f, fd = 0.25f;                                        \
        const unsigned long long t0 = clock64();                                                                  \
        asm volatile(".rept 256\n" IND "\n.endr\n" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd) : : "s20", "s21", "s22", "s23", "vcc", "scc"); \
        const unsigned long long t1 = clock64();                                                                  \
        if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;                                                \
        if (a + b + c + d == 0x7fffffff && fa + fb + fc + fd == 1.f) out[1] = 1;                                  \
    }

// operands: %0..%3 ints a,b,c,d ; %4..%7 floats
BENCH(add, "v_add_u32 %0, %0, %1", "v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0")
BENCH(fma, "v_fma_f32 %4, %4, %5, %6", "v_fma_f32 %4, %4, %5, %6\n v_fma_f32 %5, %5, %5, %6\n v_fma_f32 %6, %6, %5, %7\n v_fma_f32 %7, %7, %5, %6")
BENCH(mul, "v_mul_f32 %4, %4, %5", "v_mul_f32 %4, %4, %5\n v_mul_f32 %5, %5, %5\n v_mul_f32 %6, %6, %5\n v_mul_f32 %7, %7, %5")
BENCH(perm, "v_perm_b32 %0, %0, %1, %2", "v_perm_b32 %0, %0, %1, %2\n v_perm_b32 %1, %1, %2, %3\n v_perm_b32 %2, %2, %3, %3\n v_perm_b32 %3, %3, %1, %1")
BENCH(dot4, "v_dot4_i32_i8 %0, %1, %2, %0", "v_dot4_i32_i8 %0, %1, %2, %0\n v_dot4_i32_i8 %1, %2, %3, %1\n v_dot4_i32_i8 %2, %3, %3, %2\n v_dot4_i32_i8 %3, %1, %1, %3")
BENCH(cvtfi, "v_cvt_f32_i32 %4, %0\n v_cvt_i32_f32 %0, %4", "v_cvt_f32_i32 %4, %0\n v_cvt_f32_i32 %5, %1\n v_cvt_i32_f32 %2, %6\n v_cvt_i32_f32 %3, %7")
BENCH(med3, "v_med3_f32 %4, %4, %5, %6", "v_med3_f32 %4, %4, %5, %6\n v_med3_f32 %5, %5, %6, %7\n v_med3_f32 %6, %6, %7, %7\n v_med3_f32 %7, %7, %5, %5")
BENCH(fract, "v_fract_f32 %4, %4", "v_fract_f32 %4, %4\n v_fract_f32 %5, %5\n v_fract_f32 %6, %6\n v_fract_f32 %7, %7")
BENCH(bfi, "v_bfi_b32 %0, %1, %2, %0", "v_bfi_b32 %0, %1, %2, %0\n v_bfi_b32 %1, %2, %3, %1\n v_bfi_b32 %2, %3, %3, %2\n v_bfi_b32 %3, %1, %1, %3")
BENCH(rcp, "v_rcp_f32 %4, %4", "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7")
BENCH(cnd, "v_cndmask_b32 %0, %0, %1, vcc", "v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %1, vcc")
BENCH(mullo, "v_mul_lo_u32 %0, %0, %1", "v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %1, %1, %2\n v_mul_lo_u32 %2, %2, %3\n v_mul_lo_u32 %3, %3, %1")
BENCH(lshladd64, "v_lshl_add_u32 %0, %0, 2, %1", "v_lshl_add_u32 %0, %0, 2, %1\n v_lshl_add_u32 %1, %1, 2, %2\n v_lshl_add_u32 %2, %2, 2, %3\n v_lshl_add_u32 %3, %3, 2, %1")
BENCH(salu, "s_add_u32 s20, s20, 1", "s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1")
BENCH(mixed, "v_add_u32 %0, %0, %1\n s_add_u32 s20, s20, 1", "v_add_u32 %0, %0, %1\n s_add_u32 s20, s20, 1\n v_add_u32 %1, %1, %2\n s_add_u32 s21, s21, 1")

__global__ void k_mfma16(unsigned long long* out, int dep)
{
    v4i a = {1, 2, 3, 4}, b = {5, 6, 7, 8}, c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    const unsigned long long t0 = clock64();
    if (dep) {
#pragma unroll
        for (int i = 0; i < 64; i++) c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c0, 0, 0, 0);
    } else {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c3, 0, 0, 0);
        }
    }
    asm volatile("s_nop 7\n s_nop 7" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
    const unsigned long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (c0[0] + c1[1] + c2[2] + c3[3] == 0x7fffffff) out[1] = 1;
}

__global__ void k_mfma32(unsigned long long* out, int dep)
{
    v4i a = {1, 2, 3, 4}, b = {5, 6, 7, 8};
    v16i c0, c1;
#pragma unroll
    for (int e = 0; e < 16; e++) { c0[e] = 0; c1[e] = 0; }
    const unsigned long long t0 = clock64();
    if (dep) {
#pragma unroll
        for (int i = 0; i < 64; i++) c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
    } else {
#pragma unroll
        for (int i = 0; i < 32; i++) {
            c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
        }
    }
    asm volatile("s_nop 7\n s_nop 7" : "+v"(c0), "+v"(c1));
    const unsigned long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (c0[0] + c1[1] == 0x7fffffff) out[1] = 1;
}

// dependent LDS / global load chains (pointer chasing): latency per access
__global__ void k_lds(unsigned long long* out, int n)
{
    __shared__ int tab[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) tab[i] = (i * 7 + 64) & 1023;
    __syncthreads();
    int p = threadIdx.x;
    const unsigned long long t0 = clock64();
    for (int i = 0; i < n; i++) p = tab[p];
    const unsigned long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (p == 0x7fffffff) out[1] = 1;
}

__global__ void k_glb(unsigned long long* out, const int* tab, int n, int stride)
{
    int p = (threadIdx.x * stride) & 0xfffff;
    const unsigned long long t0 = clock64();
    for (int i = 0; i < n; i++) p = tab[p];
    const unsigned long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (p == 0x7fffffff) out[1] = 1;
}

typedef float v2f __attribute__((ext_vector_type(2)));
// packed fp32 (v_pk_fma_f32): two values per instruction -- same issue cost as a scalar fma or twice?
__global__ void k_pk(unsigned long long* out, int dep, float seed)
{
    v2f a = {seed, seed + 1.f}, b = {seed + 2.f, seed + 3.f}, c = {seed * 0.5f, seed * 0.25f}, d = {seed + 5.f, seed + 7.f};
    const v2f m = {0.999f, 1.001f}, k = {1e-3f, -1e-3f};
    const unsigned long long t0 = clock64();
    if (dep) {
#pragma unroll
        for (int i = 0; i < 256; i++) a = __builtin_elementwise_fma(a, m, k);
    } else {
#pragma unroll
        for (int i = 0; i < 64; i++) {
            a = __builtin_elementwise_fma(a, m, k);
            b = __builtin_elementwise_fma(b, m, k);
            c = __builtin_elementwise_fma(c, m, k);
            d = __builtin_elementwise_fma(d, m, k);
        }
    }
    const unsigned long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (a.x + b.x + c.x + d.x + a.y + b.y + c.y + d.y == 12345.f) out[1] = 1;
}

// byte extraction folded into the conversion (SDWA source select) -- a full-rate instruction?
__global__ void k_sdwa(unsigned long long* out, int x)
{
    int a = x, b = x + 1, c = x + 2, d = x + 3;
    float fa, fb, fc, fd;
    const unsigned long long t0 = clock64();
    asm volatile(".rept 256\n v_cvt_f32_i32_sdwa %4, sext(%0) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n"
                 " v_cvt_f32_i32_sdwa %5, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2\n"
                 " v_cvt_f32_i32_sdwa %6, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3\n"
                 " v_cvt_f32_i32_sdwa %7, sext(%3) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0\n.endr\n"
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "=v"(fa), "=v"(fb), "=v"(fc), "=v"(fd));
    const unsigned long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (fa + fb + fc + fd == 12345.f) out[1] = 1;
}

int main()
{
    unsigned long long* d; CK(hipMalloc(&d, 64));
    unsigned long long h[2];
    auto report = [&](const char* name, const char* kind, int count) {
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
        printf("%-34s %-14s %8.2f cycles/instr\n", name, kind, (double)h[0] / count);
    };
    printf("one wave per SIMD (256-thread blocks x 256), shader clock cycles (s_memtime)\n");
#define RUN(ID, N) \
    for (int r = 0; r < 2; r++) hipLaunchKernelGGL(k_dep_##ID, dim3(256), dim3(256), 0, 0, d, 1); report(#ID, "dependent", 1024 * N); \
    for (int r = 0; r < 2; r++) hipLaunchKernelGGL(k_ind_##ID, dim3(256), dim3(256), 0, 0, d, 1); report(#ID, "4 independent", 1024);
    RUN(add, 1) RUN(fma, 1) RUN(mul, 1) RUN(perm, 1) RUN(dot4, 1) RUN(cvtfi, 2) RUN(med3, 1) RUN(fract, 1) RUN(bfi, 1) RUN(rcp, 1) RUN(cnd, 1)
    RUN(mullo, 1) RUN(lshladd64, 1) RUN(salu, 1)
    for (int r = 0; r < 2; r++) hipLaunchKernelGGL(k_dep_mixed, dim3(256), dim3(256), 0, 0, d, 1); report("valu+salu alternating", "dependent", 2048);
    for (int dep = 1; dep >= 0; dep--) {
        for (int r = 0; r < 2; r++) hipLaunchKernelGGL(k_mfma16, dim3(256), dim3(256), 0, 0, d, dep);
        report("v_mfma_i32_16x16x64_i8", dep ? "dependent" : "4 independent", 64);
        for (int r = 0; r < 2; r++) hipLaunchKernelGGL(k_mfma32, dim3(256), dim3(256), 0, 0, d, dep);
        report("v_mfma_i32_32x32x32_i8", dep ? "dependent" : "2 independent", 64);
    }
    for (int dep = 1; dep >= 0; dep--) {
        for (int r = 0; r < 2; r++) hipLaunchKernelGGL(k_pk, dim3(256), dim3(256), 0, 0, d, dep, 1.5f);
        report("v_pk_fma_f32", dep ? "dependent" : "4 independent", 256);
    }
    for (int r = 0; r < 2; r++) hipLaunchKernelGGL(k_sdwa, dim3(256), dim3(256), 0, 0, d, 77);
    report("v_cvt_f32_i32_sdwa", "4 independent", 1024);
    for (int r = 0; r < 2; r++) hipLaunchKernelGGL(k_lds, dim3(256), dim3(256), 0, 0, d, 256);
    report("ds_read_b32 pointer chase", "latency", 256);
    int* tab; CK(hipMalloc(&tab, 4 << 20));
    {
        int* ht = (int*)malloc(4 << 20);
        for (int i = 0; i < (1 << 20); i++) ht[i] = (i * 17 + 4099) & 0xfffff;
        CK(hipMemcpy(tab, ht, 4 << 20, hipMemcpyHostToDevice));
        free(ht);
    }
    for (int stride : {1, 64}) {
        for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k_glb, dim3(256), dim3(256), 0, 0, d, tab, 128, stride);
        report(stride == 1 ? "global_load_dword chase (coalesced)" : "global_load_dword chase (scattered)", "latency", 128);
    }
    return 0;
}
