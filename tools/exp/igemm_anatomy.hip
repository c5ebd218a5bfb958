// Stage anatomy of the software-pipelined int8 implicit GEMM (conv_igemm_fast.h compiled with TAMD_IGEMM_STAMPS): one
// ResNet-50 3x3 layer at batch 32 (res3x_branch2b: 32 x 28 x 28 x 128 -> 128, K = 1152), random operands, per tile shape:
// launch time (events, 20 back-to-back), shader-clock stamps of wave 0 of block 0 (prologue, every stage, epilogue), and
// ablations: no MFMA / no LDS traffic / no global loads -- which resource the ~15 us of every variant is spent on.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DTAMD_IGEMM_STAMPS -I../../tengine_amd/csrc -o igemm_anatomy.bin igemm_anatomy.hip ../../tengine_amd/csrc/direct.cc -lhsa-runtime64
#include "conv_igemm_fast.h"

#include <stdio.h>
#include <stdlib.h>

#include <vector>

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)
using namespace tamd;

template <int BM, int BN, int WM, int WN, int D>
static void run(const char* name, ConvArgs a, hipStream_t st, long long* dstamps)
{
    const int tiles_n = (a.cout + BN - 1) / BN, tiles_m = (a.M + BM - 1) / BM;
    const int grid = ((tiles_m + 7) / 8) * 8 * tiles_n;
    const size_t lds = 2 * (size_t)(BM + BN) * 64;
    for (int flags : {0, 1, 2, 4, 3, 6, 7}) {
        a.dbg_flags = flags; a.dbg_stamps = dstamps;
        CK(hipMemsetAsync(dstamps, 0, 64 * 8, st));
        for (int i = 0; i < 3; i++) hipLaunchKernelGGL((conv_igemm_fast_i8_kernel<BM, BN, WM, WN, false, D>), dim3(grid), dim3(256), lds, st, a);
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 20; i++) hipLaunchKernelGGL((conv_igemm_fast_i8_kernel<BM, BN, WM, WN, false, D>), dim3(grid), dim3(256), lds, st, a);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        long long h[64];
        CK(hipMemcpy(h, dstamps, sizeof(h), hipMemcpyDeviceToHost));
        const int nk = (a.kpad / 64 + D - 1) / D * D;
        printf("%-22s flags %d (%s%s%s)  %7.2f us/launch  %4d blocks | prologue %5lld  stages:", name, flags, flags & 1 ? "noMFMA " : "", flags & 2 ? "noLDS " : "",
               flags & 4 ? "noLOAD" : "", 1e3 * ms / 20, grid, h[2] - h[0]);
        for (int k = 0; k < nk && k < 18; k++) printf(" %lld", h[3 + k] - h[2 + k]);
        printf(" | epilogue %lld  total %lld cycles\n", h[3 + nk] - h[2 + nk], h[3 + nk] - h[0]);
    }
}

int main()
{
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int N = 32, H = 28, W = 28, C = 128, CO = 128;
    ConvArgs a{};
    int8_t *x, *w, *y, *z; int* bias; float* sc; long long* ds;
    const size_t xb = (size_t)N * H * W * C + 4096, wb = (size_t)256 * 1152 + 4096;
    CK(hipMalloc(&x, xb)); CK(hipMalloc(&w, wb)); CK(hipMalloc(&y, (size_t)N * H * W * CO + 4096)); CK(hipMalloc(&z, 4096));
    CK(hipMalloc(&bias, 4096)); CK(hipMalloc(&sc, 4096)); CK(hipMalloc(&ds, 64 * 8));
    CK(hipMemset(x, 1, xb)); CK(hipMemset(w, 1, wb)); CK(hipMemset(z, 0, 4096)); CK(hipMemset(bias, 0, 4096));
    std::vector<float> s1(1024, 0.001f);
    CK(hipMemcpy(sc, s1.data(), 4096, hipMemcpyHostToDevice));
    a.x = x; a.w = w; a.bias = bias; a.wscale = sc; a.y = y; a.zeros = z;
    a.N = N; a.H = H; a.W = W; a.cs_in = C; a.ckp = C; a.OH = H; a.OW = W; a.cout = CO; a.ldc = CO; a.c_off = 0; a.c_limit = CO;
    a.KH = a.KW = 3; a.SH = a.SW = 1; a.PH = a.PW = 1; a.DH = a.DW = 1; a.cin = C; a.ktot = 9 * C; a.kpad = 9 * C; a.M = N * H * W;
    a.rq = {0.02f, 0.f, 63.7f, 0.5f, 128.25f, 255.75f, 0x1p-13f, sc}; a.cfg = -1;
    a.mg_ohw = ((1ull << 40) + H * W - 1) / (H * W); a.mg_ow = ((1ull << 40) + W - 1) / W;
    run<64, 64, 2, 2, 3>("64x64 ring3", a, st, ds);
    run<128, 64, 2, 2, 3>("128px x 64co ring3", a, st, ds);
    run<128, 128, 2, 2, 3>("128x128 ring3", a, st, ds);
    run<256, 128, 2, 2, 3>("256px x 128co ring3", a, st, ds);
    return 0;
}
