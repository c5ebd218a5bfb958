// Stage anatomy of the fused depthwise -> pointwise kernel (dwpw.hip) on the 100 MHz wall clock: the product source compiled with
// TAMD_DWPW_STAMPS on MobileNet-v1's conv5_x shape at batch 64 (14x14, 512 -> 512; random operands: timing only), with ablation
// switches (no MFMAs / no depthwise arithmetic / no epilogue requantisation) to see what each phase costs.
// Columns: us per launch (events, 20 dependent launches), then wave 0's stamps since block entry, averaged over the blocks:
//   consts in LDS | stage 0 produced | [stage k multiplied (mma issued) / stage k+1 produced + barrier] x 3 | last mma | stores issued
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm --amdgpu-mfma-vgpr-form -DTAMD_DWPW_STAMPS -I../../tengine_amd/csrc -o dwpw_anatomy.bin dwpw_anatomy.hip
#include "../../tengine_amd/csrc/dwpw.hip"

#include <stdio.h>
#include <stdlib.h>

#include <vector>

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)

namespace tamd {
thread_local std::vector<LaunchRec>* g_launch_rec = nullptr;
thread_local bool g_launch_coherent = false;
thread_local bool g_launch_beside = false;
}
using namespace tamd;

int main(int argc, char** argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 64, HW = argc > 2 ? atoi(argv[2]) : 14, C = argc > 3 ? atoi(argv[3]) : 512, COUT = argc > 4 ? atoi(argv[4]) : 512;
    const int L = 20, reps = 20;
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const size_t tens = (size_t)N * HW * HW * 512 + 65536;
    int8_t *xa, *xb, *wf, *dww; int* bias; float* scale; unsigned long long* stamps;
    CK(hipMalloc(&xa, tens)); CK(hipMalloc(&xb, tens)); CK(hipMalloc(&wf, 1 << 20)); CK(hipMalloc(&dww, 1 << 16));
    CK(hipMalloc(&bias, 1 << 16)); CK(hipMalloc(&scale, 1 << 16));
    const int blocks = (N * HW + 3) / 4;
    CK(hipMalloc(&stamps, (size_t)blocks * 16 * 8));
    CK(hipMemset(xa, 3, tens)); CK(hipMemset(xb, 3, tens)); CK(hipMemset(wf, 1, 1 << 20)); CK(hipMemset(dww, 1, 1 << 16));
    CK(hipMemset(bias, 0, 1 << 16));
    if (getenv("DWPW_RANDOM")) {            // random operands (the memset ones never toggle a multiplier input)
        std::vector<int8_t> r(tens);
        unsigned s = 12345u;
        for (auto& v : r) { s = s * 1664525u + 1013904223u; v = (int8_t)(s >> 24); }
        CK(hipMemcpy(xa, r.data(), tens, hipMemcpyHostToDevice)); CK(hipMemcpy(xb, r.data(), tens, hipMemcpyHostToDevice));
        CK(hipMemcpy(wf, r.data(), 1 << 20, hipMemcpyHostToDevice));
        for (size_t i = 3; i < (1 << 16); i += 4) r[i] = 0;      // {w0, w1, w2, 0}
        CK(hipMemcpy(dww, r.data(), 1 << 16, hipMemcpyHostToDevice));
    }
    std::vector<float> sc(1 << 14, 0.001f);
    CK(hipMemcpy(scale, sc.data(), 1 << 16, hipMemcpyHostToDevice));

    DwPwArgs a{};
    a.dw_w = dww; a.dw_bias = bias; a.dw_wscale = scale; a.dw_rq = {0.05f, 0.f, 12.f, 0.1f, 128.25f, 248.75f, 0x1p-13f, scale};
    a.pw_wfrag = wf; a.pw_bias = bias; a.pw_wscale = scale; a.pw_rq = {0.02f, 0.f, 6.f, 0.05f, 128.25f, 248.75f, 0x1p-13f, scale};
    a.N = N; a.H = a.W = a.OH = a.OW = HW; a.C = C; a.cs_in = a.cw = (C + 15) / 16 * 16; a.PH = a.PW = 1;
    a.cout = COUT; a.ldc = COUT; a.c_off = 0; a.c_limit = COUT;
    printf("dwpw %dx%dx%d -> %d, batch %d: %d blocks of 512 threads, %zu B of LDS\n", HW, HW, C, COUT, N, blocks, dwpw_lds_bytes(a.cw, COUT / 64));
    const char* names[] = {"full", "no MFMAs", "no depthwise arithmetic", "no epilogue requantisation", "no MFMAs, no depthwise", "none of the three",
                           "full, no A loads", "full, no tap loads", "full, no B reads", "none, no A loads", "none, no tap loads", "none, no A / tap loads"};
    const int abl[] = {0, 1, 2, 4, 3, 7, 8, 16, 32, 7 + 8, 7 + 16, 7 + 8 + 16};
    printf("%-28s %8s | %6s %6s | %6s %6s %6s %6s %6s %6s | %6s %6s\n", "variant", "us/lnch", "consts", "st0", "mma0", "bar1", "mma1", "bar2", "mma2", "bar3", "mma3", "stored");
    for (int v = 0; v < 12; v++) {
        a.ablate = abl[v];
        auto run = [&](int i, unsigned long long* s) {
            DwPwArgs b = a;
            b.x = (i & 1) ? xb : xa; b.y = (i & 1) ? xa : xb; b.stamps = s;
            CK(launch_dwpw(b, st));
        };
        for (int i = 0; i < 4; i++) run(i, nullptr);
        CK(hipStreamSynchronize(st));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; r++)
            for (int i = 0; i < L; i++) run(i, nullptr);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        run(0, stamps); run(1, stamps);
        CK(hipStreamSynchronize(st));
        std::vector<unsigned long long> h((size_t)blocks * 16);
        CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
        double s[16] = {0};
        unsigned long long e_min = ~0ull, x_max = 0;
        for (int b = 0; b < blocks; b++) {
            const unsigned long long* p = &h[(size_t)b * 16];
            e_min = std::min(e_min, p[0]); x_max = std::max(x_max, p[7]);
            for (int k = 1; k < 16; k++) s[k] += p[k] >= p[0] ? (double)(p[k] - p[0]) / 100.0 : 0.0;
        }
        for (int k = 1; k < 16; k++) s[k] /= blocks;
        printf("%-28s %8.2f | %6.2f %6.2f | %6.2f %6.2f %6.2f %6.2f %6.2f %6.2f | %6.2f %6.2f   (first entry to last exit %.2f us)\n", names[v], 1e3 * ms / reps / L, s[1], s[2],
               s[8], s[3], s[9], s[4], s[10], s[5], s[6], s[7], (double)(x_max - e_min) / 100.0);
    }
    return 0;
}
