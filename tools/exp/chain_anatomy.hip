// Anatomy of the chained launch (pwdw_chain_kernel) on the device clock: L identical MobileNet-v1 layers in ONE launch, the
// product source compiled with TAMD_PWDW_STAMPS.  Per layer (averaged over its blocks, microseconds since the first block of
// the launch entered): entry of its first / last block, when its blocks left the producer wait, pointwise done, tail done,
// stores acknowledged, counter bumped; "done" = the last block's counter bump, i.e. what the next layer waits for.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DTAMD_PWDW_STAMPS -DTAMD_PWDW_CHAIN_EXPERIMENT -I../../tengine_amd/csrc -o chain_anatomy.bin chain_anatomy.hip ../../tengine_amd/csrc/direct.cc -lhsa-runtime64
#include "../../tengine_amd/csrc/pwdw.hip"

#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)

using namespace tamd;

struct Case { const char* name; int H, W, cin, C, S, TH, TW, threads; };

int main()
{
    const int L = 8, reps = 30;
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    int8_t *xa, *xb, *wf, *dww; int* bias; float* scale; unsigned long long* stamps;
    const size_t big = 8 << 20;
    CK(hipMalloc(&xa, big)); CK(hipMalloc(&xb, big)); CK(hipMalloc(&wf, big)); CK(hipMalloc(&dww, 1 << 20));
    CK(hipMalloc(&bias, 1 << 16)); CK(hipMalloc(&scale, 1 << 16));
    CK(hipMalloc(&stamps, (size_t)L * 4096 * 8 * 8));
    CK(hipMemset(xa, 3, big)); CK(hipMemset(xb, 3, big)); CK(hipMemset(wf, 1, big)); CK(hipMemset(dww, 1, 1 << 20));
    CK(hipMemset(bias, 0, 1 << 16));
    std::vector<float> sc(1 << 14, 0.001f);
    CK(hipMemcpy(scale, sc.data(), 1 << 16, hipMemcpyHostToDevice));
    PwDwArgs* dlayers; int *dflags, *dsync;
    CK(hipMalloc(&dlayers, 16 * sizeof(PwDwArgs))); CK(hipMalloc(&dflags, 16 * 32 * 4)); CK(hipMalloc(&dsync, 16));

    const Case cases[] = {
        {"conv5_x 14x14 512->512 dw s1 4x7 t256", 14, 14, 512, 512, 1, 4, 7, 256},
        {"conv5_x 14x14 512->512 dw s1 7x14 t256", 14, 14, 512, 512, 1, 7, 14, 256},
        {"conv5_x 14x14 512->512 dw s1 14x14 t256", 14, 14, 512, 512, 1, 14, 14, 256},
        {"conv5_x 14x14 512->512 dw s1 4x7 t512", 14, 14, 512, 512, 1, 4, 7, 512},
        {"conv3_x 56x56 128->128 dw s1 8x8 t256", 56, 56, 128, 128, 1, 8, 8, 256},
        {"conv6_x 7x7 1024->1024 dw s1 7x7 t256", 7, 7, 1024, 1024, 1, 7, 7, 256},
    };
    for (const Case& c : cases) {
        PwDwArgs a{};
        a.wf = wf; a.bias = bias; a.wscale = scale; a.rq = {0.02f, 0.f, 6.f, 0.05f, 128.25f, 248.75f, 0x1p-13f, scale};
        a.N = 1; a.H = c.H; a.W = c.W; a.cs_in = (c.cin + 15) / 16 * 16; a.ktot = a.cs_in;
        const int real = (a.ktot + 63) / 64;
        a.steps = pwdw_steps(real); a.nsteps = (real + a.steps - 1) / a.steps * a.steps;
        a.mode = 1; a.dw_w = dww; a.dw_bias = bias; a.dw_wscale = scale; a.d_rq = {0.05f, 0.f, 12.f, 0.1f, 128.25f, 248.75f, 0x1p-13f, scale};
        a.slices = (c.C + 15) / 16; a.cw = a.slices * 16;
        a.S = c.S; a.PH = a.PW = 1; a.OH = (c.H - 1) / c.S + 1; a.OW = (c.W - 1) / c.S + 1;
        a.ldc = a.cw; a.c_off = 0; a.c_limit = a.cw;
        a.TH = std::min(c.TH, a.OH); a.TW = std::min(c.TW, a.OW);
        a.tiles_y = (a.OH + a.TH - 1) / a.TH; a.tiles_x = (a.OW + a.TW - 1) / a.TW;
        a.RH = (a.TH - 1) * a.S + 3; a.RW = (a.TW - 1) * a.S + 3;
        a.stamps = stamps;
        if (!pwdw_config_ok(a, c.threads)) { printf("%-44s config rejected\n", c.name); continue; }
        PwChainArgs ch{};
        std::vector<PwDwArgs> layers;
        int gx, gy, gz;
        for (int i = 0; i < L; i++) {
            PwDwArgs b = a;
            b.x = (i & 1) ? xb : xa; b.y = (i & 1) ? xa : xb;
            ch.variant[i] = (short)pwdw_chain_variant(b, c.threads, &gx, &gy, &gz);
            ch.gx[i] = (short)gx; ch.gy[i] = (short)gy;
            ch.first_block[i + 1] = ch.first_block[i] + gx * gy * gz;
            layers.push_back(b);
        }
        ch.nlayers = L; ch.layers = dlayers; ch.flags = dflags; ch.sync = dsync;
        const int epoch0[4] = {1, 0, 0, 0};
        CK(hipMemcpy(dlayers, layers.data(), L * sizeof(PwDwArgs), hipMemcpyHostToDevice));
        CK(hipMemset(dflags, 0, 16 * 32 * 4)); CK(hipMemcpy(dsync, epoch0, 16, hipMemcpyHostToDevice));
        const size_t lds = pwdw_lds_bytes(a, c.threads);
        const int per = ch.first_block[1], total = ch.first_block[L];
        for (int i = 0; i < 5; i++) CK(launch_pwdw_chain(ch, c.threads, lds, st));
        CK(hipStreamSynchronize(st));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; i++) CK(launch_pwdw_chain(ch, c.threads, lds, st));
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        // stand-alone launches of the same layers, for comparison
        hipEvent_t f0, f1; CK(hipEventCreate(&f0)); CK(hipEventCreate(&f1));
        for (int i = 0; i < L; i++) { PwDwArgs b = layers[i]; b.stamps = nullptr; CK(launch_pwdw(b, c.threads, st)); }
        CK(hipEventRecord(f0, st));
        for (int r = 0; r < reps; r++)
            for (int i = 0; i < L; i++) { PwDwArgs b = layers[i]; b.stamps = nullptr; CK(launch_pwdw(b, c.threads, st)); }
        CK(hipEventRecord(f1, st));
        CK(hipEventSynchronize(f1));
        float ms2; CK(hipEventElapsedTime(&ms2, f0, f1));
        std::vector<unsigned long long> h((size_t)total * 8);
        int sy[4];
        CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(sy, dsync, 16, hipMemcpyDeviceToHost));
        printf("%s: %d blocks/layer x %d layers; chained %.2f us/layer, %d eager launches %.2f us/layer; wait errors %d\n", c.name, per, L,
               1e3 * ms / reps / L, L, 1e3 * ms2 / reps / L, sy[2]);
        unsigned long long t0 = ~0ull;
        for (int b = 0; b < total; b++) t0 = std::min(t0, h[(size_t)b * 8]);
        printf("  layer  first-in  last-in | avg: waited-till  pw-done  tail-done  acked  bumped | layer done\n");
        for (int i = 0; i < L; i++) {
            double fi = 1e30, li = 0, s2 = 0, s3 = 0, s5 = 0, s7 = 0, s6 = 0, done = 0;
            for (int b = ch.first_block[i]; b < ch.first_block[i + 1]; b++) {
                const unsigned long long* s = &h[(size_t)b * 8];
                auto us = [&](int k) { return (double)(s[k] - t0) / 100.0; };
                fi = std::min(fi, us(0)); li = std::max(li, us(0));
                s2 += us(2); s3 += us(3); s5 += us(5); s7 += us(7); s6 += us(6); done = std::max(done, us(6));
            }
            printf("  %5d  %8.2f  %7.2f |      %11.2f  %7.2f  %9.2f  %5.2f  %6.2f | %8.2f\n", i, fi, li, s2 / per, s3 / per, s5 / per, s7 / per, s6 / per, done);
        }
    }
    return 0;
}
