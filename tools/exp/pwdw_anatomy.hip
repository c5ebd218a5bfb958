// Stage anatomy of the fused pointwise+depthwise kernel on the device clock: the product source (pwdw.hip) compiled with
// TAMD_PWDW_STAMPS, chained 12 launches deep in a hipGraph on MobileNet-v1 layer shapes (random operands: timing only).
// Columns: us/launch (events), gap / ramp / body as in launch_chain2.hip, then wave 0's stamps since block entry:
//   loads issued | LDS zeroed + barrier | pointwise tiles done | barrier | tail done (stores issued) | stores acked
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DTAMD_PWDW_STAMPS -I../../tengine_amd/csrc -o pwdw_anatomy.bin pwdw_anatomy.hip ../../tengine_amd/csrc/direct.cc -lhsa-runtime64
#include "../../tengine_amd/csrc/pwdw.hip"

#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)

using namespace tamd;

struct Case { const char* name; int H, W, cin, C, mode, S, TH, TW, threads; };

int main()
{
    const int L = 12, reps = 50;
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

    const Case cases[] = {
        {"conv5_x  14x14 512->512 +dw s1  4x7 256", 14, 14, 512, 512, 1, 1, 4, 7, 256},
        {"conv5_x  14x14 512->512 +dw s1  4x7 512", 14, 14, 512, 512, 1, 1, 4, 7, 512},
        {"conv5_x  14x14 512->512 +dw s1  7x14 256", 14, 14, 512, 512, 1, 1, 7, 14, 256},
        {"conv5_x  14x14 512->512 +dw s1  7x14 512", 14, 14, 512, 512, 1, 1, 7, 14, 512},
        {"conv5_x  14x14 512->512 +dw s1  14x14 512", 14, 14, 512, 512, 1, 1, 14, 14, 512},
        {"conv5_x  14x14 512->512 +dw s1  2x14 256", 14, 14, 512, 512, 1, 1, 2, 14, 256},
        {"conv2_1  112x112 32->64 +dw s2  7x8 512", 112, 112, 32, 64, 1, 2, 7, 8, 512},
        {"conv2_1  112x112 32->64 +dw s2  4x14 256", 112, 112, 32, 64, 1, 2, 4, 14, 256},
        {"conv3_2  28x28 128->256 +dw s1  7x7 512", 28, 28, 128, 256, 1, 1, 7, 7, 512},
        {"conv5_6  7x7 512->1024 +dw s1   4x4 256", 7, 7, 512, 1024, 1, 1, 4, 4, 256},
        {"conv5_6  7x7 512->1024 +dw s1   7x7 256", 7, 7, 512, 1024, 1, 1, 7, 7, 256},
        {"conv6    7x7 1024->1024 +pool     256", 7, 7, 1024, 1024, 0, 1, 1, 1, 256},
        {"conv6    7x7 1024->1024 +pool     512", 7, 7, 1024, 1024, 0, 1, 1, 1, 512},
    };
    printf("%-44s %8s %6s %6s %6s | %6s %6s %6s %6s %6s %6s\n", "case", "us/lnch", "gap", "ramp", "body", "issued", "zeroed", "pw", "barr", "tail", "acked");
    for (const Case& c : cases) {
        PwDwArgs a{};
        a.wf = wf; a.bias = bias; a.wscale = scale; a.rq = {0.02f, 0.f, 6.f, 0.05f, 128.25f, 248.75f, 0x1p-13f, scale};
        a.N = 1; a.H = c.H; a.W = c.W; a.cs_in = (c.cin + 15) / 16 * 16; a.ktot = a.cs_in;
        const int real = (a.ktot + 63) / 64;
        a.steps = pwdw_steps(real); a.nsteps = (real + a.steps - 1) / a.steps * a.steps;
        a.mode = c.mode; a.dw_w = dww; a.dw_bias = bias; a.dw_wscale = scale; a.d_rq = {0.05f, 0.f, 12.f, 0.1f, 128.25f, 248.75f, 0x1p-13f, scale};
        a.slices = (c.C + 15) / 16; a.cw = a.slices * 16;
        a.S = c.S; a.PH = a.PW = 1; a.OH = (c.H - 1) / c.S + 1; a.OW = (c.W - 1) / c.S + 1;
        a.ldc = a.cw; a.c_off = 0; a.c_limit = a.cw;
        a.TH = std::min(c.TH, a.OH); a.TW = std::min(c.TW, a.OW);
        a.tiles_y = (a.OH + a.TH - 1) / a.TH; a.tiles_x = (a.OW + a.TW - 1) / a.TW;
        a.RH = (a.TH - 1) * a.S + 3; a.RW = (a.TW - 1) * a.S + 3;
        if (c.mode == 0) { a.OH = a.OW = 1; a.TH = a.TW = 1; a.tiles_x = a.tiles_y = 1; a.RH = c.H; a.RW = c.W; a.pool_method = 1; a.p_in_scale = 0.05f; a.p_out_scale = 0.04f; }
        if (!pwdw_config_ok(a, c.threads)) { printf("%-44s config rejected\n", c.name); continue; }
        const int nblocks = a.slices * (c.mode == 0 ? 1 : a.tiles_x * a.tiles_y);
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < L; i++) {
            PwDwArgs b = a;
            b.x = (i & 1) ? xb : xa; b.y = (i & 1) ? xa : xb;
            b.stamps = stamps + (size_t)i * 4096 * 8;
            CK(launch_pwdw(b, c.threads, st));
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
        std::vector<unsigned long long> h((size_t)L * 4096 * 8);
        CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
        double gap = 0, ramp = 0, body = 0, stage[8] = {0};
        unsigned long long prev_exit = 0;
        for (int i = 0; i < L; i++) {
            unsigned long long e_min = ~0ull, e_max = 0, x_max = 0;
            double sacc[8] = {0};
            for (int b = 0; b < nblocks; b++) {
                const unsigned long long* s = &h[((size_t)i * 4096 + b) * 8];
                e_min = std::min(e_min, s[0]); e_max = std::max(e_max, s[0]); x_max = std::max(x_max, s[7]);
                for (int k = 1; k < 8; k++) sacc[k] += (s[k] >= s[0]) ? (double)(s[k] - s[0]) : 0.0;
            }
            if (i > 0) gap += (double)(e_min - prev_exit);
            ramp += (double)(e_max - e_min); body += (double)(x_max - e_min);
            for (int k = 1; k < 8; k++) stage[k] += sacc[k] / nblocks;
            prev_exit = x_max;
        }
        printf("%-44s %8.2f %6.2f %6.2f %6.2f | %6.2f %6.2f %6.2f %6.2f %6.2f %6.2f   (%d blocks)\n", c.name, 1e3 * ms / reps / L, gap / (L - 1) / 100.0,
               ramp / L / 100.0, body / L / 100.0, stage[1] / L / 100.0, stage[2] / L / 100.0, stage[3] / L / 100.0, stage[4] / L / 100.0,
               stage[5] / L / 100.0, stage[7] / L / 100.0, nblocks);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
