// Anatomy of conv_pgemm.hip (compiled with TAMD_IGEMM_STAMPS): ResNet-50 layer shapes at batch 32, random operands.  Per tile
// variant and ablation: launch time (events, 20 back to back), and from the stamps wave 0 of EVERY block leaves -- shader clock
// for the phases of a block (setup, first loads landing, K loop, epilogue), the 100 MHz wall clock for when blocks start and end
// relative to the launch's first block (dispatch ramp, tail) -- the question being where the ~12 us of a 1.5 us (MFMA) layer go.
// Round 5: the wave-grid kernels of conv_pgemm_w.hip (variants 16 ..) ride in the same loop with their geometry table; flag 64 | n << 16
// delays the second dispatch round of every XCD by n x 1024 cycles (the phase-skew experiment); argv[2] = "3x3" runs the four 3x3
// shapes of ResNet-50 only.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm --amdgpu-mfma-vgpr-form -DTAMD_IGEMM_STAMPS -DTAMD_EXPERIMENTS -I../../tengine_amd/csrc
//        -o pgemm_anatomy.bin pgemm_anatomy.hip ../../tengine_amd/csrc/direct.cc -lhsa-runtime64
#include "../../tengine_amd/csrc/conv_pgemm.hip"
#include "../../tengine_amd/csrc/conv_pgemm_w.hip"

#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <string>
#include <vector>

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)
using namespace tamd;

static long long pct(std::vector<long long> v, double p) { std::sort(v.begin(), v.end()); return v.empty() ? 0 : v[(size_t)(p * (v.size() - 1))]; }

static void run_shape(int N, int HW, int C, int CO, int K, int S, hipStream_t st)
{
    const int P = K / 2, OHW = (HW + 2 * P - K) / S + 1;
    ConvArgs a{};
    int8_t *x, *y, *z; int* bias; float* sc; long long* ds;
    const int ckp = (C + 15) / 16 * 16, ktot = K * K * ckp, kpad = (ktot + 63) / 64 * 64, cout_pad = (CO + 127) / 128 * 128;
    const size_t xb = (size_t)N * HW * HW * ckp + 4096;
    CK(hipMalloc(&x, xb)); CK(hipMalloc(&y, (size_t)N * OHW * OHW * cout_pad + 4096)); CK(hipMalloc(&z, 4096));
    CK(hipMalloc(&bias, cout_pad * 4 + 4096)); CK(hipMalloc(&sc, cout_pad * 4 + 4096));
    const int max_blocks = 16384;
    CK(hipMalloc(&ds, (size_t)max_blocks * 64));
    std::vector<int8_t> hx(xb), hw((size_t)cout_pad * kpad);
    srand(1);
    for (auto& v : hx) v = (int8_t)(rand() % 255 - 127);
    for (auto& v : hw) v = (int8_t)(rand() % 255 - 127);
    CK(hipMemcpy(x, hx.data(), xb, hipMemcpyHostToDevice));
    CK(hipMemset(z, 0, 4096)); CK(hipMemset(bias, 0, cout_pad * 4));
    std::vector<float> s1(cout_pad + 16, 0.0001f);
    CK(hipMemcpy(sc, s1.data(), cout_pad * 4, hipMemcpyHostToDevice));
    a.x = x; a.w = nullptr; a.bias = bias; a.wscale = sc; a.y = y; a.zeros = z;
    a.N = N; a.H = HW; a.W = HW; a.cs_in = ckp; a.ckp = ckp; a.OH = OHW; a.OW = OHW; a.cout = CO; a.ldc = cout_pad; a.c_off = 0; a.c_limit = CO;
    a.KH = a.KW = K; a.SH = a.SW = S; a.PH = a.PW = P; a.DH = a.DW = 1; a.cin = C; a.ktot = ktot; a.kpad = kpad; a.M = N * OHW * OHW;
    a.rq = {0.02f, 0.f, 63.7f, 0.5f, 128.25f, 255.75f, 0x1p-13f, sc}; a.cfg = -1;
    a.mg_ohw = ((1ull << 40) + OHW * OHW - 1) / (OHW * OHW); a.mg_ow = ((1ull << 40) + OHW - 1) / OHW;
    printf("=== %d x %d x %d^2 -> %d, k%d s%d: M %d, K %d, %.0f MMAC\n", N, C, HW, CO, K, S, a.M, ktot, 1e-6 * a.M * (double)CO * C * K * K);
    int8_t* packed[2] = {nullptr, nullptr};
    for (int v = 0; v < conv_pgemm_num_variants(); v++) {
        if (!conv_pgemm_applicable(a, v)) continue;
        ConvArgs ap = a;
        conv_pgemm_prepare(ap, v);
        const int bn = conv_pgemm_bn(v), slot = bn == 128;
        if (!packed[slot]) {
            std::vector<int8_t> wf(conv_pgemm_packed_bytes(ap, bn), 0);
            conv_pgemm_pack(ap, hw.data(), cout_pad, bn, wf.data());
            CK(hipMalloc(&packed[slot], wf.size()));
            CK(hipMemcpy(packed[slot], wf.data(), wf.size(), hipMemcpyHostToDevice));
        }
        ap.wfrag = packed[slot];
        int* dtab = nullptr;
        if (v & 16) {
            std::vector<int> tab;
            conv_pgemm_w_table(ap, tab);
            CK(hipMalloc(&dtab, tab.size() * 4));
            CK(hipMemcpy(dtab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
            ap.pg_tab = dtab;
        }
        const int bm = (v & 2) ? 64 : 128;
        const int tiles = ((ap.M + bm - 1) / bm) * ((CO + bn - 1) / bn);
        const int grid = (((ap.M + bm - 1) / bm + 7) / 8) * 8 * ((CO + bn - 1) / bn);
#ifdef TAMD_PG_ABLATE
        for (int flags : {0, 2, 2 | 8, 2 | 16, 2 | 4, 2 | 32, 2 | 1, 63}) {
#else
        // TAMD_ANATOMY_SKEW=1: the phase-skew experiment (profiles/r05_pgemm_anatomy_skew.txt)
        for (int flags : std::vector<int>((v & 16) && !(v & 8) && getenv("TAMD_ANATOMY_SKEW") ? std::vector<int>{0, 64 | (2 << 16), 64 | (4 << 16)} : std::vector<int>{0})) {
#endif
            ap.dbg_flags = flags; ap.dbg_stamps = nullptr;
            for (int i = 0; i < 3; i++) CK(launch_conv_pgemm(ap, st));
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < 20; i++) CK(launch_conv_pgemm(ap, st));
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            // one stamped launch
            ap.dbg_stamps = ds;
            CK(hipMemsetAsync(ds, 0, (size_t)max_blocks * 64, st));
            CK(launch_conv_pgemm(ap, st));
            CK(hipStreamSynchronize(st));
            const int reps = ((flags >> 8) & 0xff) + 1;
            std::vector<long long> hall((size_t)grid * 8 * reps);
            CK(hipMemcpy(hall.data(), ds, hall.size() * 8, hipMemcpyDeviceToHost));
            for (int rep = 0; rep < reps; rep++) {
            std::vector<long long> h(hall.begin() + (size_t)rep * grid * 8, hall.begin() + (size_t)(rep + 1) * grid * 8);
            std::vector<long long> setup, land, loop, epi, total, wstart, wend;
            long long w0 = -1;
            for (int b = 0; b < grid; b++) if (h[b * 8 + 5]) w0 = (w0 < 0 || h[b * 8] < w0) ? h[b * 8] : w0;
            int cus = 0;
            std::vector<long long> ids;
            for (int b = 0; b < grid; b++) {
                const long long* s = &h[(size_t)b * 8];
                if (!s[5]) continue;
                setup.push_back(s[2] - s[1]); land.push_back(s[3] - s[2]); loop.push_back(s[4] - s[3]); epi.push_back(s[5] - s[4]); total.push_back(s[5] - s[1]);
                wstart.push_back((s[0] - w0) * 10); wend.push_back((s[6] - w0) * 10);      // ns
                ids.push_back(s[7] & 0xffffffff00000f00ll);      // xcc id | (cu id, se id ..) bits 8-11 of HW_ID: enough to count distinct CUs roughly
            }
            std::sort(ids.begin(), ids.end());
            cus = (int)(std::unique(ids.begin(), ids.end()) - ids.begin());
            printf("%-30s pass %d flags %d (%s%s%s) %7.2f us/launch %4d tiles ns %d | cycles med/max: setup %lld/%lld land %lld/%lld loop %lld/%lld (%.0f/stage) epi %lld/%lld total %lld/%lld | wall ns: start med/max %lld/%lld end med/max %lld/%lld\n",
                   conv_pgemm_kernel_name(ap), rep, flags, flags & 1 ? "noMFMA " : "", flags & 2 ? "noEPI " : "", (std::string(flags & 4 ? "noLOAD " : "") + (flags & 8 ? "noBAR " : "") + (flags & 16 ? "noLDSRD " : "") + (flags & 32 ? "noWAIT" : "")).c_str(), 1e3 * ms / 20, tiles, ap.pg_ns,
                   pct(setup, .5), pct(setup, 1), pct(land, .5), pct(land, 1), pct(loop, .5), pct(loop, 1), (double)pct(loop, .5) / ap.pg_ns, pct(epi, .5), pct(epi, 1),
                   pct(total, .5), pct(total, 1), pct(wstart, .5), pct(wstart, 1), pct(wend, .5), pct(wend, 1));
            }
            hipEventDestroy(e0); hipEventDestroy(e1);
        }
        if (dtab) hipFree(dtab);
    }
    hipFree(x); hipFree(y); hipFree(z); hipFree(bias); hipFree(sc); hipFree(ds);
    for (auto p : packed) if (p) hipFree(p);
}

int main(int argc, char** argv)
{
    setvbuf(stdout, nullptr, _IOLBF, 0);          // a faulting kernel must not take the lines before it along
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int B = argc > 1 ? atoi(argv[1]) : 32;
    run_shape(B, 28, 128, 128, 3, 1, st);
    if (argc > 2 && std::string(argv[2]) == "3x3") {
        run_shape(B, 14, 256, 256, 3, 1, st);
        run_shape(B, 7, 512, 512, 3, 1, st);
        run_shape(B, 56, 64, 64, 3, 1, st);
        return 0;
    }
    if (argc > 2) return 0;
    run_shape(B, 14, 256, 256, 3, 1, st);
    run_shape(B, 56, 64, 64, 3, 1, st);
    run_shape(B, 14, 1024, 256, 1, 1, st);
    run_shape(B, 14, 256, 1024, 1, 1, st);
    return 0;
}
