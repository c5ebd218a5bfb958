// Planner for uint8 (per-tensor asymmetric) device graphs.
//
// The reference simulates uint8 in fp32 (see u8_kernels.hip); the device keeps every activation as a dense
// NCHW byte tensor -- the reference's own order, so graph inputs/outputs need no layout pass -- and prepares, once
// at prerun, exactly the fp32 operands the reference prepares:
//   conv (group 1)  weights -> fp32 as conv_kernel_x86.c:68-80 (interleave_uint8), transposed to [K][cout_pad];
//                   k -> (c,ky,kx) offsets of im2col_uint8 (:126-185) as a lookup table
//   conv (grouped)  weights -> fp32 as conv_kernel_ref_uint8.c:82-86, OIHW kept
//   fc              weights -> fp32 as fc_ref.c:150-160, transposed to [hidden][nout_pad]
#include <hip/hip_runtime.h>
#include "env.h"

#include <cstdlib>
#include <cstring>

#include <functional>
#include <map>
#include <mutex>

#include "graph.h"

namespace tamd {

static inline int rup(int v, int m) { return (v + m - 1) / m * m; }

static int q_of(const HTensor& t, U8Q* q, const char* what)
{
    if (t.scales.empty()) { set_error("%s %s has no quantisation parameters", what, t.name.c_str()); return -1; }
    q->scale = t.scales[0];
    q->zp = t.zps.empty() ? 0 : t.zps[0];
    return 0;
}

// `pool`: a 2x2 / stride 2 max-pool node to apply in the conv epilogue (U8PoolFuse); returns 2 when the kernel this conv gets
// cannot carry the requested fused tail (the caller plans again without it)
static int plan_conv_u8(tamd_graph* g, HNode& n, const HNode* relu = nullptr, const HNode* pool = nullptr)
{
    HTensor& x = g->tensors[n.in[0]];
    HTensor& w = g->tensors[n.in[1]];
    HTensor* b = n.in.size() > 2 ? &g->tensors[n.in[2]] : nullptr;
    HTensor& yc = g->tensors[n.out[0]];                               // the conv's own output: its quantisation parameters
    HTensor& y = relu ? g->tensors[relu->out[0]] : yc;                // where the bytes go (the fused ReLU's output)
    const tamd_conv_param& p = n.p.conv;
    if (w.dtype != TAMD_DT_UINT8 || (b && b->dtype != TAMD_DT_INT32)) { set_error("conv %s: uint8 weights / int32 bias expected", n.name.c_str()); return -1; }
    U8Q qx, qw, qy;
    if (q_of(x, &qx, "tensor") || q_of(w, &qw, "weight") || q_of(yc, &qy, "tensor")) return -1;
    U8Relu fr{};
    if (relu) { fr.on = 1; fr.slope = relu->p.relu.negative_slope; if (q_of(y, &fr.out, "tensor")) return -1; }
    const int cin_g = x.c / p.group, K = cin_g * p.kernel_h * p.kernel_w, cout = y.c;
    if ((size_t)cout * K != w.data.size()) { set_error("conv %s: weight size mismatch", n.name.c_str()); return -1; }
    const int32_t* dbias = nullptr;
    if (b) {
        std::vector<int32_t> hb((const int32_t*)b->data.data(), (const int32_t*)b->data.data() + cout);
        int32_t* d = nullptr;
        if (upload(g, hb, &d)) return -1;
        dbias = d;
    }
    Step st; st.node = n.name;
    st.macs = (double)y.n * y.h * y.w * cout * K;
    st.bytes = (double)x.n * x.c * x.h * x.w + (double)y.n * cout * y.h * y.w + 1.0 * cout * K;
    static const char* dma_env = exp_env("TAMD_U8_DMA");
    bool use_dma = dma_env && atoi(dma_env) != 0;            // measured no faster than the register-staged kernel
    // the register-staged kernel keeps the whole k -> tap table in LDS: beyond ~28k taps it does not fit next to the
    // operand tiles (160 KB per CU) and the DMA kernel (table read with scalar loads) takes over
    if (p.group == 1 && (size_t)(rup(K, 64) + 2 * (64 + 64) * 36) * 4 > 150 * 1024) use_dma = true;
    if (pool && (use_dma || p.group != 1)) return 2;       // the pooled epilogue lives in conv_u8_gemm / conv_u8_rgb3x3
    if (p.group == 1 && use_dma && relu) return 2;           // the DMA kernel has no fused ReLU tail: plan the two nodes apart
    if (p.group == 1 && use_dma && (p.kernel_h - 1) * p.dilation_h <= 15 && (p.kernel_w - 1) * p.dilation_w <= 15
        && (size_t)x.c * x.h * x.w < (1u << 24)) {
        // ---- asynchronous fp32 MFMA kernel (conv_f32_mfma.hip): fp32 copy of the input + fp32 packed weights ----
        const int Kpad = rup(K, 32), nstage = Kpad / 32;
        F32ConvArgs a{};
        a.N = x.n; a.C = x.c; a.H = x.h; a.W = x.w; a.OH = y.h; a.OW = y.w; a.cout = cout;
        a.K = K; a.Kpad = Kpad; a.SH = p.stride_h; a.SW = p.stride_w; a.PH = p.pad_h0; a.PW = p.pad_w0;
        a.tail_split = 1;
        a.cfg = conv_f32_mfma_pick(a);
        const int BM = conv_f32_mfma_bm(a.cfg), ntile = (cout + BM - 1) / BM;
        const int G = 64 / BM, NIg = 32 / G;                   // k rows per 64-float group, groups per stage
        // w_fp32 = ((float)w - (float)zp) * scale (conv_kernel_x86.c:68-80); [tile][stage][group][ (k%G)*BM + c ]
        std::vector<float> wf((size_t)ntile * nstage * 32 * BM, 0.f);
        for (int co = 0; co < cout; co++)
            for (int k = 0; k < K; k++) {
                const int r = k & 31;
                wf[(((size_t)(co / BM) * nstage + (k >> 5)) * NIg + r / G) * 64 + (r % G) * BM + co % BM] =
                    ((float)w.data[(size_t)co * K + k] - (float)qw.zp) * qw.scale;
            }
        std::vector<unsigned> lut(Kpad, 0u);
        for (int k = 0; k < K; k++) {
            const int kx = k % p.kernel_w, ky = (k / p.kernel_w) % p.kernel_h, c = k / (p.kernel_w * p.kernel_h);
            lut[k] = (unsigned)(c * x.h * x.w + ky * p.dilation_h * x.w + kx * p.dilation_w) | (unsigned)(kx * p.dilation_w) << 24
                     | (unsigned)(ky * p.dilation_h) << 28;
        }
        float* dwf = nullptr; unsigned* dlut = nullptr;
        if (upload(g, wf, &dwf) || upload(g, lut, &dlut)) return -1;
        // the fp32 copy of the input tensor (shared by every conv that reads it; refreshed once per run)
        float* xf = nullptr;
        auto it = g->f32_copy.find(n.in[0]);
        if (it == g->f32_copy.end()) {
            void* pxf = nullptr;
            if (dev_alloc(g, &pxf, x.elems() * sizeof(float), true)) return -1;
            xf = (float*)pxf;
            g->f32_copy[n.in[0]] = xf;
            const uint8_t* src = (const uint8_t*)x.dptr;
            const size_t cnt = x.elems();
            const float zp = (float)qx.zp, sc = qx.scale;
            Step dq; dq.node = x.name; dq.kernel = "dequant_u8_f32"; dq.bytes = 5.0 * cnt;
            dq.fn = [src, xf, cnt, zp, sc](hipStream_t s) { return launch_dequant_u8_f32(src, xf, cnt, zp, sc, s); };
            g->steps.push_back(dq);
        } else
            xf = it->second;
        if (!g->zero_page) { if (dev_alloc(g, &g->zero_page, 256, true)) return -1; }
        a.x = xf; a.w = dwf; a.klut = dlut; a.zeros = (const float*)g->zero_page; a.bias = dbias; a.y = (uint8_t*)y.dptr;
        a.out_img = (y.is_view ? y.cs : cout) * y.h * y.w; a.out_c0 = y.c_off;
        a.m_blocked = (cout >> 3 << 3) + (((cout - (cout >> 3 << 3)) >> 2) << 2);
        a.bias_scale = qx.scale * qw.scale;           // conv_kernel_x86.c:1723
        a.act = p.activation; a.out_scale = qy.scale; a.out_zp = qy.zp;
        st.kernel = conv_f32_mfma_kernel_name(a);
        st.bytes = 4.0 * x.elems() + (double)y.elems() + 4.0 * cout * K;
        st.fn = [a](hipStream_t s) { return launch_conv_f32_mfma(a, s); };
    } else if (p.group == 1) {
        const int Kpad = rup(K, 64), cout_pad = rup(cout, 64);     // 64: the deepest K stage of the kernel family
        U8ConvArgs a{};
        a.N = x.n; a.C = x.c; a.H = x.h; a.W = x.w; a.OH = y.h; a.OW = y.w; a.cout = cout; a.cout_pad = cout_pad;
        a.K = K; a.Kpad = Kpad; a.SH = p.stride_h; a.SW = p.stride_w; a.PH = p.pad_h0; a.PW = p.pad_w0;
        a.cfg = conv_u8_gemm_pick(a);
        if ((p.kernel_h - 1) * p.dilation_h > 15 || (p.kernel_w - 1) * p.dilation_w > 15 || (size_t)x.c * x.h * x.w >= (1u << 24)
            || conv_u8_gemm_lds(a) > 150 * 1024) {
            set_error("conv %s: kernel extent / image size / K = %d outside the packed tap table of the uint8 GEMM kernel", n.name.c_str(), K);
            return -1;
        }
        // raw bytes, [cout tile][stage][row][32 slots], slot (k%4)*8 + (k%32)/4; padding = weight zero point
        std::map<int, uint8_t*> packed;            // (BM, KC) -> device copy of the weights packed for that tile shape
        auto pack_for = [&](int cfg) -> uint8_t* {
            const int BM = conv_u8_gemm_bm(cfg), KC = conv_u8_gemm_kc(cfg), NPOS = KC / 4, nstage = rup(K, KC) / KC;
            auto it = packed.find(BM * 1000 + KC);
            if (it != packed.end()) return it->second;
            const int ntile = (cout + BM - 1) / BM;
            std::vector<uint8_t> wq((size_t)ntile * nstage * BM * KC, (uint8_t)qw.zp);
            for (int co = 0; co < cout; co++)
                for (int k = 0; k < K; k++) {
                    const int kl = k % KC;
                    wq[(((size_t)(co / BM) * nstage + k / KC) * BM + co % BM) * KC + (kl & 3) * NPOS + (kl >> 2)] = w.data[(size_t)co * K + k];
                }
            uint8_t* d = nullptr;
            if (upload(g, wq, &d)) return nullptr;
            packed[BM * 1000 + KC] = d;
            return d;
        };
        std::vector<unsigned> lut(Kpad, 0u);
        for (int k = 0; k < K; k++) {
            const int kx = k % p.kernel_w, ky = (k / p.kernel_w) % p.kernel_h, c = k / (p.kernel_w * p.kernel_h);
            lut[k] = (unsigned)(c * x.h * x.w + ky * p.dilation_h * x.w + kx * p.dilation_w) | (unsigned)(kx * p.dilation_w) << 24
                     | (unsigned)(ky * p.dilation_h) << 28;
        }
        unsigned* dlut = nullptr;
        if (upload(g, lut, &dlut)) return -1;
        a.x = (const uint8_t*)x.dptr; a.klut = dlut; a.w_scale = qw.scale; a.w_zp = (float)qw.zp; a.bias = dbias; a.y = (uint8_t*)y.dptr;
        a.out_img = (y.is_view ? y.cs : cout) * y.h * y.w; a.out_c0 = y.c_off;
        a.m_blocked = (cout >> 3 << 3) + (((cout - (cout >> 3 << 3)) >> 2) << 2);
        a.in_scale = qx.scale; a.in_zp = (float)qx.zp;
        a.bias_scale = qx.scale * qw.scale;           // conv_kernel_x86.c:1723
        a.act = p.activation; a.out_scale = qy.scale; a.out_zp = qy.zp; a.relu = fr;
        {   // the integer path's requantisation constants (u8_epilogue.h: u8i_requant), binary32 like everything around them
            const float bs = qx.scale * qw.scale;
            a.i_m = bs / qy.scale;
            a.i_qlo = 0; a.i_qhi = 255;
            if (p.activation >= 0) a.i_qlo = std::min(std::max(qy.zp, 0), 255);
            if (p.activation > 0) a.i_qhi = std::min(255, std::max(a.i_qlo, (int)roundf(6.0f / qy.scale) + qy.zp));
        }
        if (pool) {
            HTensor& yp = g->tensors[pool->out[0]];
            a.pool.on = 1;
            a.pool.y = (uint8_t*)yp.dptr;
            a.pool.out_img = (yp.is_view ? yp.cs : cout) * yp.h * yp.w; a.pool.out_c0 = yp.c_off;
            if (q_of(y, &a.pool.in, "tensor") || q_of(yp, &a.pool.out, "tensor")) return -1;
            a.pool.write_full = count_consumers(g, pool->in[0]) > 1;
            for (auto& io : g->outputs) a.pool.write_full |= (io.tensor == pool->in[0]);
            st.bytes += (double)yp.elems() - (a.pool.write_full ? 0.0 : (double)y.elems());
        }
        // ---- the opt-in INTEGER path (tamd_options.u8_integer; u8i_kernels.hip): exact int32 sums on the int8 MFMA, one rounding,
        // then the reference's own requantisation -- within one quantisation step of the reference's bytes, not identical.
        // Everything of this node that is not the convolution proper (fused ReLU / max-pool tails, concat-by-offset placement)
        // is shared with the byte-exact kernels.  Layers the kernel does not take (maps narrower than 4 columns, patches beyond
        // 512 pixels) fall through to the byte-exact family, whose bytes are inside the bar by definition.
        // (first layers -- 3 or 4 input channels against the kernel's 32-channel K step -- stay on conv_u8_rgb3x3 / the staging GEMM)
        const char* imc = exp_env("TAMD_U8_INT_MIN_C");
        if (g->opt.u8_integer && x.c <= 4 && !(exp_env("TAMD_U8I_RGB") && atoi(exp_env("TAMD_U8I_RGB")) == 0)) {
            a.i_alpha = qx.zp - 128; a.i_beta = qw.zp - 128;
            if (conv_u8i_rgb_applicable(a, p.kernel_h, p.kernel_w, p.dilation_h, p.dilation_w)) {
                conv_u8i_rgb_prepare(a);
                std::vector<int8_t> wp(conv_u8i_rgb_packed_bytes(a));
                std::vector<int32_t> cv((size_t)rup(cout, 16) + 4);
                conv_u8i_rgb_pack(a, w.data.data(), qw.zp, qx.zp, b ? (const int32_t*)b->data.data() : nullptr, wp.data(), cv.data());
                int8_t* dw = nullptr; int32_t* dc = nullptr;
                if (upload(g, wp, &dw) || upload(g, cv, &dc)) return -1;
                a.iw = dw; a.icv = dc;
                st.rd.push_back(access_of(x));
                st.wr.push_back(access_of(y));
                if (pool) st.wr.push_back(access_of(g->tensors[pool->out[0]]));
                st.deps = true;
                st.kernel = std::string("conv_u8i_rgb3x3") + (relu ? "+relu" : "") + (pool ? "+maxpool" : "");
                st.fn = [a](hipStream_t s) { return launch_conv_u8i_rgb(a, s); };
                g->steps.push_back(st);
                return 0;
            }
        }
        if (g->opt.u8_integer && x.c >= (imc ? atoi(imc) : 8)) {
            a.i_alpha = qx.zp - 128; a.i_beta = qw.zp - 128;
            // candidates: the general kernel's tile shapes (ids 0..5) and, for 1x1 / stride 1 / unpadded layers, the register-only
            // pointwise kernel's (ids 6..11).  Every one computes the same bytes (exact integer sums, one epilogue)
            const int NG = conv_u8i_num_cfgs(), NP = conv_u8i_pw_num_cfgs();
            auto iprepare = [&](U8ConvArgs& ac, int c) -> bool {
                return c < NG ? conv_u8i_prepare(ac, c, p.kernel_h, p.kernel_w, p.dilation_h, p.dilation_w) : conv_u8i_pw_prepare(ac, c - NG, p.kernel_h, p.kernel_w);
            };
            const char* pwe = tamd_pin("u8i_pw");                   // 0: never the pointwise kernel (tests), 1: only it where it applies
            std::vector<int> cands;
            for (int c = 0; c < NG + NP; c++) {
                if (c >= NG && pwe && atoi(pwe) == 0) continue;
                U8ConvArgs ac = a;
                if (iprepare(ac, c)) cands.push_back(c);
            }
            if (pwe && atoi(pwe) == 1) {
                std::vector<int> only;
                for (int c : cands) if (c >= NG) only.push_back(c);
                if (!only.empty()) cands.swap(only);
            }
            const char* ic = tamd_pin("u8i_cfg");                 // tests / fuzzing: pin one tile shape where it applies
            if (ic && *ic) {
                const int want = atoi(ic) % (NG + NP);
                if (std::find(cands.begin(), cands.end(), want) != cands.end()) cands.assign(1, want);
            }
            if (!cands.empty()) {
                std::map<int, std::pair<int8_t*, int32_t*>> ipacked;      // cout tile height -> packed weights + per-channel constants
                auto bm_of = [&](int c) { return c < NG ? conv_u8i_bm(c) : conv_u8i_pw_bm(c - NG); };
                auto iready = [&](U8ConvArgs& ac, int c) -> int {
                    if (!iprepare(ac, c)) return -1;
                    const int bm = bm_of(c);
                    auto it = ipacked.find(bm);
                    if (it == ipacked.end()) {
                        std::vector<int8_t> wp(conv_u8i_packed_bytes(ac, bm));
                        std::vector<int32_t> cv((size_t)rup(cout, bm) + 4);
                        conv_u8i_pack(ac, bm, w.data.data(), qw.zp, qx.zp, b ? (const int32_t*)b->data.data() : nullptr, wp.data(), cv.data());
                        int8_t* dw = nullptr; int32_t* dc = nullptr;
                        if (upload(g, wp, &dw) || upload(g, cv, &dc)) return -1;
                        it = ipacked.emplace(bm, std::make_pair(dw, dc)).first;
                    }
                    ac.iw = it->second.first; ac.icv = it->second.second;
                    return 0;
                };
                auto ilaunch = [NG](const U8ConvArgs& ac, int c, hipStream_t s) { return c < NG ? launch_conv_u8i(ac, s) : launch_conv_u8i_pw(ac, s); };
                auto iname = [NG](const U8ConvArgs& ac, int c) { return c < NG ? conv_u8i_kernel_name(ac) : conv_u8i_pw_kernel_name(ac); };
                // geometry heuristic: the largest tile that still gives every CU a block; the plan-time timing then decides
                auto blocks_of = [&](int c) {
                    U8ConvArgs ac = a;
                    iprepare(ac, c);
                    static const int bns[] = {64, 64, 128, 128, 128, 256};
                    const int bm = bm_of(c), bn = c < NG ? bns[c] : conv_u8i_pw_bn(c - NG);
                    const long tiles = ac.i_tw ? (long)((y.w + ac.i_tw - 1) / ac.i_tw) * ((y.h + bn / ac.i_tw - 1) / (bn / ac.i_tw)) : (y.h * y.w + bn - 1) / bn;
                    return tiles * x.n * ((cout + bm - 1) / bm);
                };
                int pick = cands[0];
                {
                    static const int pref[] = {6, 7, 9, 8, 10, 11, 3, 1, 2, 0, 5, 4};
                    long most = -1;
                    bool done = false;
                    for (int c : pref) {
                        if (std::find(cands.begin(), cands.end(), c) == cands.end()) continue;
                        const long bl = blocks_of(c);
                        if (!done && bl >= 256) { pick = c; done = true; }
                        if (!done && bl > most) { most = bl; pick = c; }
                    }
                }
                const char* iat_env = getenv("TAMD_AUTOTUNE");
                char ikey[256];
                snprintf(ikey, sizeof(ikey), "u8iconv|%s|%dx%dx%dx%d>%d k%dx%d s%d d%d%s%s", n.name.c_str(), x.n, x.c, x.h, x.w, cout, p.kernel_h, p.kernel_w,
                         p.stride_h, p.dilation_h, relu ? "+relu" : "", pool ? "+pool" : "");
                std::string icached;
                const bool itune = !(iat_env && atoi(iat_env) == 0) && st.macs >= 4e6 && cands.size() > 1;
                bool ifrom_cache = false;
                if (itune && plan_cache_get(ikey, &icached) && icached.size() >= 2 && icached[0] == 'i') {
                    const int c = atoi(icached.c_str() + 1);
                    if (std::find(cands.begin(), cands.end(), c) != cands.end()) { pick = c; ifrom_cache = true; }
                }
                if (itune && !ifrom_cache) {
                    hipEvent_t e0, e1;
                    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
                    void* flush = autotune_cold(g) ? l2_flush_buffer() : nullptr;
                    float best_ms = 1e30f;
                    std::vector<int> order{pick};
                    for (int c : cands) if (c != pick) order.push_back(c);
                    for (int c : order) {
                        U8ConvArgs ac = a;
                        if (iready(ac, c)) return -1;
                        auto launch = [&]() { return ilaunch(ac, c, g->stream); };
                        float ms = 1e30f;
                        if (launch() != hipSuccess) { (void)hipGetLastError(); continue; }
                        if (flush) { if (time_cold(g, flush, launch, &ms)) return -1; }
                        else {
                            const int reps = 10;
                            HIPCHK(hipEventRecord(e0, g->stream));
                            for (int it = 0; it < reps; it++) (void)launch();
                            HIPCHK(hipEventRecord(e1, g->stream));
                            HIPCHK(hipEventSynchronize(e1));
                            HIPCHK(hipEventElapsedTime(&ms, e0, e1));
                            ms /= reps;
                        }
                        if (getenv("TAMD_DEBUG")) fprintf(stderr, "[tamd] %s: %s %.2f us\n", n.name.c_str(), iname(ac, c), 1e3 * ms);
                        if (best_ms > 1e29f || ms < best_ms * 0.96f) { best_ms = ms; pick = c; }
                    }
                    hipEventDestroy(e0); hipEventDestroy(e1);
                    plan_cache_put(ikey, "i" + std::to_string(pick));
                }
                if (iready(a, pick)) return -1;
                st.rd.push_back(access_of(x));
                st.wr.push_back(access_of(y));
                if (pool) st.wr.push_back(access_of(g->tensors[pool->out[0]]));
                st.deps = true;
                st.kernel = std::string(iname(a, pick)) + (relu ? "+relu" : "") + (pool ? "+maxpool" : "");
                const int picked = pick;
                st.fn = [a, picked, ilaunch](hipStream_t s) { return ilaunch(a, picked, s); };
                g->steps.push_back(st);
                return 0;
            }
        }
        // plan-time autotune over the tile configurations (every one produces the same bytes: the chain order of an
        // output does not depend on the tiling); TAMD_AUTOTUNE=0 keeps the heuristic choice
        // the patch kernel (3x3 / 1x1 with whole super-steps of channels): one more candidate of the same bytes.
        // TAMD_U8_PATCH=0 never (conv_u8_patch_prepare), =1 whenever it applies (tests), otherwise it has to win the timing
        const char* pk_env = tamd_pin("u8_patch");
        const bool pk_force = pk_env && atoi(pk_env) == 1;
        int pk_best = -1;
        float* pk_w = nullptr;                     // dequantised weights in fragment order: one copy serves every tile configuration
        auto patch_for = [&](U8ConvArgs& ac, int c) -> int {          // 1: ready, 0: not applicable, -1: error
            if (!conv_u8_patch_prepare(ac, c, p.kernel_h, p.kernel_w, p.dilation_h, p.dilation_w)) return 0;
            if (!pk_w) {
                std::vector<float> wp(conv_u8_patch_packed_bytes(ac) / 4);
                conv_u8_patch_pack(ac, w.data.data(), (uint8_t)qw.zp, qw.scale, wp.data());
                if (upload(g, wp, &pk_w)) return -1;
            }
            ac.wpk = reinterpret_cast<const uint8_t*>(pk_w);
            return 1;
        };
        // conv_u8_pw shares the 1x1 fragment-order weights and the tail blocks with the patch kernel, not its switches
        auto pw_ready = [&](U8ConvArgs& ac) -> int {
            ac.pk_kh = ac.pk_kw = 1; ac.pk_dh = ac.pk_dw = 1; ac.pk_wp = 0; ac.pk_cfg = 0; ac.pk_npad = 64;
            if (!pk_w) {
                std::vector<float> wp(conv_u8_patch_packed_bytes(ac) / 4);
                conv_u8_patch_pack(ac, w.data.data(), (uint8_t)qw.zp, qw.scale, wp.data());
                if (upload(g, wp, &pk_w)) return -1;
            }
            ac.wpk = reinterpret_cast<const uint8_t*>(pk_w);
            return 0;
        };
        // conv_u8_c3 (shallow 3x3 layers of large maps) shares the 3x3 fragment-order weights and the tail blocks with the patch kernel
        auto c3_ready = [&](U8ConvArgs& ac) -> int {
            ac.pk_kh = ac.pk_kw = 3; ac.pk_dh = ac.pk_dw = 1; ac.pk_wp = 0; ac.pk_cfg = 0; ac.pk_npad = 256; ac.pk_tw = 0;
            if (!pk_w) {
                std::vector<float> wp(conv_u8_patch_packed_bytes(ac) / 4);
                conv_u8_patch_pack(ac, w.data.data(), (uint8_t)qw.zp, qw.scale, wp.data());
                if (upload(g, wp, &pk_w)) return -1;
            }
            ac.wpk = reinterpret_cast<const uint8_t*>(pk_w);
            return 0;
        };
        const char* c3_env = tamd_pin("u8_c3");                     // 0: never, 1: wherever it applies (tests)
        const bool c3_ok = conv_u8_c3_applicable(a, p.kernel_h, p.kernel_w, p.dilation_h, p.dilation_w);
        bool use_c3 = false;
        // first layers (3x3 on <= 4 channels): the per-pixel VALU kernel competes with the MFMA family (same bytes)
        const char* rgb_env = tamd_pin("u8_rgb3x3");                  // 0: never, 1: always (tests)
        const bool rgb_ok = conv_u8_rgb3x3_applicable(x.c, p.kernel_h, p.kernel_w, p.dilation_h, p.dilation_w, p.group) && cout <= 128
                            && !(rgb_env && atoi(rgb_env) == 0);
        U8ConvArgs rgb = a;
        auto rgb_ready = [&]() -> int {
            if (rgb.wf) return 0;
            rgb.wf_ld = rup(K, 4);
            std::vector<float> wf((size_t)cout * rgb.wf_ld, 0.f);
            for (int co = 0; co < cout; co++)
                for (int k = 0; k < K; k++) wf[(size_t)co * rgb.wf_ld + k] = ((float)w.data[(size_t)co * K + k] - (float)qw.zp) * qw.scale;
            float* dwf = nullptr;
            if (upload(g, wf, &dwf)) return -1;
            rgb.wf = dwf;
            return 0;
        };
        bool use_rgb = false, use_pw = false;           // conv_u8_rgb3x3 / conv_u8_pw (shallow pointwise layers of large maps)
        const char* pw_env = tamd_pin("u8_pw");                     // 0: never, 1: wherever it applies (tests)
        const char* at_env = getenv("TAMD_AUTOTUNE");
        const bool tune = !(at_env && atoi(at_env) == 0) && st.macs >= 4e6 && !tamd_pin("u8_cfg");
        // what the autotune decided last time (TAMD_PLAN_CACHE): "g<cfg>" GEMM family, "p<cfg>" patch kernel, "rgb"
        char ckey[256];
        snprintf(ckey, sizeof(ckey), "u8conv|%s|%dx%dx%dx%d>%d k%dx%d s%d d%d%s%s", n.name.c_str(), x.n, x.c, x.h, x.w, cout, p.kernel_h, p.kernel_w,
                 p.stride_h, p.dilation_h, relu ? "+relu" : "", pool ? "+pool" : "");
        std::string cached;
        bool from_cache = false;
        if (tune && !pk_force && !rgb_env && !pk_env && !pw_env && !c3_env && plan_cache_get(ckey, &cached) && cached.size() >= 2) {
            const int c = atoi(cached.c_str() + 1);
            if (cached == "rgb" && rgb_ok) { use_rgb = true; from_cache = true; }
            else if (cached == "pw" && conv_u8_pw_applicable(a, p.kernel_h, p.kernel_w)) { use_pw = true; from_cache = true; }
            else if (cached == "c3" && c3_ok) { use_c3 = true; from_cache = true; }
            else if (cached[0] == 'g' && c >= 0 && c < conv_u8_gemm_num_cfgs()) {
                U8ConvArgs ac = a; ac.cfg = c; ac.Kpad = rup(K, conv_u8_gemm_kc(c));
                if (conv_u8_gemm_lds(ac) <= 150 * 1024) { a.cfg = c; from_cache = true; }      // the same filter the autotune applies
            }
            else if (cached[0] == 'p' && c >= 0 && c < conv_u8_patch_num_cfgs()) { U8ConvArgs ac = a; if (patch_for(ac, c) == 1) { pk_best = c; from_cache = true; } }
        }
        if (tune && !from_cache) {
            hipEvent_t e0, e1;
            HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
            float best_ms = 1e30f;
            int best_cfg = a.cfg;
            // per-launch time of a candidate AS IT RUNS INSIDE A PASS (time_cold: weights evicted, input left by its producer);
            // back-to-back launches of one layer flatter the latency-bound members by 30-60 % (profiles/r03_insitu_*).
            // Small graphs / TAMD_AUTOTUNE_COLD=0: warm timing (a warm-up launch, then 5 back-to-back launches, 25 when that is
            // under 20 us)
            void* flush = autotune_cold(g) ? l2_flush_buffer() : nullptr;
            auto time_of = [&](const std::function<hipError_t()>& launch, float* out) -> int {
                if (launch() != hipSuccess) { (void)hipGetLastError(); *out = 1e30f; return 0; }
                if (flush) return time_cold(g, flush, launch, out);
                int reps = 5;
                for (int round = 0; round < 2; round++) {
                    HIPCHK(hipEventRecord(e0, g->stream));
                    for (int it = 0; it < reps; it++) (void)launch();
                    HIPCHK(hipEventRecord(e1, g->stream));
                    HIPCHK(hipEventSynchronize(e1));
                    HIPCHK(hipEventElapsedTime(out, e0, e1));
                    *out /= reps;
                    if (*out > 0.02f) break;
                    reps = 25;
                }
                return 0;
            };
            // the heuristic choice is timed first; another shape has to beat it by more than the timing noise
            std::vector<int> order{a.cfg};
            for (int c = 0; c < conv_u8_gemm_num_cfgs(); c++)
                if (c != a.cfg) order.push_back(c);
            for (int c : order) {
                U8ConvArgs ac = a; ac.cfg = c; ac.Kpad = rup(K, conv_u8_gemm_kc(c));
                if (conv_u8_gemm_lds(ac) > 150 * 1024) continue;
                if ((ac.wq = pack_for(c)) == nullptr) return -1;
                float ms = 0;
                if (time_of([&]() { return launch_conv_u8_gemm(ac, g->stream); }, &ms)) return -1;
                if (best_ms > 1e29f || ms < best_ms * 0.96f) { best_ms = ms; best_cfg = c; }
            }
            float pk_ms = 1e30f;
            const char* pcfg = tamd_pin("u8_patch_cfg");
            int named = -1;
            if (pk_force && pcfg) {
                U8ConvArgs ac = a;
                const int c = atoi(pcfg) % conv_u8_patch_num_cfgs();
                if (conv_u8_patch_prepare(ac, c, p.kernel_h, p.kernel_w, p.dilation_h, p.dilation_w)) named = c;
            }
            for (int c = 0; c < conv_u8_patch_num_cfgs(); c++) {
                // TAMD_U8_PATCH=1 pins the MFMA patch kernel (tests): the lanes configuration only competes there when it is named
                // TAMD_U8_PATCH=1 + TAMD_U8_PATCH_CFG=<c>: that configuration alone competes where it applies (tests pin forms with it);
                // without a name the lanes configuration stays out of the forced race (it pins the MFMA patch kernel)
                if (pk_force && named >= 0 && c != named) continue;
                if (pk_force && named < 0 && c == conv_u8_patch_lanes_cfg()) continue;
                U8ConvArgs ac = a;
                const int r = patch_for(ac, c);
                if (r < 0) return -1;
                if (!r) continue;
                float ms = 0;
                if (time_of([&]() { return launch_conv_u8_patch(ac, g->stream); }, &ms)) return -1;
                if (getenv("TAMD_DEBUG")) fprintf(stderr, "[tamd] %s: %s %.2f us (gemm family best %.2f us)\n", n.name.c_str(), conv_u8_patch_kernel_name(ac), 1e3 * ms, 1e3 * best_ms);
                if (pk_force ? ms < pk_ms : ms < best_ms * 0.96f) { pk_best = c; pk_ms = ms; if (!pk_force) best_ms = ms; }
            }
            if (conv_u8_pw_applicable(a, p.kernel_h, p.kernel_w)) {
                U8ConvArgs ac = a;
                if (pw_ready(ac)) return -1;
                {
                    float ms = 1e30f;
                    if (time_of([&]() { return launch_conv_u8_pw(ac, g->stream); }, &ms)) return -1;
                    if (getenv("TAMD_DEBUG")) fprintf(stderr, "[tamd] %s: %s %.2f us (best so far %.2f us)\n", n.name.c_str(), conv_u8_pw_kernel_name(ac), 1e3 * ms, 1e3 * best_ms);
                    if (ms < best_ms * 0.96f) { use_pw = true; best_ms = ms; }
                }
            }
            if (c3_ok) {
                U8ConvArgs ac = a;
                if (c3_ready(ac)) return -1;
                float ms = 1e30f;
                if (time_of([&]() { return launch_conv_u8_c3(ac, g->stream); }, &ms)) return -1;
                if (getenv("TAMD_DEBUG")) fprintf(stderr, "[tamd] %s: %s %.2f us (best so far %.2f us)\n", n.name.c_str(), conv_u8_c3_kernel_name(ac), 1e3 * ms, 1e3 * best_ms);
                if (ms < best_ms * 0.96f) { use_c3 = true; use_pw = false; pk_best = -1; best_ms = ms; }
            }
            if (rgb_ok) {
                if (rgb_ready()) return -1;
                float ms = 1e30f;
                if (time_of([&]() { return launch_conv_u8_rgb3x3(rgb, g->stream); }, &ms)) return -1;
                use_rgb = ms < best_ms * 0.96f;
            }
            hipEventDestroy(e0); hipEventDestroy(e1);
            a.cfg = best_cfg;
            plan_cache_put(ckey, use_rgb ? std::string("rgb") : use_c3 ? std::string("c3") : use_pw ? std::string("pw") : pk_best >= 0 ? "p" + std::to_string(pk_best) : "g" + std::to_string(best_cfg));
        }
        if (rgb_ok && rgb_env && atoi(rgb_env) == 1) use_rgb = true;
        if (pw_env && atoi(pw_env) == 1 && conv_u8_pw_applicable(a, p.kernel_h, p.kernel_w)) use_pw = true;
        if (use_pw && !use_rgb) {
            if (pw_ready(a)) return -1;
            st.kernel = std::string(conv_u8_pw_kernel_name(a)) + (relu ? "+relu" : "");
            st.fn = [a](hipStream_t s) { return launch_conv_u8_pw(a, s); };
            g->steps.push_back(st);
            return 0;
        }
        // everything this launch touches besides constants: the input, the output (a concat slice when it is a view), the pooled output
        st.rd.push_back(access_of(x));
        st.wr.push_back(access_of(y));
        if (pool) st.wr.push_back(access_of(g->tensors[pool->out[0]]));
        st.deps = true;
        if (c3_ok && c3_env && atoi(c3_env) == 1) use_c3 = true;
        if (use_c3 && !use_rgb) {
            if (c3_ready(a)) return -1;
            st.kernel = std::string(conv_u8_c3_kernel_name(a)) + (relu ? "+relu" : "") + (pool ? "+maxpool" : "");
            st.fn = [a](hipStream_t s) { return launch_conv_u8_c3(a, s); };
            g->steps.push_back(st);
            return 0;
        }
        if (use_rgb) {
            if (rgb_ready()) return -1;
            st.kernel = std::string(conv_u8_rgb3x3_kernel_name(rgb)) + (relu ? "+relu" : "") + (pool ? "+maxpool" : "");
            st.fn = [rgb](hipStream_t s) { return launch_conv_u8_rgb3x3(rgb, s); };
            g->steps.push_back(st);
            return 0;
        }
        a.Kpad = rup(K, conv_u8_gemm_kc(a.cfg));      // stages of the chosen depth only (the tap table stays padded to 64)
        if ((a.wq = pack_for(a.cfg)) == nullptr) return -1;
        if (!tune && !pk_force && !pk_env && pk_best < 0 && !tamd_pin("u8_cfg")) {
            // layers too small to be worth timing (< 4 MMAC): lane-level chains wherever they apply -- a GEMM launch there is 8-19 us
            // of set-up around a handful of live MFMA columns (profiles/r04_layers_mssd_uint8_b16_lanes.txt)
            U8ConvArgs ac = a;
            const int r = patch_for(ac, conv_u8_patch_lanes_cfg());
            if (r < 0) return -1;
            if (r) pk_best = conv_u8_patch_lanes_cfg();
        }
        if (pk_force && pk_best < 0) {
            const char* pc = tamd_pin("u8_patch_cfg");          // tests / fuzzing: the tile configuration to try first
            const int first = pc ? atoi(pc) % conv_u8_patch_num_cfgs() : 0;
            for (int k = 0; k < conv_u8_patch_num_cfgs() && pk_best < 0; k++) {
                const int c = (first + k) % conv_u8_patch_num_cfgs();
                U8ConvArgs ac = a;
                const int r = patch_for(ac, c);
                if (r < 0) return -1;
                if (r) pk_best = c;
            }
        }
        if (pk_best >= 0) {
            if (patch_for(a, pk_best) != 1) return -1;
            st.kernel = std::string(conv_u8_patch_kernel_name(a)) + (relu ? "+relu" : "") + (pool ? "+maxpool" : "");
            st.fn = [a](hipStream_t s) { return launch_conv_u8_patch(a, s); };
        } else {
            st.kernel = std::string(conv_u8_gemm_kernel_name(a)) + (relu ? "+relu" : "") + (pool ? "+maxpool" : "");
            st.fn = [a](hipStream_t s) { return launch_conv_u8_gemm(a, s); };
        }
    } else {
        std::vector<float> wf((size_t)cout * K);
        for (size_t i = 0; i < wf.size(); i++) wf[i] = ((float)w.data[i] - (float)qw.zp) * qw.scale;
        float* dwf = nullptr;
        if (upload(g, wf, &dwf)) return -1;
        U8DirectArgs a{};
        a.x = (const uint8_t*)x.dptr; a.wf = dwf; a.bias = dbias; a.y = (uint8_t*)y.dptr;
        a.N = x.n; a.C = x.c; a.H = x.h; a.W = x.w; a.OH = y.h; a.OW = y.w; a.cout = cout;
        a.KH = p.kernel_h; a.KW = p.kernel_w; a.SH = p.stride_h; a.SW = p.stride_w; a.PH = p.pad_h0; a.PW = p.pad_w0;
        a.DH = p.dilation_h; a.DW = p.dilation_w; a.group = p.group;
        a.out_img = (y.is_view ? y.cs : cout) * y.h * y.w; a.out_c0 = y.c_off;
        a.in_scale = qx.scale; a.in_zp = (float)qx.zp; a.w_scale = qw.scale;
        a.act = p.activation; a.out_scale = qy.scale; a.out_zp = qy.zp; a.relu = fr;
        st.kernel = relu ? "conv_u8_direct+relu" : "conv_u8_direct";
        st.fn = [a](hipStream_t s) { return launch_conv_u8_direct(a, s); };
    }
    g->steps.push_back(st);
    return 0;
}

static int plan_fc_u8(tamd_graph* g, HNode& n)
{
    HTensor& x = g->tensors[n.in[0]];
    HTensor& w = g->tensors[n.in[1]];
    HTensor* b = n.in.size() > 2 ? &g->tensors[n.in[2]] : nullptr;
    HTensor& y = g->tensors[n.out[0]];
    U8Q qx, qw, qy;
    if (q_of(x, &qx, "tensor") || q_of(w, &qw, "weight") || q_of(y, &qy, "tensor")) return -1;
    const int batch = x.dims[0], hidden = (int)(x.elems() / batch), nout = y.c, nout_pad = rup(nout, 64);
    if (w.dtype != TAMD_DT_UINT8 || (size_t)hidden * nout != w.data.size()) { set_error("fc %s: weight mismatch", n.name.c_str()); return -1; }
    if (hidden * 4 > 60000) { set_error("fc %s: hidden %d too large for the LDS row", n.name.c_str(), hidden); return -1; }
    std::vector<float> wf((size_t)hidden * nout_pad, 0.f);
    for (int o = 0; o < nout; o++)
        for (int j = 0; j < hidden; j++)
            wf[(size_t)j * nout_pad + o] = ((float)w.data[(size_t)o * hidden + j] - (float)qw.zp) * qw.scale;
    float* dwf = nullptr;
    if (upload(g, wf, &dwf)) return -1;
    U8FcArgs a{};
    a.x = (const uint8_t*)x.dptr; a.wf = dwf; a.y = (uint8_t*)y.dptr;
    if (b) {
        std::vector<int32_t> hb((const int32_t*)b->data.data(), (const int32_t*)b->data.data() + nout);
        int32_t* d = nullptr;
        if (upload(g, hb, &d)) return -1;
        a.bias = d;
        a.bias_scale = b->scales.empty() ? 0.f : b->scales[0];      // fc_ref.c:146 bias_tensor->scale
    }
    a.batch = batch; a.hidden = hidden; a.nout = nout; a.nout_pad = nout_pad;
    a.in_scale = qx.scale; a.in_zp = (float)qx.zp; a.out_scale = qy.scale; a.out_zp = qy.zp;
    Step st; st.node = n.name; st.kernel = "fc_u8";
    st.macs = (double)batch * hidden * nout; st.bytes = 4.0 * hidden * nout + batch * (hidden + nout);
    st.fn = [a](hipStream_t s) { return launch_fc_u8(a, s); };
    g->steps.push_back(st);
    return 0;
}

int plan_u8(tamd_graph* g)
{
    for (auto& t : g->tensors) {
        if (t.ttype == TAMD_TT_CONST) continue;
        nhwc_geom(t);
        t.nchw_raw = true;      // dense NCHW bytes: read_tensor / IO copy them as they are
        t.cs = 0; t.c_off = 0;
    }
    std::vector<int> alias_of(g->tensors.size(), -1);
    for (auto& n : g->nodes)
        if (n.op == TAMD_OP_DROPOUT || n.op == TAMD_OP_FLATTEN || n.op == TAMD_OP_RESHAPE) alias_of[n.out[0]] = n.in[0];   // dense NCHW: views
    for (auto& io : g->inputs) {
        HTensor& t = g->tensors[io.tensor];
        io.bytes = t.elems();
        if (dev_alloc(g, &io.stage, io.bytes, true)) return -1;
        HIPCHK(hipHostMalloc(&io.pinned, io.bytes, hipHostMallocDefault));
        t.dptr = io.stage;
    }
    // concat-by-offset: when an input carries the concat output's (scale, zero point) the reference's
    // per-element rescale roundf((u - zp) * 1 + zp) is the identity (concat_kernel_ref_uint8.c:309-352), so a conv /
    // relu / upsample whose only consumer is that concat writes its channels straight into the concat output
    std::vector<int> view_of(g->tensors.size(), -1), view_off(g->tensors.size(), 0);
    for (auto& n : g->nodes) {
        if (n.op != TAMD_OP_CONCAT) continue;
        HTensor& y = g->tensors[n.out[0]];
        const int ax = n.p.concat.axis < 0 ? n.p.concat.axis + (int)y.dims.size() : n.p.concat.axis;
        int off = 0;
        for (int i : n.in) {
            HTensor& xi = g->tensors[i];
            bool ok = ax == 1 && xi.ttype == TAMD_TT_VAR && count_consumers(g, i) == 1 && alias_of[i] < 0 && view_of[i] < 0
                      && !xi.scales.empty() && !y.scales.empty() && xi.scales[0] == y.scales[0]
                      && (xi.zps.empty() ? 0 : xi.zps[0]) == (y.zps.empty() ? 0 : y.zps[0]);
            if (ok) {
                ok = false;
                for (auto& pn : g->nodes)
                    if (!pn.out.empty() && pn.out[0] == i) ok = pn.op == TAMD_OP_CONV || pn.op == TAMD_OP_RELU || pn.op == TAMD_OP_UPSAMPLE;
            }
            if (ok) { view_of[i] = n.out[0]; view_off[i] = off; }
            off += xi.c;
        }
    }
    for (size_t i = 0; i < g->tensors.size(); i++) {
        HTensor& t = g->tensors[i];
        if (t.ttype == TAMD_TT_CONST || t.dptr || alias_of[i] >= 0 || view_of[i] >= 0) continue;
        if (dev_alloc(g, &t.dptr, t.elems(), true)) return -1;
    }
    for (size_t i = 0; i < g->tensors.size(); i++)
        if (view_of[i] >= 0) {
            HTensor& t = g->tensors[i];
            HTensor& o = g->tensors[view_of[i]];
            if (!o.dptr) { set_error("concat of concat views is not supported"); return -1; }
            t.dptr = o.dptr; t.is_view = true; t.c_off = view_off[i]; t.cs = o.c;      // cs: channels of the enclosing buffer
        }
    for (int pass = 0; pass < 4; pass++)
        for (size_t i = 0; i < g->tensors.size(); i++)
            if (alias_of[i] >= 0) g->tensors[i].dptr = g->tensors[alias_of[i]].dptr;

    // conv -> ReLU / leaky ReLU fusion (YOLOv3-tiny: 11 of them): the ReLU node is applied to the conv's own uint8
    // result in the conv epilogue when nothing else reads that result
    // SSD heads (MobileNet-SSD: 12 of them): conv -> Permute(0,2,3,1) -> Flatten -> Concat.  The permute and the
    // flatten only re-index bytes, so the concat reads the conv result in permuted order itself: perm_src[flatten
    // output] = the permute's input, and the permute launch disappears (TAMD_FUSE_PERMUTE=0 keeps it)
    std::vector<int> perm_src(g->tensors.size(), -1);
    std::vector<char> skip_perm(g->nodes.size(), 0);
    {
        const char* pe = getenv("TAMD_FUSE_PERMUTE");
        if (!(pe && atoi(pe) == 0))
            for (size_t pi = 0; pi < g->nodes.size(); pi++) {
                const HNode& pn = g->nodes[pi];
                const int* o = pn.p.perm.order;
                if (pn.op != TAMD_OP_PERMUTE || !(o[0] == 0 && o[1] == 2 && o[2] == 3 && o[3] == 1)) continue;
                if (count_consumers(g, pn.out[0]) != 1) continue;
                const HNode* fl = nullptr;
                for (auto& m : g->nodes)
                    if (m.op == TAMD_OP_FLATTEN && m.in[0] == pn.out[0]) fl = &m;
                if (!fl || count_consumers(g, fl->out[0]) != 1) continue;
                bool to_concat = false;
                for (auto& m : g->nodes)
                    if (m.op == TAMD_OP_CONCAT)
                        for (int i : m.in) to_concat |= (i == fl->out[0]);
                if (!to_concat) continue;
                perm_src[fl->out[0]] = pn.in[0];
                skip_perm[pi] = 1;
            }
    }

    const char* fuse_env = getenv("TAMD_FUSE_RELU");          // read at every prerun (tests switch it)
    std::vector<char> fused(g->nodes.size(), 0);
    for (size_t ni = 0; ni < g->nodes.size(); ni++) {
        HNode& n = g->nodes[ni];
        if (fused[ni]) continue;
        switch (n.op) {
        case TAMD_OP_INPUT: case TAMD_OP_CONST: case TAMD_OP_DROPOUT: case TAMD_OP_FLATTEN: case TAMD_OP_RESHAPE:
            break;
        case TAMD_OP_SOFTMAX: {            // the quantised part of the SSD tail: Reshape -> Softmax(axis 2) -> Flatten on mbox_conf
            HTensor& x = g->tensors[n.in[0]];
            HTensor& y = g->tensors[n.out[0]];
            if (x.is_view || y.is_view) { set_error("softmax %s on a concat view is not supported", n.name.c_str()); return -1; }
            int ax = n.p.softmax.axis < 0 ? n.p.softmax.axis + (int)x.dims.size() : n.p.softmax.axis;
            if (ax < 0 || ax >= (int)x.dims.size()) { set_error("softmax %s: bad axis", n.name.c_str()); return -1; }
            U8SoftmaxArgs a{};
            a.x = (const uint8_t*)x.dptr; a.y = (uint8_t*)y.dptr;
            a.outer = 1; a.inner = 1; a.on = x.dims[ax];
            for (int i = 0; i < ax; i++) a.outer *= x.dims[i];
            for (size_t i = ax + 1; i < x.dims.size(); i++) a.inner *= x.dims[i];
            if (q_of(x, &a.in, "tensor") || q_of(y, &a.out, "tensor")) return -1;
            Step st; st.node = n.name; st.kernel = "softmax_u8"; st.bytes = 2.0 * x.elems();
            st.fn = [a](hipStream_t s) { return launch_softmax_u8(a, s); };
            g->steps.push_back(st);
            break;
        }
        case TAMD_OP_CONV: {
            const HNode* relu = nullptr;
            size_t rj = 0;
            if (!(fuse_env && atoi(fuse_env) == 0) && count_consumers(g, n.out[0]) == 1)
                for (size_t nj = ni + 1; nj < g->nodes.size(); nj++)
                    if (g->nodes[nj].op == TAMD_OP_RELU && g->nodes[nj].in[0] == n.out[0]) { relu = &g->nodes[nj]; rj = nj; break; }
            // conv (-> ReLU) -> 2x2 stride-2 max-pool (YOLOv3-tiny: conv0..conv3): the pool is applied to the final
            // bytes in the conv epilogue -- the window's four pixels are computed by four neighbouring lanes -- when the map is
            // even-sized and has no reference "tail" pixels (OH*OW % 8 == 0); TAMD_FUSE_POOL=0 keeps the pool_u8 launch
            const HNode* pool = nullptr;
            size_t pj = 0;
            const char* pool_env = getenv("TAMD_FUSE_POOL");         // read at every prerun (tests switch it)
            {
                const int full = relu ? relu->out[0] : n.out[0];
                const HTensor& yf = g->tensors[full];
                if (!(pool_env && atoi(pool_env) == 0) && yf.dims.size() == 4
                    && yf.h % 2 == 0 && yf.w % 2 == 0 && (yf.h * yf.w) % 8 == 0 && !yf.is_view)
                    for (size_t nj = ni + 1; nj < g->nodes.size() && !pool; nj++) {
                        const HNode& m = g->nodes[nj];
                        if (m.op != TAMD_OP_POOL || m.in[0] != full || fused[nj]) continue;
                        const PoolGeom pg = pool_geom(m.p.pool, yf.h, yf.w);
                        if (m.p.pool.pool_method == 0 && pg.kh == 2 && pg.kw == 2 && pg.sh == 2 && pg.sw == 2 && pg.ph0 == 0 && pg.pw0 == 0
                            && pg.oh == yf.h / 2 && pg.ow == yf.w / 2) { pool = &m; pj = nj; }
                    }
            }
            int rc = plan_conv_u8(g, n, relu, pool);
            if (rc == 2 && pool) { pool = nullptr; rc = plan_conv_u8(g, n, relu, nullptr); }
            if (rc == 2) { relu = nullptr; rc = plan_conv_u8(g, n, nullptr); }       // kernel without the fused tail
            if (rc) return -1;
            // tensors that now only exist inside the fused launch: tamd_graph_read_tensor must refuse them instead of returning the
            // zeros of a buffer nobody writes (layer-by-layer parity tooling would be misled)
            if (g->fused_away.size() != g->tensors.size()) g->fused_away.assign(g->tensors.size(), 0);
            if (relu) { fused[rj] = 1; g->fused_away[n.out[0]] = 1; }
            if (pool) {
                fused[pj] = 1;
                const int full = relu ? relu->out[0] : n.out[0];
                bool written = count_consumers(g, full) > 1;           // == U8PoolFuse::write_full
                for (auto& io : g->outputs) written = written || io.tensor == full;
                if (!written) g->fused_away[full] = 1;
            }
            break;
        }
        case TAMD_OP_FC:
            if (plan_fc_u8(g, n)) return -1;
            break;
        case TAMD_OP_POOL: {
            HTensor& x = g->tensors[n.in[0]];
            HTensor& y = g->tensors[n.out[0]];
            PoolGeom pg = pool_geom(n.p.pool, x.h, x.w);
            U8PoolArgs a{};
            a.x = (const uint8_t*)x.dptr; a.y = (uint8_t*)y.dptr;
            a.N = x.n; a.C = x.c; a.H = x.h; a.W = x.w; a.OH = y.h; a.OW = y.w;
            a.KH = pg.kh; a.KW = pg.kw; a.SH = pg.sh; a.SW = pg.sw; a.PH = pg.ph0; a.PW = pg.pw0;
            a.method = n.p.pool.pool_method; a.caffe_flavor = n.p.pool.caffe_flavor;
            if (q_of(x, &a.in, "tensor") || q_of(y, &a.out, "tensor")) return -1;
            Step st; st.node = n.name; st.kernel = "pool_u8";
            st.bytes = (double)x.elems() + (double)y.elems();
            st.fn = [a](hipStream_t s) { return launch_pool_u8(a, s); };
            g->steps.push_back(st);
            break;
        }
        case TAMD_OP_RELU: case TAMD_OP_UPSAMPLE: {
            HTensor& x = g->tensors[n.in[0]];
            HTensor& y = g->tensors[n.out[0]];
            U8MapArgs a{};
            a.x = (const uint8_t*)x.dptr; a.y = (uint8_t*)y.dptr;
            a.N = x.n; a.C = x.c; a.H = x.h; a.W = x.w;
            a.scale = n.op == TAMD_OP_UPSAMPLE ? (int)n.p.ups.scale : 1;
            a.out_img = (y.is_view ? y.cs : y.c) * y.h * y.w; a.out_c0 = y.c_off;
            a.slope = n.op == TAMD_OP_RELU ? n.p.relu.negative_slope : 0.f;
            if (q_of(x, &a.in, "tensor") || q_of(y, &a.out, "tensor")) return -1;
            const bool up = n.op == TAMD_OP_UPSAMPLE;
            Step st; st.node = n.name; st.kernel = up ? "upsample_u8" : "relu_u8";
            st.bytes = (double)x.elems() + (double)y.elems();
            st.fn = [a, up](hipStream_t s) { return up ? launch_upsample_u8(a, s) : launch_relu_u8(a, s); };
            g->steps.push_back(st);
            break;
        }
        case TAMD_OP_PERMUTE: {
            if (skip_perm[ni]) break;                 // folded into the concat that reads it (below)
            HTensor& x = g->tensors[n.in[0]];
            HTensor& y = g->tensors[n.out[0]];
            const int* o = n.p.perm.order;
            if (!(o[0] == 0 && o[1] == 2 && o[2] == 3 && o[3] == 1)) { set_error("permute %s: only order (0,2,3,1) is supported on the device", n.name.c_str()); return -1; }
            U8CatArgs a{};
            a.x = (const uint8_t*)x.dptr; a.y = (uint8_t*)y.dptr;
            a.N = x.dims[0]; a.in_img = (int)(x.elems() / x.dims[0]);
            a.perm_c = x.dims[1]; a.perm_p = x.dims[2] * x.dims[3];
            a.out_img = a.in_img; a.out_off = 0; a.identity = 1;
            a.in.scale = a.out.scale = 1.f;
            Step st; st.node = n.name; st.kernel = "permute_u8"; st.bytes = 2.0 * x.elems();
            st.fn = [a](hipStream_t s) { return launch_flatcat_u8(a, s); };
            g->steps.push_back(st);
            break;
        }
        case TAMD_OP_PRIORBOX: {          // shapes-only node: evaluated here, once (graph_infer.hip priorbox_eval); no launch at run
            HTensor& x = g->tensors[n.in[0]];
            HTensor& y = g->tensors[n.out[0]];
            const HTensor& img = g->tensors[n.in[1]];
            U8Q qy;
            if (q_of(y, &qy, "tensor")) return -1;
            std::vector<float> boxes;
            std::vector<uint8_t> q;
            priorbox_eval(n.p.priorbox, x.dims[2], x.dims[3], img.dims[2], img.dims[3], &boxes);
            if (boxes.size() != y.elems()) { set_error("priorbox %s: output shape mismatch", n.name.c_str()); return -1; }
            priorbox_quant_u8(boxes, qy.scale, qy.zp, &q);
            HIPCHK(hipMemcpyAsync(y.dptr, q.data(), q.size(), hipMemcpyHostToDevice, g->stream));
            HIPCHK(hipStreamSynchronize(g->stream));
            y.prerun_const = true;
            break;
        }
        case TAMD_OP_CONCAT: {
            HTensor& y = g->tensors[n.out[0]];
            int ax = n.p.concat.axis < 0 ? n.p.concat.axis + (int)y.dims.size() : n.p.concat.axis;
            if (ax < 0 || ax >= (int)y.dims.size()) { set_error("concat %s: bad axis", n.name.c_str()); return -1; }
            // dense tensors: for every index in front of the axis, each input is one contiguous slice of the output's slice
            // (axis 1 of [n][c][...]: one slice per image)
            size_t outer = 1, inner = 1;
            for (int d = 0; d < ax; d++) outer *= (size_t)y.dims[d];
            for (size_t d = ax + 1; d < y.dims.size(); d++) inner *= (size_t)y.dims[d];
            const int out_img = (int)(y.dims[ax] * inner);
            bool all_const = true;
            for (int i : n.in) all_const &= g->tensors[i].prerun_const;
            int off = 0;
            // two to eight copied inputs (the SSD heads: six per concat): one launch for all of them (TAMD_FUSE_CONCAT=0: one each)
            const char* fc_env = getenv("TAMD_FUSE_CONCAT");
            U8CatMulti multi{};
            const size_t first_step = g->steps.size();
            for (int i : n.in) {
                HTensor& x = g->tensors[i];
                const int in_img = (int)(x.dims[ax] * inner);
                if (x.is_view && x.dptr == y.dptr) { off += in_img; continue; }       // written in place by its producer
                U8CatArgs a{};
                a.x = (const uint8_t*)x.dptr; a.y = (uint8_t*)y.dptr;
                a.N = (int)outer; a.in_img = in_img; a.out_img = out_img; a.out_off = off;
                if (q_of(x, &a.in, "tensor") || q_of(y, &a.out, "tensor")) return -1;
                // roundf((u - zp) * 1 + zp) == u: equal parameters make the rescale a copy
                // (.. and a SINGLE input is copied byte for byte whatever the quantisation says: concat_kernel_ref_uint8.c:47-58 --
                //  round 6, found by tools/fuzz_heads.py on the int8 twin of this rule)
                a.identity = (a.in.scale == a.out.scale && a.in.zp == a.out.zp) || n.in.size() == 1;
                const char* kname = "concat_u8";
                if (perm_src[i] >= 0) {                                  // Permute(0,2,3,1) -> Flatten -> this concat
                    HTensor& s = g->tensors[perm_src[i]];
                    a.x = (const uint8_t*)s.dptr; a.perm_c = s.dims[1]; a.perm_p = s.dims[2] * s.dims[3];
                    kname = "permute_concat_u8";
                }
                Step st; st.node = n.name; st.kernel = kname; st.bytes = 2.0 * x.elems();
                st.once = all_const;                                     // e.g. mbox_priorbox: PriorBox outputs only
                {                                                        // reads its source, writes ITS slot of every outer slice
                    Access r = access_of(perm_src[i] >= 0 ? g->tensors[perm_src[i]] : x), wa;
                    wa.base = (const char*)y.dptr; wa.size = y.elems(); wa.period = (size_t)out_img; wa.off = (size_t)off; wa.len = (size_t)in_img;
                    st.rd.push_back(r); st.wr.push_back(wa); st.deps = !y.is_view;
                }
                st.fn = [a](hipStream_t s) { return launch_flatcat_u8(a, s); };
                g->steps.push_back(st);
                if (multi.count < 8) { multi.src[multi.count] = a; multi.rescale[multi.count] = a.in.scale / a.out.scale; }
                multi.count++;
                off += in_img;
            }
            if (multi.count >= 2 && multi.count <= 8 && !all_const && !y.is_view && !(fc_env && atoi(fc_env) == 0)) {
                Step st = g->steps[first_step];
                st.kernel = g->steps[first_step].kernel + "<x" + std::to_string(multi.count) + ">";
                st.bytes = 0; st.rd.clear(); st.wr.clear();
                for (size_t k = first_step; k < g->steps.size(); k++) {
                    st.bytes += g->steps[k].bytes;
                    st.rd.insert(st.rd.end(), g->steps[k].rd.begin(), g->steps[k].rd.end());
                    st.wr.insert(st.wr.end(), g->steps[k].wr.begin(), g->steps[k].wr.end());
                    if (g->steps[k].kernel != g->steps[first_step].kernel) st.kernel = "concat_u8<x" + std::to_string(multi.count) + ">";
                }
                st.fn = [multi](hipStream_t s) { return launch_flatcat_multi_u8(multi, s); };
                g->steps.resize(first_step);
                g->steps.push_back(st);
            }
            y.prerun_const = all_const;
            break;
        }
        case TAMD_OP_ELTWISE: {
            HTensor& xa = g->tensors[n.in[0]];
            HTensor& xb = g->tensors[n.in[1]];
            HTensor& y = g->tensors[n.out[0]];
            if (xa.dims != xb.dims) { set_error("eltwise %s: broadcast not supported", n.name.c_str()); return -1; }
            U8EltArgs a{};
            a.a = (const uint8_t*)xa.dptr; a.b = (const uint8_t*)xb.dptr; a.y = (uint8_t*)y.dptr;
            a.count = xa.elems(); a.type = n.p.elt.type;
            if (a.type != 0 && a.type != 2 && a.type != 4 && a.type != 6) { set_error("eltwise %s: type %d unsupported", n.name.c_str(), a.type); return -1; }
            if (q_of(xa, &a.qa, "tensor") || q_of(xb, &a.qb, "tensor") || q_of(y, &a.out, "tensor")) return -1;
            Step st; st.node = n.name; st.kernel = "eltwise_u8"; st.bytes = 3.0 * xa.elems();
            st.fn = [a](hipStream_t s) { return launch_eltwise_u8(a, s); };
            g->steps.push_back(st);
            break;
        }
        default:
            set_error("op %d (%s) is not supported on the device for uint8", n.op, n.name.c_str());
            return -1;
        }
    }
    for (auto& io : g->outputs) {
        HTensor& t = g->tensors[io.tensor];
        io.bytes = t.elems();
        HIPCHK(hipHostMalloc(&io.pinned, io.bytes, hipHostMallocDefault));
        io.stage = t.dptr;
    }
    return 0;
}

}  // namespace tamd
