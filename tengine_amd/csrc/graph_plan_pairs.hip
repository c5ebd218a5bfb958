// int8 planner, the pair fusions: pointwise conv + its single consumer (depthwise 3x3 | global pooling) in one launch (pwdw.hip), and
// depthwise 3x3 + the pointwise conv behind it (dwpw.hip).  Split out of graph.hip in round 6.
#include "graph.h"
#include "graph_internal.h"
#include "env.h"

#include <stdarg.h>
#include <stddef.h>
#include <stdio.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <mutex>

#include "epilogue.h"
#include "graph_plan.h"

namespace tamd {

// ---- pointwise conv + its single consumer (depthwise 3x3 | global pooling) in one launch: pwdw.hip ---------------------
// Which node, if any, can ride in pointwise conv `ni`'s launch.  *tmode: 1 depthwise 3x3, 0 global pooling.
int find_pwdw_tail(tamd_graph* g, size_t ni, int* tmode, int* prod)
{
    const HNode& n = g->nodes[ni];
    if (n.op != TAMD_OP_CONV || n.in.size() < 2) return -1;
    const tamd_conv_param& p = n.p.conv;
    const HTensor& x = g->tensors[n.in[0]];
    const HTensor& y = g->tensors[n.out[0]];
    if (p.group != 1 || y.is_view || x.dtype != TAMD_DT_INT8 || count_consumers(g, n.out[0]) != 1) return -1;
    if (x.nchw_raw) {
        // the network's first conv, gathered from the NCHW graph input: patch rows of 4 consecutive bytes (KW <= 4, no
        // x dilation), at most 16 rows (c, ky) = one 64-deep K step; row offsets of 24 bits, ky*DH of 4
        if (x.c > 4 || p.kernel_w > 4 || p.dilation_w != 1 || x.c * p.kernel_h > 16 || p.dilation_h * (p.kernel_h - 1) > 15
            || (long)x.c * x.h * x.w >= (1L << 24) || p.pad_h0 < 0 || p.pad_w0 < 0)
            return -1;
        *prod = 1;
    } else {
        if (p.kernel_h != 1 || p.kernel_w != 1 || p.stride_h != 1 || p.stride_w != 1 || p.pad_h0 || p.pad_h1 || p.pad_w0 || p.pad_w1) return -1;
        *prod = 0;
    }
    for (auto& o : g->outputs) if (o.tensor == n.out[0]) return -1;
    for (size_t nj = ni + 1; nj < g->nodes.size(); nj++) {
        const HNode& c = g->nodes[nj];
        if (c.in.empty() || c.in[0] != n.out[0]) continue;
        const HTensor& o = g->tensors[c.out[0]];
        if (c.op == TAMD_OP_CONV && c.in.size() >= 2) {
            const tamd_conv_param& q = c.p.conv;
            const bool dw3 = q.group > 1 && q.group == y.c && o.c == y.c && q.kernel_h == 3 && q.kernel_w == 3 && q.dilation_h == 1
                             && q.dilation_w == 1 && q.stride_h == q.stride_w && (q.stride_h == 1 || q.stride_h == 2) && q.pad_h0 >= 0
                             && q.pad_w0 >= 0 && q.pad_h0 <= 2 && q.pad_w0 <= 2;
            if (!dw3 || o.scales.empty() || g->tensors[c.in[1]].scales.empty()) return -1;
            *tmode = 1;
            return (int)nj;
        }
        if (c.op == TAMD_OP_POOL && *prod == 0) {
            const PoolGeom pg = pool_geom(c.p.pool, y.h, y.w);
            const int m = c.p.pool.pool_method;
            if (pg.oh != 1 || pg.ow != 1 || pg.kh != y.h || pg.kw != y.w || pg.ph0 || pg.pw0 || (m != 0 && m != 1) || y.h * y.w > 1024 || o.scales.empty())
                return -1;
            *tmode = 0;
            return (int)nj;
        }
        return -1;
    }
    return -1;
}

// The two nodes were just planned as steps [s0, s0 + 2); build the fused launch, and keep whichever is faster
// (plan-time measurement; without autotune: fuse the small-map cases where launches, not bytes, are the cost).
// TAMD_FUSE_PWDW=0 never fuses, =2 always fuses; TAMD_PWDW_CFG="TH,TW,threads" pins the tile configuration (tests).
int plan_pwdw(tamd_graph* g, HNode& pw, HNode& tl, int tmode, int prod, size_t s0)
{
    const char* fenv = getenv("TAMD_FUSE_PWDW");                 // read at every prerun
    const int fmode = fenv ? atoi(fenv) : 1;
    if (fmode == 0) return 0;
    HTensor& x = g->tensors[pw.in[0]];
    HTensor& w = g->tensors[pw.in[1]];
    HTensor* b = pw.in.size() > 2 ? &g->tensors[pw.in[2]] : nullptr;
    HTensor& mid = g->tensors[pw.out[0]];
    HTensor& y = g->tensors[tl.out[0]];
    const tamd_conv_param& pp = pw.p.conv;
    const int cin = x.c, C = mid.c, slices = (C + 15) / 16, cw = slices * 16;
    const int Kw = prod == 1 ? cin * pp.kernel_h * pp.kernel_w : cin;           // weight row length in the model
    const int K = prod == 1 ? cin * pp.kernel_h * 4 : cin;                      // reduction length as the kernel walks it
    const int ktot = prod == 1 ? K : rup(cin, 16), steps = pwdw_steps((ktot + 63) / 64), nsteps = rup((ktot + 63) / 64, steps);
    if (w.elems() != (size_t)C * Kw || (b && b->elems() < (size_t)C)) return 0;
    PwDwArgs a{};
    {
        const RqFold rq = fold_requant(RQ_CONV_HCL, pp.activation, x.scales[0], mid.scales[0], w, C);
        const int8_t* wd = (const int8_t*)w.data.data();
        std::vector<int8_t> wrows;
        if (prod == 1) {                // k = (c*KH + ky)*4 + kx: rows padded to 4 taps
            wrows.assign((size_t)C * K, 0);
            for (int c = 0; c < C; c++)
                for (int r = 0; r < cin * pp.kernel_h; r++)
                    for (int kx = 0; kx < pp.kernel_w; kx++) wrows[(size_t)c * K + r * 4 + kx] = wd[(size_t)c * Kw + r * pp.kernel_w + kx];
            wd = wrows.data();
        }
        const std::vector<int8_t> wf = pack_pw_panel(wd, C, K, nsteps);
        std::vector<int32_t> bp(cw, 0);
        for (int c = 0; c < C; c++) bp[c] = b ? ((const int32_t*)b->data.data())[c] : 0;
        int8_t* d0; int32_t* d1;
        if (upload(g, wf, &d0) || upload(g, bp, &d1) || upload_rq(g, rq, cw, &a.wscale, &a.rq)) return -1;
        a.wf = d0; a.bias = d1;
    }
    a.prod = prod;
    a.coherent = (g->opt.direct_dispatch && !exp_plain_kernels()) ? 1 : 0;
    // the larger operand is the one every XCD should fetch only its share of (pwdw.hip: block -> XCD mapping)
    a.tile_major = (double)x.h * x.w * (prod == 1 ? x.c : x.cs) * (slices >= 8 ? 8 : slices) > (double)C * ktot * 8.0 ? 1 : 0;
    if (slices > 65535) a.tile_major = 0;
    if (prod == 1) {
        std::vector<unsigned> rows(16, 0u);
        for (int r = 0; r < cin * pp.kernel_h; r++) {
            const int ky = r % pp.kernel_h, ci = r / pp.kernel_h;
            rows[r] = (unsigned)(ci * x.h * x.w + ky * pp.dilation_h * x.w) | ((unsigned)(ky * pp.dilation_h) << 28);
        }
        unsigned* dt;
        if (upload(g, rows, &dt)) return -1;
        a.taps = dt; a.in_C = cin; a.in_H = x.h; a.in_W = x.w;
        a.fSH = pp.stride_h; a.fSW = pp.stride_w; a.fPH = pp.pad_h0; a.fPW = pp.pad_w0;
    }
    a.x = (const int8_t*)x.dptr + (prod == 1 ? 0 : x.c_off);
    a.N = x.n; a.H = mid.h; a.W = mid.w; a.cs_in = x.cs; a.ktot = ktot; a.nsteps = nsteps; a.steps = steps;
    a.mode = tmode; a.cw = cw; a.slices = slices;
    a.y = (int8_t*)y.dptr; a.ldc = y.cs; a.c_off = y.c_off;
    a.c_limit = y.is_view ? C : std::min(rup(C, 16), y.cs - y.c_off);
    a.S = 1; a.OH = a.OW = 1; a.TH = a.TW = 1; a.tiles_x = a.tiles_y = 1; a.RH = mid.h; a.RW = mid.w;
    if (tmode == 1) {
        const tamd_conv_param& q = tl.p.conv;
        HTensor& dwt = g->tensors[tl.in[1]];
        HTensor* db = tl.in.size() > 2 ? &g->tensors[tl.in[2]] : nullptr;
        if (dwt.elems() != (size_t)C * 9 || (db && db->elems() < (size_t)C)) return 0;
        const RqFold rq = fold_requant(conv_mode(q, g->formula_batch ? g->formula_batch : mid.n, C, C), q.activation, mid.scales[0], y.scales[0], dwt, C);
        const int8_t* wd = (const int8_t*)dwt.data.data();
        std::vector<int8_t> wp((size_t)3 * cw * 4, 0);
        for (int c = 0; c < C; c++)
            for (int r = 0; r < 3; r++)
                for (int kx = 0; kx < 3; kx++) wp[((size_t)r * cw + c) * 4 + kx] = wd[(size_t)c * 9 + r * 3 + kx];
        std::vector<int32_t> bp(cw, 0);
        for (int c = 0; c < C; c++) bp[c] = db ? ((const int32_t*)db->data.data())[c] : 0;
        int8_t* d0; int32_t* d1;
        if (upload(g, wp, &d0) || upload(g, bp, &d1) || upload_rq(g, rq, cw, &a.dw_wscale, &a.d_rq)) return -1;
        a.dw_w = d0; a.dw_bias = d1;
        a.S = q.stride_h; a.PH = q.pad_h0; a.PW = q.pad_w0; a.OH = y.h; a.OW = y.w;
    } else {
        a.pool_method = tl.p.pool.pool_method; a.p_in_scale = mid.scales[0]; a.p_out_scale = y.scales[0];
    }

    // ---- tile configurations: (TH, TW, threads) ranked by a small cost model, the best few timed on the device ---------
    struct Cfg { int th, tw, threads; double cost; int sl; };      // sl: 16-channel slices per block (pwdw.hip)
    std::vector<Cfg> cfgs;
    auto with_tiles = [&](PwDwArgs v, int th, int tw, int sl = 1) {
        v.TH = th; v.TW = tw; v.tiles_y = (v.OH + th - 1) / th; v.tiles_x = (v.OW + tw - 1) / tw;
        v.RH = (th - 1) * v.S + 3; v.RW = (tw - 1) * v.S + 3;
        v.sl = sl; v.slices = (slices + sl - 1) / sl;
        return v;
    };
    if (tmode == 0) {
        cfgs.push_back({1, 1, 256, 0.0, 1});
        cfgs.push_back({1, 1, 512, 1.0, 1});
    } else {
        std::vector<int> ths, tws;
        for (int v : {1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 28, a.OH}) if (v <= a.OH && std::find(ths.begin(), ths.end(), v) == ths.end()) ths.push_back(v);
        for (int v : {4, 6, 7, 8, 14, 16, 28, 56, a.OW}) if (v <= a.OW && std::find(tws.begin(), tws.end(), v) == tws.end()) tws.push_back(v);
        for (int th : ths)
            for (int tw : tws)
                for (int threads : {256, 512})
                  for (int sl : {1, 2, 4}) {
                    if (sl > 1 && (slices % sl != 0 || nsteps > steps)) continue;
                    const PwDwArgs v = with_tiles(a, th, tw, sl);
                    if (!pwdw_config_ok(v, threads)) continue;
                    // two slices per block halve the grid: offered where two blocks per CU remain (the batched early layers it is for;
                    // a batch-1 launch is a latency chain, its blocks must stay many and short)
                    if (sl > 1 && (double)a.N * v.tiles_y * v.tiles_x * v.slices < 512.0) continue;
                    // instruction slots of the busiest wave (a lone wave issues one instruction per 4 cycles): pointwise tiles
                    // (address + K steps + requantisation) and depthwise tasks, on top of a fixed prologue
                    const int nw = threads / 64;
                    const double vp = (double)std::min(v.RH, a.H) * std::min(v.RW, a.W);
                    const double tiles_w = std::ceil(std::ceil(vp / 16.0) / nw);
                    const int twl = a.S == 1 ? 2 : 1;
                    const double tasks_t = std::ceil((double)th * ((tw + twl - 1) / twl) * 4.0 * sl / threads);
                    // (a second slice repeats the K steps and the requantisation of a tile, not its address arithmetic, load and loop control)
                    const double block = 250.0 + tiles_w * (45.0 + sl * (25.0 + 3.0 * nsteps)) + tasks_t * (a.S == 1 ? 150.0 : 100.0) + vp * ktot / 400.0;
                    const double blocks = (double)a.N * v.tiles_y * v.tiles_x * v.slices;
                    const double rounds = std::ceil(blocks / (256.0 * (threads == 256 ? 2 : 1)));
                    cfgs.push_back({th, tw, threads, rounds * block * (threads == 256 && blocks > 256 ? 1.3 : 1.0), sl});
                }
        std::sort(cfgs.begin(), cfgs.end(), [](const Cfg& l, const Cfg& r) { return l.cost < r.cost; });
        // the best few of EACH block width go to the device: the model ranks within a width, the race decides between them
        std::vector<Cfg> keep;
        for (int sl : {1, 2, 4}) {
            int n = 0;
            for (auto& c : cfgs)
                if (c.sl == sl && n < (sl == 1 ? 8 : 6)) { keep.push_back(c); n++; }
        }
        cfgs = keep;
    }
    if (const char* pin = tamd_pin("pwdw_cfg")) {
        int th = 0, tw = 0, threads = 0, sl = 1;      // "THxTWxthreads" or "THxTWxthreadsx2" / "..x4" (two / four slices per block)
        if (sscanf(pin, "%dx%dx%dx%d", &th, &tw, &threads, &sl) >= 3 && tmode == 1) {
            th = std::min(th, a.OH); tw = std::min(tw, a.OW);
            if ((sl != 2 && sl != 4) || slices % sl != 0 || nsteps > steps) sl = 1;
            if (th >= 1 && tw >= 1 && pwdw_config_ok(with_tiles(a, th, tw, sl), threads)) { cfgs.clear(); cfgs.push_back({th, tw, threads, 0.0, sl}); }
        }
    }
    if (cfgs.empty()) return 0;
    Step& sa = g->steps[s0];
    Step& sb = g->steps[s0 + 1];
    const bool autotune = autotune_enabled() && sa.macs >= 4e6;
    size_t best = 0;
    // fused by construction (no race): a link of a LATENCY chain only -- batch 1.  Everything batched keeps the race against the two-launch
    // plan below (batch 8 at 56x56 or batch 16 at 19x19 are throughput launches; round 6's first evidence pass showed what the
    // wider rule costs where the race is skipped: MobileNet-v1 b64's 7x7 tail fused by construction ran conv6/sep + pool6 in 34.9 us
    // against 9.9 + 3.5 us as two launches, and conv5_6/sep + conv6/dw in 18.5 against 7.5 + 5.8 -- 287 instead of 265 us per step)
    bool fuse = fmode == 2 || (a.N == 1 && (double)a.N * a.H * a.W <= 32768.0);
    // cost model inputs below use the map the tail reads (a.H x a.W) and the reduction depth
    char ckey[256];
    snprintf(ckey, sizeof(ckey), "pwdw|%s|n%d %dx%d k%d m%d f%d c%zu", sa.node.c_str(), a.N, a.H, a.W, a.ktot, tmode, fmode, cfgs.size());
    std::string cached;
    int cf = 0, cb = 0;
    bool from_cache = false;
    if (autotune && plan_cache_get(ckey, &cached) && sscanf(cached.c_str(), "%d,%d", &cf, &cb) == 2 && cb >= 0 && cb < (int)cfgs.size()) {
        // a cached index is only as good as the file it came from: the configuration must still launch here
        const PwDwArgs v = tmode == 1 ? with_tiles(a, cfgs[cb].th, cfgs[cb].tw, cfgs[cb].sl) : a;
        if (!cf || launch_pwdw(v, cfgs[cb].threads, g->stream) == hipSuccess) { fuse = cf != 0; best = (size_t)cb; from_cache = true; }
        else (void)hipGetLastError();
    }
    if (from_cache) {
    } else if (autotune) {
        float best_ms = 1e30f;
        for (size_t c = 0; c < cfgs.size(); c++) {
            const PwDwArgs v = tmode == 1 ? with_tiles(a, cfgs[c].th, cfgs[c].tw, cfgs[c].sl) : a;
            const int threads = cfgs[c].threads;
            float ms;
            if (time_fn(g, [v, threads](hipStream_t s) { return launch_pwdw(v, threads, s); }, &ms)) return -1;
            if (ms < best_ms) { best_ms = ms; best = c; }
        }
        // A small pair is a link of a latency chain (batch 1: 3.4 us per dependent launch whatever it does): one launch instead of two
        // is right by construction there, and the race -- which times a launch back to back with ITSELF, i.e. its throughput -- gets
        // exactly these wrong now and then (conv6/sep + pool6 left as two launches: 54.9 instead of 51.4 us per MobileNet-v1 pass,
        // profiles/r05_ab_b1_call12_vs_now_v2.txt, r05_ab_firstdw_pingpong_mobilenet_v1_b1.txt).  Timed: the batched pairs only.
        if (fmode != 2 && !fuse) {
            float ta, tb;
            if (time_fn(g, sa.fn, &ta) || time_fn(g, sb.fn, &tb)) return -1;
            fuse = best_ms < 0.97f * (ta + tb);
        }
        plan_cache_put(ckey, std::to_string(fuse ? 1 : 0) + "," + std::to_string(best));
    }
    if (!fuse) return 0;
    const PwDwArgs v = tmode == 1 ? with_tiles(a, cfgs[best].th, cfgs[best].tw, cfgs[best].sl) : a;
    const int threads = cfgs[best].threads;
    Step st;
    st.node = sa.node + "+" + sb.node;
    char nm[48];
    if (tmode == 1) snprintf(nm, sizeof(nm), "%s_i8<s%d,%dx%d,%d%s>", prod == 1 ? "firstdw" : "pwdw", a.S, v.TH, v.TW, threads, v.sl == 4 ? ",c64" : v.sl == 2 ? ",c32" : "");
    else snprintf(nm, sizeof(nm), "pwpool_i8<%d>", threads);
    st.kernel = nm;
    st.macs = sa.macs + sb.macs;
    st.bytes = sa.bytes + sb.bytes;      // SURVEY 8(d) accounting, per layer: the intermediate tensor still counts as algorithmic bytes
    st.fn = [v, threads](hipStream_t s) { return launch_pwdw(v, threads, s); };
    if (prod == 1 && tmode == 1) {       // the first layer pair: reads the graph input, writes the depthwise output, nothing else (run_steps: wrap)
        st.rd.push_back(access_of(x)); st.wr.push_back(access_of(y)); st.deps = true;
    }
    g->steps.resize(s0);
    g->steps.push_back(st);
    g->fused_away[pw.out[0]] = 1;
    return 0;
}

// ---- depthwise 3x3 (stride 1) + the pointwise conv that consumes it in one launch: dwpw.hip ------------------------------------
// Called with the pair planned as two steps at s0, s0 + 1 (the depthwise step, then whichever GEMM-family member the pointwise race
// chose).  Large batches only: at batch 1 the pointwise conv pairs with the depthwise BEHIND it instead (pwdw.hip), which this
// fusion would take away.  TAMD_FUSE_DWPW=0 never, =2 always (tests); default: the faster of the two by plan-time timing.
int plan_dwpw(tamd_graph* g, HNode& dw, HNode& pw, size_t s0)
{
    const char* env = getenv("TAMD_FUSE_DWPW");
    const int fmode = env ? atoi(env) : 1;
    if (!fmode || !g_last_dw_valid || !g_last_gemm_valid || !dwpw_applicable(g_last_dw, g_last_gemm)) return 0;
    const DwArgs& d = g_last_dw;
    const ConvArgs& c = g_last_gemm;
    if (fmode != 2 && (long)d.N * d.OH * d.OW < 4096) return 0;
    const HTensor& w = g->tensors[pw.in[1]];
    std::vector<int8_t> wp(dwpw_packed_bytes(c.cout, c.cin));
    dwpw_pack((const int8_t*)w.data.data(), c.cout, c.cin, wp.data());
    int8_t* dwf = nullptr;
    if (upload(g, wp, &dwf)) return -1;
    DwPwArgs a{};
    a.x = d.x; a.dw_w = d.w; a.dw_bias = d.bias; a.dw_wscale = d.wscale; a.dw_rq = d.rq;
    a.pw_wfrag = dwf; a.pw_bias = c.bias; a.pw_wscale = c.wscale; a.pw_rq = c.rq;
    a.y = c.y;
    a.N = d.N; a.H = d.H; a.W = d.W; a.C = d.C; a.cs_in = d.cs_in; a.cw = d.cw; a.OH = d.OH; a.OW = d.OW; a.PH = d.PH; a.PW = d.PW;
    a.cout = c.cout; a.ldc = c.ldc; a.c_off = c.c_off; a.c_limit = c.c_limit;
    Step& sa = g->steps[s0];
    Step& sb = g->steps[s0 + 1];
    bool fuse = fmode == 2;
    if (fmode != 2) {
        char ckey[256];
        snprintf(ckey, sizeof(ckey), "dwpw|%s|n%d %dx%d c%d>%d", sa.node.c_str(), d.N, d.OH, d.OW, d.C, c.cout);
        std::string cached;
        if (autotune_enabled() && plan_cache_get(ckey, &cached)) fuse = cached == "1";
        else if (autotune_enabled()) {
            float tf, ta, tb;
            if (time_fn(g, [a](hipStream_t s) { return launch_dwpw(a, s); }, &tf) || time_fn(g, sa.fn, &ta) || time_fn(g, sb.fn, &tb)) return -1;
            fuse = tf < 0.97f * (ta + tb);
            if (getenv("TAMD_DEBUG")) fprintf(stderr, "[tamd] %s + %s: dwpw %.2f us vs %.2f + %.2f us -> %s\n", sa.node.c_str(), sb.node.c_str(), 1e3 * tf, 1e3 * ta, 1e3 * tb, fuse ? "fused" : "two launches");
            plan_cache_put(ckey, fuse ? "1" : "0");
        }
    }
    if (!fuse) return 0;
    Step st;
    st.node = sa.node + "+" + sb.node;
    st.kernel = "dwpw_i8";
    st.macs = sa.macs + sb.macs;
    st.bytes = sa.bytes + sb.bytes;      // SURVEY 8(d) accounting, per layer: the intermediate tensor still counts as algorithmic bytes
    st.fn = [a](hipStream_t s) { return launch_dwpw(a, s); };
    st.rd.push_back(access_of(g->tensors[dw.in[0]])); st.wr.push_back(access_of(g->tensors[pw.out[0]])); st.deps = true;
    g->steps.resize(s0);
    g->steps.push_back(st);
    g->fused_away[dw.out[0]] = 1;
    return 1;
}

}  // namespace tamd
