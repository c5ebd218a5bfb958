// The int8 planner: device buffers (arena), requantisation folds, plan-time timing, the convolution / pooling planners and plan_i8()
// itself -- node list -> launch list (NHWC int8).  The pair fusions (pointwise + depthwise, depthwise + pointwise) are in
// graph_plan_pairs.hip.  Split out of graph.hip in round 6.
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

// ---------------------------------------------------------------------------------------------
// planner
// ---------------------------------------------------------------------------------------------
int dev_alloc(tamd_graph* g, void** p, size_t bytes, bool zero)
{
    // slack: the pointwise kernels read whole 64-byte K steps, up to 8 of them past a pixel row's last channel (those
    // bytes meet zero weights, but must be readable behind the last pixel of a buffer too)
    const size_t slack = 1024;
    const char* ae = tamd_pin("arena");                       // 0: one hipMalloc per buffer (round 1-3 behaviour; A/B runs)
    if (ae && atoi(ae) == 0) {
        HIPCHK(hipMalloc(p, bytes + slack));
        g->dev_allocs.push_back(*p);
        if (zero) HIPCHK(hipMemsetAsync(*p, 0, bytes + slack, g->stream));    // never the legacy stream: it would collide with another thread's capture
        return 0;
    }
    // bump allocation out of a few large chunks: a model's tensors, weights and per-channel vectors are hundreds of buffers, and
    // as separate hipMalloc ranges each brings its own page-table fragment -- inside a pass every launch then begins with
    // translation misses on memory it last touched a step ago.  One contiguous range per 32 MB .. 1 GB maps with large fragments.
    const size_t need = (bytes + slack + 255) & ~(size_t)255;
    DevArena* a = g->arenas.empty() ? nullptr : &g->arenas.back();
    if (!a || a->used + need > a->cap) {
        size_t cap = g->arenas.empty() ? ((size_t)32 << 20) : std::min<size_t>(2 * g->arenas.back().cap, (size_t)1 << 30);
        cap = (std::max(cap, need) + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
        DevArena na;
        HIPCHK(hipMalloc((void**)&na.base, cap));
        na.cap = cap;
        g->dev_allocs.push_back(na.base);
        g->arenas.push_back(na);
        a = &g->arenas.back();
    }
    *p = a->base + a->used;
    a->used += need;
    if (zero) HIPCHK(hipMemsetAsync(*p, 0, bytes + slack, g->stream));
    return 0;
}

void nhwc_geom(HTensor& t)
{
    if (t.dims.size() == 4) { t.n = t.dims[0]; t.c = t.dims[1]; t.h = t.dims[2]; t.w = t.dims[3]; }
    else if (t.dims.size() == 2) { t.n = t.dims[0]; t.c = t.dims[1]; t.h = t.w = 1; }
    else { t.n = 1; t.c = (int)t.elems(); t.h = t.w = 1; }
}

int count_consumers(const tamd_graph* g, int tensor)
{
    int c = 0;
    for (auto& n : g->nodes)
        for (int i : n.in) c += (i == tensor);
    for (auto& o : g->outputs) c += (o.tensor == tensor);
    return c;
}


// which formula the reference's score() selection lands on (conv_hcl_x86.c:351-371,
// conv_dw_hcl_x86.c:508-543, conv_ref.c:197-200)
int conv_mode(const tamd_conv_param& p, int batch, int cin, int cout)
{
    if (p.group == 1) return RQ_CONV_HCL;
    int cin_g = cin / p.group, cout_g = cout / p.group;
    if (p.kernel_h == p.kernel_w && batch == 1 && p.group > 1 && cin_g == 1 && cout_g == 1 && p.pad_h0 == p.pad_h1
        && p.pad_w0 == p.pad_w1 && p.dilation_h == 1 && p.dilation_w == 1 && p.kernel_h == 3
        && ((p.stride_h == 1 && p.stride_w == 1) || (p.stride_h == 2 && p.stride_w == 2)))
        return RQ_CONV_HCL;
    return RQ_CONV_REF;
}


// the reference's three requantisation formulas folded into (m1, m2[c], lo, hi, out_scale) -- epilogue.h.
// Host float arithmetic here is binary32, unfused (-ffp-contract=off), exactly the reference's expressions.
RqFold fold_requant(int mode, int act, float in_s, float out_s, const HTensor& w, int cout)
{
    RqFold r;
    r.m2.resize(cout);
    for (int i = 0; i < cout; i++) r.m2[i] = w.scales.size() == (size_t)cout ? w.scales[i] : w.scales[0];
    r.m1 = in_s; r.out_scale = out_s; r.lo = -FLT_MAX; r.hi = FLT_MAX;
    if (mode == RQ_CONV_HCL) {
        if (act == 0) r.lo = 0.f;
        if (act > 0) { r.lo = 0.f; r.hi = 6.f; }
    } else if (mode == RQ_CONV_REF) {
        r.m1 = 1.0f;
        for (int i = 0; i < cout; i++) { volatile float d = in_s * r.m2[i]; r.m2[i] = d; }
        if (act == 1) { r.lo = -1.f; r.hi = 1.f; }
        else if (act >= 0) { r.lo = 0.f; if (act == 6) r.hi = 6.f; }
    } else {   // RQ_FC
        r.m1 = 1.0f;
        for (int i = 0; i < cout; i++) { volatile float d = in_s * r.m2[i]; volatile float q = d / out_s; r.m2[i] = q; }
        r.out_scale = 1.0f;
    }
    return r;
}

// RqArgs of epilogue.h for one node: the reference chain's constants (the +-127.49 * out_scale saturation folded into lo / hi)
// and the fast path's window / multipliers.  Host float arithmetic here is binary32, unfused: q(lo) / q(hi) are the
// reference's own sat127(round(x / out_scale)) on the clamp bounds.  The fold is used only when every factor is an ordinary
// normal number (the error bound of epilogue.h assumes no underflow in the chain); otherwise thr = 2 hands every value to the chain.
static int host_q(float x, float s)
{
    volatile float d = x / s;
    const float r = roundf(d);
    return r > 127.f ? 127 : (r < -127.f ? -127 : (int)r);
}
static RqArgs host_rq(const RqFold& r, int cpad, std::vector<float>* mf, std::vector<float>* m2)
{
    RqArgs q{};
    volatile float lim = 127.49f * r.out_scale;
    q.m1 = r.m1; q.out_scale = r.out_scale;
    q.lo = std::max(r.lo, -(float)lim);
    q.hi = std::min(r.hi, (float)lim);
    auto ordinary = [](double v) { return std::isfinite(v) && std::fabs(v) >= 1e-30 && std::fabs(v) <= 1e30; };
    bool ok = ordinary(r.m1) && ordinary(r.out_scale) && r.out_scale > 0.f && r.m1 > 0.f && q.lo <= q.hi;
    for (float v : r.m2) ok = ok && (v == 0.f || (ordinary(v) && ordinary((double)r.m1 * v) && ordinary((double)r.m1 * v / r.out_scale)));
    mf->assign(cpad, 0.f);
    m2->assign(cpad, 1.f);
    for (size_t c = 0; c < r.m2.size() && c < (size_t)cpad; c++) {
        (*m2)[c] = r.m2[c];
        if (ok) (*mf)[c] = (float)((double)r.m1 * (double)r.m2[c] / (double)r.out_scale);
    }
    q.thr = ok ? 0x1p-13f : 2.0f;
    q.ylo = ok ? 128.f + (float)host_q(q.lo, r.out_scale) + 0.25f : 1.25f;
    q.yhi = ok ? 128.f + (float)host_q(q.hi, r.out_scale) + 0.75f : 255.75f;
    return q;
}
// timing experiments only (tools/exp/xcd_local.sh): TAMD_EXP_PLAIN_KERNELS=1 plans the ordinary (non-coherent) kernel
// instances under direct dispatch; TAMD_EXP_NOFENCE=1 strips the fences of ordinary launches AND skips the self-check -- the bytes
// of such a graph are NOT trustworthy (stale L1 lines), only its clock is looked at
bool exp_plain_kernels() { const char* e = exp_env("TAMD_EXP_PLAIN_KERNELS"); return e && atoi(e) == 1; }

// uploads both per-channel vectors; *wscale = the fast-path multipliers, rq->m2 = the chain's factors
int upload_rq(tamd_graph* g, const RqFold& r, int cpad, const float** wscale, RqArgs* rq)
{
    std::vector<float> mf, m2;
    *rq = host_rq(r, cpad, &mf, &m2);
    float *d0, *d1;
    if (upload(g, mf, &d0) || upload(g, m2, &d1)) return -1;
    *wscale = d0; rq->m2 = d1;
    return 0;
}

void* l2_flush_buffer()
{
    static std::mutex mu;
    static std::map<int, void*> per_dev;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    auto it = per_dev.find(dev);
    if (it != per_dev.end()) return it->second;
    void* p = nullptr;
    if (hipMalloc(&p, kL2FlushBytes) != hipSuccess) { (void)hipGetLastError(); p = nullptr; }
    per_dev[dev] = p;
    return p;
}

bool autotune_cold(tamd_graph* g)
{
    if (g->autotune_cold < 0) {
        const char* e = exp_env("TAMD_AUTOTUNE_COLD");              // 0: always warm, 1: always cold
        size_t bytes = 0;
        for (const HTensor& t : g->tensors)
            bytes += (t.ttype == TAMD_TT_VAR || t.ttype == TAMD_TT_INPUT) && t.n > 0 ? (size_t)t.n * t.h * t.w * (t.cs > 0 ? t.cs : t.c) : t.elems() * (t.dtype == TAMD_DT_FP32 ? 4 : 1);
        g->autotune_cold = e ? (atoi(e) != 0) : bytes > (size_t)(48u << 20);      // tensors + weights of one pass vs 32 MB of L2
    }
    return g->autotune_cold == 1;
}

// one candidate the way it runs inside a pass: the fill evicts its weights (and everything else) from the L2s, the step planned
// just before it -- as a rule the producer of its input -- runs again and leaves that input where a pass leaves it, then the
// candidate is timed on its own.  Five samples, the slowest dropped.
int time_cold(tamd_graph* g, void* flush, const std::function<hipError_t()>& launch, float* ms_out)
{
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    const Step* prev = nullptr;
    for (size_t i = g->steps.size(); i-- > 0 && !prev;)
        if (!g->steps[i].once) prev = &g->steps[i];
    float tot = 0.f, worst = 0.f;
    const int reps = 5;
    for (int it = 0; it < reps; it++) {
        float t = 0;
        HIPCHK(hipMemsetAsync(flush, it, kL2FlushBytes, g->stream));
        if (prev) (void)prev->fn(g->stream);
        HIPCHK(hipEventRecord(e0, g->stream));
        (void)launch();
        HIPCHK(hipEventRecord(e1, g->stream));
        HIPCHK(hipEventSynchronize(e1));
        HIPCHK(hipEventElapsedTime(&t, e0, e1));
        tot += t;
        worst = std::max(worst, t);
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    (void)hipGetLastError();
    *ms_out = (tot - worst) / (reps - 1);
    return 0;
}

// average duration of one launch of `fn` on the graph's stream (plan-time autotune): back to back, or each launch behind an
// L2-evicting fill (autotune_cold)
int time_fn(tamd_graph* g, const std::function<hipError_t(hipStream_t)>& fn, float* ms_out)
{
    hipEvent_t e0, e1;
    *ms_out = 1e30f;
    hipError_t err = fn(g->stream);
    if (err == hipSuccess) err = fn(g->stream);
    if (err != hipSuccess) { (void)hipGetLastError(); return 0; }
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    if (void* flush = autotune_cold(g) ? l2_flush_buffer() : nullptr) {
        hipEventDestroy(e0); hipEventDestroy(e1);
        return time_cold(g, flush, [&]() { return fn(g->stream); }, ms_out);
    }
    // best of two timed bursts (the ranking decides the plan: run-to-run noise of a single burst showed up as 5-10 % swings of
    // whole-model times); short kernels (batch-1 layers are a few microseconds) get longer bursts
    float ms = 1e30f;
    int reps = 8;
    for (int round = 0; round < 3; round++) {
        float t = 0;
        HIPCHK(hipEventRecord(e0, g->stream));
        for (int it = 0; it < reps; it++) (void)fn(g->stream);
        HIPCHK(hipEventRecord(e1, g->stream));
        HIPCHK(hipEventSynchronize(e1));
        HIPCHK(hipEventElapsedTime(&t, e0, e1));
        t /= reps;
        if (round == 0 && t <= 0.02f) { reps = 40; continue; }      // re-measure short kernels with a longer burst
        ms = std::min(ms, t);
        if (round == 0) round = 1;                                  // long kernel: bursts 0 and 2
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    *ms_out = ms;
    return 0;
}

bool autotune_enabled()
{
    const char* at_env = getenv("TAMD_AUTOTUNE");
    return !(at_env && atoi(at_env) == 0);
}


// pointwise weight panel in MFMA fragment order: [16-channel slice][64-deep K step][lane = (k block of 16) * 16 + channel][16 B];
// `wd` = [C][K] int8 rows (1x1 conv: K = cin; first conv: K = cin*KH*KW in OIHW order), zero padded to nsteps * 64
std::vector<int8_t> pack_pw_panel(const int8_t* wd, int C, int K, int nsteps)
{
    const int slices = (C + 15) / 16;
    std::vector<int8_t> wf((size_t)slices * nsteps * 1024, 0);
    for (int c = 0; c < C; c++)
        for (int k = 0; k < K; k++)
            wf[((size_t)((c >> 4) * nsteps + (k >> 6)) * 64 + ((k >> 4) & 3) * 16 + (c & 15)) * 16 + (k & 15)] = wd[(size_t)c * K + k];
    return wf;
}


struct FusedElt {            // an eltwise (+ReLU) node folded into the epilogue of the conv that produces its later operand
    int res_tensor;          // the other eltwise operand
    int elt_tensor;          // the eltwise node's own output (its scale)
    int out_tensor;          // where the result is stored: elt_tensor, or the ReLU's output when one follows
    int type;
    bool conv_is_first, relu;
};

// the arguments of the last first-layer convolution / pooling step planned on this thread: plan() reads them back when it turns
// the pair into ONE launch (conv_first_pool.hip)
static thread_local FirstArgs g_last_first;
static thread_local bool g_last_first_valid = false;
static thread_local PoolArgs g_last_pool;
// ... of the last depthwise 3x3 / implicit-GEMM convolution planned on this thread (dwpw.hip: depthwise -> pointwise in one launch)
thread_local DwArgs g_last_dw;
thread_local bool g_last_dw_valid = false;
thread_local ConvArgs g_last_gemm;
thread_local bool g_last_gemm_valid = false;

static int plan_conv(tamd_graph* g, HNode& n, bool as_fc, const FusedElt* fz = nullptr)
{
    HTensor& x = g->tensors[n.in[0]];
    HTensor& w = g->tensors[n.in[1]];
    HTensor* b = n.in.size() > 2 ? &g->tensors[n.in[2]] : nullptr;
    HTensor& y = g->tensors[n.out[0]];
    if (x.dtype != TAMD_DT_INT8 || w.dtype != TAMD_DT_INT8 || y.dtype != TAMD_DT_INT8) {
        set_error("conv/fc %s: only int8 is implemented on the device in this round (dtype %d)", n.name.c_str(), x.dtype);
        return -1;
    }
    if (x.scales.empty() || y.scales.empty() || w.scales.empty()) { set_error("%s: missing quant params", n.name.c_str()); return -1; }
    tamd_conv_param p{};
    int mode;
    if (as_fc) {   // FC == "valid" convolution whose kernel covers the whole input map; weight [out][c*h*w]
        p.kernel_h = x.h; p.kernel_w = x.w; p.stride_h = p.stride_w = 1; p.dilation_h = p.dilation_w = 1;
        p.group = 1; p.activation = -1; p.input_channel = x.c; p.output_channel = y.c;
        mode = RQ_FC;
        if ((size_t)w.elems() != (size_t)y.c * x.c * x.h * x.w) { set_error("fc %s: weight size mismatch", n.name.c_str()); return -1; }
    } else {
        p = n.p.conv;
        mode = conv_mode(p, g->formula_batch ? g->formula_batch : x.n, x.c, y.c);      // (a half of a pair: the whole graph's batch decides, graph.h)
    }
    const int cout = y.c, cin = x.c, group = p.group;
    const int cin_g = cin / group;
    const RqFold rqf = fold_requant(mode, p.activation, x.scales[0], y.scales[0], w, cout);
    const std::vector<float>& ws = rqf.m2;     // m2[c]
    const float in_scale = rqf.m1, out_scale = rqf.out_scale, rq_lo = rqf.lo, rq_hi = rqf.hi;
    const int8_t* wd = (const int8_t*)w.data.data();
    const int32_t* bd = b ? (const int32_t*)b->data.data() : nullptr;
    const int KH = p.kernel_h, KW = p.kernel_w;
    const double macs = (double)y.n * y.h * y.w * cout * cin_g * KH * KW;
    const double abytes = (double)x.n * x.h * x.w * cin + (double)y.n * y.h * y.w * cout + (double)cout * cin_g * KH * KW + 4.0 * cout;

    Step st;
    st.node = n.name; st.macs = macs; st.bytes = abytes;
    const bool is_dw = (group > 1 && group == cin && cout == cin);
    if (x.nchw_raw && group == 1 && cin <= 4 && cin * KH * KW <= 224 && cout <= 128
        && p.dilation_h * (KH - 1) < 256 && p.dilation_w * (KW - 1) < 256) {
        // ---- first layer from the NCHW graph input on MFMA ----
        const char* rows_env = tamd_pin("first_rows");                   // 0: always the generic gather kernel (tests; read at every prerun)
        const int kwp = (rows_env && atoi(rows_env) == 0) ? 0 : conv_first_kwp(cin, KH, KW, p.dilation_w);
        const int kreal = cin * KH * KW, kp = kwp ? rup(cin * KH * kwp, 32) : rup(kreal, 32), cpad = rup(cout, 32);
        std::vector<int8_t> wp((size_t)cpad * kp, 0);
        for (int co = 0; co < cout; co++) {
            if (!kwp) { memcpy(&wp[(size_t)co * kp], wd + (size_t)co * kreal, kreal); continue; }   // OIHW row as stored
            for (int r = 0; r < cin * KH; r++)                              // kx padded to kwp: a patch row is kwp consecutive bytes
                memcpy(&wp[(size_t)co * kp + (size_t)r * kwp], wd + (size_t)co * kreal + (size_t)r * KW, KW);
        }
        std::vector<int32_t> bp(cpad, 0);
        for (int c = 0; c < cout; c++) bp[c] = bd ? bd[c] : 0;
        FirstArgs a{};
        int8_t* dw_; int32_t* db_;
        if (upload(g, wp, &dw_) || upload(g, bp, &db_) || upload_rq(g, rqf, cpad, &a.wscale, &a.rq)) return -1;
        a.x = (const int8_t*)x.dptr; a.w = dw_; a.bias = db_; a.y = (int8_t*)y.dptr;
        a.N = x.n; a.C = cin; a.H = x.h; a.W = x.w; a.OH = y.h; a.OW = y.w; a.cout = cout; a.ldc = y.cs; a.c_off = y.c_off;
        a.c_limit = y.is_view ? cout : std::min(rup(cout, 16), y.cs - y.c_off);
        a.KH = KH; a.KW = KW; a.SH = p.stride_h; a.SW = p.stride_w; a.PH = p.pad_h0; a.PW = p.pad_w0;
        a.DH = p.dilation_h; a.DW = p.dilation_w; a.kp = kp; a.kwp = kwp;
        st.kernel = "conv_first_i8";
        st.fn = [a](hipStream_t s) { return launch_conv_first(a, s); };
        g_last_first = a; g_last_first_valid = true;
    } else if (x.nchw_raw || group != 1) {
        if (!x.nchw_raw && is_dw && KH == 3 && KW == 3 && p.dilation_h == 1 && p.dilation_w == 1 && p.stride_h == p.stride_w
            && (p.stride_h == 1 || p.stride_h == 2)) {
            // ---- depthwise 3x3 ----
            const int cw = rup(cin, 16);
            // [3 rows][cw] dwords {w[r][0], w[r][1], w[r][2], 0}: one v_dot4 operand per (row, channel)
            std::vector<int8_t> wp((size_t)3 * cw * 4, 0);
            for (int c = 0; c < cin; c++)
                for (int r = 0; r < 3; r++)
                    for (int kx = 0; kx < 3; kx++) wp[((size_t)r * cw + c) * 4 + kx] = wd[(size_t)c * 9 + r * 3 + kx];
            std::vector<int32_t> bp(cw, 0);
            for (int c = 0; c < cin; c++) bp[c] = bd ? bd[c] : 0;
            DwArgs a{};
            int8_t* dw_; int32_t* db_;
            if (upload(g, wp, &dw_) || upload(g, bp, &db_) || upload_rq(g, rqf, cw, &a.wscale, &a.rq)) return -1;
            a.x = (const int8_t*)x.dptr + x.c_off; a.w = dw_; a.bias = db_;
            a.y = (int8_t*)y.dptr;
            a.N = x.n; a.H = x.h; a.W = x.w; a.C = cin; a.cs_in = x.cs; a.cw = cw; a.OH = y.h; a.OW = y.w;
            a.ldc = y.cs; a.c_off = y.c_off; a.S = p.stride_h; a.PH = p.pad_h0; a.PW = p.pad_w0;
            st.kernel = dwconv3x3_kernel_name(a);
            st.fn = [a](hipStream_t s) { return launch_dwconv3x3(a, s); };
            g_last_dw = a; g_last_dw_valid = true;
        } else {
            // ---- generic direct (first layer from NCHW, grouped, non-3x3 depthwise) ----
            std::vector<int8_t> wv(wd, wd + w.elems());
            DirectArgs a{};
            int8_t* dw_; int32_t* db_ = nullptr;
            if (upload(g, wv, &dw_) || upload_rq(g, rqf, rup(cout, 4), &a.wscale, &a.rq)) return -1;
            if (bd) { std::vector<int32_t> bv(bd, bd + cout); if (upload(g, bv, &db_)) return -1; }
            a.x = (const int8_t*)x.dptr + (x.nchw_raw ? 0 : x.c_off); a.w = dw_; a.bias = db_;
            a.y = (int8_t*)y.dptr;
            a.N = x.n; a.C = cin; a.H = x.h; a.W = x.w; a.cs_in = x.nchw_raw ? 0 : x.cs;
            a.OH = y.h; a.OW = y.w; a.cout = cout; a.ldc = y.cs; a.c_off = y.c_off;
            a.KH = KH; a.KW = KW; a.SH = p.stride_h; a.SW = p.stride_w; a.PH = p.pad_h0; a.PW = p.pad_w0;
            a.DH = p.dilation_h; a.DW = p.dilation_w; a.group = group;
            st.kernel = "conv_direct_i8";
            st.fn = [a](hipStream_t s) { return launch_conv_direct(a, s); };
        }
    } else {
        // ---- implicit GEMM on MFMA ----
        const int ckp = rup(cin, 16);
        const int ktot = KH * KW * ckp;
        const int kpad = rup(ktot, 64);
        const int cout_pad = rup(cout, 128);
        if (KH * KW > 128) { set_error("conv %s: kernel %dx%d too large", n.name.c_str(), KH, KW); return -1; }
        std::vector<int8_t> wp((size_t)cout_pad * kpad + 256, 0);      // + tail: deep-K stages may read past the last row
        for (int co = 0; co < cout; co++)
            for (int ci = 0; ci < cin; ci++)
                for (int ky = 0; ky < KH; ky++)
                    for (int kx = 0; kx < KW; kx++)
                        wp[(size_t)co * kpad + (size_t)(ky * KW + kx) * ckp + ci] = wd[(((size_t)co * cin + ci) * KH + ky) * KW + kx];
        std::vector<int32_t> bp(cout_pad, 0);
        for (int c = 0; c < cout; c++) bp[c] = bd ? bd[c] : 0;
        ConvArgs a{};
        int8_t* dw_; int32_t* db_;
        if (upload(g, wp, &dw_) || upload(g, bp, &db_) || upload_rq(g, rqf, cout_pad, &a.wscale, &a.rq)) return -1;
        a.x = (const int8_t*)x.dptr + x.c_off; a.w = dw_; a.bias = db_; a.y = (int8_t*)y.dptr;
        a.N = x.n; a.H = x.h; a.W = x.w; a.cs_in = x.cs; a.ckp = ckp; a.OH = y.h; a.OW = y.w; a.cout = cout;
        a.ldc = y.cs; a.c_off = y.c_off; a.c_limit = y.is_view ? cout : std::min(rup(cout, 16), y.cs - y.c_off);
        a.KH = KH; a.KW = KW; a.SH = p.stride_h; a.SW = p.stride_w; a.PH = p.pad_h0; a.PW = p.pad_w0;
        a.DH = p.dilation_h; a.DW = p.dilation_w; a.cin = cin; a.ktot = ktot; a.kpad = kpad;
        if (!g->zero_page) { if (dev_alloc(g, &g->zero_page, 256, true)) return -1; }
        a.zeros = (const int8_t*)g->zero_page;
        a.mg_ohw = ((1ull << 40) + (unsigned)(y.h * y.w) - 1) / (unsigned)(y.h * y.w);
        a.mg_ow = ((1ull << 40) + (unsigned)y.w - 1) / (unsigned)y.w;
        a.M = y.n * y.h * y.w;
        a.cfg = -1;
        g_last_gemm = a; g_last_gemm_valid = !fz;
        if (fz) {      // conv -> eltwise (-> relu) in one launch: the conv's own int8 rounding is kept, see epilogue.h
            HTensor& r = g->tensors[fz->res_tensor];
            HTensor& o = g->tensors[fz->out_tensor];
            a.elt.res = (const int8_t*)r.dptr; a.elt.res_ldc = r.cs; a.elt.res_c_off = r.c_off;
            a.elt.type = fz->type; a.elt.conv_is_first = fz->conv_is_first ? 1 : 0;
            a.elt.s_conv = y.scales[0]; a.elt.s_res = r.scales[0];
            a.elt.out_scale = g->tensors[fz->elt_tensor].scales[0];
            a.elt.relu = fz->relu ? (o.scales[0] == a.elt.out_scale ? 2 : 1) : 0; a.elt.relu_out_scale = o.scales[0];
            {   // SUM (+ scale-keeping ReLU): the two-fma tail of epilogue.h when its error bound holds (S = mc + mr <= 2)
                const double sc = a.elt.s_conv, sr = a.elt.s_res, so = a.elt.out_scale;
                auto ordinary = [](double v) { return std::isfinite(v) && v >= 1e-30 && v <= 1e30; };
                const bool ok = fz->type == 2 && a.elt.relu != 1 && ordinary(sc) && ordinary(sr) && ordinary(so) && (sc + sr) / so <= 2.0;
                a.elt.thr = 0.f;
                if (ok && !(tamd_pin("elt_fold") && atoi(tamd_pin("elt_fold")) == 0)) {
                    const float e = 0x1p-13f;
                    a.elt.mc = (float)(sc / so); a.elt.mr = (float)(sr / so);
                    a.elt.k0 = (float)(128.5 + (double)e - 128.0 * ((double)a.elt.mc + (double)a.elt.mr));
                    a.elt.ylo = a.elt.relu ? 128.25f : 1.25f; a.elt.yhi = 255.75f; a.elt.thr = 2.f * e;
                }
            }
            a.y = (int8_t*)o.dptr; a.ldc = o.cs; a.c_off = o.c_off;
            a.c_limit = o.is_view ? cout : std::min(rup(cout, 16), o.cs - o.c_off);
            st.bytes += (double)r.n * r.h * r.w * r.c;
        }
        // candidates: every kernel of the family computes the same bytes (exact integer GEMM + the same epilogue), so
        // the choice is purely a matter of speed
        struct Cand { std::string name; std::function<hipError_t(hipStream_t)> fn; };
        std::vector<Cand> cands;
        // (the fused eltwise tail lives in the conv_igemm / conv_igemm2 / pw_stream epilogues)
        if (!fz && gemm_direct_applicable(a)) cands.push_back({"gemm_direct_i8", [a](hipStream_t s) { return launch_gemm_direct(a, s); }});
        if (pw_stream_applicable(a)) cands.push_back({"pw_stream_i8", [a](hipStream_t s) { return launch_pw_stream(a, s); }});
        if (pw_rows_applicable(a)) cands.push_back({"pw_rows_i8", [a](hipStream_t s) { return launch_pw_rows(a, s); }});
        if (conv_igemm2_applicable(a)) cands.push_back({conv_igemm2_kernel_name(a), [a](hipStream_t s) { return launch_conv_igemm2(a, s); }});
        // lean-loop kernels (conv_pgemm.hip): fragment-ordered weights, k x k activations as an LDS-resident patch
        {
            int8_t* packed[2] = {nullptr, nullptr};       // per cout-tile width (64 / 128), packed on first use
            int* geom[2] = {nullptr, nullptr};            // conv_pgemm_w.hip: the per-tile geometry table, per pixel-tile height (128 / 64)
            for (int v = 0; v < conv_pgemm_num_variants(); v++) {
                if (!conv_pgemm_applicable(a, v)) continue;
                if ((v & 2) && a.M >= 65536) continue;    // 64-pixel tiles: only where 128-pixel tiles leave CUs idle
                ConvArgs ap = a;
                conv_pgemm_prepare(ap, v);
                const int bn = conv_pgemm_bn(v), slot = bn == 128;
                if (!packed[slot]) {
                    std::vector<int8_t> wf(conv_pgemm_packed_bytes(ap, bn), 0);
                    conv_pgemm_pack(ap, wp.data(), cout_pad, bn, wf.data());
                    if (upload(g, wf, &packed[slot])) return -1;
                }
                ap.wfrag = packed[slot];
                if (v & 16) {
                    const int gs = (v & 2) ? 1 : 0;
                    if (!geom[gs]) {
                        std::vector<int> tab;
                        conv_pgemm_w_table(ap, tab);
                        if (upload(g, tab, &geom[gs])) return -1;
                    }
                    ap.pg_tab = geom[gs];
                }
                cands.push_back({conv_pgemm_kernel_name(ap), [ap](hipStream_t s) { return launch_conv_pgemm(ap, s); }});
            }
        }
        // small maps (batch-1 tails, 1x1-map FC): the lean 16-channel-slice kernel of pwdw.hip without a tail
        const bool is1x1 = KH == 1 && KW == 1 && p.stride_h == 1 && p.stride_w == 1 && !p.pad_h0 && !p.pad_h1 && !p.pad_w0 && !p.pad_w1;
        if (!fz && is1x1 && a.M <= 4096 && !(exp_env("TAMD_PW_SMALL") && atoi(exp_env("TAMD_PW_SMALL")) == 0)) {
            PwDwArgs v{};
            const int slices = (cout + 15) / 16, cws = slices * 16;
            const int steps = pwdw_steps((ckp + 63) / 64), nsteps = rup((ckp + 63) / 64, steps);
            std::vector<int8_t> w2(wd, wd + (size_t)cout * cin);
            const std::vector<int8_t> wf = pack_pw_panel(w2.data(), cout, cin, nsteps);
            std::vector<int32_t> b2(cws, 0);
            for (int c = 0; c < cout; c++) b2[c] = bd ? bd[c] : 0;
            int8_t* d0; int32_t* d1;
            if (upload(g, wf, &d0) || upload(g, b2, &d1) || upload_rq(g, rqf, cws, &v.wscale, &v.rq)) return -1;
            v.wf = d0; v.bias = d1;
            v.x = a.x; v.N = x.n; v.H = x.h; v.W = x.w; v.cs_in = x.cs; v.ktot = ckp; v.nsteps = nsteps; v.steps = steps;
            v.mode = 2; v.prod = 0; v.slices = slices; v.cw = cws;
            v.coherent = (g->opt.direct_dispatch && !exp_plain_kernels()) ? 1 : 0;
            v.tile_major = (double)x.h * x.w * x.cs > (double)cout * ckp && slices <= 65535 ? 1 : 0;
            v.y = a.y; v.ldc = a.ldc; v.c_off = a.c_off; v.c_limit = a.c_limit;
            v.S = 1; v.OH = x.h; v.OW = x.w; v.TW = x.w; v.tiles_x = 1; v.RH = 1; v.RW = x.w;
            for (int px : {64, 128, 256}) {          // pixels per block: 1, 2, 4 tiles of 16 per wave at 256 threads
                int th = std::max(1, std::min(x.h, px / std::max(1, x.w)));
                v.TH = th; v.tiles_y = (x.h + th - 1) / th;
                bool dup = false;
                for (auto& c : cands) dup |= c.name == "pw_small_i8<" + std::to_string(th) + ">";
                if (dup || !pwdw_config_ok(v, 256)) continue;
                const PwDwArgs vc = v;
                cands.push_back({"pw_small_i8<" + std::to_string(th) + ">", [vc](hipStream_t s) { return launch_pwdw(vc, 256, s); }});
            }
        }
        const bool heuristic_done = !cands.empty();
        const bool autotune = autotune_enabled() && st.macs >= 5e5;
        if (!heuristic_done || autotune) {
            if (autotune) {
                for (int c = 0; c < conv_igemm_num_cfgs(); c++) {
                    if ((c == 1 || c == 3) && cout > 256 && a.M > 4096) continue;       // slivers: never competitive there
                    if (!conv_igemm_cfg_ok(a, c)) continue;
                    ConvArgs ac = a; ac.cfg = c;
                    cands.push_back({conv_igemm_kernel_name(ac), [ac](hipStream_t s) { return launch_conv_igemm(ac, s); }});
                }
            } else
                cands.push_back({conv_igemm_kernel_name(a), [a](hipStream_t s) { return launch_conv_igemm(a, s); }});
        }
        if (const char* force = getenv("TAMD_FORCE_GEMM")) {     // tests: pin one member of the family (read at every prerun)
            const std::string want = force;
            std::vector<Cand> only;
            for (int c = 0; c < conv_igemm_num_cfgs(); c++) {
                ConvArgs ac = a; ac.cfg = c;
                if (want == "igemm" + std::to_string(c) && conv_igemm_cfg_ok(a, c)) only.push_back({conv_igemm_kernel_name(ac), [ac](hipStream_t s) { return launch_conv_igemm(ac, s); }});
            }
            for (auto& c : cands)
                if (c.name.find(want) == 0) only.push_back(c);
            if (!only.empty()) cands = only;
        }
        size_t best = 0;
        char ckey[256];
        snprintf(ckey, sizeof(ckey), "gemm|%s|%dx%dx%dx%d>%d k%dx%d s%d%s", n.name.c_str(), x.n, x.c, x.h, x.w, cout, KH, KW, p.stride_h, fz ? "+elt" : "");
        std::string cached;
        bool from_cache = false;
        if (autotune && cands.size() > 1 && plan_cache_get(ckey, &cached))
            for (size_t c = 0; c < cands.size() && !from_cache; c++)
                if (cands[c].name == cached) { best = c; from_cache = true; }
        if (autotune && cands.size() > 1 && !from_cache) {
            // plan-time autotune: a few timed launches of each candidate on the real buffers (outputs are overwritten
            // again by the first real run); the heuristics above remain the fallback (TAMD_AUTOTUNE=0)
            float best_ms = 1e30f;
            for (size_t c = 0; c < cands.size(); c++) {
                float ms;
                if (time_fn(g, cands[c].fn, &ms)) return -1;
                if (ms > 1e29f) continue;
                // the heuristic candidates come first: a later one has to win by more than the timing noise
                if (best_ms > 1e29f || ms < best_ms * 0.96f) { best_ms = ms; best = c; }
            }
            plan_cache_put(ckey, cands[best].name);
        }
        st.kernel = cands[best].name + (fz ? (fz->relu ? "+eltwise+relu" : "+eltwise") : "");
        st.fn = cands[best].fn;
    }
    if (!fz) {                           // reads its input, writes its output (constants aside), one launch: all a convolution / FC step touches
        st.rd.push_back(access_of(x)); st.wr.push_back(access_of(y)); st.deps = true;
    }
    g->steps.push_back(st);
    return 0;
}


static int plan_pool(tamd_graph* g, HNode& n)
{
    HTensor& x = g->tensors[n.in[0]];
    HTensor& y = g->tensors[n.out[0]];
    PoolGeom pg = pool_geom(n.p.pool, x.h, x.w);
    PoolArgs a{};
    a.x = (const int8_t*)x.dptr + x.c_off; a.y = (int8_t*)y.dptr;
    a.N = x.n; a.H = x.h; a.W = x.w; a.C = x.c; a.cs_in = x.cs; a.OH = y.h; a.OW = y.w; a.ldc = y.cs; a.c_off = y.c_off;
    a.KH = pg.kh; a.KW = pg.kw; a.SH = pg.sh; a.SW = pg.sw; a.PH = pg.ph0; a.PW = pg.pw0;
    a.method = n.p.pool.pool_method; a.caffe_flavor = n.p.pool.caffe_flavor;
    a.in_scale = x.scales[0]; a.out_scale = y.scales[0];
    const PoolArgs av = a;
    g_last_pool = a;
    Step st; st.node = n.name; st.kernel = "pool_i8";
    st.bytes = (double)x.n * x.h * x.w * x.c + (double)y.n * y.h * y.w * y.c;
    st.fn = [av](hipStream_t s) { return launch_pool(av, s); };
    g->steps.push_back(st);
    return 0;
}


int plan_i8(tamd_graph* g)
{
    // ---- 1. geometry + device buffers for every non-const tensor -------------------------------
    for (auto& t : g->tensors) if (t.ttype != TAMD_TT_CONST) nhwc_geom(t);
    // concat outputs own a buffer; their inputs become views when layouts allow (concat-by-offset:
    // concat/concat_kernel_ref_int8.c with in_scale == out_scale is a pure copy)
    // An input that cannot be written in place (its scale differs -> the reference rescales, concat_kernel_ref_int8.c:70-80;
    // channel count / offset not a multiple of 16; produced or also consumed by a kernel that does not address channel
    // slices; a graph input) keeps its own buffer and is copied by concat_copy_i8 at the concat's position.
    std::vector<int> view_of(g->tensors.size(), -1), view_off(g->tensors.size(), 0);
    std::vector<int> alias_of(g->tensors.size(), -1);
    // ---- DENSE tensors (round 6: the SSD head plumbing of an int8 graph).  The convolution stack lives in NHWC buffers
    // with padded channels; what Permute(0,2,3,1) / PriorBox produce, and everything that is only a re-reading of it (Flatten, Reshape,
    // Concat on any axis, Softmax on any axis), lives in the reference's own dense element order (HTensor::nchw_raw, cs = 0): those ops
    // are byte copies / views there (permute_ref.c:305-343, flatten_ref.c:74-80, reshape_ref.c:76-90), and a graph output needs no
    // layout pass.  perm_src[t] >= 0: t is NHWC tensor perm_src[t] seen through Permute(0,2,3,1) (+ views) -- a Concat reads the NHWC
    // buffer itself (flatcat_i8, kind 1) and the permuted copy is never written unless something else reads it too.
    const size_t NT = g->tensors.size();
    std::vector<char> dense(NT, 0), flat_concat(g->nodes.size(), 0), need_buf(NT, 0);
    std::vector<int> perm_src(NT, -1);
    {
        auto root = [&](int t) { for (int hop = 0; hop < 16 && alias_of[t] >= 0; hop++) t = alias_of[t]; return t; };
        for (size_t ni = 0; ni < g->nodes.size(); ni++) {
            HNode& n = g->nodes[ni];
            if (n.op == TAMD_OP_INPUT || n.op == TAMD_OP_CONST || n.in.empty() || n.out.empty()) continue;
            HTensor& x = g->tensors[n.in[0]];
            const int yo = n.out[0];
            switch (n.op) {
            case TAMD_OP_PERMUTE: {
                const int* o = n.p.perm.order;
                if (x.dims.size() != 4 || dense[n.in[0]] || !(o[0] == 0 && o[1] == 2 && o[2] == 3 && o[3] == 1)) {
                    set_error("permute %s: only order (0, 2, 3, 1) of a 4-D convolution-stack tensor runs on the device", n.name.c_str());
                    return -1;
                }
                dense[yo] = 1; perm_src[yo] = n.in[0];
                break;
            }
            case TAMD_OP_PRIORBOX: dense[yo] = 1; break;
            case TAMD_OP_RESHAPE: dense[yo] = 1; if (dense[n.in[0]]) { alias_of[yo] = n.in[0]; perm_src[yo] = perm_src[n.in[0]]; } break;
            case TAMD_OP_DROPOUT: case TAMD_OP_FLATTEN:
                if (dense[n.in[0]]) { dense[yo] = 1; alias_of[yo] = n.in[0]; perm_src[yo] = perm_src[n.in[0]]; }
                else {
                    // Flatten of an H x W map: the [N, C*H*W] result is the SAME NCHW element order; on the device it stays the NHWC
                    // buffer and keeps the 4-D geometry, so a following FC (== conv whose kernel covers the map), the NCHW output
                    // conversion and a flat Concat (kind 2) all see (c, h, w).  Set HERE, in node order: the Concat case below looks
                    // at the geometry of its inputs (a 2-D tensor that is a flattened MAP must not take the channel-concat path --
                    // it did until round 6 and wrote N*H*W rows into an N-row output)
                    alias_of[yo] = n.in[0];
                    HTensor& yy = g->tensors[yo];
                    yy.n = x.n; yy.c = x.c; yy.h = x.h; yy.w = x.w;
                }
                break;
            case TAMD_OP_SOFTMAX: if (dense[n.in[0]]) dense[yo] = 1; break;
            case TAMD_OP_CONCAT: {
                HTensor& y = g->tensors[yo];
                const int ax = n.p.concat.axis < 0 ? n.p.concat.axis + (int)y.dims.size() : n.p.concat.axis;
                bool any_dense = false, any_map = false;
                for (int i : n.in) { any_dense |= dense[i] != 0; any_map |= g->tensors[i].h * g->tensors[i].w > 1; }
                if (any_dense || ax != 1 || (y.dims.size() != 4 && any_map)) {
                    flat_concat[ni] = 1; dense[yo] = 1;
                }
                break;
            }
            default:
                for (int i : n.in)
                    if (g->tensors[i].ttype != TAMD_TT_CONST && dense[i]) {
                        set_error("%s reads %s, a tensor in dense (permuted / flattened) order: only Flatten, Reshape, Concat and Softmax do on the device", n.name.c_str(), g->tensors[i].name.c_str());
                        return -1;
                    }
            }
        }
        // who needs the bytes of a dense tensor in memory: everything but the flat Concat of a lazily permuted tensor (and its views)
        for (size_t ni = 0; ni < g->nodes.size(); ni++) {
            const HNode& n = g->nodes[ni];
            for (int i : n.in) {
                if (g->tensors[i].ttype == TAMD_TT_CONST || !dense[i]) continue;
                const bool is_alias_op = !n.out.empty() && alias_of[n.out[0]] == i;
                if (is_alias_op) continue;
                if (!(flat_concat[ni] && perm_src[i] >= 0)) need_buf[root(i)] = 1;
            }
        }
        for (auto& io : g->outputs) if (dense[io.tensor]) need_buf[root(io.tensor)] = 1;
        for (size_t t = 0; t < NT; t++) {
            if (!dense[t]) continue;
            if (need_buf[root((int)t)]) perm_src[t] = -1;              // materialised: its readers take the dense bytes
            HTensor& d = g->tensors[t];
            d.nchw_raw = true; d.cs = 0; d.c_off = 0;
        }
        for (size_t t = 0; t < NT; t++) {
            HTensor& d = g->tensors[t];
            if (!dense[t] || alias_of[t] >= 0) continue;
            if (d.dtype != TAMD_DT_INT8) { set_error("tensor %s: dtype %d not supported on the device yet", d.name.c_str(), d.dtype); return -1; }
            if (perm_src[t] >= 0) continue;                             // never written: read through its NHWC source
            if (dev_alloc(g, &d.dptr, d.elems(), true)) return -1;
        }
    }
    auto producer_op = [&](int t) { for (auto& n : g->nodes) if (!n.out.empty() && n.out[0] == t) return n.op; return -1; };
    auto slice_capable = [](int op) { return op == TAMD_OP_CONV || op == TAMD_OP_FC || op == TAMD_OP_POOL; };
    for (auto& n : g->nodes) {
        if (n.op != TAMD_OP_CONCAT || flat_concat[&n - g->nodes.data()]) continue;
        HTensor& y = g->tensors[n.out[0]];
        int ax = n.p.concat.axis < 0 ? n.p.concat.axis + (int)y.dims.size() : n.p.concat.axis;
        if (ax != 1 || y.dims.size() < 2) { set_error("concat %s: only the channel axis is supported on the device", n.name.c_str()); return -1; }
        if (y.scales.empty()) { set_error("concat %s: missing quant params", n.name.c_str()); return -1; }
        int off = 0;
        for (int i : n.in) {
            HTensor& x = g->tensors[i];
            if (x.dtype != y.dtype || x.scales.empty()) { set_error("concat %s: input %s: dtype / quant params", n.name.c_str(), x.name.c_str()); return -1; }
            bool ok = (x.c % 16 == 0) && (off % 16 == 0) && (x.scales[0] == y.scales[0] || n.in.size() == 1) && x.ttype == TAMD_TT_VAR
                      && view_of[i] < 0 && slice_capable(producer_op(i));
            for (auto& c : g->nodes)            // every other reader must cope with a channel slice too
                for (int ci : c.in)
                    if (ci == i && &c != &n && !(slice_capable(c.op) || c.op == TAMD_OP_CONCAT)) ok = false;
            int readers = 0;
            for (int ci : n.in) readers += (ci == i);
            if (readers > 1) ok = false;        // the same tensor twice: one copy per position
            if (ok) { view_of[i] = n.out[0]; view_off[i] = off; }
            off += x.c;
        }
    }
    // (identity ops -- Dropout, Flatten -- alias their input: alias_of, set by the dense analysis above)
    // graph inputs: NCHW staging; first conv with <=4 channels reads NCHW directly
    for (auto& io : g->inputs) {
        HTensor& t = g->tensors[io.tensor];
        io.bytes = t.elems() * esize(t.dtype);
        if (dev_alloc(g, &io.stage, io.bytes + 64, true)) return -1;     // slack: the first-layer kernel over-reads the last row by < 8 bytes
        HIPCHK(hipHostMalloc(&io.pinned, io.bytes, hipHostMallocDefault));
        bool direct = (t.dims.size() == 4 && t.c <= 4 && count_consumers(g, io.tensor) == 1);
        if (direct) {
            for (auto& n : g->nodes)
                if (n.op == TAMD_OP_CONV && n.in[0] == io.tensor && n.p.conv.group != 1) direct = false;
                else if (n.op != TAMD_OP_CONV && !n.in.empty() && n.in[0] == io.tensor) direct = false;
        }
        if (direct) { t.nchw_raw = true; t.dptr = io.stage; t.cs = 0; }
    }
    std::vector<size_t> own;                     // tensors that own a buffer (not a constant, a raw input, a view or an alias)
    for (size_t i = 0; i < g->tensors.size(); i++) {
        HTensor& t = g->tensors[i];
        if (t.ttype == TAMD_TT_CONST || t.nchw_raw) continue;
        if (view_of[i] >= 0 || alias_of[i] >= 0) continue;
        if (t.dtype != TAMD_DT_INT8) { set_error("tensor %s: dtype %d not supported on the device yet", t.name.c_str(), t.dtype); return -1; }
        t.cs = rup(t.c, 16);
        // a 1x1-map graph output written by conv/fc/pool (dword stores) keeps its channels dense, so the
        // NHWC buffer IS the reference's NCHW order and no output layout pass is needed
        if (t.h * t.w == 1 && t.c % 4 == 0 && count_consumers(g, (int)i) == 1) {
            bool is_out = false, dword_producer = false;
            for (auto& o : g->outputs) is_out |= (o.tensor == (int)i);
            for (auto& n : g->nodes)
                if (!n.out.empty() && n.out[0] == (int)i)
                    dword_producer = (n.op == TAMD_OP_CONV || n.op == TAMD_OP_FC || n.op == TAMD_OP_POOL || n.op == TAMD_OP_SOFTMAX);   // (softmax: byte stores, any stride)
            if (is_out && dword_producer) t.cs = t.c;
        }
        own.push_back(i);
    }
    // ---- activation buffers.  Tensors whose lifetimes cannot overlap share device memory (tamd_options.keep_tensors = 0, the
    // default): a pass then touches a fraction of the bytes -- ResNet-50 at batch 32 owns 345 MB of activations one by one, more
    // than the 256 MB last-level cache, but never has more than ~65 MB of them alive.  Lifetime of a buffer, in node positions
    // (the launch list follows the node order, except that a fused tail runs at ITS PRODUCER's position and a fused
    // eltwise / ReLU at the position of the convolution that absorbs it): written from `birth` = the earliest producer within two
    // hops above the node that produces it (covers both exceptions, conservatively), read until `death` = the last node that
    // names it (or a view / alias of it) as an input.  A launch reads and writes in one go, so buffers with birth == death of
    // another never share.  Graph inputs / outputs and tensors with padding channels (cs != c: their padding bytes are zero from
    // the allocation on and stay zero) keep their own buffers.
    g->pooled.assign(g->tensors.size(), 0);
    {
        const char* pe = getenv("TAMD_POOL");
        const bool pool = pe ? atoi(pe) != 0 : !g->opt.keep_tensors;
        const int NN = (int)g->nodes.size();
        auto root_of = [&](int t) { for (int hop = 0; hop < 8; hop++) { if (view_of[t] >= 0) t = view_of[t]; else if (alias_of[t] >= 0) t = alias_of[t]; else break; } return t; };
        std::vector<int> prod(g->tensors.size(), -1), birth(g->tensors.size(), NN), death(g->tensors.size(), -1);
        for (int ni = 0; ni < NN; ni++)
            for (int o : g->nodes[ni].out) prod[o] = ni;
        auto up = [&](int ni) { int e = ni; for (int i : g->nodes[ni].in) if (g->tensors[i].ttype != TAMD_TT_CONST && prod[i] >= 0) e = std::min(e, prod[i]); return e; };
        for (int ni = 0; ni < NN; ni++) {
            const HNode& n = g->nodes[ni];
            if (n.op == TAMD_OP_INPUT || n.op == TAMD_OP_CONST) continue;
            int e = ni;
            for (int i : n.in) if (g->tensors[i].ttype != TAMD_TT_CONST && prod[i] >= 0) e = std::min(e, up(prod[i]));
            for (int o : n.out) { const int r = root_of(o); birth[r] = std::min(birth[r], e); death[r] = std::max(death[r], ni); }
            for (int i : n.in) if (g->tensors[i].ttype != TAMD_TT_CONST) { const int r = root_of(i); death[r] = std::max(death[r], ni); }
        }
        // an NHWC tensor that a flat Concat reads THROUGH its Permute (perm_src) is read at the Concat's position, not at the Permute's
        for (int ni = 0; ni < NN; ni++)
            if (flat_concat[ni])
                for (int i : g->nodes[ni].in)
                    if (perm_src[i] >= 0) { const int r = root_of(perm_src[i]); death[r] = std::max(death[r], ni); }
        std::vector<char> pinned(g->tensors.size(), 0);
        for (auto& io : g->inputs) pinned[root_of(io.tensor)] = 1;
        for (auto& io : g->outputs) pinned[root_of(io.tensor)] = 1;
        struct Blk { size_t t, bytes, off; int b, d; };
        std::vector<Blk> blks;
        auto bytes_of = [&](const HTensor& t) { return ((size_t)t.n * t.h * t.w * t.cs + 1024 + 255) & ~(size_t)255; };
        for (size_t i : own) {
            HTensor& t = g->tensors[i];
            g->unpooled_bytes += bytes_of(t);
            if (pool && !pinned[i] && t.cs == t.c && death[i] >= 0 && birth[i] <= death[i]) blks.push_back({i, bytes_of(t), 0, birth[i], death[i]});
            else if (dev_alloc(g, &t.dptr, (size_t)t.n * t.h * t.w * t.cs, true)) return -1;
        }
        // greedy by size: the largest buffers first, each at the lowest offset that is free over its whole lifetime
        std::sort(blks.begin(), blks.end(), [](const Blk& a, const Blk& b) { return a.bytes != b.bytes ? a.bytes > b.bytes : a.t < b.t; });
        size_t total = 0;
        for (size_t k = 0; k < blks.size(); k++) {
            std::vector<std::pair<size_t, size_t>> busy;           // [offset, end) of placed buffers alive at the same time
            for (size_t j = 0; j < k; j++)
                if (blks[j].b <= blks[k].d && blks[k].b <= blks[j].d) busy.push_back({blks[j].off, blks[j].off + blks[j].bytes});
            std::sort(busy.begin(), busy.end());
            size_t off = 0;
            for (auto& r : busy) { if (off + blks[k].bytes <= r.first) break; off = std::max(off, r.second); }
            blks[k].off = off;
            total = std::max(total, off + blks[k].bytes);
        }
        if (!blks.empty()) {
            void* base = nullptr;
            if (dev_alloc(g, &base, total, true)) return -1;
            for (const Blk& b : blks) { g->tensors[b.t].dptr = (char*)base + b.off; g->pooled[b.t] = 1; }
            g->pool_bytes = total;
        }
        for (size_t i = 0; i < g->tensors.size(); i++)                 // views / aliases of a shared buffer are shared too
            if (g->tensors[i].ttype != TAMD_TT_CONST && (view_of[i] >= 0 || alias_of[i] >= 0) && g->pooled[root_of((int)i)]) g->pooled[i] = 1;
        if (getenv("TAMD_DEBUG"))
            fprintf(stderr, "[tamd] activations: %.1f MB one buffer per tensor, %zu of %zu buffers share %.1f MB\n", g->unpooled_bytes / 1048576.0, blks.size(),
                    own.size(), g->pool_bytes / 1048576.0);
    }
    // resolve views / aliases (nodes are in topological order; resolve chains iteratively)
    for (int pass = 0; pass < 4; pass++)
        for (size_t i = 0; i < g->tensors.size(); i++) {
            HTensor& t = g->tensors[i];
            if (view_of[i] >= 0) {
                HTensor& o = g->tensors[view_of[i]];
                t.dptr = o.dptr; t.cs = o.cs; t.c_off = o.c_off + view_off[i]; t.is_view = true;
            } else if (alias_of[i] >= 0) {
                HTensor& o = g->tensors[alias_of[i]];
                t.dptr = o.dptr; t.cs = o.cs; t.c_off = o.c_off; t.is_view = o.is_view;
            }
        }
    // input layout steps
    for (auto& io : g->inputs) {
        HTensor& t = g->tensors[io.tensor];
        if (t.nchw_raw) continue;
        LayoutArgs a{io.stage, t.dptr, t.n, t.c, t.h, t.w, t.cs, esize(t.dtype)};
        Step st; st.node = t.name; st.kernel = "nchw_to_nhwc";
        st.fn = [a](hipStream_t s) { return launch_nchw_to_nhwc(a, s); };
        g->in_steps.push_back(st);
    }
    // ---- 2. compile nodes ---------------------------------------------------------------------
    g->fused_away.assign(g->tensors.size(), 0);
    std::vector<char> fused(g->nodes.size(), 0);
    // conv -> eltwise (-> relu) fusion (ResNet: branch2c / branch1 + residual add + relu): the eltwise is
    // folded into the LATER of its two producers when that one is a group-1 GEMM conv whose output feeds nothing else
    std::vector<FusedElt> fuse_at(g->nodes.size());
    std::vector<char> has_fuse(g->nodes.size(), 0);
    const char* fuse_env = getenv("TAMD_FUSE_ELTWISE");
    auto producer = [&](int t) { for (size_t i = 0; i < g->nodes.size(); i++) if (!g->nodes[i].out.empty() && g->nodes[i].out[0] == t) return (int)i; return -1; };
    for (size_t ei = 0; ei < g->nodes.size() && !(fuse_env && atoi(fuse_env) == 0); ei++) {
        HNode& e = g->nodes[ei];
        if (e.op != TAMD_OP_ELTWISE || e.in.size() != 2) continue;
        const int ty = e.p.elt.type;
        if (ty != 0 && ty != 2 && ty != 4 && ty != 6) continue;
        HTensor& ta = g->tensors[e.in[0]];
        HTensor& tb = g->tensors[e.in[1]];
        HTensor& te = g->tensors[e.out[0]];
        if (ta.dims != tb.dims || ta.is_view || tb.is_view || te.is_view || ta.ttype == TAMD_TT_CONST || tb.ttype == TAMD_TT_CONST) continue;
        const int pa = producer(e.in[0]), pb = producer(e.in[1]);
        const int later = std::max(pa, pb), conv_in = later == pa ? 0 : 1;
        if (later < 0 || later >= (int)ei) continue;
        HNode& c = g->nodes[later];
        if (c.op != TAMD_OP_CONV || c.p.conv.group != 1 || g->tensors[c.in[0]].nchw_raw || has_fuse[later]) continue;
        if (c.p.conv.kernel_h * c.p.conv.kernel_w > 128 || count_consumers(g, e.in[conv_in]) != 1) continue;
        FusedElt fz{};
        fz.res_tensor = e.in[1 - conv_in]; fz.elt_tensor = e.out[0]; fz.out_tensor = e.out[0]; fz.type = ty;
        fz.conv_is_first = conv_in == 0; fz.relu = false;
        size_t relu_node = 0;
        if (count_consumers(g, e.out[0]) == 1)
            for (size_t nj = ei + 1; nj < g->nodes.size(); nj++) {
                HNode& r = g->nodes[nj];
                if (r.op == TAMD_OP_RELU && r.in[0] == e.out[0] && r.p.relu.negative_slope == 0.f && !g->tensors[r.out[0]].is_view) {
                    fz.relu = true; fz.out_tensor = r.out[0]; relu_node = nj;
                    break;
                }
            }
        fuse_at[later] = fz; has_fuse[later] = 1; fused[ei] = 1;
        g->fused_away[e.in[conv_in]] = 1;        // the conv's own int8 result only exists in registers
        if (fz.relu) { fused[relu_node] = 1; g->fused_away[e.out[0]] = 1; }
    }
    for (size_t ni = 0; ni < g->nodes.size(); ni++) {
        HNode& n = g->nodes[ni];
        if (fused[ni]) continue;
        switch (n.op) {
        case TAMD_OP_INPUT: case TAMD_OP_CONST: case TAMD_OP_DROPOUT: case TAMD_OP_FLATTEN:
            break;
        case TAMD_OP_PERMUTE: {            // permute_ref.c:305-343, order (0, 2, 3, 1): a byte permutation -- written only when somebody reads it
            HTensor& x = g->tensors[n.in[0]];
            HTensor& y = g->tensors[n.out[0]];
            if (perm_src[n.out[0]] >= 0) { g->fused_away[n.out[0]] = 1; break; }       // its Concat reads the NHWC buffer itself
            FlatCatI8Args a{};
            a.nsrc = 1; a.y = (int8_t*)y.dptr; a.outer = x.n; a.out_row = x.c * x.h * x.w; a.row_begin = 0; a.row_len = a.out_row;
            a.src[0] = FlatCatI8Src{(const int8_t*)x.dptr + x.c_off, 1, a.out_row, x.c, x.h * x.w, x.cs, 0, 1.f, 1};
            Step st; st.node = n.name; st.kernel = "permute_i8"; st.bytes = 2.0 * x.elems();
            st.fn = [a](hipStream_t s) { return launch_flatcat_i8(a, s); };
            g->steps.push_back(st);
            break;
        }
        case TAMD_OP_RESHAPE: {            // reshape_ref.c:76-90 (NCHW): the same bytes under another shape
            HTensor& x = g->tensors[n.in[0]];
            HTensor& y = g->tensors[n.out[0]];
            if (alias_of[n.out[0]] >= 0) { if (perm_src[n.out[0]] >= 0) g->fused_away[n.out[0]] = 1; break; }     // dense input: a view
            FlatCatI8Args a{};              // convolution-stack input: its NCHW element order, written once
            a.nsrc = 1; a.y = (int8_t*)y.dptr; a.outer = x.n; a.out_row = x.c * x.h * x.w; a.row_begin = 0; a.row_len = a.out_row;
            a.src[0] = FlatCatI8Src{(const int8_t*)x.dptr + x.c_off, 2, a.out_row, x.c, x.h * x.w, x.cs, 0, 1.f, 1};
            Step st; st.node = n.name; st.kernel = "reshape_i8"; st.bytes = 2.0 * x.elems();
            st.fn = [a](hipStream_t s) { return launch_flatcat_i8(a, s); };
            g->steps.push_back(st);
            break;
        }
        case TAMD_OP_PRIORBOX: {           // shapes only: evaluated here, once (graph_infer.hip priorbox_eval), quantised as priorbox_ref.c:195-210
            HTensor& x = g->tensors[n.in[0]];
            HTensor& y = g->tensors[n.out[0]];
            const HTensor& img = g->tensors[n.in[1]];
            if (y.scales.empty()) { set_error("priorbox %s: missing quant params", n.name.c_str()); return -1; }
            std::vector<float> boxes;
            std::vector<int8_t> q;
            priorbox_eval(n.p.priorbox, x.dims[2], x.dims[3], img.dims[2], img.dims[3], &boxes);
            if (boxes.size() != y.elems()) { set_error("priorbox %s: output shape mismatch", n.name.c_str()); return -1; }
            priorbox_quant_i8(boxes, y.scales[0], &q);
            HIPCHK(hipMemcpyAsync(y.dptr, q.data(), q.size(), hipMemcpyHostToDevice, g->stream));
            HIPCHK(hipStreamSynchronize(g->stream));
            y.prerun_const = true;
            break;
        }
        case TAMD_OP_CONCAT: {
            HTensor& y = g->tensors[n.out[0]];
            if (flat_concat[ni]) {
                // dense output: for every index in front of the axis each input is one contiguous run of the output's run; up to
                // kFlatCatMax inputs per launch (the six SSD heads of a Concat: one launch), each read in the reference's element order
                const int ax = n.p.concat.axis < 0 ? n.p.concat.axis + (int)y.dims.size() : n.p.concat.axis;
                if (y.scales.empty()) { set_error("concat %s: missing quant params", n.name.c_str()); return -1; }
                long outer = 1, inner = 1;
                for (int d = 0; d < ax; d++) outer *= y.dims[d];
                for (size_t d = ax + 1; d < y.dims.size(); d++) inner *= y.dims[d];
                const int out_row = (int)(y.dims[ax] * inner);
                bool all_const = true;
                for (int i : n.in) all_const &= g->tensors[i].prerun_const;
                int begin = 0;
                for (size_t k0 = 0; k0 < n.in.size(); k0 += kFlatCatMax) {
                    FlatCatI8Args a{};
                    a.y = (int8_t*)y.dptr; a.outer = outer; a.out_row = out_row; a.row_begin = begin;
                    Step st; st.node = n.name; st.kernel = "flatcat_i8"; st.once = all_const;
                    for (size_t k = k0; k < std::min(n.in.size(), k0 + (size_t)kFlatCatMax); k++) {
                        HTensor& x = g->tensors[n.in[k]];
                        if (x.dtype != y.dtype || x.scales.empty()) { set_error("concat %s: input %s: dtype / quant params", n.name.c_str(), x.name.c_str()); return -1; }
                        FlatCatI8Src s{};
                        s.begin = begin; s.identity = n.in.size() == 1;            // concat_kernel_ref_int8.c:47-57: a single input is copied as it is
                        volatile float rs = x.scales[0] / y.scales[0];             // :70 rescale = in_scale / out_scale
                        s.rescale = rs;
                        if (perm_src[n.in[k]] >= 0) {                              // Permute(0,2,3,1) (-> Flatten) of an NHWC tensor, read in place
                            HTensor& p = g->tensors[perm_src[n.in[k]]];
                            s.x = (const int8_t*)p.dptr + p.c_off; s.kind = 1; s.C = p.c; s.HW = p.h * p.w; s.cs = p.cs; s.chunk = p.c * p.h * p.w;
                            if (p.n != outer) { set_error("concat %s: a permuted input on axis %d", n.name.c_str(), ax); return -1; }
                            st.kernel = "permute_concat_i8";
                        } else if (dense[n.in[k]]) {
                            s.x = (const int8_t*)x.dptr; s.kind = 0; s.chunk = (int)(x.dims[ax] * inner);
                        } else {                                                   // a convolution-stack tensor, read in NCHW element order
                            s.x = (const int8_t*)x.dptr + x.c_off; s.kind = 2; s.C = x.c; s.HW = x.h * x.w; s.cs = x.cs;
                            s.chunk = (int)((size_t)x.n * x.c * x.h * x.w / (size_t)outer);
                        }
                        a.src[a.nsrc++] = s;
                        begin += s.chunk;
                        st.bytes += 2.0 * outer * s.chunk;
                    }
                    a.row_len = begin - a.row_begin;
                    st.fn = [a](hipStream_t s) { return launch_flatcat_i8(a, s); };
                    g->steps.push_back(st);
                }
                if (begin != out_row) { set_error("concat %s: the inputs do not add up to the output", n.name.c_str()); return -1; }
                y.prerun_const = all_const;
                break;
            }
            int off = 0;
            for (int i : n.in) {
                HTensor& x = g->tensors[i];
                if (!(view_of[i] == n.out[0] && view_off[i] == off && x.is_view)) {
                    CatCopyArgs a{};
                    a.x = (const int8_t*)x.dptr + x.c_off; a.y = (int8_t*)y.dptr;
                    a.pixels = (long)x.n * x.h * x.w; a.C = x.c; a.cs_in = x.cs; a.ldc = y.cs; a.c_off = y.c_off + off;
                    volatile float rs = x.scales[0] / y.scales[0];      // concat_kernel_ref_int8.c:70: rescale = in_scale / out_scale
                    a.rescale = rs;
                    a.identity = n.in.size() == 1;                      // :47-57: a single input is copied as it is
                    Step st; st.node = n.name; st.kernel = "concat_copy_i8"; st.bytes = 2.0 * a.pixels * x.c;
                    st.fn = [a](hipStream_t s) { return launch_concat_copy_i8(a, s); };
                    g->steps.push_back(st);
                }
                off += x.c;
            }
            break;
        }
        case TAMD_OP_CONV: {
            int tmode = -1, prod = 0;
            // depthwise 3x3 whose only consumer is a pointwise conv (large batches): one launch, the depthwise map stays in LDS (dwpw.hip).
            // `try_dwpw(dwi)`: the depthwise node dwi has just been planned as the LAST step (g_last_dw describes it); plans its
            // pointwise consumer behind it and lets plan_dwpw turn the two steps into one.  1: fused (the consumer is marked), 0: the
            // depthwise step stands alone and the consumer goes through the ordinary path later, -1: error
            auto try_dwpw = [&](size_t dwi) -> int {
                HNode& d = g->nodes[dwi];
                const HTensor& dy = g->tensors[d.out[0]];
                const char* dp_env = getenv("TAMD_FUSE_DWPW");
                if ((dp_env && atoi(dp_env) == 0) || !g_last_dw_valid || d.p.conv.stride_h != 1 || count_consumers(g, d.out[0]) != 1 || dy.is_view) return 0;
                if ((long)dy.n * dy.h * dy.w < 4096 && !(dp_env && atoi(dp_env) == 2)) return 0;
                for (auto& o : g->outputs) if (o.tensor == d.out[0]) return 0;
                int pw_node = -1;
                for (size_t nj = dwi + 1; nj < g->nodes.size(); nj++)
                    if (g->nodes[nj].op == TAMD_OP_CONV && g->nodes[nj].in.size() >= 2 && g->nodes[nj].in[0] == d.out[0] && !fused[nj] && !has_fuse[nj]
                        && g->nodes[nj].p.conv.group == 1 && g->nodes[nj].p.conv.kernel_h == 1 && g->nodes[nj].p.conv.kernel_w == 1) { pw_node = (int)nj; break; }
                if (pw_node < 0) return 0;
                {   // what dwpw_applicable will ask of the shapes, before the consumer is planned (and its weights uploaded) for nothing
                    const HTensor& py = g->tensors[g->nodes[pw_node].out[0]];
                    const tamd_conv_param& q = g->nodes[pw_node].p.conv;
                    if (py.c % 64 != 0 || py.c > 512 || dy.w > 16 || q.stride_h != 1 || q.stride_w != 1 || q.pad_h0 || q.pad_w0 || q.pad_h1 || q.pad_w1) return 0;
                }
                const size_t sdw = g->steps.size() - 1;
                g_last_gemm_valid = false;
                if (plan_conv(g, g->nodes[pw_node], false)) return -1;
                int r = 0;
                if (g->steps.size() == sdw + 2) r = plan_dwpw(g, d, g->nodes[pw_node], sdw);
                if (r < 0) return -1;
                if (r == 1) { fused[pw_node] = 1; return 1; }
                g->steps.resize(sdw + 1);                      // not fused: forget the trial plan of the consumer
                return 0;
            };
            if (!has_fuse[ni] && n.p.conv.group > 1 && n.p.conv.group == g->tensors[n.in[0]].c && n.p.conv.kernel_h == 3 && n.p.conv.stride_h == 1
                && !g->tensors[n.in[0]].nchw_raw) {
                const size_t s0 = g->steps.size();
                g_last_dw_valid = false;
                if (plan_conv(g, n, false)) return -1;
                if (g->steps.size() == s0 + 1 && try_dwpw(ni) < 0) return -1;
                break;
            }
            // stem: first-layer convolution whose only consumer is a MAX pool 3x3 / 2 -> one launch, the conv map stays in LDS
            // (TAMD_FIRST_POOL=0: two launches, for A/B runs and the fused == unfused tests)
            if (!has_fuse[ni] && g->tensors[n.in[0]].nchw_raw && count_consumers(g, n.out[0]) == 1) {
                int pool_node = -1;
                for (size_t nj = ni + 1; nj < g->nodes.size(); nj++)
                    if (g->nodes[nj].op == TAMD_OP_POOL && g->nodes[nj].in[0] == n.out[0] && !fused[nj]) { pool_node = (int)nj; break; }
                const char* fp_env = tamd_pin("first_pool");
                bool is_out = false;
                for (auto& o : g->outputs) is_out |= (o.tensor == n.out[0]);
                if (pool_node >= 0 && !is_out && !(fp_env && atoi(fp_env) == 0)) {
                    const size_t s0 = g->steps.size();
                    g_last_first_valid = false;
                    if (plan_conv(g, n, false)) return -1;
                    if (plan_pool(g, g->nodes[pool_node])) return -1;
                    fused[pool_node] = 1;
                    if (g->steps.size() == s0 + 2 && g_last_first_valid && conv_first_pool_applicable(g_last_first, g_last_pool)) {
                        const FirstPoolArgs fa = conv_first_pool_args(g_last_first, g_last_pool);
                        Step st;
                        st.node = g->steps[s0].node + "+" + g->steps[s0 + 1].node;
                        st.kernel = "conv_first_pool_i8";
                        st.macs = g->steps[s0].macs;
                        st.bytes = g->steps[s0].bytes + g->steps[s0 + 1].bytes;      // SURVEY 8(d) accounting, per layer: the conv map still counts
                        st.fn = [fa](hipStream_t s) { return launch_conv_first_pool(fa, s); };
                        st.rd.push_back(access_of(g->tensors[n.in[0]])); st.wr.push_back(access_of(g->tensors[g->nodes[pool_node].out[0]])); st.deps = true;
                        g->steps.resize(s0);
                        g->steps.push_back(st);
                        g->fused_away[n.out[0]] = 1;
                    }
                    break;
                }
            }
            const int tail = has_fuse[ni] ? -1 : find_pwdw_tail(g, ni, &tmode, &prod);
            if (tail >= 0 && !fused[tail]) {
                // the pair is planned here, the tail ahead of its node order (its only input is this conv's output), then
                // possibly replaced by ONE fused launch
                const size_t s0 = g->steps.size();
                g_last_dw_valid = false;
                if (plan_conv(g, n, false)) return -1;
                if (tmode == 0 ? plan_pool(g, g->nodes[tail]) : plan_conv(g, g->nodes[tail], false)) return -1;
                fused[tail] = 1;
                g_last_dw_valid = g_last_dw_valid && tmode == 1;
                // Where the depthwise tail can go together with ITS consumer (dwpw.hip: batched 14x14-class maps, stride 1), that pairing is
                // tried FIRST.  In a chain pw, dw, pw, dw, .. either pairing covers every layer once per period, and in a pass dwpw is the
                // cheaper period (MobileNet-v1 b64: 16.4 us against 22.9 us for the pwdw pair with two slices per block) -- but the
                // plan-time race, which times a launch back to back with itself, saw the pwdw pair at < 18.8 us and took it
                // (profiles/r05_layers_mobilenet_v1_int8_b64.txt, the evidence plan: the 14x14 block 120 us; with this order 95 us)
                int took = 0;
                if (tmode == 1 && g->steps.size() == s0 + 2) { took = try_dwpw((size_t)tail); if (took < 0) return -1; }
                if (!took && g->steps.size() == s0 + 2 && plan_pwdw(g, n, g->nodes[tail], tmode, prod, s0)) return -1;
                break;
            }
            if (plan_conv(g, n, false, has_fuse[ni] ? &fuse_at[ni] : nullptr)) return -1;
            break;
        }
        case TAMD_OP_FC:
            if (plan_conv(g, n, true)) return -1;
            break;
        case TAMD_OP_POOL:
            if (plan_pool(g, n)) return -1;
            break;
        case TAMD_OP_SOFTMAX: {            // ResNet-50's prob; softmax_kernel_ref_int8.c over the channel axis
            HTensor& x = g->tensors[n.in[0]];
            HTensor& y = g->tensors[n.out[0]];
            int ax = n.p.softmax.axis < 0 ? n.p.softmax.axis + (int)x.dims.size() : n.p.softmax.axis;
            if (x.scales.empty() || y.scales.empty()) { set_error("softmax %s: missing quant params", n.name.c_str()); return -1; }
            const bool general = dense[n.in[0]] || (x.dims.size() == 4 && (ax == 2 || ax == 3));
            if (general) {
                // round 6: any axis of a dense tensor (the SSD tail: Reshape -> Softmax(axis 2) on [N, priors, classes]) and the spatial
                // axes of an NHWC tensor -- the same kernel with strided addressing (kernels.h: SoftmaxI8Args)
                if (ax < 0 || ax >= (int)x.dims.size() || x.dims[ax] < 1 || x.dims[ax] > kSoftmaxI8MaxC) {
                    set_error("softmax %s: axis %d of at most %d values", n.name.c_str(), ax, kSoftmaxI8MaxC);
                    return -1;
                }
                SoftmaxI8Args a{};
                a.x = (const int8_t*)x.dptr + x.c_off; a.y = (int8_t*)y.dptr + y.c_off;
                a.C = x.dims[ax]; a.in_scale = x.scales[0]; a.out_scale = y.scales[0];
                if (dense[n.in[0]]) {
                    long outer = 1, inner = 1;
                    for (int d = 0; d < ax; d++) outer *= x.dims[d];
                    for (size_t d = ax + 1; d < x.dims.size(); d++) inner *= x.dims[d];
                    a.positions = outer * inner; a.d1 = a.d2 = inner; a.is1 = a.os1 = (long)a.C * inner; a.is2 = a.os2 = 0; a.istride = a.ostride = inner;
                } else if (ax == 3) {
                    a.positions = (long)x.n * x.h * x.c; a.d1 = a.d2 = x.c;
                    a.is1 = (long)x.w * x.cs; a.os1 = (long)y.w * y.cs; a.is2 = a.os2 = 0; a.istride = x.cs; a.ostride = y.cs;
                } else {
                    a.positions = (long)x.n * x.w * x.c; a.d1 = (long)x.w * x.c; a.d2 = x.c;
                    a.is1 = (long)x.h * x.w * x.cs; a.os1 = (long)y.h * y.w * y.cs; a.is2 = x.cs; a.os2 = y.cs;
                    a.istride = (long)x.w * x.cs; a.ostride = (long)y.w * y.cs;
                }
                Step st; st.node = n.name; st.kernel = "softmax_i8"; st.bytes = 2.0 * (double)x.elems();
                st.fn = [a](hipStream_t s) { return launch_softmax_i8(a, s); };
                g->steps.push_back(st);
                break;
            }
            if (ax != 1 || (x.dims.size() != 2 && x.dims.size() != 4) || x.c < 1 || x.c > kSoftmaxI8MaxC) {
                set_error("softmax %s is not supported on the device: int8 softmax runs over the channel axis of a 2-D / 4-D tensor of at most %d channels",
                          n.name.c_str(), kSoftmaxI8MaxC);
                return -1;
            }
            if (x.dims.size() == 2 && x.h * x.w != 1) {
                // a 2-D tensor that is the flattened view of an H x W > 1 map keeps the map's NHWC geometry on the device: its "channel
                // axis" is C, the reference normalises over all C*H*W values in NCHW order.  The plugin leaves such a node to
                // the CPU device (hip_device.cc: node_runs_on_device); through the C ABI it is refused here
                set_error("softmax %s is not supported on the device: its 2-D input is the flattened view of a %d x %d map", n.name.c_str(), x.h, x.w);
                return -1;
            }
            if (x.scales.empty() || y.scales.empty()) { set_error("softmax %s: missing quant params", n.name.c_str()); return -1; }
            SoftmaxI8Args a{};
            a.x = (const int8_t*)x.dptr + x.c_off; a.y = (int8_t*)y.dptr + y.c_off;
            a.positions = (long)x.n * x.h * x.w; a.C = x.c; a.cs_in = x.cs; a.cs_out = y.cs;
            a.in_scale = x.scales[0]; a.out_scale = y.scales[0];
            Step st; st.node = n.name; st.kernel = "softmax_i8"; st.bytes = 2.0 * a.positions * x.c;
            st.fn = [a](hipStream_t s) { return launch_softmax_i8(a, s); };
            g->steps.push_back(st);
            break;
        }
        case TAMD_OP_RELU: {
            HTensor& x = g->tensors[n.in[0]];
            HTensor& y = g->tensors[n.out[0]];
            if (x.is_view || y.is_view) { set_error("relu %s on a concat view is not supported", n.name.c_str()); return -1; }
            ReluArgs a{(const int8_t*)x.dptr, (int8_t*)y.dptr, (size_t)x.n * x.h * x.w * x.cs, n.p.relu.negative_slope, x.scales[0], y.scales[0]};
            Step st; st.node = n.name; st.kernel = "relu_i8"; st.bytes = 2.0 * x.n * x.h * x.w * x.c;
            st.fn = [a](hipStream_t s) { return launch_relu(a, s); };
            g->steps.push_back(st);
            break;
        }
        case TAMD_OP_ELTWISE: {
            HTensor& xa = g->tensors[n.in[0]];
            HTensor& xb = g->tensors[n.in[1]];
            HTensor* y = &g->tensors[n.out[0]];
            if (xa.is_view || xb.is_view || y->is_view || xa.dims != xb.dims) { set_error("eltwise %s: views / broadcast not supported", n.name.c_str()); return -1; }
            EltArgs a{};
            a.a = (const int8_t*)xa.dptr; a.b = (const int8_t*)xb.dptr; a.count = (size_t)xa.n * xa.h * xa.w * xa.cs;
            a.type = n.p.elt.type; a.sa = xa.scales[0]; a.sb = xb.scales[0]; a.out_scale = y->scales[0];
            if (a.type != 0 && a.type != 2 && a.type != 4 && a.type != 6) { set_error("eltwise %s: type %d unsupported", n.name.c_str(), a.type); return -1; }
            std::string kname = "eltwise_i8";
            double bytes = 3.0 * xa.n * xa.h * xa.w * xa.c;
            // fuse the standalone ReLU that follows (ResNet: 16 x eltwise -> relu), SURVEY §8f-1
            if (count_consumers(g, n.out[0]) == 1) {
                for (size_t nj = ni + 1; nj < g->nodes.size(); nj++) {
                    HNode& r = g->nodes[nj];
                    if (r.op == TAMD_OP_RELU && r.in[0] == n.out[0] && r.p.relu.negative_slope == 0.f) {
                        HTensor& ry = g->tensors[r.out[0]];
                        if (ry.is_view) break;
                        a.fuse_relu = ry.scales[0] == a.out_scale ? 2 : 1; a.relu_out_scale = ry.scales[0];
                        y = &ry; fused[nj] = 1; kname = "eltwise_relu_i8";
                        g->fused_away[n.out[0]] = 1;
                        break;
                    }
                }
            }
            a.y = (int8_t*)y->dptr;
            Step st; st.node = n.name; st.kernel = kname; st.bytes = bytes;
            st.fn = [a](hipStream_t s) { return launch_eltwise(a, s); };
            g->steps.push_back(st);
            break;
        }
        default:
            set_error("op %d (%s) is not supported on the device", n.op, n.name.c_str());
            return -1;
        }
    }
    // ---- 3. outputs: NHWC -> the reference's NCHW order ------------------------------------------
    for (auto& io : g->outputs) {
        HTensor& t = g->tensors[io.tensor];
        io.bytes = t.elems() * esize(t.dtype);
        HIPCHK(hipHostMalloc(&io.pinned, io.bytes, hipHostMallocDefault));
        if (t.nchw_raw) { io.stage = t.dptr; continue; }                 // dense tensors are the reference's element order already
        if (t.h * t.w == 1 && t.cs == t.c && t.c_off == 0) { io.stage = t.dptr; continue; }
        if (dev_alloc(g, &io.stage, io.bytes, true)) return -1;
        LayoutArgs a{(const int8_t*)t.dptr + t.c_off, io.stage, t.n, t.c, t.h, t.w, t.cs, esize(t.dtype)};
        Step st; st.node = t.name; st.kernel = "nhwc_to_nchw";
        st.fn = [a](hipStream_t s) { return launch_nhwc_to_nchw(a, s); };
        g->out_steps.push_back(st);
    }
    return 0;
}

// io_slot < 0: the device-resident launch list; 0 | 1: with the upload of every input from / the download of every output to

}  // namespace tamd
