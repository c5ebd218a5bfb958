// Planner for fp32 device graphs (config #1 plumbing, parity bar 1e-4, order-free).
// Dense NCHW fp32 tensors -- the reference's own order, no layout pass at the graph edges.  group == 1 convolutions
// and FC run on the matrix cores (conv_f32_mfma.hip, LDS-DMA operand ring).  3x3 / stride 1 convolutions also have a
// Winograd F(2,3) form (winograd_f32.hip; the reference's CPU backend uses F(4,3), wino_conv_kernel_x86.c -- the same
// mathematical result): the planner times both and keeps the faster (TAMD_F32_WINOGRAD=0 never, 1 always).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include <algorithm>

#include "graph.h"

namespace tamd {

static inline int rup(int v, int m) { return (v + m - 1) / m * m; }

// conv (group 1) or FC as [cout] x [pixels] x [K] GEMM on the matrix cores
static int plan_gemm_f32(tamd_graph* g, HNode& n, const float* xdev, int N, int C, int H, int W, int OH, int OW, int cout,
                         int KH, int KW, int SH, int SW, int PH, int PW, int DH, int DW, int act, float* ydev, int oimg, int oc0)
{
    HTensor& w = g->tensors[n.in[1]];
    HTensor* b = n.in.size() > 2 ? &g->tensors[n.in[2]] : nullptr;
    const int K = C * KH * KW, Kpad = rup(K, 32), nstage = Kpad / 32;
    if (w.dtype != TAMD_DT_FP32 || (size_t)cout * K * 4 != w.data.size()) { set_error("%s: fp32 weight size mismatch", n.name.c_str()); return -1; }
    if ((KH - 1) * DH > 15 || (KW - 1) * DW > 15 || (size_t)C * H * W >= (1u << 24)) { set_error("%s: kernel extent / image size outside the packed tap table", n.name.c_str()); return -1; }
    F32ConvArgs a{};
    a.N = N; a.C = C; a.H = H; a.W = W; a.OH = OH; a.OW = OW; a.cout = cout; a.K = K; a.Kpad = Kpad;
    a.SH = SH; a.SW = SW; a.PH = PH; a.PW = PW;
    a.tail_split = 0;
    a.cfg = conv_f32_mfma_pick(a);
    const int BM = conv_f32_mfma_bm(a.cfg), ntile = (cout + BM - 1) / BM, G = 64 / BM, NIg = 32 / G;
    const float* wsrc = (const float*)w.data.data();
    std::vector<float> wf((size_t)ntile * nstage * 32 * BM, 0.f);
    for (int co = 0; co < cout; co++)
        for (int k = 0; k < K; k++) {
            const int r = k & 31;
            wf[(((size_t)(co / BM) * nstage + (k >> 5)) * NIg + r / G) * 64 + (r % G) * BM + co % BM] = wsrc[(size_t)co * K + k];
        }
    std::vector<unsigned> lut(Kpad, 0u);
    for (int k = 0; k < K; k++) {
        const int kx = k % KW, ky = (k / KW) % KH, c = k / (KW * KH);
        lut[k] = (unsigned)(c * H * W + ky * DH * W + kx * DW) | (unsigned)(kx * DW) << 24 | (unsigned)(ky * DH) << 28;
    }
    float* dwf = nullptr; unsigned* dlut = nullptr;
    if (upload(g, wf, &dwf) || upload(g, lut, &dlut)) return -1;
    if (b) {
        if (b->dtype != TAMD_DT_FP32) { set_error("%s: fp32 bias expected", n.name.c_str()); return -1; }
        std::vector<float> hb((const float*)b->data.data(), (const float*)b->data.data() + cout);
        float* d = nullptr;
        if (upload(g, hb, &d)) return -1;
        a.bias_f32 = d;
    }
    if (!g->zero_page) { if (dev_alloc(g, &g->zero_page, 256, true)) return -1; }
    a.x = xdev; a.w = dwf; a.klut = dlut; a.zeros = (const float*)g->zero_page; a.out_f32 = ydev;
    a.out_img = oimg; a.out_c0 = oc0; a.act = act; a.out_scale = 1.f;
    Step st; st.node = n.name; st.kernel = conv_f32_mfma_kernel_name(a);
    st.macs = (double)N * OH * OW * cout * K;
    st.bytes = 4.0 * ((double)N * C * H * W + (double)N * cout * OH * OW + (double)cout * K);
    st.fn = [a](hipStream_t s) { return launch_conv_f32_mfma(a, s); };
    g->steps.push_back(st);
    return 0;
}

// per-launch time of a list of launches run back to back (best of two bursts)
static int time_steps(tamd_graph* g, const std::vector<std::function<hipError_t(hipStream_t)>>& fns, float* ms_out)
{
    hipEvent_t e0, e1;
    *ms_out = 1e30f;
    for (auto& f : fns)
        if (f(g->stream) != hipSuccess) { (void)hipGetLastError(); return 0; }
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    for (int round = 0; round < 2; round++) {
        float t = 0;
        HIPCHK(hipEventRecord(e0, g->stream));
        for (int it = 0; it < 10; it++)
            for (auto& f : fns) (void)f(g->stream);
        HIPCHK(hipEventRecord(e1, g->stream));
        HIPCHK(hipEventSynchronize(e1));
        HIPCHK(hipEventElapsedTime(&t, e0, e1));
        *ms_out = std::min(*ms_out, t / 10);
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    return 0;
}

// Winograd F(2,3) form of the 3x3 / stride 1 convolution whose direct form is the LAST step of g->steps: replaces it when it
// is faster (or when TAMD_F32_WINOGRAD=1).  U = G g G^T with G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1], evaluated in double
// and rounded once.
static int plan_winograd_f32(tamd_graph* g, HNode& n, const HTensor& x, const HTensor& y, int oimg)
{
    const char* env = getenv("TAMD_F32_WINOGRAD");                // read at every prerun (tests switch it)
    const int mode = env ? atoi(env) : -1;
    const tamd_conv_param& p = n.p.conv;
    if (mode == 0 || p.group != 1 || p.kernel_h != 3 || p.kernel_w != 3 || p.stride_h != 1 || p.stride_w != 1 || p.dilation_h != 1
        || p.dilation_w != 1) return 0;
    HTensor& w = g->tensors[n.in[1]];
    HTensor* b = n.in.size() > 2 ? &g->tensors[n.in[2]] : nullptr;
    F32WinoArgs a{};
    a.N = x.n; a.C = x.c; a.H = x.h; a.W = x.w; a.OH = y.h; a.OW = y.w; a.cout = y.c; a.PH = p.pad_h0; a.PW = p.pad_w0;
    a.TH = (y.h + 1) / 2; a.TW = (y.w + 1) / 2; a.T = x.n * a.TH * a.TW; a.Tpad = rup(a.T, 64);
    a.Cpad = rup(x.c, 16); a.Mpad = rup(y.c, 64);
    a.out_img = oimg; a.out_c0 = y.c_off; a.act = p.activation;
    const size_t ws_bytes = 16ull * ((size_t)a.Cpad + a.Mpad) * a.Tpad * 4;
    if (ws_bytes > (1ull << 30)) return 0;                        // transformed tensors of this layer would not be "small"
    // only worth timing when the multiplications matter at all (the plan-time decision is by measurement anyway)
    if (mode != 1 && (double)a.T * y.c * x.c < 2e6) return 0;
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    const float* wsrc = (const float*)w.data.data();
    std::vector<float> U((size_t)16 * a.Mpad * a.Cpad, 0.f);
    for (int co = 0; co < y.c; co++)
        for (int c = 0; c < x.c; c++) {
            const float* gk = wsrc + ((size_t)co * x.c + c) * 9;
            double t[4][3];
            for (int i = 0; i < 4; i++)
                for (int j = 0; j < 3; j++) t[i][j] = G[i][0] * gk[j] + G[i][1] * gk[3 + j] + G[i][2] * gk[6 + j];
            for (int i = 0; i < 4; i++)
                for (int j = 0; j < 4; j++)
                    U[((size_t)(4 * i + j) * a.Mpad + co) * a.Cpad + c] = (float)(t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2]);
        }
    // transformed weights, bias and the V / M workspaces (up to 1 GiB per layer) belong to the graph only if Winograd WINS the
    // timing below: plain allocations until then, freed when the direct convolution is kept (they used to stay in dev_allocs --
    // hundreds of MB of dead HBM per batched fp32 network)
    float* dU = nullptr; float* db = nullptr;
    void* dV = nullptr; void* dM = nullptr;
    std::vector<void*> mine;
    auto drop = [&]() { for (void* q : mine) (void)hipFree(q); mine.clear(); };
    auto grab = [&](void** q, size_t bytes) -> int {
        if (hipMalloc(q, bytes + 1024) != hipSuccess) { (void)hipGetLastError(); set_error("winograd workspace: out of device memory"); drop(); return -1; }
        mine.push_back(*q);
        return 0;
    };
    if (grab((void**)&dU, U.size() * 4)) return -1;
    if (hipMemcpy(dU, U.data(), U.size() * 4, hipMemcpyHostToDevice) != hipSuccess) { drop(); set_error("winograd weight upload failed"); return -1; }
    if (b) {
        if (grab((void**)&db, (size_t)y.c * 4) || hipMemcpy(db, b->data.data(), (size_t)y.c * 4, hipMemcpyHostToDevice) != hipSuccess) { drop(); return -1; }
    }
    if (grab(&dV, 16ull * a.Cpad * a.Tpad * 4) || grab(&dM, 16ull * a.Mpad * a.Tpad * 4)) return -1;
    a.x = (const float*)x.dptr; a.U = dU; a.bias = db; a.V = (float*)dV; a.M = (float*)dM; a.y = (float*)y.dptr;
    Step direct = g->steps.back();
    std::vector<std::function<hipError_t(hipStream_t)>> wino = {[a](hipStream_t s) { return launch_wino_in_f32(a, s); },
                                                                 [a](hipStream_t s) { return launch_wino_gemm_f32(a, s); },
                                                                 [a](hipStream_t s) { return launch_wino_out_f32(a, s); }};
    bool use = mode == 1;
    if (mode != 1) {
        float tw = 0, td = 0;
        if (time_steps(g, wino, &tw) || time_steps(g, {direct.fn}, &td)) { drop(); return -1; }
        use = tw * 3.f < td * 0.97f;                              // time_steps reports per launch: three against one
        if (getenv("TAMD_DEBUG")) fprintf(stderr, "tengine_amd: %s: winograd F(2,3) %.2f us vs direct %.2f us -> %s\n", n.name.c_str(), 3e3f * tw, 1e3f * td, use ? "winograd" : "direct");
    }
    if (!use) { HIPCHK(hipStreamSynchronize(g->stream)); drop(); return 0; }
    for (void* q : mine) g->dev_allocs.push_back(q);           // from here on the graph owns them (freed by tamd_graph_destroy)
    g->steps.pop_back();
    const char* names[3] = {"wino_in_f32", "wino_gemm_f32<F(2,3)>", "wino_out_f32"};
    for (int k = 0; k < 3; k++) {
        Step st; st.node = n.name; st.kernel = names[k];
        if (k == 1) { st.macs = 16.0 * a.T * y.c * x.c; st.bytes = direct.bytes; }     // SURVEY 8(d) bytes of the node, once
        st.fn = wino[k];
        g->steps.push_back(st);
    }
    return 0;
}

int plan_f32(tamd_graph* g)
{
    for (auto& t : g->tensors) {
        if (t.ttype == TAMD_TT_CONST) continue;
        nhwc_geom(t);
        t.nchw_raw = true;
        t.cs = 0; t.c_off = 0;
    }
    std::vector<int> alias_of(g->tensors.size(), -1);
    for (auto& n : g->nodes)
        if (n.op == TAMD_OP_DROPOUT || n.op == TAMD_OP_FLATTEN || n.op == TAMD_OP_RESHAPE) alias_of[n.out[0]] = n.in[0];   // dense NCHW: views
    for (auto& io : g->inputs) {
        HTensor& t = g->tensors[io.tensor];
        io.bytes = t.elems() * 4;
        if (dev_alloc(g, &io.stage, io.bytes, true)) return -1;
        HIPCHK(hipHostMalloc(&io.pinned, io.bytes, hipHostMallocDefault));
        t.dptr = io.stage;
    }
    // concat-by-offset: a conv / relu / upsample whose only consumer is a channel concat writes its
    // channels straight into the concat output (kernel arguments out_img / out_c0); the concat launch disappears
    std::vector<int> view_of(g->tensors.size(), -1), view_off(g->tensors.size(), 0);
    for (auto& n : g->nodes) {
        if (n.op != TAMD_OP_CONCAT) continue;
        HTensor& y = g->tensors[n.out[0]];
        const int ax = n.p.concat.axis < 0 ? n.p.concat.axis + (int)y.dims.size() : n.p.concat.axis;
        int off = 0;
        for (int i : n.in) {
            HTensor& xi = g->tensors[i];
            bool ok = ax == 1 && xi.ttype == TAMD_TT_VAR && count_consumers(g, i) == 1 && alias_of[i] < 0 && view_of[i] < 0;
            if (ok) {
                ok = false;
                for (auto& pn : g->nodes)
                    if (!pn.out.empty() && pn.out[0] == i)
                        ok = pn.op == TAMD_OP_CONV || pn.op == TAMD_OP_RELU || pn.op == TAMD_OP_RELU6 || pn.op == TAMD_OP_UPSAMPLE;
            }
            if (ok) { view_of[i] = n.out[0]; view_off[i] = off; }
            off += xi.c;
        }
    }
    for (size_t i = 0; i < g->tensors.size(); i++) {
        HTensor& t = g->tensors[i];
        if (t.ttype == TAMD_TT_CONST || t.dptr || alias_of[i] >= 0 || view_of[i] >= 0) continue;
        if (dev_alloc(g, &t.dptr, t.elems() * 4, true)) return -1;
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
    // output placement of a tensor: elements per image of the buffer it lives in, first channel
    auto out_img = [](const HTensor& t) { return (t.is_view ? t.cs : t.c) * t.h * t.w; };

    for (auto& n : g->nodes) {
        if (n.op == TAMD_OP_INPUT || n.op == TAMD_OP_CONST || n.op == TAMD_OP_DROPOUT || n.op == TAMD_OP_FLATTEN || n.op == TAMD_OP_RESHAPE) continue;
        HTensor& x = g->tensors[n.in[0]];
        HTensor& y = g->tensors[n.out[0]];
        switch (n.op) {
        case TAMD_OP_CONV: {
            const tamd_conv_param& p = n.p.conv;
            if (p.group == 1) {
                if (plan_gemm_f32(g, n, (const float*)x.dptr, x.n, x.c, x.h, x.w, y.h, y.w, y.c, p.kernel_h, p.kernel_w, p.stride_h,
                                  p.stride_w, p.pad_h0, p.pad_w0, p.dilation_h, p.dilation_w, p.activation, (float*)y.dptr, out_img(y), y.c_off)) return -1;
                if (plan_winograd_f32(g, n, x, y, out_img(y))) return -1;
            } else {
                HTensor& w = g->tensors[n.in[1]];
                HTensor* b = n.in.size() > 2 ? &g->tensors[n.in[2]] : nullptr;
                std::vector<float> hw((const float*)w.data.data(), (const float*)w.data.data() + w.data.size() / 4);
                float* dw = nullptr; float* db = nullptr;
                if (upload(g, hw, &dw)) return -1;
                if (b) {
                    std::vector<float> hb((const float*)b->data.data(), (const float*)b->data.data() + y.c);
                    if (upload(g, hb, &db)) return -1;
                }
                F32DirectArgs a{};
                a.x = (const float*)x.dptr; a.w = dw; a.bias = db; a.y = (float*)y.dptr;
                a.N = x.n; a.C = x.c; a.H = x.h; a.W = x.w; a.OH = y.h; a.OW = y.w; a.cout = y.c;
                a.KH = p.kernel_h; a.KW = p.kernel_w; a.SH = p.stride_h; a.SW = p.stride_w; a.PH = p.pad_h0; a.PW = p.pad_w0;
                a.DH = p.dilation_h; a.DW = p.dilation_w; a.group = p.group;
                a.out_img = out_img(y); a.out_c0 = y.c_off; a.act = p.activation;
                Step st; st.node = n.name; st.kernel = "conv_f32_direct";
                st.macs = (double)y.elems() * (x.c / p.group) * p.kernel_h * p.kernel_w;
                st.bytes = 4.0 * ((double)x.elems() + (double)y.elems());
                st.fn = [a](hipStream_t s) { return launch_conv_f32_direct(a, s); };
                g->steps.push_back(st);
            }
            break;
        }
        case TAMD_OP_FC: {
            const int batch = x.dims[0], hidden = (int)(x.elems() / batch);
            if (plan_gemm_f32(g, n, (const float*)x.dptr, batch, hidden, 1, 1, 1, 1, y.c, 1, 1, 1, 1, 0, 0, 1, 1, -1, (float*)y.dptr, y.c, 0)) return -1;
            break;
        }
        case TAMD_OP_POOL: {
            PoolGeom pg = pool_geom(n.p.pool, x.h, x.w);
            F32PoolArgs a{};
            a.x = (const float*)x.dptr; a.y = (float*)y.dptr;
            a.N = x.n; a.C = x.c; a.H = x.h; a.W = x.w; a.OH = y.h; a.OW = y.w;
            a.KH = pg.kh; a.KW = pg.kw; a.SH = pg.sh; a.SW = pg.sw; a.PH = pg.ph0; a.PW = pg.pw0;
            a.method = n.p.pool.pool_method; a.caffe_flavor = n.p.pool.caffe_flavor;
            Step st; st.node = n.name; st.kernel = "pool_f32"; st.bytes = 4.0 * ((double)x.elems() + (double)y.elems());
            st.fn = [a](hipStream_t s) { return launch_pool_f32(a, s); };
            g->steps.push_back(st);
            break;
        }
        case TAMD_OP_RELU: case TAMD_OP_RELU6: case TAMD_OP_UPSAMPLE: {
            F32MapArgs a{};
            a.x = (const float*)x.dptr; a.y = (float*)y.dptr;
            a.N = x.n; a.C = x.c; a.H = x.h; a.W = x.w;
            a.scale = n.op == TAMD_OP_UPSAMPLE ? (int)n.p.ups.scale : 1;
            a.out_img = out_img(y); a.out_c0 = y.c_off;
            a.slope = n.op == TAMD_OP_RELU ? n.p.relu.negative_slope : 0.f;
            const int mode = n.op == TAMD_OP_RELU ? 0 : (n.op == TAMD_OP_UPSAMPLE ? 2 : 3);
            Step st; st.node = n.name; st.kernel = mode == 2 ? "upsample_f32" : "relu_f32"; st.bytes = 4.0 * ((double)x.elems() + (double)y.elems());
            st.fn = [a, mode](hipStream_t s) { return launch_map_f32(a, mode, s); };
            g->steps.push_back(st);
            break;
        }
        case TAMD_OP_PRIORBOX: {          // shapes-only node: evaluated here, once (graph_infer.hip priorbox_eval); no launch at run
            const HTensor& img = g->tensors[n.in[1]];
            std::vector<float> boxes;
            priorbox_eval(n.p.priorbox, x.dims[2], x.dims[3], img.dims[2], img.dims[3], &boxes);
            if (boxes.size() != y.elems()) { set_error("priorbox %s: output shape mismatch", n.name.c_str()); return -1; }
            HIPCHK(hipMemcpyAsync(y.dptr, boxes.data(), boxes.size() * 4, hipMemcpyHostToDevice, g->stream));
            HIPCHK(hipStreamSynchronize(g->stream));
            y.prerun_const = true;
            break;
        }
        case TAMD_OP_CONCAT: {
            int ax = n.p.concat.axis < 0 ? n.p.concat.axis + (int)y.dims.size() : n.p.concat.axis;
            if (ax < 0 || ax >= (int)y.dims.size()) { set_error("concat %s: bad axis", n.name.c_str()); return -1; }
            // dense tensors: any axis is a channel concat of the [outer][dims[ax]][1][inner] view
            int outer = 1, inner = 1;
            for (int d = 0; d < ax; d++) outer *= y.dims[d];
            for (size_t d = ax + 1; d < y.dims.size(); d++) inner *= y.dims[d];
            bool all_const = true;
            for (int i : n.in) all_const &= g->tensors[i].prerun_const;
            int off = 0;
            for (int i : n.in) {
                HTensor& xi = g->tensors[i];
                if (xi.is_view && xi.dptr == y.dptr) { off += xi.dims[ax]; continue; }      // written in place by its producer
                F32MapArgs a{};
                a.x = (const float*)xi.dptr; a.y = (float*)y.dptr;
                a.N = outer; a.C = xi.dims[ax]; a.H = 1; a.W = inner; a.scale = 1;
                a.out_img = y.dims[ax] * inner; a.out_c0 = off;
                Step st; st.node = n.name; st.kernel = "concat_f32"; st.bytes = 8.0 * xi.elems();
                st.once = all_const;
                st.fn = [a](hipStream_t s) { return launch_map_f32(a, 1, s); };
                g->steps.push_back(st);
                off += xi.dims[ax];
            }
            y.prerun_const = all_const;
            break;
        }
        case TAMD_OP_ELTWISE: {
            HTensor& xb = g->tensors[n.in[1]];
            const int type = n.p.elt.type;
            if (x.dims != xb.dims || (type != 0 && type != 2 && type != 4 && type != 6)) { set_error("eltwise %s: broadcast / type %d unsupported", n.name.c_str(), type); return -1; }
            const float* pa = (const float*)x.dptr; const float* pb = (const float*)xb.dptr; float* py = (float*)y.dptr;
            const size_t cnt = x.elems();
            Step st; st.node = n.name; st.kernel = "eltwise_f32"; st.bytes = 12.0 * cnt;
            st.fn = [pa, pb, py, cnt, type](hipStream_t s) { return launch_eltwise_f32(pa, pb, py, cnt, type, s); };
            g->steps.push_back(st);
            break;
        }
        case TAMD_OP_SOFTMAX: {
            const float* px = (const float*)x.dptr; float* py = (float*)y.dptr;
            int ax = n.p.softmax.axis < 0 ? n.p.softmax.axis + (int)x.dims.size() : n.p.softmax.axis;
            if (ax < 0 || ax >= (int)x.dims.size()) { set_error("softmax %s: bad axis", n.name.c_str()); return -1; }
            int N = 1, inner = 1;
            const int C = x.dims[ax];
            for (int i = 0; i < ax; i++) N *= x.dims[i];
            for (size_t i = ax + 1; i < x.dims.size(); i++) inner *= x.dims[i];
            Step st; st.node = n.name; st.kernel = "softmax_f32"; st.bytes = 8.0 * x.elems();
            st.fn = [px, py, N, C, inner](hipStream_t s) { return launch_softmax_f32(px, py, N, C, inner, s); };
            g->steps.push_back(st);
            break;
        }
        default:
            set_error("op %d (%s) is not supported on the device for fp32", n.op, n.name.c_str());
            return -1;
        }
    }
    for (auto& io : g->outputs) {
        HTensor& t = g->tensors[io.tensor];
        io.bytes = t.elems() * 4;
        HIPCHK(hipHostMalloc(&io.pinned, io.bytes, hipHostMallocDefault));
        io.stage = t.dptr;
    }
    return 0;
}

}  // namespace tamd
