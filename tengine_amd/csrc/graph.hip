// Graph IR, shape inference, planner and executor behind the C ABI of include/tengine_amd.h.
//
// Mirrors what a Tengine device backend does between pre_run and run (CUDA pattern:
// source/device/cuda/cuda_executor.cc:136-204), re-designed for MI355X:
//   prerun : infer shapes (restating source/operator/prototype/*.c), choose NHWC device layouts,
//            repack weights once (the role of conv_hcl_prerun, conv_kernel_x86.c:2137-2209), compile
//            the node list into a launch list and capture it into ONE hipGraph;
//   run    : H2D inputs -> hipGraphLaunch -> D2H outputs on a private stream, pinned bounce buffers.
#include "graph.h"
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

namespace tamd {

// prerun (plan + hipGraph capture) and the device-synchronous frees are serialised process-wide: HIP rejects legacy-
// stream / synchronous operations of one host thread while another one captures (seen as "operation would make the
// legacy stream depend on a capturing blocking stream" under tools/exp/stress_threads.py).  run/launch are not affected.
static std::mutex g_capture_mutex;

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    if (getenv("TAMD_VERBOSE")) fprintf(stderr, "tengine_amd: %s\n", g_err);
}

static inline int rup(int v, int m) { return (v + m - 1) / m * m; }
static inline int esize(int dt) { return (dt == TAMD_DT_FP32 || dt == TAMD_DT_INT32) ? 4 : (dt == TAMD_DT_FP16 ? 2 : 1); }
static int cdiv_c(int a, int b) { return a / b; }  // C semantics (truncation), as the reference

// ---------------------------------------------------------------------------------------------
// shape inference (restating the reference's infer_shape functions)
// ---------------------------------------------------------------------------------------------
// pooling_param.h:59-105
static int pool_out_size(int input, int kernel, int stride, int pad, int caffe)
{
    int output = 1;
    if (pad >= 0) {
        if (caffe == 1) {
            output = 2 + cdiv_c(input - kernel + 2 * pad - 1, stride);
            if (pad > 0 && ((output - 1) * stride >= input + pad)) output--;
        } else if (caffe == 2)
            output = 1 + cdiv_c(input - kernel + pad, stride);
        else
            output = 1 + cdiv_c(input - kernel + 2 * pad, stride);
    } else
        output = 1 + cdiv_c(input - 1, stride);
    return output;
}
static void pool_real_pads(int out, int in, int kernel, int stride, int pad_org, int* pad0, int* pad1)
{
    int total = (out - 1) * stride + kernel;
    int pad_num = total - in;
    if (pad_num < 0) pad_num = 0;
    if (pad_org < 0) { *pad0 = pad_num / 2; *pad1 = pad_num - pad_org; }
    else { *pad0 = pad_org; *pad1 = pad_num - pad_org; }
}

// pooling.c:36-100
PoolGeom pool_geom(const tamd_pool_param& p, int h, int w)
{
    PoolGeom g{};
    int glob = p.global;
    if (p.kernel_h == h && p.kernel_w == w && p.pad_w0 == 0 && p.pad_w1 == 0 && p.pad_h0 == 0 && p.pad_h1 == 0) glob = 1;
    if (glob) { g = {1, 1, h, w, 1, 1, 0, 0}; return g; }
    int caffe = p.caffe_flavor & ~0x100;
    g.oh = pool_out_size(h, p.kernel_h, p.stride_h, p.pad_h0, p.caffe_flavor);
    g.ow = pool_out_size(w, p.kernel_w, p.stride_w, p.pad_w0, p.caffe_flavor);
    g.kh = p.kernel_h; g.kw = p.kernel_w; g.sh = p.stride_h; g.sw = p.stride_w;
    int d;
    if (caffe != 2) {
        pool_real_pads(g.oh, h, p.kernel_h, p.stride_h, p.pad_h0, &g.ph0, &d);
        pool_real_pads(g.ow, w, p.kernel_w, p.stride_w, p.pad_w0, &g.pw0, &d);
    } else { g.ph0 = p.pad_h0 / 2; g.pw0 = p.pad_w0 / 2; }
    return g;
}

// ---------------------------------------------------------------------------------------------
// PriorBox: the SSD anchor boxes depend on tensor SHAPES and node parameters only, so the node is evaluated
// once at prerun and its output tensor is a device constant; the reference recomputes the same numbers at every run
// (priorbox_ref.c:53-175).  Arithmetic types follow that file: sizes are truncated to int (:110,:121); the (min, max)
// prior is the double sqrt of the int product; ratio priors are int * | / double sqrt(ratio), stored as float; corners are
// (centre -+ size * 0.5f) / extent in float; a flipped prior swaps the sizes AND the extents it divides by (:146-150).
// ---------------------------------------------------------------------------------------------
int priorbox_count(const tamd_priorbox_param& p)
{
    return (1 + (p.max_size_num > 0 ? 1 : 0) + p.aspect_ratio_num * (p.flip ? 2 : 1)) * p.min_size_num;       // priorbox.c:37-64
}

void priorbox_eval(const tamd_priorbox_param& p, int feat_h, int feat_w, int data_h, int data_w, std::vector<float>* out)
{
    struct Prior { float w, h; bool flipped; };
    std::vector<Prior> cell;                                     // the priors of one feature-map cell, in output order
    for (int s = 0; s < p.min_size_num; s++) {
        const int mn = (int)p.min_size[s];
        cell.push_back({(float)mn, (float)mn, false});
        if (p.max_size_num > 0) {
            const float q = (float)std::sqrt((double)(mn * (int)p.max_size[s]));
            cell.push_back({q, q, false});
        }
        for (int r = 0; r < p.aspect_ratio_num; r++) {
            const double root = std::sqrt((double)p.aspect_ratio[r]);
            const Prior pr{(float)(mn * root), (float)(mn / root), false};
            cell.push_back(pr);
            if (p.flip) cell.push_back({pr.w, pr.h, true});
        }
    }
    const bool own_image = p.image_h != 0 && p.image_w != 0, own_step = p.step_h != 0 && p.step_w != 0;
    const float iw = (float)(own_image ? p.image_w : data_w), ih = (float)(own_image ? p.image_h : data_h);
    const float step_w = own_step ? p.step_w : iw / (float)feat_w, step_h = own_step ? p.step_h : ih / (float)feat_h;
    const size_t dim = (size_t)feat_h * feat_w * cell.size() * 4;
    out->assign(2 * dim, 0.f);
    float* o = out->data();
    for (int y = 0; y < feat_h; y++)
        for (int x = 0; x < feat_w; x++) {
            const float cx = ((float)x + p.offset) * step_w, cy = ((float)y + p.offset) * step_h;
            for (const Prior& pr : cell) {
                const float hx = (pr.flipped ? pr.h : pr.w) * 0.5f, hy = (pr.flipped ? pr.w : pr.h) * 0.5f;
                const float ex = pr.flipped ? ih : iw, ey = pr.flipped ? iw : ih;
                o[0] = (cx - hx) / ex; o[1] = (cy - hy) / ey; o[2] = (cx + hx) / ex; o[3] = (cy + hy) / ey;
                o += 4;
            }
        }
    if (p.clip)
        for (size_t i = 0; i < dim; i++) (*out)[i] = std::min(std::max((*out)[i], 0.f), 1.f);
    for (size_t i = 0; i < dim; i++) (*out)[dim + i] = p.variance[i & 3];
}

// its quantisation (priorbox_ref.c:178-213): uint8 truncates (int)(f / scale + zp); int8 rounds
void priorbox_quant_u8(const std::vector<float>& f, float scale, int zp, std::vector<uint8_t>* q)
{
    q->resize(f.size());
    for (size_t i = 0; i < f.size(); i++) {
        const int u = (int)(f[i] / scale + (float)zp);
        (*q)[i] = (uint8_t)std::min(std::max(u, 0), 255);
    }
}

static int infer_shapes(tamd_graph* g)
{
    for (auto& n : g->nodes) {
        if (n.op == TAMD_OP_INPUT || n.op == TAMD_OP_CONST) continue;
        if (n.in.empty() || n.out.empty()) { set_error("node %s has no io", n.name.c_str()); return -1; }
        HTensor& x = g->tensors[n.in[0]];
        HTensor& y = g->tensors[n.out[0]];
        switch (n.op) {
        case TAMD_OP_CONV: {   // convolution.c:35-145
            if (n.in.size() < 2) { set_error("conv %s: no weight tensor", n.name.c_str()); return -1; }
            if (x.dims.size() != 4) { set_error("conv %s: input is not 4-D", n.name.c_str()); return -1; }
            if (g->tensors[n.in[1]].dims.size() != 4) { set_error("conv %s: weight is not 4-D", n.name.c_str()); return -1; }
            tamd_conv_param& p = n.p.conv;
            if (p.kernel_w == 0) { p.kernel_w = 1; p.pad_w0 = p.pad_w1 = 0; }
            if (p.kernel_h == 0) p.kernel_h = 1;
            if (p.stride_w == 0) p.stride_w = 1;
            if (p.stride_h == 0) p.stride_h = 1;
            if (p.dilation_h == 0) p.dilation_h = 1;
            if (p.dilation_w == 0) p.dilation_w = 1;
            p.input_channel = x.dims[1];
            const HTensor& w = g->tensors[n.in[1]];
            int h = x.dims[2], wd = x.dims[3], oh, ow;
            if (p.pad_h0 < 0) {
                oh = (h - 1) / p.stride_h + 1;
                int pad_num = (oh - 1) * p.stride_h + p.kernel_h - h;
                if (p.pad_h0 == -1) { p.pad_h0 = pad_num / 2; p.pad_h1 = pad_num - pad_num / 2; }
                else { p.pad_h1 = pad_num / 2; p.pad_h0 = pad_num - pad_num / 2; }
            } else
                oh = (h - p.dilation_h * (p.kernel_h - 1) - 1 + p.pad_h0 + p.pad_h1) / p.stride_h + 1;
            if (p.pad_w0 < 0) {
                ow = (wd - 1) / p.stride_w + 1;
                int pad_num = (ow - 1) * p.stride_w + p.kernel_w - wd;
                if (p.pad_w0 == -1) { p.pad_w0 = pad_num / 2; p.pad_w1 = pad_num - pad_num / 2; }
                else { p.pad_w1 = pad_num / 2; p.pad_w0 = pad_num - pad_num / 2; }
            } else
                ow = (wd - p.dilation_w * (p.kernel_w - 1) - 1 + p.pad_w0 + p.pad_w1) / p.stride_w + 1;
            y.dims = {x.dims[0], w.dims[0], oh ? oh : 1, ow ? ow : 1};
            break;
        }
        case TAMD_OP_FC: {
            if (n.in.size() < 2 || g->tensors[n.in[1]].dims.empty() || x.dims.empty()) { set_error("fc %s: no weight tensor", n.name.c_str()); return -1; }
            int nout = n.p.fc.num_output ? n.p.fc.num_output : g->tensors[n.in[1]].dims[0];
            y.dims = {x.dims[0], nout};
            break;
        }
        case TAMD_OP_POOL: {
            PoolGeom pg = pool_geom(n.p.pool, x.dims[2], x.dims[3]);
            y.dims = {x.dims[0], x.dims[1], pg.oh, pg.ow};
            break;
        }
        case TAMD_OP_RELU: case TAMD_OP_RELU6: case TAMD_OP_ELTWISE: case TAMD_OP_DROPOUT: case TAMD_OP_SOFTMAX:
            y.dims = x.dims;
            break;
        case TAMD_OP_CONCAT: {
            int ax = n.p.concat.axis < 0 ? n.p.concat.axis + (int)x.dims.size() : n.p.concat.axis;
            y.dims = x.dims;
            int s = 0;
            for (int i : n.in) s += g->tensors[i].dims[ax];
            y.dims[ax] = s;
            break;
        }
        case TAMD_OP_UPSAMPLE: {
            int sc = (int)n.p.ups.scale;
            y.dims = {x.dims[0], x.dims[1], x.dims[2] * sc, x.dims[3] * sc};
            break;
        }
        case TAMD_OP_PERMUTE: {           // permute.c infer_shape: out.dims[i] = in.dims[order[i]]
            if (x.dims.size() != 4) { set_error("permute %s: only 4-D tensors", n.name.c_str()); return -1; }
            y.dims.resize(4);
            for (int i = 0; i < 4; i++) {
                const int o = n.p.perm.order[i];
                if (o < 0 || o > 3) { set_error("permute %s: bad order", n.name.c_str()); return -1; }
                y.dims[i] = x.dims[o];
            }
            break;
        }
        case TAMD_OP_RESHAPE: {           // the resolved shape travels in the parameter; only the batch may have been re-set
            const tamd_reshape_param& rp = n.p.reshape;
            if (rp.dim_num < 1 || rp.dim_num > 8) { set_error("reshape %s: bad shape", n.name.c_str()); return -1; }
            y.dims.assign(rp.dims, rp.dims + rp.dim_num);
            size_t rest = 1;
            for (int i = 1; i < rp.dim_num; i++) rest *= (size_t)std::max(1, rp.dims[i]);
            if (y.elems() != x.elems() && rest && x.elems() % rest == 0) y.dims[0] = (int)(x.elems() / rest);
            if (y.elems() != x.elems()) { set_error("reshape %s: element count changes", n.name.c_str()); return -1; }
            break;
        }
        case TAMD_OP_FLATTEN: {
            int f = 1;
            for (size_t i = 1; i < x.dims.size(); i++) f *= x.dims[i];
            y.dims = {x.dims[0], f};
            break;
        }
        case TAMD_OP_PRIORBOX: {          // priorbox.c:33-75: [n][2][feat_h * feat_w * num_priors * 4][1]
            const tamd_priorbox_param& pb = n.p.priorbox;
            if (x.dims.size() != 4 || n.in.size() < 2 || g->tensors[n.in[1]].dims.size() != 4) { set_error("priorbox %s: needs a 4-D feature map and the 4-D image tensor", n.name.c_str()); return -1; }
            if (pb.min_size_num < 1 || pb.min_size_num > TAMD_PRIORBOX_MAX || pb.aspect_ratio_num < 0 || pb.aspect_ratio_num > TAMD_PRIORBOX_MAX
                || (pb.max_size_num != 0 && pb.max_size_num != pb.min_size_num)) { set_error("priorbox %s: bad size / ratio counts", n.name.c_str()); return -1; }
            y.dims = {x.dims[0], 2, x.dims[2] * x.dims[3] * priorbox_count(pb) * 4, 1};
            break;
        }
        default:
            set_error("infer_shape: unsupported op %d (%s)", n.op, n.name.c_str());
            return -1;
        }
    }
    return 0;
}


// ---------------------------------------------------------------------------------------------
// validation: the model bytes may come off the wire (RCCL broadcast, tm2_reader.cc checks the container); the planners
// index constant payloads by operator parameters, so parameters and payload sizes are reconciled ONCE here -- a
// malformed graph fails prerun with a message instead of over-reading the host heap
// ---------------------------------------------------------------------------------------------
static int validate_graph(tamd_graph* g)
{
    auto bad = [&](const HNode& n, const char* what) { set_error("%s: %s", n.name.c_str(), what); return -1; };
    for (auto& n : g->nodes) {
        // priorbox_ref.c fills image 0 of its output only (:99-175); what a batch > 1 tensor holds behind it is undefined there
        if (n.op == TAMD_OP_PRIORBOX && g->tensors[n.out[0]].dims[0] != 1) return bad(n, "PriorBox is defined for batch 1 only");
        if (n.op != TAMD_OP_CONV && n.op != TAMD_OP_FC) continue;
        const HTensor& x = g->tensors[n.in[0]];
        HTensor& w = g->tensors[n.in[1]];
        const HTensor& y = g->tensors[n.out[0]];
        if (w.ttype != TAMD_TT_CONST) return bad(n, "weights must be a constant tensor");
        const size_t es = (size_t)esize(w.dtype);
        size_t want = 0;
        int cout = 0;
        if (n.op == TAMD_OP_CONV) {
            const tamd_conv_param& p = n.p.conv;
            cout = y.dims[1];
            if (p.group < 1 || p.kernel_h < 1 || p.kernel_w < 1 || p.stride_h < 1 || p.stride_w < 1 || p.dilation_h < 1 || p.dilation_w < 1)
                return bad(n, "kernel / stride / dilation / group must be positive");
            if (x.dims[1] % p.group || cout % p.group) return bad(n, "group does not divide the channel counts");
            if (y.dims[2] < 1 || y.dims[3] < 1) return bad(n, "empty output map");
            want = (size_t)cout * (x.dims[1] / p.group) * p.kernel_h * p.kernel_w;
            if (w.dims[0] != cout) return bad(n, "weight dims[0] != output channels");
        } else {
            cout = y.dims[1];
            const size_t hidden = x.elems() / (size_t)std::max(1, x.dims[0]);
            want = (size_t)cout * hidden;
            // fc_ref.c:351-356 switches to a transposed read (need_trans) when weight dims[0] != num_output, but the operator's own
            // infer_shape (operator/prototype/fc.c:43-97) sizes the output from weight dims[0] and insists on dims[1] == hidden,
            // so a consistent model never gets there; such a node is refused here (the plugin leaves it to the CPU device)
            if (w.dims.size() != 2 || w.dims[0] != cout || (size_t)w.dims[1] != hidden) return bad(n, "fc weight must be [num_output][hidden]");
        }
        if (w.elems() != want || w.data.size() != want * es) return bad(n, "weight size does not match the operator parameters");
        if (!w.scales.empty() && w.scales.size() != 1 && w.scales.size() != (size_t)cout) return bad(n, "weight scale count is neither 1 nor the output channel count");
        if (n.in.size() > 2) {
            const HTensor& b = g->tensors[n.in[2]];
            if (b.ttype != TAMD_TT_CONST || b.elems() < (size_t)cout || b.data.size() < (size_t)cout * esize(b.dtype) || esize(b.dtype) != 4)
                return bad(n, "bias must be a constant of at least one 32-bit value per output channel");
        }
    }
    return 0;
}

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

enum { RQ_CONV_HCL = 0, RQ_CONV_REF = 1, RQ_FC = 2 };   // A1 / A2 / A5 of SURVEY Appendix A (epilogue.h)

// which formula the reference's score() selection lands on (conv_hcl_x86.c:351-371,
// conv_dw_hcl_x86.c:508-543, conv_ref.c:197-200)
static int conv_mode(const tamd_conv_param& p, int batch, int cin, int cout)
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
struct RqFold { float m1, lo, hi, out_scale; std::vector<float> m2; };
static RqFold fold_requant(int mode, int act, float in_s, float out_s, const HTensor& w, int cout)
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
static bool exp_plain_kernels() { const char* e = exp_env("TAMD_EXP_PLAIN_KERNELS"); return e && atoi(e) == 1; }

// uploads both per-channel vectors; *wscale = the fast-path multipliers, rq->m2 = the chain's factors
static int upload_rq(tamd_graph* g, const RqFold& r, int cpad, const float** wscale, RqArgs* rq)
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
static int time_fn(tamd_graph* g, const std::function<hipError_t(hipStream_t)>& fn, float* ms_out)
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

static bool autotune_enabled()
{
    static const char* at_env = getenv("TAMD_AUTOTUNE");
    return !(at_env && atoi(at_env) == 0);
}

// ---- plan cache (TAMD_PLAN_CACHE=<file>): what the plan-time autotune decided, "<site>|<node>|<shape>" -> choice ---------------
// A first prerun measures as usual and writes the file; later preruns of the same model take the recorded choices WITHOUT
// launching anything -- a profiler then sees the run's own launches only (round 2's rocprofv3 CSVs were 99 % autotune
// dispatches), the plan no longer depends on one box's timing noise, and prerun drops from seconds to the packing time.
// The table is process-wide (graphs of one process share it) and guarded by a mutex; a file that changed on disk since it was
// read (size or modification time) is read again at the next lookup.
struct PlanCache {
    bool loaded = false, dirty = false;
    std::string path;
    long long stamp = 0;                                  // size ^ mtime of the file as read / written
    std::map<std::string, std::string> kv;     // the file's entries + this process's
    std::map<std::string, std::string> mine;   // what THIS process decided since the file was read (merged over the file at flush)
};
static std::mutex g_plan_cache_mu;
static long long file_stamp(const std::string& path)
{
    struct stat st;
    if (path.empty() || stat(path.c_str(), &st) != 0) return 0;
    return (long long)st.st_size * 1000003ll ^ (long long)st.st_mtim.tv_sec * 1000000007ll ^ (long long)st.st_mtim.tv_nsec;
}
// first line of a plan file: what the choices were made FOR.  A file written by another library version, for another
// architecture or with another candidate list is ignored as a whole (and overwritten at the next flush): a stale choice
// could name a configuration this build no longer launches
static std::string plan_cache_header()
{
    return std::string("#tamd-plan v2 gfx950 ") + tamd_version() + " gemm" + std::to_string(conv_igemm_num_cfgs()) + "/" + std::to_string(conv_pgemm_num_variants())
           + " u8" + std::to_string(conv_u8_gemm_num_cfgs()) + "/" + std::to_string(conv_u8_patch_num_cfgs());
}
static void plan_cache_read(const std::string& path, std::map<std::string, std::string>* kv)
{
    FILE* f = path.empty() ? nullptr : fopen(path.c_str(), "r");
    if (!f) return;
    char line[512];
    bool first = true, ok = false;
    while (fgets(line, sizeof(line), f)) {
        std::string l = line;
        while (!l.empty() && (l.back() == '\n' || l.back() == '\r')) l.pop_back();
        if (first) { first = false; ok = l == plan_cache_header(); if (!ok) break; continue; }
        const size_t tab = l.find('\t');
        if (tab == std::string::npos) continue;
        (*kv)[l.substr(0, tab)] = l.substr(tab + 1);
    }
    fclose(f);
    if (!ok) kv->clear();
}
static PlanCache& plan_cache_locked()                     // call with g_plan_cache_mu held
{
    static PlanCache pc;
    const char* p = getenv("TAMD_PLAN_CACHE");
    const std::string want = p ? p : "";
    if (!pc.loaded || pc.path != want || (!pc.dirty && file_stamp(want) != pc.stamp)) {
        pc = PlanCache();
        pc.loaded = true; pc.path = want; pc.stamp = file_stamp(want);
        plan_cache_read(want, &pc.kv);
    }
    return pc;
}
bool plan_cache_get(const std::string& key, std::string* v)
{
    std::lock_guard<std::mutex> lk(g_plan_cache_mu);
    PlanCache& pc = plan_cache_locked();
    auto it = pc.kv.find(key);
    if (pc.path.empty() || it == pc.kv.end()) return false;
    *v = it->second;
    return true;
}
void plan_cache_put(const std::string& key, const std::string& v)
{
    std::lock_guard<std::mutex> lk(g_plan_cache_mu);
    PlanCache& pc = plan_cache_locked();
    if (pc.path.empty()) return;
    pc.kv[key] = v;
    pc.mine[key] = v;
    pc.dirty = true;
}
// Several processes may share one file (the ranks of a multi-GPU job): the entries on disk are merged with this process's own
// decisions (ours win), written to a temporary file and renamed over the old one -- a reader sees the old file or the new one,
// never half of either.
static void plan_cache_flush()
{
    std::lock_guard<std::mutex> lk(g_plan_cache_mu);
    PlanCache& pc = plan_cache_locked();
    if (pc.path.empty() || !pc.dirty) return;
    std::map<std::string, std::string> merged;
    plan_cache_read(pc.path, &merged);
    for (auto& e : pc.mine) merged[e.first] = e.second;
    const std::string tmp = pc.path + ".tmp." + std::to_string((long)getpid());
    if (FILE* f = fopen(tmp.c_str(), "w")) {
        fprintf(f, "%s\n", plan_cache_header().c_str());
        for (auto& e : merged) fprintf(f, "%s\t%s\n", e.first.c_str(), e.second.c_str());
        fclose(f);
        if (rename(tmp.c_str(), pc.path.c_str()) != 0) (void)remove(tmp.c_str());
    }
    pc.kv = merged;
    pc.dirty = false;
    pc.stamp = file_stamp(pc.path);
}

// pointwise weight panel in MFMA fragment order: [16-channel slice][64-deep K step][lane = (k block of 16) * 16 + channel][16 B];
// `wd` = [C][K] int8 rows (1x1 conv: K = cin; first conv: K = cin*KH*KW in OIHW order), zero padded to nsteps * 64
static std::vector<int8_t> pack_pw_panel(const int8_t* wd, int C, int K, int nsteps)
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
static thread_local DwArgs g_last_dw;
static thread_local bool g_last_dw_valid = false;
static thread_local ConvArgs g_last_gemm;
static thread_local bool g_last_gemm_valid = false;

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
        mode = conv_mode(p, x.n, x.c, y.c);
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

// ---- pointwise conv + its single consumer (depthwise 3x3 | global pooling) in one launch: pwdw.hip ---------------------
// Which node, if any, can ride in pointwise conv `ni`'s launch.  *tmode: 1 depthwise 3x3, 0 global pooling.
static int find_pwdw_tail(tamd_graph* g, size_t ni, int* tmode, int* prod)
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
static int plan_pwdw(tamd_graph* g, HNode& pw, HNode& tl, int tmode, int prod, size_t s0)
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
        const RqFold rq = fold_requant(conv_mode(q, mid.n, C, C), q.activation, mid.scales[0], y.scales[0], dwt, C);
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
    // fused by construction (no race): a link of a LATENCY chain only -- batch 1, or a pair whose launches cannot fill the machine
    // (<= 32 k pixels AND at most 256 pixel rows of 64, i.e. fewer blocks than CUs).  Batch 8 at 56x56 or batch 16 at 19x19 are
    // throughput launches: they keep the race against the two-launch plan below
    bool fuse = fmode == 2 || ((double)a.N * a.H * a.W <= 32768.0 && (a.N == 1 || (double)a.N * a.H * a.W <= 256.0 * 64.0));
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
static int plan_dwpw(tamd_graph* g, HNode& dw, HNode& pw, size_t s0)
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

static int plan(tamd_graph* g)
{
    // ---- 1. geometry + device buffers for every non-const tensor -------------------------------
    for (auto& t : g->tensors) if (t.ttype != TAMD_TT_CONST) nhwc_geom(t);
    // concat outputs own a buffer; their inputs become views when layouts allow (concat-by-offset:
    // concat/concat_kernel_ref_int8.c with in_scale == out_scale is a pure copy)
    // An input that cannot be written in place (its scale differs -> the reference rescales, concat_kernel_ref_int8.c:70-80;
    // channel count / offset not a multiple of 16; produced or also consumed by a kernel that does not address channel
    // slices; a graph input) keeps its own buffer and is copied by concat_copy_i8 at the concat's position.
    std::vector<int> view_of(g->tensors.size(), -1), view_off(g->tensors.size(), 0);
    auto producer_op = [&](int t) { for (auto& n : g->nodes) if (!n.out.empty() && n.out[0] == t) return n.op; return -1; };
    auto slice_capable = [](int op) { return op == TAMD_OP_CONV || op == TAMD_OP_FC || op == TAMD_OP_POOL; };
    for (auto& n : g->nodes) {
        if (n.op != TAMD_OP_CONCAT) continue;
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
    // identity ops alias their input
    std::vector<int> alias_of(g->tensors.size(), -1);
    for (auto& n : g->nodes) {
        if (n.op == TAMD_OP_DROPOUT || n.op == TAMD_OP_FLATTEN) {
            HTensor& x = g->tensors[n.in[0]];
            // Flatten of an H x W map: the [N, C*H*W] result is the SAME NCHW element order; on the device it stays the NHWC
            // buffer and keeps the 4-D geometry, so a following FC (== conv whose kernel covers the map) and the NCHW
            // output conversion both see (c, h, w)
            alias_of[n.out[0]] = n.in[0];
            HTensor& yy = g->tensors[n.out[0]];
            yy.n = x.n; yy.c = x.c; yy.h = x.h; yy.w = x.w;
        }
    }
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
    static const char* fuse_env = getenv("TAMD_FUSE_ELTWISE");
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
        case TAMD_OP_CONCAT: {
            HTensor& y = g->tensors[n.out[0]];
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
// the pinned buffers of that I/O slot as the first / last launches
static int run_steps(tamd_graph* g, hipStream_t s, int io_slot = -1)
{
    if (io_slot >= 0)
        for (auto& io : g->inputs) {
            hipError_t e = launch_copy_bytes(io.stage, io_slot ? io.pinned2 : io.pinned, io.bytes, s);
            if (e != hipSuccess) { set_error("input upload launch failed: %s", hipGetErrorString(e)); return -1; }
        }
    // while the launch list is being recorded for the direct path: a step that provably touches nothing its predecessors since
    // the last ORDERED launch touch (Step::deps, rd, wr -- the twelve SSD head convolutions, the concat copies behind them) is
    // marked to run beside them; everything else keeps the barrier bit.  OFF unless TAMD_DIRECT_OVERLAP=1: measured on
    // MobileNet-SSD b16 (21 of 59 packets lose the bit) it buys 0.5-4 % -- the packet processor does not spread such short
    // dispatches the way a second queue would -- and an unordered launch is one more thing that has to be right
    const char* ov_env = exp_env("TAMD_DIRECT_OVERLAP");
    const bool overlap = g_launch_rec && ov_env && atoi(ov_env) == 1;
    // ... and the same idea across passes queued back to back (TAMD_DIRECT_WRAP=1; off by default for the same reason: measured
    // 51.4 vs 51.5 us per MobileNet pass -- this packet processor does not start a barrier-free dispatch early).  When the first launch of
    // a pass touches nothing the LAST launch of the previous pass touches (MobileNet: conv1+dw reads the input and writes its own
    // tensor, fc7 reads pool6 and writes the logits; everything in between completed before fc7 started), it needs no barrier
    // bit: pass k+1 starts while pass k's last kernel drains.  Only the list without upload / download launches is marked.
    const char* wr_env = exp_env("TAMD_DIRECT_WRAP");
    const bool wrap = g_launch_rec && io_slot < 0 && wr_env && atoi(wr_env) == 1;
    const size_t rec0 = g_launch_rec ? g_launch_rec->size() : 0;
    const Step *first_step = nullptr, *last_step = nullptr;
    size_t last_step_recs = 0;
    std::vector<const Step*> open;                     // the steps since (and including) the last ordered one
    for (auto* v : {&g->in_steps, &g->steps, &g->out_steps})
        for (auto& st : *v) {
            if (overlap) {
                bool beside = st.deps && !open.empty();
                for (size_t i = 0; i < open.size() && beside; i++) beside = open[i]->deps && !step_conflict(st, *open[i]);
                if (beside) launch_rec_beside();
                else open.clear();
                open.push_back(&st);
            }
            const size_t before = g_launch_rec ? g_launch_rec->size() : 0;
            hipError_t e = st.fn(s);
            launch_rec_clear_flags();          // a step that launched nothing must not leave its flags to the next step's launch
            if (e != hipSuccess) { set_error("launch %s (%s) failed: %s", st.kernel.c_str(), st.node.c_str(), hipGetErrorString(e)); return -1; }
            if (!first_step) first_step = &st;
            last_step = &st;
            last_step_recs = g_launch_rec ? g_launch_rec->size() - before : 0;
        }
    // the last step must be ONE ordered launch (its barrier bit says: everything before it is complete) for the argument to hold
    if (wrap && first_step && last_step && first_step != last_step && first_step->deps && last_step->deps && last_step_recs == 1
        && !(overlap && open.size() > 1) && !step_conflict(*first_step, *last_step) && g_launch_rec->size() > rec0)
        (*g_launch_rec)[rec0].wrap = true;
    if (io_slot >= 0)
        for (auto& io : g->outputs) {
            hipError_t e = launch_copy_bytes(io_slot ? io.pinned2 : io.pinned, io.stage, io.bytes, s);
            if (e != hipSuccess) { set_error("output download launch failed: %s", hipGetErrorString(e)); return -1; }
        }
    return 0;
}

// one launch of the host-to-host list of I/O slot `slot` on the graph's stream
static int launch_io(tamd_graph* g, int slot)
{
    if (g->hexec_io[slot][0]) {
        hipGraphExec_t e = g->hexec_io[slot][g->next_io[slot]];
        g->next_io[slot] ^= 1;
        HIPCHK(hipGraphLaunch(e, g->stream));
        return 0;
    }
    return run_steps(g, g->stream, slot);
}

}  // namespace tamd

using namespace tamd;

// One direct pass against the eager pass of the same launch list, every graph output compared byte for byte.  Two pseudo-random
// inputs: eager(A) -> want; eager(B) leaves B's results in every buffer; direct(A) must bring want back -- a pass that writes
// nothing, or the wrong thing, shows up, and outputs that are prerun constants (PriorBox) are the same in all three.  0: identical.
static void selfcheck_noise(const tamd_graph* g, const IOBind& io, unsigned* lcg_state, std::vector<unsigned char>* noise)
{
    unsigned lcg = *lcg_state;
    noise->resize(io.bytes);
    if (g->tensors[io.tensor].dtype == TAMD_DT_FP32) {           // finite, moderate floats
        float* f = (float*)noise->data();
        for (size_t i = 0; i < io.bytes / 4; i++) { lcg = lcg * 1664525u + 1013904223u; f[i] = (float)((int)(lcg >> 20) - 2048) / 1024.f; }
    } else
        for (size_t i = 0; i < io.bytes; i++) { lcg = lcg * 1664525u + 1013904223u; (*noise)[i] = (unsigned char)(lcg >> 24); }
    *lcg_state = lcg;
}

static int direct_selfcheck(tamd_graph* g)
{
    std::vector<std::vector<unsigned char>> want, got;
    auto fill_inputs = [&](unsigned seed) -> int {
        std::vector<unsigned char> noise;
        unsigned lcg = seed;
        for (auto& io : g->inputs) {
            selfcheck_noise(g, io, &lcg, &noise);
            HIPCHK(hipMemcpy(io.stage, noise.data(), io.bytes, hipMemcpyHostToDevice));
        }
        return 0;
    };
    auto snapshot = [&](std::vector<std::vector<unsigned char>>& dst) -> int {
        dst.clear();
        for (auto& io : g->outputs) {
            dst.emplace_back(io.bytes);
            HIPCHK(hipMemcpy(dst.back().data(), io.stage, io.bytes, hipMemcpyDeviceToHost));
        }
        return 0;
    };
    if (fill_inputs(0x5EED1234u) || run_steps(g, g->stream)) return -1;
    HIPCHK(hipStreamSynchronize(g->stream));
    if (snapshot(want)) return -1;
    if (fill_inputs(0x0BADF00Du) || run_steps(g, g->stream)) return -1;
    HIPCHK(hipStreamSynchronize(g->stream));
    if (fill_inputs(0x5EED1234u)) return -1;
    HIPCHK(hipDeviceSynchronize());
    if (direct_submit(g->direct) || direct_wait(g->direct)) { set_error("direct pass failed: %s", direct_last_error()); return -1; }
    if (snapshot(got)) return -1;
    for (auto& io : g->inputs) HIPCHK(hipMemset(io.stage, 0, io.bytes));
    HIPCHK(hipDeviceSynchronize());
    for (size_t i = 0; i < want.size(); i++)
        if (want[i] != got[i]) { set_error("the direct pass does not reproduce the eager pass (output %zu differs)", i); return -1; }
    return 0;
}

// ---- host-to-host lists: graph outputs stored straight into the pinned host buffers ---------------------------------------------
// A blocking run_graph is upload kernel -> launch list -> download kernel -> closing packet.  The download kernel copies a few
// hundred bytes (MobileNet: 1000) that the last compute launch has just written; it costs a launch boundary, a kernel and an
// HBM round trip for nothing.  In the RECORDED list of an I/O slot every kernel argument that holds an output's device staging
// address is re-pointed at the slot's pinned host buffer (device-mapped: the download kernel already writes there), and the
// download launch is dropped; the closing packet's system-scope release makes the stores visible to the host as before.  Only
// outputs nobody else reads on the device qualify (a consumer would otherwise read host memory), and only when at least one
// argument matched; the patched program must then reproduce the eager list byte for byte (direct_io_selfcheck) or it is rebuilt
// with its download launches.
static bool io_zero_copy_wanted()
{
    const char* e = getenv("TAMD_IO_ZERO_COPY");
    return !(e && atoi(e) == 0);
}

static int patch_pointer(std::vector<LaunchRec>& recs, size_t nrecs, const void* from, const void* to)
{
    int hits = 0;
    for (size_t r = 0; r < nrecs; r++)
        for (size_t off = 0; off + 8 <= recs[r].args.size(); off += 8) {
            const void* v;
            memcpy(&v, recs[r].args.data() + off, 8);
            if (v == from) { memcpy(recs[r].args.data() + off, &to, 8); hits++; }
        }
    return hits;
}

// recs = the recorded host-to-host list of `slot`: [uploads][in_steps, steps, out_steps][one download launch per output].
// Returns true when every download launch could be dropped.
static bool zero_copy_outputs(tamd_graph* g, std::vector<LaunchRec>& recs, int slot)
{
    const size_t nout = g->outputs.size();
    if (nout == 0 || recs.size() <= nout + g->inputs.size()) return false;
    const size_t body = recs.size() - nout;
    std::vector<LaunchRec> trial(recs.begin(), recs.begin() + body);
    for (auto& io : g->outputs) {
        const HTensor& t = g->tensors[io.tensor];
        if (io.stage == t.dptr && count_consumers(g, io.tensor) != 1) return false;      // read again on the device
        if (t.prerun_const) return false;                                                  // written once at prerun, not by the list
        void* dev = nullptr;
        if (hipHostGetDevicePointer(&dev, slot ? io.pinned2 : io.pinned, 0) != hipSuccess) { (void)hipGetLastError(); return false; }
        if (patch_pointer(trial, body, io.stage, dev) < 1) return false;
    }
    recs.swap(trial);
    return true;
}

// The mirror image for graph inputs (TAMD_IO_ZERO_COPY_IN=1; off by default until it measures faster): the first compute launch
// reads the slot's pinned host buffer itself (device-mapped, uncached on the device side) and the upload launches are dropped.
// recs = [one upload launch per input][the rest]; every later argument that holds an input's staging address is re-pointed.
static bool zero_copy_inputs(tamd_graph* g, std::vector<LaunchRec>& recs, int slot)
{
    const char* e = exp_env("TAMD_IO_ZERO_COPY_IN");
    const size_t nin = g->inputs.size();
    if (!(e && atoi(e) == 1) || nin == 0 || recs.size() <= nin) return false;
    std::vector<LaunchRec> trial(recs.begin() + nin, recs.end());
    for (auto& io : g->inputs) {
        void* dev = nullptr;
        if (hipHostGetDevicePointer(&dev, slot ? io.pinned2 : io.pinned, 0) != hipSuccess) { (void)hipGetLastError(); return false; }
        if (patch_pointer(trial, trial.size(), io.stage, dev) < 1) return false;
    }
    recs.swap(trial);
    return true;
}

// the host-to-host program of `slot` against the eager list of the same slot (upload and download launches included), every
// pinned output compared byte for byte.  0: identical.
static int direct_io_selfcheck(tamd_graph* g, DirectProgram* pio, int slot)
{
    std::vector<unsigned char> noise;
    std::vector<std::vector<unsigned char>> want;
    unsigned lcg = 0xC0FFEE11u + (unsigned)slot;
    for (auto& io : g->inputs) {
        selfcheck_noise(g, io, &lcg, &noise);
        memcpy(slot ? io.pinned2 : io.pinned, noise.data(), io.bytes);
    }
    if (run_steps(g, g->stream, slot)) return -1;
    HIPCHK(hipStreamSynchronize(g->stream));
    for (auto& io : g->outputs) {
        unsigned char* pin = (unsigned char*)(slot ? io.pinned2 : io.pinned);
        want.emplace_back(pin, pin + io.bytes);
        memset(pin, 0xA5, io.bytes);
    }
    HIPCHK(hipDeviceSynchronize());
    unsigned long long b = 0;
    if (direct_submit(pio, true, &b) || direct_wait_burst(pio, b)) { set_error("direct host-to-host pass failed: %s", direct_last_error()); return -1; }
    for (size_t i = 0; i < g->outputs.size(); i++)
        if (memcmp(want[i].data(), slot ? g->outputs[i].pinned2 : g->outputs[i].pinned, g->outputs[i].bytes) != 0) {
            set_error("the direct host-to-host pass does not reproduce the eager list (output %zu differs)", i);
            return -1;
        }
    return 0;
}

// A graph lives on the device it was pre-run on; its entry points may be called from any host thread -- ONE at a time per graph
// (include/tengine_amd.h "Threading": run state and the single-producer HSA queue are not locked) -- whose current HIP
// device is whatever that thread used last (events, eager launches and temporary allocations would land on the wrong
// device otherwise).  hipSetDevice is a thread-local assignment when nothing changes.
// One graph = one thread at a time: every entry point that changes the graph or touches its buffers holds the graph for the
// duration of the call (nested entry points of the SAME thread pass).  Calls from different threads one after the other are fine --
// Tengine's scheduler does that -- two at once are a caller's bug that used to show up as corrupted launch lists; now the second
// call fails with an error.
namespace {
struct OneThread {
    tamd_graph* g;
    bool ok = true, outer = false;
    explicit OneThread(tamd_graph* g_) : g(g_)
    {
        static thread_local char marker;
        const unsigned long me = (unsigned long)(uintptr_t)&marker;
        if (!g) return;
        unsigned long none = 0;
        if (g->owner.compare_exchange_strong(none, me)) outer = true;
        else if (none != me) ok = false;
    }
    ~OneThread() { if (g && outer) g->owner.store(0); }
};
}  // namespace
#define TAMD_ONE_THREAD(g_)                                                                                                          \
    OneThread one_thread_(g_);                                                                                                       \
    if (!one_thread_.ok) { set_error("this tamd_graph is inside a call on another thread: one graph = one thread at a time (include/tengine_amd.h)"); return -1; }

static int bind_device(tamd_graph* g)
{
    if (!g) { set_error("null graph"); return -1; }
    HIPCHK(hipSetDevice(g->gpu));
    return 0;
}

// the TG_DEBUG_TIME analogue (source/device/cpu/cpu_dump.c:607-697 prints per-node times at postrun): one table per device
// graph on stderr when tamd_options.profile is set or TG_DEBUG_TIME=1 is in the environment, as the CPU device honours it
static void dump_profile(tamd_graph* g)
{
    const int n = (int)g->steps.size();
    if (n == 0) return;
    std::vector<tamd_kernel_info> k(n);
    if (tamd_graph_profile(g, 10, k.data(), n) < 0) return;
    double tot = 0;
    for (auto& e : k) tot += e.ms;
    fprintf(stderr, "Tengine HIP device graph: %d launches, %.3f ms per run (sum of launches)\n", n, tot);
    for (int i = 0; i < n; i++)
        fprintf(stderr, "  %3d %-40s %-32s %8.2f us %6.2f%% %9.2f MMAC %9.1f KB\n", i, k[i].node, k[i].kernel, 1e3 * k[i].ms,
                tot > 0 ? 100.0 * k[i].ms / tot : 0.0, k[i].macs / 1e6, k[i].bytes / 1e3);
}

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

int tamd_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int tamd_init(int gpu_index)
{
    int n = tamd_device_count();
    if (n <= 0) { set_error("no HIP device visible: the tengine_amd backend needs an MI355X (it has no CPU fallback)"); return -1; }
    if (gpu_index < 0 || gpu_index >= n) { set_error("gpu_index %d out of range (%d devices)", gpu_index, n); return -1; }
    HIPCHK(hipSetDevice(gpu_index));
    return 0;
}

int tamd_shutdown(void) { return 0; }
const char* tamd_last_error(void) { return g_err; }
const char* tamd_version(void) { return "tengine_amd 0.4 (gfx950)"; }

int tamd_op_supported(int op, int dtype)
{
    if (dtype != TAMD_DT_INT8 && dtype != TAMD_DT_UINT8 && dtype != TAMD_DT_FP32) return 0;
    if (op == TAMD_OP_UPSAMPLE) return dtype != TAMD_DT_INT8;      // nearest upsample: uint8 / fp32 graphs
    if (op == TAMD_OP_RELU6) return dtype == TAMD_DT_FP32;
    if (op == TAMD_OP_SOFTMAX) return 1;                           // int8: over the channel axis only (tamd_node_supported)
    if (op == TAMD_OP_RESHAPE || op == TAMD_OP_PRIORBOX) return dtype != TAMD_DT_INT8;   // dense NCHW device tensors: uint8 / fp32 graphs
    if (op == TAMD_OP_PERMUTE) return dtype == TAMD_DT_UINT8;      // SSD heads (Permute -> Flatten -> Concat), uint8 graphs
    switch (op) {
    case TAMD_OP_INPUT: case TAMD_OP_CONST: case TAMD_OP_CONV: case TAMD_OP_FC: case TAMD_OP_POOL: case TAMD_OP_RELU:
    case TAMD_OP_ELTWISE: case TAMD_OP_CONCAT: case TAMD_OP_DROPOUT: case TAMD_OP_FLATTEN:
        return 1;
    default:
        return 0;
    }
}

// What allocator.describe cannot say with an operator list alone: is THIS node, with these parameters and tensors, one the
// planners compile?  The Tengine plugin asks before it claims a subgraph (hip_device.cc: subgraph_runs_on_device), so
// anything else lands on the CPU device instead of failing pre_run.  Mirrors the planners' own conditions.
int tamd_node_supported(const tamd_node_desc* n, const tamd_tensor_desc* in, int n_in, const tamd_tensor_desc* out, int n_out)
{
    if (!n || n_out < 1 || !out) return 0;
    const int dt = out[0].dtype;
    if (!tamd_op_supported(n->op, dt)) return 0;
    for (int i = 0; i < n_out; i++)
        if (out[i].dtype != dt || (dt != TAMD_DT_FP32 && out[i].quant_num != 1)) return 0;
    for (int i = 0; i < n_in; i++)
        if (in[i].ttype != TAMD_TT_CONST && (in[i].dtype != dt || (dt != TAMD_DT_FP32 && in[i].quant_num != 1))) return 0;
    auto elems = [](const tamd_tensor_desc& t) { size_t e = 1; for (int i = 0; i < t.dim_num; i++) e *= (size_t)t.dims[i]; return e; };
    switch (n->op) {
    case TAMD_OP_CONV: {
        if (n_in < 2 || !n->param || in[0].dim_num != 4 || in[1].dim_num != 4 || in[1].ttype != TAMD_TT_CONST) return 0;
        const tamd_conv_param& p = *(const tamd_conv_param*)n->param;
        const int cin = in[0].dims[1], cout = in[1].dims[0];
        const int kh = p.kernel_h ? p.kernel_h : 1, kw = p.kernel_w ? p.kernel_w : 1;
        if (p.group < 1 || cin % p.group || cout % p.group) return 0;
        if (elems(in[1]) != (size_t)cout * (cin / p.group) * kh * kw) return 0;
        if (in[1].quant_num != 0 && in[1].quant_num != 1 && in[1].quant_num != cout) return 0;
        if (dt != TAMD_DT_FP32 && in[1].quant_num == 0) return 0;
        if (dt == TAMD_DT_UINT8 && in[1].quant_num != 1) return 0;                 // per-tensor weights (conv_kernel_x86.c:76-79)
        if (dt == TAMD_DT_INT8 && p.group == 1 && kh * kw > 128 && cin > 4) return 0;  // tap table of the implicit GEMM
        if (n_in > 2 && (in[2].ttype != TAMD_TT_CONST || elems(in[2]) < (size_t)cout)) return 0;
        return 1;
    }
    case TAMD_OP_FC: {
        if (n_in < 2 || in[1].ttype != TAMD_TT_CONST || in[0].dim_num < 2) return 0;
        const size_t hidden = elems(in[0]) / (size_t)(in[0].dims[0] > 0 ? in[0].dims[0] : 1);
        const int nout = out[0].dim_num > 1 ? out[0].dims[1] : 0;
        if (nout < 1 || in[1].dim_num != 2 || in[1].dims[0] != nout || (size_t)in[1].dims[1] != hidden) return 0;   // [num_output][hidden] only (fc.c:43-97)
        if (n->param && ((const tamd_fc_param*)n->param)->num_output && ((const tamd_fc_param*)n->param)->num_output != nout) return 0;
        if (dt == TAMD_DT_UINT8 && hidden * 4 > 60000) return 0;                    // fc_u8 keeps the input row in LDS
        return 1;
    }
    case TAMD_OP_ELTWISE: {
        const int ty = n->param ? ((const tamd_eltwise_param*)n->param)->type : -1;
        if (n_in != 2 || (ty != 0 && ty != 2 && ty != 4 && ty != 6)) return 0;
        if (in[0].dim_num != in[1].dim_num) return 0;
        for (int i = 0; i < in[0].dim_num; i++) if (in[0].dims[i] != in[1].dims[i]) return 0;
        return in[0].ttype != TAMD_TT_CONST && in[1].ttype != TAMD_TT_CONST;
    }
    case TAMD_OP_CONCAT: {
        int ax = n->param ? ((const tamd_concat_param*)n->param)->axis : 1;
        if (ax < 0) ax += out[0].dim_num;
        if (dt == TAMD_DT_INT8) return ax == 1 && out[0].dim_num >= 2;       // NHWC device tensors: channel concat
        return ax >= 0 && ax < out[0].dim_num;                               // dense NCHW: any axis
    }
    case TAMD_OP_SOFTMAX: {
        if (dt != TAMD_DT_INT8) return 1;
        // int8 tensors are NHWC on the device: the channel axis of a 2-D / 4-D tensor is the contiguous one (softmax_i8_kernel)
        if (n_in < 1 || (in[0].dim_num != 2 && in[0].dim_num != 4) || in[0].ttype == TAMD_TT_CONST) return 0;
        int ax = n->param ? ((const tamd_softmax_param*)n->param)->axis : 1;
        if (ax < 0) ax += in[0].dim_num;
        return ax == 1 && in[0].dims[1] >= 1 && in[0].dims[1] <= kSoftmaxI8MaxC;
    }
    case TAMD_OP_PRIORBOX: {
        if (!n->param || n_in < 2 || in[0].dim_num != 4 || in[1].dim_num != 4 || out[0].dim_num < 1 || out[0].dims[0] != 1) return 0;
        const tamd_priorbox_param& p = *(const tamd_priorbox_param*)n->param;
        return p.min_size_num >= 1 && p.min_size_num <= TAMD_PRIORBOX_MAX && p.aspect_ratio_num >= 0 && p.aspect_ratio_num <= TAMD_PRIORBOX_MAX
               && (p.max_size_num == 0 || p.max_size_num == p.min_size_num);
    }
    case TAMD_OP_PERMUTE: {
        if (!n->param || out[0].dim_num != 4) return 0;
        const int* o = ((const tamd_permute_param*)n->param)->order;
        return o[0] == 0 && o[1] == 2 && o[2] == 3 && o[3] == 1;
    }
    case TAMD_OP_UPSAMPLE: {
        const float sc = n->param ? ((const tamd_upsample_param*)n->param)->scale : 0.f;
        return sc >= 1.f && sc == (float)(int)sc;
    }
    case TAMD_OP_POOL: {
        if (!n->param || in[0].dim_num != 4) return 0;
        const int m = ((const tamd_pool_param*)n->param)->pool_method;
        return m == 0 || m == 1;
    }
    default:
        return 1;
    }
}

tamd_graph* tamd_graph_create(void) { return new tamd_graph(); }

int tamd_graph_add_tensor(tamd_graph* g, const tamd_tensor_desc* d)
{
    if (!g || !d) return -1;
    HTensor t;
    t.dtype = d->dtype; t.ttype = d->ttype;
    t.dims.assign(d->dims, d->dims + d->dim_num);
    if (d->name) t.name = d->name;
    if (d->quant_num > 0 && d->scales) {
        t.scales.assign(d->scales, d->scales + d->quant_num);
        if (d->zero_points) t.zps.assign(d->zero_points, d->zero_points + d->quant_num);
        else t.zps.assign(d->quant_num, 0);
    }
    if (d->ttype == TAMD_TT_CONST) {
        size_t bytes = t.elems() * esize(t.dtype);
        t.data.resize(bytes);
        if (d->data) memcpy(t.data.data(), d->data, bytes);     // NULL payload == zero-filled (tm2_serializer.c:240-246)
    }
    g->tensors.push_back(std::move(t));
    return (int)g->tensors.size() - 1;
}

int tamd_graph_add_node(tamd_graph* g, const tamd_node_desc* d)
{
    if (!g || !d) return -1;
    HNode n;
    n.op = d->op;
    if (d->op == TAMD_OP_SOFTMAX) n.p.softmax.axis = 1;
    if (d->name) n.name = d->name;
    for (int i = 0; i < d->input_num; i++) {
        if (d->inputs[i] < 0 || d->inputs[i] >= (int)g->tensors.size()) { set_error("node %s: bad input tensor", n.name.c_str()); return -1; }
        n.in.push_back(d->inputs[i]);
    }
    for (int i = 0; i < d->output_num; i++) {
        if (d->outputs[i] < 0 || d->outputs[i] >= (int)g->tensors.size()) { set_error("node %s: bad output tensor", n.name.c_str()); return -1; }
        n.out.push_back(d->outputs[i]);
    }
    if (d->param) {
        switch (d->op) {
        case TAMD_OP_CONV: n.p.conv = *(const tamd_conv_param*)d->param; break;
        case TAMD_OP_FC: n.p.fc = *(const tamd_fc_param*)d->param; break;
        case TAMD_OP_POOL: n.p.pool = *(const tamd_pool_param*)d->param; break;
        case TAMD_OP_RELU: n.p.relu = *(const tamd_relu_param*)d->param; break;
        case TAMD_OP_ELTWISE: n.p.elt = *(const tamd_eltwise_param*)d->param; break;
        case TAMD_OP_CONCAT: n.p.concat = *(const tamd_concat_param*)d->param; break;
        case TAMD_OP_UPSAMPLE: n.p.ups = *(const tamd_upsample_param*)d->param; break;
        case TAMD_OP_PERMUTE: n.p.perm = *(const tamd_permute_param*)d->param; break;
        case TAMD_OP_SOFTMAX: n.p.softmax = *(const tamd_softmax_param*)d->param; break;
        case TAMD_OP_RESHAPE: n.p.reshape = *(const tamd_reshape_param*)d->param; break;
        case TAMD_OP_PRIORBOX: n.p.priorbox = *(const tamd_priorbox_param*)d->param; break;
        default: break;
        }
    }
    g->nodes.push_back(std::move(n));
    return (int)g->nodes.size() - 1;
}

int tamd_graph_set_inputs(tamd_graph* g, int n, const int* ids)
{
    g->inputs.clear();
    for (int i = 0; i < n; i++) { IOBind b; b.tensor = ids[i]; g->inputs.push_back(b); }
    return 0;
}

int tamd_graph_set_outputs(tamd_graph* g, int n, const int* ids)
{
    g->outputs.clear();
    for (int i = 0; i < n; i++) { IOBind b; b.tensor = ids[i]; g->outputs.push_back(b); }
    return 0;
}

int tamd_graph_set_batch(tamd_graph* g, int batch)
{
    if (!g) { set_error("null graph"); return -1; }
    TAMD_ONE_THREAD(g);
    if (g->prepared) { set_error("set_batch after prerun"); return -1; }
    for (auto& io : g->inputs) g->tensors[io.tensor].dims[0] = batch;
    return 0;
}

int tamd_graph_prerun(tamd_graph* g, const tamd_options* opt)
{
    TAMD_ONE_THREAD(g);
    if (!g) return -1;
    if (g->prepared) return 0;
    std::lock_guard<std::mutex> lk(g_capture_mutex);
    tamd_options o{};
    o.dev_name = "HIP"; o.size = (int)sizeof(tamd_options); o.gpu_index = 0; o.use_hip_graph = 1; o.profile = 0;
    if (opt) {             // options may be NULL (scheduler.c:49-59); only the fields the caller's blob really holds are read
        const int have = opt->size;
        if (have >= (int)(offsetof(tamd_options, gpu_index) + sizeof(int))) o.gpu_index = opt->gpu_index;
        if (have >= (int)(offsetof(tamd_options, use_hip_graph) + sizeof(int))) o.use_hip_graph = opt->use_hip_graph;
        if (have >= (int)(offsetof(tamd_options, profile) + sizeof(int))) o.profile = opt->profile;
        if (have >= (int)(offsetof(tamd_options, direct_dispatch) + sizeof(int))) o.direct_dispatch = opt->direct_dispatch;
        if (have >= (int)(offsetof(tamd_options, keep_tensors) + sizeof(int))) o.keep_tensors = opt->keep_tensors;
        if (have >= (int)(offsetof(tamd_options, u8_integer) + sizeof(int))) o.u8_integer = opt->u8_integer;
    }
    if (const char* ui = getenv("TAMD_U8_INT")) o.u8_integer = atoi(ui) != 0;
    if (const char* dd = getenv("TAMD_DIRECT_DISPATCH")) o.direct_dispatch = atoi(dd) != 0;
    // a tool that intercepts HSA queues (rocprofv3) crashes in its doorbell handler on packets it did not see HIP write
    // (ROCm 7.2: SIGSEGV inside the interceptor on the first pass, profiles/r02_direct_dispatch.txt): under such a tool the graph
    // keeps the hipGraph replay -- the same kernels with the same arguments, so per-kernel figures are unaffected
    if (o.direct_dispatch && !getenv("TAMD_DIRECT_UNDER_TOOLS")) {
        const char* pre = getenv("LD_PRELOAD");
        if (getenv("HSA_TOOLS_LIB") || getenv("ROCP_TOOL_LIBRARIES") || getenv("ROCPROFILER_LIBRARY") || (pre && strstr(pre, "rocprof"))) o.direct_dispatch = 0;
    }
    if (const char* dt = getenv("TG_DEBUG_TIME")) { if (atoi(dt) == 1) o.profile = 1; }     // cpu_define.h:41-43
    const auto prerun_t0 = std::chrono::steady_clock::now();
    g->opt = o;
    if (infer_shapes(g) || validate_graph(g)) return -1;       // a malformed graph is refused before the device is touched
    if (tamd_init(o.gpu_index)) return -1;
    // the planner picks the COHERENT kernel instances (agent-scope loads / write-through stores, slower under a hipGraph) only
    // where direct dispatch can exist at all: HSA agent and loader extension are asked before anything is planned
    if (o.direct_dispatch && !(o.use_hip_graph && direct_probe(o.gpu_index))) { o.direct_dispatch = 0; g->opt = o; }
    g->gpu = o.gpu_index;
    HIPCHK(hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking));
    // one quantisation scheme per device graph (the reference's splitter hands over homogeneous subgraphs)
    bool any_u8 = false, any_f32 = false, any_i8 = false;
    for (auto& t : g->tensors)
        if (t.ttype != TAMD_TT_CONST) (t.dtype == TAMD_DT_UINT8 ? any_u8 : t.dtype == TAMD_DT_FP32 ? any_f32 : any_i8) = true;
    if ((int)any_u8 + (int)any_f32 + (int)any_i8 > 1) { set_error("mixed int8 / uint8 / fp32 activations in one device graph"); return -1; }
    if (any_u8 ? plan_u8(g) : any_f32 ? plan_f32(g) : plan(g)) return -1;
    // launches whose inputs are all prerun constants (the Concat of the PriorBox outputs) run now and never again
    for (auto& st : g->steps)
        if (st.once) HIPCHK(st.fn(g->stream));
    g->steps.erase(std::remove_if(g->steps.begin(), g->steps.end(), [](const Step& st) { return st.once; }), g->steps.end());
    for (auto* v : {&g->inputs, &g->outputs})
        for (auto& io : *v) HIPCHK(hipHostMalloc(&io.pinned2, std::max<size_t>(io.bytes, 16), hipHostMallocDefault));
    HIPCHK(hipDeviceSynchronize());
    if (o.use_hip_graph) {
        // one warm eager pass (module load), then capture compute + output layout launches
        if (run_steps(g, g->stream)) return -1;
        HIPCHK(hipStreamSynchronize(g->stream));
        HIPCHK(hipStreamBeginCapture(g->stream, hipStreamCaptureModeThreadLocal));
        int rc = run_steps(g, g->stream);
        hipError_t e = hipStreamEndCapture(g->stream, &g->hgraph);
        if (rc) return -1;
        HIPCHK(e);
        const char* ne = exp_env("TAMD_GRAPH_EXECS");
        g->nexec = ne ? std::max(1, std::min(4, atoi(ne))) : 3;
        for (int i = 0; i < g->nexec; i++) HIPCHK(hipGraphInstantiate(&g->hexecs[i], g->hgraph, nullptr, nullptr, 0));
        g->hexec = g->hexecs[0];
        const char* ioenv = exp_env("TAMD_IO_GRAPH");
        for (int slot = 0; slot < 2 && !(ioenv && atoi(ioenv) == 0); slot++) {      // the host-to-host variants (upload / download as launches)
            HIPCHK(hipStreamBeginCapture(g->stream, hipStreamCaptureModeThreadLocal));
            rc = run_steps(g, g->stream, slot);
            e = hipStreamEndCapture(g->stream, &g->hgraph_io[slot]);
            if (rc) return -1;
            HIPCHK(e);
            for (int i = 0; i < 2; i++) HIPCHK(hipGraphInstantiate(&g->hexec_io[slot][i], g->hgraph_io[slot], nullptr, nullptr, 0));
        }
        if (o.direct_dispatch) {
            // one more eager pass with the launch recorder on (launch_rec.h), then the list as AQL packets (direct.cc); every
            // module of the list has launched by now, so its code object is loaded and its kernel descriptors resolve
            std::vector<LaunchRec> recs;
            g_launch_rec = &recs;
            rc = run_steps(g, g->stream);
            g_launch_rec = nullptr;
            if (rc) return -1;
            HIPCHK(hipStreamSynchronize(g->stream));
            const char* why = "";
            g->direct = direct_build(g->gpu, g->stream, recs, &why);
            if (!g->direct) fprintf(stderr, "tengine_amd: direct dispatch not used (hipGraph replay instead): %s\n", why);
            // the packets carry hand-built argument segments (hidden arguments at the code-object-v5 offsets) and hand-picked
            // fence scopes: ONE direct pass must reproduce the eager pass byte for byte on a non-trivial input, or the graph
            // keeps its hipGraph (a different ROCm, a renamed kernel, a stale line would otherwise be silently wrong outputs)
            if (g->direct && !exp_env("TAMD_EXP_NOFENCE") && direct_selfcheck(g)) {
                fprintf(stderr, "tengine_amd: direct dispatch DISABLED for this graph: %s (hipGraph replay instead)\n", g_err);
                direct_destroy(g->direct);
                g->direct = nullptr;
            }
            for (int slot = 0; slot < 2 && g->direct && g->hexec_io[slot][0]; slot++) {
                // the host-to-host list of I/O slot 0 | 1 (upload launch, compute, download launch) on the same queue:
                // tamd_graph_run and the asynchronous pair
                recs.clear();
                g_launch_rec = &recs;
                rc = run_steps(g, g->stream, slot);
                g_launch_rec = nullptr;
                if (rc) return -1;
                HIPCHK(hipStreamSynchronize(g->stream));
                // outputs straight into the pinned host buffers (no download launch) where the list allows it
                std::vector<LaunchRec> with_downloads = recs;
                bool zc = io_zero_copy_wanted() && zero_copy_outputs(g, recs, slot);
                if (zero_copy_inputs(g, recs, slot)) zc = true;        // (a failed self-check below falls back to the full list)
                DirectProgram* pio = direct_build(g->gpu, g->stream, recs, &why, g->direct);
                bool zc_ok = zc && pio;
                if (pio && direct_io_selfcheck(g, pio, slot)) {
                    if (zc) fprintf(stderr, "tengine_amd: zero-copy outputs DISABLED for this graph: %s\n", g_err);
                    direct_destroy(pio);
                    pio = nullptr; zc_ok = false; why = g_err;
                    if (zc) {                                // once more with the download launches
                        pio = direct_build(g->gpu, g->stream, with_downloads, &why, g->direct);
                        if (pio && direct_io_selfcheck(g, pio, slot)) { direct_destroy(pio); pio = nullptr; why = g_err; }
                    }
                }
                if (slot == 0) g->io_zero_copy = zc_ok;
                else g->io_zero_copy2 = zc_ok;
                if (!pio) fprintf(stderr, "tengine_amd: direct dispatch not used for host-to-host runs (slot %d): %s\n", slot, why);
                (slot ? g->direct_io2 : g->direct_io) = pio;
            }
            if (!g->direct_io || !g->direct_io2) {          // both or none: the asynchronous pair alternates between them
                if (g->direct_io2) { direct_destroy(g->direct_io2); g->direct_io2 = nullptr; }
            }
            if (getenv("TAMD_DEBUG") && g->direct_io)
                fprintf(stderr, "[tamd] host-to-host list: %d packets%s\n", direct_packets(g->direct_io), g->io_zero_copy ? ", outputs stored straight into the pinned host buffers" : "");
        }
    }
    plan_cache_flush();
    g->prerun_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - prerun_t0).count();
    g->prepared = true;
    return 0;
}

int tamd_graph_input_num(const tamd_graph* g) { return (int)g->inputs.size(); }
int tamd_graph_output_num(const tamd_graph* g) { return (int)g->outputs.size(); }

static int fill_desc(const HTensor& t, int* dims8, int* dtype)
{
    for (size_t i = 0; i < t.dims.size() && i < 8; i++) dims8[i] = t.dims[i];
    if (dtype) *dtype = t.dtype;
    return (int)t.dims.size();
}

int tamd_graph_input_desc(const tamd_graph* g, int idx, int* dims8, int* dtype)
{
    if (idx < 0 || idx >= (int)g->inputs.size()) return -1;
    return fill_desc(g->tensors[g->inputs[idx].tensor], dims8, dtype);
}

int tamd_graph_output_desc(const tamd_graph* g, int idx, int* dims8, int* dtype, float* scale, int* zp)
{
    if (idx < 0 || idx >= (int)g->outputs.size()) return -1;
    const HTensor& t = g->tensors[g->outputs[idx].tensor];
    if (scale) *scale = t.scales.empty() ? 0.f : t.scales[0];
    if (zp) *zp = t.zps.empty() ? 0 : t.zps[0];
    return fill_desc(t, dims8, dtype);
}

// (the setters change io.host_in / io.host_out, which run / run_async / wait read: they hold the graph like every other entry point --
//  a set_input from a second thread DURING a run is refused instead of racing)
int tamd_graph_set_input(tamd_graph* g, int idx, const void* host, size_t bytes)
{
    if (!g) { set_error("null graph"); return -1; }
    TAMD_ONE_THREAD(g);
    if (idx < 0 || idx >= (int)g->inputs.size()) { set_error("bad input index"); return -1; }
    if (g->prepared && bytes != g->inputs[idx].bytes) { set_error("input %d: %zu bytes given, %zu expected", idx, bytes, g->inputs[idx].bytes); return -1; }
    g->inputs[idx].host_in = host;
    return 0;
}

int tamd_graph_set_output(tamd_graph* g, int idx, void* host, size_t bytes)
{
    if (!g) { set_error("null graph"); return -1; }
    TAMD_ONE_THREAD(g);
    if (idx < 0 || idx >= (int)g->outputs.size()) { set_error("bad output index"); return -1; }
    if (g->prepared && bytes != g->outputs[idx].bytes) { set_error("output %d: %zu bytes given, %zu expected", idx, bytes, g->outputs[idx].bytes); return -1; }
    g->outputs[idx].host_out = host;
    return 0;
}

// passes submitted by direct dispatch are not on the HIP stream: everything that touches the tensors waits for them first
static int direct_drain(tamd_graph* g)
{
    if (g->direct && g->direct_busy) {
        if (direct_wait(g->direct)) { set_error("direct dispatch: %s", direct_last_error()); return -1; }
        g->direct_busy = false;
    }
    return 0;
}

// the direct path of this graph is unusable (queue fault, a burst that never completed): forget the runs in flight -- their
// results are lost, the caller has been told -- and go back to the hipGraph executables, which every entry point still has
static void direct_abandon(tamd_graph* g, const char* why)
{
    fprintf(stderr, "tengine_amd: direct dispatch abandoned for this graph (%s): hipGraph replay from here on\n", why);
    g->inflight.erase(std::remove_if(g->inflight.begin(), g->inflight.end(), [](const Inflight& f) { return f.direct; }), g->inflight.end());
    // direct_destroy waits for what is still running unless the queue has faulted; a hung burst is bounded by TAMD_DIRECT_TIMEOUT_S
    if (g->direct_io2) { direct_destroy(g->direct_io2); g->direct_io2 = nullptr; }
    if (g->direct_io) { direct_destroy(g->direct_io); g->direct_io = nullptr; }
    if (g->direct) { direct_destroy(g->direct); g->direct = nullptr; }
    g->direct_busy = false;
}

static inline long long now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static bool close_on_last_packet()
{
    const char* e = tamd_pin("direct_close_on_last");          // 0: a separate barrier packet closes the burst (round 2-3 behaviour)
    return !(e && atoi(e) == 0);
}

int tamd_graph_upload_inputs(tamd_graph* g)
{
    TAMD_ONE_THREAD(g);
    if (bind_device(g)) return -1;
    if (!g->inflight.empty()) { set_error("tamd_graph_upload_inputs while asynchronous runs are in flight (they own the pinned buffers): collect them with tamd_graph_wait first"); return -1; }
    if (direct_drain(g)) return -1;
    if (!g->prepared) { set_error("graph not prepared"); return -1; }
    for (auto& io : g->inputs) {
        if (!io.host_in) { set_error("input buffer not set"); return -1; }
        memcpy(io.pinned, io.host_in, io.bytes);
        HIPCHK(hipMemcpyAsync(io.stage, io.pinned, io.bytes, hipMemcpyHostToDevice, g->stream));
    }
    g->stream_dirty = true;
    return 0;
}

// A zero-copy host-to-host run leaves its outputs in the pinned host buffers ONLY (the launch that would have written the device
// staging buffer was re-pointed): whoever reads the device copy next -- tamd_graph_output_device (the RCCL gather),
// tamd_graph_read_tensor of a 1x1-map output -- gets it refreshed from the pinned slot first.
static int stage_from_pinned(tamd_graph* g)
{
    if (!g->out_fresh_in) return 0;
    if (!g->inflight.empty()) { set_error("the outputs of the last host-to-host run live in a pinned buffer that a run in flight may overwrite: tamd_graph_wait first"); return -1; }
    if (direct_drain(g)) return -1;
    for (auto& io : g->outputs)
        HIPCHK(hipMemcpyAsync(io.stage, g->out_fresh_in == 2 ? io.pinned2 : io.pinned, io.bytes, hipMemcpyHostToDevice, g->stream));
    HIPCHK(hipStreamSynchronize(g->stream));
    g->out_fresh_in = 0;
    return 0;
}

int tamd_graph_launch(tamd_graph* g)
{
    TAMD_ONE_THREAD(g);
    if (bind_device(g)) return -1;
    if (!g->prepared) { set_error("graph not prepared"); return -1; }
    g->out_fresh_in = 0;                       // the pass writes the staging buffers itself
    if (g->direct) {
        // the pass reads what the stream wrote (uploaded inputs): drain it before the first packet of a burst
        if (!g->direct_busy) { HIPCHK(hipStreamSynchronize(g->stream)); g->stream_dirty = false; }
        g->direct_busy = true;
        if (direct_submit(g->direct)) { set_error("direct dispatch: %s", direct_last_error()); return -1; }
        return 0;
    }
    g->stream_dirty = true;
    if (g->hexec) {
        hipGraphExec_t e = g->hexecs[g->next_exec];
        g->next_exec = (g->next_exec + 1) % g->nexec;
        HIPCHK(hipGraphLaunch(e, g->stream));
        return 0;
    }
    return run_steps(g, g->stream);
}

int tamd_graph_direct_packets(const tamd_graph* g) { return g && g->direct ? direct_packets(g->direct) : 0; }
int tamd_graph_direct_meta_packets(const tamd_graph* g) { return g && g->direct ? direct_meta_packets(g->direct) : 0; }
double tamd_graph_prerun_ms(const tamd_graph* g) { return g ? g->prerun_ms : 0.0; }

int tamd_graph_sync(tamd_graph* g)
{
    TAMD_ONE_THREAD(g);
    if (bind_device(g)) return -1;
    if (direct_drain(g)) return -1;
    // asynchronous runs that were submitted and not collected yet are device work too (their outputs stay in the pinned slots
    // until tamd_graph_wait delivers them)
    if (g->direct_io && !g->inflight.empty() && direct_wait_all(g->direct_io)) { set_error("direct dispatch: %s", direct_last_error()); return -1; }
    HIPCHK(hipStreamSynchronize(g->stream));
    g->stream_dirty = false;
    return 0;
}

int tamd_graph_download_outputs(tamd_graph* g)
{
    TAMD_ONE_THREAD(g);
    if (bind_device(g)) return -1;
    if (!g->inflight.empty()) { set_error("tamd_graph_download_outputs while asynchronous runs are in flight (they own the pinned buffers): collect them with tamd_graph_wait first"); return -1; }
    if (direct_drain(g)) return -1;
    if (g->out_fresh_in) {                     // the last pass was a zero-copy host-to-host run: its pinned slot IS the newest copy
        for (auto& io : g->outputs)
            if (io.host_out) memcpy(io.host_out, g->out_fresh_in == 2 ? io.pinned2 : io.pinned, io.bytes);
        return 0;
    }
    for (auto& io : g->outputs) HIPCHK(hipMemcpyAsync(io.pinned, io.stage, io.bytes, hipMemcpyDeviceToHost, g->stream));
    HIPCHK(hipStreamSynchronize(g->stream));
    for (auto& io : g->outputs)
        if (io.host_out) memcpy(io.host_out, io.pinned, io.bytes);
    return 0;
}

int tamd_graph_run(tamd_graph* g)
{
    TAMD_ONE_THREAD(g);
    if (!g || !g->prepared) { set_error("graph not prepared"); return -1; }
    if (!g->inflight.empty()) { set_error("tamd_graph_run while asynchronous runs are in flight: collect them with tamd_graph_wait first"); return -1; }
    if (bind_device(g)) return -1;
    if (direct_drain(g)) return -1;
    static const bool trace = getenv("TAMD_H2H_TRACE") && atoi(getenv("TAMD_H2H_TRACE")) == 1;
    long long t[6] = {0, 0, 0, 0, 0, 0};
    if (trace) t[0] = now_ns();
    for (auto& io : g->inputs) {
        if (!io.host_in) { set_error("input buffer not set"); return -1; }
        memcpy(io.pinned, io.host_in, io.bytes);
    }
    if (trace) t[1] = now_ns();
    if (g->direct_io) {
        // the same list as AQL packets: system-scope acquire in front; the burst is closed by the list's last packet (or a barrier
        // packet behind it).  The graph's HIP stream is drained only when something may be pending on it: what the pass reads was
        // either written by the pass itself (the upload launch) or by stream work this library knows about
        if (g->stream_dirty || g->stream_exposed) { HIPCHK(hipStreamSynchronize(g->stream)); g->stream_dirty = false; }
        if (trace) t[2] = now_ns();
        unsigned long long b = 0;
        int rc = close_on_last_packet() ? direct_submit(g->direct_io, true, &b) : (direct_submit(g->direct_io) || direct_close(g->direct_io, &b));
        if (trace) t[3] = now_ns();
        if (!rc) rc = direct_wait_burst(g->direct_io, b);
        if (rc) {
            set_error("direct dispatch: %s", direct_last_error());
            direct_abandon(g, g_err);
            return -1;
        }
    } else {
        if (trace) t[2] = t[3] = now_ns();
        if (launch_io(g, 0)) return -1;
        HIPCHK(hipStreamSynchronize(g->stream));
    }
    if (trace) t[4] = now_ns();
    g->out_fresh_in = (g->direct_io && g->io_zero_copy) ? 1 : 0;
    for (auto& io : g->outputs)
        if (io.host_out) memcpy(io.host_out, io.pinned, io.bytes);
    if (trace) {
        t[5] = now_ns();
        for (int i = 0; i < 5; i++) g->h2h_ns[i] += t[i + 1] - t[i];
        g->h2h_runs++;
    }
    return 0;
}

// ---- asynchronous runs: interface.async_run / async_wait of struct interface (source/device/device.h:60-63), which the
// reference's scheduler never reaches (run_graph(graph, 0) is rejected, scheduler.c:75-79).  Two runs may be in flight:
// while the device works on run k the host already stages run k+1 (its own pinned buffers), so launch and completion
// latencies overlap with device work instead of adding to every image.  Everything stays on the graph's one in-order
// stream: run k+1's H2D queues behind run k's D2H, results cannot mix.
int tamd_graph_run_async(tamd_graph* g)
{
    TAMD_ONE_THREAD(g);
    if (!g || !g->prepared) { set_error("graph not prepared"); return -1; }
    if (bind_device(g)) return -1;
    if (g->inflight.size() >= 2) { set_error("two runs are already in flight: call tamd_graph_wait first"); return -1; }
    if (direct_drain(g)) return -1;
    const int slot = g->next_slot;
    for (auto& io : g->inputs) {
        if (!io.host_in) { set_error("input buffer not set"); return -1; }
        memcpy(slot ? io.pinned2 : io.pinned, io.host_in, io.bytes);
    }
    Inflight f;
    f.slot = slot;
    for (auto& io : g->outputs) f.host_out.push_back(io.host_out);
    if (g->direct_io && g->direct_io2) {
        // the run is ONE burst on the graph's own HSA queue: the slot's host-to-host list (system-scope acquire in front: the
        // pinned input was just written by the host), closed by a barrier packet that releases at system scope and counts the
        // queue's completion signal down.  The second run's packets queue behind the first one's closing packet (barrier bit on
        // every packet): the device goes from run k's download straight into run k+1's upload, the host is never in between.
        DirectProgram* p = slot ? g->direct_io2 : g->direct_io;
        if (g->inflight.empty() && (g->stream_dirty || g->stream_exposed)) { HIPCHK(hipStreamSynchronize(g->stream)); g->stream_dirty = false; }
        const int rc = close_on_last_packet() ? direct_submit(p, true, &f.burst) : (direct_submit(p) || direct_close(p, &f.burst));
        if (rc) {
            // packets may be in the ring without a closing packet: the queue cannot be trusted any more
            set_error("direct dispatch: %s", direct_last_error());
            direct_abandon(g, g_err);
            return -1;
        }
        f.direct = true;
    } else {
        if (!g->slot_done[slot]) HIPCHK(hipEventCreateWithFlags(&g->slot_done[slot], hipEventDisableTiming));
        g->stream_dirty = true;
        if (launch_io(g, slot)) return -1;
        f.done = g->slot_done[slot];
        HIPCHK(hipEventRecord(f.done, g->stream));
    }
    g->inflight.push_back(f);
    g->next_slot ^= 1;
    return 0;
}

// blocks until the OLDEST run in flight is complete and its outputs are in the buffers that were set when it was submitted
int tamd_graph_wait(tamd_graph* g)
{
    TAMD_ONE_THREAD(g);
    if (!g || g->inflight.empty()) { set_error("tamd_graph_wait: no run in flight"); return -1; }
    if (bind_device(g)) return -1;
    const Inflight f = g->inflight.front();
    if (f.direct) {
        if (direct_wait_burst(g->direct_io, f.burst)) {
            // the run is lost; so is everything queued behind it.  Drop the bookkeeping (the graph would otherwise refuse every
            // entry point with "runs in flight" until it is destroyed) and leave the direct path
            set_error("direct dispatch: %s", direct_last_error());
            direct_abandon(g, g_err);
            return -1;
        }
    } else
        HIPCHK(hipEventSynchronize(f.done));
    for (size_t i = 0; i < g->outputs.size(); i++)
        if (f.host_out[i]) memcpy(f.host_out[i], f.slot ? g->outputs[i].pinned2 : g->outputs[i].pinned, g->outputs[i].bytes);
    g->out_fresh_in = (f.direct && (f.slot ? g->io_zero_copy2 : g->io_zero_copy)) ? 1 + f.slot : 0;
    g->inflight.erase(g->inflight.begin());
    return 0;
}

int tamd_graph_inflight(const tamd_graph* g) { return g ? (int)g->inflight.size() : 0; }

int tamd_graph_output_device(tamd_graph* g, int idx, void** dptr, size_t* bytes)
{
    if (!g) { set_error("null graph"); return -1; }
    TAMD_ONE_THREAD(g);
    if (idx < 0 || idx >= (int)g->outputs.size() || !g->prepared) return -1;
    if (g->out_fresh_in && (bind_device(g) || stage_from_pinned(g))) return -1;
    *dptr = g->outputs[idx].stage;
    *bytes = g->outputs[idx].bytes;
    return 0;
}

// once the caller holds the stream it may queue work there that this library cannot see: every direct burst drains it first again
void* tamd_graph_stream(tamd_graph* g) { g->stream_exposed = true; g->stream_dirty = true; return (void*)g->stream; }

int tamd_graph_time_launches(tamd_graph* g, int iters, float* total_ms)
{
    TAMD_ONE_THREAD(g);
    if (bind_device(g)) return -1;
    if (g->direct) {        // the passes are not on the stream: host clock around submit .. complete
        if (direct_drain(g)) return -1;
        HIPCHK(hipStreamSynchronize(g->stream));
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < iters; i++)
            if (tamd_graph_launch(g)) return -1;
        if (direct_drain(g)) return -1;
        *total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        return 0;
    }
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    HIPCHK(hipEventRecord(e0, g->stream));
    for (int i = 0; i < iters; i++)
        if (tamd_graph_launch(g)) return -1;
    HIPCHK(hipEventRecord(e1, g->stream));
    HIPCHK(hipEventSynchronize(e1));
    HIPCHK(hipEventElapsedTime(total_ms, e0, e1));
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return 0;
}

int tamd_graph_kernel_num(const tamd_graph* g) { return (int)g->steps.size(); }

int tamd_graph_profile(tamd_graph* g, int iters, tamd_kernel_info* out, int max_out)
{
    TAMD_ONE_THREAD(g);
    if (bind_device(g)) return -1;
    if (!g->prepared) { set_error("graph not prepared"); return -1; }
    if (!g->inflight.empty()) { set_error("tamd_graph_profile while asynchronous runs are in flight: collect them with tamd_graph_wait first"); return -1; }
    if (direct_drain(g)) return -1;
    g->out_fresh_in = 0;                       // the launches below (and the closing pass) write the staging buffers themselves
    int n = std::min((int)g->steps.size(), max_out);
    std::vector<hipEvent_t> ev(2 * g->steps.size());
    for (auto& e : ev) HIPCHK(hipEventCreate(&e));
    std::vector<double> acc(g->steps.size(), 0.0);
    // Each launch is repeated `iters` times back to back between ONE event pair (launches are idempotent:
    // same inputs, same outputs), so the ~2 us cost of the event records themselves is amortised away and
    // the figure is the in-order stream's per-launch duration, as rocprofv3 --kernel-trace reports it.
    for (size_t i = 0; i < g->steps.size(); i++) {
        HIPCHK(hipEventRecord(ev[2 * i], g->stream));
        for (int it = 0; it < iters; it++) {
            hipError_t e = g->steps[i].fn(g->stream);
            if (e != hipSuccess) { set_error("profile launch failed: %s", hipGetErrorString(e)); return -1; }
        }
        HIPCHK(hipEventRecord(ev[2 * i + 1], g->stream));
    }
    HIPCHK(hipStreamSynchronize(g->stream));
    for (size_t i = 0; i < g->steps.size(); i++) {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]));
        acc[i] = (double)ms;              // total of `iters` launches; divided by iters below
    }
    // leave every tensor holding the result of ONE forward pass again
    if (run_steps(g, g->stream)) return -1;
    HIPCHK(hipStreamSynchronize(g->stream));
    for (auto& e : ev) hipEventDestroy(e);
    for (int i = 0; i < n; i++) {
        memset(&out[i], 0, sizeof(out[i]));
        snprintf(out[i].node, sizeof(out[i].node), "%s", g->steps[i].node.c_str());
        snprintf(out[i].kernel, sizeof(out[i].kernel), "%s", g->steps[i].kernel.c_str());
        out[i].macs = g->steps[i].macs;
        out[i].bytes = g->steps[i].bytes;
        out[i].ms = (float)(acc[i] / iters);
    }
    return n;
}

int tamd_graph_tensor_num(const tamd_graph* g) { return (int)g->tensors.size(); }

int tamd_graph_tensor_desc(const tamd_graph* g, int idx, int* dims8, int* dtype)
{
    if (idx < 0 || idx >= (int)g->tensors.size()) return -1;
    return fill_desc(g->tensors[idx], dims8, dtype);
}

int tamd_graph_read_tensor(tamd_graph* g, int idx, void* host, size_t bytes)
{
    if (!g) { set_error("null graph"); return -1; }
    TAMD_ONE_THREAD(g);
    if (idx < 0 || idx >= (int)g->tensors.size() || !g->prepared) return -1;
    if (bind_device(g)) return -1;
    HTensor& t = g->tensors[idx];
    if (t.ttype == TAMD_TT_CONST) { memcpy(host, t.data.data(), std::min(bytes, t.data.size())); return 0; }
    if ((size_t)idx < g->fused_away.size() && g->fused_away[idx]) {
        set_error("read_tensor: %s was fused into its consumer's launch and never reaches memory (TAMD_FUSE_PWDW=0 / "
                  "TAMD_FUSE_ELTWISE=0 / TAMD_FUSE_RELU=0 at prerun materialise it)", t.name.c_str());
        return -1;
    }
    if ((size_t)idx < g->pooled.size() && g->pooled[idx]) {
        set_error("read_tensor: %s shares its device memory with tensors of other lifetimes and does not survive the pass "
                  "(tamd_options.keep_tensors = 1 / TAMD_POOL=0 at prerun gives every tensor its own buffer)", t.name.c_str());
        return -1;
    }
    size_t need = t.elems() * esize(t.dtype);
    if (bytes != need) { set_error("read_tensor: %zu bytes given, %zu needed", bytes, need); return -1; }
    if (g->out_fresh_in)                       // an output whose staging buffer is the tensor itself (1x1 map): see stage_from_pinned
        for (auto& io : g->outputs)
            if (io.tensor == idx && io.stage == t.dptr && stage_from_pinned(g)) return -1;
    if (direct_drain(g)) return -1;
    HIPCHK(hipStreamSynchronize(g->stream));
    if (t.nchw_raw && t.is_view) {      // NCHW channel slice of a concat buffer (uint8 / fp32 planners): one row per image
        const size_t es = esize(t.dtype), img = (size_t)t.c * t.h * t.w * es;
        HIPCHK(hipMemcpy2DAsync(host, img, (const char*)t.dptr + (size_t)t.c_off * t.h * t.w * es, (size_t)t.cs * t.h * t.w * es, img,
                                (size_t)t.n, hipMemcpyDeviceToHost, g->stream));
        HIPCHK(hipStreamSynchronize(g->stream));
        return 0;
    }
    if (t.nchw_raw) {
        HIPCHK(hipMemcpyAsync(host, t.dptr, need, hipMemcpyDeviceToHost, g->stream));
        HIPCHK(hipStreamSynchronize(g->stream));
        return 0;
    }
    void* tmp = nullptr;
    HIPCHK(hipMalloc(&tmp, need));
    LayoutArgs a{(const int8_t*)t.dptr + t.c_off, tmp, t.n, t.c, t.h, t.w, t.cs, esize(t.dtype)};
    hipError_t e = launch_nhwc_to_nchw(a, g->stream);
    if (e != hipSuccess) { hipFree(tmp); set_error("layout launch failed"); return -1; }
    HIPCHK(hipMemcpyAsync(host, tmp, need, hipMemcpyDeviceToHost, g->stream));
    HIPCHK(hipStreamSynchronize(g->stream));
    { std::lock_guard<std::mutex> lk(g_capture_mutex); hipFree(tmp); }
    return 0;
}

void tamd_graph_destroy(tamd_graph* g)
{
    if (!g) return;
    if (g->prepared) (void)bind_device(g);
    if (g->prepared && g->opt.profile) dump_profile(g);
    if (g->h2h_runs > 0)
        fprintf(stderr, "[tamd] blocking run, host side, mean of %lld runs (us): copy in %.2f | stream drain %.2f | submit %.2f | wait %.2f | copy out %.2f\n", g->h2h_runs,
                1e-3 * g->h2h_ns[0] / g->h2h_runs, 1e-3 * g->h2h_ns[1] / g->h2h_runs, 1e-3 * g->h2h_ns[2] / g->h2h_runs, 1e-3 * g->h2h_ns[3] / g->h2h_runs,
                1e-3 * g->h2h_ns[4] / g->h2h_runs);
    // runs submitted and never collected are still device work: direct_destroy below waits for every closed burst before the
    // queue, the kernel arguments and the tensors go away
    g->inflight.clear();
    if (g->stream) hipStreamSynchronize(g->stream);
    std::lock_guard<std::mutex> lk(g_capture_mutex);      // hipFree is device-synchronous: not while another thread captures
    if (g->direct_io2) { direct_destroy(g->direct_io2); g->direct_io2 = nullptr; }
    if (g->direct_io) { direct_destroy(g->direct_io); g->direct_io = nullptr; }
    if (g->direct) { direct_destroy(g->direct); g->direct = nullptr; }
    for (int i = 0; i < g->nexec; i++) if (g->hexecs[i]) hipGraphExecDestroy(g->hexecs[i]);
    for (int slot = 0; slot < 2; slot++) {
        for (int i = 0; i < 2; i++) if (g->hexec_io[slot][i]) hipGraphExecDestroy(g->hexec_io[slot][i]);
        if (g->hgraph_io[slot]) hipGraphDestroy(g->hgraph_io[slot]);
    }
    if (g->hgraph) hipGraphDestroy(g->hgraph);
    for (void* p : g->dev_allocs) hipFree(p);
    for (auto* v : {&g->inputs, &g->outputs})
        for (auto& io : *v) { if (io.pinned) hipHostFree(io.pinned); if (io.pinned2) hipHostFree(io.pinned2); }
    for (auto& e : g->slot_done) if (e) hipEventDestroy(e);
    if (g->stream) hipStreamDestroy(g->stream);
    delete g;
}

}  // extern "C"
