// The C ABI of include/tengine_amd.h that builds, describes, pre-runs, profiles, reads and destroys a device graph.
//
// Mirrors what a Tengine device backend does between pre_run and run (CUDA pattern:
// source/device/cuda/cuda_executor.cc:136-204), re-designed for MI355X:
//   prerun : infer shapes (graph_infer.hip, restating source/operator/prototype/*.c), choose NHWC device layouts,
//            repack weights once (the role of conv_hcl_prerun, conv_kernel_x86.c:2137-2209), compile
//            the node list into a launch list (graph_plan*.hip, graph_u8.hip, graph_f32.hip), capture it into ONE hipGraph
//            and record it as AQL packets for direct dispatch (direct.cc);
//   run    : graph_exec.hip -- inputs -> one pass over the launch list -> outputs, pinned bounce buffers.
// The units of the graph layer are listed in graph_internal.h (one translation unit until round 5).
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

using namespace tamd;

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
const char* tamd_last_error(void) { return last_error(); }
const char* tamd_version(void) { return "tengine_amd 0.4 (gfx950)"; }

int tamd_op_supported(int op, int dtype)
{
    if (dtype != TAMD_DT_INT8 && dtype != TAMD_DT_UINT8 && dtype != TAMD_DT_FP32) return 0;
    if (op == TAMD_OP_UPSAMPLE) return dtype != TAMD_DT_INT8;      // nearest upsample: uint8 / fp32 graphs
    if (op == TAMD_OP_RELU6) return dtype == TAMD_DT_FP32;
    if (op == TAMD_OP_SOFTMAX) return 1;                           // (which axes: tamd_node_supported)
    if (op == TAMD_OP_RESHAPE || op == TAMD_OP_PRIORBOX) return 1; // dense device tensors: uint8 / fp32 graphs; int8 since round 6 (graph_plan.hip)
    if (op == TAMD_OP_PERMUTE) return dtype != TAMD_DT_FP32;       // SSD heads (Permute -> Flatten -> Concat): uint8, and int8 since round 6
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
        if (dt == TAMD_DT_INT8) {
            // NHWC device tensors: the channel concat; any other axis only for what is dense on the device anyway -- the PriorBox layout
            // [1][2][K][1] (priorbox.c:33-75: mbox_priorbox is a Concat on axis 2) and tensors of other ranks (Reshape / Flatten results)
            if (ax == 1) return out[0].dim_num >= 2;
            if (ax < 0 || ax >= out[0].dim_num) return 0;
            return out[0].dim_num != 4 || (out[0].dims[0] == 1 && out[0].dims[1] == 2 && out[0].dims[3] == 1);
        }
        return ax >= 0 && ax < out[0].dim_num;                               // dense NCHW: any axis
    }
    case TAMD_OP_SOFTMAX: {
        if (dt != TAMD_DT_INT8) return 1;
        // int8 tensors are NHWC on the device: the channel axis of a 2-D / 4-D tensor is the contiguous one; since round 6 the spatial axes
        // of a 4-D tensor and any axis >= 1 of a 3-D tensor (a Reshape result: dense on the device) run through the same kernel, strided
        if (n_in < 1 || in[0].dim_num < 2 || in[0].dim_num > 4 || in[0].ttype == TAMD_TT_CONST) return 0;
        int ax = n->param ? ((const tamd_softmax_param*)n->param)->axis : 1;
        if (ax < 0) ax += in[0].dim_num;
        if (ax < 1 || ax >= in[0].dim_num) return 0;
        if (in[0].dim_num == 2 && ax != 1) return 0;
        return in[0].dims[ax] >= 1 && in[0].dims[ax] <= kSoftmaxI8MaxC;
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
    // a batched graph of batch-wise independent operators: two half-batch device graphs side by side (graph_pair.hip)
    if (const int pr = pair_try_prerun(g, opt)) return pr < 0 ? -1 : 0;
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
    if (any_u8 ? plan_u8(g) : any_f32 ? plan_f32(g) : plan_i8(g)) return -1;
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
                fprintf(stderr, "tengine_amd: direct dispatch DISABLED for this graph: %s (hipGraph replay instead)\n", last_error());
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
                    if (zc) fprintf(stderr, "tengine_amd: zero-copy outputs DISABLED for this graph: %s\n", last_error());
                    direct_destroy(pio);
                    pio = nullptr; zc_ok = false; why = last_error();
                    if (zc) {                                // once more with the download launches
                        pio = direct_build(g->gpu, g->stream, with_downloads, &why, g->direct);
                        if (pio && direct_io_selfcheck(g, pio, slot)) { direct_destroy(pio); pio = nullptr; why = last_error(); }
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

int tamd_graph_halves(const tamd_graph* g) { return g && g->half[0] ? 2 : 0; }
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
    if (g->half[0]) return pair_set_input(g, idx, host, bytes);
    g->inputs[idx].host_in = host;
    return 0;
}

int tamd_graph_set_output(tamd_graph* g, int idx, void* host, size_t bytes)
{
    if (!g) { set_error("null graph"); return -1; }
    TAMD_ONE_THREAD(g);
    if (idx < 0 || idx >= (int)g->outputs.size()) { set_error("bad output index"); return -1; }
    if (g->prepared && bytes != g->outputs[idx].bytes) { set_error("output %d: %zu bytes given, %zu expected", idx, bytes, g->outputs[idx].bytes); return -1; }
    if (g->half[0]) return pair_set_output(g, idx, host, bytes);
    g->outputs[idx].host_out = host;
    return 0;
}


int tamd_graph_kernel_num(const tamd_graph* g) { return g->half[0] ? pair_kernel_num(g) : (int)g->steps.size(); }

int tamd_graph_profile(tamd_graph* g, int iters, tamd_kernel_info* out, int max_out)
{
    TAMD_ONE_THREAD(g);
    if (bind_device(g)) return -1;
    if (!g->prepared) { set_error("graph not prepared"); return -1; }
    if (g->half[0]) return pair_profile(g, iters, out, max_out);
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
    if (g->half[0]) return pair_read_tensor(g, idx, host, bytes);
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
    if (g->half[0]) { pair_destroy(g); return; }          // (the halves print their own tables, release their own queues and memory)
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

