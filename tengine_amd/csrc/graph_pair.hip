// A batched graph as TWO half-batch device graphs behind one tamd_graph (tamd_options.split_batch, round 6).
//
// Graphs of this library are independent objects with their own HSA queue.  A batch submitted as two graphs of half the batch lets the
// launch boundaries, ramps and tile tails of one half overlap the other half's work: a launch of ResNet-50's res4 stage lives 5.5 us
// inside its blocks and costs 9.1 us end to end (profiles/r05_pgemm_anatomy_v3_final_forms.txt).  Measured with two graph objects
// driven by the host (profiles/r06_split_batch_direct.txt, bench.py's former side figure): ResNet-50 b32 +8.2 %, MobileNet-v1 b64
// +6.5 %, outputs identical.  Here the pair lives behind the C ABI: the parent keeps the IR (descriptions, constants), the two halves
// are ordinary graphs of B / 2 images (own stream, HSA queue, arena, packed weights, host-to-host lists), and every entry point
// forwards -- host buffers as two contiguous halves (dimension 0 is the batch in the reference's NCHW order: tensor.c / the tmfile).
// The reference runs a batch as one loop over images inside every operator (conv_kernel_x86.c:2241-2263 `for (int i = 0; i < batch
// ...)`), so images never interact in the operators admitted here and the halves' bytes are the one graph's bytes.
#include "graph.h"
#include "graph_internal.h"
#include "env.h"

#include <stddef.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <mutex>

namespace tamd {

static tamd_graph* clone_ir(const tamd_graph* g)
{
    tamd_graph* c = new tamd_graph();
    c->tensors = g->tensors;
    c->nodes = g->nodes;
    for (auto& io : g->inputs) { IOBind b; b.tensor = io.tensor; c->inputs.push_back(b); }
    for (auto& io : g->outputs) { IOBind b; b.tensor = io.tensor; c->outputs.push_back(b); }
    c->is_half = true;
    return c;
}

// operators that treat the images of a batch independently (the plugin's list, hip_device.cc: split_wanted)
static bool ops_allow_split(const tamd_graph* g)
{
    for (auto& n : g->nodes)
        switch (n.op) {
        case TAMD_OP_INPUT: case TAMD_OP_CONST: case TAMD_OP_CONV: case TAMD_OP_FC: case TAMD_OP_POOL: case TAMD_OP_RELU:
        case TAMD_OP_ELTWISE: case TAMD_OP_DROPOUT:
            break;
        case TAMD_OP_CONCAT: if (n.p.concat.axis < 1) return false; break;
        case TAMD_OP_SOFTMAX: if (n.p.softmax.axis < 1) return false; break;
        default: return false;
        }
    return true;
}

// every activation tensor that a node touches carries `batch` as dimension 0 (the halves of a host buffer are then contiguous)
static bool carries_batch(const tamd_graph* g, int batch)
{
    for (auto& n : g->nodes) {
        if (n.op == TAMD_OP_INPUT || n.op == TAMD_OP_CONST) continue;
        for (auto* v : {&n.in, &n.out})
            for (int ti : *v) {
                const HTensor& t = g->tensors[ti];
                if (t.ttype == TAMD_TT_CONST) continue;
                if (t.dims.size() < 2 || t.dims[0] != batch) return false;
            }
    }
    for (auto* v : {&g->inputs, &g->outputs})
        for (auto& io : *v) {
            const HTensor& t = g->tensors[io.tensor];
            if (t.ttype == TAMD_TT_CONST || t.dims.size() < 2 || t.dims[0] != batch) return false;
        }
    return true;
}

// 0: the graph stays one launch list (the caller pre-runs it as before); 1: it is a pair now, prepared; < 0: error
int pair_try_prerun(tamd_graph* g, const tamd_options* opt)
{
    if (g->is_half || g->inputs.empty() || g->outputs.empty()) return 0;
    // tamd_options.split_batch: 0 default rule, 1 never, 2 wherever possible; TAMD_SPLIT_BATCH: 0 never, 1 default rule, 2 wherever possible
    int mode = 0, direct = 0;
    if (opt && opt->size >= (int)(offsetof(tamd_options, split_batch) + sizeof(int))) mode = opt->split_batch;
    if (opt && opt->size >= (int)(offsetof(tamd_options, direct_dispatch) + sizeof(int))) direct = opt->direct_dispatch;
    if (mode != 1 && mode != 2) mode = 0;      // (a caller compiled against round 5's struct of the same padded size passes padding here)
    if (mode == 1) return 0;                   // a caller that splits batches by itself (the plugin): final, the switch below is then ITS switch
    if (const char* e = getenv("TAMD_SPLIT_BATCH")) { const int v = atoi(e); mode = v == 0 ? 1 : v == 2 ? 2 : 0; }
    if (const char* dd = getenv("TAMD_DIRECT_DISPATCH")) direct = atoi(dd) != 0;
    if (mode == 1) return 0;
    for (auto& io : g->inputs)
        if (io.tensor < 0 || io.tensor >= (int)g->tensors.size() || g->tensors[io.tensor].dims.empty()) return 0;
    const int B = g->tensors[g->inputs[0].tensor].dims[0];
    if (B < 2 || B % 2 || !ops_allow_split(g)) return 0;
    if (mode != 2) {
        // the default rule: where it was measured to pay -- int8 launch lists dispatched directly, from batch 16 on (device-resident:
        // ResNet-50 b32 +8.2 %, MobileNet-v1 b64 +6.5 %, b16 / b32 halves of those; the byte-exact uint8 configs gain 0.6-1.7 %, their
        // fp32-MFMA launches are long enough to hide their own boundaries: profiles/r06_split_batch_direct.txt)
        if (!direct || B < 16) return 0;
        for (auto& t : g->tensors)
            if (t.ttype != TAMD_TT_CONST && t.dtype != TAMD_DT_INT8) return 0;
    }
    // shapes first, on a clone: anything that does not carry the batch in front (a 1-D tensor, a Concat of constants) keeps the graph in
    // one piece -- not an error
    {
        tamd_graph* probe = clone_ir(g);
        for (auto& io : probe->inputs) probe->tensors[io.tensor].dims[0] = B / 2;
        const bool ok = infer_shapes(probe) == 0 && carries_batch(probe, B / 2);
        tamd_graph_destroy(probe);
        if (!ok) return 0;
    }
    const auto t0 = std::chrono::steady_clock::now();
    tamd_graph* h[2] = {nullptr, nullptr};
    auto drop = [&]() { for (auto*& c : h) if (c) { tamd_graph_destroy(c); c = nullptr; } };
    for (int k = 0; k < 2; k++) {
        h[k] = clone_ir(g);
        h[k]->formula_batch = B;               // batch-dependent reference formulas follow the WHOLE batch (graph.h)
        for (auto& io : h[k]->inputs) h[k]->tensors[io.tensor].dims[0] = B / 2;
        // anything the halves cannot do: the whole batch as one launch list, as before (its own prerun reports what is wrong, if anything is)
        if (tamd_graph_prerun(h[k], opt)) { drop(); return 0; }
    }
    // the parent describes the whole batch: its own shapes (it is never planned)
    if (infer_shapes(g) || validate_graph(g)) { drop(); return -1; }
    if (!carries_batch(g, B)) { drop(); set_error("split_batch: the full-batch graph does not infer like its halves"); return -1; }
    g->opt = h[0]->opt;
    g->gpu = h[0]->gpu;
    for (auto* v : {&g->inputs, &g->outputs})
        for (auto& io : *v) io.bytes = g->tensors[io.tensor].elems() * esize(g->tensors[io.tensor].dtype);
    for (size_t i = 0; i < g->inputs.size(); i++)
        if (g->inputs[i].bytes != 2 * h[0]->inputs[i].bytes) { drop(); set_error("split_batch: input %zu of a half is not half the input", i); return -1; }
    for (size_t i = 0; i < g->outputs.size(); i++)
        if (g->outputs[i].bytes != 2 * h[0]->outputs[i].bytes) { drop(); set_error("split_batch: output %zu of a half is not half the output", i); return -1; }
    g->half[0] = h[0]; g->half[1] = h[1];
    g->pair_out.assign(g->outputs.size(), nullptr);
    // buffers bound before prerun (the plugin and tm_benchmark bind after it; the ABI allows both)
    for (size_t i = 0; i < g->inputs.size(); i++)
        if (g->inputs[i].host_in && pair_set_input(g, (int)i, g->inputs[i].host_in, g->inputs[i].bytes)) return -1;
    for (size_t i = 0; i < g->outputs.size(); i++)
        if (g->outputs[i].host_out && pair_set_output(g, (int)i, g->outputs[i].host_out, g->outputs[i].bytes)) return -1;
    g->prerun_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    g->prepared = true;
    if (getenv("TAMD_DEBUG"))
        fprintf(stderr, "[tamd] batch %d as two device graphs of %d images side by side (%zu + %zu launches per pass%s)\n", B, B / 2, h[0]->steps.size(),
                h[1]->steps.size(), h[0]->direct && h[1]->direct ? ", direct dispatch on two HSA queues" : "");
    return 1;
}

int pair_set_input(tamd_graph* g, int idx, const void* host, size_t bytes)
{
    g->inputs[idx].host_in = host;
    for (int k = 0; k < 2; k++)
        if (tamd_graph_set_input(g->half[k], idx, host ? (const char*)host + k * (bytes / 2) : nullptr, bytes / 2)) return -1;
    return 0;
}

int pair_set_output(tamd_graph* g, int idx, void* host, size_t bytes)
{
    g->outputs[idx].host_out = host;
    for (int k = 0; k < 2; k++)
        if (tamd_graph_set_output(g->half[k], idx, host ? (char*)host + k * (bytes / 2) : nullptr, bytes / 2)) return -1;
    return 0;
}

int pair_both(tamd_graph* g, int (*fn)(tamd_graph*))
{
    for (int k = 0; k < 2; k++)
        if (fn(g->half[k])) return -1;
    return 0;
}

// the blocking host-to-host run: both halves submitted (each its own burst: upload launch, launch list, outputs into its pinned slot),
// then both collected -- the second half's staging copy and upload overlap the first half's compute
int pair_run(tamd_graph* g)
{
    if (tamd_graph_inflight(g->half[0]) || tamd_graph_inflight(g->half[1])) { set_error("tamd_graph_run while asynchronous runs are in flight: collect them with tamd_graph_wait first"); return -1; }
    if (tamd_graph_run_async(g->half[0])) return -1;
    if (tamd_graph_run_async(g->half[1])) { (void)tamd_graph_wait(g->half[0]); return -1; }
    const int r0 = tamd_graph_wait(g->half[0]), r1 = tamd_graph_wait(g->half[1]);
    return (r0 || r1) ? -1 : 0;
}

int pair_run_async(tamd_graph* g)
{
    if (tamd_graph_inflight(g->half[0]) >= 2) { set_error("two runs are already in flight: call tamd_graph_wait first"); return -1; }
    if (tamd_graph_run_async(g->half[0])) return -1;
    if (tamd_graph_run_async(g->half[1])) { (void)tamd_graph_wait(g->half[0]); return -1; }
    return 0;
}

int pair_wait(tamd_graph* g)
{
    if (tamd_graph_inflight(g->half[0]) == 0) { set_error("tamd_graph_wait: no run in flight"); return -1; }
    const int r0 = tamd_graph_wait(g->half[0]), r1 = tamd_graph_wait(g->half[1]);
    return (r0 || r1) ? -1 : 0;
}

int pair_direct_packets(const tamd_graph* g, bool meta)
{
    return meta ? tamd_graph_direct_meta_packets(g->half[0]) + tamd_graph_direct_meta_packets(g->half[1])
                : tamd_graph_direct_packets(g->half[0]) + tamd_graph_direct_packets(g->half[1]);
}

const char* pair_direct_packet_name(const tamd_graph* g, int i)
{
    const int n0 = tamd_graph_direct_packets(g->half[0]);
    return i < n0 ? tamd_graph_direct_packet_name(g->half[0], i) : tamd_graph_direct_packet_name(g->half[1], i - n0);
}

// the stamped passes of the first half, then of the second (each half alone on the device: per-packet durations, not the overlap)
int pair_direct_timestamps(tamd_graph* g, int passes, double* dur_us, double* gap_us, int max_packets)
{
    const int n0 = tamd_graph_direct_packets(g->half[0]), n1 = tamd_graph_direct_packets(g->half[1]);
    if (n0 <= 0 || n1 <= 0) { set_error("tamd_graph_direct_timestamps: the graph does not dispatch directly (tamd_options.direct_dispatch)"); return -1; }
    if (max_packets < n0 + n1) { set_error("tamd_graph_direct_timestamps: %d packets, room for %d", n0 + n1, max_packets); return -1; }
    if (tamd_graph_direct_timestamps(g->half[0], passes, dur_us, gap_us, n0) < 0) return -1;
    if (tamd_graph_direct_timestamps(g->half[1], passes, dur_us + n0, gap_us + n0, n1) < 0) return -1;
    return n0 + n1;
}

// the device copy of an output: the two halves' staging buffers gathered into one buffer of the pair (refreshed at every call; the
// halves' own buffers stay where their launch lists write them)
int pair_output_device(tamd_graph* g, int idx, void** dptr, size_t* bytes)
{
    if (bind_device(g)) return -1;
    const size_t nb = g->outputs[idx].bytes;
    if (!g->pair_out[idx]) HIPCHK(hipMalloc(&g->pair_out[idx], nb > 0 ? nb : 16));
    for (int k = 0; k < 2; k++) {
        void* p = nullptr;
        size_t b = 0;
        if (tamd_graph_sync(g->half[k]) || tamd_graph_output_device(g->half[k], idx, &p, &b)) return -1;
        if (b != nb / 2) { set_error("output_device: a half holds %zu bytes, %zu expected", b, nb / 2); return -1; }
        HIPCHK(hipMemcpy((char*)g->pair_out[idx] + k * b, p, b, hipMemcpyDeviceToDevice));
    }
    *dptr = g->pair_out[idx];
    *bytes = nb;
    return 0;
}

// `iters` passes of both halves, the host's clock around first submit .. last completion (what a caller of launch() + sync() sees)
int pair_time_launches(tamd_graph* g, int iters, float* total_ms)
{
    if (pair_both(g, tamd_graph_sync)) return -1;
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < iters; i++)
        if (pair_both(g, tamd_graph_launch)) return -1;
    if (pair_both(g, tamd_graph_sync)) return -1;
    *total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return 0;
}

int pair_kernel_num(const tamd_graph* g) { return tamd_graph_kernel_num(g->half[0]) + tamd_graph_kernel_num(g->half[1]); }

// per-launch figures of the first half's list, then of the second's (every launch timed alone, as for one graph)
int pair_profile(tamd_graph* g, int iters, tamd_kernel_info* out, int max_out)
{
    const int n0 = tamd_graph_profile(g->half[0], iters, out, max_out);
    if (n0 < 0) return -1;
    const int n1 = max_out > n0 ? tamd_graph_profile(g->half[1], iters, out + n0, max_out - n0) : 0;
    return n1 < 0 ? -1 : n0 + n1;
}

int pair_read_tensor(tamd_graph* g, int idx, void* host, size_t bytes)
{
    const HTensor& t = g->tensors[idx];
    if (t.ttype == TAMD_TT_CONST) { memcpy(host, t.data.data(), std::min(bytes, t.data.size())); return 0; }
    const size_t need = t.elems() * esize(t.dtype);
    if (bytes != need) { set_error("read_tensor: %zu bytes given, %zu needed", bytes, need); return -1; }
    if (t.dims.empty() || t.dims[0] != 2 * g->half[0]->tensors[idx].dims[0]) { set_error("read_tensor: %s does not carry the batch", t.name.c_str()); return -1; }
    for (int k = 0; k < 2; k++)
        if (tamd_graph_read_tensor(g->half[k], idx, (char*)host + k * (bytes / 2), bytes / 2)) return -1;
    return 0;
}

void pair_destroy(tamd_graph* g)
{
    for (auto*& c : g->half) if (c) { tamd_graph_destroy(c); c = nullptr; }
    if (g->prepared) (void)bind_device(g);
    for (void* p : g->pair_out) if (p) { std::lock_guard<std::mutex> lk(g_capture_mutex); hipFree(p); }
    delete g;
}

}  // namespace tamd
