// Host-side graph IR, planner and executor of the backend (private).
#pragma once
#include <hip/hip_runtime.h>

#include <functional>
#include <atomic>
#include <map>
#include <string>
#include <vector>

#include "../../include/tengine_amd.h"
#include "kernels.h"

namespace tamd {

void set_error(const char* fmt, ...);

struct HTensor {               // host IR tensor == the parts of struct tensor the backend reads
    int dtype = 0, ttype = TAMD_TT_VAR;
    std::vector<int> dims;     // NCHW etc.
    std::vector<uint8_t> data; // const payload (copied)
    std::vector<float> scales;
    std::vector<int> zps;
    std::string name;
    // device view (valid after prerun for var/input tensors)
    void* dptr = nullptr;      // base of the NHWC buffer this tensor lives in
    int n = 0, h = 0, w = 0, c = 0;
    int cs = 0;                // channel stride of the *buffer* (bytes per pixel for int8)
    int c_off = 0;             // channel offset inside the buffer pixel (concat-by-offset views)
    bool is_view = false;      // aliases another tensor's buffer
    bool nchw_raw = false;     // graph input kept in NCHW for a direct first conv
    bool prerun_const = false; // written once at prerun (PriorBox and what only depends on it), never by a run
    size_t elems() const { size_t e = 1; for (int d : dims) e *= (size_t)d; return e; }
};

union NodeParam {
    tamd_conv_param conv;
    tamd_fc_param fc;
    tamd_pool_param pool;
    tamd_relu_param relu;
    tamd_eltwise_param elt;
    tamd_concat_param concat;
    tamd_upsample_param ups;
    tamd_permute_param perm;
    tamd_softmax_param softmax;
    tamd_reshape_param reshape;
    tamd_priorbox_param priorbox;
};

struct HNode {
    int op = 0;
    std::vector<int> in, out;
    NodeParam p{};
    std::string name;
};

// what a step touches, for the dependency-aware barrier bits of the direct path: bytes [base, base + size) and, when period > 0,
// only the slice [off, off + len) of every `period` bytes (a concat input's slot, a channel range of an NCHW concat buffer)
struct Access {
    const char* base = nullptr;
    size_t size = 0, off = 0, len = 0, period = 0;
};
inline bool access_overlap(const Access& a, const Access& b)
{
    if (a.base + a.size <= b.base || b.base + b.size <= a.base) return false;
    if (a.base == b.base && a.size == b.size && a.period > 0 && a.period == b.period) return !(a.off + a.len <= b.off || b.off + b.len <= a.off);
    return true;
}

struct Step {                  // one device launch of the compiled plan
    std::string node, kernel;
    double macs = 0, bytes = 0;
    std::function<hipError_t(hipStream_t)> fn;
    bool once = false;         // every input is a prerun constant (PriorBox outputs): launched once at the end of prerun
    // rd / wr list EVERYTHING the step's launch reads / writes in device memory that another step may write (constants left
    // out) -- only then is deps set, and only a step with deps may run beside its predecessors (graph_exec.hip run_steps)
    bool deps = false;
    std::vector<Access> rd, wr;
};
inline bool step_conflict(const Step& a, const Step& b)          // RAW, WAR or WAW between two steps that both carry deps
{
    for (auto& w : a.wr) { for (auto& x : b.wr) if (access_overlap(w, x)) return true; for (auto& x : b.rd) if (access_overlap(w, x)) return true; }
    for (auto& r : a.rd) for (auto& x : b.wr) if (access_overlap(r, x)) return true;
    return false;
}
inline Access access_of(const HTensor& t)                         // dense tensor, or the channel slice of the concat buffer it lives in
{
    Access a;
    a.base = (const char*)t.dptr;
    const size_t esz = t.dtype == TAMD_DT_FP32 ? 4 : 1;
    if (t.nchw_raw || t.cs <= 0) {                                // NCHW bytes (uint8 / fp32 graphs, raw graph inputs)
        if (t.is_view && t.cs > 0) {
            a.period = (size_t)t.cs * t.h * t.w * esz; a.off = (size_t)t.c_off * t.h * t.w * esz; a.len = (size_t)t.c * t.h * t.w * esz;
            a.size = a.period * (size_t)t.n;
        } else {
            a.size = t.elems() * esz; a.len = a.size;
        }
    } else {                                                      // NHWC int8 buffer, cs bytes per pixel; a view owns channels [c_off, c_off + c)
        a.size = (size_t)t.n * t.h * t.w * t.cs;
        if (t.is_view) { a.period = (size_t)t.cs; a.off = (size_t)t.c_off; a.len = (size_t)t.c; }
        else a.len = a.size;
    }
    return a;
}

struct IOBind {
    int tensor = -1;
    const void* host_in = nullptr;   // re-read at every run
    void* host_out = nullptr;
    size_t bytes = 0;
    void* stage = nullptr;           // device buffer in the reference's NCHW order
    void* pinned = nullptr;          // pinned host bounce buffer (slot 0 of the asynchronous runs)
    void* pinned2 = nullptr;         // slot 1: tamd_graph_run_async keeps two runs in flight
};

struct Inflight {                    // one tamd_graph_run_async() that tamd_graph_wait() has not collected yet
    int slot = 0;
    std::vector<void*> host_out;     // where the caller wants this run's outputs (set_output at submit time)
    hipEvent_t done = nullptr;       // hipGraph path
    bool direct = false;             // direct path: the run is burst `burst` of the graph's HSA queue (direct.cc)
    unsigned long long burst = 0;
};

struct PoolGeom { int oh, ow, kh, kw, sh, sw, ph0, pw0; };

// device memory of a graph comes out of a few large allocations (bump-allocated, 256-byte granules): one hipMalloc per tensor /
// weight blob scatters a model over hundreds of separately mapped ranges, and inside a pass every launch then starts with
// translation misses on pages it last touched a whole step ago
struct DevArena { char* base = nullptr; size_t cap = 0, used = 0; };

}  // namespace tamd

#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            tamd::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return -1;                                                                            \
        }                                                                                         \
    } while (0)

struct tamd_graph {
    // "one graph = one thread at a time" (include/tengine_amd.h), enforced: the token of the thread inside a call that changes the graph
    // or touches its buffers (0: nobody).  A second thread's call fails with an error instead of racing (graph_internal.h: OneThread)
    std::atomic<unsigned long> owner{0};
    std::vector<tamd::HTensor> tensors;
    std::vector<tamd::HNode> nodes;
    std::vector<tamd::IOBind> inputs, outputs;
    std::vector<tamd::Step> steps;      // compute launches
    std::vector<tamd::Step> in_steps;   // input layout launches (after H2D)
    std::vector<tamd::Step> out_steps;  // output layout launches (before D2H)
    std::vector<void*> dev_allocs;
    std::vector<tamd::DevArena> arenas;       // dev_alloc(): bump allocation out of these (each is also in dev_allocs)
    std::vector<char> pooled;           // tensors whose buffer is shared with other tensors of disjoint lifetime (read_tensor refuses them)
    size_t pool_bytes = 0, unpooled_bytes = 0;     // activation arena with / without lifetime sharing (TAMD_DEBUG prints both)
    std::map<int, float*> f32_copy;     // uint8 tensor -> its dequantised fp32 copy (input of the fp32 MFMA conv kernel)
    std::vector<char> fused_away;       // tensors that only exist inside a fused launch (read_tensor refuses them)
    void* zero_page = nullptr;          // 256 zero bytes (out-of-image taps of the LDS-DMA conv kernel)
    hipStream_t stream = nullptr;
    hipGraph_t hgraph = nullptr;
    hipGraphExec_t hexec = nullptr;     // == hexecs[0]
    // several instances of the same captured graph, launched round-robin: back-to-back replays of ONE hipGraphExec_t
    // leave a ~9 us hole between them on the device (the next launch is not queued behind the running one); with
    // independent instances the next replay's packets are already in the queue (profiles/r02_*replay*)
    hipGraphExec_t hexecs[4] = {nullptr, nullptr, nullptr, nullptr};
    int nexec = 0, next_exec = 0;
    // host-to-host runs (tamd_graph_run / _run_async): the same launch list with the input upload in front and the output
    // download behind it as copy KERNELS on the device-mapped pinned buffers of I/O slot 0 | 1, two instances per slot
    hipGraph_t hgraph_io[2] = {nullptr, nullptr};
    hipGraphExec_t hexec_io[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    int next_io[2] = {0, 0};
    std::vector<tamd::Inflight> inflight;      // FIFO, at most 2
    hipEvent_t slot_done[2] = {nullptr, nullptr};
    int next_slot = 0;
    tamd::DirectProgram* direct = nullptr;     // tamd_options.direct_dispatch: the launch list as AQL packets (direct.cc)
    tamd::DirectProgram* direct_io = nullptr;  // .. the host-to-host list of I/O slot 0 (tamd_graph_run), same HSA queue
    tamd::DirectProgram* direct_io2 = nullptr; // .. of I/O slot 1 (the second asynchronous run in flight)
    int autotune_cold = -1;                    // plan-time timing mode, decided once (autotune_cold())
    double prerun_ms = 0;                      // wall time of tamd_graph_prerun (planning, autotune, capture)
    bool direct_busy = false;                  // passes submitted since the last wait
    bool stream_dirty = true;                  // something may be pending on `stream` (uploads, eager launches): drain it before a direct burst
    bool stream_exposed = false;               // tamd_graph_stream() handed the stream out: the caller may queue work this library cannot see
    bool io_zero_copy = false;                 // the host-to-host lists store graph outputs straight into the pinned host buffers
    bool io_zero_copy2 = false;                //   .. the list of I/O slot 1 does
    int out_fresh_in = 0;                      // 0: the outputs' device staging buffers hold the last pass; 1 / 2: only pinned slot 0 / 1 does
                                               // (a zero-copy host-to-host run): stage_from_pinned() before anything reads the device copy
    // TAMD_H2H_TRACE=1: where a blocking tamd_graph_run spends its time on the host (ns): copy in, stream drain, submit, wait, copy out
    long long h2h_ns[5] = {0, 0, 0, 0, 0};
    long long h2h_runs = 0;
    tamd_options opt{};
    bool prepared = false;
    int gpu = 0;
    // round 6: a batched graph of batch-wise independent operators is compiled as TWO device graphs of half the batch each
    // (graph_pair.hip): half[0] runs images [0, B / 2), half[1] the rest, side by side on their own HSA queues.  This object then
    // keeps the IR (descriptions, constants) and forwards every entry point; it owns no stream, launch list or device tensor.
    tamd_graph* half[2] = {nullptr, nullptr};
    bool is_half = false;                      // one of the two halves of such a pair: never split again
    // ... and the batch of the WHOLE graph it is a part of (0: its own).  The reference picks the formula of a 3x3 depthwise
    // convolution by batch == 1 (conv_dw_hcl_x86.c:508-543 score(): the hand-written kernel at batch 1, ref_conv_int8 otherwise --
    // another requantisation chain, graph_plan.hip: conv_mode): the halves of a 2-image graph run ONE image each and must still
    // produce the 2-image graph's bytes (found by tools/fuzz_split.py: 7 of 931 graphs differed before this field existed)
    int formula_batch = 0;
    std::vector<void*> pair_out;               // tamd_graph_output_device of a pair: the two halves gathered into one buffer per output
};

namespace tamd {

// planner helpers shared by graph_plan.hip (int8, NHWC) and graph_u8.hip (uint8, NCHW)
PoolGeom pool_geom(const tamd_pool_param& p, int h, int w);
int dev_alloc(tamd_graph* g, void** p, size_t bytes, bool zero);
// plan-time autotune: candidates of a graph whose pass moves far more bytes than the L2s hold are timed COLD -- every timed
// launch behind a fill of kL2FlushBytes (l2_flush_buffer(): one per device, kept for the life of the process) -- because that
// is how they run inside a pass; small graphs (batch-1 classifiers live in the L2s from step to step) keep back-to-back timing
constexpr size_t kL2FlushBytes = 64u << 20;
void* l2_flush_buffer();
bool autotune_cold(tamd_graph* g);
// TAMD_PLAN_CACHE=<file>: "<site>|<node>|<shape>" -> what the plan-time autotune chose (plan_cache.hip)
bool plan_cache_get(const std::string& key, std::string* v);
void plan_cache_put(const std::string& key, const std::string& v);
int time_cold(tamd_graph* g, void* flush, const std::function<hipError_t()>& launch, float* ms_out);      // graph_plan.hip
void nhwc_geom(HTensor& t);
int count_consumers(const tamd_graph* g, int tensor);
int priorbox_count(const tamd_priorbox_param& p);
void priorbox_eval(const tamd_priorbox_param& p, int feat_h, int feat_w, int data_h, int data_w, std::vector<float>* out);
void priorbox_quant_u8(const std::vector<float>& f, float scale, int zp, std::vector<uint8_t>* q);
void priorbox_quant_i8(const std::vector<float>& f, float scale, std::vector<int8_t>* q);
int plan_u8(tamd_graph* g);        // graph_u8.hip: every activation tensor is uint8
int plan_f32(tamd_graph* g);       // graph_f32.hip: every activation tensor is fp32

template <typename T>
int upload(tamd_graph* g, const std::vector<T>& host, T** dev)
{
    void* p = nullptr;
    if (dev_alloc(g, &p, host.size() * sizeof(T), false)) return -1;
    HIPCHK(hipMemcpyAsync(p, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice, g->stream));   // own stream only
    HIPCHK(hipStreamSynchronize(g->stream));
    *dev = (T*)p;
    return 0;
}

}  // namespace tamd
