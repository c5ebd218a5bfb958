// What the translation units of the graph layer share (private).  Since round 6 the layer is six units instead of one:
//   graph.hip             the C ABI that builds, describes, pre-runs, profiles, reads and destroys a graph
//   graph_infer.hip       shape inference, validation, the PriorBox evaluator, the error string
//   graph_plan.hip        the int8 planner (arena, requantisation folds, plan-time timing, conv / pool planners, plan_i8)
//   graph_plan_pairs.hip  .. its pair fusions (pwdw, dwpw); graph_plan.h is what those two share
//   plan_cache.hip        TAMD_PLAN_CACHE
//   graph_pair.hip        a batched graph as two half-batch graphs side by side behind one handle (tamd_options.split_batch)
//   graph_exec.hip        run_steps, the direct path's self-checks, zero-copy lists, the run-side entry points
// (graph_u8.hip and graph_f32.hip are the uint8 / fp32 planners, as before.)
#pragma once
#include <mutex>

#include "graph.h"
#include "launch_rec.h"

namespace tamd {

// prerun (plan + hipGraph capture) and the device-synchronous frees are serialised process-wide (graph_infer.hip)
extern std::mutex g_capture_mutex;
const char* last_error();                    // the calling thread's error string (set_error)

static inline int rup(int v, int m) { return (v + m - 1) / m * m; }
static inline int esize(int dt) { return (dt == TAMD_DT_FP32 || dt == TAMD_DT_INT32) ? 4 : (dt == TAMD_DT_FP16 ? 2 : 1); }
static inline int cdiv_c(int a, int b) { return a / b; }  // C semantics (truncation), as the reference

int infer_shapes(tamd_graph* g);             // graph_infer.hip
int validate_graph(tamd_graph* g);
int plan_i8(tamd_graph* g);                  // graph_plan.hip: every activation tensor is int8
void plan_cache_flush();                     // plan_cache.hip

// graph_exec.hip
int run_steps(tamd_graph* g, hipStream_t s, int io_slot = -1);
int launch_io(tamd_graph* g, int slot);
int direct_selfcheck(tamd_graph* g);
bool io_zero_copy_wanted();
bool zero_copy_outputs(tamd_graph* g, std::vector<LaunchRec>& recs, int slot);
bool zero_copy_inputs(tamd_graph* g, std::vector<LaunchRec>& recs, int slot);
int direct_io_selfcheck(tamd_graph* g, DirectProgram* pio, int slot);
int bind_device(tamd_graph* g);
int direct_drain(tamd_graph* g);
void direct_abandon(tamd_graph* g, const char* why);
int stage_from_pinned(tamd_graph* g);

// graph_pair.hip: a batched graph as two half-batch device graphs behind one tamd_graph (tamd_options.split_batch); every entry point
// of the C ABI forwards to these when g->half[0] is set
int pair_try_prerun(tamd_graph* g, const tamd_options* opt);     // 0: stays one launch list; 1: a pair now, prepared; < 0: error
int pair_set_input(tamd_graph* g, int idx, const void* host, size_t bytes);
int pair_set_output(tamd_graph* g, int idx, void* host, size_t bytes);
int pair_both(tamd_graph* g, int (*fn)(tamd_graph*));
int pair_run(tamd_graph* g);
int pair_run_async(tamd_graph* g);
int pair_wait(tamd_graph* g);
int pair_direct_packets(const tamd_graph* g, bool meta);
const char* pair_direct_packet_name(const tamd_graph* g, int i);
int pair_direct_timestamps(tamd_graph* g, int passes, double* dur_us, double* gap_us, int max_packets);
int pair_output_device(tamd_graph* g, int idx, void** dptr, size_t* bytes);
int pair_time_launches(tamd_graph* g, int iters, float* total_ms);
int pair_kernel_num(const tamd_graph* g);
int pair_profile(tamd_graph* g, int iters, tamd_kernel_info* out, int max_out);
int pair_read_tensor(tamd_graph* g, int idx, void* host, size_t bytes);
void pair_destroy(tamd_graph* g);

// One graph = one thread at a time: every entry point that changes the graph or touches its buffers holds the graph for the
// duration of the call (nested entry points of the SAME thread pass).  Calls from different threads one after the other are fine --
// Tengine's scheduler does that -- two at once are a caller's bug that used to show up as corrupted launch lists; now the second
// call fails with an error.
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

}  // namespace tamd

#define TAMD_ONE_THREAD(g_)                                                                                                          \
    tamd::OneThread one_thread_(g_);                                                                                                 \
    if (!one_thread_.ok) { tamd::set_error("this tamd_graph is inside a call on another thread: one graph = one thread at a time (include/tengine_amd.h)"); return -1; }
