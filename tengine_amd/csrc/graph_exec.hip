// The executor: one pass over the launch list (eager / recorded for the direct path), the direct path's self-checks, the zero-copy
// host-to-host lists, and the run-side entry points of the C ABI (upload / launch / sync / download / run / run_async / wait).
// Split out of graph.hip in round 6.
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

namespace tamd {

int run_steps(tamd_graph* g, hipStream_t s, int io_slot)
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
int launch_io(tamd_graph* g, int slot)
{
    if (g->hexec_io[slot][0]) {
        hipGraphExec_t e = g->hexec_io[slot][g->next_io[slot]];
        g->next_io[slot] ^= 1;
        HIPCHK(hipGraphLaunch(e, g->stream));
        return 0;
    }
    return run_steps(g, g->stream, slot);
}


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

int direct_selfcheck(tamd_graph* g)
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
bool io_zero_copy_wanted()
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
bool zero_copy_outputs(tamd_graph* g, std::vector<LaunchRec>& recs, int slot)
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
bool zero_copy_inputs(tamd_graph* g, std::vector<LaunchRec>& recs, int slot)
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
int direct_io_selfcheck(tamd_graph* g, DirectProgram* pio, int slot)
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

int bind_device(tamd_graph* g)
{
    if (!g) { set_error("null graph"); return -1; }
    HIPCHK(hipSetDevice(g->gpu));
    return 0;
}

// passes submitted by direct dispatch are not on the HIP stream: everything that touches the tensors waits for them first
int direct_drain(tamd_graph* g)
{
    if (g->direct && g->direct_busy) {
        if (direct_wait(g->direct)) { set_error("direct dispatch: %s", direct_last_error()); return -1; }
        g->direct_busy = false;
    }
    return 0;
}

// the direct path of this graph is unusable (queue fault, a burst that never completed): forget the runs in flight -- their
// results are lost, the caller has been told -- and go back to the hipGraph executables, which every entry point still has
void direct_abandon(tamd_graph* g, const char* why)
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

// A zero-copy host-to-host run leaves its outputs in the pinned host buffers ONLY (the launch that would have written the device
// staging buffer was re-pointed): whoever reads the device copy next -- tamd_graph_output_device (the RCCL gather),
// tamd_graph_read_tensor of a 1x1-map output -- gets it refreshed from the pinned slot first.
int stage_from_pinned(tamd_graph* g)
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

}  // namespace tamd

using namespace tamd;

extern "C" {

int tamd_graph_upload_inputs(tamd_graph* g)
{
    TAMD_ONE_THREAD(g);
    if (bind_device(g)) return -1;
    if (g->half[0]) return pair_both(g, tamd_graph_upload_inputs);
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

int tamd_graph_launch(tamd_graph* g)
{
    TAMD_ONE_THREAD(g);
    if (bind_device(g)) return -1;
    if (!g->prepared) { set_error("graph not prepared"); return -1; }
    if (g->half[0]) return pair_both(g, tamd_graph_launch);
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

int tamd_graph_direct_packets(const tamd_graph* g) { return g && g->half[0] ? pair_direct_packets(g, false) : g && g->direct ? direct_packets(g->direct) : 0; }
int tamd_graph_direct_meta_packets(const tamd_graph* g) { return g && g->half[0] ? pair_direct_packets(g, true) : g && g->direct ? direct_meta_packets(g->direct) : 0; }
const char* tamd_graph_direct_packet_name(const tamd_graph* g, int i) { return g && g->half[0] ? pair_direct_packet_name(g, i) : g && g->direct ? direct_packet_name(g->direct, i) : ""; }

int tamd_graph_direct_timestamps(tamd_graph* g, int passes, double* dur_us, double* gap_us, int max_packets)
{
    if (!g) { set_error("null graph"); return -1; }
    TAMD_ONE_THREAD(g);
    if (bind_device(g)) return -1;
    if (g->half[0]) return pair_direct_timestamps(g, passes, dur_us, gap_us, max_packets);
    if (!g->prepared || !g->direct) { set_error("tamd_graph_direct_timestamps: the graph does not dispatch directly (tamd_options.direct_dispatch)"); return -1; }
    if (!g->inflight.empty()) { set_error("tamd_graph_direct_timestamps while asynchronous runs are in flight"); return -1; }
    if (max_packets < direct_packets(g->direct)) { set_error("tamd_graph_direct_timestamps: %d packets, room for %d", direct_packets(g->direct), max_packets); return -1; }
    if (direct_drain(g)) return -1;
    HIPCHK(hipStreamSynchronize(g->stream));
    g->stream_dirty = false;
    g->out_fresh_in = 0;                       // the passes write the staging buffers themselves
    const int n = direct_timestamps(g->direct, passes, dur_us, gap_us);
    if (n < 0) { set_error("direct dispatch: %s", direct_last_error()); return -1; }
    return n;
}
double tamd_graph_prerun_ms(const tamd_graph* g) { return g ? g->prerun_ms : 0.0; }

int tamd_graph_sync(tamd_graph* g)
{
    TAMD_ONE_THREAD(g);
    if (bind_device(g)) return -1;
    if (g->half[0]) return pair_both(g, tamd_graph_sync);
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
    if (g->half[0]) return pair_both(g, tamd_graph_download_outputs);
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
    if (g->half[0]) return pair_run(g);
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
            direct_abandon(g, last_error());
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
    if (g->half[0]) return pair_run_async(g);
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
            direct_abandon(g, last_error());
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
    if (g && g->half[0]) return pair_wait(g);
    if (!g || g->inflight.empty()) { set_error("tamd_graph_wait: no run in flight"); return -1; }
    if (bind_device(g)) return -1;
    const Inflight f = g->inflight.front();
    if (f.direct) {
        if (direct_wait_burst(g->direct_io, f.burst)) {
            // the run is lost; so is everything queued behind it.  Drop the bookkeeping (the graph would otherwise refuse every
            // entry point with "runs in flight" until it is destroyed) and leave the direct path
            set_error("direct dispatch: %s", direct_last_error());
            direct_abandon(g, last_error());
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

int tamd_graph_inflight(const tamd_graph* g) { return !g ? 0 : g->half[0] ? tamd_graph_inflight(g->half[0]) : (int)g->inflight.size(); }

int tamd_graph_output_device(tamd_graph* g, int idx, void** dptr, size_t* bytes)
{
    if (!g) { set_error("null graph"); return -1; }
    TAMD_ONE_THREAD(g);
    if (idx < 0 || idx >= (int)g->outputs.size() || !g->prepared) return -1;
    if (g->half[0]) return pair_output_device(g, idx, dptr, bytes);
    if (g->out_fresh_in && (bind_device(g) || stage_from_pinned(g))) return -1;
    *dptr = g->outputs[idx].stage;
    *bytes = g->outputs[idx].bytes;
    return 0;
}

// once the caller holds the stream it may queue work there that this library cannot see: every direct burst drains it first again
void* tamd_graph_stream(tamd_graph* g)
{
    if (g->half[0]) { (void)tamd_graph_stream(g->half[1]); return tamd_graph_stream(g->half[0]); }   // (a pair: the first half's stream; both halves drain theirs from now on)
    g->stream_exposed = true; g->stream_dirty = true;
    return (void*)g->stream;
}

int tamd_graph_time_launches(tamd_graph* g, int iters, float* total_ms)
{
    TAMD_ONE_THREAD(g);
    if (bind_device(g)) return -1;
    if (g->half[0]) return pair_time_launches(g, iters, total_ms);
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

}  // extern "C"
