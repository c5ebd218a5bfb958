// Direct dispatch: a device graph's launch list replayed as AQL packets on a private HSA queue.
//
// Why: at batch 1 the graph is fifteen dependent launches of 2-4 microseconds, and the boundary between two of them is what
// the step is made of.  Measured on the same chain of short kernels (tools/exp/aql_chain.cpp, profiles/r02_aql_chain.txt):
// hipGraph replay 1.6-1.75 us per boundary, eager HIP launches 1.47 us (host-bound below ~2.6 us of kernel), raw AQL packets
// with the barrier bit and agent-scope fences 1.26 us -- and no hole between replays, because the host only writes 64-byte
// packets into the ring and rings the doorbell once per pass; it is far ahead of the device.
//
// How: prerun records the launch list once (launch_rec.h: kernel handle, geometry, explicit argument bytes).  Each record is
// resolved to its kernel descriptor through the HSA loader's view of the code objects HIP has already loaded (kernel name from
// hipKernelNameRefByPtr -> "<name>.kd" symbol), its argument segment is completed with the code-object-v5 hidden arguments
// (block counts, group sizes, grid dimensionality at align8(explicit size) + 0 / 12 / 64), and all segments live in one device
// buffer written once: the arguments never change between passes (tensors are at fixed addresses).  A pass = one
// kernel-dispatch packet per record, every packet with the barrier bit (the list is a dependency chain) and agent-scope acquire /
// release -- what one kernel needs to see of the previous one across the XCDs' L2s.  System scope is paid once per BURST, not
// per pass: the first packet after the graph's stream was drained acquires at system scope (inputs uploaded by a copy engine),
// and direct_wait closes the burst with one barrier packet that releases at system scope (outputs read by the host or a
// copy engine) and carries the completion signal.  Launches of the COHERENT kernel instances (pwdw.hip: agent-scope loads of
// what other launches wrote, write-through stores of what other launches read) carry fence scope "none": nothing has to be
// written back or invalidated around them, which is 0.84 us per boundary instead of 1.26 (tools/exp/aql_chain.cpp) and leaves
// the read-only weights resident in the L2s across launches and passes.  Measured on MobileNet-v1 batch 1: 59.4 us per pass with system scope at
// both ends of every pass, 57.0 with it at the ends of the burst, 61.8 for the hipGraph replay (profiles/r02_direct_dispatch.txt).
//
// Not covered, by construction: stream ordering with the graph's HIP stream (tamd_graph_sync / download / run wait for the
// queue; graph_exec.hip drains the stream before the first packet of a burst is written).  Kernels with a scratch frame pass their
// private segment size in the packet; the runtime backs the queue's scratch on demand as it does for HIP's queues.
#include <hip/hip_runtime.h>
#include "env.h"
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <hsa/hsa_ven_amd_loader.h>
#include <cxxabi.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>

#include "codeobj_meta.h"
#include "launch_rec.h"

namespace tamd {

thread_local std::vector<LaunchRec>* g_launch_rec = nullptr;
thread_local bool g_launch_coherent = false;
thread_local bool g_launch_beside = false;

namespace {

struct KernelSym { uint64_t object; uint32_t kernarg, group, priv; };

struct DeviceCtx {
    bool tried = false, ok = false;
    hsa_agent_t agent{}, cpu{};
    bool have_cpu = false;
    std::map<std::string, KernelSym> syms;
    std::map<std::string, HiddenLayout> layouts;        // "<kernel>.kd" -> hidden-argument offsets from the code objects' metadata notes
    std::map<uint64_t, bool> seen_objects;              // storage base of the code objects already parsed
};
std::mutex g_mu;
std::map<int, DeviceCtx> g_ctx;
hsa_ven_amd_loader_1_03_pfn_t g_loader{};
bool g_loader_ok = false;

struct AgentPick { int want_bus, want_dev, index, seen; hsa_agent_t by_bdf, by_index, cpu; bool have_bdf, have_index, have_cpu; };

hsa_status_t pick_agent(hsa_agent_t a, void* data)
{
    AgentPick* p = (AgentPick*)data;
    hsa_device_type_t t;
    if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
    if (t == HSA_DEVICE_TYPE_CPU && !p->have_cpu) { p->cpu = a; p->have_cpu = true; }
    if (t != HSA_DEVICE_TYPE_GPU) return HSA_STATUS_SUCCESS;
    uint32_t bdf = 0;
    if (hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &bdf) == HSA_STATUS_SUCCESS
        && (int)((bdf >> 8) & 0xff) == p->want_bus && (int)((bdf >> 3) & 0x1f) == p->want_dev && !p->have_bdf) { p->by_bdf = a; p->have_bdf = true; }
    if (p->seen == p->index) { p->by_index = a; p->have_index = true; }
    p->seen++;
    return HSA_STATUS_SUCCESS;
}

struct SymScan { DeviceCtx* ctx; };

hsa_status_t scan_symbol(hsa_executable_t, hsa_agent_t, hsa_executable_symbol_t s, void* data)
{
    DeviceCtx* ctx = ((SymScan*)data)->ctx;
    hsa_symbol_kind_t kind;
    if (hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_TYPE, &kind) != HSA_STATUS_SUCCESS || kind != HSA_SYMBOL_KIND_KERNEL) return HSA_STATUS_SUCCESS;
    uint32_t len = 0;
    if (hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_NAME_LENGTH, &len) != HSA_STATUS_SUCCESS || len == 0 || len > 4096) return HSA_STATUS_SUCCESS;
    std::string name(len, '\0');
    if (hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_NAME, &name[0]) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
    KernelSym k{};
    if (hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &k.object) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
    (void)hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &k.kernarg);
    (void)hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &k.group);
    (void)hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &k.priv);
    ctx->syms[name] = k;
    return HSA_STATUS_SUCCESS;
}

// the code object a loaded code object came from, when the loader still holds it in host memory (HIP's fat-binary images do):
// its AMDGPU metadata note lists every kernel's arguments, hidden ones included (codeobj_meta.h)
hsa_status_t scan_code_object(hsa_executable_t, hsa_loaded_code_object_t lco, void* data)
{
    DeviceCtx* ctx = ((SymScan*)data)->ctx;
    if (!g_loader.hsa_ven_amd_loader_loaded_code_object_get_info) return HSA_STATUS_SUCCESS;
    uint32_t st = 0;
    if (g_loader.hsa_ven_amd_loader_loaded_code_object_get_info(lco, HSA_VEN_AMD_LOADER_LOADED_CODE_OBJECT_INFO_CODE_OBJECT_STORAGE_TYPE, &st) != HSA_STATUS_SUCCESS
        || st != HSA_VEN_AMD_LOADER_CODE_OBJECT_STORAGE_TYPE_MEMORY) return HSA_STATUS_SUCCESS;
    uint64_t base = 0, size = 0;
    if (g_loader.hsa_ven_amd_loader_loaded_code_object_get_info(lco, HSA_VEN_AMD_LOADER_LOADED_CODE_OBJECT_INFO_CODE_OBJECT_STORAGE_MEMORY_BASE, &base) != HSA_STATUS_SUCCESS
        || g_loader.hsa_ven_amd_loader_loaded_code_object_get_info(lco, HSA_VEN_AMD_LOADER_LOADED_CODE_OBJECT_INFO_CODE_OBJECT_STORAGE_MEMORY_SIZE, &size) != HSA_STATUS_SUCCESS
        || !base || !size) return HSA_STATUS_SUCCESS;
    if (ctx->seen_objects.count(base)) return HSA_STATUS_SUCCESS;
    ctx->seen_objects[base] = true;
    (void)codeobj_hidden_layouts((const void*)(uintptr_t)base, (size_t)size, ctx->layouts);
    return HSA_STATUS_SUCCESS;
}

hsa_status_t scan_executable(hsa_executable_t exe, void* data)
{
    DeviceCtx* ctx = ((SymScan*)data)->ctx;
    (void)hsa_executable_iterate_agent_symbols(exe, ctx->agent, scan_symbol, data);
    if (g_loader.hsa_ven_amd_loader_executable_iterate_loaded_code_objects)
        (void)g_loader.hsa_ven_amd_loader_executable_iterate_loaded_code_objects(exe, scan_code_object, data);
    return HSA_STATUS_SUCCESS;
}

// (re)reads the kernel symbols of every code object loaded so far (HIP loads a module's code object at its first launch)
void rescan(DeviceCtx* ctx)
{
    if (!g_loader_ok) return;
    SymScan sc{ctx};
    (void)g_loader.hsa_ven_amd_loader_iterate_executables(scan_executable, &sc);
}

DeviceCtx* device_ctx(int gpu)
{
    DeviceCtx& c = g_ctx[gpu];
    if (c.tried) return c.ok ? &c : nullptr;
    c.tried = true;
    if (hsa_init() != HSA_STATUS_SUCCESS) return nullptr;           // reference counted: HIP holds the runtime already
    if (!g_loader_ok)
        g_loader_ok = hsa_system_get_major_extension_table(HSA_EXTENSION_AMD_LOADER, 1, sizeof(g_loader), &g_loader) == HSA_STATUS_SUCCESS
                      && g_loader.hsa_ven_amd_loader_iterate_executables != nullptr;
    if (!g_loader_ok) return nullptr;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, gpu) != hipSuccess) return nullptr;
    AgentPick p{};
    p.want_bus = prop.pciBusID; p.want_dev = prop.pciDeviceID; p.index = gpu;
    if (hsa_iterate_agents(pick_agent, &p) != HSA_STATUS_SUCCESS) return nullptr;
    if (!p.have_bdf && !p.have_index) return nullptr;
    c.agent = p.have_bdf ? p.by_bdf : p.by_index;
    c.cpu = p.cpu; c.have_cpu = p.have_cpu;
    c.ok = true;
    return &c;
}

}  // namespace

struct DirectQueue {                                     // one HSA queue, shared by the programs of a graph
    hsa_queue_t* q = nullptr;
    hsa_signal_t done{};                                // counts bursts down from kStart
    uint64_t bursts = 0;                                // closed so far (direct_close)
    bool open = false;                                  // passes submitted since the last direct_close
    int refs = 0;
    std::atomic<int> fault{0};                          // hsa_status_t handed to the queue's error callback (0: healthy)
    static constexpr hsa_signal_value_t kStart = (hsa_signal_value_t)1 << 40;
};

namespace {
thread_local char g_direct_err[256] = "";
void direct_err(const char* what, int code) { snprintf(g_direct_err, sizeof(g_direct_err), "%s (%d)", what, code); }

// a kernel that faults (bad argument layout, scratch failure, memory violation) puts the queue into the error state: the read
// index stops and no signal fires.  The runtime reports it here; every wait below looks at the flag instead of spinning forever
void queue_fault(hsa_status_t status, hsa_queue_t*, void* data)
{
    ((DirectQueue*)data)->fault.store((int)status ? (int)status : -1);
}

double wait_limit_s()
{
    static const double lim = getenv("TAMD_DIRECT_TIMEOUT_S") ? atof(getenv("TAMD_DIRECT_TIMEOUT_S")) : 30.0;
    return lim;
}
}  // namespace

const char* direct_last_error() { return g_direct_err; }

bool direct_probe(int gpu)
{
    std::lock_guard<std::mutex> lk(g_mu);
    return device_ctx(gpu) != nullptr;
}

struct DirectProgram {
    DirectQueue* dq = nullptr;

    std::vector<hsa_kernel_dispatch_packet_t> pkts;     // bodies; headers are written last, per pass
    std::vector<uint16_t> hdr;                          // header of packet i inside a burst (the first packet of a burst: h_open)
    void* kernargs = nullptr;
    uint16_t h_open = 0, h_close = 0, h_wrap = 0;        // first packet of a burst / closing barrier packet / first packet of a later pass
    int n_meta = 0;                                      // packets whose hidden-argument offsets were read from code-object metadata
    hsa_agent_t agent{};                                 // direct_timestamps: the agent whose clock the dispatch times are in
    std::vector<std::string> names;                      // kernel symbol of packet i (direct_packet_name)
};

DirectProgram* direct_build(int gpu, hipStream_t stream, const std::vector<LaunchRec>& recs, const char** why, DirectProgram* share)
{
    static const char* reason = "";
    *why = reason;
    if (recs.empty()) { *why = "empty launch list"; return nullptr; }
    std::lock_guard<std::mutex> lk(g_mu);
    DeviceCtx* ctx = device_ctx(gpu);
    if (!ctx) { *why = "no HSA agent / loader extension for this device"; return nullptr; }
    DirectProgram* p = new DirectProgram;
    std::vector<unsigned char> blob;
    std::vector<size_t> offs;
    std::vector<char> coherent;                         // packet i is a coherent launch (pwdw.hip)
    bool scanned = false;
    int n_meta = 0;                                     // launches whose hidden-argument offsets came from the code object's metadata
    for (const LaunchRec& r : recs) {
        const char* nm = hipKernelNameRefByPtr(r.func, stream);
        if (!nm) { *why = "kernel without a name"; delete p; return nullptr; }
        const std::string key = std::string(nm) + ".kd";
        auto it = ctx->syms.find(key);
        if (it == ctx->syms.end() && !scanned) { rescan(ctx); scanned = true; it = ctx->syms.find(key); }
        if (it == ctx->syms.end()) { *why = "kernel descriptor not found among the loaded code objects"; delete p; return nullptr; }
        const KernelSym& k = it->second;
        // scratch (the out-of-line hand-over paths of a few epilogues reserve a call frame): the packet carries the size and the
        // runtime backs the queue's scratch on demand, as it does for HIP's own queues; TAMD_DIRECT_SCRATCH=0 refuses instead
        static const bool allow_scratch = !(exp_env("TAMD_DIRECT_SCRATCH") && atoi(exp_env("TAMD_DIRECT_SCRATCH")) == 0);
        if (k.priv != 0 && !allow_scratch) { *why = "a kernel of the list uses scratch memory"; delete p; return nullptr; }
        if (r.args.size() > k.kernarg) { *why = "recorded arguments exceed the kernel's argument segment"; delete p; return nullptr; }
        // hidden arguments: at the offsets the kernel's own metadata lists (codeobj_meta.h).  Only where the code object is no
        // longer readable (or carries no note) the code-object-v5 defaults stand in -- block counts at the first 8-byte boundary
        // behind the explicit arguments, group sizes at +12, grid dimensionality at +64 -- and the prerun self-check
        // (graph_exec.hip direct_selfcheck) is what then vouches for them.  TAMD_DIRECT_META=0 forces the defaults (tests).
        HiddenLayout hl;
        {
            static const bool use_meta = !(exp_env("TAMD_DIRECT_META") && atoi(exp_env("TAMD_DIRECT_META")) == 0);
            auto lt = ctx->layouts.find(key);
            if (use_meta && lt != ctx->layouts.end()) {
                hl = lt->second;
                if ((size_t)hl.explicit_end > ((r.args.size() + 7) & ~(size_t)7) || (size_t)hl.explicit_end + 8 < r.args.size()) {
                    *why = "recorded arguments do not match the kernel's metadata"; delete p; return nullptr;
                }
                n_meta++;
            } else {
                const int hid = (int)((r.args.size() + 7) & ~(size_t)7);
                for (int a3 = 0; a3 < 3; a3++) { hl.block_count[a3] = hid + 4 * a3; hl.group_size[a3] = hid + 12 + 2 * a3; hl.remainder[a3] = hid + 18 + 2 * a3; }
                hl.grid_dims = hid + 64;
            }
        }
        size_t seg = k.kernarg;
        for (int a3 = 0; a3 < 3; a3++) seg = std::max<size_t>(seg, std::max(hl.block_count[a3] + 4, std::max(hl.group_size[a3], hl.remainder[a3]) + 2));
        seg = std::max<size_t>(seg, std::max<size_t>(hl.grid_dims + 2, r.args.size()));
        const size_t off = (blob.size() + 255) & ~(size_t)255;
        blob.resize(off + seg, 0);
        memcpy(blob.data() + off, r.args.data(), r.args.size());
        const uint32_t bc[3] = {r.grid.x, r.grid.y, r.grid.z};
        const uint16_t gs[3] = {(uint16_t)r.block.x, (uint16_t)r.block.y, (uint16_t)r.block.z};
        const uint16_t dims = r.grid.z * r.block.z > 1 ? 3 : (r.grid.y * r.block.y > 1 ? 2 : 1), zero16 = 0;
        for (int a3 = 0; a3 < 3; a3++) {
            if (hl.block_count[a3] >= 0) memcpy(blob.data() + off + hl.block_count[a3], &bc[a3], 4);
            if (hl.group_size[a3] >= 0) memcpy(blob.data() + off + hl.group_size[a3], &gs[a3], 2);
            if (hl.remainder[a3] >= 0) memcpy(blob.data() + off + hl.remainder[a3], &zero16, 2);       // grids are whole multiples of the group
        }
        if (hl.grid_dims >= 0) memcpy(blob.data() + off + hl.grid_dims, &dims, 2);
        offs.push_back(off);
        hsa_kernel_dispatch_packet_t pk{};
        pk.setup = (uint16_t)(dims << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS);
        pk.workgroup_size_x = gs[0]; pk.workgroup_size_y = gs[1]; pk.workgroup_size_z = gs[2];
        pk.grid_size_x = r.grid.x * r.block.x; pk.grid_size_y = r.grid.y * r.block.y; pk.grid_size_z = r.grid.z * r.block.z;
        pk.private_segment_size = k.priv;
        pk.group_segment_size = k.group + r.shmem;
        pk.kernel_object = k.object;
        p->pkts.push_back(pk);
        {                                                // readable kernel name for direct_packet_name (measurement output)
            int st = 0;
            char* dm = abi::__cxa_demangle(nm, nullptr, nullptr, &st);
            p->names.push_back(st == 0 && dm ? std::string(dm) : std::string(nm));
            free(dm);
        }
        // kernels that exchange their tensors with agent-scope accesses (sc1 loads, write-through stores): flagged by their launcher
        const bool allow_none = tamd_pin_int("direct_coherent", 1) != 0;      // (read at every prerun: a test flips it inside one process)
        coherent.push_back(allow_none && r.coherent);
    }
    if (hipMalloc(&p->kernargs, blob.size()) != hipSuccess || hipMemcpy(p->kernargs, blob.data(), blob.size(), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipGetLastError();
        *why = "kernel-argument buffer"; direct_destroy(p); return nullptr;
    }
    // tools that intercept the queue (rocprofv3) read kernel-argument segments from the host, as they can for HIP's own
    // (BAR-mapped) argument buffers: map this one for the CPU agent too; without a host mapping it simply stays device-only
    if (ctx->have_cpu) {
        hsa_agent_t both[2] = {ctx->agent, ctx->cpu};
        (void)hsa_amd_agents_allow_access(2, both, nullptr, p->kernargs);
    }
    for (size_t i = 0; i < p->pkts.size(); i++) p->pkts[i].kernarg_address = (char*)p->kernargs + offs[i];
    if (share) {
        p->dq = share->dq;
    } else {
        p->dq = new DirectQueue;
        uint32_t qsize = 1024;
        while (qsize < 8 * p->pkts.size()) qsize *= 2;
        if (hsa_queue_create(ctx->agent, qsize, HSA_QUEUE_TYPE_SINGLE, queue_fault, p->dq, UINT32_MAX, UINT32_MAX, &p->dq->q) != HSA_STATUS_SUCCESS) {
            *why = "hsa_queue_create"; p->dq->q = nullptr; p->dq->refs = 1; direct_destroy(p); return nullptr;
        }
        if (hsa_signal_create(DirectQueue::kStart, 0, nullptr, &p->dq->done) != HSA_STATUS_SUCCESS) { *why = "hsa_signal_create"; p->dq->refs = 1; direct_destroy(p); return nullptr; }
        // experiment (tools/exp): run the whole list on a subset of the CUs -- "first32" = mask bits 0..31,
        // "stride8" = every 8th bit (tools/exp/cumask_probe.hip tells which of the two is one XCD on this stack)
        if (const char* cm = exp_env("TAMD_DIRECT_CU_MASK")) {
            uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (!strcmp(cm, "first32")) mask[0] = 0xffffffffu;
            else if (!strcmp(cm, "stride8")) for (int i = 0; i < 8; i++) mask[i] = 0x01010101u;
            else if (!strcmp(cm, "first64")) mask[0] = mask[1] = 0xffffffffu;
            else if (!strcmp(cm, "stride4")) for (int i = 0; i < 8; i++) mask[i] = 0x11111111u;
            if (mask[0]) {
                const hsa_status_t st = hsa_amd_queue_cu_set_mask(p->dq->q, 256, mask);
                if (getenv("TAMD_DEBUG")) fprintf(stderr, "[tamd] direct queue CU mask %s: %s\n", cm, st == HSA_STATUS_SUCCESS ? "set" : "REFUSED");
            }
        }
    }
    p->dq->refs++;
    p->agent = ctx->agent;
    if (p->pkts.size() * 2 > p->dq->q->size) { *why = "launch list longer than the shared queue"; direct_destroy(p); return nullptr; }
    auto header = [](int type, int acq, int rel, bool barrier = true) {
        return (uint16_t)((type << HSA_PACKET_HEADER_TYPE) | ((barrier ? 1 : 0) << HSA_PACKET_HEADER_BARRIER) | (acq << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE)
                          | (rel << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
    };
    // fences inside a burst.  An ordinary launch acquires and releases at agent scope.  A coherent launch reads what other
    // launches wrote through agent-scope loads and writes through: nothing to write back behind it (release none), and
    // nothing to invalidate in front of it -- unless its predecessor is an ordinary launch: the first convolution reads the
    // graph input with ordinary loads, and in the host-to-host list that input was just written by the upload launch.
    const int K = HSA_PACKET_TYPE_KERNEL_DISPATCH;
    int n_beside = 0;
    for (size_t i = 0; i < p->pkts.size(); i++) {
        // a launch the graph marked independent of its predecessors (LaunchRec::beside) carries no barrier bit: the packet
        // processor starts it while they run; whatever depends on it comes later with the bit set and waits for ALL of them
        // -- and no acquire either: what it reads was released before the last ordered launch began, and that launch's acquire
        // already dropped every stale line; nothing it reads has been written since (or it would not be independent)
        const bool beside = i > 0 && recs[i].beside;
        n_beside += beside;
        static const bool exp_nofence = exp_env("TAMD_EXP_NOFENCE") != nullptr;        // timing experiment: bytes not trustworthy (graph.hip)
        if (!coherent[i] && exp_nofence) p->hdr.push_back(header(K, HSA_FENCE_SCOPE_NONE, HSA_FENCE_SCOPE_NONE, !beside));
        else if (!coherent[i]) p->hdr.push_back(header(K, beside ? HSA_FENCE_SCOPE_NONE : HSA_FENCE_SCOPE_AGENT, HSA_FENCE_SCOPE_AGENT, !beside));
        else p->hdr.push_back(header(K, (i > 0 && !coherent[i - 1]) ? HSA_FENCE_SCOPE_AGENT : HSA_FENCE_SCOPE_NONE, HSA_FENCE_SCOPE_NONE));
    }
    if (getenv("TAMD_DEBUG")) fprintf(stderr, "[tamd] direct: %zu packets, %d without the barrier bit, hidden arguments of %d from code-object metadata\n", p->pkts.size(), n_beside, n_meta);
    p->n_meta = n_meta;
    p->h_open = header(K, HSA_FENCE_SCOPE_SYSTEM, coherent[0] ? HSA_FENCE_SCOPE_NONE : HSA_FENCE_SCOPE_AGENT);
    p->h_close = header(HSA_PACKET_TYPE_BARRIER_AND, HSA_FENCE_SCOPE_AGENT, HSA_FENCE_SCOPE_SYSTEM);
    // a pass queued right behind another pass of the same burst: when the graph found the first launch independent of the
    // previous pass's last launches (LaunchRec::wrap), its packet carries no barrier bit -- it starts while they finish; the
    // second packet has the bit again and waits for both.  The fences stay what they are (the input may have been rewritten in
    // between; a coherent first launch reads it with agent-scope loads as before).
    p->h_wrap = (!recs.empty() && recs[0].wrap) ? (uint16_t)(p->hdr[0] & ~(1u << HSA_PACKET_HEADER_BARRIER)) : p->hdr[0];      // same fences, no barrier bit
    if (getenv("TAMD_DEBUG") && p->h_wrap != p->hdr[0]) fprintf(stderr, "[tamd] direct: the first launch of a pass runs beside the end of the previous pass\n");
    return p;
}

int direct_packets(const DirectProgram* p) { return p ? (int)p->pkts.size() : 0; }
const char* direct_packet_name(const DirectProgram* p, int i) { return (p && i >= 0 && i < (int)p->names.size()) ? p->names[i].c_str() : ""; }

// The directly dispatched pass under the HSA runtime's OWN dispatch profiling (hsa_amd_profiling_set_profiler_enabled +
// hsa_amd_profiling_get_dispatch_time: the packet processor stamps the start and the end of every dispatch into its completion
// signal -- the very timestamps rocprofv3's kernel trace reports, without the tool's queue interception, which does not survive
// packets it did not see HIP write).  `passes` passes of the program, each packet with a completion signal of its own (that is
// the only change to the packets: same headers, same fences, same barrier bits), one burst per pass.
//   dur_us[i]  mean duration of packet i (end - start)
//   gap_us[i]  mean time from the end of packet i to the start of packet i + 1 (the launch boundary as the device saw it);
//              gap_us[n - 1] = from the end of the last packet to the start of the first packet of the NEXT pass (0 for the last pass)
// A measurement entry point (round 6: the timed path had never been seen by anything but the host's clock); not used by a run.
int direct_timestamps(DirectProgram* p, int passes, double* dur_us, double* gap_us)
{
    const int n = (int)p->pkts.size();
    if (n == 0 || passes < 1) return -1;
    DirectQueue* dq = p->dq;
    if (direct_wait_all(p)) return -1;
    // On this stack (ROCm 7.x) hsa_amd_profiling_get_dispatch_time hands out SYSTEM-domain ticks already (HSA_SYSTEM_INFO_TIMESTAMP_FREQUENCY =
    // 1 GHz): measured in round 6, calls 9 / 10 -- the sums of the stamps of a batched pass equal the host's clock within 0.5 % in that unit
    // (10.0 x off with the agent's 100 MHz figure), and hsa_amd_profiling_convert_tick_to_system_domain applied on top returns non-monotonic
    // values.  So: no conversion, system frequency.
    uint64_t freq = 0;
    if (hsa_system_get_info(HSA_SYSTEM_INFO_TIMESTAMP_FREQUENCY, &freq) != HSA_STATUS_SUCCESS || !freq) { direct_err("no timestamp frequency", 0); return -1; }
    if (hsa_amd_profiling_set_profiler_enabled(dq->q, 1) != HSA_STATUS_SUCCESS) { direct_err("hsa_amd_profiling_set_profiler_enabled", 0); return -1; }
    // the first pass after profiling is switched on is not stamped (the packet processor picks the queue property up with the next
    // doorbell: call 10 read start == end == a constant for all of its packets): one extra pass in front, dropped below
    passes += passes < 20 ? 3 : 1;
    std::vector<hsa_signal_t> sig((size_t)n * passes);
    for (auto& s : sig)
        if (hsa_signal_create(1, 0, nullptr, &s) != HSA_STATUS_SUCCESS) { direct_err("hsa_signal_create", 0); return -1; }
    int rc = 0;
    hsa_queue_t* q = dq->q;
    hsa_kernel_dispatch_packet_t* base = (hsa_kernel_dispatch_packet_t*)q->base_address;
    const uint64_t mask = q->size - 1;
    for (int ps = 0; ps < passes && !rc; ps++) {
        // all passes go out back to back (one burst), as the timed loop of bench.py submits them
        const uint64_t idx0 = hsa_queue_add_write_index_relaxed(q, (uint64_t)n);
        while (idx0 + n - hsa_queue_load_read_index_scacquire(q) > q->size)
            if (dq->fault.load()) { rc = -1; break; }
        for (int i = 0; i < n && !rc; i++) {
            hsa_kernel_dispatch_packet_t* d = base + ((idx0 + i) & mask);
            const hsa_kernel_dispatch_packet_t& s = p->pkts[i];
            d->setup = s.setup;
            d->workgroup_size_x = s.workgroup_size_x; d->workgroup_size_y = s.workgroup_size_y; d->workgroup_size_z = s.workgroup_size_z;
            d->reserved0 = 0;
            d->grid_size_x = s.grid_size_x; d->grid_size_y = s.grid_size_y; d->grid_size_z = s.grid_size_z;
            d->private_segment_size = s.private_segment_size; d->group_segment_size = s.group_segment_size;
            d->kernel_object = s.kernel_object; d->kernarg_address = s.kernarg_address; d->reserved2 = 0;
            d->completion_signal = sig[(size_t)ps * n + i];
            const uint16_t h = i == 0 ? (dq->open ? p->h_wrap : p->h_open) : p->hdr[i];
            __atomic_store_n(&d->header, h, __ATOMIC_RELEASE);
        }
        const uint64_t to_end = q->size - (idx0 & mask);
        if (to_end < (uint64_t)n) hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)(idx0 + to_end - 1));
        hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)(idx0 + n - 1));
        dq->open = true;
        // the warm-up pass completes before the stamped ones are written: with a handful of short passes the whole burst used to be in the
        // ring before the packet processor had looked at the first doorbell, and the first stamped pass read start == end as well
        // (tests/test_gpu_split_batch.py: 5 passes of a batch-2 MobileNet; 30+ passes never showed it)
        if (ps == 0 && !rc) rc = direct_wait(p);
    }
    if (!rc) rc = direct_wait(p);
    std::vector<double> t0((size_t)n * passes, 0.0), t1((size_t)n * passes, 0.0);
    int first_ok = 1, n_bad = 0, last_bad = -1;              // first pass whose packets (and every later pass's) all carry stamps
    for (size_t k = 0; k < sig.size() && !rc; k++) {
        hsa_amd_profiling_dispatch_time_t t{};
        if (hsa_amd_profiling_get_dispatch_time(p->agent, sig[k], &t) != HSA_STATUS_SUCCESS) { direct_err("hsa_amd_profiling_get_dispatch_time", (int)k); rc = -1; break; }
        if (getenv("TAMD_DEBUG") && k >= (size_t)n && k < (size_t)n + 4)
            fprintf(stderr, "[tamd] stamp %zu: start %llu end %llu (%lld ticks), system timestamp frequency %llu Hz\n", k, (unsigned long long)t.start,
                    (unsigned long long)t.end, (long long)(t.end - t.start), (unsigned long long)freq);
        // (the queue property is picked up by the packet processor some time after the switch: passes in front of the first fully
        //  stamped one are dropped below -- at least the warm-up pass, on short launch lists sometimes one or two more)
        if (t.end <= t.start) { first_ok = std::max(first_ok, (int)(k / (size_t)n) + 1); n_bad++; last_bad = (int)k; }
        const uint64_t origin = 0;
        t0[k] = 1e6 * (double)(t.start - origin) / (double)freq; t1[k] = 1e6 * (double)(t.end - origin) / (double)freq;
    }
    (void)hsa_amd_profiling_set_profiler_enabled(dq->q, 0);
    for (auto& s : sig) (void)hsa_signal_destroy(s);
    if (rc) return -1;
    if (first_ok >= passes) {
        char msg[200];
        snprintf(msg, sizeof(msg), "no pass carries dispatch stamps throughout: %d of %d packets without (the last one: packet %d of pass %d, %s)", n_bad, n * passes,
                 last_bad % n, last_bad / n, direct_packet_name(p, last_bad % n));
        direct_err(msg, passes - 1);
        return -1;
    }
    for (int i = 0; i < n; i++) {
        double d = 0, g = 0;
        int ng = 0;
        for (int ps = first_ok; ps < passes; ps++) {     // (pass 0: the unstamped warm-up)
            const size_t k = (size_t)ps * n + i;
            d += t1[k] - t0[k];
            if (k + 1 < sig.size()) { g += t0[k + 1] - t1[k]; ng++; }
        }
        dur_us[i] = d / (passes - first_ok);
        gap_us[i] = ng ? g / ng : 0.0;
    }
    return n;
}
int direct_meta_packets(const DirectProgram* p) { return p ? p->n_meta : 0; }

// close_burst: this pass is the last of its burst and its LAST packet closes it -- it releases at system scope and carries the
// queue's completion signal itself, so no barrier packet follows (one packet less for the packet processor to walk: what a
// blocking host-to-host run waits for is the end of that kernel, not the retirement of an empty packet behind it)
int direct_submit(DirectProgram* p, bool close_burst, unsigned long long* burst)
{
    const uint64_t n = p->pkts.size();
    DirectQueue* dq = p->dq;
    hsa_queue_t* q = dq->q;
    if (dq->fault.load()) { direct_err("the HSA queue is in the error state (a dispatched kernel faulted)", dq->fault.load()); return -1; }
    const uint64_t idx0 = hsa_queue_add_write_index_relaxed(q, n);
    if (idx0 + n - hsa_queue_load_read_index_scacquire(q) > q->size) {          // ring full: the packet processor is behind
        const auto t0 = std::chrono::steady_clock::now();
        while (idx0 + n - hsa_queue_load_read_index_scacquire(q) > q->size) {
            if (dq->fault.load() || std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > wait_limit_s()) {
                // the slots were reserved but never written: the queue is unusable from here on
                if (!dq->fault.load()) dq->fault.store(-2);
                direct_err("the HSA queue does not drain (faulted kernel or hung device)", dq->fault.load());
                return -1;
            }
        }
    }
    hsa_kernel_dispatch_packet_t* base = (hsa_kernel_dispatch_packet_t*)q->base_address;
    const uint64_t mask = q->size - 1;
    for (uint64_t i = 0; i < n; i++) {
        hsa_kernel_dispatch_packet_t* d = base + ((idx0 + i) & mask);
        const hsa_kernel_dispatch_packet_t& s = p->pkts[i];
        d->setup = s.setup;
        d->workgroup_size_x = s.workgroup_size_x; d->workgroup_size_y = s.workgroup_size_y; d->workgroup_size_z = s.workgroup_size_z;
        d->reserved0 = 0;
        d->grid_size_x = s.grid_size_x; d->grid_size_y = s.grid_size_y; d->grid_size_z = s.grid_size_z;
        d->private_segment_size = s.private_segment_size; d->group_segment_size = s.group_segment_size;
        d->kernel_object = s.kernel_object; d->kernarg_address = s.kernarg_address; d->reserved2 = 0;
        d->completion_signal.handle = 0;
        uint16_t h = i == 0 ? (dq->open ? p->h_wrap : p->h_open) : p->hdr[i];
        if (close_burst && i + 1 == n) {
            // the closing packet carries the completion signal, so it must retire LAST: barrier bit forced (a packet recorded
            // "beside" its predecessor -- independent heads with TAMD_DIRECT_OVERLAP=1 -- has none and could complete, and
            // decrement the signal, while earlier packets of the pass still run) and system-scope release
            d->completion_signal = dq->done;
            h = (uint16_t)((h & ~(3u << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE)) | (HSA_FENCE_SCOPE_SYSTEM << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE)
                           | (1u << HSA_PACKET_HEADER_BARRIER));
        }
        __atomic_store_n(&d->header, h, __ATOMIC_RELEASE);
    }
    // one doorbell per pass -- two when the pass wraps around the end of the ring, so that a queue interceptor is never handed
    // a batch that is not contiguous in memory
    const uint64_t to_end = q->size - (idx0 & mask);
    if (to_end < n) hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)(idx0 + to_end - 1));
    hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)(idx0 + n - 1));
    if (close_burst) {
        dq->bursts++;
        dq->open = false;
        if (burst) *burst = dq->bursts;
    } else
        dq->open = true;
    return 0;
}

// closes the burst: one barrier packet behind everything submitted to the queue (barrier bit: it waits for the last kernel),
// system-scope release, completion signal.  The signal counts bursts down from kStart, so a burst number is a point on ONE
// timeline: "burst b has completed" == signal <= kStart - b, and the asynchronous runs of a graph (two bursts in flight) need no
// signal of their own.
int direct_close(DirectProgram* p, unsigned long long* burst)
{
    DirectQueue* dq = p->dq;
    if (burst) *burst = dq->bursts;
    if (!dq->open) return 0;
    if (dq->fault.load()) { direct_err("the HSA queue is in the error state (a dispatched kernel faulted)", dq->fault.load()); return -1; }
    hsa_queue_t* q = dq->q;
    const uint64_t idx = hsa_queue_add_write_index_relaxed(q, 1);
    const auto t0 = std::chrono::steady_clock::now();
    while (idx + 1 - hsa_queue_load_read_index_scacquire(q) > q->size) {
        if (dq->fault.load() || std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > wait_limit_s()) {
            if (!dq->fault.load()) dq->fault.store(-2);
            direct_err("the HSA queue does not drain (faulted kernel or hung device)", dq->fault.load());
            return -1;
        }
    }
    hsa_barrier_and_packet_t* b = (hsa_barrier_and_packet_t*)q->base_address + (idx & (q->size - 1));
    b->reserved0 = 0; b->reserved1 = 0; b->reserved2 = 0;
    for (int i = 0; i < 5; i++) b->dep_signal[i].handle = 0;
    b->completion_signal = dq->done;
    __atomic_store_n(&b->header, p->h_close, __ATOMIC_RELEASE);
    hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)idx);
    dq->bursts++;
    dq->open = false;
    if (burst) *burst = dq->bursts;
    return 0;
}

// The host spins on the signal first (a blocking run is a latency measurement: an interrupt-driven wake-up costs more than the
// pass) and only then sleeps on it -- in slices, looking at the queue's fault flag between them, never forever.
int direct_wait_burst(DirectProgram* p, unsigned long long burst)
{
    DirectQueue* dq = p->dq;
    const hsa_signal_value_t target = DirectQueue::kStart - (hsa_signal_value_t)burst;
    if (hsa_signal_load_scacquire(dq->done) <= target) return 0;
    if (hsa_signal_wait_scacquire(dq->done, HSA_SIGNAL_CONDITION_LT, target + 1, 2000000 /* timestamp ticks of spinning */, HSA_WAIT_STATE_ACTIVE) <= target) return 0;
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        // ~50 ms slices (the timeout is in timestamp ticks of the system clock, 100 MHz here; only its order of magnitude matters)
        if (hsa_signal_wait_scacquire(dq->done, HSA_SIGNAL_CONDITION_LT, target + 1, 5000000, HSA_WAIT_STATE_BLOCKED) <= target) return 0;
        if (dq->fault.load()) { direct_err("a dispatched kernel faulted: the HSA queue is in the error state", dq->fault.load()); return -1; }
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > wait_limit_s()) {
            // a pass that is merely slow is an error for THIS wait, not a dead queue: only the runtime's fault callback marks the
            // queue unusable (a later wait on the same burst may still succeed)
            direct_err("timeout waiting for a direct pass (TAMD_DIRECT_TIMEOUT_S)", -3);
            return -1;
        }
    }
}

int direct_wait(DirectProgram* p)
{
    unsigned long long b = 0;
    if (!p->dq->open) return 0;
    if (direct_close(p, &b)) return -1;
    return direct_wait_burst(p, b);
}

// everything ever submitted to the program's queue: the open burst (closed here) AND the bursts that were closed at submit
// time and never collected (asynchronous runs: direct_close without a wait)
int direct_wait_all(DirectProgram* p)
{
    if (direct_wait(p)) return -1;
    return direct_wait_burst(p, p->dq->bursts);
}

void direct_destroy(DirectProgram* p)
{
    if (!p) return;
    if (p->dq) {
        // packets may still be running: the open burst, and bursts closed at submit time that nobody waited for (asynchronous
        // runs) -- kernel arguments, tensors and the queue itself must outlive them
        if (p->dq->q && !p->dq->fault.load()) (void)direct_wait_all(p);
        if (--p->dq->refs <= 0) {
            if (p->dq->q) (void)hsa_queue_destroy(p->dq->q);
            if (p->dq->done.handle) (void)hsa_signal_destroy(p->dq->done);
            delete p->dq;
        }
    }
    if (p->kernargs) (void)hipFree(p->kernargs);
    delete p;
}

}  // namespace tamd
