// Launch recorder: every kernel launch of the backend goes through launch_rec() (kernels.h redefines hipLaunchKernelGGL to
// it).  Normally it only launches.  While a recording list is installed (prerun, direct dispatch) it also keeps what the
// launch WAS -- kernel, geometry, dynamic LDS bytes and the explicit kernel-argument segment, packed with the kernel's own
// parameter types at their natural alignment (the layout the code object's metadata gives the by-value arguments) -- so that
// direct.cc can replay the list as AQL packets without going through the HIP launch path.
#pragma once
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>
#include <vector>

namespace tamd {

struct LaunchRec {
    const void* func;                   // host-side kernel handle (what hipLaunchKernel takes)
    dim3 grid, block;
    unsigned shmem;                     // dynamic LDS bytes
    std::vector<unsigned char> args;    // explicit kernel-argument segment
    bool coherent = false;              // the kernel exchanges its tensors with agent-scope accesses (launch_rec_coherent): no fences
    bool beside = false;                // independent of every launch since the last ordered one (launch_rec_beside): no barrier bit
    bool wrap = false;                  // (first record only) independent of the END of the previous pass: in a burst of passes it starts beside it
};

extern thread_local std::vector<LaunchRec>* g_launch_rec;       // direct.cc; non-null while a launch list is being recorded
extern thread_local bool g_launch_coherent;                     // direct.cc; set by launch_rec_coherent() for the NEXT launch
// a launcher calls this right before it launches a COHERENT kernel instance (pwdw.hip): the record says so explicitly -- the
// packet's fence scopes do not hang on a naming convention
inline void launch_rec_coherent() { g_launch_coherent = true; }
extern thread_local bool g_launch_beside;                       // direct.cc; set by launch_rec_beside() for the NEXT launch
// the graph calls this before a step whose reads and writes touch nothing the launches since the last ordered launch write or
// read (graph_exec.hip run_steps): its first launch may start while they still run -- its packet goes out without the barrier bit
inline void launch_rec_beside() { g_launch_beside = true; }
// both flags belong to ONE step: a step that launched nothing (or failed before its launch) must not hand them to the next one
inline void launch_rec_clear_flags() { g_launch_coherent = false; g_launch_beside = false; }

template <typename T>
inline void rec_pack(std::vector<unsigned char>& b, const T& v)
{
    const size_t off = (b.size() + alignof(T) - 1) & ~(alignof(T) - 1);
    b.resize(off + sizeof(T));
    memcpy(b.data() + off, &v, sizeof(T));
}

template <typename... P, typename... A>
inline void launch_rec(void (*kernel)(P...), dim3 grid, dim3 block, size_t shmem, hipStream_t s, A&&... a)
{
    static_assert(sizeof...(P) == sizeof...(A), "kernel launched with the wrong number of arguments");
    if (g_launch_rec) {
        LaunchRec r;
        r.func = reinterpret_cast<const void*>(kernel);
        r.grid = grid; r.block = block; r.shmem = (unsigned)shmem;
        (rec_pack<P>(r.args, static_cast<P>(a)), ...);
        r.coherent = g_launch_coherent;
        r.beside = g_launch_beside;
        g_launch_rec->push_back(std::move(r));
    }
    g_launch_coherent = false;
    g_launch_beside = false;
    kernel<<<grid, block, shmem, s>>>(static_cast<P>(a)...);
}

// ---- direct dispatch (direct.cc) ---------------------------------------------------------------------------------------------
struct DirectProgram;
// Builds the AQL form of a recorded launch list for HIP device `gpu`; nullptr (with *why) when something in the list cannot be
// dispatched directly (a kernel that needs scratch memory, an unresolved symbol, no HSA queue): the caller keeps the hipGraph.
// `share`: a program of the same graph whose HSA queue (and burst state) the new one uses as well.
DirectProgram* direct_build(int gpu, hipStream_t stream, const std::vector<LaunchRec>& recs, const char** why, DirectProgram* share = nullptr);
bool direct_probe(int gpu);                 // HSA agent + loader extension present for HIP device `gpu` (cheap; asked before planning)
// one pass over the list: packets + one doorbell; returns without waiting; -1: queue fault / ring stuck.  close_burst: the pass's
// last packet closes the burst itself (system-scope release + the completion signal), *burst = ticket for direct_wait_burst
int direct_submit(DirectProgram* p, bool close_burst = false, unsigned long long* burst = nullptr);
int direct_close(DirectProgram* p, unsigned long long* burst);   // closes the burst with a barrier packet that carries the completion signal;
                                                                 // does not wait.  *burst: ticket for direct_wait_burst
int direct_wait_burst(DirectProgram* p, unsigned long long burst);   // until that burst (and everything before it) has completed; -1: queue fault / timeout
int direct_wait(DirectProgram* p);          // direct_close + direct_wait_burst: until every pass of the OPEN burst has completed
int direct_wait_all(DirectProgram* p);      // .. and every burst closed earlier without a wait (asynchronous runs)
const char* direct_last_error();            // what the last -1 of this thread was about
int direct_packets(const DirectProgram* p);
const char* direct_packet_name(const DirectProgram* p, int i);       // kernel symbol of packet i
// `passes` back-to-back passes with every packet stamped by the HSA runtime's dispatch profiling: mean duration of packet i and mean
// gap to the next packet's start, microseconds (direct.cc); returns the packet count, -1 on error
int direct_timestamps(DirectProgram* p, int passes, double* dur_us, double* gap_us);
int direct_meta_packets(const DirectProgram* p);      // of those, how many took their hidden-argument offsets from code-object metadata
void direct_destroy(DirectProgram* p);

}  // namespace tamd
