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
};

extern thread_local std::vector<LaunchRec>* g_launch_rec;       // direct.cc; non-null while a launch list is being recorded

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
        g_launch_rec->push_back(std::move(r));
    }
    kernel<<<grid, block, shmem, s>>>(static_cast<P>(a)...);
}

// ---- direct dispatch (direct.cc) ---------------------------------------------------------------------------------------------
struct DirectProgram;
// Builds the AQL form of a recorded launch list for HIP device `gpu`; nullptr (with *why) when something in the list cannot be
// dispatched directly (a kernel that needs scratch memory, an unresolved symbol, no HSA queue): the caller keeps the hipGraph.
// `share`: a program of the same graph whose HSA queue (and burst state) the new one uses as well.
DirectProgram* direct_build(int gpu, hipStream_t stream, const std::vector<LaunchRec>& recs, const char** why, DirectProgram* share = nullptr);
int direct_submit(DirectProgram* p);        // one pass over the list: packets + one doorbell; returns without waiting
int direct_wait(DirectProgram* p);          // until every submitted pass has completed
int direct_packets(const DirectProgram* p);
void direct_destroy(DirectProgram* p);

}  // namespace tamd
