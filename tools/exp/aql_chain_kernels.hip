// device side of aql_chain.cpp: a dependent chain link that works for `ticks` of the 100 MHz wall clock on `blocks` workgroups
#include <hip/hip_runtime.h>
extern "C" __global__ void chain_link(unsigned long long* counter, unsigned* scratch, int ticks)
{
    const unsigned long long t0 = wall_clock64();
    unsigned v = scratch[blockIdx.x * 64 + (threadIdx.x & 63)];
    while (wall_clock64() - t0 < (unsigned long long)ticks) v = v * 1664525u + 1013904223u;
    scratch[blockIdx.x * 64 + (threadIdx.x & 63)] = v;
    if (blockIdx.x == 0 && threadIdx.x == 0) counter[0] = counter[0] + 1;      // plain read-modify-write: only a correctly ordered chain counts right
}
