// Shared pieces of the lean-loop implicit-GEMM kernels (conv_pgemm.hip, conv_pgemm_w.hip): LDS-DMA copy, counted-wait immediates,
// multiply-high division, the device-clock stamps of tools/exp/pgemm_anatomy.hip.
#pragma once
#include <stdlib.h>

#include <type_traits>

#include "epilogue.h"
#include "gemm_epilogue.h"
#include "kernels.h"

namespace tamd {

typedef int v4i_p __attribute__((ext_vector_type(4)));
typedef int v16i_p __attribute__((ext_vector_type(16)));

#define PG_GLDS16(gptr, lptr)                                                                          \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),           \
                                     (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

#define PG_GLDS4(gptr, lptr)                                                                           \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),           \
                                     (__attribute__((address_space(3))) void*)(lptr), 4, 0, 0)

// s_waitcnt immediate (gfx9 encoding): vmcnt = n, lgkmcnt = 0, expcnt untouched
#define PG_WAITCNT(n) ((((n) & 15) | (7 << 4) | (0 << 8) | (((n) >> 4) << 14)))

// floor(v / d) by multiply-high: mg = ceil(2^40 / d), exact for v < 2^24, d < 2^16 (the launcher checks both)
__device__ __forceinline__ int pg_div(int v, unsigned long long mg) { return (int)(((unsigned long long)(unsigned)v * mg) >> 40); }

template <int N> struct pg_int { static constexpr int value = N; };

// tools/exp/pgemm_anatomy.hip only: per-block device-clock stamps (wave 0) and ablation switches
#ifdef TAMD_IGEMM_STAMPS
#define PG_STAMP(i) do { if (a.dbg_stamps && threadIdx.x == 0) a.dbg_stamps[((size_t)pg_rep * gridDim.x + blockIdx.x) * 8 + (i)] = ((i) == 0 || (i) == 6) ? (long long)wall_clock64() : (long long)clock64(); } while (0)
// the whole tile computation pg_reps times in ONE launch (dbg_flags >> 8 extra passes): the second pass runs the same code with a
// warm instruction cache -- how much of a block's time is instruction fetch?
#define PG_REPS (((a.dbg_flags >> 8) & 0xff) + 1)
#else
#define PG_STAMP(i) do { } while (0)
#define PG_REPS 1
#endif
#ifdef TAMD_PG_ABLATE            // run-time ablation switches (each costs a scalar branch where it is tested: only for A/B runs)
#define PG_ON(bit) (!(a.dbg_flags & (bit)))
#else
#define PG_ON(bit) true
#endif

}  // namespace tamd
