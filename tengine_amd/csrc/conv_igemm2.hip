// int8 implicit-GEMM convolution, large-problem schedule: async global->LDS copies, 3-stage ring.
//
// Same arithmetic, operand roles, weight packing and fused epilogue as conv_igemm.hip (reference chain
// conv_kernel_x86.c:187-242, :963-1007, :1008-1630, :1796-1893).  What changes is how the MFMA pipe is fed,
// for the layers where the contraction dominates (ResNet 3x3 / wide 1x1 at batch >= 8):
//   * tiles are staged with global_load_lds_dwordx4 (LDS-DMA): no VGPR round trip, no ds_write pass; the
//     K loop keeps TWO 64-deep stages in flight while the third is consumed -- one s_barrier per stage,
//     counted s_waitcnt vmcnt(N), never a drain inside the loop;
//   * LDS-DMA writes lane-linear (wave base + lane*16), so rows cannot be padded; bank conflicts of the
//     ds_read_b128 fragment reads are removed by swizzling the SOURCE granule instead: LDS slot (row, q)
//     holds k-granule q ^ ((row>>2)&3), readers apply the same XOR (conflict-free for the 16-lane groups
//     {0-3,12-15,20-27}.. of ds_read_b128, 64 banks);
//   * out-of-image im2col taps read a 16-byte zero page instead of being predicated;
//   * 128 x BN block tile, 4 waves of 64 x BN/2; XCD-aware tile map as in conv_igemm.hip.
#include <stdlib.h>

#include "epilogue.h"
#include "gemm_epilogue.h"
#include "env.h"
#include "kernels.h"

namespace tamd {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define TAMD_GLDS16(gptr, lptr)                                                                       \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),          \
                                     (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

template <int BN, bool IS1X1, int STAGES>
__global__ __launch_bounds__(256) void conv_igemm2_i8_kernel(ConvArgs a)
{
    constexpr int BM = 128, BK = 64;
    static_assert(STAGES >= 3 && STAGES <= 6, "ring depth");
    constexpr int ROWS = BM + BN;                 // rows staged per K stage: [0,BN) weights, [BN,BN+BM) pixels
    constexpr int NI = ROWS / 64;                 // LDS-DMA instructions per wave per stage
    constexpr int NA = BN / 64;                   // .. of which for the weight tile
    constexpr int NB = BM / 64;
    constexpr int STAGE_BYTES = ROWS * BK;
    constexpr int WN = 2, WM = 2;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    static_assert(TN >= 1, "BN too small");

    // ONE LDS object and NO other LDS/VMEM reads inside the K loop: hipcc drains vmcnt(0) in front of any
    // ds_read or ordinary load it cannot disambiguate from the in-flight LDS-DMA, which would serialise the ring
    extern __shared__ __attribute__((aligned(16))) int8_t smem[];

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wm = wave % WM, wn = wave / WM;

    const int tiles_n = (a.cout + BN - 1) / BN;
    const int tiles_m = (a.M + BM - 1) / BM;
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int tile_m = (local / tiles_n) * 8 + xcd;
    const int tile_n = local % tiles_n;
    if (tile_m >= tiles_m) return;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int ntaps = a.KH * a.KW;
    // tap -> (ky, kx) without a table: floor(tap/KW) == (tap*magic)>>16 for tap < 128, KW <= 16
    const int kw_magic = 65536 / a.KW + 1;

    // ---- loader state: this lane always moves LDS slot (row = 16*blk + lane>>2, q = lane&3) ----
    const int lrow = lane >> 2;
    const int gk = (lane & 3) ^ ((lane >> 4) & 3);        // source k-granule for that slot (swizzle)
    const int8_t* wsrc[NA];
#pragma unroll
    for (int j = 0; j < NA; j++) wsrc[j] = a.w + (size_t)(n0 + (j * 4 + wave) * 16 + lrow) * a.kpad + gk * 16;
    const int8_t* xbase[NB];
    int iy0[NB], ix0[NB];
    bool rvalid[NB];
#pragma unroll
    for (int j = 0; j < NB; j++) {
        const int m = m0 + (j * 4 + wave) * 16 + lrow;
        rvalid[j] = m < a.M;
        const int mm = rvalid[j] ? m : 0;
        if (IS1X1) {
            xbase[j] = a.x + (size_t)mm * a.cs_in + gk * 16;
            iy0[j] = ix0[j] = 0;
        } else {
            const int ohw = a.OH * a.OW;
            const int n = mm / ohw, rem = mm - n * ohw;
            const int oy = rem / a.OW, ox = rem - oy * a.OW;
            xbase[j] = a.x + (size_t)n * a.H * a.W * a.cs_in;
            iy0[j] = oy * a.SH - a.PH;
            ix0[j] = ox * a.SW - a.PW;
        }
    }
    int tap = (gk * 16) / a.ckp;
    int ci = (gk * 16) - tap * a.ckp;

    auto issue = [&](int s) {                               // stage s -> ring slot s % STAGES
        int8_t* base = smem + (s % STAGES) * STAGE_BYTES;
#pragma unroll
        for (int j = 0; j < NA; j++)
            TAMD_GLDS16(wsrc[j] + (size_t)s * BK, base + ((j * 4 + wave) * 16) * BK);
        if (IS1X1) {
            const bool kvalid = (s * BK + gk * 16) < a.ktot;
#pragma unroll
            for (int j = 0; j < NB; j++) {
                const int8_t* src = (rvalid[j] && kvalid) ? xbase[j] + (size_t)s * BK : a.zeros;
                TAMD_GLDS16(src, base + (BN + (j * 4 + wave) * 16) * BK);
            }
        } else {
            const bool tvalid = tap < ntaps;
            const int ky = (tap * kw_magic) >> 16, kx = tap - ky * a.KW;
            const int dy = ky * a.DH, dx = kx * a.DW;
#pragma unroll
            for (int j = 0; j < NB; j++) {
                const int iy = iy0[j] + dy, ix = ix0[j] + dx;
                const bool ok = rvalid[j] && tvalid && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
                const int8_t* src = ok ? xbase[j] + ((size_t)iy * a.W + ix) * a.cs_in + ci : a.zeros;
                TAMD_GLDS16(src, base + (BN + (j * 4 + wave) * 16) * BK);
            }
            ci += BK;
            while (ci >= a.ckp) { ci -= a.ckp; tap++; }
        }
    };

    v16i acc[TN][TM];
    igemm_acc_from_bias<TM, TN>(acc, a.bias, n0, wn, hi);      // the epilogue adds nothing (gemm_epilogue.h)

    // fragment read offsets: row r = 32*tile + l31, k-granule g = 2*kk + hi lives in slot g ^ ((r>>2)&3)
    const int sw = (l31 >> 2) & 3;
    const int koff0 = ((0 + hi) ^ sw) * 16, koff1 = ((2 + hi) ^ sw) * 16;
    const int a_row0 = (wn * TN * 32 + l31) * BK;
    const int b_row0 = (BN + wm * TM * 32 + l31) * BK;

    const int nk = a.kpad / BK;
    constexpr int LA = STAGES - 1;                 // stages in flight ahead of the one being consumed
#pragma unroll
    for (int p = 0; p < LA; p++)
        if (p < nk) issue(p);
    for (int s = 0; s < nk; s++) {
        // my own copies of stage s have landed (the up to LA-1 younger stages may stay in flight) ...
        const int rem = nk - s;                    // stages issued and not yet consumed = min(LA, rem)
        if (rem >= LA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((LA - 1) * NI) : "memory");
        else if (LA >= 3 && rem == LA - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((LA >= 3 ? LA - 2 : 0) * NI) : "memory");
        else if (LA >= 4 && rem == LA - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((LA >= 4 ? LA - 3 : 0) * NI) : "memory");
        else if (LA >= 5 && rem == LA - 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((LA >= 5 ? LA - 4 : 0) * NI) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // ... and so have everyone else's; every wave is also done reading the slot stage s+LA will overwrite
        __builtin_amdgcn_s_barrier();
        if (s + LA < nk) issue(s + LA);
        const int8_t* base = smem + (s % STAGES) * STAGE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
            const int ko = kk == 0 ? koff0 : koff1;
            v4i af[TN], bf[TM];
#pragma unroll
            for (int i = 0; i < TN; i++) af[i] = *reinterpret_cast<const v4i*>(base + a_row0 + i * 32 * BK + ko);
#pragma unroll
            for (int j = 0; j < TM; j++) bf[j] = *reinterpret_cast<const v4i*>(base + b_row0 + j * 32 * BK + ko);
#pragma unroll
            for (int i = 0; i < TN; i++)
#pragma unroll
                for (int j = 0; j < TM; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    }

    igemm_epilogue<TM, TN>(a, acc, m0, n0, wm, wn, l31, hi);       // gemm_epilogue.h
}

template <int BN, int STAGES>
static hipError_t launch2(const ConvArgs& a, hipStream_t s, bool is1x1)
{
    const int tiles_n = (a.cout + BN - 1) / BN;
    const int tiles_m = (a.M + 127) / 128;
    const int grid = ((tiles_m + 7) / 8) * 8 * tiles_n;
    const size_t lds = STAGES * (size_t)(128 + BN) * 64;
    if (is1x1) {
        auto k = conv_igemm2_i8_kernel<BN, true, STAGES>;
        if (lds > 65536) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, s, a);
    } else {
        auto k = conv_igemm2_i8_kernel<BN, false, STAGES>;
        if (lds > 65536) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, s, a);
    }
    return hipGetLastError();
}

static int env_int(const char* name, int dflt)        // experiment builds only (csrc/env.h): the product always takes the default
{
    const char* e = exp_env(name);
    return e ? atoi(e) : dflt;
}

// large problems only: enough 128-row tiles to occupy the chip and a K loop long enough for the ring to matter
bool conv_igemm2_applicable(const ConvArgs& a)
{
    static const int mode = env_int("TAMD_IGEMM2", 1);
    if (mode == 0) return false;
    const long blocks = (long)((a.M + 127) / 128) * ((a.cout + 127) / 128);
    return a.zeros != nullptr && a.kpad >= 128 && blocks >= 64 && a.KW <= 16 && a.KH * a.KW <= 128;
}

// 128-wide cout tiles when they still give every CU a block, else 64-wide (twice the blocks)
static bool use_bn128(const ConvArgs& a)
{
    static const int force = env_int("TAMD_IGEMM2_BN", 0);
    if (force == 64) return false;
    if (force == 128) return a.cout > 64;
    const long blocks128 = (long)((a.M + 127) / 128) * ((a.cout + 127) / 128);
    return a.cout > 64 && blocks128 >= 224;
}

const char* conv_igemm2_kernel_name(const ConvArgs& a) { return use_bn128(a) ? "conv_igemm2_i8<128x128x64,ring>" : "conv_igemm2_i8<128x64x64,ring>"; }

hipError_t launch_conv_igemm2(const ConvArgs& a, hipStream_t s)
{
    static const int stages = env_int("TAMD_IGEMM2_STAGES", 3);
    const ConvArgs& b = a;
    const bool is1x1 = (a.KH == 1 && a.KW == 1 && a.SH == 1 && a.SW == 1 && a.PH == 0 && a.PW == 0);
    if (use_bn128(a)) {
        if (stages <= 3) return launch2<128, 3>(b, s, is1x1);
        if (stages == 4) return launch2<128, 4>(b, s, is1x1);
        return launch2<128, 6>(b, s, is1x1);
    }
    if (stages <= 3) return launch2<64, 3>(b, s, is1x1);
    if (stages == 4) return launch2<64, 4>(b, s, is1x1);
    return launch2<64, 6>(b, s, is1x1);
}

}  // namespace tamd
