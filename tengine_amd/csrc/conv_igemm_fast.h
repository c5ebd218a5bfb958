// The software-pipelined int8 implicit-GEMM kernel with the buffer-load operand loader (see conv_igemm.hip for the family and
// the launchers; kept in a header so that tools/exp/igemm_anatomy.hip can instantiate single configurations with stage stamps).
#pragma once
#include <type_traits>

#include "epilogue.h"
#include "gemm_epilogue.h"
#include "kernels.h"

namespace tamd {

typedef int v4i_f __attribute__((ext_vector_type(4)));

#ifdef TAMD_IGEMM_STAMPS
#define IG_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0 && a.dbg_stamps && (i) < 64) a.dbg_stamps[(i)] = clock64(); } while (0)
#else
#define IG_STAMP(i) do { } while (0)
#endif

// ---- the ring kernel with an instruction-lean loader (the generic loader above spends ~100 instructions per 64-deep stage on
// 64-bit address arithmetic, bounds tests and selects, and a wave issues one instruction per ~4.5 cycles: the stage is
// issue-bound at a quarter of the MFMA rate whatever the memory system does -- profiles/r02_instruction_issue_rates*).
// Eligible when a stage never straddles two filter taps (roundup(cin,16) % 64 == 0: every ResNet / VGG-style layer):
//   * operands are fetched with BUFFER loads: per-thread byte offset in a VGPR that never changes, per-stage offset in an
//     SGPR (k position, filter tap) -- no per-load address arithmetic at all for the weights, two VALU for an activation row;
//   * out-of-image taps use the buffer's range check instead of a select: the row's precomputed tap-validity bit is shifted
//     into bit 31 of the offset, which puts it past num_records and the load returns zeros;
//   * the tap walk (kx, ky, offsets) is scalar code.
template <int BM, int BN, int WM, int WN, bool IS1X1, int D>
__global__ __launch_bounds__(256) void conv_igemm_fast_i8_kernel(ConvArgs a)
{
    constexpr int BK = 64, RPP = 64;
    constexpr int PA = (BN + RPP - 1) / RPP, PB = (BM + RPP - 1) / RPP;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    static_assert(WM * WN == 4 && TM >= 1 && TN >= 1 && BN % 64 == 0 && BM % 64 == 0, "tile shape");
    constexpr unsigned OOB = 0x80000000u;

    extern __shared__ __attribute__((aligned(16))) int8_t smem[];       // 2 x (BN + BM) rows of 64 B, granules XOR-swizzled

    IG_STAMP(0);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, hi = lane >> 5;
    const int wm = wave % WM, wn = wave / WM;
    const int tiles_n = (a.cout + BN - 1) / BN, tiles_m = (a.M + BM - 1) / BM;
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int tile_m = (local / tiles_n) * 8 + xcd, tile_n = local % tiles_n;
    if (tile_m >= tiles_m) return;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int ntaps = a.KH * a.KW;

    // buffer resources: weights; activations with the base moved back by the largest negative tap offset, so that per-row
    // offsets are non-negative (bounds are enforced through bit 31 of the offset, not through num_records)
    const int shift = IS1X1 ? 0 : (a.PH * a.W + a.PW) * a.cs_in;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (int)OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x - shift), 0, (int)OOB, 0x00020000);

    const int ohw = a.OH * a.OW;
    const unsigned long long mg_ohw = a.mg_ohw, mg_ow = a.mg_ow;                    // host-computed (graph_plan.hip plan_conv)
    const int q = t & 3, r0 = t >> 2;
    unsigned voffA[PA], voffB[PB], inval[PB];      // inval: bit t set = tap t of this row is outside the image (or the row is)
    int ldsA[PA], ldsB[PB];
#pragma unroll
    for (int p = 0; p < PA; p++) {
        const int row = r0 + p * RPP;
        voffA[p] = (unsigned)((n0 + row) * a.kpad + q * 16);
        ldsA[p] = row * BK + ((q ^ ((row >> 2) & 3)) << 4);
    }
#pragma unroll
    for (int p = 0; p < PB; p++) {
        const int row = r0 + p * RPP, m = m0 + row;
        const bool rv = m < a.M;
        const int mm = rv ? m : 0;
        ldsB[p] = BN * BK + row * BK + ((q ^ ((row >> 2) & 3)) << 4);
        if (IS1X1) {
            voffB[p] = rv ? (unsigned)(mm * a.cs_in + q * 16) : OOB;
            inval[p] = 0;
        } else {
            // m -> (n, oy, ox) without integer divisions: floor(m / d) == (m * ceil(2^40 / d)) >> 40 for m < 2^24, d < 2^16
            // (the launcher checks both; the error term m / 2^40 < 2^-16 < 1 / d)
            const int n = (int)(((unsigned long long)(unsigned)mm * mg_ohw) >> 40), rem = mm - n * ohw;
            const int oy = (int)(((unsigned long long)(unsigned)rem * mg_ow) >> 40), ox = rem - oy * a.OW;
            const int iy0 = oy * a.SH - a.PH, ix0 = ox * a.SW - a.PW;
            voffB[p] = (unsigned)(((n * a.H + iy0) * a.W + ix0) * a.cs_in + shift + q * 16);
            unsigned bad = rv ? 0u : 0xffffffffu;
            int tp = 0;
            for (int ky = 0; ky < a.KH; ky++) {
                const bool rowbad = (unsigned)(iy0 + ky * a.DH) >= (unsigned)a.H;
                for (int kx2 = 0; kx2 < a.KW; kx2++, tp++)
                    if (rowbad || (unsigned)(ix0 + kx2 * a.DW) >= (unsigned)a.W) bad |= 1u << tp;
            }
            inval[p] = bad;
        }
    }
    // scalar walk over K: position inside the tap (ci), tap index, its byte offset
    int ci = 0, tap = 0, kx = 0, off_row = 0, off_tap = 0;
    const int nk_real = a.kpad / BK;                    // stages that carry weights (the rest of a padded ring multiply zeros)
    const int dxs = a.DW * a.cs_in, dys = a.DH * a.W * a.cs_in;

    v4i_f ra[D][PA], rb[D][PB];
    auto gload = [&](int kt, v4i_f (&A)[PA], v4i_f (&B)[PB]) {
#pragma unroll
        for (int p = 0; p < PA; p++) A[p] = __builtin_amdgcn_raw_buffer_load_b128(rw, (int)voffA[p], kt * BK, 0);
        if (IS1X1) {
            const unsigned dead = kt * BK < a.ktot ? 0u : OOB;                        // uniform
#pragma unroll
            for (int p = 0; p < PB; p++) B[p] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(voffB[p] | dead), kt * BK, 0);
        } else {
            const bool live = tap < ntaps;                                             // uniform
            const int sh = 31 - (live ? tap : 0);
            const unsigned dead = live ? 0u : OOB;
            const int soff = off_tap + ci;
#pragma unroll
            for (int p = 0; p < PB; p++)
                B[p] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(voffB[p] | ((inval[p] << sh) & OOB) | dead), soff, 0);
            ci += BK;
            if (ci >= a.ckp) {
                ci = 0; tap++; kx++; off_tap += dxs;
                if (kx == a.KW) { kx = 0; off_row += dys; off_tap = off_row; }
            }
        }
    };
    auto lstore = [&](int buf, const v4i_f (&A)[PA], const v4i_f (&B)[PB]) {
        int8_t* sb = smem + buf * (BM + BN) * BK;
#pragma unroll
        for (int p = 0; p < PA; p++) *reinterpret_cast<v4i_f*>(sb + ldsA[p]) = A[p];
#pragma unroll
        for (int p = 0; p < PB; p++) *reinterpret_cast<v4i_f*>(sb + ldsB[p]) = B[p];
    };

    v16i_t acc[TN][TM];
    igemm_acc_from_bias<TM, TN>(acc, a.bias, n0, wn, hi);      // the epilogue adds nothing (gemm_epilogue.h)

    const int nk = (nk_real + D - 1) / D * D;
    IG_STAMP(1);
    gload(0, ra[0], rb[0]);
    if constexpr (D > 2) gload(1, ra[1], rb[1]);
    if constexpr (D > 3) gload(2, ra[2], rb[2]);
    // fragment addresses inside a stage buffer (fixed per lane)
    int fa[TN][2], fb[TM][2];
#pragma unroll
    for (int kk = 0; kk < 2; kk++) {
#pragma unroll
        for (int i = 0; i < TN; i++) { const int row = (wn * TN + i) * 32 + l31; fa[i][kk] = row * BK + (((kk * 2 + hi) ^ ((row >> 2) & 3)) << 4); }
#pragma unroll
        for (int j = 0; j < TM; j++) { const int row = (wm * TM + j) * 32 + l31; fb[j][kk] = BN * BK + row * BK + (((kk * 2 + hi) ^ ((row >> 2) & 3)) << 4); }
    }
    auto stage = [&](int kt, auto U) {
        constexpr int u = decltype(U)::value, un = (u + D - 1) % D;
        const int buf = kt & 1;
        IG_STAMP(2 + kt);
#ifdef TAMD_IGEMM_STAMPS
        if (!(a.dbg_flags & 2)) lstore(buf, ra[u], rb[u]);
        if (!(a.dbg_flags & 4)) gload(kt + D - 1, ra[un], rb[un]);
        if (!(a.dbg_flags & 2)) __syncthreads();
        if (a.dbg_flags & 1) return;
#else
        lstore(buf, ra[u], rb[u]);
        gload(kt + D - 1, ra[un], rb[un]);
        __syncthreads();
#endif
        const int8_t* sb = smem + buf * (BM + BN) * BK;
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
            v4i_f af[TN], bf[TM];
#pragma unroll
            for (int i = 0; i < TN; i++) af[i] = *reinterpret_cast<const v4i_f*>(sb + fa[i][kk]);
#pragma unroll
            for (int j = 0; j < TM; j++) bf[j] = *reinterpret_cast<const v4i_f*>(sb + fb[j][kk]);
#pragma unroll
            for (int i = 0; i < TN; i++)
#pragma unroll
                for (int j = 0; j < TM; j++) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    };
    for (int kt0 = 0; kt0 < nk; kt0 += D) {
        stage(kt0, std::integral_constant<int, 0>{});
        stage(kt0 + 1, std::integral_constant<int, 1>{});
        if constexpr (D > 2) stage(kt0 + 2, std::integral_constant<int, 2>{});
        if constexpr (D > 3) stage(kt0 + 3, std::integral_constant<int, 3>{});
    }
    IG_STAMP(2 + nk);
    igemm_epilogue<TM, TN>(a, acc, m0, n0, wm, wn, l31, hi);
    IG_STAMP(3 + nk);
}

}  // namespace tamd
