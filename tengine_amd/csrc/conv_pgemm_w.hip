// int8 3x3 implicit-GEMM convolution ("w": WM x WN waves per K group): the successor of round 3's unrolled-taps kernel.
//
// Same arithmetic, operand roles, weight fragment packing and fused epilogue as conv_pgemm.hip (reference chain
// conv_kernel_x86.c:187-242 im2col, :963-1007 pack, :1008-1630 sgemm_i8, :1796-1893 epilogue).  The K loop: the block keeps the
// input PATCH of its pixel tile resident in LDS (granule-major, no halo columns: conv_pgemm.hip), weights arrive through a
// counted-wait LDS-DMA ring, two 64-channel chunks = 18 stages are straight-line code so that ring slot, tap offset, register set
// and every s_waitcnt immediate are compile-time; a stage's issue order is pinned with sched_barrier (wait + barrier, MFMA 0, the
// copies of a later stage, the fragment reads of the NEXT stage spread behind the remaining MFMAs).  What round 5 changed is a block's FIXED cost -- on
// ResNet-50's 3x3 layers at batch 32 a CU sees 0.8 - 1.5 tiles per launch and 48 % of a block's life was set-up, landing and
// epilogue (profiles/r03_pgemm_anatomy_v4_taps_ring6.txt):
//   * SET-UP FROM A TABLE.  The planner writes, per pixel tile, the NHWC pixel index of every patch unit (or -1: a zero row /
//     beyond the patch) and, per output pixel, its patch origin and left / right edge bits (conv_pgemm_w_table).  A lane's
//     geometry is two or three dword loads issued before the first weight copies -- their latency hides under the issue of
//     those copies -- instead of ~25 multiply-high divisions and selects per lane.
//   * EIGHT WAVES per block (4 x 2 or 2 x 4 wave grids, 32 x 32 or 32 x 64 accumulators per wave): half the epilogue per
//     wave, half the set-up copies per wave, two waves per SIMD from ONE block so that a lone block on a CU (res4 / res5
//     launch 196 - 200 tiles on 256 CUs) still overlaps one wave's instruction issue with the other's MFMAs.  With 64-cout
//     tiles a weight stage is four 1-KB pieces: waves 0-3 copy weights, waves 4-7 copy the patch (role-specific counted waits).
//   * BIAS IN THE ACCUMULATORS: they start at the int32 bias (exact: integer addition) instead of zero -- four adds per
//     packed dword less in the epilogue.
//   * A ZERO AREA BEHIND EACH PATCH BUFFER: a tap that falls off the left / right edge is a per-lane choice of the BASE address
//     (per filter column, made once): the zero area's unit with the lane's own bank quad.  Adding the stage's scalar offset
//     keeps the quad, so an edge lane never collides with its neighbours and the K loop has no select -- the single zero
//     unit of conv_pgemm.hip put every edge lane on bank quad 0: a 2-way conflict in one of the two 16-lane groups of most
//     B reads (28 % of the LDS cycles, profiles/r04_pmc_mfma_resnet50_int8_b32.csv).
//   * KS = 2 (two K groups of 2 x 2 waves; res5: 72 stages per tile): both groups ds_add_u32 their partial sums into ONE zeroed
//     row-major LDS tile -- one pass, one barrier -- and ALL eight waves requantise it from there, eight consecutive channels
//     of a pixel per lane (8-byte stores, 64 contiguous bytes per pixel from eight lanes).
#include "conv_pgemm_common.h"

namespace tamd {

// patch pieces a copying wave issues at tap `tp` of a chunk: BS 1: granule tp at taps 0..3; BS 3: granules 0, 1 at taps 0, 1 and
// granules 2, 3 at tap 2 (they must have landed at the chunk's last barrier, tap 6, with D - 4 stages of copies still in flight)
constexpr int pgw_patch_at(int tp, int bs, int npc) { return bs == 1 ? (tp < 4 ? npc : 0) : (tp < 2 ? npc : tp == 2 ? 2 * npc : 0); }
// copies a wave has issued during the stages that may still be in flight at the barrier on top of pair-local stage u: the barrier
// covers the fragment reads of stages u+1 .. u+BS, whose copies were issued D stages ahead -- everything issued during stages
// u+BS+1-D .. u-1 may fly on
constexpr int pgw_inflight(int u, int d, int bs, int nt, int na, int npc)
{
    int w = 0;
    for (int v = u + bs + 1 - d; v <= u - 1; v++) w += na + pgw_patch_at(((v + 4 * nt) % (2 * nt)) % nt, bs, npc);
    return w;
}

// BS: taps per barrier.  1: wait + barrier on top of every 64-deep stage (conv_pgemm.hip's schedule).  3: ONE barrier per filter
// row -- the stages in between only wait for the wave's own fragment reads, so the waves of a block drift up to two stages apart
// and a stage no longer costs the whole wait -> barrier -> issue -> LDS-latency chain (round 5 anatomy: 300 - 400 cycles per stage
// whatever the MFMA count).  The ring then needs D + 2 slots (a copy issued in stage v lands in the slot of stage v - 2, which
// every wave has read before the last barrier) and 7 stages of lead for the same three stages of copies in flight at a barrier.
template <int BM, int BN, int WM, int WN, int KS, int NPC, int BS>
__global__ __launch_bounds__(64 * WM * WN * KS) void conv_pgemm_w_i8_kernel(ConvArgs a)
{
    constexpr int KW = 3, NT = 9, RS = BS == 3 ? 9 : 6, D = BS == 3 ? 7 : 5;
    constexpr int NW = WM * WN;                          // waves of a K group
    constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
    constexpr int STG = BN * 64;                         // bytes of a weight stage
    constexpr int PCS = BN / 16;                         // .. in 1-KB LDS-DMA pieces
    constexpr bool ROLES = PCS < NW;                     // fewer pieces than waves: waves 0-3 copy weights, waves 4-7 the patch
    constexpr int WWAVES = ROLES ? PCS : NW;             // waves that copy weights
    constexpr int NA = PCS / WWAVES;                     // weight pieces per copying wave per stage
    constexpr int PWAVES = ROLES ? NW - PCS : NW;        // waves that copy patch pieces
    constexpr int PC = BN + 4;                           // KS > 1: dword pitch of the row-major partial-sum tile (16-B aligned rows)
    static_assert(TM >= 1 && TN >= 1 && (NW == 4 || NW == 8), "wave grid");
    static_assert(!ROLES || (NW == 8 && PCS == 4), "role split: 8 waves over 64 couts");
    static_assert(BS == 1 || BS == 3, "taps per barrier");
    static_assert((2 * NT) % RS == 0 && RS >= D + (BS == 3 ? 2 : 1) && NT % BS == 0, "ring slots static; a slot is refilled after its last reader passed a barrier");
    static_assert(BS == 3 || NT - (D - 2) > 4, "patch granules land before their chunk");
    static_assert(BS == 1 || (NT - BS) + BS + 1 - D > 2, "BS 3: no patch copy among those still in flight at a chunk's last barrier");
    static_assert(KS == 1 || NW == 4, "K groups are 2 x 2 waves");

    extern __shared__ __attribute__((aligned(16))) int8_t smem[];

    for (int pg_rep = 0; pg_rep < PG_REPS; pg_rep++) {
    if (pg_rep) __syncthreads();
    PG_STAMP(0);
    PG_STAMP(1);
    const int t = threadIdx.x, lane = t & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int grp = KS > 1 ? wave / NW : 0, wq = wave - grp * NW, wm = wq % WM, wn = wq / WM;
    const int tiles_n = a.pg_tiles_n, tiles_m = a.pg_tiles_m;
    const int per_xcd = (tiles_m + 7) >> 3;
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int lm = tiles_n == 1 ? local : (int)__umulhi((unsigned)local, a.mg_tn), tile_n = local - lm * tiles_n;      // local / tiles_n (conv_pgemm_w_applicable checks the range)
    const int tile_m = xcd * per_xcd + lm;
    if (tile_m >= tiles_m) return;
#ifdef TAMD_IGEMM_STAMPS
    // phase-skew experiment (tools/exp/pgemm_anatomy.hip): the blocks of an XCD's second dispatch round start late
    if ((a.dbg_flags & 64) && local >= 32)
        for (int q = 0; q < ((a.dbg_flags >> 16) & 0xff); q++) __builtin_amdgcn_s_sleep(16);
#endif
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int npad = a.pg_npad;                          // multiple of 64, <= 512
    const int BUFS = npad * 64 + a.pg_zarea;             // a patch buffer: four granule planes of npad units, then the zero area (a multiple of 256 B)
    const int GRP = RS * STG + 2 * BUFS;                 // LDS bytes of a K group: [ring][patch buffer 0][patch buffer 1]
    const int EOFF = KS * GRP;                           // bias / multiplier vectors of the cout tile (1 KB)
    const int TOFF = EOFF + 1024;                        // table rows of each wave: (NPC + TM) x 256 B
    [[maybe_unused]] const int AOFF = TOFF + NW * KS * (NPC + TM) * 256;       // KS > 1: the partial-sum tile
    int8_t* const lds = smem + grp * GRP;
    const int nchunks = a.ckp >> 6, nck = nchunks / KS;  // 64-channel chunks; this group takes grp, grp + KS, ..

    // ---- the epilogue's per-channel vectors: one LDS-DMA piece by the last wave (lanes 0-31 the bias granules, lanes 32-63 the multipliers')
    if (wave == NW * KS - 1) {
        const int gq = (lane & 31) % (BN / 4);
        const int8_t* src = lane < 32 ? (const int8_t*)(a.bias + n0) + gq * 16 : (const int8_t*)(a.wscale + n0) + gq * 16;
        PG_GLDS16(src, smem + EOFF);
    }
    // ---- the lane's geometry: its table rows, fetched by LDS-DMA as well (one dword per lane) into the wave's own staging area and
    // issued BEFORE the weight copies, whose issue hides the round trip.  Through LDS-DMA and not as register loads because the
    // wait for a register load is the compiler's: behind the branches of this set-up it becomes vmcnt(0), i.e. a wait for the weight
    // copies too -- the patch copies would start only after the weights have landed.  Here the wait is counted by hand.
    const int* tb = a.pg_tab + (size_t)tile_m * a.pg_ts;
    const bool wcopy = !ROLES || wq < WWAVES, pcopy = !ROLES || wq >= WWAVES;      // wave-uniform
    const int wcq = wq, pcq = ROLES ? wq - WWAVES : wq;
    const int pieces = npad >> 6;
    int8_t* const tl = smem + TOFF + wave * ((NPC + TM) * 256);
    int pq[NPC];
#pragma unroll
    for (int j = 0; j < NPC; j++) {
        // 64-unit pieces of the patch; where the piece count is not a multiple of the copying waves the surplus copies repeat the
        // LAST piece (same bytes to the same place): every copying wave issues the same number of loads, the counted waits stay static
        pq[j] = (j * PWAVES + pcq) < pieces ? (j * PWAVES + pcq) : pieces - 1;
        if (pcopy) PG_GLDS4(tb + pq[j] * 64 + lane, tl + j * 256);
    }
#pragma unroll
    for (int j = 0; j < TM; j++) PG_GLDS4(tb + npad + (wm * TM + j) * 32 + l31, tl + (NPC + j) * 256);

    // ---- weights: the copies of stages 0..D-1 -----------------------------------------------------------------------------------
    const int8_t* wc = a.wfrag + ((size_t)tile_n * a.pg_ns + (size_t)grp * NT) * STG + lane * 16;      // stage 0 of chunk `grp`
    auto stage_off = [](int v) { return ((v / NT) * KS * NT + v % NT) * STG; };
    auto issue_a = [&](int off, int slot) {              // off: bytes from wc (compile-time per call site)
#pragma unroll
        for (int i = 0; i < NA; i++) PG_GLDS16(wc + off + (i * WWAVES + wcq) * 1024, lds + slot * STG + (i * WWAVES + wcq) * 1024);
    };
    if (wcopy) {
#pragma unroll
        for (int p = 0; p < D; p++) issue_a(stage_off(p), p);
    }
    {
        const v4i_p z = {0, 0, 0, 0};
        const int zu = a.pg_zarea >> 4;                  // both zero areas of every K group (<= 1 K units each: conv_pgemm_w_applicable)
#pragma unroll
        for (int b = 0; b < 2 * KS; b++)
            for (int q = t; q < zu; q += 64 * NW * KS) *reinterpret_cast<v4i_p*>(smem + (b >> 1) * GRP + RS * STG + (b & 1) * BUFS + npad * 64 + q * 16) = z;
        if constexpr (KS > 1)
            for (int q = t; q < BM * PC / 4; q += 64 * NW * KS) reinterpret_cast<v4i_p*>(smem + AOFF)[q] = z;
    }

    // the table rows have landed: only the weight copies issued after them may still be in flight
    if (wcopy) __builtin_amdgcn_s_waitcnt(PG_WAITCNT(D * NA));
    else __builtin_amdgcn_s_waitcnt(PG_WAITCNT(0));
    asm volatile("" ::: "memory");
    int tp[NPC];
    unsigned tf[TM];
#pragma unroll
    for (int j = 0; j < NPC; j++) tp[j] = pcopy ? *reinterpret_cast<const int*>(tl + j * 256 + lane * 4) : -1;
#pragma unroll
    for (int j = 0; j < TM; j++) tf[j] = *reinterpret_cast<const unsigned*>(tl + (NPC + j) * 256 + lane * 4);

    // ---- the patch: chunk `grp`, granules 0..3 into buffer 0 ---------------------------------------------------------------------
    const int8_t* psrc[NPC];
    int pstep[NPC];
#pragma unroll
    for (int j = 0; j < NPC; j++) {
        const bool ok = tp[j] >= 0;
        psrc[j] = ok ? a.x + (size_t)(unsigned)tp[j] * (unsigned)a.cs_in : a.zeros;
        pstep[j] = ok ? 16 : 0;
    }
    auto issue_patch = [&](int c, int g, int buf) {      // granule g of chunk c -> patch buffer buf
#pragma unroll
        for (int j = 0; j < NPC; j++)
            PG_GLDS16(psrc[j] + (c * 4 + g) * pstep[j], lds + RS * STG + buf * BUFS + (g * npad + pq[j] * 64) * 16);
    };
    if (pcopy) {
#pragma unroll
        for (int g = 0; g < 4; g++) issue_patch(grp, g, 0);
    }

    int toff[NT];                                        // tap -> byte offset inside a patch granule plane (scalars)
#pragma unroll
    for (int k = 0; k < NT; k++) toff[k] = ((k / KW) * a.DH * a.pg_wp + (k % KW) * a.DW) * 16;
    const int afr = grp * GRP + (wn * TN * 2) * 1024 + lane * 16;
    // B fragment base per (pixel tile, k half, filter COLUMN): the lane's patch unit of tap (0, 0) in granule plane 2 kk + hi -- or,
    // where that column falls off the image for this lane's pixel, the zero-area unit of the same bank quad
    int bfr[TM][2][KW];
#pragma unroll
    for (int j = 0; j < TM; j++) {
        const int pp0 = (int)(tf[j] << 16) >> 16;        // patch unit of tap (0, 0); -PW .. npad
        const int zb = grp * GRP + RS * STG + npad * 64 + ((pp0 << 4) & 0xF0);
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
            const int nat = grp * GRP + RS * STG + ((kk * 2 + hi) * npad + pp0) * 16;
#pragma unroll
            for (int kx = 0; kx < KW; kx++) bfr[j][kk][kx] = ((tf[j] >> (16 + kx)) & 1) ? zb : nat;
        }
    }

    v16i_p acc[TN][TM];
    // fragment registers: fr[set][r], r < 2*TN: weights (cout tile r/2, k half r%2); then activations (pixel tile, k half)
    constexpr int NR = 2 * TN + 2 * TM, NM = 2 * TN * TM;
    v4i_p fr[2][NR];
    auto read_one = [&](auto P, auto R, auto SLOT, auto KX, int boff) {
        constexpr int p = decltype(P)::value, r = decltype(R)::value, slot = decltype(SLOT)::value, kx = decltype(KX)::value;
        if constexpr (r < 2 * TN) fr[p][r] = *reinterpret_cast<const v4i_p*>(smem + afr + slot * STG + r * 1024);
        else {
            constexpr int j = (r - 2 * TN) / 2, kk = (r - 2 * TN) % 2;
            fr[p][r] = *reinterpret_cast<const v4i_p*>(smem + bfr[j][kk][kx] + boff);
        }
    };
    auto mfma_one = [&](auto P, auto Mi) {
        constexpr int p = decltype(P)::value, m = decltype(Mi)::value;
        constexpr int kk = m / (TN * TM), i = (m / TM) % TN, j = m % TM;
        acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fr[p][i * 2 + kk], fr[p][2 * TN + j * 2 + kk], acc[i][j], 0, 0, 0);
    };
#define PGW_SB() __builtin_amdgcn_sched_barrier(0)

    PG_STAMP(2);
    __builtin_amdgcn_s_waitcnt(PG_WAITCNT(0));           // every copy has landed, the zero areas (and the partial-sum tile) are written
    __builtin_amdgcn_s_barrier();
    // accumulators start at the bias (K group 0) / at zero
#pragma unroll
    for (int i = 0; i < TN; i++)
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) {
            int4 b4 = make_int4(0, 0, 0, 0);
            if (grp == 0) b4 = *reinterpret_cast<const int4*>(smem + EOFF + ((wn * TN + i) * 32 + 8 * g4 + 4 * hi) * 4);
#pragma unroll
            for (int j = 0; j < TM; j++) {
                acc[i][j][4 * g4 + 0] = b4.x; acc[i][j][4 * g4 + 1] = b4.y; acc[i][j][4 * g4 + 2] = b4.z; acc[i][j][4 * g4 + 3] = b4.w;
            }
        }
    static_for<0, NR>([&](auto R) { read_one(pg_int<0>{}, R, pg_int<0>{}, pg_int<0>{}, toff[0]); });
    PG_STAMP(3);

    // Two 64-channel chunks = 2 * NT stages, straight-line (conv_pgemm.hip: the stage's issue order and why it is pinned).
    // ROLE 0: every wave copies weights and patch pieces; 1: this wave copies weights only; 2: patch pieces only.  The role is a
    // compile-time constant of the loop (one scalar branch in front of it), so the counted waits stay immediates.
    auto kloop = [&](auto ROLE) {
        constexpr int role = decltype(ROLE)::value;
        auto half = [&](auto H, int ci) {
            constexpr int h = decltype(H)::value;
            const int cnext = ci + h + 1 < nck ? ci + h + 1 : nck - 1;      // patch granules issued now: the group's NEXT chunk (the last one re-fetches itself)
            static_for<0, NT>([&](auto T) {
                constexpr int tpos = decltype(T)::value, u = h * NT + tpos, p = u & 1;
                // copies issued by the previous D-2 stages may stay in flight; everything older -- stage u+1's weights included -- has
                // landed.  A patch-only wave: the next chunk's granules (issued in taps 0-3) are first read by the fragment reads of
                // this chunk's LAST stage.  (Every form also waits for the wave's own fragment reads of the previous stage.)
                if constexpr (tpos % BS == 0) {
                    constexpr int W = role == 0 ? pgw_inflight(u, D, BS, NT, NA, NPC) : role == 1 ? (D - BS - 1) * NA : (tpos == NT - BS ? 0 : 63);
                    __builtin_amdgcn_s_waitcnt(PG_WAITCNT(W));
                    __builtin_amdgcn_s_barrier();
                } else
                    __builtin_amdgcn_s_waitcnt(PG_WAITCNT(63));          // the wave's own fragment reads only
                PGW_SB();
                mfma_one(pg_int<p>{}, pg_int<0>{});
                PGW_SB();
                if constexpr (role != 1) {
                    if constexpr (BS == 1 && tpos < 4) issue_patch(grp + cnext * KS, tpos, h ^ 1);
                    if constexpr (BS == 3 && tpos < 3) issue_patch(grp + cnext * KS, tpos, h ^ 1);
                    if constexpr (BS == 3 && tpos == 2) issue_patch(grp + cnext * KS, 3, h ^ 1);
                }
                if constexpr (role != 2) issue_a(stage_off(u + D), (u + D) % RS);
                PGW_SB();
                constexpr int un = (u + 1) % (2 * NT), tn = un % NT;                           // the stage whose fragments are read now
                const int boff = (un / NT) * BUFS + toff[tn];
                constexpr int slots = NM > 4 ? NM / 2 : NM - 1;                    // MFMAs 1 .. slots get fragment reads in front of them
                constexpr int per = (NR + slots - 1) / slots;
                static_for<1, NM>([&](auto Mi) {
                    constexpr int m = decltype(Mi)::value;
                    static_for<0, per>([&](auto Q) {
                        constexpr int r = (m - 1) * per + decltype(Q)::value;
                        if constexpr (m <= slots && r < NR) read_one(pg_int<p ^ 1>{}, pg_int<r>{}, pg_int<(u + 1) % RS>{}, pg_int<tn % KW>{}, boff);
                    });
                    PGW_SB();
                    mfma_one(pg_int<p>{}, Mi);
                    PGW_SB();
                });
            });
        };
        for (int ci = 0; ci < nck; ci += 2) {
            half(pg_int<0>{}, ci);
            if (ci + 1 < nck) half(pg_int<1>{}, ci);
            wc += (size_t)2 * KS * NT * STG;
        }
    };
    if constexpr (!ROLES) kloop(pg_int<0>{});
    else if (wcopy) kloop(pg_int<1>{});
    else kloop(pg_int<2>{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the ring's trailing copies must not outlive the workgroup
    PG_STAMP(4);

    if constexpr (KS == 1) {
        igemm_epilogue_src<TM, TN>(a, acc, m0, n0, wm, wn, l31, hi, EpiFromLdsNoBias{smem + EOFF, n0, 128});
    } else {
        // both K groups add their partial sums into the zeroed row-major tile [pixel][cout] (pitch PC dwords) ..
        int* const tile = reinterpret_cast<int*>(smem + AOFF);
#pragma unroll
        for (int i = 0; i < TN; i++)
#pragma unroll
            for (int j = 0; j < TM; j++) {
                const int px = (wm * TM + j) * 32 + l31, cb = (wn * TN + i) * 32 + 4 * hi;
#pragma unroll
                for (int e = 0; e < 16; e++)
                    __hip_atomic_fetch_add(tile + px * PC + cb + 8 * (e >> 2) + (e & 3), acc[i][j][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        __builtin_amdgcn_s_waitcnt(PG_WAITCNT(0));
        __builtin_amdgcn_s_barrier();
        // .. and every wave requantises its share: eight consecutive channels of one pixel per lane
        const Rq rq = a.rq;
        const EpiFromLdsNoBias src{smem + EOFF, n0, 128};
        const bool wide8 = ((a.ldc | a.c_off) & 7) == 0 && (!a.elt.res || ((a.elt.res_ldc | a.elt.res_c_off) & 7) == 0);
        const float inv_elt = a.elt.res ? __fdiv_rn(1.0f, a.elt.out_scale) : 1.f;
        const float inv_relu = (a.elt.res && a.elt.relu) ? __fdiv_rn(1.0f, a.elt.relu_out_scale) : 1.f;
        constexpr int UPR = BN / 8, THREADS = 64 * NW * KS;
#pragma unroll
        for (int it = 0; it < BM * UPR / THREADS; it++) {
            const int un = it * THREADS + t, px = un / UPR, c8 = (un % UPR) * 8;
            const v4i_p v0 = *reinterpret_cast<const v4i_p*>(tile + px * PC + c8), v1 = *reinterpret_cast<const v4i_p*>(tile + px * PC + c8 + 4);
            const int c0 = n0 + c8, m = m0 + px;
            unsigned p0 = requant4(v0[0], v0[1], v0[2], v0[3], src.scale4(c0), c0, rq);
            unsigned p1 = requant4(v1[0], v1[1], v1[2], v1[3], src.scale4(c0 + 4), c0 + 4, rq);
            if (m < a.M && c0 < a.c_limit) {
                const bool two = c0 + 4 < a.c_limit;
                if (a.elt.res) {
                    const int8_t* rp = a.elt.res + (size_t)m * a.elt.res_ldc + a.elt.res_c_off + c0;
                    const unsigned r0 = *reinterpret_cast<const unsigned*>(rp), r1 = two ? *reinterpret_cast<const unsigned*>(rp + 4) : 0u;
                    if (a.elt.thr > 0.f) { p0 = elt_sum4_fold(p0, r0, a.elt); p1 = elt_sum4_fold(p1, r1, a.elt); }
                    else { p0 = fuse_elt4(p0, r0, a.elt, inv_elt, inv_relu); p1 = fuse_elt4(p1, r1, a.elt, inv_elt, inv_relu); }
                }
                int8_t* yp = a.y + (size_t)m * a.ldc + a.c_off + c0;
                if (wide8 && two) *reinterpret_cast<uint2*>(yp) = make_uint2(p0, p1);
                else {
                    *reinterpret_cast<unsigned*>(yp) = p0;
                    if (two) *reinterpret_cast<unsigned*>(yp + 4) = p1;
                }
            }
        }
    }
    PG_STAMP(5);
    PG_STAMP(6);
#ifdef TAMD_IGEMM_STAMPS
    if (a.dbg_stamps && threadIdx.x == 0) {
        unsigned xcc = 0, hwid = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        a.dbg_stamps[(size_t)blockIdx.x * 8 + 7] = ((long long)xcc << 32) | hwid;
    }
#endif
    }
}

// ---- host side -----------------------------------------------------------------------------------------------------------------
// variant = 16 | tile bits of conv_pgemm.hip (bit 0: BN 128, bit 1: BM 64) | bit 2: KS 2 (two K groups of 2 x 2 waves)
//              | bit 3: 2 x 2 waves (conv_pgemm.hip's wave grid with this file's set-up) | bit 5: one barrier per three taps
static int pw_bn(int v) { return (v & 1) ? 128 : 64; }
static int pw_bm(int v) { return (v & 2) ? 64 : 128; }
static int pw_ks(int v) { return (v & 4) ? 2 : 1; }
static bool pw_w4(int v) { return (v & 8) != 0; }
static bool pw_b3(int v) { return (v & 32) != 0; }

static int pw_patch_pixels(const ConvArgs& a, int bm)
{
    const int Hp = (a.OH - 1) * a.SH + (a.KH - 1) * a.DH + 1, Wp = a.W;
    const int ohw = a.OH * a.OW;
    int worst = 0;
    for (int m0 = 0; m0 < a.M; m0 += bm) {
        const int ml = std::min(m0 + bm - 1, a.M - 1);
        const int r0 = (m0 / ohw) * Hp + ((m0 % ohw) / a.OW) * a.SH;
        const int r1 = (ml / ohw) * Hp + ((ml % ohw) / a.OW) * a.SH + (a.KH - 1) * a.DH;
        worst = std::max(worst, (r1 - r0 + 1) * Wp);
    }
    return worst;
}

static int pw_npad(const ConvArgs& a, int v) { return (pw_patch_pixels(a, pw_bm(v)) + 63) / 64 * 64; }

// zero area behind a patch buffer: the largest tap offset past a bank-quad unit (< 256 B), rounded to 256 B
static int pw_zarea(const ConvArgs& a) { return ((2 * a.DH * a.W + 2 * a.DW) * 16 + 256 + 255) / 256 * 256; }

static size_t pw_lds(const ConvArgs& a, int v, int npad)
{
    const int bn = pw_bn(v), bm = pw_bm(v), ks = pw_ks(v);
    return (size_t)ks * ((pw_b3(v) ? 9 : 6) * (size_t)bn * 64 + 2 * ((size_t)npad * 64 + pw_zarea(a))) + 1024 + 8 * 4 * 256 + (ks > 1 ? (size_t)bm * (bn + 4) * 4 : 0);
}

bool conv_pgemm_w_applicable(const ConvArgs& a, int v)
{
    const char* e = getenv("TAMD_PGEMM_W");              // read at every prerun: A/B runs build both forms in one process (tools/exp/ab_step.py)
    if ((e && atoi(e) == 0) || !(v & 16) || a.zeros == nullptr) return false;
    if (a.KH != 3 || a.KW != 3 || a.ckp % 64 != 0 || a.cs_in < a.ckp || a.PH < 0 || a.PW < 0 || a.PW > 255) return false;
    const int bn = pw_bn(v), bm = pw_bm(v), ks = pw_ks(v);
    if (bn == 128 && a.cout <= 64) return false;
    if (bm == 128 && a.M <= 64) return false;
    if (ks == 2 && (bn == 128 || pw_w4(v) || (a.ckp / 64) % 2 != 0)) return false;
    if (!pw_w4(v) && ks == 1 && bm == 64 && bn == 64) return false;              // 64 x 64 has four 32 x 32 wave tiles: KS 2 or the 2 x 2 form
    if (pw_w4(v) && bn == 128) return false;                                    // the 2 x 2 form is instantiated for 64-cout tiles only
    if (pw_b3(v) && !(pw_w4(v) || (ks == 2 && bm == 64) || (bn == 128 && bm == 128))) return false;      // one barrier per filter row: the forms instantiated below
#ifndef TAMD_EXPERIMENTS
    // forms that lost on every ResNet-50 shape (profiles/r05_pgemm_anatomy_*.txt) are only built for tools/exp/pgemm_anatomy.hip:
    // the two K groups (ks2w), eight waves over 64 couts with copy roles (128x64 w8) and the 2 x 4 wave grid (64x128 w8)
    if (ks == 2 || (!pw_w4(v) && !(bn == 128 && bm == 128))) return false;
#endif
    if ((long)a.N * a.H * a.W >= (1L << 31) || a.M >= (1 << 24)) return false;
    const long tiles_m = (a.M + bm - 1) / bm, tiles_n = (a.cout + bn - 1) / bn;
    if (((tiles_m + 7) / 8 + 1) * tiles_n >= (1L << 31) / tiles_n) return false;     // local / tiles_n by multiply-high
    const int npad = pw_npad(a, v);
    if (npad > 512 || a.W >= 32768) return false;
    return pw_lds(a, v, npad) <= 160 * 1024;
}

void conv_pgemm_w_prepare(ConvArgs& a, int v)
{
    a.pg_ns = (a.ckp / 64) * 9;
    a.pg_variant = v;
    a.pg_hp = (a.OH - 1) * a.SH + 2 * a.DH + 1;
    a.pg_wp = a.W;
    a.pg_npad = pw_npad(a, v);
    a.pg_zarea = pw_zarea(a);
    a.pg_tiles_m = (a.M + pw_bm(v) - 1) / pw_bm(v);
    a.pg_tiles_n = (a.cout + pw_bn(v) - 1) / pw_bn(v);
    a.mg_tn = a.pg_tiles_n > 1 ? (unsigned)(((1ull << 32) + (unsigned)a.pg_tiles_n - 1) / (unsigned)a.pg_tiles_n) : 0u;      // one cout tile: the kernel does not divide
    a.pg_ts = a.pg_npad + pw_bm(v);
    a.mg_hp = a.mg_wp = 0;
}

// the per-tile geometry table (see the kernel): tiles_m x (npad + BM) dwords
void conv_pgemm_w_table(const ConvArgs& a, std::vector<int>& out)
{
    const int bm = pw_bm(a.pg_variant), npad = a.pg_npad, ts = a.pg_ts, Hp = a.pg_hp, Wp = a.pg_wp, ohw = a.OH * a.OW;
    out.assign((size_t)a.pg_tiles_m * ts, -1);
    for (int tm = 0; tm < a.pg_tiles_m; tm++) {
        int* tb = out.data() + (size_t)tm * ts;
        const int m0 = tm * bm, ml = std::min(m0 + bm - 1, a.M - 1);
        const int na = m0 / ohw, oya = (m0 % ohw) / a.OW, nb = ml / ohw, oyb = (ml % ohw) / a.OW;
        const int R0 = na * Hp + oya * a.SH;
        const int NP = (nb * Hp + oyb * a.SH + 2 * a.DH - R0 + 1) * Wp;
        for (int pp = 0; pp < npad && pp < NP; pp++) {
            const int vrow = pp / Wp, ix = pp % Wp, VR = R0 + vrow, n = VR / Hp, iy = VR - n * Hp - a.PH;
            if (iy >= 0 && iy < a.H && n < a.N) tb[pp] = (n * a.H + iy) * a.W + ix;
        }
        for (int p = 0; p < bm; p++) {
            const int m = std::min(m0 + p, a.M - 1);
            const int n = m / ohw, oy = (m % ohw) / a.OW, ox = m % a.OW;
            const int ix0 = ox * a.SW - a.PW;
            const int pp0 = (n * Hp + oy * a.SH - R0) * Wp + ix0;
            unsigned e = 0;
            for (int kx = 0; kx < 3; kx++)
                if ((unsigned)(ix0 + kx * a.DW) >= (unsigned)a.W) e |= 1u << kx;
            tb[npad + p] = (int)(((unsigned)pp0 & 0xffffu) | (e << 16));
        }
    }
}

const char* conv_pgemm_w_kernel_name(const ConvArgs& a)
{
    static const char* w8[2][4] = {{"conv_pgemm_i8<128x64,3x3,w8>", "conv_pgemm_i8<128x128,3x3,w8>", "?", "conv_pgemm_i8<64x128,3x3,w8>"},
                                   {"?", "conv_pgemm_i8<128x128,3x3,w8b3>", "?", "?"}};
    static const char* k2[2][4] = {{"conv_pgemm_i8<128x64,3x3,ks2w>", "?", "conv_pgemm_i8<64x64,3x3,ks2w>", "?"}, {"?", "?", "conv_pgemm_i8<64x64,3x3,ks2wb3>", "?"}};
    static const char* w4[2][4] = {{"conv_pgemm_i8<128x64,3x3,w4t>", "?", "conv_pgemm_i8<64x64,3x3,w4t>", "?"},
                                   {"conv_pgemm_i8<128x64,3x3,w4b3>", "?", "conv_pgemm_i8<64x64,3x3,w4b3>", "?"}};
    const int v = a.pg_variant, b3 = pw_b3(v);
    return pw_ks(v) == 2 ? k2[b3][v & 3] : pw_w4(v) ? w4[b3][v & 3] : w8[b3][v & 3];
}

template <typename K>
static hipError_t pw_go(K k, const ConvArgs& a, int threads, hipStream_t s)
{
    const int grid = ((a.pg_tiles_m + 7) / 8) * 8 * a.pg_tiles_n;
    const size_t lds = pw_lds(a, a.pg_variant, a.pg_npad);
    if (lds > 65536) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3(grid), dim3(threads), lds, s, a);
    return hipGetLastError();
}

#define PW_K(BM, BN, WM, WN, KS, NPC, BS) conv_pgemm_w_i8_kernel<BM, BN, WM, WN, KS, NPC, BS>
hipError_t launch_conv_pgemm_w(const ConvArgs& a, hipStream_t s)
{
    const int v = a.pg_variant, npad = a.pg_npad;
    const bool b3 = pw_b3(v), wide = npad > 256;        // wide: two patch pieces per copying wave of a four-wave copy crew
#ifdef TAMD_EXPERIMENTS
    if (pw_ks(v) == 2) {
        if (v & 2) {
            if (b3) return wide ? pw_go(PW_K(64, 64, 2, 2, 2, 2, 3), a, 512, s) : pw_go(PW_K(64, 64, 2, 2, 2, 1, 3), a, 512, s);
            return wide ? pw_go(PW_K(64, 64, 2, 2, 2, 2, 1), a, 512, s) : pw_go(PW_K(64, 64, 2, 2, 2, 1, 1), a, 512, s);
        }
        return wide ? pw_go(PW_K(128, 64, 2, 2, 2, 2, 1), a, 512, s) : pw_go(PW_K(128, 64, 2, 2, 2, 1, 1), a, 512, s);
    }
    if (!pw_w4(v) && (v & 3) == 0) return wide ? pw_go(PW_K(128, 64, 4, 2, 1, 2, 1), a, 512, s) : pw_go(PW_K(128, 64, 4, 2, 1, 1, 1), a, 512, s);
    if (!pw_w4(v) && (v & 3) == 3) return pw_go(PW_K(64, 128, 2, 4, 1, 1, 1), a, 512, s);
#endif
    if (pw_w4(v)) {
        if (v & 2) {
            if (b3) return wide ? pw_go(PW_K(64, 64, 2, 2, 1, 2, 3), a, 256, s) : pw_go(PW_K(64, 64, 2, 2, 1, 1, 3), a, 256, s);
            return wide ? pw_go(PW_K(64, 64, 2, 2, 1, 2, 1), a, 256, s) : pw_go(PW_K(64, 64, 2, 2, 1, 1, 1), a, 256, s);
        }
        if (b3) return wide ? pw_go(PW_K(128, 64, 2, 2, 1, 2, 3), a, 256, s) : pw_go(PW_K(128, 64, 2, 2, 1, 1, 3), a, 256, s);
        return wide ? pw_go(PW_K(128, 64, 2, 2, 1, 2, 1), a, 256, s) : pw_go(PW_K(128, 64, 2, 2, 1, 1, 1), a, 256, s);
    }
    if ((v & 3) == 1) return b3 ? pw_go(PW_K(128, 128, 4, 2, 1, 1, 3), a, 512, s) : pw_go(PW_K(128, 128, 4, 2, 1, 1, 1), a, 512, s);
    return hipErrorInvalidValue;
}

}  // namespace tamd
