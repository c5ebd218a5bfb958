// int8 implicit-GEMM convolution, lean-loop schedule ("pgemm": Patch / Panel GEMM).
//
// Same arithmetic, operand roles and fused epilogue as conv_igemm.hip (reference chain conv_kernel_x86.c:187-242 im2col,
// :963-1007 pack, :1008-1630 sgemm_i8, :1796-1893 epilogue).  What this kernel changes is everything AROUND the MFMAs: the
// round-2 anatomy (profiles/r02_igemm_anatomy_*) showed 60-100 instructions per wave per 64-deep K stage for 2-8 MFMAs -- the
// kernels were instruction-issue bound at 10 % of the int8 MFMA peak.  Here a stage is ~30 instructions for 8 MFMAs:
//
//   * WEIGHTS are packed by the planner in MFMA fragment order, [cout tile][stage][fragment (32 couts x 32 k)][lane][16 B]:
//     a stage of a block is ONE contiguous BN*64-byte piece, copied by global_load_lds_dwordx4 in 1-KB lane-linear pieces
//     (perfectly coalesced) and read back with ds_read_b128 at lane*16 + immediate -- no swizzle, no address arithmetic.
//   * ACTIVATIONS of a k x k convolution (PATCH): the block keeps the input PATCH of its 128 output pixels -- every input
//     row they touch, halo included, 64 channels at a time -- resident in LDS, granule-major [16-B channel granule][patch
//     pixel], zero-filled outside the image at load time.  A filter tap is then a uniform offset (ky*Wp + kx) * 16 bytes
//     added to the lane's fragment address: no im2col re-reads (each activation byte enters the CU once per 64 channels
//     instead of KH*KW times), no border tests and no pointer arithmetic in the K loop.  The next 64 channels' patch is
//     fetched while the current one is multiplied (two patch buffers).
//   * 1x1 convolutions (!PATCH): activation rows ride in the stage ring next to the weights, row-major with the source-side
//     XOR granule swizzle of conv_igemm2.hip.
//   * LDS-DMA ring of LA stages, counted s_waitcnt vmcnt(N), ONE raw s_barrier per stage; the fragments of stage s+1 are
//     read into a second register set BEFORE the MFMAs of stage s are issued, so LDS latency hides under the matrix pipe
//     even with one wave per SIMD.  Every stage issues the same number of loads (padding loads go to a dump area), so the
//     counted waits are exact.
//   * XCD-aware tile map: an XCD (blocks b, b+8, ..) owns a CONTIGUOUS range of pixel tiles -> its L2 holds one eighth of
//     the input plus the weights; neighbouring tiles share their halo rows in that L2.
#include "conv_pgemm_common.h"

namespace tamd {

template <int BM, int BN, bool PATCH, int LA, int NPC>
__global__ __launch_bounds__(256) void conv_pgemm_i8_kernel(ConvArgs a)
{
    constexpr int TM = BM / 64, TN = BN / 64;            // 32x32 MFMA tiles per wave; waves: 2 along pixels x 2 along couts
    constexpr int NA = BN / 64;                          // 1-KB LDS-DMA pieces of a weight stage per wave
    constexpr int NB = PATCH ? 0 : BM / 64;              // .. of an activation stage (1x1)
    constexpr int NP_ = PATCH ? NPC : 0;                 // patch (or padding) loads per wave per stage
    constexpr int STAGE = BN * 64 + (PATCH ? 0 : BM * 64);
    constexpr int RING = LA * STAGE;
    constexpr int PER_STAGE = NA + NB + NP_;             // loads every wave issues per stage, always
    constexpr int WAITN = (LA - 2) * PER_STAGE;          // loads that may stay in flight when stage s+1 must have landed
    static_assert(LA >= 3 && LA <= 6, "ring depth");
    static_assert(WAITN <= 60, "vmcnt is 6 bits");

    // ONE LDS object (a second __shared__ makes hipcc drain vmcnt(0) in front of every ds_read of a glds pipeline)
    extern __shared__ __attribute__((aligned(16))) int8_t smem[];

    [[maybe_unused]] const int pg_rep = 0;
    PG_STAMP(0);
    PG_STAMP(1);
    const int t = threadIdx.x, lane = t & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);       // scalar: LDS-DMA destinations (M0) stay in SALU
    const int wm = wave & 1, wn = wave >> 1;
    const int tiles_n = (a.cout + BN - 1) / BN, tiles_m = (a.M + BM - 1) / BM;
    const int per_xcd = (tiles_m + 7) >> 3;
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int lm = local / tiles_n, tile_n = local - lm * tiles_n;
    const int tile_m = xcd * per_xcd + lm;
    if (tile_m >= tiles_m) return;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int ns = a.pg_ns;

    // ---- weights: this cout tile's stages, one contiguous BN*64-byte piece each ------------------------------------------
    const int8_t* wt = a.wfrag + (size_t)tile_n * ns * (BN * 64) + lane * 16;
    auto issue_a = [&](int sl, int slot) {
        const int8_t* src = wt + (size_t)sl * (BN * 64);
        int8_t* dst = smem + slot * STAGE;
#pragma unroll
        for (int i = 0; i < NA; i++) PG_GLDS16(src + (i * 4 + wave) * 1024, dst + (i * 4 + wave) * 1024);
    };

    const int ohw = a.OH * a.OW;
    // ---- 1x1: activation rows through the ring (row-major 64-B rows, source-swizzled granules) -------------------------------
    const int8_t* brow[NB > 0 ? NB : 1];
    if constexpr (!PATCH) {
        const int gk = (lane & 3) ^ ((lane >> 4) & 3);
#pragma unroll
        for (int i = 0; i < NB; i++) {
            int m = m0 + (i * 4 + wave) * 16 + (lane >> 2);
            m = m < a.M ? m : a.M - 1;                                   // rows past M repeat the last pixel; never stored
            size_t off;
            if (a.SH == 1 && a.SW == 1) off = (size_t)m * a.cs_in;
            else {
                const int n = pg_div(m, a.mg_ohw), rem = m - n * ohw, oy = pg_div(rem, a.mg_ow), ox = rem - oy * a.OW;
                off = ((size_t)(n * a.H + oy * a.SH) * a.W + ox * a.SW) * a.cs_in;
            }
            brow[i] = a.x + off + gk * 16;
        }
    }
    auto issue_b = [&](int sl, int slot) {
        if constexpr (!PATCH) {
            int8_t* dst = smem + slot * STAGE + BN * 64;
#pragma unroll
            for (int i = 0; i < NB; i++) PG_GLDS16(brow[i] + (size_t)sl * 64, dst + (i * 4 + wave) * 1024);
        }
    };

    // ---- k x k: the input patch of this pixel tile --------------------------------------------------------------------------
    // Virtual padded input: image n occupies rows [n*Hp, (n+1)*Hp), Hp = (OH-1)*SH + (KH-1)*DH + 1, row r <-> input row r - PH;
    // columns [0, Wp), Wp = (OW-1)*SW + (KW-1)*DW + 1, column c <-> input column c - PW.  Output pixel (n, oy, ox) reads, for
    // tap (ky, kx), virtual row n*Hp + oy*SH + ky*DH, column ox*SW + kx*DW.  The patch = virtual rows [R0, R1] of the tile.
    const int Hp = a.pg_hp, Wp = a.pg_wp, npad = a.pg_npad;
    const int ntaps = a.KH * a.KW;
    int R0 = 0;
    const int8_t* psrc[NPC > 0 ? NPC : 1];
    int pstep[NPC > 0 ? NPC : 1];
    if constexpr (PATCH) {
        const int ml = (m0 + BM - 1) < a.M ? (m0 + BM - 1) : a.M - 1;
        const int na = pg_div(m0, a.mg_ohw), oya = pg_div(m0 - na * ohw, a.mg_ow);
        const int nb = pg_div(ml, a.mg_ohw), oyb = pg_div(ml - nb * ohw, a.mg_ow);
        R0 = na * Hp + oya * a.SH;
        const int NP = (nb * Hp + oyb * a.SH + (a.KH - 1) * a.DH - R0 + 1) * Wp;
#pragma unroll
        for (int j = 0; j < NPC; j++) {
            const int pp = (j * 4 + wave) * 64 + lane;
            const int vrow = pg_div(pp, a.mg_wp), col = pp - vrow * Wp;
            const int VR = R0 + vrow, n = pg_div(VR, a.mg_hp);
            const int iy = VR - n * Hp - a.PH, ix = col - a.PW;
            const bool ok = pp < NP && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W && n < a.N;
            psrc[j] = ok ? a.x + ((size_t)(n * a.H + iy) * a.W + ix) * a.cs_in : a.zeros;
            pstep[j] = ok ? 16 : 0;
        }
    }
    // granule g (16 channels) of channel chunk c into patch buffer pb; !real: padding loads (zero page -> dump area)
    auto issue_patch = [&](int c, int g, int pb, bool real) {
        if constexpr (PATCH) {
#pragma unroll
            for (int j = 0; j < NPC; j++) {
                const int q = j * 4 + wave;
                const bool use = real && q * 64 < npad;                                  // wave-uniform
                int8_t* dst = use ? smem + RING + pb * (npad * 64) + (g * npad + q * 64) * 16 : smem + RING + 2 * npad * 64 + wave * 1024;
                const int8_t* src = psrc[j] + (real ? (c * 4 + g) * pstep[j] : 0);
                PG_GLDS16(real ? src : a.zeros, dst);
            }
        }
    };

    // ---- fragment addresses ------------------------------------------------------------------------------------------------
    const int afr = (wn * TN * 2) * 1024 + lane * 16;              // + slot * STAGE + (i*2 + kk) * 1024
    int bfr[TM][2];                                                // + per-stage uniform offset
#pragma unroll
    for (int j = 0; j < TM; j++) {
        if constexpr (PATCH) {
            int m = m0 + (wm * TM + j) * 32 + l31;
            m = m < a.M ? m : a.M - 1;
            const int n = pg_div(m, a.mg_ohw), rem = m - n * ohw, oy = pg_div(rem, a.mg_ow), ox = rem - oy * a.OW;
            const int pp0 = (n * Hp + oy * a.SH - R0) * Wp + ox * a.SW;
#pragma unroll
            for (int kk = 0; kk < 2; kk++) bfr[j][kk] = RING + ((kk * 2 + hi) * npad + pp0) * 16;
        } else {
            const int r = (wm * TM + j) * 32 + l31, sw = (r >> 2) & 3;
#pragma unroll
            for (int kk = 0; kk < 2; kk++) bfr[j][kk] = BN * 64 + r * 64 + (((kk * 2 + hi) ^ sw) << 4);
        }
    }

    v16i_p acc[TN][TM];
    igemm_acc_from_bias<TM, TN>(acc, a.bias, n0, wn, hi);      // the epilogue adds nothing (gemm_epilogue.h)

    v4i_p af[2][TN][2], bf[2][TM][2];
    auto read_frags = [&](auto P, int slot, int boff) {
        constexpr int p = decltype(P)::value;
        const int8_t* ab = smem + afr + slot * STAGE;
#pragma unroll
        for (int i = 0; i < TN; i++)
#pragma unroll
            for (int kk = 0; kk < 2; kk++) af[p][i][kk] = *reinterpret_cast<const v4i_p*>(ab + (i * 2 + kk) * 1024);
#pragma unroll
        for (int j = 0; j < TM; j++)
#pragma unroll
            for (int kk = 0; kk < 2; kk++) bf[p][j][kk] = *reinterpret_cast<const v4i_p*>(smem + bfr[j][kk] + boff);
    };
    auto mfmas = [&](auto P) {
        constexpr int p = decltype(P)::value;
#pragma unroll
        for (int kk = 0; kk < 2; kk++)
#pragma unroll
            for (int i = 0; i < TN; i++)
#pragma unroll
                for (int j = 0; j < TM; j++) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[p][i][kk], bf[p][j][kk], acc[i][j], 0, 0, 0);
    };

    // ---- scalar walk over K.  Stage s = chunk c (64 channels) * ntaps + tap; "cur" = the stage being multiplied, "rd" = the
    // stage whose fragments are being read (cur + 1) ----------------------------------------------------------------------------
    const int nchunks = a.ckp >> 6;
    int t_cur = 0, c_cur = 0;                       // tap / chunk of stage cur
    int kx_r = 0, ky_r = 0, toff_r = 0, pb_r = 0;   // tap walk of stage rd: offset in patch pixels, patch buffer
    const int row_step = a.DH * Wp - a.KW * a.DW;
    auto boff_rd = [&](int slot) { return PATCH ? pb_r * (npad * 64) + toff_r * 16 : slot * STAGE; };
    auto advance_rd = [&]() {
        if constexpr (PATCH) {
            kx_r++; toff_r += a.DW;
            if (kx_r == a.KW) {
                kx_r = 0; ky_r++; toff_r += row_step;
                if (ky_r == a.KH) { ky_r = 0; toff_r = 0; pb_r ^= 1; }
            }
        }
    };

    // ---- prologue: chunk 0's patch, the first LA stages; everything landed before the first fragment read -----------------------
    if constexpr (PATCH) {
#pragma unroll
        for (int g = 0; g < 4; g++) issue_patch(0, g, 0, true);
    }
#pragma unroll
    for (int p = 0; p < LA; p++) {
        const int sl = p < ns ? p : ns - 1;
        issue_a(sl, p);
        issue_b(sl, p);
    }
    // the epilogue's per-channel vectors, fetched now instead of at the start of the epilogue (see the unrolled-taps kernel)
    const int EOFF = RING + (PATCH ? 2 * npad * 64 + 4096 : 0);
    if (wave == 0) {
        const int gq = (lane & 31) % (BN / 4);
        const int8_t* src = lane < 32 ? (const int8_t*)(a.bias + n0) + gq * 16 : (const int8_t*)(a.wscale + n0) + gq * 16;
        PG_GLDS16(src, smem + EOFF);
    }
    PG_STAMP(2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    read_frags(pg_int<0>{}, 0, boff_rd(0));
    advance_rd();
    PG_STAMP(3);

    int slot_w = 0, slot_r = 1 % LA;                // ring slot of stage cur (refilled with stage cur + LA) / of stage rd
    auto body = [&](auto P, int s) {
        constexpr int p = decltype(P)::value;
        // my copies of stage s+1 (and of every older load) have landed; my fragment reads of stage s are complete
        // (the builtin, not inline asm: hipcc's own wait insertion then knows the LDS counter is zero here and does not
        // put an lgkmcnt(0) -- i.e. a wait for the reads of stage s+1 issued below -- in front of this stage's MFMAs)
        __builtin_amdgcn_s_waitcnt(PG_WAITCNT(WAITN));
        // .. and so have everyone else's: slot_w (stage s, now in registers everywhere) may be refilled
        __builtin_amdgcn_s_barrier();
        {
            const int sl = s + LA < ns ? s + LA : ns - 1;          // past the end: a harmless repeat keeps the load count uniform
            if constexpr (PATCH) issue_patch(c_cur + 1, t_cur, (c_cur + 1) & 1, t_cur < 4 && c_cur + 1 < nchunks);
            if (PG_ON(4)) {
                issue_a(sl, slot_w);
                issue_b(sl, slot_w);
            }
        }
        read_frags(pg_int<p ^ 1>{}, slot_r, boff_rd(slot_r));
        advance_rd();
        slot_w = slot_w + 1 == LA ? 0 : slot_w + 1;
        slot_r = slot_r + 1 == LA ? 0 : slot_r + 1;
        if constexpr (PATCH) { t_cur++; if (t_cur == ntaps) { t_cur = 0; c_cur++; } }
        if (PG_ON(1)) mfmas(P);
    };
    for (int s = 0; s < ns; s += 2) {
        body(pg_int<0>{}, s);
        if (s + 1 < ns) body(pg_int<1>{}, s + 1);
    }
    // the ring's trailing (repeat) copies must not outlive the workgroup: its LDS is handed to the next one
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PG_STAMP(4);

    if (PG_ON(2)) igemm_epilogue_src<TM, TN>(a, acc, m0, n0, wm, wn, l31, hi, EpiFromLdsNoBias{smem + EOFF, n0, 128});       // gemm_epilogue.h
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

// ---- host side ---------------------------------------------------------------------------------------------------------------
static int pg_env(const char* name, int dflt)
{
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

enum { PG_ROWS = 0, PG_PATCH = 2 };      // 1x1 | any other k x k (generic patch kernel; 3x3 has its own kernels in conv_pgemm_w.hip)
static int pg_kind(const ConvArgs& a)
{
    return (a.KH == 1 && a.KW == 1 && a.PH == 0 && a.PW == 0) ? PG_ROWS : PG_PATCH;
}

// patch pixels the worst pixel tile of `bm` outputs needs (rows it touches, halo included, times the padded row length)
static int pg_pitch(const ConvArgs& a) { return (a.OW - 1) * a.SW + (a.KW - 1) * a.DW + 1; }

static int pg_patch_pixels(const ConvArgs& a, int bm)
{
    const int Hp = (a.OH - 1) * a.SH + (a.KH - 1) * a.DH + 1, Wp = pg_pitch(a);
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

int conv_pgemm_stages(const ConvArgs& a) { return pg_kind(a) == PG_ROWS ? (a.ckp + 63) / 64 : (a.ckp / 64) * a.KH * a.KW; }

// variants 0 .. 3 (this file): bit 0: BN 128 (else 64); bit 1: BM 64 (else 128).  Variants with bit 4 (16 .. 31, 48 .. 63): the 3x3
// kernels of conv_pgemm_w.hip (same tile bits and packed weights; its own bits 2, 3, 5).  Everything else is unused: the round-3
// unrolled-taps kernel (3x3, its split-K and 3-slot-ring forms) that lived here lost to conv_pgemm_w.hip on every ResNet-50
// shape (profiles/r05_pgemm_anatomy_v2_b3.txt) and was removed in round 5.
static constexpr int PG_LA = 4;            // ring depth
int conv_pgemm_num_variants() { return 64; }
int conv_pgemm_bn(int variant) { return (variant & 1) ? 128 : 64; }
static int pg_bm(int variant) { return (variant & 2) ? 64 : 128; }

static int pg_npad(const ConvArgs& a, int variant)
{
    const int kind = pg_kind(a);
    if (kind == PG_ROWS) return 0;
    return (pg_patch_pixels(a, pg_bm(variant)) + 63) / 64 * 64;
}

static size_t pg_lds(const ConvArgs& a, int variant, int npad)
{
    const int bn = conv_pgemm_bn(variant), bm = pg_bm(variant);
    switch (pg_kind(a)) {
    case PG_ROWS: return (size_t)PG_LA * (bn + bm) * 64 + 1024;
    default: return (size_t)PG_LA * bn * 64 + 2 * (size_t)npad * 64 + 4096 + 1024;
    }
}

bool conv_pgemm_applicable(const ConvArgs& a, int variant)
{
    static const int mode = pg_env("TAMD_PGEMM", 1);
    if (variant & 16) return mode && conv_pgemm_w_applicable(a, variant);
    if (variant >= 4 || !mode || a.zeros == nullptr || a.M >= (1 << 24) || a.OH * a.OW >= 65536) return false;
    const int bn = conv_pgemm_bn(variant), bm = pg_bm(variant), kind = pg_kind(a);
    if (bn == 128 && a.cout <= 64) return false;
    if (bm == 128 && a.M <= 64) return false;
    if (kind == PG_ROWS) return a.ckp >= 32;
    // a 3x3 layer the dedicated kernels take (same pixel tile, four waves, table-driven set-up) is not offered the generic one as well
    if (a.KH == 3 && a.KW == 3 && conv_pgemm_w_applicable(a, 16 | 8 | (variant & 2))) return false;
    const int Hp = (a.OH - 1) * a.SH + (a.KH - 1) * a.DH + 1, Wp = (a.OW - 1) * a.SW + (a.KW - 1) * a.DW + 1;
    if (a.ckp % 64 != 0 || a.cs_in < a.ckp) return false;
    if (a.KH * a.KW < PG_LA + 4 || a.KH * a.KW > 64) return false;        // a chunk's patch loads (4 stages) land LA-1 stages later, inside the chunk
    if ((long)a.N * Hp >= (1L << 24) || Wp >= 65536 || Hp >= 65536) return false;
    if (a.PH < 0 || a.PW < 0) return false;
    const int npad = pg_npad(a, variant);
    if (npad > 512) return false;
    return pg_lds(a, variant, npad) <= 160 * 1024;
}

// fills the patch geometry fields of `a` for `variant` (the planner then attaches the packed weights: conv_pgemm_pack)
void conv_pgemm_prepare(ConvArgs& a, int variant)
{
    if (variant & 16) { conv_pgemm_w_prepare(a, variant); return; }
    a.pg_ns = conv_pgemm_stages(a);
    a.pg_variant = variant;
    if (pg_kind(a) == PG_ROWS) { a.pg_hp = a.pg_wp = 1; a.pg_npad = 0; a.mg_hp = a.mg_wp = 0; return; }
    a.pg_hp = (a.OH - 1) * a.SH + (a.KH - 1) * a.DH + 1;
    a.pg_wp = pg_pitch(a);
    a.pg_npad = pg_npad(a, variant);
    a.mg_hp = ((1ull << 40) + (unsigned)a.pg_hp - 1) / (unsigned)a.pg_hp;
    a.mg_wp = ((1ull << 40) + (unsigned)a.pg_wp - 1) / (unsigned)a.pg_wp;
}

// `w` = the family's [cout_pad][kpad] layout (k = tap * ckp + ci) -> fragment order for cout tiles of `bn`
void conv_pgemm_pack(const ConvArgs& a, const int8_t* w, int cout_pad, int bn, int8_t* out)
{
    const int ns = conv_pgemm_stages(a), ntaps = a.KH * a.KW;
    const bool pw = pg_kind(a) == PG_ROWS;
    const int tiles_n = (a.cout + bn - 1) / bn;
    for (int tn = 0; tn < tiles_n; tn++)
        for (int s = 0; s < ns; s++) {
            const int c = pw ? s : s / ntaps, tap = pw ? 0 : s % ntaps;
            int8_t* st = out + ((size_t)tn * ns + s) * bn * 64;
            for (int col = 0; col < bn; col++) {
                const int co = tn * bn + col;
                for (int kb = 0; kb < 64; kb++) {
                    const int ci = c * 64 + kb;
                    int8_t v = 0;
                    if (co < cout_pad && ci < a.ckp) v = w[(size_t)co * a.kpad + (size_t)tap * a.ckp + ci];
                    const int frag = (col >> 5) * 2 + (kb >> 5), ln = ((kb >> 4) & 1) * 32 + (col & 31);
                    st[(frag * 64 + ln) * 16 + (kb & 15)] = v;
                }
            }
        }
}
// the kernels prefetch up to (two chunk pairs of taps + the ring) past the last stage without a test: readable zeros behind the data
size_t conv_pgemm_packed_bytes(const ConvArgs& a, int bn)
{
    return ((size_t)((a.cout + bn - 1) / bn) * conv_pgemm_stages(a) + 4 * (size_t)a.KH * a.KW + 8) * bn * 64;
}

const char* conv_pgemm_kernel_name(const ConvArgs& a)
{
    static const char* names[2][4] = {{"conv_pgemm_i8<128x64,rows>", "conv_pgemm_i8<128x128,rows>", "conv_pgemm_i8<64x64,rows>", "conv_pgemm_i8<64x128,rows>"},
                                      {"conv_pgemm_i8<128x64,patch>", "conv_pgemm_i8<128x128,patch>", "conv_pgemm_i8<64x64,patch>", "conv_pgemm_i8<64x128,patch>"}};
    if (a.pg_variant & 16) return conv_pgemm_w_kernel_name(a);
    return names[pg_kind(a) == PG_ROWS ? 0 : 1][a.pg_variant & 3];
}

template <typename K>
static hipError_t pg_go(K k, const ConvArgs& a, int bm, int bn, int threads, hipStream_t s)
{
    const int tiles_n = (a.cout + bn - 1) / bn, tiles_m = (a.M + bm - 1) / bm;
    const int grid = ((tiles_m + 7) / 8) * 8 * tiles_n;
    const size_t lds = pg_lds(a, a.pg_variant, a.pg_npad);
    if (lds > 65536) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3(grid), dim3(threads), lds, s, a);
    return hipGetLastError();
}

template <int BM, int BN>
static hipError_t pg_launch_mn(const ConvArgs& a, hipStream_t s)
{
    switch (pg_kind(a)) {
    case PG_ROWS: return pg_go(conv_pgemm_i8_kernel<BM, BN, false, PG_LA, 0>, a, BM, BN, 256, s);
    default:
        if (a.pg_npad <= 256) return pg_go(conv_pgemm_i8_kernel<BM, BN, true, PG_LA, 1>, a, BM, BN, 256, s);
        return pg_go(conv_pgemm_i8_kernel<BM, BN, true, PG_LA, 2>, a, BM, BN, 256, s);
    }
}

hipError_t launch_conv_pgemm(const ConvArgs& a, hipStream_t s)
{
    if (a.pg_variant & 16) return launch_conv_pgemm_w(a, s);
    switch (a.pg_variant & 3) {
    case 0: return pg_launch_mn<128, 64>(a, s);
    case 1: return pg_launch_mn<128, 128>(a, s);
    case 2: return pg_launch_mn<64, 64>(a, s);
    default: return pg_launch_mn<64, 128>(a, s);
    }
}

}  // namespace tamd
