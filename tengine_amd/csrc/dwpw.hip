// int8 depthwise 3x3 (stride 1) -> pointwise 1x1 in ONE launch: the depthwise result is the GEMM's B operand and never leaves LDS.
//
// Arithmetic = dwconv.hip followed by the implicit-GEMM family, value for value: the depthwise output is formed as the int8 tensor
// the reference stores between the two nodes (conv_dw_hcl_x86.c:42-95,97-269 / batch > 1: conv_kernel_ref_int8.c:42-177, with the
// epilogue formula the planner selected for that node; bias, activation, requantisation, saturation), then multiplied
// (conv_kernel_x86.c:1008-1630 sgemm_i8) and requantised with the pointwise node's own constants (:1796-1893).
//
// Why (round 4): MobileNet-v1 at batch 64 runs its 14x14 block unfused -- five (depthwise 8.4 us + pointwise 10.2-11.5 us) pairs, each
// writing and re-reading a 6.4 MB tensor -- because the pointwise -> depthwise fusion (pwdw.hip) loses there to halo recomputation
// (profiles/r04_pwdw_cfgs_b64.txt).  The other direction has no halo: a block owns 64 output pixels and ALL output channels.
//
//   * block = 512 threads, tile = 4 rows x 16 columns of the depthwise OUTPUT map, the rows counted through the whole batch (row index
//     = image * OH + oy, so a 14-row map leaves no partial row tiles; columns >= OW are dead lanes);
//   * K loop over the channels in stages of 128: (a) every thread computes one (4 adjacent pixels x 4 channels) unit of the depthwise
//     layer -- dwconv.hip's scheme: dword loads with lanes along channels, 4x4 byte transposes, v_dot4 rows -- requantises it and
//     writes four dwords into the stage's B buffer, granule-major [16-B channel granule][pixel] (a B fragment read is then 32
//     consecutive 16-B units: conflict-free); (b) wave w multiplies output channels [64 w, 64 w + 64) x 64 pixels: 16
//     v_mfma_i32_32x32x32_i8 per stage, A fragments straight from global memory in fragment order ([32-cout tile][32-k step][lane][16 B],
//     packed by the planner), requested behind the MFMAs of the previous stage; two B buffers, one barrier per stage;
//   * epilogue: requant4 + half-wave regroup -> one 16-byte store per lane and tile (gemm_epilogue.h's scheme on this tile's pixel map).
//
// Round 6 (the ISA of round 4's form, `s_waitcnt` by `s_waitcnt`): every stage fetched the depthwise weights / bias / multipliers of its 128
// channels from global memory at the top of its depthwise phase, BEHIND the eight A-fragment loads it had just issued -- loads return in
// order, so the `vmcnt(0)` in front of the first requantisation waited for the fragments too: one exposed L2 round trip of 8 KB per wave
// and stage; and the epilogue fetched the pointwise multipliers one vector at a time, sixteen `global_load_dwordx4` + `vmcnt(0)` pairs
// in a row.  Now the per-channel constants of BOTH nodes are copied into LDS once (20 bytes per depthwise channel + 4 per output
// channel, requested before anything else), every global load of the stage loop is unconditional (a stage index past the end is
// clamped: the compiler's wait counting then knows how many younger loads are in flight instead of falling back to vmcnt(0)), and
// the A fragments of stage s + 1 are requested k step by k step right behind the MFMAs that read stage s's.
#include <type_traits>

#include "dw_common.h"
#include "epilogue.h"
#include "kernels.h"

namespace tamd {

typedef int v4i_dp __attribute__((ext_vector_type(4)));
typedef int v16i_dp __attribute__((ext_vector_type(16)));

#ifdef TAMD_DWPW_STAMPS      // tools/exp/dwpw_anatomy.hip: stage time stamps (100 MHz wall clock) of wave 0 of every block, ablation switches
#define DWPW_STAMP(i) do { if (threadIdx.x == 0 && a.stamps) a.stamps[(size_t)blockIdx.x * 16 + (i)] = wall_clock64(); } while (0)
#define DWPW_ON(bit) (!(a.ablate & (bit)))
#else
#define DWPW_STAMP(i) do { } while (0)
#define DWPW_ON(bit) true
#endif

// NW = waves along the output channels (cout <= 64 * NW); 512 threads always (the depthwise stage needs 512 units per 128 channels)
constexpr int DWPW_KST = 128;                            // channels per stage
typedef int8_t dwpw_bs_t[DWPW_KST / 16][64][16];         // one B buffer: [channel granule][pixel][16 B]

// dynamic LDS of a block: [2 B buffers][dw weights 3 x cw dwords][dw bias cw][dw multipliers cw][pw multipliers 64 NW]
__host__ __device__ inline size_t dwpw_lds_bytes(int cw, int nw) { return 2 * sizeof(dwpw_bs_t) + (size_t)20 * cw + (size_t)256 * nw; }

// WIN: both requantisations (the depthwise node's and the pointwise node's) in the one-binade form of epilogue.h; checked once by the kernel
template <int NW, int WIN>
__device__ __forceinline__ void dwpw_body(const DwPwArgs& a, int8_t* smem)
{
    constexpr int KST = DWPW_KST;
    dwpw_bs_t* bs = reinterpret_cast<dwpw_bs_t*>(smem);
    int8_t* const cW = smem + 2 * sizeof(dwpw_bs_t);     // depthwise weights, DwArgs::w's own layout: [3 rows][cw channels][4 B]
    int8_t* const cB = cW + (size_t)12 * a.cw;           // depthwise bias [cw] int32
    int8_t* const cS = cB + (size_t)4 * a.cw;            // depthwise multipliers [cw] float
    int8_t* const pS = cS + (size_t)4 * a.cw;            // pointwise multipliers [64 NW] float
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), l31 = lane & 31, hi = lane >> 5;
    const int rows_total = a.N * a.OH;
    const int gr0 = blockIdx.x * 4;                      // first row (image * OH + oy) of the tile
    DWPW_STAMP(0);

    // ---- the per-channel constants of both nodes -> LDS: requested before everything else (they are needed first), stored once the
    // other loads of the prologue are on their way.  16-byte unit i of the LDS block [cW | cB | cS | pS] comes from one of four arrays
    const int nw4 = 3 * a.cw / 4, nb4 = a.cw / 4, cunits = nw4 + 2 * nb4 + NW * 16;      // (cw is a multiple of 16)
    auto const_unit = [&](int i) {
        i = min(i, cunits - 1);                          // unconditional load; units past the end are not stored
        const uint4* src = i < nw4 ? reinterpret_cast<const uint4*>(a.dw_w) + i
                         : i < nw4 + nb4 ? reinterpret_cast<const uint4*>(a.dw_bias) + (i - nw4)
                         : i < nw4 + 2 * nb4 ? reinterpret_cast<const uint4*>(a.dw_wscale) + (i - nw4 - nb4)
                         : reinterpret_cast<const uint4*>(a.pw_wscale) + (i - nw4 - 2 * nb4);
        return *src;
    };
    const uint4 cu0 = const_unit(t), cu1 = const_unit(t + 512);

    // ---- this thread's depthwise unit: pixels (row ur, columns 4 ucg .. 4 ucg + 3), channels 4 cq .. 4 cq + 3 of the stage ----
    // t >> 5 = 2 wave + hi: the tile row ur = wave >> 1 is the same for the whole wave, so everything that hangs on it -- image, input
    // rows, their validity -- is scalar: a tap's address is (scalar row pointer + stage offset) + (per-lane pixel / channel offset), the
    // form a global load takes as SGPR base + 32-bit VGPR offset with no vector address arithmetic per load
    const int cq = t & 31, ur = wave >> 1, ucg = 2 * (wave & 1) + hi;
    const int ugr = min(gr0 + ur, rows_total - 1);       // (a dead row repeats the last live one: computed, never stored)
    const int un = ugr / a.OH, uoy = ugr - un * a.OH;
    const int8_t* xn = a.x + (size_t)un * a.H * a.W * a.cs_in;
    const int ixb = 4 * ucg - a.PW;
    int pixoff[6];
    unsigned colok = 0;
#pragma unroll
    for (int p = 0; p < 6; p++) {
        const int ix = ixb + p;
        const bool ok = (unsigned)ix < (unsigned)a.W;
        pixoff[p] = (ok ? ix : 0) * a.cs_in;
        colok |= ok ? 1u << p : 0u;
    }
    const int8_t* xrow[3];
    unsigned rowok = 0;
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const int iy = uoy - a.PH + r;
        const bool ok = (unsigned)iy < (unsigned)a.H;
        xrow[r] = xn + (size_t)((ok ? iy : 0) * a.W) * a.cs_in;
        rowok |= ok ? 1u << r : 0u;
    }
    const Rq drq = a.dw_rq;
    const int nst = (a.C + KST - 1) / KST;

    // depthwise of stage `st` -> B buffer `buf`.  Channels past C (a ragged last stage) read the tensor's zero padding / repeat: their
    // weights are zero rows of the pointwise fragments, so whatever they hold is multiplied by 0.  A stage index past the end repeats the
    // last stage's addresses (requested, never used): every load of the loop is unconditional.
    unsigned raw[2][3][6];                               // two sets: the taps of stage s + 2 are requested before stage s + 1 is computed
    auto dw_load = [&](auto D, int st) {
        constexpr int d = decltype(D)::value;
        const int sb = min(st, nst - 1) * KST;                                      // scalar
        const int lc = min(sb + 4 * cq, a.cw - 4) - sb;                             // keep the dword inside the padded channel row
#pragma unroll
        for (int p = 0; p < 6; p++) {
            const unsigned vo = (unsigned)(pixoff[p] + lc);
            if (!DWPW_ON(16)) continue;
#pragma unroll
            for (int r = 0; r < 3; r++) raw[d][r][p] = *reinterpret_cast<const unsigned*>(xrow[r] + sb + vo);
        }
    };
    auto dw_compute = [&](auto D, int st, int buf) {
        constexpr int d = decltype(D)::value;
        const int c = min(st * KST + 4 * cq, a.cw - 4);
        unsigned wrow[3][4];
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const uint4 v = *reinterpret_cast<const uint4*>(cW + ((size_t)r * a.cw + c) * 4);
            wrow[r][0] = v.x; wrow[r][1] = v.y; wrow[r][2] = v.z; wrow[r][3] = v.w;
        }
        const int4 b4 = *reinterpret_cast<const int4*>(cB + (size_t)c * 4);
        const float4 s4 = *reinterpret_cast<const float4*>(cS + (size_t)c * 4);
        int acc[4][4];
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int k = 0; k < 4; k++) acc[j][k] = k == 0 ? b4.x : k == 1 ? b4.y : k == 2 ? b4.z : b4.w;      // the dot chain starts at the bias
#pragma unroll
        for (int r = 0; r < 3; r++) {
            if (!DWPW_ON(2)) { acc[0][0] += (int)(raw[d][r][0] ^ raw[d][r][1] ^ raw[d][r][2] ^ raw[d][r][3] ^ raw[d][r][4] ^ raw[d][r][5]); continue; }
            unsigned d0[4], d1[4], f0[4], f1[4];
#pragma unroll
            for (int q = 0; q < 4; q++) d0[q] = ((rowok >> r) & (colok >> q) & 1u) ? raw[d][r][q] : 0u;
            d1[0] = ((rowok >> r) & (colok >> 4) & 1u) ? raw[d][r][4] : 0u;
            d1[1] = ((rowok >> r) & (colok >> 5) & 1u) ? raw[d][r][5] : 0u;
            d1[2] = 0u; d1[3] = 0u;
            transpose4x4(d0, f0);
            transpose4x4(d1, f1);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                acc[0][k] = __builtin_amdgcn_sdot4((int)f0[k], (int)wrow[r][k], acc[0][k], false);
                acc[1][k] = __builtin_amdgcn_sdot4((int)f0[k], (int)(wrow[r][k] << 8), acc[1][k], false);
                acc[2][k] = __builtin_amdgcn_sdot4((int)__builtin_amdgcn_alignbyte(f1[k], f0[k], 2), (int)wrow[r][k], acc[2][k], false);
                acc[3][k] = __builtin_amdgcn_sdot4((int)__builtin_amdgcn_alignbyte(f1[k], f0[k], 3), (int)wrow[r][k], acc[3][k], false);
            }
        }
        const int cl = 4 * cq;                           // channel inside the stage
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const unsigned pk = requant4<WIN>(acc[j][0], acc[j][1], acc[j][2], acc[j][3], s4, c, drq);
            *reinterpret_cast<unsigned*>(&bs[buf][cl >> 4][ur * 16 + 4 * ucg + j][cl & 15]) = pk;
        }
    };

    // ---- pointwise: wave w -> output channels [64 w, 64 w + 64): two 32-row tiles, A fragments one stage ahead in registers ----
    const int nk32 = nst * (KST / 32);
    const int8_t* wf = a.pw_wfrag + ((size_t)(wave * 2) * nk32 * 64 + lane) * 16;       // tile 2w, step 0, this lane
    v4i_dp af[2][KST / 32];                              // [cout tile][k step]: ONE stage of fragments
    auto a_load_ks = [&](int st, int ks) {
        if (!DWPW_ON(8) && st > 0) return;
#pragma unroll
        for (int i = 0; i < 2; i++)
            af[i][ks] = *reinterpret_cast<const v4i_dp*>(wf + ((size_t)i * nk32 + (size_t)min(st, nst - 1) * (KST / 32) + ks) * 1024);
    };
    // the accumulators start at the pointwise bias (C/D layout of the 32x32 MFMA: register e of lane (pixel, hi) is channel
    // 8 (e >> 2) + 4 hi + (e & 3) of the tile): the epilogue requantises them as they are
    v16i_dp acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) {
            const int4 b4 = wave < NW ? *reinterpret_cast<const int4*>(a.pw_bias + (wave * 2 + i) * 32 + 8 * g4 + 4 * hi) : make_int4(0, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 2; j++) { acc[i][j][4 * g4 + 0] = b4.x; acc[i][j][4 * g4 + 1] = b4.y; acc[i][j][4 * g4 + 2] = b4.z; acc[i][j][4 * g4 + 3] = b4.w; }
        }
    // stage `buf` multiplied; NEXT >= 0: the fragments of stage NEXT requested into the same registers, k step by k step, right behind
    // the MFMAs that read them (the hardware orders the overwrite behind the read; the fetch then has the rest of the stage to land)
    auto mma = [&](int buf, int next) {
        if (wave >= NW) return;
#pragma unroll
        for (int ks = 0; ks < KST / 32; ks++) {
            if (!DWPW_ON(1)) { if (next >= 0) a_load_ks(next, ks); continue; }
            v4i_dp bf[2];
#pragma unroll
            for (int j = 0; j < 2; j++) bf[j] = DWPW_ON(32) ? *reinterpret_cast<const v4i_dp*>(&bs[buf][ks * 2 + hi][j * 32 + l31][0]) : af[j][ks];
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[i][ks], bf[j], acc[i][j], 0, 0, 0);
            if (next >= 0) a_load_ks(next, ks);
        }
    };

    // ---- the stages: stage s multiplies while the fragments of stage s + 1 are on their way, then the depthwise layer of stage s + 1 is
    // computed into the other B buffer -- its ~150 VALU instructions run beside the matrix pipe -- and the taps of stage s + 3 are
    // requested; one barrier per stage
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    if (wave < NW) {
#pragma unroll
        for (int ks = 0; ks < KST / 32; ks++) a_load_ks(0, ks);
    }
    dw_load(I0{}, 0);
    dw_load(I1{}, 1);
    if (t < cunits) reinterpret_cast<uint4*>(cW)[t] = cu0;
    if (t + 512 < cunits) reinterpret_cast<uint4*>(cW)[t + 512] = cu1;
    for (int i = t + 1024; i < cunits; i += 512) reinterpret_cast<uint4*>(cW)[i] = const_unit(i);      // more than 768 depthwise channels
    __syncthreads();                                     // the constants are in LDS
    DWPW_STAMP(1);
    dw_compute(I0{}, 0, 0);
    dw_load(I0{}, 2);
    __syncthreads();
    DWPW_STAMP(2);
    // half(st, D, buf): stage st (B buffer buf) multiplied, stage st + 1 produced from tap set D into the other buffer
    auto half = [&](int st, auto D, int buf) {
        mma(buf, st + 1);
        DWPW_STAMP(8 + st);
        dw_compute(D, st + 1, buf ^ 1);
        dw_load(D, st + 3);
        __syncthreads();
        DWPW_STAMP(3 + st);
    };
    int st = 0;
    for (; st + 2 < nst; st += 2) { half(st, I1{}, 0); half(st + 1, I0{}, 1); }
    if (st + 1 < nst) { half(st, I1{}, 0); mma(1, -1); }
    else mma(0, -1);
    DWPW_STAMP(6);
    if (wave >= NW) return;

    // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane & 31 (pixel), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) (cout) ----
    const Rq prq = a.pw_rq;
    float4 s4s[2][4];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) s4s[i][g4] = *reinterpret_cast<const float4*>(pS + (size_t)((wave * 2 + i) * 32 + 8 * g4 + 4 * hi) * 4);
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int p = j * 32 + l31, pr = p >> 4, pc = p & 15;
        const int gr = gr0 + pr;
        const bool live = gr < rows_total && pc < a.OW;
        int8_t* yp = a.y + ((size_t)gr * a.OW + pc) * a.ldc + a.c_off;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int cb = (wave * 2 + i) * 32;
            unsigned pk[4];
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) {
                const int c = cb + 8 * g4 + 4 * hi;
                if (!DWPW_ON(4)) { pk[g4] = (unsigned)(acc[i][j][4 * g4 + 0] ^ acc[i][j][4 * g4 + 1] ^ acc[i][j][4 * g4 + 2] ^ acc[i][j][4 * g4 + 3]); continue; }
                pk[g4] = requant4<WIN>(acc[i][j][4 * g4 + 0], acc[i][j][4 * g4 + 1], acc[i][j][4 * g4 + 2], acc[i][j][4 * g4 + 3], s4s[i][g4], c, prq);
            }
            half_wave_regroup(pk);
            const int c16 = cb + hi * 16;
            if (live && c16 < a.c_limit) *reinterpret_cast<uint4*>(yp + c16) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        }
    }
    DWPW_STAMP(7);
}

// (Round 6 also built the depthwise layer as diagonal-weight MFMAs -- v_mfma_i32_16x16x64_i8, k = kx * 16 + channel, the B operand one
// ds_read_b128 of an LDS copy of the input patch per tap row -- bit-exact, ~105 instead of ~260 vector instructions per wave and stage, and
// 8 % SLOWER in the graph: the patch is re-read from LDS 12 KB per wave and stage on top of the pointwise B reads, and this kernel's
// phases add up instead of overlapping.  tools/exp/patches/r06_dwpw_depthwise_on_mfma.patch, profiles/r06_ab_dwpw_depthwise_on_mfma_*.)
template <int NW>
__global__ __launch_bounds__(512) void dwpw_i8_kernel(DwPwArgs a)
{
    extern __shared__ __attribute__((aligned(16))) int8_t dwpw_smem[];
    if (rq_win(a.dw_rq) && rq_win(a.pw_rq)) dwpw_body<NW, 1>(a, dwpw_smem);
    else dwpw_body<NW, 0>(a, dwpw_smem);
}

// depthwise 3x3 stride 1 (any padding the map allows) feeding a pointwise 1x1 stride-1 convolution with no padding; output channels
// in whole 64-channel wave slices up to 512, destination on 16-channel granularity (16-byte stores); OW <= 16 (one tile row spans the map);
// at most 2048 (padded) channels: their constants live in LDS (dwpw_lds_bytes <= 64 KB)
bool dwpw_applicable(const DwArgs& d, const ConvArgs& p)
{
    if (d.S != 1 || d.OW > 16 || d.C % 4 != 0 || d.cw < 16 || d.cw % 16 != 0 || d.cw > 2048) return false;
    if (p.KH != 1 || p.KW != 1 || p.SH != 1 || p.SW != 1 || p.PH != 0 || p.PW != 0 || p.elt.res) return false;
    if (p.cout % 64 != 0 || p.cout > 512 || p.cin != d.C) return false;
    if (((p.c_limit | p.c_off | p.ldc) & 15) != 0 || p.c_limit < p.cout) return false;
    return p.N == d.N && p.H == d.OH && p.W == d.OW;
}

// A-fragment order of the pointwise weights: [32-cout tile][32-k step][lane = half * 32 + row][16 B]; `w` = [cout][cin] int8 (OIHW, 1x1);
// k steps padded to whole 128-channel stages with zeros
size_t dwpw_packed_bytes(int cout, int cin) { return (size_t)((cout + 31) / 32) * ((cin + 127) / 128 * 4) * 1024; }
void dwpw_pack(const int8_t* w, int cout, int cin, int8_t* out)
{
    const int nk32 = (cin + 127) / 128 * 4;
    for (size_t i = 0; i < dwpw_packed_bytes(cout, cin); i++) out[i] = 0;
    for (int co = 0; co < cout; co++)
        for (int k = 0; k < cin; k++) {
            const int tile = co >> 5, row = co & 31, ks = k >> 5, half = (k >> 4) & 1;
            out[(((size_t)tile * nk32 + ks) * 64 + half * 32 + row) * 16 + (k & 15)] = w[(size_t)co * cin + k];
        }
}

hipError_t launch_dwpw(const DwPwArgs& a, hipStream_t s)
{
    const int blocks = (a.N * a.OH + 3) / 4;
    switch (a.cout / 64) {
    case 1: hipLaunchKernelGGL(dwpw_i8_kernel<1>, dim3(blocks), dim3(512), dwpw_lds_bytes(a.cw, 1), s, a); break;
    case 2: hipLaunchKernelGGL(dwpw_i8_kernel<2>, dim3(blocks), dim3(512), dwpw_lds_bytes(a.cw, 2), s, a); break;
    case 3: hipLaunchKernelGGL(dwpw_i8_kernel<3>, dim3(blocks), dim3(512), dwpw_lds_bytes(a.cw, 3), s, a); break;
    case 4: hipLaunchKernelGGL(dwpw_i8_kernel<4>, dim3(blocks), dim3(512), dwpw_lds_bytes(a.cw, 4), s, a); break;
    case 5: hipLaunchKernelGGL(dwpw_i8_kernel<5>, dim3(blocks), dim3(512), dwpw_lds_bytes(a.cw, 5), s, a); break;
    case 6: hipLaunchKernelGGL(dwpw_i8_kernel<6>, dim3(blocks), dim3(512), dwpw_lds_bytes(a.cw, 6), s, a); break;
    case 7: hipLaunchKernelGGL(dwpw_i8_kernel<7>, dim3(blocks), dim3(512), dwpw_lds_bytes(a.cw, 7), s, a); break;
    case 8: hipLaunchKernelGGL(dwpw_i8_kernel<8>, dim3(blocks), dim3(512), dwpw_lds_bytes(a.cw, 8), s, a); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace tamd
