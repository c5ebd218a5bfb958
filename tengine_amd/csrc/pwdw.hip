// Pointwise (1x1) int8 convolution fused with the node that consumes it: the following depthwise 3x3 convolution
// (stride 1 | 2) or global pooling.  One launch instead of two, the intermediate tensor never leaves the CU.
//
// Replaces, pair by pair, the reference chain  conv_hcl_run (im2col_int8 -> input_pack4_int8 -> sgemm_i8 -> requant,
// conv_kernel_x86.c:187-242, :963-1007, :1008-1630, :1796-1893)  ->  convdw3x3s1/s2_int8_sse
// (conv_dw_hcl_x86.c:97-269, :271-445; batch > 1: ref_conv_int8, conv_kernel_ref_int8.c:42-177)  or  ->  the int8
// pooling reference (pooling_kernel_ref_int8.c:84-189).  Bit-exact: the pointwise result is requantised to int8 with
// the very epilogue the stand-alone kernel uses (epilogue.h) before the tail consumes it -- same integers, same float
// operations, only the HBM round trip of the intermediate tensor is gone.
//
// Why: at batch 1 (BASELINE configs[1]) every launch costs a 1.25 us dependent-launch gap + 0.15 us dispatch ramp,
// and a lone wave issues one instruction per 4 cycles, so a short kernel's body is instruction-count bound
// (profiles/r02_launch_chain2_device_clock_anatomy.txt).  MobileNet-v1 is 13 x (depthwise, pointwise): fusing
// pointwise_i with depthwise_{i+1} removes 12 of its 29 launches -- and the depthwise needs no halo exchange when a
// block owns a CHANNEL slice, because depthwise never mixes channels.
//
// Block = (16-channel slice of the pointwise output) x (TH x TW tile of depthwise outputs) of one image.
//   1. pointwise GEMM over the tile's input region ((TH-1)*S+3) x ((TW-1)*S+3), clipped to the image:
//      v_mfma_i32_16x16x64_i8, A = weights (rows = the 16 channels; pre-packed in fragment order so a load instruction
//      reads 1 KB contiguous), B = NHWC activations (cols = 16 region pixels, 64 contiguous K bytes each); the waves
//      split the 16-pixel tiles, all K steps of a tile in flight at once, the next tile's loads issued before this
//      tile's MFMAs.  Fused requantisation -> int8 -> LDS [region pixel][16 channels]; pixels outside the image are
//      the depthwise zero padding (LDS pre-zeroed).
//   2. depthwise 3x3 from LDS exactly as dwconv.hip does from memory: lane = 4 channels x a strip of outputs, 4x4 byte
//      transposes + one v_dot4_i32_i8 per filter row, its own requantisation, dword stores to the NHWC output.
//      (global pooling: per-channel sum / max over the region, the reference's float sequence, one dword per 4 channels)
// The same block structure serves two more jobs, selected by template parameters:
//   * PROD 1: the producer is the network's FIRST convolution, gathered straight from the NCHW graph input (<= 64 taps:
//     MobileNet / SSD conv1 3x3x3) through a k -> (plane offset, dy, dx) table -- conv1 + depthwise 2_1 become one launch;
//   * MODE 4: no tail at all -- a small-map pointwise conv / 1x1-map FC whose tile results go from the accumulator
//     registers to memory (the batch-1 layers that have no depthwise behind them: MobileNet fc7).
#include "dw_common.h"
#include "epilogue.h"
#include "kernels.h"

namespace tamd {

typedef int v4i __attribute__((ext_vector_type(4)));

#ifdef TAMD_PWDW_STAMPS      // tools/exp/pwdw_anatomy.hip: stage time stamps (s_memrealtime) of wave 0 of every block
#define PWDW_STAMP(i) do { if (threadIdx.x == 0 && a.stamps) a.stamps[(size_t)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + (i)] = wall_clock64(); } while (0)
#define PWDW_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define PWDW_STAMP(i) do { } while (0)
#define PWDW_DRAIN() do { } while (0)
#endif

// MODE 0: global pooling tail; 1: depthwise stride 1, two outputs per lane; 2: depthwise stride 2; 3: depthwise stride 1,
// one output per lane (small tiles: more lanes busy, shorter chains).  STEPS: 64-deep K steps held in registers; the
// planner pads the weight panel with zero steps to a multiple of STEPS and the activation buffers carry slack behind
// their last pixel, so the K loop needs no predicates (bytes read past cin meet zero weights).  CHUNKED: the panel is
// several times STEPS deep (K > 1024), weights are re-read per tile.
// A lone wave issues one instruction every 4 cycles and these launches are a few hundred instructions long, so the code
// is written for instruction count: host-folded requantisation constants, no integer divisions, no predicated loads.
// MODE 4: no tail (results stored from registers).  PROD 1: first-layer gather from the NCHW graph input (STEPS == 1).
//
// CHAINED: the block runs inside pwdw_chain_kernel -- several of these layers in ONE launch, ordered by flags instead of
// kernel boundaries (see there).  Everything that does not depend on the previous layer (weight fragments, bias / scale
// vectors, the LDS zero fill) is done BEFORE the block waits for its producers; the block publishes its own flag at the end.
struct ChainDep {
    const int* wait_counter;   // finished-block counter of the previous layer of the chain (nullptr: nothing to wait for)
    int wait_target;           // its value once every block of that layer has finished in THIS run (counters only grow: epoch * blocks)
    int* my_counter;           // this layer's counter
    int* err;                  // set when a wait gives up (bounded spin: a logic error must not hang the GPU)
};

// Tensors handed from layer to layer INSIDE a chained launch never rely on the (per-XCD, mutually incoherent) L2s: a chained
// block stores its results write-through (agent-scope stores, sc1) and reads its input with agent-scope loads (sc1: served
// by the memory side, never by a stale L1 / L2 line).  Ordering then needs no cache maintenance at all: producer = wait for
// its own stores (vmcnt) -> barrier -> one relaxed fetch-add; consumer = poll -> barrier.  (A release / acquire pair per block
// -- buffer_wbl2 / buffer_inv -- serialises on the XCD's L2: measured 10-19 us per layer with 128-512 blocks a layer.)
__device__ __forceinline__ void chain_wait(const ChainDep& d)
{
    if (d.wait_counter) {
        if (threadIdx.x == 0) {          // ONE uncached load per block and round: hundreds of blocks watch the same word
            bool done = false;
            for (int spin = 0; spin < (1 << 17) && !done; spin++) {
                // counters and targets wrap together (unsigned arithmetic): compare their distance
                done = (int)((unsigned)__hip_atomic_load(d.wait_counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - (unsigned)d.wait_target) >= 0;
                if (!done) __builtin_amdgcn_s_sleep(8);
            }
            if (!done) *d.err = 1;
        }
        __syncthreads();                 // also keeps the compiler from moving the input loads above the poll
    }
}

__device__ __forceinline__ void chain_signal(const ChainDep& d)
{
    // coherent stand-alone launch (pwdw_i8_coh_kernel): the kernel boundary orders the layers.  No explicit wait for the
    // write-through stores here: S_ENDPGM does it ("the hardware implicitly executes S_WAITCNT 0 before executing this
    // instruction", GCN3 / Vega / CDNA ISA manuals, SOPP S_ENDPGM; vmcnt counts stores until they are acknowledged, for sc1 stores by
    // the memory side), a dispatch completes when its last wave has ended, and the next packet carries the barrier bit.
    if (!d.my_counter) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this thread's write-through stores have reached the memory side
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(d.my_counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void chain_store(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// WIN: both requantisations (the pointwise conv's and its depthwise consumer's) in the one-binade form of epilogue.h; the kernels
// check the two windows once (pwdw_windows)
// SL: 16-channel slices per block (1 | 2 | 4).  The slices of a block share everything that is per pixel tile -- the tile's address arithmetic, its
// activation load (PROD 1: the patch gather), the loop control -- which is 30 of the ~50 vector instructions a tile costs: the early
// MobileNet pairs at batch 64 are bound by exactly those (73-86 % VALU busy, profiles/r05_pmc_sq_activity_mobilenet_v1_int8_b64.csv).
// The intermediate tensor is slice-major in LDS ([slice][region pixel][16 channels]), so the depthwise phase only widens its channel index.
template <int STEPS, int MODE, bool CHUNKED, int PROD, bool CHAINED, int WIN = 0, int SL = 1>
__device__ __forceinline__ void pwdw_block(const PwDwArgs& a, unsigned* __restrict__ inter, const int bx, const int by, const int bz,
                                           const int nthreads, const ChainDep& dep)
{
    constexpr int S = MODE == 2 ? 2 : 1;
    static_assert(SL == 1 || ((SL == 2 || SL == 4) && !CHUNKED && MODE >= 1 && MODE <= 3), "several slices: register-resident K, depthwise tails");
    constexpr bool PINGPONG = !CHUNKED && STEPS <= 8 && PROD == 0;      // a second operand buffer: the next tile's loads fly under this tile's MFMAs
    PWDW_STAMP(0);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, nwaves = nthreads >> 6;
    const int l15 = lane & 15, kb = lane >> 4;
    // block -> XCD is round-robin over the linear block index, and every XCD has its own L2: with the channel slice in grid.x
    // an XCD sees every tile (the input is fetched once per XCD, the weights once in total); with the tile in grid.x it sees
    // every slice of its own tiles (input once in total, weights once per XCD).  The planner puts the larger operand on the
    // "once in total" side (PMC: the early layers read 6-8x their input with slices first, profiles/r02_traffic_*).
    const int slice = a.tile_major ? bz : bx, tx = a.tile_major ? bx : by;
    int ty = a.tile_major ? by : bz, n = 0;
    if (a.N > 1) { n = ty / a.tiles_y; ty -= n * a.tiles_y; }
    const int c_base = slice * 16 * SL;

    // ---- loads that depend on nothing but the block index go out first ----------------------------------------
    v4i bias_v[SL];                                        // the MFMA chain of a tile starts at the bias: no addition in the epilogue
    float4 ps[SL];
#pragma unroll
    for (int k = 0; k < SL; k++) {
        const int4 pb = *reinterpret_cast<const int4*>(a.bias + c_base + 16 * k + 4 * kb);
        bias_v[k] = v4i{pb.x, pb.y, pb.z, pb.w};
        ps[k] = *reinterpret_cast<const float4*>(a.wscale + c_base + 16 * k + 4 * kb);
    }
    const int8_t* wfp = a.wf + ((size_t)slice * SL * a.nsteps * 64 + lane) * 16;      // nsteps: padded to a multiple of STEPS; slice k: + k * nsteps KB
    v4i af[SL][STEPS];
    if (!CHUNKED) {
#pragma unroll
        for (int k = 0; k < SL; k++)
#pragma unroll
            for (int u = 0; u < STEPS; u++) af[k][u] = *reinterpret_cast<const v4i*>(wfp + ((size_t)k * a.nsteps + u) * 1024);
    }
    const int cq = t & (4 * SL - 1);                       // depthwise phase: this thread's channel quad of the block's 16 * SL channels
    const int c0 = c_base + cq * 4;
    unsigned wrow[3][4], wsh[3][4];
    int4 db = {0, 0, 0, 0};
    float4 ds = {1.f, 1.f, 1.f, 1.f};
    if (MODE != 0 && MODE != 4) {
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const uint4 v = *reinterpret_cast<const uint4*>(a.dw_w + ((size_t)r * a.cw + c0) * 4);
            wrow[r][0] = v.x; wrow[r][1] = v.y; wrow[r][2] = v.z; wrow[r][3] = v.w;
        }
        db = *reinterpret_cast<const int4*>(a.dw_bias + c0);
        ds = *reinterpret_cast<const float4*>(a.dw_wscale + c0);
    }

    // ---- geometry of this block (uniform) ---------------------------------------------------------------------
    int iy0 = 0, ix0 = 0, th = 1, tw = 1, vy0 = 0, vx0 = 0, vy1 = a.H, vx1 = a.W, RW = a.W;
    if (MODE == 4) {                 // rows [ty*TH, ty*TH + TH) of the map, full width
        vy0 = ty * a.TH; vy1 = min(vy0 + a.TH, a.H);
    } else if (MODE != 0) {
        const int oy0 = ty * a.TH, ox0 = tx * a.TW;
        iy0 = oy0 * S - a.PH; ix0 = ox0 * S - a.PW;
        th = min(a.TH, a.OH - oy0); tw = min(a.TW, a.OW - ox0);
        vy0 = max(iy0, 0); vx0 = max(ix0, 0);
        vy1 = min(iy0 + (th - 1) * S + 3, a.H); vx1 = min(ix0 + (tw - 1) * S + 3, a.W);
        RW = a.RW;
    }
    const int VW = vx1 - vx0, VP = (vy1 - vy0) * VW;
    const float inv_vw = __builtin_amdgcn_rcpf((float)VW);
    const int ntiles = (VP + 15) >> 4;
    const Rq rq = a.rq;                  // folded on the host (graph_plan.hip: host_rq)
    const int8_t* xn = PROD == 0 ? a.x + (size_t)n * a.H * a.W * a.cs_in + kb * 16
                                 : a.x + (size_t)n * a.in_C * a.in_H * a.in_W;
    // PROD 1: the four patch rows (c, ky) of this lane's 16 K bytes: k = row * 4 + kx, a row is FOUR consecutive input bytes
    // (kx = 0..KW-1 real, the rest meets zero weights) -- one unaligned dword load per row instead of KW byte gathers
    // (conv_first.hip's row trick).  Table entry: (c*in_H*in_W + ky*DH*in_W) | ky*DH << 28; padding rows are 0.
    unsigned rows[4];
    if (PROD == 1) {
        const uint4 v = *reinterpret_cast<const uint4*>(a.taps + kb * 4);
        rows[0] = v.x; rows[1] = v.y; rows[2] = v.z; rows[3] = v.w;
    }

    // region pixel of lane l15 in tile i: LDS dword index (negative: no pixel) and input address; lanes past the last
    // pixel (and waves past the last tile) re-read the last pixel instead of branching around their loads
    int piy = 0, pix = 0;      // map coordinates of the located pixel (MODE 4 stores, PROD 1 gathers)
    auto locate = [&](int i, const int8_t*& xp) -> int {
        const int v = i * 16 + l15, vc = min(v, VP - 1);
        // v / VW without an integer division: (v + 0.5) / VW is at least 0.5 / VW away from an integer, the float error
        // (v < 2^14, 1-ulp rcp) is orders of magnitude smaller
        const int vy = (int)(((float)vc + 0.5f) * inv_vw), vx = vc - vy * VW;
        const int iy = vy0 + vy, ix = vx0 + vx;
        piy = iy; pix = ix;
        xp = PROD == 0 ? xn + (unsigned)((iy * a.W + ix) * a.cs_in) : xn;
        const int slot = MODE == 4 ? (iy * a.W + ix) : ((iy - iy0) * RW + (ix - ix0)) * 4 + kb;
        return v < VP ? slot : -1;
    };
    // chained: agent-scope (sc1) buffer loads of the previous layer's write-through output
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, 0x7fffffff, 0x00020000);
    auto load_b = [&](const int8_t* xp, v4i (&bf)[STEPS], int chunk) {
        if (PROD == 0 && CHAINED) {
            const int voff = (int)(xp - a.x);
#pragma unroll
            for (int u = 0; u < STEPS; u++) bf[u] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, voff, (chunk * STEPS + u) * 64, 16 /* sc1 */);
        } else if (PROD == 0) {
#pragma unroll
            for (int u = 0; u < STEPS; u++) bf[u] = *reinterpret_cast<const v4i*>(xp + (chunk * STEPS + u) * 64);
        } else {
            // patch rows of conv output pixel (piy, pix).  Column handling is the same for every row: bytes left of the image
            // are shifted in as zeros, bytes right of it masked; rows above / below the image are zero.  All four loads are
            // unconditional (clamped addresses) so they fly together.
            const int iyb = piy * a.fSH - a.fPH, ixb = pix * a.fSW - a.fPW;
            const int sft = max(-ixb, 0), xs = max(ixb, 0), nvalid = a.in_W - ixb;
            const bool colok = nvalid > 0 && sft < 4;
            const unsigned cmask = nvalid < 4 ? (1u << (8 * max(nvalid, 0))) - 1u : ~0u;
            const int base = iyb * a.in_W + xs;
            unsigned raw[4];
            bool ok[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int iy = iyb + (int)(rows[j] >> 28);
                ok[j] = colok && (unsigned)iy < (unsigned)a.in_H;
                __builtin_memcpy(&raw[j], xp + (ok[j] ? base + (int)(rows[j] & 0xffffffu) : 0), 4);
            }
#pragma unroll
            for (int j = 0; j < 4; j++) bf[0][j] = ok[j] ? (int)((raw[j] << (8 * sft)) & cmask) : 0;
        }
    };
    const int soff = (a.RH * a.RW + 4) * 4;               // dwords between the slices' copies of the region
    auto finish = [&](const v4i& acc, int slot, int k = 0) {
        const unsigned p = requant4<WIN>(acc[0], acc[1], acc[2], acc[3], ps[k], c_base + 16 * k + 4 * kb, rq);      // (the accumulator started at the bias)
        if (MODE == 4) {
            if (slot >= 0 && c_base + 4 * kb < a.c_limit) {
                unsigned* dst = reinterpret_cast<unsigned*>(a.y + ((size_t)n * a.H * a.W + slot) * a.ldc + a.c_off + c_base + 4 * kb);
                if (CHAINED) chain_store(dst, p); else *dst = p;
            }
        } else if (slot >= 0)
            inter[(SL > 1 ? k * soff : 0) + slot] = p;
    };
    // one pixel tile: all K steps of every slice against the operand in `b`
    auto tile = [&](const v4i (&b)[STEPS], int slot) {
#pragma unroll
        for (int k = 0; k < SL; k++) {
            v4i acc = bias_v[k];
#pragma unroll
            for (int u = 0; u < STEPS; u++) acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[k][u], b[u], acc, 0, 0, 0);
            finish(acc, slot, k);
        }
    };

    // first tile's activations go out before the LDS is prepared (chained: after the producers' flags, and the LDS first)
    v4i b0[STEPS];
    const int8_t* xp0;
    int slot0 = locate(wave, xp0);
    if ((!CHAINED || !dep.wait_counter) && !CHUNKED) load_b(xp0, b0, 0);
    PWDW_STAMP(1);
    if (MODE != 0 && MODE != 4) {
        // depthwise zero padding: everything the pointwise phase does not overwrite
        const uint4 z = {0u, 0u, 0u, 0u};
        for (int i = t; i < (a.RH * a.RW + 4) * SL; i += nthreads) reinterpret_cast<uint4*>(inter)[i] = z;
        if (MODE == 1) {
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int c = 0; c < 4; c++) wsh[r][c] = wrow[r][c] << 8;     // taps of the second output of a lane: {0, w0, w1, w2}
        }
        if (!CHAINED || !dep.wait_counter) __syncthreads();    // (chain_wait ends with a barrier of its own)
    }
    if (CHAINED && dep.wait_counter) {
        chain_wait(dep);
        if (!CHUNKED) load_b(xp0, b0, 0);
    }
    PWDW_STAMP(2);

    if (!CHUNKED) {
        for (int i = wave; i < ntiles; i += 2 * nwaves) {
            // b0 holds tile i; b1 takes tile i + nwaves while b0 is multiplied, then the roles swap
            v4i b1[PINGPONG ? STEPS : 1];
            const int8_t* xp1 = xn;
            int slot1 = -1;
            const bool second = i + nwaves < ntiles;
            if (PINGPONG && second) { slot1 = locate(i + nwaves, xp1); load_b(xp1, reinterpret_cast<v4i (&)[STEPS]>(b1), 0); }
            tile(b0, slot0);
            if (!second) break;
            const bool third = i + 2 * nwaves < ntiles;
            if (PINGPONG) {
                if (third) { slot0 = locate(i + 2 * nwaves, xp0); load_b(xp0, b0, 0); }
                tile(reinterpret_cast<v4i (&)[STEPS]>(b1), slot1);
            } else {
                // one operand buffer (16 steps in registers): tiles one after the other
                slot0 = locate(i + nwaves, xp0); load_b(xp0, b0, 0);
                tile(b0, slot0);
                if (third) { slot0 = locate(i + 2 * nwaves, xp0); load_b(xp0, b0, 0); }
            }
        }
    } else {
        const int nchunks = a.nsteps / STEPS;
        for (int i = wave; i < ntiles; i += nwaves) {
            if (i != wave) slot0 = locate(i, xp0);
            v4i acc = bias_v[0];
            for (int ch = 0; ch < nchunks; ch++) {
#pragma unroll
                for (int u = 0; u < STEPS; u++) af[0][u] = *reinterpret_cast<const v4i*>(wfp + (size_t)(ch * STEPS + u) * 1024);
                load_b(xp0, b0, ch);
#pragma unroll
                for (int u = 0; u < STEPS; u++) acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[0][u], b0[u], acc, 0, 0, 0);
            }
            finish(acc, slot0);
        }
    }
    PWDW_STAMP(3);
    if (MODE == 4) {
        if (CHAINED) chain_signal(dep);
        return;
    }
    __syncthreads();
    PWDW_STAMP(4);

    if (MODE != 0) {
        // ---- depthwise 3x3 from LDS (dwconv.hip's scheme, one 4-pixel fragment per row) --------------------------
        constexpr int TWL = MODE == 1 ? 2 : 1;                 // outputs per lane
        const int strips = (tw + TWL - 1) / TWL;
        const int ntask = th * strips;
        const float inv_strips = __builtin_amdgcn_rcpf((float)strips);
        const Rq drq = a.d_rq;
        int8_t* yn = a.y + ((size_t)(n * a.OH + ty * a.TH) * a.OW + tx * a.TW) * a.ldc + a.c_off + c0;
        constexpr int QSH = SL == 4 ? 4 : SL == 2 ? 3 : 2;    // threads per task position: one per channel quad of the block
        for (int q = t >> QSH; q < ntask; q += nthreads >> QSH) {
            const int oyl = (int)(((float)q + 0.5f) * inv_strips), st = q - oyl * strips;
            const unsigned* row = inter + (SL > 1 ? (cq >> 2) * soff : 0) + ((oyl * S) * RW + st * TWL * S) * 4 + (cq & 3);
            int acc[TWL][4];
#pragma unroll
            for (int j = 0; j < TWL; j++)
#pragma unroll
                for (int c = 0; c < 4; c++) acc[j][c] = c == 0 ? db.x : c == 1 ? db.y : c == 2 ? db.z : db.w;      // the dot chain starts at the bias
#pragma unroll
            for (int r = 0; r < 3; r++) {
                // a strip's 4th column may lie past the region row (odd tile widths): its products carry a zero tap or
                // feed an output that is not stored; the buffer has 4 pixels of slack behind the last row
                const unsigned d[4] = {row[(r * RW + 0) * 4], row[(r * RW + 1) * 4], row[(r * RW + 2) * 4], row[(r * RW + 3) * 4]};
                unsigned frag[4];
                transpose4x4(d, frag);
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    acc[0][c] = __builtin_amdgcn_sdot4((int)frag[c], (int)wrow[r][c], acc[0][c], false);
                    if (TWL == 2) acc[1][c] = __builtin_amdgcn_sdot4((int)frag[c], (int)wsh[r][c], acc[1][c], false);
                }
            }
#pragma unroll
            for (int j = 0; j < TWL; j++) {
                const int oxl = st * TWL + j;
                const unsigned p = requant4<WIN>(acc[j][0], acc[j][1], acc[j][2], acc[j][3], ds, c0, drq);
                if (oxl < tw && c0 < a.c_limit) {
                    unsigned* dst = reinterpret_cast<unsigned*>(yn + ((size_t)oyl * a.OW + oxl) * a.ldc);
                    if (CHAINED) chain_store(dst, p); else *dst = p;
                }
            }
        }
    } else if (wave == 0) {
        // ---- global pooling over the region (pooling_kernel_ref_int8.c:84-189; misc_kernels.hip global_pool_i8): one wave,
        // lane = (pixel phase 0..3, channel 0..15): a short serial sum, two cross-lane steps, one requantisation per channel
        const int ch = lane & 15, part = lane >> 4;
        int s = a.pool_method == 0 ? -128 : 0;
        for (int p = part; p < VP; p += 4) {
            const int e = sx8(inter[p * 4 + (ch >> 2)], ch & 3);
            s = a.pool_method == 0 ? (s > e ? s : e) : s + e;
        }
#pragma unroll
        for (int m = 16; m < 64; m <<= 1) {
            const int o = __shfl_xor(s, m, 64);
            s = a.pool_method == 0 ? (s > o ? s : o) : s + o;
        }
        int q;
        if (a.pool_method == 0) {
            q = round_sat(__fmul_rn((float)s, __fdiv_rn(a.p_in_scale, a.p_out_scale)));
        } else {
            float f = __fmul_rn((float)s, a.p_in_scale);
            f = __fdiv_rn(f, (float)VP);
            q = round_sat(__fdiv_rn(f, a.p_out_scale));
        }
        if (part == 0 && c_base + ch < a.c_limit) {
            int8_t* dst = a.y + (size_t)n * a.ldc + a.c_off + c_base + ch;
            if (CHAINED) __hip_atomic_store(dst, (int8_t)q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *dst = (int8_t)q;
        }
    }
    PWDW_STAMP(5);
    PWDW_DRAIN();
    PWDW_STAMP(7);
    if (CHAINED) chain_signal(dep);
    PWDW_STAMP(6);
}

// both windows of the launch in the one-binade form?  (host: PWDW_LAUNCH picks the kernel instance)
static bool pwdw_windows(const PwDwArgs& a, int mode)
{
#ifdef TAMD_EXP_PWDW_NOWIN          // tools/exp A/B builds only
    return false;
#endif
    return rq_win(a.rq) && (mode == 0 || mode == 4 || rq_win(a.d_rq));
}

#ifndef TAMD_PWDW_STAMPS
static_assert(sizeof(PwDwArgs) == 304, "PwDwArgs must not grow: see the note at PwDwArgs::sl (kernels.h)");
#endif

template <int STEPS, int MODE, bool CHUNKED, int PROD, int WIN, int SL>
__global__ __launch_bounds__(512) void pwdw_i8_kernel(PwDwArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned inter[];       // [slice][region pixel][4 dwords = 16 channels]
    const ChainDep none = {nullptr, 0, nullptr, nullptr};
    pwdw_block<STEPS, MODE, CHUNKED, PROD, false, WIN, SL>(a, inter, blockIdx.x, blockIdx.y, blockIdx.z, blockDim.x, none);
}

// The COHERENT form of the same launch (PwDwArgs::coherent, used under direct dispatch -- direct.cc): every tensor the launch
// receives from or hands to another launch travels with agent-scope accesses -- input rows by sc1 buffer loads (served by the
// memory side, never by a stale L1 / L2 line), results by write-through sc1 stores -- exactly the per-access form of the
// chained experiment below.  Such a launch needs NO cache maintenance at its boundaries: its AQL packet carries fence scope
// "none" (0.84 us per boundary against 1.26 us with agent-scope fences, tools/exp/aql_chain.cpp), and the weights / bias /
// multiplier vectors, which nobody ever writes, stay valid in the L2s from one pass to the next.  The first convolution's
// graph input (PROD 1) is read with ordinary loads: it changes only between bursts, behind a system-scope acquire.
template <int STEPS, int MODE, bool CHUNKED, int PROD, int WIN, int SL>
__global__ __launch_bounds__(512) void pwdw_i8_coh_kernel(PwDwArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned inter[];
    const ChainDep none = {nullptr, 0, nullptr, nullptr};
    pwdw_block<STEPS, MODE, CHUNKED, PROD, true, WIN, SL>(a, inter, blockIdx.x, blockIdx.y, blockIdx.z, blockDim.x, none);
}

#ifdef TAMD_PWDW_CHAIN_EXPERIMENT
// =================================================================================================================
// A CHAIN of these layers in one launch (batch-1 MobileNet: conv1+dw2_1 ... conv6/sep+pool6, fc7).
// A dependent launch costs 1.25-1.45 us of gap + dispatch ramp, and the new kernel's first loads (weights, bias) only start
// behind it.  A flag hand-over between two resident workgroups costs ~0.55 us (tools/exp/flag_handoff.hip,
// profiles/r02_flag_handoff_device_clock.txt).  So the blocks of ALL layers of the chain are one grid, layer after layer in
// block-index order (workgroups are dispatched in index order: a block's producers are resident or finished before it starts,
// so waiting on them cannot deadlock).  A block fetches its weights, prepares its LDS, THEN waits until the previous layer's
// finished-block counter (one release fetch-add per block; counters only grow, the target is epoch * blocks) says the layer
// is complete (one uncached load per block and polling round) and runs the layer body -- with write-through stores and
// agent-scope loads for the tensors that travel inside the chain (see chain_wait), so no cache is flushed or invalidated.  "Previous layer complete"
// implies all earlier layers complete, so any tensor of the chain may be read.  The last finisher of the last layer bumps
// the epoch for the next launch.  Waits are bounded; a give-up raises sync[2] (checked by the host in tests / at prerun).
// =================================================================================================================
template <int STEPS, int MODE, int PROD>
__device__ __forceinline__ void chain_variant(const PwDwArgs& a, unsigned* inter, int bx, int by, int bz, int nthreads, const ChainDep& d)
{
    pwdw_block<STEPS, MODE, false, PROD, true>(a, inter, bx, by, bz, nthreads, d);
}

template <int MODE>
__device__ __forceinline__ void chain_steps(int steps, const PwDwArgs& a, unsigned* inter, int bx, int by, int bz, int nthreads, const ChainDep& d)
{
    switch (steps) {
    case 1: chain_variant<1, MODE, 0>(a, inter, bx, by, bz, nthreads, d); break;
    case 2: chain_variant<2, MODE, 0>(a, inter, bx, by, bz, nthreads, d); break;
    case 4: chain_variant<4, MODE, 0>(a, inter, bx, by, bz, nthreads, d); break;
    case 8: chain_variant<8, MODE, 0>(a, inter, bx, by, bz, nthreads, d); break;
    default: chain_variant<16, MODE, 0>(a, inter, bx, by, bz, nthreads, d); break;
    }
}

__global__ __launch_bounds__(512) void pwdw_chain_kernel(const PwChainArgs c)
{
    extern __shared__ __attribute__((aligned(16))) unsigned inter[];
    const int b = blockIdx.x;
    int l = 0;
    while (l + 1 < c.nlayers && b >= c.first_block[l + 1]) l++;
    const int local = b - c.first_block[l];
    const int gx = c.gx[l], gy = c.gy[l];
    const int bz = local / (gx * gy), rem = local - bz * gx * gy, by = rem / gx, bx = rem - by * gx;
    const int epoch = __hip_atomic_load(c.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ChainDep d;                       // counters: one per layer, a cache line apart
    d.wait_counter = l ? c.flags + 32 * (l - 1) : nullptr;
    d.wait_target = l ? (int)((unsigned)epoch * (unsigned)(c.first_block[l] - c.first_block[l - 1])) : 0;
    d.my_counter = c.flags + 32 * l;
    d.err = c.sync + 2;
    const PwDwArgs a = c.layers[l];
    const int nthreads = blockDim.x;
    const int v = c.variant[l];                 // MODE | PROD << 3 | steps << 4
    const int mode = v & 7, steps = v >> 4;
    if (v & 8) {
        if (mode == 1) chain_variant<1, 1, 1>(a, inter, bx, by, bz, nthreads, d);
        else if (mode == 2) chain_variant<1, 2, 1>(a, inter, bx, by, bz, nthreads, d);
        else chain_variant<1, 3, 1>(a, inter, bx, by, bz, nthreads, d);
    } else {
        switch (mode) {
        case 0: chain_steps<0>(steps, a, inter, bx, by, bz, nthreads, d); break;
        case 1: chain_steps<1>(steps, a, inter, bx, by, bz, nthreads, d); break;
        case 2: chain_steps<2>(steps, a, inter, bx, by, bz, nthreads, d); break;
        case 3: chain_steps<3>(steps, a, inter, bx, by, bz, nthreads, d); break;
        default: chain_steps<4>(steps, a, inter, bx, by, bz, nthreads, d); break;
        }
    }
    // the last block of the last layer to finish opens the next epoch (every block has read the current one by then)
    if (l == c.nlayers - 1 && threadIdx.x == 0) {
        const int nlast = c.first_block[l + 1] - c.first_block[l];
        if (atomicAdd(c.sync + 1, 1) == nlast - 1) {
            __hip_atomic_store(c.sync + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(c.sync, epoch + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // next reader: the next launch
        }
    }
}
#endif   // TAMD_PWDW_CHAIN_EXPERIMENT

// This file is compiled twice: as itself (every launch with ONE slice per block -- all batch-1 launches -- plus the host-side helpers) and,
// through pwdw_slices.hip (TAMD_PWDW_SLICES_TU), as a second code object that holds the two- and four-slice instances.  In one code
// object the batch-1 kernels sit scattered among six times as many instances and the batch-1 step loses 0.4-0.9 %
// (profiles/r05_ab_slices4_layers_mobilenet_v1_b1.txt: the same instructions, 2-6 % slower per isolated launch).
#ifndef TAMD_PWDW_SLICES_TU
size_t pwdw_lds_bytes(const PwDwArgs& a, int threads)
{
    if (a.mode == 2) return 0;
    if (a.mode == 0) return ((size_t)a.H * a.W + 4) * 16;
    return ((size_t)a.RH * a.RW + 8) * 16 * (a.sl > 1 ? a.sl : 1);
}

bool pwdw_config_ok(const PwDwArgs& a, int threads)
{
    if (threads != 256 && threads != 512) return false;
    if (pwdw_lds_bytes(a, threads) > 64 * 1024) return false;
    if (a.prod == 1 && (a.nsteps != 1 || a.mode != 1)) return false;
    if (a.sl != 0 && a.sl != 1 && a.sl != 2 && a.sl != 4) return false;
    if (a.sl > 1 && (a.mode != 1 || a.nsteps > a.steps)) return false;       // several slices per block: depthwise tails, register-resident K
    if (a.sl == 4 && a.steps > 2) return false;                              // (four: K <= 128, the fragments of all slices stay in registers)
    if (a.mode == 0) return a.H * a.W <= 1024;
    if (a.mode == 2) return a.TH >= 1 && (long)a.TH * a.W < 16384;
    return a.TH >= 1 && a.TW >= 1 && a.RH * a.RW < 16384;
}

// K steps kept in registers for a layer of `nsteps` real 64-deep steps: the planner pads the weight panel to a multiple
int pwdw_steps(int nsteps)
{
    if (nsteps <= 1) return 1;
    if (nsteps <= 2) return 2;
    if (nsteps <= 4) return 4;
    if (nsteps <= 8) return 8;
    if (nsteps <= 16) return 16;
    int best = 16, pad = (nsteps + 15) / 16 * 16;
    for (int s : {8, 4}) { const int p = (nsteps + s - 1) / s * s; if (p < pad) { pad = p; best = s; } }
    return best;
}
#endif

// plain or coherent instance of one variant, with the general or the one-binade requantisation (a kernel each: a branch at the top of
// one kernel costs the batch-1 launches 0.1-0.3 us -- more scalar state, a longer preamble -- profiles/r05_ab_window_layers_mobilenet_v1_b1.txt)
#define PWDW_LAUNCH_WS(STEPS_, MODE_, CHUNKED_, PROD_, WIN_, SL_)                                                             \
    do {                                                                                                                     \
        if (a.coherent) { launch_rec_coherent(); hipLaunchKernelGGL((pwdw_i8_coh_kernel<STEPS_, MODE_, CHUNKED_, PROD_, WIN_, SL_>), grid, dim3(threads), lds, s, a); } \
        else hipLaunchKernelGGL((pwdw_i8_kernel<STEPS_, MODE_, CHUNKED_, PROD_, WIN_, SL_>), grid, dim3(threads), lds, s, a); \
    } while (0)
#define PWDW_LAUNCH_S(STEPS_, MODE_, CHUNKED_, PROD_, SL_)                                                                   \
    do {                                                                                                                     \
        if (pwdw_windows(a, MODE_)) PWDW_LAUNCH_WS(STEPS_, MODE_, CHUNKED_, PROD_, 1, SL_);                                  \
        else PWDW_LAUNCH_WS(STEPS_, MODE_, CHUNKED_, PROD_, 0, SL_);                                                         \
    } while (0)
#ifndef TAMD_PWDW_SLICES_TU
#define PWDW_LAUNCH(STEPS_, MODE_, CHUNKED_, PROD_) PWDW_LAUNCH_S(STEPS_, MODE_, CHUNKED_, PROD_, 1)
#define PWDW_LAUNCH_DW(STEPS_, MODE_, CHUNKED_, PROD_) PWDW_LAUNCH_S(STEPS_, MODE_, CHUNKED_, PROD_, 1)
#define LAUNCH_PWDW_NAME launch_pwdw
#else
// depthwise tails with two / four 16-channel slices per block (PwDwArgs::sl; register-resident K, four: K <= 128)
#define PWDW_LAUNCH(STEPS_, MODE_, CHUNKED_, PROD_) return hipErrorInvalidValue
#define PWDW_LAUNCH_DW(STEPS_, MODE_, CHUNKED_, PROD_)                                                                       \
    do {                                                                                                                     \
        if constexpr (!(CHUNKED_)) { if (a.sl == 2) { PWDW_LAUNCH_S(STEPS_, MODE_, false, PROD_, 2); break; } }              \
        if constexpr (!(CHUNKED_) && (STEPS_) <= 2) { if (a.sl == 4) { PWDW_LAUNCH_S(STEPS_, MODE_, false, PROD_, 4); break; } } \
        return hipErrorInvalidValue;                                                                                         \
    } while (0)
#define LAUNCH_PWDW_NAME launch_pwdw_slices
#endif

template <int STEPS, bool CHUNKED>
static hipError_t launch_steps(const PwDwArgs& a, int threads, hipStream_t s)
{
    const size_t lds = pwdw_lds_bytes(a, threads);
    if (a.mode == 2) {
        const dim3 sm(a.slices, 1, ((a.H + a.TH - 1) / a.TH) * a.N);
        const dim3 grid = a.tile_major ? dim3(sm.y, sm.z, sm.x) : sm;
        PWDW_LAUNCH(STEPS, 4, CHUNKED, 0);
        return hipGetLastError();
    }
    const dim3 sm(a.slices, a.mode == 0 ? 1 : a.tiles_x, a.mode == 0 ? a.N : a.tiles_y * a.N);
    const dim3 grid = a.tile_major ? dim3(sm.y, sm.z, sm.x) : sm;
    if (a.mode == 0) PWDW_LAUNCH(STEPS, 0, CHUNKED, 0);
    else if (a.S == 2) PWDW_LAUNCH_DW(STEPS, 2, CHUNKED, 0);
    else if (a.TH * ((a.TW + 1) / 2) * 4 >= threads) PWDW_LAUNCH_DW(STEPS, 1, CHUNKED, 0);
    else PWDW_LAUNCH_DW(STEPS, 3, CHUNKED, 0);      // small tile: one output per lane
    return hipGetLastError();
}

static hipError_t launch_first(const PwDwArgs& a, int threads, hipStream_t s)
{
    const dim3 sm(a.slices, a.tiles_x, a.tiles_y * a.N);
    const dim3 grid = a.tile_major ? dim3(sm.y, sm.z, sm.x) : sm;
    const size_t lds = pwdw_lds_bytes(a, threads);
    if (a.S == 2) PWDW_LAUNCH_DW(1, 2, false, 1);
    else if (a.TH * ((a.TW + 1) / 2) * 4 >= threads) PWDW_LAUNCH_DW(1, 1, false, 1);
    else PWDW_LAUNCH_DW(1, 3, false, 1);
    return hipGetLastError();
}

#ifdef TAMD_PWDW_CHAIN_EXPERIMENT
// the template instance launch_pwdw() would pick for this layer, as a code for pwdw_chain_kernel, and its grid
int pwdw_chain_variant(const PwDwArgs& a, int threads, int* gx, int* gy, int* gz)
{
    if (a.nsteps > a.steps) return -1;                      // chunked K loops stay stand-alone
    int mode;
    dim3 sm;
    if (a.prod == 1 || (a.mode != 2 && a.mode != 0)) {
        sm = dim3(a.slices, a.tiles_x, a.tiles_y * a.N);
        mode = a.S == 2 ? 2 : (a.TH * ((a.TW + 1) / 2) * 4 >= threads ? 1 : 3);
    } else if (a.mode == 2) {
        sm = dim3(a.slices, 1, ((a.H + a.TH - 1) / a.TH) * a.N);
        mode = 4;
    } else {
        sm = dim3(a.slices, 1, a.N);
        mode = 0;
    }
    const dim3 grid = a.tile_major ? dim3(sm.y, sm.z, sm.x) : sm;
    *gx = (int)grid.x; *gy = (int)grid.y; *gz = (int)grid.z;
    if (grid.x > 32767 || grid.y > 32767) return -1;
    return mode | (a.prod == 1 ? 8 : 0) | (a.prod == 1 ? 1 : a.steps) << 4;
}

hipError_t launch_pwdw_chain(const PwChainArgs& c, int threads, size_t lds, hipStream_t s)
{
    hipLaunchKernelGGL(pwdw_chain_kernel, dim3(c.first_block[c.nlayers]), dim3(threads), lds, s, c);
    return hipGetLastError();
}
#endif

#ifndef TAMD_PWDW_SLICES_TU
hipError_t launch_pwdw_slices(const PwDwArgs& a, int threads, hipStream_t s);      // pwdw_slices.hip
#endif

hipError_t LAUNCH_PWDW_NAME(const PwDwArgs& a, int threads, hipStream_t s)
{
#ifndef TAMD_PWDW_SLICES_TU
    if (a.sl > 1) return launch_pwdw_slices(a, threads, s);
#endif
    if (a.prod == 1) return launch_first(a, threads, s);
    const bool chunked = a.nsteps > a.steps;      // a.nsteps is a multiple of a.steps == pwdw_steps(real steps)
    switch (a.steps) {
    case 1: return launch_steps<1, false>(a, threads, s);
    case 2: return launch_steps<2, false>(a, threads, s);
    case 4: return chunked ? launch_steps<4, true>(a, threads, s) : launch_steps<4, false>(a, threads, s);
    case 8: return chunked ? launch_steps<8, true>(a, threads, s) : launch_steps<8, false>(a, threads, s);
    default: return chunked ? launch_steps<16, true>(a, threads, s) : launch_steps<16, false>(a, threads, s);
    }
}

}  // namespace tamd
