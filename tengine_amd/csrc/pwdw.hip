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
#include "dw_common.h"
#include "epilogue.h"
#include "kernels.h"

namespace tamd {

typedef int v4i __attribute__((ext_vector_type(4)));

// MODE 0: global pooling tail; 1: depthwise stride 1; 2: depthwise stride 2.  STEPS: 64-deep K steps held in registers.
template <int STEPS, int MODE>
__global__ __launch_bounds__(512) void pwdw_i8_kernel(PwDwArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned inter[];       // [region pixel][4 dwords = 16 channels]
    constexpr int S = MODE == 2 ? 2 : 1;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, nwaves = blockDim.x >> 6;
    const int l15 = lane & 15, kb = lane >> 4;
    const int slice = blockIdx.x, tx = blockIdx.y;
    int ty = blockIdx.z, n = 0;
    if (a.N > 1) { n = ty / a.tiles_y; ty -= n * a.tiles_y; }
    const int c_base = slice * 16;

    // ---- loads that depend on nothing but the block index go out first ----------------------------------------
    const int4 pb = *reinterpret_cast<const int4*>(a.bias + c_base + 4 * kb);
    const float4 ps = *reinterpret_cast<const float4*>(a.wscale + c_base + 4 * kb);
    const int8_t* wfp = a.wf + ((size_t)slice * a.nsteps * 64 + lane) * 16;
    const int nchunks = (a.nsteps + STEPS - 1) / STEPS;
    const v4i zero4 = {0, 0, 0, 0};
    v4i af[STEPS];
    if (nchunks == 1) {
#pragma unroll
        for (int u = 0; u < STEPS; u++) af[u] = u < a.nsteps ? *reinterpret_cast<const v4i*>(wfp + u * 1024) : zero4;
    }
    const int cq = t & 3;                                  // tail phase: this thread's channel quad of the slice
    const int c0 = c_base + cq * 4;
    unsigned wrow[3][4];
    int4 db = {0, 0, 0, 0};
    float4 ds = {1.f, 1.f, 1.f, 1.f};
    if (MODE != 0) {
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
    if (MODE != 0) {
        const int oy0 = ty * a.TH, ox0 = tx * a.TW;
        iy0 = oy0 * S - a.PH; ix0 = ox0 * S - a.PW;
        th = min(a.TH, a.OH - oy0); tw = min(a.TW, a.OW - ox0);
        vy0 = max(iy0, 0); vx0 = max(ix0, 0);
        vy1 = min(iy0 + (th - 1) * S + 3, a.H); vx1 = min(ix0 + (tw - 1) * S + 3, a.W);
        RW = a.RW;
        // depthwise zero padding: everything the pointwise phase does not overwrite
        const uint4 z = {0u, 0u, 0u, 0u};
        for (int i = t; i < a.RH * a.RW + 4; i += blockDim.x) reinterpret_cast<uint4*>(inter)[i] = z;
        __syncthreads();
    }
    const int VW = vx1 - vx0, VP = (vy1 - vy0) * VW;
    const float inv_vw = __builtin_amdgcn_rcpf((float)VW);
    const int ntiles = (VP + 15) >> 4;
    const int klim = a.ktot - kb * 16;                     // this lane's 16 K bytes of step u are real iff u*64 < klim
    const Rq rq = make_rq(a.m1, a.lo, a.hi, a.out_scale);
    const int8_t* xn = a.x + (size_t)n * a.H * a.W * a.cs_in + kb * 16;

    // region pixel of lane l15 in tile i: LDS slot and input address
    auto locate = [&](int i, int& slot, const int8_t*& xp) -> bool {
        const int v = i * 16 + l15;
        // v / VW without an integer division: (v + 0.5) / VW is at least 0.5 / VW away from an integer, the float error
        // (v < 2^14, 1-ulp rcp) is orders of magnitude smaller
        const int vy = (int)(((float)v + 0.5f) * inv_vw), vx = v - vy * VW;
        const int iy = vy0 + vy, ix = vx0 + vx;
        slot = (iy - iy0) * RW + (ix - ix0);
        xp = xn + (size_t)(unsigned)((iy * a.W + ix) * a.cs_in);
        return v < VP;
    };

    if (nchunks == 1) {
        v4i bf[STEPS];
        int slot; const int8_t* xp;
        bool valid = false;
        if (wave < ntiles) {
            valid = locate(wave, slot, xp);
#pragma unroll
            for (int u = 0; u < STEPS; u++) bf[u] = (valid && u * 64 < klim) ? *reinterpret_cast<const v4i*>(xp + u * 64) : zero4;
        }
        for (int i = wave; i < ntiles; i += nwaves) {
            v4i bn[STEPS];
            int slot_n = 0; const int8_t* xp_n = xn;
            bool valid_n = false;
            if (i + nwaves < ntiles) {
                valid_n = locate(i + nwaves, slot_n, xp_n);
#pragma unroll
                for (int u = 0; u < STEPS; u++) bn[u] = (valid_n && u * 64 < klim) ? *reinterpret_cast<const v4i*>(xp_n + u * 64) : zero4;
            }
            v4i acc = zero4;
#pragma unroll
            for (int u = 0; u < STEPS; u++)
                if (u < a.nsteps) acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[u], bf[u], acc, 0, 0, 0);
            const unsigned p = requant4(acc[0] + pb.x, acc[1] + pb.y, acc[2] + pb.z, acc[3] + pb.w, ps, rq);
            if (valid) inter[slot * 4 + kb] = p;
            if (i + nwaves < ntiles) {
#pragma unroll
                for (int u = 0; u < STEPS; u++) bf[u] = bn[u];
                slot = slot_n; valid = valid_n;
            }
        }
    } else {
        // deep K (> 8 steps): chunks of STEPS, weights re-read per tile (only the 7x7 layers get here: <= 4 tiles a block)
        for (int i = wave; i < ntiles; i += nwaves) {
            int slot; const int8_t* xp;
            const bool valid = locate(i, slot, xp);
            v4i acc = zero4;
            for (int ch = 0; ch < nchunks; ch++) {
                v4i bf[STEPS];
#pragma unroll
                for (int u = 0; u < STEPS; u++) {
                    const int s = ch * STEPS + u;
                    af[u] = s < a.nsteps ? *reinterpret_cast<const v4i*>(wfp + (size_t)s * 1024) : zero4;
                    bf[u] = (valid && s * 64 < klim) ? *reinterpret_cast<const v4i*>(xp + s * 64) : zero4;
                }
#pragma unroll
                for (int u = 0; u < STEPS; u++) acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[u], bf[u], acc, 0, 0, 0);
            }
            const unsigned p = requant4(acc[0] + pb.x, acc[1] + pb.y, acc[2] + pb.z, acc[3] + pb.w, ps, rq);
            if (valid) inter[slot * 4 + kb] = p;
        }
    }
    __syncthreads();

    if (MODE != 0) {
        // ---- depthwise 3x3 from LDS (dwconv.hip's scheme, one 4-pixel fragment per row) --------------------------
        constexpr int TWL = S == 1 ? 2 : 1;                    // outputs per lane
        const int strips = (tw + TWL - 1) / TWL;
        const int ntask = th * strips;
        const float inv_strips = __builtin_amdgcn_rcpf((float)strips);
        const Rq drq = make_rq(a.d_m1, a.d_lo, a.d_hi, a.d_out_scale);
        int8_t* yn = a.y + ((size_t)(n * a.OH + ty * a.TH) * a.OW + tx * a.TW) * a.ldc + a.c_off + c0;
        for (int q = t >> 2; q < ntask; q += blockDim.x >> 2) {
            const int oyl = (int)(((float)q + 0.5f) * inv_strips), st = q - oyl * strips;
            const unsigned* row = inter + ((oyl * S) * RW + st * TWL * S) * 4 + cq;
            int acc[TWL][4];
#pragma unroll
            for (int j = 0; j < TWL; j++)
#pragma unroll
                for (int c = 0; c < 4; c++) acc[j][c] = 0;
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
                    if (TWL == 2) acc[1][c] = __builtin_amdgcn_sdot4((int)frag[c], (int)(wrow[r][c] << 8), acc[1][c], false);
                }
            }
#pragma unroll
            for (int j = 0; j < TWL; j++) {
                const int oxl = st * TWL + j;
                const unsigned p = requant4(acc[j][0] + db.x, acc[j][1] + db.y, acc[j][2] + db.z, acc[j][3] + db.w, ds, drq);
                if (oxl < tw && c0 < a.c_limit) *reinterpret_cast<unsigned*>(yn + ((size_t)oyl * a.OW + oxl) * a.ldc) = p;
            }
        }
    } else {
        // ---- global pooling over the region (pooling_kernel_ref_int8.c:84-189; misc_kernels.hip global_pool_i8) ------
        unsigned* red = inter + (size_t)(VP + 4) * 4;          // [wave][4 quads][4]
        int s4[4];
#pragma unroll
        for (int b = 0; b < 4; b++) s4[b] = a.pool_method == 0 ? -128 : 0;
        for (int p = t >> 2; p < VP; p += blockDim.x >> 2) {
            const unsigned v = inter[p * 4 + cq];
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int e = sx8(v, b);
                s4[b] = a.pool_method == 0 ? (s4[b] > e ? s4[b] : e) : s4[b] + e;
            }
        }
#pragma unroll
        for (int m = 4; m < 64; m <<= 1)
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int o = __shfl_xor(s4[b], m, 64);
                s4[b] = a.pool_method == 0 ? (s4[b] > o ? s4[b] : o) : s4[b] + o;
            }
        if (lane < 4) {
#pragma unroll
            for (int b = 0; b < 4; b++) red[(wave * 4 + lane) * 4 + b] = (unsigned)s4[b];
        }
        __syncthreads();
        if (t < 4) {
            int q[4];
#pragma unroll
            for (int b = 0; b < 4; b++) {
                int r = (int)red[t * 4 + b];
                for (int w = 1; w < nwaves; w++) {
                    const int e = (int)red[(w * 4 + t) * 4 + b];
                    r = a.pool_method == 0 ? (r > e ? r : e) : r + e;
                }
                if (a.pool_method == 0) {
                    q[b] = round_sat(__fmul_rn((float)r, __fdiv_rn(a.p_in_scale, a.p_out_scale)));
                } else {
                    float f = __fmul_rn((float)r, a.p_in_scale);
                    f = __fdiv_rn(f, (float)VP);
                    q[b] = round_sat(__fdiv_rn(f, a.p_out_scale));
                }
            }
            if (c0 < a.c_limit) *reinterpret_cast<unsigned*>(a.y + (size_t)n * a.ldc + a.c_off + c0) = pack4(q[0], q[1], q[2], q[3]);
        }
    }
}

size_t pwdw_lds_bytes(const PwDwArgs& a, int threads)
{
    if (a.mode == 0) return ((size_t)a.H * a.W + 4) * 16 + (size_t)(threads / 64) * 64;
    return ((size_t)a.RH * a.RW + 8) * 16;
}

bool pwdw_config_ok(const PwDwArgs& a, int threads)
{
    if (threads != 256 && threads != 512) return false;
    if (pwdw_lds_bytes(a, threads) > 64 * 1024) return false;
    if (a.mode == 0) return a.H * a.W <= 1024;
    return a.TH >= 1 && a.TW >= 1 && a.RH * a.RW < 16384;
}

template <int STEPS>
static hipError_t launch_steps(const PwDwArgs& a, int threads, hipStream_t s)
{
    const dim3 grid(a.slices, a.mode == 0 ? 1 : a.tiles_x, a.mode == 0 ? a.N : a.tiles_y * a.N);
    const size_t lds = pwdw_lds_bytes(a, threads);
    if (a.mode == 0) hipLaunchKernelGGL((pwdw_i8_kernel<STEPS, 0>), grid, dim3(threads), lds, s, a);
    else if (a.S == 1) hipLaunchKernelGGL((pwdw_i8_kernel<STEPS, 1>), grid, dim3(threads), lds, s, a);
    else hipLaunchKernelGGL((pwdw_i8_kernel<STEPS, 2>), grid, dim3(threads), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_pwdw(const PwDwArgs& a, int threads, hipStream_t s)
{
    if (a.nsteps <= 1) return launch_steps<1>(a, threads, s);
    if (a.nsteps <= 2) return launch_steps<2>(a, threads, s);
    if (a.nsteps <= 4) return launch_steps<4>(a, threads, s);
    return launch_steps<8>(a, threads, s);
}

}  // namespace tamd
