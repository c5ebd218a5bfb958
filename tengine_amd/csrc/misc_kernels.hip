// Bandwidth-bound glue ops of the config graphs, int8 NHWC on device.
// Each follows the reference's dequant -> fp32 -> requant formula op for op (bit-exact), vectorised
// 4..16 bytes per lane with lanes along the contiguous channel dimension.
#include "epilogue.h"
#include "kernels.h"

namespace tamd {

__device__ __forceinline__ int sxb(unsigned v, int b) { return (int)(signed char)((v >> (8 * b)) & 0xff); }

// ---- pooling: pooling/pooling_kernel_ref_int8.c:84-189 ---------------------------------------------
// max: y = round((float)max_q * (in_scale/out_scale)); avg: f=(float)sum*in_scale; f=f/(float)pool_size;
// y = round(f/out_scale); pool_size = in-image taps unless caffe_flavor (window clipped to in+pad).
__global__ __launch_bounds__(256) void pool_i8_kernel(PoolArgs a)
{
    const int cg = (a.C + 3) / 4;          // channel groups of THIS tensor (it may be a view in a wider buffer)
    const long total = (long)a.N * a.OH * a.OW * cg;
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c4 = (int)(idx % cg); idx /= cg;
    const int px = (int)(idx % a.OW); idx /= a.OW;
    const int py = (int)(idx % a.OH);
    const int n = (int)(idx / a.OH);

    int hs = py * a.SH - a.PH, he = hs + a.KH;
    if (he > a.H + a.PH) he = a.H + a.PH;
    int ws = px * a.SW - a.PW, we = ws + a.KW;
    if (we > a.W + a.PW) we = a.W + a.PW;
    int pool_size = 1;
    if (a.caffe_flavor) pool_size = (he - hs) * (we - ws);
    hs = hs > 0 ? hs : 0;
    ws = ws > 0 ? ws : 0;
    he = he < a.H ? he : a.H;
    we = we < a.W ? we : a.W;
    if (!a.caffe_flavor) pool_size = (he - hs) * (we - ws);

    const int8_t* xn = a.x + (size_t)n * a.H * a.W * a.cs_in + c4 * 4;
    int q[4];
    if (a.method == 0) {
        const unsigned f0 = *reinterpret_cast<const unsigned*>(xn + ((size_t)hs * a.W + ws) * a.cs_in);
        int m[4] = {sxb(f0, 0), sxb(f0, 1), sxb(f0, 2), sxb(f0, 3)};
        for (int iy = hs; iy < he; iy++)
            for (int ix = ws; ix < we; ix++) {
                const unsigned v = *reinterpret_cast<const unsigned*>(xn + ((size_t)iy * a.W + ix) * a.cs_in);
#pragma unroll
                for (int b = 0; b < 4; b++) { int t = sxb(v, b); m[b] = m[b] > t ? m[b] : t; }
            }
        const float rq = __fdiv_rn(a.in_scale, a.out_scale);
#pragma unroll
        for (int b = 0; b < 4; b++) q[b] = round_sat(__fmul_rn((float)m[b], rq));
    } else {
        int s[4] = {0, 0, 0, 0};
        for (int iy = hs; iy < he; iy++)
            for (int ix = ws; ix < we; ix++) {
                const unsigned v = *reinterpret_cast<const unsigned*>(xn + ((size_t)iy * a.W + ix) * a.cs_in);
#pragma unroll
                for (int b = 0; b < 4; b++) s[b] += sxb(v, b);
            }
#pragma unroll
        for (int b = 0; b < 4; b++) {
            float f = __fmul_rn((float)s[b], a.in_scale);
            f = __fdiv_rn(f, (float)pool_size);
            q[b] = round_sat(__fdiv_rn(f, a.out_scale));
        }
    }
    *reinterpret_cast<unsigned*>(a.y + (((size_t)n * a.OH + py) * a.OW + px) * a.ldc + a.c_off + c4 * 4) =
        pack4(q[0], q[1], q[2], q[3]);
}

// global pooling (window == whole map, no padding): 16 pixel lanes x 16 channel quads per block, LDS
// tree-free reduce; same formulas as above with pool_size = H*W.
__global__ __launch_bounds__(256) void global_pool_i8_kernel(PoolArgs a)
{
    __shared__ int red[16][16][4];
    const int t = threadIdx.x;
    const int cq = t & 15, pl = t >> 4;
    const int c4 = blockIdx.x * 16 + cq;
    const int n = blockIdx.y;
    const int cg = (a.C + 3) / 4;
    const int hw = a.H * a.W;
    const bool cvalid = c4 < cg;
    int s[4] = {0, 0, 0, 0};
    if (a.method == 0) s[0] = s[1] = s[2] = s[3] = -128;
    if (cvalid) {
        const int8_t* xn = a.x + (size_t)n * hw * a.cs_in + c4 * 4;
        for (int p = pl; p < hw; p += 16) {
            const unsigned v = *reinterpret_cast<const unsigned*>(xn + (size_t)p * a.cs_in);
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int e = sxb(v, b);
                s[b] = a.method == 0 ? (s[b] > e ? s[b] : e) : s[b] + e;
            }
        }
    }
#pragma unroll
    for (int b = 0; b < 4; b++) red[pl][cq][b] = s[b];
    __syncthreads();
    if (pl != 0 || !cvalid) return;
    int q[4];
#pragma unroll
    for (int b = 0; b < 4; b++) {
        int r = red[0][cq][b];
        for (int k = 1; k < 16; k++) {
            const int e = red[k][cq][b];
            r = a.method == 0 ? (r > e ? r : e) : r + e;
        }
        if (a.method == 0) {
            q[b] = round_sat(__fmul_rn((float)r, __fdiv_rn(a.in_scale, a.out_scale)));
        } else {
            float f = __fmul_rn((float)r, a.in_scale);
            f = __fdiv_rn(f, (float)hw);
            q[b] = round_sat(__fdiv_rn(f, a.out_scale));
        }
    }
    *reinterpret_cast<unsigned*>(a.y + (size_t)n * a.ldc + a.c_off + c4 * 4) = pack4(q[0], q[1], q[2], q[3]);
}

hipError_t launch_pool(const PoolArgs& a, hipStream_t s)
{
    if (a.OH == 1 && a.OW == 1 && a.KH == a.H && a.KW == a.W && a.PH == 0 && a.PW == 0) {
        const int cg = (a.C + 3) / 4;
        hipLaunchKernelGGL(global_pool_i8_kernel, dim3((cg + 15) / 16, a.N), dim3(256), 0, s, a);
        return hipGetLastError();
    }
    const long total = (long)a.N * a.OH * a.OW * ((a.C + 3) / 4);
    hipLaunchKernelGGL(pool_i8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ---- eltwise (+ optionally the standalone ReLU that follows it in ResNet): eltwise_ref.c:589-640,833-837
// a=(float)qa*sa ; b=(float)qb*sb ; f = a op b ; y = round(f/out_scale) clamp +-127
// fused relu (relu_kernel_ref_int8.c:40-94 applied to y): f2=(float)y*out_scale ; f2<0 -> 0 ;
// y2 = round(f2/relu_out_scale)
__global__ __launch_bounds__(256) void eltwise_i8_kernel(EltArgs a)
{
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
    if (i >= a.count) return;
    const uint4 va = *reinterpret_cast<const uint4*>(a.a + i);
    const uint4 vb = *reinterpret_cast<const uint4*>(a.b + i);
    const unsigned pa[4] = {va.x, va.y, va.z, va.w}, pb[4] = {vb.x, vb.y, vb.z, vb.w};
    const float inv_out = __fdiv_rn(1.0f, a.out_scale);
    const float inv_relu = a.fuse_relu ? __fdiv_rn(1.0f, a.relu_out_scale) : 1.0f;
    unsigned out[4];
#pragma unroll
    for (int d = 0; d < 4; d++) {
        int q[4];
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const float fa = __fmul_rn((float)sxb(pa[d], b), a.sa), fb = __fmul_rn((float)sxb(pb[d], b), a.sb);
            float f;
            switch (a.type) {
            case 0: f = __fmul_rn(fa, fb); break;
            case 2: f = __fadd_rn(fa, fb); break;
            case 4: f = __fsub_rn(fa, fb); break;
            default: f = fa > fb ? fa : fb; break;
            }
            int y = round_div_sat(f, a.out_scale, inv_out);
            if (a.fuse_relu == 2) {
                y = y < 0 ? 0 : y;      // relu_out_scale == out_scale: round(fl(fl(y*s)/s)) == y (epilogue.h fuse_elt4)
            } else if (a.fuse_relu) {
                float f2 = __fmul_rn((float)y, a.out_scale);
                f2 = f2 < 0.f ? 0.f : f2;
                y = round_div_sat(f2, a.relu_out_scale, inv_relu);
            }
            q[b] = y;
        }
        out[d] = pack4(q[0], q[1], q[2], q[3]);
    }
    *reinterpret_cast<uint4*>(a.y + i) = make_uint4(out[0], out[1], out[2], out[3]);
}

hipError_t launch_eltwise(const EltArgs& a, hipStream_t s)
{
    const size_t n16 = (a.count + 15) / 16;
    hipLaunchKernelGGL(eltwise_i8_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ---- host <-> device staging as ordinary launches: the pinned host buffers are device-mapped, so a run's input upload and
// output download are two more nodes of its hipGraph instead of copy-engine commands the compute queue has to hand over to
// and back from (each hand-over is a signal round trip of ~10 us; a batch-1 run has two of them)
__global__ __launch_bounds__(256) void copy_bytes_kernel(uint4* __restrict__ dst, const uint4* __restrict__ src, size_t n16, size_t bytes)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n16) dst[i] = src[i];
    if (i == 0)
        for (size_t b = n16 * 16; b < bytes; b++) reinterpret_cast<uint8_t*>(dst)[b] = reinterpret_cast<const uint8_t*>(src)[b];
}

hipError_t launch_copy_bytes(void* dst, const void* src, size_t bytes, hipStream_t s)
{
    const size_t n16 = bytes / 16;
    hipLaunchKernelGGL(copy_bytes_kernel, dim3((unsigned)((n16 + 255) / 256 + (n16 == 0))), dim3(256), 0, s, (uint4*)dst, (const uint4*)src, n16, bytes);
    return hipGetLastError();
}

// ---- concat input that is not written in place: concat/concat_kernel_ref_int8.c:60-95 (the same text at every rank/axis)
// rescale = in_scale / out_scale ; q = roundf((float)x * rescale) ; q > 127 -> 127 ; q < -127 -> **+127** (the reference's
// lower clamp assigns the wrong sign at all ten sites, :83-84 ...; identical results carry the defect along, as the
// oracle's orc_requant_copy_int8 does).  One thread per output byte: these copies are launch-latency sized.
__global__ __launch_bounds__(256) void concat_copy_i8_kernel(CatCopyArgs a)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.pixels * a.C) return;
    const long pix = idx / a.C;
    const int c = (int)(idx - pix * a.C);
    const int x = a.x[pix * a.cs_in + c];
    int q = x;
    if (!a.identity) {
        q = (int)roundf(__fmul_rn((float)x, a.rescale));
        if (q > 127) q = 127;
        else if (q < -127) q = 127;
    }
    a.y[pix * a.ldc + a.c_off + c] = (int8_t)q;
}

hipError_t launch_concat_copy_i8(const CatCopyArgs& a, hipStream_t s)
{
    const long total = a.pixels * a.C;
    hipLaunchKernelGGL(concat_copy_i8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ---- dense int8 tensors (round 6): Permute(0,2,3,1) (permute_ref.c:305-343, a byte permutation), Flatten (flatten_ref.c:74-80) and
// Reshape (reshape_ref.c, NCHW: byte copies) are views of what this kernel writes; Concat re-scales like concat_copy_i8 above
// (concat_kernel_ref_int8.c: the same text at every rank / axis, the +127 lower clamp included).  One thread per output byte.
__global__ __launch_bounds__(256) void flatcat_i8_kernel(FlatCatI8Args a)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.outer * a.row_len) return;
    const long o = idx / a.row_len;
    const int r = a.row_begin + (int)(idx - o * a.row_len);
    int si = 0;
#pragma unroll
    for (int k = 1; k < kFlatCatMax; k++)
        if (k < a.nsrc && r >= a.src[k].begin) si = k;
    const FlatCatI8Src& s = a.src[si];
    const int e = r - s.begin;
    size_t at;
    if (s.kind == 0) at = (size_t)o * s.chunk + e;
    else if (s.kind == 1) at = ((size_t)o * s.HW + e / s.C) * s.cs + e % s.C;
    else {
        const size_t L = (size_t)o * s.chunk + e, img = (size_t)s.C * s.HW;
        const size_t n = L / img, rem = L - n * img;
        at = (n * s.HW + rem % s.HW) * s.cs + rem / s.HW;
    }
    const int x = s.x[at];
    int q = x;
    if (!s.identity) {
        q = (int)roundf(__fmul_rn((float)x, s.rescale));
        if (q > 127) q = 127;
        else if (q < -127) q = 127;
    }
    a.y[(size_t)o * a.out_row + r] = (int8_t)q;
}

hipError_t launch_flatcat_i8(const FlatCatI8Args& a, hipStream_t s)
{
    if (a.nsrc < 1 || a.nsrc > kFlatCatMax) return hipErrorInvalidValue;
    const long total = a.outer * a.row_len;
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(flatcat_i8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ---- relu / leaky relu: relu/relu_kernel_ref_int8.c:40-94 -------------------------------------------
__global__ __launch_bounds__(256) void relu_i8_kernel(ReluArgs a)
{
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
    if (i >= a.count) return;
    const uint4 v = *reinterpret_cast<const uint4*>(a.x + i);
    const unsigned pv[4] = {v.x, v.y, v.z, v.w};
    const float inv_out = __fdiv_rn(1.0f, a.out_scale);
    unsigned out[4];
#pragma unroll
    for (int d = 0; d < 4; d++) {
        int q[4];
#pragma unroll
        for (int b = 0; b < 4; b++) {
            float f = __fmul_rn((float)sxb(pv[d], b), a.in_scale);
            if (f < 0.f) f = (a.slope == 0.f) ? 0.f : __fmul_rn(f, a.slope);
            q[b] = round_div_sat(f, a.out_scale, inv_out);
        }
        out[d] = pack4(q[0], q[1], q[2], q[3]);
    }
    *reinterpret_cast<uint4*>(a.y + i) = make_uint4(out[0], out[1], out[2], out[3]);
}

hipError_t launch_relu(const ReluArgs& a, hipStream_t s)
{
    const size_t n16 = (a.count + 15) / 16;
    hipLaunchKernelGGL(relu_i8_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ---- softmax over the channel axis: softmax/softmax_kernel_ref_int8.c:41-117 over softmax_kernel_ref.h:35-85 ------------------
// f = (float)q * in_scale; per position: max over the axis; o = (float)exp((double)(f - max)) -- the reference calls C `exp` on a
// float argument, i.e. the DOUBLE routine, and rounds to float on the store (:67); the sum is accumulated in fp32 IN AXIS ORDER
// (:68) -- a dependent chain, so ONE lane adds the exponentials out of LDS (16-byte reads, 16 values ahead of the chain; the row
// is zero-padded to a multiple of 32, and x + 0.0f == x); o / sum; y = round(o / out_scale), clamp +-127.
// exp runs in fp64 here too (ocml, <= 1 ulp): see softmax_u8_kernel (u8_kernels.hip) for what that can and cannot change.
// A TEAM of TW waves works on one position (NHWC keeps the axis contiguous); TPB teams per block.  Two forms: <1, 4> for short
// axes on many positions (class scores over a map: the waves of a block never meet, a wave past the last position just leaves)
// and <4, 1> for long axes on few positions (ResNet-50's prob, 1000 channels x batch: the fp64 exponentials and the two
// divisions per output spread over 256 threads; 17.4 -> see profiles/r04_softmax_i8_resnet50_b32*.txt).
template <int TW, int TPB>
__global__ __launch_bounds__(64 * TW * TPB) void softmax_i8_kernel(SoftmaxI8Args a)
{
    static_assert(TW == 1 || TPB == 1, "a multi-wave team owns its block (block-level barriers)");
    extern __shared__ __attribute__((aligned(16))) float softmax_e[];
    constexpr int T = 64 * TW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tid = TW == 1 ? lane : (int)threadIdx.x, team = TW == 1 ? wave : 0;
    const long p = (long)blockIdx.x * TPB + team;
    if (p >= a.positions) return;                      // whole teams
    const int pitch = (a.C + 31) & ~31;                // + 16 floats the chain's read-ahead touches and never adds
    float* e = softmax_e + (size_t)team * (pitch + 16);
    float* red = softmax_e + (size_t)TPB * (pitch + 16);          // TW > 1: the waves' maxima [TW], then the sum [TW]
    auto team_sync = [&]() {
        if constexpr (TW == 1) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        } else {
            __syncthreads();
        }
    };
    // where position p starts and how far apart the axis' elements are (kernels.h: SoftmaxI8Args; d1 == 0: the contiguous channel axis)
    const long xs = a.d1 > 0 ? a.istride : 1, ys = a.d1 > 0 ? a.ostride : 1;
    const int8_t* x = a.d1 > 0 ? a.x + (size_t)((p / a.d1) * a.is1 + ((p % a.d1) / a.d2) * a.is2 + (p % a.d2)) : a.x + (size_t)p * a.cs_in;
    int8_t* y = a.d1 > 0 ? a.y + (size_t)((p / a.d1) * a.os1 + ((p % a.d1) / a.d2) * a.os2 + (p % a.d2)) : a.y + (size_t)p * a.cs_out;
    float mx = -__builtin_inff();
    for (int j = tid; j < a.C; j += T) mx = fmaxf(mx, __fmul_rn((float)x[j * xs], a.in_scale));
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d, 64));
    if constexpr (TW > 1) {
        if (lane == 0) red[wave] = mx;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < TW; w++) mx = fmaxf(mx, red[w]);
    }
    for (int j = tid; j < pitch; j += T)
        e[j] = j < a.C ? (float)exp((double)__fsub_rn(__fmul_rn((float)x[j * xs], a.in_scale), mx)) : 0.f;
    team_sync();
    float sum = 0.f;
    if (tid == 0) {
        const float4* q = reinterpret_cast<const float4*>(e);
        float4 b0[4], b1[4];
#pragma unroll
        for (int k = 0; k < 4; k++) b0[k] = q[k];
        for (int j = 0; j < pitch; j += 32) {
#pragma unroll
            for (int k = 0; k < 4; k++) b1[k] = q[(j >> 2) + 4 + k];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                sum = __fadd_rn(sum, b0[k].x); sum = __fadd_rn(sum, b0[k].y); sum = __fadd_rn(sum, b0[k].z); sum = __fadd_rn(sum, b0[k].w);
            }
#pragma unroll
            for (int k = 0; k < 4; k++) b0[k] = q[(j >> 2) + 8 + k];            // the last round reads the 16 spare floats
#pragma unroll
            for (int k = 0; k < 4; k++) {
                sum = __fadd_rn(sum, b1[k].x); sum = __fadd_rn(sum, b1[k].y); sum = __fadd_rn(sum, b1[k].z); sum = __fadd_rn(sum, b1[k].w);
            }
        }
        if constexpr (TW > 1) red[TW] = sum;
    }
    if constexpr (TW == 1) sum = __shfl(sum, 0, 64);
    else { __syncthreads(); sum = red[TW]; }
    for (int j = tid; j < a.C; j += T) y[j * ys] = (int8_t)round_sat(__fdiv_rn(__fdiv_rn(e[j], sum), a.out_scale));
}

hipError_t launch_softmax_i8(const SoftmaxI8Args& a, hipStream_t s)
{
    if (a.C < 1 || a.C > kSoftmaxI8MaxC) return hipErrorInvalidValue;
    const size_t row = (size_t)(((a.C + 31) & ~31) + 16) * sizeof(float);
    if (a.C <= 256) {
        hipLaunchKernelGGL((softmax_i8_kernel<1, 4>), dim3((unsigned)((a.positions + 3) / 4)), dim3(256), 4 * row, s, a);
    } else {
        hipLaunchKernelGGL((softmax_i8_kernel<4, 1>), dim3((unsigned)a.positions), dim3(256), row + 8 * sizeof(float), s, a);
    }
    return hipGetLastError();
}

// ---- layout at the subgraph edges (the IR is NCHW: source/operator/prototype/convolution.c:60-70) ----
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(LayoutArgs a)
{
    const long total = (long)a.N * a.H * a.W * a.cs;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % a.cs);
    long p = idx / a.cs;
    const int w = (int)(p % a.W); p /= a.W;
    const int h = (int)(p % a.H);
    const int n = (int)(p / a.H);
    T v = 0;
    if (c < a.C) v = reinterpret_cast<const T*>(a.src)[(((size_t)n * a.C + c) * a.H + h) * a.W + w];
    reinterpret_cast<T*>(a.dst)[idx] = v;
}

template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(LayoutArgs a)
{
    const long total = (long)a.N * a.C * a.H * a.W;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int w = (int)(idx % a.W);
    long p = idx / a.W;
    const int h = (int)(p % a.H); p /= a.H;
    const int c = (int)(p % a.C);
    const int n = (int)(p / a.C);
    reinterpret_cast<T*>(a.dst)[idx] = reinterpret_cast<const T*>(a.src)[(((size_t)n * a.H + h) * a.W + w) * a.cs + c];
}

hipError_t launch_nchw_to_nhwc(const LayoutArgs& a, hipStream_t s)
{
    const long total = (long)a.N * a.H * a.W * a.cs;
    dim3 g((unsigned)((total + 255) / 256));
    if (a.elem == 1) hipLaunchKernelGGL(nchw_to_nhwc_kernel<int8_t>, g, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, g, dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_nhwc_to_nchw(const LayoutArgs& a, hipStream_t s)
{
    const long total = (long)a.N * a.C * a.H * a.W;
    dim3 g((unsigned)((total + 255) / 256));
    if (a.elem == 1) hipLaunchKernelGGL(nhwc_to_nchw_kernel<int8_t>, g, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, g, dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace tamd
