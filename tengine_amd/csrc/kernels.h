// Device-kernel argument blocks and host launchers (private to the backend).
// Device layout: activations are NHWC int8 with a channel stride `cs` that is a multiple of 16 bytes
// (so every pixel row is 16-B aligned and K-contiguous for MFMA operand loads); conv weights are
// repacked once at prerun to [cout_pad][kpad] with k = (ky*KW+kx)*cin_pad + ci, zero padded.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

namespace tamd {

struct EltFuse {              // eltwise (+ReLU) node applied in the conv epilogue (epilogue.h: fuse_elt4)
    const int8_t* res;        // the other eltwise operand (NHWC, same pixels); nullptr: no fusion
    int res_ldc, res_c_off;   // its channel stride / offset
    int type;                 // ELT_* (0 prod, 2 sum, 4 sub, 6 max)
    int conv_is_first;        // the conv output is eltwise input 0 (matters for sub)
    float s_conv, s_res;      // scales of the conv output tensor and of the residual tensor
    float out_scale;          // eltwise output scale
    int relu;                 // a ReLU (slope 0) node follows
    float relu_out_scale;
    // SUM (+ ReLU that keeps the eltwise scale) in two fused multiply-adds per value (epilogue.h: elt_sum16_fold), folded by the
    // planner: mc = RN32(s_conv / out_scale), mr = RN32(s_res / out_scale), k0 = RN32(128.5 + e - 128 * (mc + mr)), window
    // [ylo, yhi] = [128 + (relu ? 0 : -127) + 0.25, 255.75], hand-over threshold thr = 2e; thr = 0: not applicable
    float mc, mr, k0, ylo, yhi, thr;
};

// Requantisation constants of one conv / FC node, folded by the planner (graph_plan.hip: fold_requant + host_rq; the arithmetic and
// its exactness argument are in epilogue.h).  The kernels' per-channel vector `wscale[]` holds the FAST-path multiplier
// M[c] = RN32(m1 * m2[c] / out_scale); the reference chain's own factors stay here for the values the fast path hands over.
struct RqArgs {
    float m1, lo, hi, out_scale;   // reference chain: f = clamp(fl(fl((float)acc * m1) * m2[c]), lo, hi), q = sat127(round(f / out_scale));
                                   // lo / hi already fold the +-127.49 * out_scale saturation
    float ylo, yhi;                // fast path: clamp of the biased value, 128 + q(lo) + 0.25 / 128 + q(hi) + 0.75
    float thr;                     // fast path: hand over when fract(y) < thr (2^-13; 2.0 = always, when the constants are out of range)
    const float* m2;               // [as wscale] m2[c] of the reference chain, read on the hand-over path only
};

struct ConvArgs {
    const int8_t* x;       // NHWC input, channel stride cs_in
    const int8_t* w;       // packed weights [cout_pad][kpad]
    const int32_t* bias;   // [cout_pad] (zeros when the node has no bias)
    const float* wscale;   // [cout_pad] fast-path multipliers M[c] (see RqArgs)
    int8_t* y;             // NHWC output base
    int N, H, W, cs_in;    // cs_in: bytes between consecutive input pixels
    int ckp;               // K bytes per tap = roundup(cin,16) (== cs_in unless x is a concat view)
    int OH, OW;
    int cout;              // logical output channels
    int ldc;               // output channel stride (bytes per pixel)
    int c_off;             // channel offset inside the output pixel (concat-by-offset)
    int c_limit;           // channels [0,c_limit) of this conv may be stored (multiple of 4)
    int KH, KW, SH, SW, PH, PW, DH, DW;
    int cin;               // logical input channels per group
    int ktot;              // KH*KW*ckp
    int kpad;              // weight row stride (multiple of the K tile)
    int M;                 // N*OH*OW
    RqArgs rq;             // requantisation constants, see epilogue.h (wscale[] holds M[c])
    const int8_t* zeros;   // >= 16 zero bytes (source of out-of-image taps for the LDS-DMA kernel)
    unsigned long long mg_ohw, mg_ow;   // ceil(2^40 / (OH*OW)), ceil(2^40 / OW): pixel index -> (n, oy, ox) by multiply-high (conv_igemm_fast.h)
    int cfg;               // tile configuration of the chosen GEMM kernel (-1: the launcher's heuristic), set by the planner
    // conv_pgemm.hip only (conv_pgemm_prepare / the planner fill these; other kernels ignore them)
    const int8_t* wfrag;   // weights in MFMA fragment order: [cout tile of BN][stage][BN/32 x 2 fragments][64 lanes][16 B]
    int pg_ns;             // 64-deep K stages; stage = (64-channel chunk) * KH*KW + tap
    int pg_variant;        // tile shape, see conv_pgemm_bn()
    int pg_npad;           // patch pixels of the worst pixel tile, rounded up to 64 (k x k convolutions)
    int pg_hp, pg_wp;      // rows per image / columns of the virtual padded input the patch is cut from
    unsigned long long mg_hp, mg_wp;     // ceil(2^40 / pg_hp), ceil(2^40 / pg_wp)
    // conv_pgemm_w.hip only (conv_pgemm_w_prepare / conv_pgemm_w_table)
    const int* pg_tab;     // per pixel tile: [pg_npad] NHWC pixel index of each patch unit (-1: zeros), [BM] patch origin | edge bits << 16 of each output pixel
    int pg_ts;             // dwords per tile of pg_tab
    int pg_tiles_m, pg_tiles_n;
    int pg_zarea;          // bytes of the zero area behind each patch buffer (edge taps)
    unsigned mg_tn;        // ceil(2^32 / pg_tiles_n) (pg_tiles_n > 1)
#ifdef TAMD_IGEMM_STAMPS
    long long* dbg_stamps; // tools/exp/igemm_anatomy.hip only: s_memtime at the stage boundaries of wave 0 of block 0
    int dbg_flags;         // .. ablation: 1 no MFMA, 2 no LDS traffic, 4 no global loads
#endif
    EltFuse elt;           // fused eltwise(+relu) tail (conv_igemm / conv_igemm2 only); y/ldc/c_off then describe ITS output
};

struct DwArgs {
    const int8_t* x;       // NHWC
    const int8_t* w;       // [3 rows][cw] dwords {w[r][0],w[r][1],w[r][2],0}, cw = roundup(C,16)
    const int32_t* bias;   // [cw]
    const float* wscale;   // [cw]
    int8_t* y;
    int N, H, W, C, cs_in, cw, OH, OW, ldc, c_off;
    int S, PH, PW;
    RqArgs rq;             // requantisation constants, see epilogue.h (wscale[] holds M[c])
};

struct DwPwArgs {          // depthwise 3x3 (stride 1) -> pointwise 1x1 in one launch (dwpw.hip): the depthwise result only exists in LDS
    const int8_t* x;       // NHWC input of the depthwise conv (channel offset applied)
    const int8_t* dw_w;    // DwArgs::w
    const int32_t* dw_bias; const float* dw_wscale; RqArgs dw_rq;
    const int8_t* pw_wfrag;        // pointwise weights in A-fragment order (dwpw_pack)
    const int32_t* pw_bias; const float* pw_wscale; RqArgs pw_rq;
    int8_t* y;             // NHWC output of the pointwise conv
    int N, H, W, C, cs_in, cw, OH, OW, PH, PW;     // depthwise geometry (C channels in and out, cw = roundup(C, 16))
    int cout, ldc, c_off, c_limit;                 // pointwise output
#ifdef TAMD_DWPW_STAMPS
    unsigned long long* stamps;                    // tools/exp/dwpw_anatomy.hip only: 16 per block
    int ablate;                                    // .. 1: no MFMAs, 2: no depthwise arithmetic, 4: no epilogue requantisation, 8: no A-fragment loads after stage 0, 16: no tap loads, 32: no B-fragment reads
#endif
};

struct PwDwArgs {          // pointwise conv + its consumer in one launch (pwdw.hip)
    const int8_t* x;       // NHWC input of the pointwise conv (channel offset applied)
    const int8_t* wf;      // pointwise weights in MFMA fragment order: [16-channel slice][64-deep K step][64 lanes][16 B]
    const int32_t* bias;   // [slices * 16]
    const float* wscale;   // [slices * 16] (M[c] of epilogue.h)
    RqArgs rq;             // pointwise requantisation constants
    int N, H, W, cs_in, ktot;          // H x W: the pointwise map == the tail's input map
    int nsteps, steps;                 // 64-deep K steps of the (zero padded) weight panel, a multiple of `steps` = pwdw_steps()
#ifdef TAMD_PWDW_STAMPS
    unsigned long long* stamps;        // tools/exp/pwdw_anatomy.hip only
#endif
    int coherent;          // 1: the coherent instance (agent-scope loads of x, write-through stores of y): its launch needs no cache
                           // maintenance at its boundaries (pwdw.hip: pwdw_i8_coh_kernel; set when the graph dispatches directly)
    int mode;              // tail: 0 global pooling, 1 depthwise 3x3, 2 none (the tile results are stored)
    int prod;              // producer: 0 pointwise conv of an NHWC tensor, 1 first-layer conv gathered from the NCHW graph input
    const unsigned* taps;  // prod 1: [16] patch row (c, ky) -> (c*in_H*in_W + ky*DH*in_W) | ky*DH << 28, zero padded; k = row*4 + kx
    int in_C, in_H, in_W;  // prod 1: the NCHW input;  H x W above is then the first conv's OUTPUT map
    int fSH, fSW, fPH, fPW;            // prod 1: stride, leading pads
    int sl;                // 16-channel slices per block: 2 | 4 (depthwise tails with a register-resident K; slices = cout / (16 sl)), else one.
                           // HERE, in what was a 4-byte alignment hole: the struct must not grow -- its size decides which 64-byte line of the
                           // argument segment holds the hidden block size the kernels read late, and 8 bytes more cost every batch-1 launch
                           // with a depthwise tail 0.15-0.2 us (a cold scalar-cache line on the critical path;
                           // profiles/r05_ab_b1_call12_vs_now.txt: identical instructions, +3-6 % per launch)
    const int8_t* dw_w;    // as DwArgs::w
    const int32_t* dw_bias;
    const float* dw_wscale;
    RqArgs d_rq;
    int cw;                // depthwise weight row length (channels rounded up to 16)
    int S, PH, PW, OH, OW; // depthwise stride / leading pads / output map
    int8_t* y;             // NHWC output of the tail
    int ldc, c_off, c_limit;
    int TH, TW, tiles_y, tiles_x, slices;      // depthwise output tile per block, grid (slices: blocks along the channels)
    int tile_major;        // 1: grid = (tile_x, tile_y * N, slice) instead of (slice, tile_x, tile_y * N): which operand an XCD's L2 shares
    int RH, RW;            // input region of a tile: (TH-1)*S+3, (TW-1)*S+3
    int pool_method;       // mode 0: 0 max, 1 avg
    float p_in_scale, p_out_scale;
};

#ifdef TAMD_PWDW_CHAIN_EXPERIMENT      // tools/exp/chain_anatomy.hip only (why the chained launch is not in the product)
struct PwChainArgs {       // several PwDwArgs layers as ONE launch, ordered by counters (pwdw.hip: pwdw_chain_kernel)
    const PwDwArgs* layers;    // device array [nlayers]
    int nlayers;
    int first_block[17];       // blocks of layer l: [first_block[l], first_block[l + 1])
    short variant[16];         // kernel variant of layer l: MODE | PROD << 3 | steps << 4 (pwdw_chain_variant)
    short gx[16], gy[16];      // its grid (x, y; z follows from the block count)
    int* flags;                // [32 * nlayers] finished-block counter of layer l at [32 * l] (a cache line each); never reset
    int* sync;                 // [0] epoch, [1] finished blocks of the last layer, [2] a bounded wait gave up
};
#endif

struct DirectArgs {        // generic direct conv (any group / cin), also NCHW-input first layers
    const int8_t* x;
    const int8_t* w;       // OIHW as in the model
    const int32_t* bias;   // may be null
    const float* wscale;
    int8_t* y;             // NHWC
    int N, C, H, W, cs_in; // cs_in == 0 => x is NCHW (graph input), else NHWC with that stride
    int OH, OW, cout, ldc, c_off;
    int KH, KW, SH, SW, PH, PW, DH, DW, group;
    RqArgs rq;             // requantisation constants, see epilogue.h (wscale[] holds M[c])
};

struct FirstArgs {         // first layer from the NCHW graph input (C <= 4) on MFMA
    const int8_t* x;       // NCHW
    const int8_t* w;       // [cout_pad32][kp], k = (c*KH+ky)*KW+kx (OIHW order), zero padded
    const int32_t* bias;   // [cout_pad32]
    const float* wscale;   // [cout_pad32]
    int8_t* y;             // NHWC
    int N, C, H, W, OH, OW, cout, ldc, c_off, c_limit;
    int KH, KW, SH, SW, PH, PW, DH, DW;
    int kp;                // roundup(C*KH*KW, 32) <= 256
    int kwp;               // 0: k = OIHW order (gather kernel); 4 / 8: k = (c*KH+ky)*kwp + kx, kp = roundup(C*KH*kwp, 32) <= 192
    RqArgs rq;             // requantisation constants, see epilogue.h (wscale[] holds M[c])
};

struct PoolArgs {
    const int8_t* x; int8_t* y;
    int N, H, W, C, cs_in, OH, OW, ldc, c_off;
    int KH, KW, SH, SW, PH, PW;
    int method, caffe_flavor;
    float in_scale, out_scale;
};

struct FirstPoolArgs {     // the stem in one launch: FirstArgs' convolution (7 x KW, stride 2, C = 3) + MAX pool 3x3 / 2 / pad 0 (conv_first_pool.hip)
    const int8_t* x;       // NCHW graph input
    const int8_t* w;       // FirstArgs::w with kwp == 8
    const int32_t* bias; const float* wscale;
    int8_t* y;             // NHWC pooled map
    int N, C, H, W, OH, OW, cout;      // OH x OW: the conv map (never stored)
    int KH, KW, PH, PW, kp;
    int POH, POW, ldc, c_off;          // pooled map and its destination
    float pool_in_scale, pool_out_scale;
    RqArgs rq;
};

struct EltArgs {
    const int8_t* a; const int8_t* b; int8_t* y;
    size_t count;          // bytes (padded NHWC buffers, identical geometry)
    int type;
    float sa, sb, out_scale;
    int fuse_relu;         // 1: apply the following standalone ReLU node (slope 0) in the same pass
    float relu_out_scale;
};

struct ReluArgs {
    const int8_t* x; int8_t* y; size_t count; float slope, in_scale, out_scale;
};

struct SoftmaxI8Args {     // softmax over ONE axis of an int8 tensor: the channels of an NHWC tensor (a view's channel slice included), or --
                           // round 6 -- any axis of a dense (reference-order) or NHWC tensor
    const int8_t* x; int8_t* y;
    long positions;        // the (outer, inner) index pairs: N * H * W for the channel axis
    int C, cs_in, cs_out;  // C = length of the axis.  Position p starts at p * cs_in / p * cs_out ...
    float in_scale, out_scale;
    // ... unless d1 > 0 (the general form): start = (p / d1) * s1 + ((p % d1) / d2) * s2 + (p % d2), element j of the axis `stride`
    // bytes further on; i-fields for the input, o-fields for the output.  Dense [outer][A][inner]: d1 = d2 = inner, s1 = A * inner, s2 = 0,
    // stride = inner.  NHWC over W: d1 = d2 = C, s1 = W * cs, stride = cs.  NHWC over H: d1 = W * C, d2 = C, s1 = H * W * cs, s2 = cs,
    // stride = W * cs.
    long d1, d2;
    long is1, is2, os1, os2;
    long istride, ostride;
};
constexpr int kSoftmaxI8MaxC = 16000;      // the axis' exponentials live in LDS as floats (<= 64 KB with the chain's padding)

struct CatCopyArgs {       // one concat input that cannot be written in place: (re-scaling) copy into its channel slice
    const int8_t* x; int8_t* y;
    long pixels;
    int C, cs_in, ldc, c_off;
    float rescale;         // in_scale / out_scale
    int identity;          // plain byte copy (single-input concat)
};

// ---- int8 tensors in the reference's DENSE element order (round 6: the SSD head plumbing of an int8 graph -- Permute(0,2,3,1), Flatten,
// Reshape, Concat on any axis, PriorBox).  One launch copies up to kFlatCatMax inputs of a Concat into their slots of every outer slice
// of the dense output, each read in the element order the reference sees and re-scaled as concat_kernel_ref_int8.c does.
struct FlatCatI8Src {
    const int8_t* x;
    int kind;              // 0: dense, element e of outer slice o at o * chunk + e
                           // 1: NHWC tensor seen through Permute(0,2,3,1) (+ Flatten): e = p * C + c   -> (o * HW + p) * cs + c
                           // 2: NHWC tensor in NCHW element order: L = o * chunk + e = (n * C + c) * HW + p   -> (n * HW + p) * cs + c
    int chunk;             // elements per outer slice
    int C, HW, cs;
    int begin;             // first element of this input inside an output slice
    float rescale;         // in_scale / out_scale
    int identity;          // plain byte copy (single-input concat, a lone Permute / layout copy)
};
constexpr int kFlatCatMax = 8;
struct FlatCatI8Args {
    FlatCatI8Src src[kFlatCatMax];
    int nsrc;
    int8_t* y;
    long outer;
    int out_row;           // elements of one outer slice of the output
    int row_begin, row_len;   // this launch covers [row_begin, row_begin + row_len) of every slice
};

struct LayoutArgs {        // NCHW <-> NHWC(cs) int8 / generic element size
    const void* src; void* dst; int N, C, H, W, cs; int elem;
};

// launchers (return hipError_t of the launch)
hipError_t launch_conv_igemm(const ConvArgs& a, hipStream_t s);
const char* conv_igemm_kernel_name(const ConvArgs& a);   // tile shape the launcher will pick
int conv_igemm_num_cfgs();                               // ConvArgs::cfg values the autotuner may try
bool conv_igemm_cfg_ok(const ConvArgs& a, int cfg);
hipError_t launch_gemm_direct(const ConvArgs& a, hipStream_t s);   // 1x1, small-M / latency-bound shapes
bool gemm_direct_applicable(const ConvArgs& a);
hipError_t launch_conv_igemm2(const ConvArgs& a, hipStream_t s);  // LDS-DMA 3-stage ring, large problems
bool conv_igemm2_applicable(const ConvArgs& a);
const char* conv_igemm2_kernel_name(const ConvArgs& a);
// lean-loop implicit GEMM: fragment-ordered weights by LDS-DMA, k x k activations as an LDS-resident input patch (conv_pgemm.hip)
int conv_pgemm_num_variants();
int conv_pgemm_bn(int variant);                                    // cout tile of a variant (the weight packing unit)
bool conv_pgemm_applicable(const ConvArgs& a, int variant);
void conv_pgemm_prepare(ConvArgs& a, int variant);                 // fills pg_* / mg_hp / mg_wp
size_t conv_pgemm_packed_bytes(const ConvArgs& a, int bn);
void conv_pgemm_pack(const ConvArgs& a, const int8_t* w, int cout_pad, int bn, int8_t* out);   // w: [cout_pad][kpad] family layout
const char* conv_pgemm_kernel_name(const ConvArgs& a);
hipError_t launch_conv_pgemm(const ConvArgs& a, hipStream_t s);
// wave-grid form for 3x3 (conv_pgemm_w.hip): variants 16 .. 31 of the family above; the same packed weights, plus a geometry table
bool conv_pgemm_w_applicable(const ConvArgs& a, int variant);
void conv_pgemm_w_prepare(ConvArgs& a, int variant);               // fills pg_* except pg_tab
void conv_pgemm_w_table(const ConvArgs& a, std::vector<int>& out); // after prepare: the table the planner uploads (pg_tab)
const char* conv_pgemm_w_kernel_name(const ConvArgs& a);
hipError_t launch_conv_pgemm_w(const ConvArgs& a, hipStream_t s);
hipError_t launch_pw_stream(const ConvArgs& a, hipStream_t s);     // 1x1, shallow K, many pixels
bool pw_stream_applicable(const ConvArgs& a);
hipError_t launch_pw_rows(const ConvArgs& a, hipStream_t s);       // 1x1, shallow K, many pixels: row-major epilogue, persistent pipelined waves
bool pw_rows_applicable(const ConvArgs& a);
hipError_t launch_conv_first(const FirstArgs& a, hipStream_t s);
bool dwpw_applicable(const DwArgs& d, const ConvArgs& p);
size_t dwpw_packed_bytes(int cout, int cin);
void dwpw_pack(const int8_t* w, int cout, int cin, int8_t* out);        // w: [cout][cin] (OIHW, 1x1)
hipError_t launch_dwpw(const DwPwArgs& a, hipStream_t s);
bool conv_first_pool_applicable(const FirstArgs& c, const PoolArgs& p);
FirstPoolArgs conv_first_pool_args(const FirstArgs& c, const PoolArgs& p);
hipError_t launch_conv_first_pool(const FirstPoolArgs& a, hipStream_t s);
int conv_first_kwp(int C, int KH, int KW, int DW);
hipError_t launch_dwconv3x3(const DwArgs& a, hipStream_t s);
const char* dwconv3x3_kernel_name(const DwArgs& a);   // variant <stride, fragments per row> the launcher will pick
hipError_t launch_conv_direct(const DirectArgs& a, hipStream_t s);
hipError_t launch_pwdw(const PwDwArgs& a, int threads, hipStream_t s);
size_t pwdw_lds_bytes(const PwDwArgs& a, int threads);
#ifdef TAMD_PWDW_CHAIN_EXPERIMENT
int pwdw_chain_variant(const PwDwArgs& a, int threads, int* gx, int* gy, int* gz);
hipError_t launch_pwdw_chain(const PwChainArgs& c, int threads, size_t lds, hipStream_t s);
#endif
bool pwdw_config_ok(const PwDwArgs& a, int threads);
int pwdw_steps(int nsteps);
hipError_t launch_pool(const PoolArgs& a, hipStream_t s);
hipError_t launch_eltwise(const EltArgs& a, hipStream_t s);
hipError_t launch_relu(const ReluArgs& a, hipStream_t s);
hipError_t launch_softmax_i8(const SoftmaxI8Args& a, hipStream_t s);
hipError_t launch_concat_copy_i8(const CatCopyArgs& a, hipStream_t s);
hipError_t launch_flatcat_i8(const FlatCatI8Args& a, hipStream_t s);
hipError_t launch_copy_bytes(void* dst, const void* src, size_t bytes, hipStream_t s);   // 16-byte aligned buffers
hipError_t launch_nchw_to_nhwc(const LayoutArgs& a, hipStream_t s);
hipError_t launch_nhwc_to_nchw(const LayoutArgs& a, hipStream_t s);

// ---- uint8 (per-tensor asymmetric) ---------------------------------------------------------------------------
// The reference SIMULATES uint8 in fp32: operands are dequantised, the convolution is an fp32 GEMM
// whose per-element summation order is fixed by the reference's register tiling, and only the result is
// requantised.  Byte-identical outputs therefore need the same fp32 operations in the same order: one fused
// multiply-add chain per output element -- which is exactly how v_mfma_f32_16x16x4f32 accumulates (measured,
// profiles/r01_mfma_f32_is_sequential_fma_chain.txt), so the path runs on the matrix cores (u8_conv_*.hip, u8_kernels.hip).
// uint8 activations stay in the reference's dense NCHW order on the device.
struct U8Q { float scale; int zp; };
// a ReLU node folded into the producing conv: applied to the conv's own uint8 result in registers
// (relu_kernel_ref_uint8.c:48-95 on that byte), so the bytes equal the two-launch sequence
struct U8Relu { int on; float slope; U8Q out; };

struct U8PoolFuse {            // a 2x2 / stride 2 / unpadded MAX-pool node applied to the conv's final bytes in the conv epilogue
    int on;                    // the kernel then enumerates output pixels window-major: j = 4 * window + 2 * dy + dx (so the four
                               // pixels of a window sit in four neighbouring lanes); needs OH, OW even and OH*OW % 8 == 0
    int write_full;            // the unpooled tensor has other readers (or is a graph output): store it as well
    uint8_t* y;                // pooled output, NCHW, image stride out_img bytes, first channel at out_c0
    int out_img, out_c0;
    U8Q in, out;               // quantisation of the pool node's input (== what the conv epilogue produced) and output
};

struct U8ConvArgs {            // group == 1: conv_kernel_x86.c sgemm_fp order (u8_conv_gemm.hip: conv_u8_gemm)
    const uint8_t* x;          // NCHW
    const uint8_t* wq;         // raw uint8 weights, [cout tile of BM][stage of 32 k][BM rows][32 slots]; BM =
                               // conv_u8_gemm_bm(cfg); slot of k inside its stage: (k%4)*8 + (k%32)/4 (class-major);
                               // padding holds the weight zero point (dequantises to exactly 0)
    const unsigned* klut;      // [Kpad] packed tap table: (c*H*W + ky*DH*W + kx*DW) | kx*DW << 24 | ky*DH << 28
                               // (padding rows: 0 -- their weights are 0 and fma(x, 0, s) == s for finite x)
    const int32_t* bias;       // may be null
    uint8_t* y;                // NCHW, image stride out_img bytes, first channel at out_c0
    int N, C, H, W, OH, OW, cout, cout_pad, K, Kpad;
    int SH, SW, PH, PW;
    int out_img, out_c0;       // bytes per output image / channel offset (concat outputs are written in place)
    int cfg;                   // conv_u8_gemm_pick(): block tile (channels x pixels)
    int m_blocked;             // rows below this sit in an 8- or 4-row block of the reference's sgemm_fp
    float in_scale, in_zp;     // zero point as float (exact)
    float w_scale, w_zp;
    float bias_scale;          // in_scale * w_scale
    int act;
    float out_scale; int out_zp;
    U8Relu relu;               // fused ReLU / leaky ReLU node (y, out_img, out_c0 then describe ITS output)
    const float* wf;           // conv_u8_rgb3x3 only: dequantised weights, [cout][wf_ld] rows in OIHW k order
    int wf_ld;
    U8PoolFuse pool;           // fused max-pool node (conv_u8_gemm main tiles and conv_u8_rgb3x3)
    // conv_u8_patch (u8_conv_patch.hip): the main pixels (j < (OH*OW)&~7) from an LDS-resident fp32 input patch
    const uint8_t* wpk;        // DEQUANTISED weights (floats) in MFMA A-fragment order, [16-row tile][super-step of 4*SS k][float4 group][lane]
    int pk_cfg;                // -1: not used; else tile configuration of launch_conv_u8_patch
    int pk_npad, pk_wp;        // floats per channel plane of the patch (3x3: 256 | 512, 1x1: the pixel tile) / patch row pitch (input columns incl. halo)
    int pk_kh, pk_kw, pk_dh, pk_dw;    // filter shape / dilation (the GEMM kernel gets them through klut)
    int pk_tw;                 // 0: a pixel tile is a run of consecutive pixels (rows of the whole map width in the patch); > 0 (round 4): 2-D pixel
                               // tiles of 8 rows x pk_tw columns (= the configuration's pixel-tile size / 8), the patch is the tile + its halo
    // conv_u8i (u8i_kernels.hip): the opt-in INTEGER path -- exact int32 sums on the int8 MFMA, results within one step of the reference
    const int8_t* iw;          // (w ^ 0x80) in MFMA A-fragment order [cout tile][step = (32-channel chunk, tap)][32-row fragment][lane][16 B]
    const int32_t* icv;        // per output channel: bias - alpha * sum_k w'_k + Kp * alpha * beta, padded to the cout tile
    int i_cfg, i_npad, i_nchunks;      // tile configuration / pixels per patch granule plane (128 | 256 | 512) / 32-channel chunks
    int i_cgs;                 // log2 of the 32-channel groups a chunk (= one barrier) holds: 0 | 1 | 2
    int i_dbg;                 // anatomy runs only (TAMD_U8I_ABLATE): 1 no stores, 2 no requantisation, 4 no input loads -- wrong bytes by design
    int i_tw;                  // 0: linear pixel tiles; 8 | 16: 2-D tiles of this width (maps too wide for a linear tile's bounding box)
    int i_alpha, i_beta;       // in_zp - 128, w_zp - 128
    float i_m;                 // requantisation multiplier fl(fl(in_scale * w_scale) / out_scale) (u8_epilogue.h: u8i_requant)
    int i_qlo, i_qhi;          // clamp window of the result: [0, 255] narrowed by the conv's own activation
};

struct U8DirectArgs {          // grouped / depthwise: conv_kernel_ref_uint8.c order (conv_u8_direct), also FC
    const uint8_t* x; const float* wf;   // wf: [cout][cin_g*KH*KW] dequantised weights, OIHW order
    const int32_t* bias; uint8_t* y;
    int N, C, H, W, OH, OW, cout, KH, KW, SH, SW, PH, PW, DH, DW, group;
    int out_img, out_c0;
    float in_scale, in_zp, w_scale;
    int act;
    float out_scale; int out_zp;
    U8Relu relu;
};

struct U8FcArgs {              // fc_ref.c:121-207
    const uint8_t* x; const float* wf;   // wf: [hidden][nout_pad] dequantised, zero padded
    const int32_t* bias; uint8_t* y;
    int batch, hidden, nout, nout_pad;
    float in_scale, in_zp, bias_scale, out_scale; int out_zp;
};

struct U8PoolArgs {
    const uint8_t* x; uint8_t* y;
    int N, C, H, W, OH, OW, KH, KW, SH, SW, PH, PW, method, caffe_flavor;
    U8Q in, out;
};

struct U8MapArgs {             // relu / leaky, concat slice copy, nearest upsample: one output byte per input byte
    const uint8_t* x; uint8_t* y;
    int N, C, H, W;            // INPUT geometry
    int out_img, out_c0;       // output image stride (bytes) / channel offset (concat)
    int scale;                 // upsample factor (1 otherwise)
    float slope;               // relu
    U8Q in, out;
};

struct U8CatArgs {             // flat per-image slice copy: concat on axis 1 of any rank, Permute(0,2,3,1), or both at once
    const uint8_t* x; uint8_t* y;
    int N, in_img;             // images, bytes per input image
    int perm_c, perm_p;        // 0: source index == j; else source = (j % perm_c) * perm_p + j / perm_c  (NCHW read in NHWC order)
    int out_img, out_off;      // output image stride / byte offset of this slice inside the output image
    int identity;              // 1: plain byte copy (permute, or concat input carrying the output's scale and zero point)
    U8Q in, out;
};

struct U8CatMulti {            // every input of one concat node in ONE launch (blockIdx.z = input): MobileNet-SSD's mbox_loc / mbox_conf
    U8CatArgs src[8];
    float rescale[8];
    int count;
};
hipError_t launch_flatcat_multi_u8(const U8CatMulti& m, hipStream_t s);

struct U8SoftmaxArgs {         // softmax over the middle axis of [outer][on][inner]
    const uint8_t* x; uint8_t* y;
    int outer, on, inner;
    U8Q in, out;
};

struct U8EltArgs {
    const uint8_t* a; const uint8_t* b; uint8_t* y; size_t count; int type;
    U8Q qa, qb, out;
};

int conv_u8_gemm_pick(const U8ConvArgs& a);        // geometry fields only
int conv_u8_gemm_bm(int cfg);                      // channel rows per block tile (weight packing unit)
int conv_u8_gemm_kc(int cfg);                      // K stage depth (weight packing unit)
int conv_u8_gemm_num_cfgs();
size_t conv_u8_gemm_lds(const U8ConvArgs& a);      // dynamic LDS bytes of the chosen configuration
hipError_t launch_conv_u8_gemm(const U8ConvArgs& a, hipStream_t s);
// main pixels from an LDS-resident fp32 patch (3x3 and 1x1, group 1); tail pixels, if any, on the VALU by extra blocks of the same launch
int conv_u8_patch_num_cfgs();
int conv_u8_patch_lanes_cfg();                                      // the configuration index of the lane-level chain kernel (conv_u8_lanes)
int conv_u8_patch_bm(int cfg);
int conv_u8_patch_ss(const U8ConvArgs& a);                         // MFMA steps per super-step (9: 3x3, 4: 1x1); 0: shape not supported
bool conv_u8_patch_prepare(U8ConvArgs& a, int cfg, int KH, int KW, int DH, int DW);     // fills pk_*; false: not applicable
const char* conv_u8_patch_kernel_name(const U8ConvArgs& a);
size_t conv_u8_patch_packed_bytes(const U8ConvArgs& a);
void conv_u8_patch_pack(const U8ConvArgs& a, const uint8_t* w, uint8_t w_zp, float w_scale, float* out);       // w: [cout][K] as in the model
hipError_t launch_conv_u8_patch(const U8ConvArgs& a, hipStream_t s);
// shallow pointwise layers of large maps (K = 32 | 64): weights resident in registers, B straight from the NCHW input, one wave per tile
bool conv_u8_pw_applicable(const U8ConvArgs& a, int KH, int KW);
const char* conv_u8_pw_kernel_name(const U8ConvArgs& a);
// shallow 3x3 layers of large maps (C = 16 | 32): weights resident in registers, the 3x3 gather straight from the NCHW input, one wave per tile
bool conv_u8_c3_applicable(const U8ConvArgs& a, int KH, int KW, int DH, int DW);
const char* conv_u8_c3_kernel_name(const U8ConvArgs& a);
hipError_t launch_conv_u8_c3(const U8ConvArgs& a, hipStream_t s);
hipError_t launch_conv_u8_pw(const U8ConvArgs& a, hipStream_t s);
// the integer path (u8i_kernels.hip); shares U8ConvArgs (geometry, fused ReLU / max-pool tails) with the byte-exact kernels
int conv_u8i_num_cfgs();
int conv_u8i_bm(int cfg);
bool conv_u8i_prepare(U8ConvArgs& a, int cfg, int KH, int KW, int DH, int DW);          // fills i_cfg / i_npad / i_nchunks / pk_k*; false: not applicable
size_t conv_u8i_packed_bytes(const U8ConvArgs& a, int bm);                          // bm: channel rows of a block tile (conv_u8i_bm / conv_u8i_pw_bm)
void conv_u8i_pack(const U8ConvArgs& a, int bm, const uint8_t* w, int w_zp, int in_zp, const int32_t* bias, int8_t* out, int32_t* cvec);    // w: [cout][K] as in the model
// pointwise layers (1x1, stride 1, no padding): register-only transposes, no LDS, dword stores
int conv_u8i_pw_num_cfgs();
int conv_u8i_pw_bm(int cfg);
int conv_u8i_pw_bn(int cfg);
bool conv_u8i_pw_prepare(U8ConvArgs& a, int cfg, int KH, int KW);
const char* conv_u8i_pw_kernel_name(const U8ConvArgs& a);
hipError_t launch_conv_u8i_pw(const U8ConvArgs& a, hipStream_t s);
const char* conv_u8i_kernel_name(const U8ConvArgs& a);
hipError_t launch_conv_u8i(const U8ConvArgs& a, hipStream_t s);
// first layers of the integer path (3x3, <= 4 input channels): the whole K in one v_mfma_i32_16x16x64_i8
bool conv_u8i_rgb_applicable(const U8ConvArgs& a, int KH, int KW, int DH, int DW);
void conv_u8i_rgb_prepare(U8ConvArgs& a);          // fills i_npad (patch pixels of a full 16x16 window)
size_t conv_u8i_rgb_packed_bytes(const U8ConvArgs& a);
void conv_u8i_rgb_pack(const U8ConvArgs& a, const uint8_t* w, int w_zp, int in_zp, const int32_t* bias, int8_t* out, int32_t* cvec);
hipError_t launch_conv_u8i_rgb(const U8ConvArgs& a, hipStream_t s);
bool conv_u8_rgb3x3_applicable(int cin, int kh, int kw, int dh, int dw, int group);
hipError_t launch_conv_u8_rgb3x3(const U8ConvArgs& a, hipStream_t s);
const char* conv_u8_rgb3x3_kernel_name(const U8ConvArgs& a);      // "conv_u8_rgb3x3_mfma" (main pixels on the matrix cores) | "conv_u8_rgb3x3"
const char* conv_u8_gemm_kernel_name(const U8ConvArgs& a);
hipError_t launch_conv_u8_direct(const U8DirectArgs& a, hipStream_t s);

// group == 1 convolution as an fp32 GEMM on the matrix cores, operands streamed with LDS-DMA (conv_f32_mfma.hip).
// Serves uint8 models (x = dequantised copy of the byte tensor, uint8 epilogue) and fp32 models (fp32 epilogue).
struct F32ConvArgs {
    const float* x;            // NCHW fp32
    const float* w;            // packed fp32 weights: [cout tile of BM][stage of 32 k][row group][64], see graph_u8.hip
    const unsigned* klut;      // [Kpad] packed tap table (as U8ConvArgs::klut)
    const float* zeros;        // >= 4 zero bytes: source of out-of-image taps
    const int32_t* bias;       // uint8 models (may be null)
    const float* bias_f32;     // fp32 models (may be null)
    uint8_t* y;                // uint8 output (NCHW) ...
    float* out_f32;            // ... or fp32 output when non-null
    int N, C, H, W, OH, OW, cout, K, Kpad;
    int SH, SW, PH, PW;
    int out_img, out_c0;       // elements per output image / channel offset
    int cfg, m_blocked;
    int tail_split;            // 1: uint8 models (the reference's tail-pixel summation order); 0: fp32 models (order-free)
    float bias_scale;
    int act;
    float out_scale; int out_zp;
};
int conv_f32_mfma_pick(const F32ConvArgs& a);      // geometry fields only
int conv_f32_mfma_bm(int cfg);
size_t conv_f32_mfma_lds(int cfg);
const char* conv_f32_mfma_kernel_name(const F32ConvArgs& a);
hipError_t launch_conv_f32_mfma(const F32ConvArgs& a, hipStream_t s);
hipError_t launch_dequant_u8_f32(const uint8_t* x, float* y, size_t n, float zp, float scale, hipStream_t s);

// Winograd F(2x2, 3x3) for fp32 3x3 / stride 1 / dilation 1 / group 1 convolutions (winograd_f32.hip): three launches
struct F32WinoArgs {
    const float* x;            // NCHW input
    const float* U;            // transformed weights [16][Mpad][Cpad] (host: G g G^T), zero padded
    const float* bias;         // may be null
    float* V;                  // workspace [16][Cpad][Tpad]: transformed input tiles
    float* M;                  // workspace [16][Mpad][Tpad]: products
    float* y;                  // NCHW output, image stride out_img elements, first channel out_c0
    int N, C, H, W, OH, OW, cout, PH, PW;
    int TH, TW, T, Tpad;       // tiles per image (rows, columns), tiles in total, padded to 64
    int Cpad, Mpad;            // cin padded to 16, cout padded to 64
    int out_img, out_c0, act;
};
hipError_t launch_wino_in_f32(const F32WinoArgs& a, hipStream_t s);
hipError_t launch_wino_gemm_f32(const F32WinoArgs& a, hipStream_t s);
hipError_t launch_wino_out_f32(const F32WinoArgs& a, hipStream_t s);

// ---- fp32 models (f32_kernels.hip): dense NCHW fp32 ------------------------------------------------------------------
struct F32DirectArgs {         // grouped / depthwise convolution
    const float* x; const float* w; const float* bias; float* y;
    int N, C, H, W, OH, OW, cout, KH, KW, SH, SW, PH, PW, DH, DW, group;
    int out_img, out_c0, act;
};
struct F32PoolArgs {
    const float* x; float* y;
    int N, C, H, W, OH, OW, KH, KW, SH, SW, PH, PW, method, caffe_flavor;
};
struct F32MapArgs {            // relu / leaky / relu6, concat slice copy, nearest upsample
    const float* x; float* y;
    int N, C, H, W;            // INPUT geometry
    int out_img, out_c0, scale;
    float slope;
};
hipError_t launch_conv_f32_direct(const F32DirectArgs& a, hipStream_t s);
hipError_t launch_pool_f32(const F32PoolArgs& a, hipStream_t s);
hipError_t launch_map_f32(const F32MapArgs& a, int mode, hipStream_t s);   // 0 relu/leaky, 1 concat copy, 2 upsample, 3 relu6
hipError_t launch_eltwise_f32(const float* a, const float* b, float* y, size_t count, int type, hipStream_t s);
hipError_t launch_softmax_f32(const float* x, float* y, int N, int C, int inner, hipStream_t s);
hipError_t launch_fc_u8(const U8FcArgs& a, hipStream_t s);
hipError_t launch_pool_u8(const U8PoolArgs& a, hipStream_t s);
hipError_t launch_relu_u8(const U8MapArgs& a, hipStream_t s);
hipError_t launch_requant_copy_u8(const U8MapArgs& a, hipStream_t s);
hipError_t launch_upsample_u8(const U8MapArgs& a, hipStream_t s);
hipError_t launch_flatcat_u8(const U8CatArgs& a, hipStream_t s);
hipError_t launch_eltwise_u8(const U8EltArgs& a, hipStream_t s);
hipError_t launch_softmax_u8(const U8SoftmaxArgs& a, hipStream_t s);

}  // namespace tamd

// every launch of the backend goes through the launch recorder (launch_rec.h): a plain launch, plus -- while prerun records
// the launch list for direct dispatch -- a copy of what was launched
#include "launch_rec.h"
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) ::tamd::launch_rec(kernel, grid, block, shmem, stream, __VA_ARGS__)
