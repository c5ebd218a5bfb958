// Device-kernel argument blocks and host launchers (private to the backend).
// Device layout: activations are NHWC int8 with a channel stride `cs` that is a multiple of 16 bytes
// (so every pixel row is 16-B aligned and K-contiguous for MFMA operand loads); conv weights are
// repacked once at prerun to [cout_pad][kpad] with k = (ky*KW+kx)*cin_pad + ci, zero padded.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tamd {

struct ConvArgs {
    const int8_t* x;       // NHWC input, channel stride cs_in
    const int8_t* w;       // packed weights [cout_pad][kpad]
    const int32_t* bias;   // [cout_pad] (zeros when the node has no bias)
    const float* wscale;   // [cout_pad]
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
    float m1, lo, hi, out_scale;   // requantisation constants, see epilogue.h (wscale[] holds m2[c])
    const int8_t* zeros;   // >= 16 zero bytes (source of out-of-image taps for the LDS-DMA kernel)
    int dbg;               // perf experiments only (TAMD_IGEMM2_DBG), 0 in production
};

struct DwArgs {
    const int8_t* x;       // NHWC
    const int8_t* w;       // [3 rows][cw] dwords {w[r][0],w[r][1],w[r][2],0}, cw = roundup(C,16)
    const int32_t* bias;   // [cw]
    const float* wscale;   // [cw]
    int8_t* y;
    int N, H, W, C, cs_in, cw, OH, OW, ldc, c_off;
    int S, PH, PW;
    float m1, lo, hi, out_scale;   // requantisation constants, see epilogue.h (wscale[] holds m2[c])
};

struct DirectArgs {        // generic direct conv (any group / cin), also NCHW-input first layers
    const int8_t* x;
    const int8_t* w;       // OIHW as in the model
    const int32_t* bias;   // may be null
    const float* wscale;
    int8_t* y;             // NHWC
    int N, C, H, W, cs_in; // cs_in == 0 => x is NCHW (graph input), else NHWC with that stride
    int OH, OW, cout, ldc, c_off;
    int KH, KW, SH, SW, PH, PW, DH, DW, group;
    float m1, lo, hi, out_scale;   // requantisation constants, see epilogue.h (wscale[] holds m2[c])
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
    float m1, lo, hi, out_scale;   // requantisation constants, see epilogue.h (wscale[] holds m2[c])
};

struct PoolArgs {
    const int8_t* x; int8_t* y;
    int N, H, W, C, cs_in, OH, OW, ldc, c_off;
    int KH, KW, SH, SW, PH, PW;
    int method, caffe_flavor;
    float in_scale, out_scale;
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

struct LayoutArgs {        // NCHW <-> NHWC(cs) int8 / generic element size
    const void* src; void* dst; int N, C, H, W, cs; int elem;
};

// launchers (return hipError_t of the launch)
hipError_t launch_conv_igemm(const ConvArgs& a, hipStream_t s);
const char* conv_igemm_kernel_name(const ConvArgs& a);   // tile shape the launcher will pick
hipError_t launch_gemm_direct(const ConvArgs& a, hipStream_t s);   // 1x1, small-M / latency-bound shapes
bool gemm_direct_applicable(const ConvArgs& a);
hipError_t launch_conv_igemm2(const ConvArgs& a, hipStream_t s);  // LDS-DMA 3-stage ring, large problems
bool conv_igemm2_applicable(const ConvArgs& a);
const char* conv_igemm2_kernel_name(const ConvArgs& a);
hipError_t launch_pw_stream(const ConvArgs& a, hipStream_t s);     // 1x1, shallow K, many pixels
bool pw_stream_applicable(const ConvArgs& a);
hipError_t launch_conv_first(const FirstArgs& a, hipStream_t s);
hipError_t launch_dwconv3x3(const DwArgs& a, hipStream_t s);
hipError_t launch_conv_direct(const DirectArgs& a, hipStream_t s);
hipError_t launch_pool(const PoolArgs& a, hipStream_t s);
hipError_t launch_eltwise(const EltArgs& a, hipStream_t s);
hipError_t launch_relu(const ReluArgs& a, hipStream_t s);
hipError_t launch_nchw_to_nhwc(const LayoutArgs& a, hipStream_t s);
hipError_t launch_nhwc_to_nchw(const LayoutArgs& a, hipStream_t s);

}  // namespace tamd
