/*
 * TEST INFRASTRUCTURE ONLY.  CPU restatement ("oracle") of the reference's arithmetic for the
 * int8 / fp32 Conv2D + FC hot path and the glue ops of the config graphs.  Plain C, NCHW buffers,
 * no dependency on the reference tree.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may call into this file; the product (tengine_amd/) never links it.
 *
 * Pinning: tests/test_oracle_vs_reference.py runs every function here side by side with the real
 * reference library (oracle/_ref, built from the unmodified sources by oracle/build_ref.py) on
 * seeded random layers and whole models -- bit-exact for int8 -- and tests/test_golden_kat.py
 * checks it against the inline known-answer vectors of the reference's own device tests
 * (tests/op/test_opendla_op_convolution.cpp etc., restated in tests/golden/).
 *
 * Each function cites the reference file:line it follows (paths relative to the reference root,
 * source/device/cpu/op/...).  All float arithmetic is binary32, one rounding per written
 * operation, never fused: build with -ffp-contract=off (oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* which formula the reference's kernel selection (score()) lands on -- SURVEY §8 a1:
 *   conv_hcl_x86.c:351-371      group==1, dtype in {fp32,u8,i8}            -> ORC_CONV_HCL (A1)
 *   conv_dw_hcl_x86.c:508-543   dw 3x3 s1|s2, batch 1, sym pad, dil 1       -> ORC_CONV_HCL (A1, same epilogue text)
 *   conv_ref.c:197-200          everything else                             -> ORC_CONV_REF (A2)           */
enum { ORC_CONV_HCL = 0, ORC_CONV_REF = 1 };

ORC_API int orc_conv_int8_variant(int batch, int group, int cin_g, int cout_g, int kh, int kw, int sh, int sw,
                                  int ph0, int ph1, int pw0, int pw1, int dh, int dw)
{
    if (group == 1)
        return ORC_CONV_HCL;
    if (kh == kw && batch == 1 && group > 1 && cin_g == 1 && cout_g == 1 && ph0 == ph1 && pw0 == pw1 && dh == 1
        && dw == 1 && kh == 3 && ((sh == 1 && sw == 1) || (sh == 2 && sw == 2)))
        return ORC_CONV_HCL;
    return ORC_CONV_REF;
}

static inline int8_t sat_i8(int v)
{
    if (v > 127) v = 127;
    if (v < -127) v = -127;
    return (int8_t)v;
}

/* Requantising epilogue.
 * A1  conv/x86/conv_kernel_x86.c:1826-1889 == conv/x86/conv_dw_hcl_x86.c:197-261,373-436
 *     f = (float)(acc+bias) * in_scale * w_scale[c]   (two multiplies, in that order)
 *     act==0 -> relu ; act>0 -> clamp [0,6] (ANY positive code) ; q = round(f / out_scale) ; clamp +-127
 * A2  conv/conv_kernel_ref_int8.c:72-78,137-167
 *     f = (float)(acc+bias) * (in_scale*w_scale[c])   (pre-multiplied) ; act>=0: relu unless act==1,
 *     act==1 -> clamp[-1,1], act==6 -> min 6 ; q = round(f/out_scale) ; clamp +-127                      */
static inline int8_t requant_conv(int32_t acc, float in_scale, float w_scale, float out_scale, int act, int variant)
{
    float f;
    if (variant == ORC_CONV_HCL)
    {
        f = (float)acc * in_scale;
        f = f * w_scale;
        if (act == 0 && f < 0) f = 0;
        if (act > 0)
        {
            if (f < 0) f = 0;
            if (f > 6) f = 6;
        }
    }
    else
    {
        float d = in_scale * w_scale;
        f = (float)acc * d;
        if (act >= 0)
        {
            if (f < 0 && act != 1) f = 0;
            if (f > 1 && act == 1) f = 1;
            if (f > 6 && act == 6) f = 6;
            if (f < -1 && act == 1) f = -1;
        }
    }
    float q = f / out_scale;
    return sat_i8((int)round((double)q));
}

/* int8 convolution, NCHW input [n][cin][h][w], OIHW weight [cout][cin/group][kh][kw], int32 bias.
 * Integer part: conv_kernel_x86.c:187-242 (im2col: top/left pad = pad_h0/pad_w0, out-of-image taps
 * contribute 0) + :1008-1630 (sgemm_i8, exact int32) ; dw: conv_dw_hcl_x86.c:97-269 ; naive:
 * conv_kernel_ref_int8.c:86-136.  All three compute the same exact integer sum.                        */
ORC_API int orc_conv2d_int8(const int8_t* x, const int8_t* w, const int32_t* bias, int8_t* y, int n, int cin, int h,
                            int wd, int cout, int oh, int ow, int kh, int kw, int sh, int sw, int ph0, int pw0,
                            int dh, int dw, int group, int act, float in_scale, const float* w_scales,
                            float out_scale, int variant)
{
    int cin_g = cin / group, cout_g = cout / group;
#pragma omp parallel for collapse(2)
    for (int b = 0; b < n; b++)
        for (int oc = 0; oc < cout; oc++)
        {
            int g = oc / cout_g;
            const int8_t* wk = w + (size_t)oc * cin_g * kh * kw;
            for (int oy = 0; oy < oh; oy++)
                for (int ox = 0; ox < ow; ox++)
                {
                    int32_t acc = 0;
                    for (int kc = 0; kc < cin_g; kc++)
                    {
                        const int8_t* xc = x + ((size_t)b * cin + (size_t)g * cin_g + kc) * h * wd;
                        for (int ky = 0; ky < kh; ky++)
                        {
                            int iy = oy * sh - ph0 + ky * dh;
                            if (iy < 0 || iy >= h) continue;
                            for (int kx = 0; kx < kw; kx++)
                            {
                                int ix = ox * sw - pw0 + kx * dw;
                                if (ix < 0 || ix >= wd) continue;
                                acc += (int32_t)xc[iy * wd + ix] * (int32_t)wk[(kc * kh + ky) * kw + kx];
                            }
                        }
                    }
                    if (bias) acc += bias[oc];
                    y[(((size_t)b * cout + oc) * oh + oy) * ow + ox] =
                        requant_conv(acc, in_scale, w_scales[oc], out_scale, act, variant);
                }
        }
    return 0;
}

/* int8 fully connected -- fc/fc_ref.c:209-297:  r[o] = (in_scale*w_scale[o]) / out_scale ;
 * y = sat(roundf((float)(acc+bias[o]) * r[o]), +-127).  weight [out][hidden] (need_trans==0).           */
ORC_API int orc_fc_int8(const int8_t* x, const int8_t* w, const int32_t* bias, int8_t* y, int batch, int hidden,
                        int nout, float in_scale, const float* w_scales, float out_scale)
{
#pragma omp parallel for
    for (int o = 0; o < nout; o++)
    {
        float r = (in_scale * w_scales[o]) / out_scale;
        for (int b = 0; b < batch; b++)
        {
            int32_t acc = bias ? bias[o] : 0;
            for (int j = 0; j < hidden; j++) acc += (int32_t)x[(size_t)b * hidden + j] * (int32_t)w[(size_t)o * hidden + j];
            float f = (float)acc * r;
            y[(size_t)b * nout + o] = sat_i8((int)roundf(f));
        }
    }
    return 0;
}

/* int8 pooling, NCHW -- pooling/pooling_kernel_ref_int8.c:84-189.
 * max: y = round((float)max_q * (in_scale/out_scale)) ; avg: f=(float)sum*in_scale; f=f/(float)pool_size;
 * y = round(f/out_scale) ; pool_size counts in-image taps unless caffe_flavor (then the padded window
 * clipped to in+pad).  method 0 max, 1 avg (pooling_param.h:28-32).                                       */
ORC_API int orc_pool_int8(const int8_t* x, int8_t* y, int n, int c, int h, int w, int oh, int ow, int kh, int kw,
                          int sh, int sw, int ph0, int pw0, int method, int caffe_flavor, float in_scale,
                          float out_scale)
{
    float requant = in_scale / out_scale;
    for (int b = 0; b < n; b++)
        for (int ch = 0; ch < c; ch++)
        {
            const int8_t* xc = x + ((size_t)b * c + ch) * h * w;
            for (int py = 0; py < oh; py++)
                for (int px = 0; px < ow; px++)
                {
                    int hs = py * sh - ph0, he = hs + kh;
                    if (he > h + ph0) he = h + ph0;
                    int ws = px * sw - pw0, we = ws + kw;
                    if (we > w + pw0) we = w + pw0;
                    int pool_size = 1;
                    if (caffe_flavor) pool_size = (he - hs) * (we - ws);
                    if (hs < 0) hs = 0;
                    if (ws < 0) ws = 0;
                    if (he > h) he = h;
                    if (we > w) we = w;
                    if (!caffe_flavor) pool_size = (he - hs) * (we - ws);
                    int8_t* out = y + (((size_t)b * c + ch) * oh + py) * ow + px;
                    if (method == 0)
                    {
                        int8_t m = xc[hs * w + ws];
                        for (int iy = hs; iy < he; iy++)
                            for (int ix = ws; ix < we; ix++)
                                if (xc[iy * w + ix] > m) m = xc[iy * w + ix];
                        *out = sat_i8((int)round((double)((float)m * requant)));
                    }
                    else
                    {
                        int32_t s = 0;
                        for (int iy = hs; iy < he; iy++)
                            for (int ix = ws; ix < we; ix++) s += xc[iy * w + ix];
                        float f = (float)s * in_scale;
                        f = f / (float)pool_size;
                        *out = sat_i8((int)round((double)(f / out_scale)));
                    }
                }
        }
    return 0;
}

/* int8 relu / leaky relu -- relu/relu_kernel_ref_int8.c:40-94:
 * f=(float)q*in_scale ; f<0 ? f*slope(or 0) : f ; y = round(f/out_scale) clamp +-127                    */
ORC_API int orc_relu_int8(const int8_t* x, int8_t* y, size_t count, float slope, float in_scale, float out_scale)
{
    for (size_t i = 0; i < count; i++)
    {
        float f = (float)x[i] * in_scale;
        if (f < 0) f = (slope == 0) ? 0 : f * slope;
        y[i] = sat_i8((int)round((double)(f / out_scale)));
    }
    return 0;
}

/* concat int8, one input slice -- concat/concat_kernel_ref_int8.c:60-95 (same text for every rank / axis):
 * rescale = in_scale / out_scale ; q = roundf((float)x * rescale) ; q > 127 -> 127 ; and q < -127 -> **+127**: the
 * reference's lower clamp assigns the wrong sign at all ten sites (:83-84, :113-114, ... :430-431).  "Identical to
 * the reference" includes that defect, so it is restated, not repaired.                                          */
ORC_API int orc_requant_copy_int8(const int8_t* x, int8_t* y, size_t count, float in_scale, float out_scale)
{
    float rescale = in_scale / out_scale;
    for (size_t i = 0; i < count; i++)
    {
        int q = (int)roundf((float)x[i] * rescale);
        if (q > 127) q = 127;
        else if (q < -127) q = 127;
        y[i] = (int8_t)q;
    }
    return 0;
}

/* int8 eltwise, same-shape operands -- eltwise/eltwise_ref.c:589-640,833-837:
 * a=(float)qa*sa ; b=(float)qb*sb ; f = a (+|*|max|-) b ; y = round(f/out_scale) clamp +-127
 * type codes: eltwise_param.h (0 PROD, 2 SUM, 4 SUB, 6 MAX)                                              */
ORC_API int orc_eltwise_int8(const int8_t* a, const int8_t* b, int8_t* y, size_t count, int type, float sa,
                             float sb, float out_scale)
{
    for (size_t i = 0; i < count; i++)
    {
        float fa = (float)a[i] * sa, fb = (float)b[i] * sb, f;
        switch (type)
        {
        case 0: f = fa * fb; break;
        case 2: f = fa + fb; break;
        case 4: f = fa - fb; break;
        case 6: f = fa > fb ? fa : fb; break;
        default: return -1;
        }
        y[i] = sat_i8((int)round((double)(f / out_scale)));
    }
    return 0;
}

/* int8 softmax -- softmax/softmax_kernel_ref_int8.c:41-117 with softmax_kernel_ref.h:35-85: f = (float)q * in_scale
 * (:86-89); per (outer, inner) position: max over the axis (softmax_kernel_ref.h:35-50), out = (float)exp((double)(f - max))
 * -- C `exp`, i.e. the DOUBLE routine, rounded to float on the store (:67) --, sum accumulated in fp32 in axis order
 * (:68), out / sum (:72-79); y = round(out / out_scale) clamp +-127 (:101-111).  Zero points are not read.      */
ORC_API int orc_softmax_int8(const int8_t* x, int8_t* y, int out_size, int on_size, int in_size, float in_scale,
                             float out_scale)
{
    float* f = (float*)malloc(sizeof(float) * (size_t)on_size * in_size);
    float* o = (float*)malloc(sizeof(float) * (size_t)on_size * in_size);
    float* mx = (float*)malloc(sizeof(float) * (size_t)in_size);
    float* sm = (float*)malloc(sizeof(float) * (size_t)in_size);
    for (int i = 0; i < out_size; i++)
    {
        const int8_t* xi = x + (size_t)i * on_size * in_size;
        int8_t* yi = y + (size_t)i * on_size * in_size;
        for (int j = 0; j < on_size * in_size; j++) f[j] = (float)xi[j] * in_scale;
        for (int l = 0; l < in_size; l++) mx[l] = f[l];
        for (int j = 0; j < on_size; j++)
            for (int l = 0; l < in_size; l++)
                if (mx[l] < f[j * in_size + l]) mx[l] = f[j * in_size + l];
        for (int l = 0; l < in_size; l++) sm[l] = 0.f;
        for (int j = 0; j < on_size; j++)
            for (int l = 0; l < in_size; l++)
            {
                o[j * in_size + l] = (float)exp((double)(f[j * in_size + l] - mx[l]));
                sm[l] = sm[l] + o[j * in_size + l];
            }
        for (int j = 0; j < on_size * in_size; j++)
        {
            float v = o[j] / sm[j % in_size];
            yi[j] = sat_i8((int)round((double)(v / out_scale)));
        }
    }
    free(f); free(o); free(mx); free(sm);
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * fp32 (tolerance 1e-4, order-free): conv = im2col+sgemm+bias+relu/relu6 (conv_kernel_x86.c:1632-1701),
 * fc (fc_ref.c:52-83), pooling (pooling_kernel_ref_fp32.c).  Accumulated in double here so the oracle is
 * the *more* accurate side of the 1e-4 comparison.
 * ---------------------------------------------------------------------------------------------- */
ORC_API int orc_conv2d_fp32(const float* x, const float* w, const float* bias, float* y, int n, int cin, int h, int wd,
                            int cout, int oh, int ow, int kh, int kw, int sh, int sw, int ph0, int pw0, int dh, int dw,
                            int group, int act)
{
    int cin_g = cin / group, cout_g = cout / group;
#pragma omp parallel for collapse(2)
    for (int b = 0; b < n; b++)
        for (int oc = 0; oc < cout; oc++)
        {
            int g = oc / cout_g;
            const float* wk = w + (size_t)oc * cin_g * kh * kw;
            for (int oy = 0; oy < oh; oy++)
                for (int ox = 0; ox < ow; ox++)
                {
                    double acc = 0;
                    for (int kc = 0; kc < cin_g; kc++)
                    {
                        const float* xc = x + ((size_t)b * cin + (size_t)g * cin_g + kc) * h * wd;
                        for (int ky = 0; ky < kh; ky++)
                        {
                            int iy = oy * sh - ph0 + ky * dh;
                            if (iy < 0 || iy >= h) continue;
                            for (int kx = 0; kx < kw; kx++)
                            {
                                int ix = ox * sw - pw0 + kx * dw;
                                if (ix < 0 || ix >= wd) continue;
                                acc += (double)xc[iy * wd + ix] * (double)wk[(kc * kh + ky) * kw + kx];
                            }
                        }
                    }
                    if (bias) acc += bias[oc];
                    float f = (float)acc;
                    if (act == 0 && f < 0) f = 0;
                    if (act > 0)
                    {
                        if (f < 0) f = 0;
                        if (f > 6) f = 6;
                    }
                    y[(((size_t)b * cout + oc) * oh + oy) * ow + ox] = f;
                }
        }
    return 0;
}

ORC_API int orc_fc_fp32(const float* x, const float* w, const float* bias, float* y, int batch, int hidden, int nout)
{
#pragma omp parallel for
    for (int o = 0; o < nout; o++)
        for (int b = 0; b < batch; b++)
        {
            double acc = bias ? bias[o] : 0;
            for (int j = 0; j < hidden; j++) acc += (double)x[(size_t)b * hidden + j] * (double)w[(size_t)o * hidden + j];
            y[(size_t)b * nout + o] = (float)acc;
        }
    return 0;
}

ORC_API int orc_pool_fp32(const float* x, float* y, int n, int c, int h, int w, int oh, int ow, int kh, int kw, int sh,
                          int sw, int ph0, int pw0, int method, int caffe_flavor)
{
    for (int b = 0; b < n; b++)
        for (int ch = 0; ch < c; ch++)
        {
            const float* xc = x + ((size_t)b * c + ch) * h * w;
            for (int py = 0; py < oh; py++)
                for (int px = 0; px < ow; px++)
                {
                    int hs = py * sh - ph0, he = hs + kh;
                    if (he > h + ph0) he = h + ph0;
                    int ws = px * sw - pw0, we = ws + kw;
                    if (we > w + pw0) we = w + pw0;
                    int pool_size = 1;
                    if (caffe_flavor) pool_size = (he - hs) * (we - ws);
                    if (hs < 0) hs = 0;
                    if (ws < 0) ws = 0;
                    if (he > h) he = h;
                    if (we > w) we = w;
                    if (!caffe_flavor) pool_size = (he - hs) * (we - ws);
                    float* out = y + (((size_t)b * c + ch) * oh + py) * ow + px;
                    if (method == 0)
                    {
                        float m = xc[hs * w + ws];
                        for (int iy = hs; iy < he; iy++)
                            for (int ix = ws; ix < we; ix++)
                                if (xc[iy * w + ix] > m) m = xc[iy * w + ix];
                        *out = m;
                    }
                    else
                    {
                        double s = 0;
                        for (int iy = hs; iy < he; iy++)
                            for (int ix = ws; ix < we; ix++) s += xc[iy * w + ix];
                        *out = (float)(s / pool_size);
                    }
                }
        }
    return 0;
}

/* ================================================================================================
 * uint8 (per-tensor asymmetric) -- the reference SIMULATES uint8 in fp32.
 * The reference is built -O3 -std=gnu99 -mfma, i.e. with GCC's default -ffp-contract=fast: every
 * `a*b + c` in its C source is ONE fused multiply-add.  This file is built -ffp-contract=off, so each fusion
 * the reference gets is written out as fmaf() here and nothing else is fused.
 * ============================================================================================== */
static inline uint8_t sat_u8(int v)
{
    if (v > 255) v = 255;
    if (v < 0) v = 0;
    return (uint8_t)v;
}

/* The fp32 GEMM the reference simulates uint8 convolution with: conv/x86/conv_kernel_x86.c:322-960 (sgemm_fp,
 * __AVX__ branch -- oracle/build_ref.py compiles the reference with -mfma, which implies AVX, like the
 * reference's own x86 build).  Per output element (row m of M = cout, column j of N = out_h*out_w of ONE image)
 * the summation ORDER depends on where the element sits in the 8x8 register tiling; restated exactly:
 *   j <  N&~7                     : one fused chain  s = fma(col[k], w[k], s), k = 0..K-1          (:349-520,651-762,
 *                                                                                                    :838-877)
 *   j >= N&~7, m in an 8- or 4-row block : four interleaved fused chains s_r over k = r (mod 4), k < K&~3, then
 *                                   s = ((0 + (s0+s1)) + (s2+s3)), then the fused chain over the K%4 tail (:531-583,
 *                                   :769-811)
 *   j >= N&~7, m in the last M%4 rows    : the same four lane chains (mul+add, contracted to fma by the compiler),
 *                                   s = ((s0+s1)+s2)+s3, then the fused tail chain (:905-935)
 * col[] is the im2col column in (c, ky, kx) order with 0.0f at out-of-image taps (:126-185).                 */
static float sgemm_fp_element(const float* col, const float* w, int K, int full_col, int blocked_row)
{
    float s = 0.f;
    int k = 0;
    if (full_col)
    {
        for (; k < K; k++) s = fmaf(col[k], w[k], s);
        return s;
    }
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (; k + 3 < K; k += 4)
    {
        s0 = fmaf(w[k], col[k], s0);
        s1 = fmaf(w[k + 1], col[k + 1], s1);
        s2 = fmaf(w[k + 2], col[k + 2], s2);
        s3 = fmaf(w[k + 3], col[k + 3], s3);
    }
    if (blocked_row)
    {
        s0 = s0 + s1;
        s2 = s2 + s3;
        s = s + s0;
        s = s + s2;
    }
    else
    {
        s = s0 + s1;
        s = s + s2;
        s = s + s3;
    }
    for (; k < K; k++) s = fmaf(w[k], col[k], s);
    return s;
}

/* uint8 convolution.
 * variant ORC_CONV_REF -- conv/conv_kernel_ref_uint8.c:42-195, bit-exact restatement: operands dequantised to
 *   fp32 (:74-95), ONE sequential fused chain total = fma(x, w, total) in kc->kh->kw order over in-image taps
 *   (:125-152), + bias_fp32 = (float)b*in_s*k_s (:88-95,155), activation as the naive ref (:157-175),
 *   round(total/out_s) + out_zp, clamp [0,255] (:177-182).  Used by the reference for every depthwise /
 *   grouped uint8 conv (conv_dw_hcl_x86.c:533 rejects uint8).
 * variant ORC_CONV_HCL -- conv/x86/conv_kernel_x86.c:68-80 (weights -> fp32), :126-185 (im2col_uint8 -> fp32),
 *   sgemm_fp in its exact summation order (sgemm_fp_element above), then :1703-1794: s = s + (float)bias *
 *   (in_s*k_s) (product rounded first), relu / relu6 for ANY positive activation code, (int)(round(s/out_s) + out_zp), clamp [0,255]. */
ORC_API int orc_conv2d_uint8(const uint8_t* x, const uint8_t* w, const int32_t* bias, uint8_t* y, int n, int cin, int h,
                             int wd, int cout, int oh, int ow, int kh, int kw, int sh, int sw, int ph0, int pw0, int dh,
                             int dw, int group, int act, float in_scale, int in_zp, float w_scale, int w_zp,
                             float out_scale, int out_zp, int variant)
{
    int cin_g = cin / group, cout_g = cout / group;
    int K = cin_g * kh * kw, N = oh * ow;
    float* wf = (float*)malloc(sizeof(float) * (size_t)cout * K);
    for (size_t i = 0; i < (size_t)cout * K; i++) wf[i] = ((float)w[i] - (float)w_zp) * w_scale;
    int m_blocked = (cout_g >> 3 << 3) + (((cout_g - (cout_g >> 3 << 3)) >> 2) << 2);
#pragma omp parallel
    {
        float* col = (float*)malloc(sizeof(float) * (size_t)K);
#pragma omp for collapse(2)
        for (int b = 0; b < n; b++)
            for (int j = 0; j < N; j++)
            {
                int oy = j / ow, ox = j % ow;
                for (int g = 0; g < group; g++)
                {
                    int kk = 0;
                    for (int kc = 0; kc < cin_g; kc++)
                    {
                        const uint8_t* xc = x + ((size_t)b * cin + (size_t)g * cin_g + kc) * h * wd;
                        for (int ky = 0; ky < kh; ky++)
                            for (int kx = 0; kx < kw; kx++, kk++)
                            {
                                int iy = oy * sh - ph0 + ky * dh, ix = ox * sw - pw0 + kx * dw;
                                col[kk] = (iy < 0 || iy >= h || ix < 0 || ix >= wd)
                                              ? 0.f
                                              : ((float)xc[iy * wd + ix] - (float)in_zp) * in_scale;
                            }
                    }
                    for (int m = 0; m < cout_g; m++)
                    {
                        int oc = g * cout_g + m;
                        const float* wk = wf + (size_t)oc * K;
                        float total;
                        if (variant == ORC_CONV_REF)
                        {
                            total = 0.f;
                            /* out-of-image taps are skipped there; fma(0, w, t) == t, so keeping them is identical */
                            for (int k = 0; k < K; k++) total = fmaf(col[k], wk[k], total);
                            if (bias)
                            {
                                float bf = (float)bias[oc] * in_scale;
                                bf = bf * w_scale;
                                total = total + bf;
                            }
                            if (act >= 0)
                            {
                                if (total < 0 && act != 1) total = 0;
                                if (total > 1 && act == 1) total = 1;
                                if (total > 6 && act == 6) total = 6;
                                if (total < -1 && act == 1) total = -1;
                            }
                        }
                        else
                        {
                            total = sgemm_fp_element(col, wk, K, j < (N & ~7), m < m_blocked);
                            /* :1733-1743 -- the loop-invariant product (float)bias * bias_scale is hoisted out of the
                             * pixel loop by the compiler and ADDED (vmulss once, then vaddps): two roundings, not an fma.
                             * Read off the reference object's disassembly; an fma here differs once per ~2e5 outputs. */
                            if (bias) total = total + (float)bias[oc] * (in_scale * w_scale);
                            if (act == 0 && total < 0) total = 0;
                            if (act > 0)
                            {
                                if (total < 0) total = 0;
                                if (total > 6) total = 6;
                            }
                        }
                        int out = (int)(round((double)(total / out_scale)) + out_zp);
                        y[(((size_t)b * cout + oc) * oh + oy) * ow + ox] = sat_u8(out);
                    }
                }
            }
        free(col);
    }
    free(wf);
    return 0;
}

/* uint8 softmax -- softmax/softmax_kernel_ref_uint8.c:40-119 with softmax_kernel_ref.h:35-85: dequantise to fp32,
 * per (outer, inner) position: max over the axis (:35-50), out = (float)exp((double)(x - max)) -- C `exp`, i.e. the
 * DOUBLE routine, rounded to float on the store (:67) --, sum accumulated in fp32 in axis order (:68), out / sum
 * (:72-79), then (int)(round(out / out_scale) + out_zp), clamp [0,255] (:103-113).                              */
ORC_API int orc_softmax_uint8(const uint8_t* x, uint8_t* y, int out_size, int on_size, int in_size, float in_scale,
                              int in_zp, float out_scale, int out_zp)
{
    float* f = (float*)malloc(sizeof(float) * (size_t)on_size * in_size);
    float* o = (float*)malloc(sizeof(float) * (size_t)on_size * in_size);
    float* mx = (float*)malloc(sizeof(float) * (size_t)in_size);
    float* sm = (float*)malloc(sizeof(float) * (size_t)in_size);
    for (int i = 0; i < out_size; i++)
    {
        const uint8_t* xi = x + (size_t)i * on_size * in_size;
        uint8_t* yi = y + (size_t)i * on_size * in_size;
        for (int j = 0; j < on_size * in_size; j++) f[j] = ((float)xi[j] - (float)(uint8_t)in_zp) * in_scale;
        for (int l = 0; l < in_size; l++) mx[l] = f[l];
        for (int j = 0; j < on_size; j++)
            for (int l = 0; l < in_size; l++)
                if (mx[l] < f[j * in_size + l]) mx[l] = f[j * in_size + l];
        for (int l = 0; l < in_size; l++) sm[l] = 0.f;
        for (int j = 0; j < on_size; j++)
            for (int l = 0; l < in_size; l++)
            {
                o[j * in_size + l] = (float)exp((double)(f[j * in_size + l] - mx[l]));
                sm[l] = sm[l] + o[j * in_size + l];
            }
        for (int j = 0; j < on_size * in_size; j++)
        {
            float v = o[j] / sm[j % in_size];
            int u = (int)(round((double)(v / out_scale)) + (double)(uint8_t)out_zp);
            yi[j] = sat_u8(u);
        }
    }
    free(f); free(o); free(mx); free(sm);
    return 0;
}

/* PriorBox -- priorbox/priorbox_ref.c:53-199.  Depends on shapes and parameters only.  C arithmetic kept as written
 * there: `int min_size_ = param->min_size[s]` truncates (:110); the square prior of (min, max) is the DOUBLE sqrt of the
 * INT product (:123-126); the aspect-ratio priors multiply / divide that int by the DOUBLE sqrt of the float ratio
 * (:136-139); every corner is (centre -+ size * 0.5f) / image extent in float; the flipped prior divides x by image_h and
 * y by image_w (:146-150, as written there); clip (:157-164) covers the boxes, then out_dim/4 copies of variance[4]
 * (:166-175).  Output: [1][2][out_dim][1] floats, out_dim = feat_h * feat_w * num_priors * 4 (priorbox.c:37-75).        */
ORC_API int orc_priorbox_f32(float* out, int feat_h, int feat_w, int data_h, int data_w, int image_h_p, int image_w_p,
                             float step_h_p, float step_w_p, float offset, const float* min_size, int n_min,
                             const float* max_size, int n_max, const float* ratio, int n_ratio, const float* variance,
                             int flip, int clip)
{
    if (n_max > 0 && n_max != n_min) return -1;                            /* priorbox.c:48-61 */
    const int num_priors = (n_ratio * (flip ? 2 : 1) + 1 + (n_max > 0 ? 1 : 0)) * n_min;
    const int dim = feat_h * feat_w * num_priors * 4;
    int image_w, image_h;
    if (image_h_p == 0 || image_w_p == 0) { image_w = data_w; image_h = data_h; }
    else { image_w = image_w_p; image_h = image_h_p; }
    float step_w, step_h;
    if (step_h_p == 0 || step_w_p == 0) { step_w = (float)(image_w) / feat_w; step_h = (float)(image_h) / feat_h; }
    else { step_w = step_w_p; step_h = step_h_p; }
    for (int h = 0; h < feat_h; ++h)
    {
        float* box = out + h * num_priors * 4 * feat_w;
        for (int w = 0; w < feat_w; ++w)
        {
            float center_x = (w + offset) * step_w;
            float center_y = (h + offset) * step_h;
            float bw, bh;
            for (int s = 0; s < n_min; ++s)
            {
                int mn = min_size[s];
                bw = bh = mn;
                box[0] = (center_x - bw * 0.5f) / image_w; box[1] = (center_y - bh * 0.5f) / image_h;
                box[2] = (center_x + bw * 0.5f) / image_w; box[3] = (center_y + bh * 0.5f) / image_h;
                box += 4;
                if (n_max > 0)
                {
                    int mx = max_size[s];
                    bw = bh = sqrt(mn * mx);
                    box[0] = (center_x - bw * 0.5f) / image_w; box[1] = (center_y - bh * 0.5f) / image_h;
                    box[2] = (center_x + bw * 0.5f) / image_w; box[3] = (center_y + bh * 0.5f) / image_h;
                    box += 4;
                }
                for (int r = 0; r < n_ratio; ++r)
                {
                    float ar = ratio[r];
                    bw = mn * sqrt(ar);
                    bh = mn / sqrt(ar);
                    box[0] = (center_x - bw * 0.5f) / image_w; box[1] = (center_y - bh * 0.5f) / image_h;
                    box[2] = (center_x + bw * 0.5f) / image_w; box[3] = (center_y + bh * 0.5f) / image_h;
                    box += 4;
                    if (flip)
                    {
                        box[0] = (center_x - bh * 0.5f) / image_h; box[1] = (center_y - bw * 0.5f) / image_w;
                        box[2] = (center_x + bh * 0.5f) / image_h; box[3] = (center_y + bw * 0.5f) / image_w;
                        box += 4;
                    }
                }
            }
        }
    }
    if (clip)
        for (int d = 0; d < dim; ++d) out[d] = out[d] < 0.f ? 0.f : (out[d] > 1.f ? 1.f : out[d]);
    float* v = out + dim;
    for (int i = 0; i < dim / 4; i++, v += 4) { v[0] = variance[0]; v[1] = variance[1]; v[2] = variance[2]; v[3] = variance[3]; }
    return num_priors;
}

/* its quantisation: uint8 (int)(f / scale + zp) -- TRUNCATION, float arithmetic -- clamp [0,255] (priorbox_ref.c:178-195);
 * int8 round(f / scale) clamp [-127,127] (:197-213)                                                                    */
ORC_API int orc_priorbox_quant_uint8(const float* f, uint8_t* y, size_t n, float scale, int zp)
{
    for (size_t i = 0; i < n; i++)
    {
        int u = (int)(f[i] / scale + zp);
        y[i] = u > 255 ? 255 : (u < 0 ? 0 : u);
    }
    return 0;
}

ORC_API int orc_priorbox_quant_int8(const float* f, int8_t* y, size_t n, float scale)
{
    for (size_t i = 0; i < n; i++)
    {
        int q = round(f[i] / scale);
        y[i] = (int8_t)(q > 127 ? 127 : (q < -127 ? -127 : q));
    }
    return 0;
}

/* fc uint8 -- fc/fc_ref.c:121-207: data = (float)bias*bias_scale; data = fma(xf, wf, data) j ascending;
 * round(data/out_s) + out_zp, clamp [0,255].  bias_scale == bias_tensor->scale.                            */
ORC_API int orc_fc_uint8(const uint8_t* x, const uint8_t* w, const int32_t* bias, uint8_t* y, int batch, int hidden,
                         int nout, float in_scale, int in_zp, float w_scale, int w_zp, float bias_scale, float out_scale,
                         int out_zp)
{
#pragma omp parallel for
    for (int o = 0; o < nout; o++)
        for (int b = 0; b < batch; b++)
        {
            float data = bias ? (float)bias[o] * bias_scale : 0.f;
            for (int j = 0; j < hidden; j++)
            {
                float xf = ((float)x[(size_t)b * hidden + j] - (float)in_zp) * in_scale;
                float wf = ((float)w[(size_t)o * hidden + j] - (float)w_zp) * w_scale;
                data = fmaf(xf, wf, data);
            }
            y[(size_t)b * nout + o] = sat_u8((int)round((double)(data / out_scale)) + out_zp);
        }
    return 0;
}

/* pooling uint8 -- pooling/pooling_kernel_ref_uint8.c:91-200: dequantise all, fp32 max / sequential fp32 sum
 * (rows then columns) / pool_size, then round(f/out_s) + out_zp with ONLY the upper clamp (:193-196; a negative
 * value wraps through the uint8 store).                                                                    */
ORC_API int orc_pool_uint8(const uint8_t* x, uint8_t* y, int n, int c, int h, int w, int oh, int ow, int kh, int kw,
                           int sh, int sw, int ph0, int pw0, int method, int caffe_flavor, float in_scale, int in_zp,
                           float out_scale, int out_zp)
{
    for (int b = 0; b < n; b++)
        for (int ch = 0; ch < c; ch++)
        {
            const uint8_t* xc = x + ((size_t)b * c + ch) * h * w;
            for (int py = 0; py < oh; py++)
                for (int px = 0; px < ow; px++)
                {
                    int hs = py * sh - ph0, he = hs + kh;
                    if (he > h + ph0) he = h + ph0;
                    int ws = px * sw - pw0, we = ws + kw;
                    if (we > w + pw0) we = w + pw0;
                    int pool_size = 1;
                    if (caffe_flavor) pool_size = (he - hs) * (we - ws);
                    if (hs < 0) hs = 0;
                    if (ws < 0) ws = 0;
                    if (he > h) he = h;
                    if (we > w) we = w;
                    if (!caffe_flavor) pool_size = (he - hs) * (we - ws);
                    float f;
                    if (method == 0)
                    {
                        f = (float)((int)xc[hs * w + ws] - in_zp) * in_scale;
                        for (int iy = hs; iy < he; iy++)
                            for (int ix = ws; ix < we; ix++)
                            {
                                float v = (float)((int)xc[iy * w + ix] - in_zp) * in_scale;
                                f = f > v ? f : v;
                            }
                    }
                    else
                    {
                        float s = 0.f;
                        for (int iy = hs; iy < he; iy++)
                            for (int ix = ws; ix < we; ix++) s += (float)((int)xc[iy * w + ix] - in_zp) * in_scale;
                        f = s / pool_size;
                    }
                    int od = (int)round((double)(f / out_scale)) + out_zp;
                    y[(((size_t)b * c + ch) * oh + py) * ow + px] = (uint8_t)(od > 255 ? 255 : od);
                }
        }
    return 0;
}

/* relu / leaky uint8 -- relu/relu_kernel_ref_uint8.c:48-95: f=((float)u-zp_in)*in_s; f<0 -> f*slope (or 0);
 * y = round(f/out_s + out_zp) (zero point added INSIDE the round), clamp [0,255]                           */
ORC_API int orc_relu_uint8(const uint8_t* x, uint8_t* y, size_t count, float slope, float in_scale, int in_zp,
                           float out_scale, int out_zp)
{
    for (size_t i = 0; i < count; i++)
    {
        float f = ((float)x[i] - (float)in_zp) * in_scale;
        if (f < 0) f = (slope == 0) ? 0 : f * slope;
        y[i] = sat_u8((int)round((double)(f / out_scale + (float)out_zp)));
    }
    return 0;
}

/* concat uint8, one input slice -- concat/concat_kernel_ref_uint8.c:309-352:
 * y = roundf(fma((float)(u - zp_in), in_s/out_s, (float)out_zp)), clamp [0,255]                            */
ORC_API int orc_requant_copy_uint8(const uint8_t* x, uint8_t* y, size_t count, float in_scale, int in_zp,
                                   float out_scale, int out_zp)
{
    float rescale = in_scale / out_scale;
    for (size_t i = 0; i < count; i++)
        y[i] = sat_u8((int)roundf(fmaf((float)((int)x[i] - in_zp), rescale, (float)out_zp)));
    return 0;
}

/* nearest upsample uint8 -- upsample/upsample_ref.c:74-130: dequantise, replicate (in = out / scale),
 * y = round(f/out_s + out_zp), clamp [0,255]                                                               */
ORC_API int orc_upsample_uint8(const uint8_t* x, uint8_t* y, int n, int c, int h, int w, int scale, float in_scale,
                               int in_zp, float out_scale, int out_zp)
{
    int oh = h * scale, ow = w * scale;
    for (int b = 0; b < n * c; b++)
        for (int oy = 0; oy < oh; oy++)
            for (int ox = 0; ox < ow; ox++)
            {
                float f = ((float)x[((size_t)b * h + oy / scale) * w + ox / scale] - (float)in_zp) * in_scale;
                y[((size_t)b * oh + oy) * ow + ox] = sat_u8((int)round((double)(f / out_scale + (float)out_zp)));
            }
    return 0;
}

/* eltwise uint8 (same-shape operands) -- eltwise/eltwise_ref.c:311-585: a=(float)(u0-zp0)*s0, b likewise,
 * fp32 op, round(r/out_s) + out_zp, clamp [0,255].  type = enum (eltwise_param.h): 0 prod, 2 sum, 4 sub, 6 max.       */
ORC_API int orc_eltwise_uint8(const uint8_t* a, const uint8_t* b, uint8_t* y, size_t count, int type, float sa, int za,
                              float sb, int zb, float out_scale, int out_zp)
{
    for (size_t i = 0; i < count; i++)
    {
        float fa = (float)((int)a[i] - za) * sa, fb = (float)((int)b[i] - zb) * sb, r;
        switch (type)
        {
        case 0: r = fa * fb; break;
        case 2: r = fa + fb; break;
        case 4: r = fa - fb; break;
        case 6: r = fa > fb ? fa : fb; break;
        default: return -1;
        }
        y[i] = sat_u8((int)round((double)(r / out_scale)) + out_zp);
    }
    return 0;
}
