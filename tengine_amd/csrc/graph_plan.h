// What the two units of the int8 planner share (graph_plan.hip, graph_plan_pairs.hip); private.
#pragma once
#include "graph_internal.h"
#include "epilogue.h"

namespace tamd {

// a1: which reference formula a convolution node's result follows (conv_mode), and its requantisation folded for the device
enum { RQ_CONV_HCL = 0, RQ_CONV_REF = 1, RQ_FC = 2 };   // A1 / A2 / A5 of SURVEY Appendix A (epilogue.h)
int conv_mode(const tamd_conv_param& p, int batch, int cin, int cout);
struct RqFold { float m1, lo, hi, out_scale; std::vector<float> m2; };
RqFold fold_requant(int mode, int act, float in_s, float out_s, const HTensor& w, int cout);
int upload_rq(tamd_graph* g, const RqFold& r, int cpad, const float** wscale, RqArgs* rq);
bool exp_plain_kernels();
// plan-time timing of one candidate launch; whether the plan-time races run at all (TAMD_AUTOTUNE)
int time_fn(tamd_graph* g, const std::function<hipError_t(hipStream_t)>& fn, float* ms_out);
bool autotune_enabled();
std::vector<int8_t> pack_pw_panel(const int8_t* wd, int C, int K, int nsteps);

// the arguments of the last depthwise 3x3 / implicit-GEMM convolution planned on this thread (dwpw.hip: depthwise -> pointwise in one
// launch reads them back)
extern thread_local DwArgs g_last_dw;
extern thread_local bool g_last_dw_valid;
extern thread_local ConvArgs g_last_gemm;
extern thread_local bool g_last_gemm_valid;

// graph_plan_pairs.hip
int find_pwdw_tail(tamd_graph* g, size_t ni, int* tmode, int* prod);
int plan_pwdw(tamd_graph* g, HNode& pw, HNode& tl, int tmode, int prod, size_t s0);
int plan_dwpw(tamd_graph* g, HNode& dw, HNode& pw, size_t s0);

}  // namespace tamd
