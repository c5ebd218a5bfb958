/*
 * tengine_amd.h -- C ABI of the MI355X (gfx950) device backend for Tengine's quantised CNN hot path.
 *
 * This is the drop-in boundary.  Plain pointers and sizes only; no C++/torch types.  Every entry
 * point names the reference interface it replaces (paths relative to the reference root):
 *
 *   reference (source/device/device.h:40-108)          this library
 *   ---------------------------------------------------------------------------------------------
 *   interface.init / release_device                     tamd_init / tamd_shutdown
 *   serializer load_mem  (tm2_serializer.c:915-936)     tamd_graph_load_tm2        (same tmfile bytes)
 *   create_graph_node/tensor (c_api.c)                  tamd_graph_add_tensor / tamd_graph_add_node
 *   allocator.describe (cuda_device.cc:47-83)           tamd_op_supported
 *   interface.pre_run  (scheduler.c:49-59)              tamd_graph_prerun
 *   interface.run      (scheduler.c:134)                tamd_graph_run  (host in -> H2D -> launches -> D2H)
 *   interface.async_run / async_wait (device.h:60-63)   tamd_graph_run_async / tamd_graph_wait
 *   interface.post_run / release_graph                  tamd_graph_destroy
 *   set_tensor_buffer / get_tensor_buffer (c_api.c)     tamd_graph_set_input / tamd_graph_get_output
 *
 * The Tengine device plugin (tengine_amd/device/hip_device.cc -> register_hip_device(), the
 * symbol name cmake/registry.cmake:11-31 derives from `hip_device.cc`) is a thin translator from
 * `struct subgraph` onto this ABI; see INTEGRATION.md.
 *
 * Conventions follow the reference: every int function returns 0 on success, a negative value on
 * failure (c_api.c:463-497); tamd_last_error() gives the message.  Tensors cross the boundary in
 * the reference's layouts (activations NCHW, conv weights OIHW, FC weights [out][hidden], int32
 * bias, per-tensor activation scales, per-out-channel weight scales); the NHWC / MFMA-packed device
 * layouts are private to the backend.
 */
#ifndef TENGINE_AMD_H
#define TENGINE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TAMD_API __attribute__((visibility("default")))

/* data types == TENGINE_DT_* (source/api/c_api.h:58-63) */
enum { TAMD_DT_FP32 = 0, TAMD_DT_FP16 = 1, TAMD_DT_INT8 = 2, TAMD_DT_UINT8 = 3, TAMD_DT_INT32 = 4 };
/* tensor types == TENSOR_TYPE_* (c_api.h:70-74) */
enum { TAMD_TT_VAR = 1, TAMD_TT_CONST = 2, TAMD_TT_INPUT = 3 };

/* operator codes of this ABI (the plugin maps OP_* of source/operator/op.h:38-145 onto these;
 * the tm2 loader maps TM2_OPTYPE_* of tm2_format.h:157-264) */
enum {
    TAMD_OP_INPUT = 0,
    TAMD_OP_CONST = 1,
    TAMD_OP_CONV = 2,    /* param: tamd_conv_param  */
    TAMD_OP_FC = 3,      /* param: tamd_fc_param    */
    TAMD_OP_POOL = 4,    /* param: tamd_pool_param  */
    TAMD_OP_RELU = 5,    /* param: tamd_relu_param  */
    TAMD_OP_ELTWISE = 6, /* param: tamd_eltwise_param */
    TAMD_OP_CONCAT = 7,  /* param: tamd_concat_param */
    TAMD_OP_DROPOUT = 8, /* identity */
    TAMD_OP_UPSAMPLE = 9,/* param: tamd_upsample_param */
    TAMD_OP_RELU6 = 10,
    TAMD_OP_FLATTEN = 11,
    TAMD_OP_SOFTMAX = 12, /* param: tamd_softmax_param (NULL: axis 1); fp32 / uint8 graphs: any axis; int8 graphs: any axis >= 1 of a
                           * 2-D / 3-D / 4-D tensor, <= 16000 values along it (softmax_kernel_ref_int8.c:41-117; the spatial axes of an
                           * NHWC tensor and the axes of a dense tensor since round 6)                                               */
    TAMD_OP_PERMUTE = 13, /* param: tamd_permute_param; uint8 and (round 6) int8 graphs, order (0,2,3,1) -- the SSD head permute */
    TAMD_OP_RESHAPE = 14, /* param: tamd_reshape_param; a view of a dense tensor (uint8 / fp32 graphs; int8 since round 6: what a
                           * Permute / Flatten / Reshape / PriorBox / flat Concat produces is dense on the device)                  */
    TAMD_OP_PRIORBOX = 15,/* param: tamd_priorbox_param; uint8 / fp32 and (round 6) int8 graphs, batch 1.  Depends on shapes only:
                           * evaluated ONCE at prerun (priorbox_ref.c:53-213 re-computes the same numbers at every run), no launch   */
    TAMD_OP_NUM
};

/* field meaning == struct conv_param (source/operator/prototype/convolution_param.h:28-45);
 * activation: -1 none, 0 relu, 1 relu1, 6 relu6 */
typedef struct tamd_conv_param {
    int kernel_h, kernel_w, stride_h, stride_w;
    int pad_h0, pad_h1, pad_w0, pad_w1;
    int dilation_h, dilation_w;
    int input_channel, output_channel, group, activation;
} tamd_conv_param;

typedef struct tamd_fc_param { int num_output; } tamd_fc_param;             /* fc_param.h */

/* == struct pool_param (pooling_param.h:34-57): pool_method 0 max / 1 avg; pads are the *original*
 * (model) pads, resolved exactly like infer_shape (pooling.c:36-100) */
typedef struct tamd_pool_param {
    int pool_method, kernel_h, kernel_w, stride_h, stride_w;
    int pad_h0, pad_h1, pad_w0, pad_w1;
    int global, caffe_flavor;
} tamd_pool_param;

typedef struct tamd_relu_param { float negative_slope; } tamd_relu_param;   /* relu_param.h */
typedef struct tamd_eltwise_param { int type; int caffe_flavor; float shift, power, scale; } tamd_eltwise_param;
typedef struct tamd_concat_param { int axis; } tamd_concat_param;
typedef struct tamd_upsample_param { float scale; } tamd_upsample_param;
typedef struct tamd_permute_param { int order[4]; } tamd_permute_param;      /* permute_param.h: order0..order3 */
typedef struct tamd_softmax_param { int axis; } tamd_softmax_param;          /* softmax_param.h */
/* the RESOLVED output shape of a Reshape node (what reshape.c:37-160 infers from re_shape / is_mxnet / is_onnx); the batch
 * dimension follows tamd_graph_set_batch */
typedef struct tamd_reshape_param { int dim_num; int dims[8]; } tamd_reshape_param;

/* == struct priorbox_param (source/operator/prototype/priorbox_param.h:28-52) with the float vectors inline; num_priors /
 * out_dim are re-derived as priorbox.c:37-64 does.  inputs: [0] feature map, [1] the image ("data") -- shapes only */
#define TAMD_PRIORBOX_MAX 8
typedef struct tamd_priorbox_param {
    int min_size_num, max_size_num, aspect_ratio_num;
    float min_size[TAMD_PRIORBOX_MAX], max_size[TAMD_PRIORBOX_MAX], aspect_ratio[TAMD_PRIORBOX_MAX], variance[4];
    int flip, clip, image_h, image_w;
    float step_h, step_w, offset;
} tamd_priorbox_param;

/* == the quantisation-relevant part of struct tensor (source/graph/tensor.h:43-102) */
typedef struct tamd_tensor_desc {
    int dtype;            /* TAMD_DT_*                                            */
    int ttype;            /* TAMD_TT_*                                            */
    int dim_num;
    int dims[8];          /* NCHW / OIHW / [out][hidden] / [n]                    */
    const void* data;     /* const payload (host, copied at prerun), else NULL    */
    int quant_num;        /* 0 none, 1 per-tensor, N per-channel                  */
    const float* scales;  /* quant_num entries                                    */
    const int* zero_points;
    const char* name;     /* optional                                             */
} tamd_tensor_desc;

typedef struct tamd_node_desc {
    int op;               /* TAMD_OP_*                                            */
    int input_num;
    const int* inputs;    /* tensor indices                                       */
    int output_num;
    const int* outputs;
    const void* param;    /* op specific struct above, or NULL                    */
    const char* name;
} tamd_node_desc;

/* == the device option blob passed through set_context_device (first field is dev_name by the
 * reference's convention, trt_define.h:35-41 / scheduler.c:49-57) */
typedef struct tamd_options {
    const char* dev_name; /* "HIP"                                                */
    int size;             /* sizeof(tamd_options) as the CALLER compiled it: set_context_device copies dev_opt_size bytes
                           * and hands pre_run only the pointer (c_api.c:183-210, scheduler.c:49-57), so the blob
                           * carries its own size; fields beyond it keep their defaults (0: only dev_name is read) */
    int gpu_index;        /* HIP device ordinal                                   */
    int use_hip_graph;    /* 1: capture the launch list into a hipGraph (default) */
    int profile;          /* 1: print a per-launch timing table on stderr when the graph is released -- what
                           * TG_DEBUG_TIME=1 makes the CPU device do (cpu_define.h:41-43, cpu_dump.c:607-697); the
                           * environment variable itself is honoured too */
    int direct_dispatch;  /* 1: tamd_graph_launch replays the launch list as AQL packets on the graph's own HSA queue instead
                           * of hipGraphLaunch (shorter launch boundaries, no hole between replays; csrc/direct.cc).  The
                           * passes are ordered among themselves and observed by tamd_graph_sync / _download_outputs /
                           * _run; work on tamd_graph_stream() is NOT ordered behind them.  TAMD_DIRECT_DISPATCH=0|1
                           * overrides.  Falls back to the hipGraph silently when the list cannot be dispatched directly. */
    int keep_tensors;     /* 1: every intermediate tensor keeps a buffer of its own, so tamd_graph_read_tensor can return any of
                           * them after a run (the TG_DEBUG_DATA analogue).  0 (default): tensors whose lifetimes do not overlap
                           * share device memory -- a pass then touches a fraction of the bytes, which is what keeps the batched
                           * configs inside the device's last-level cache; read_tensor refuses such tensors.  TAMD_POOL=0|1
                           * overrides (0 = keep). */
    int u8_integer;       /* 1: uint8 convolutions (group 1) run as EXACT int32 sums on the int8 matrix cores and are requantised
                           * once -- every CONVOLUTION is within ONE quantisation step of the reference CPU backend on the reference's
                           * own inputs (teacher-forced, tests/test_gpu_u8_int.py), not byte-identical (the reference simulates uint8
                           * in fp32, conv_kernel_x86.c:68-80,1703-1794).  The bound is per layer, NOT end to end: errors compound,
                           * MobileNet-SSD's outputs differ from the reference's by up to 4-5 steps in 33-36 % of the bytes
                           * (profiles/r04_u8int_pytest.txt).  0 (default): the byte-exact fp32 chains.  TAMD_U8_INT=0|1 overrides. */
    int split_batch;      /* A batched graph whose operators treat the images of a batch independently (Convolution, FullyConnected,
                           * Pooling, ReLU, Eltwise, Dropout, Concat / Softmax on an axis >= 1) and whose activation tensors all carry
                           * the same even batch B as dimension 0 can be compiled as TWO device graphs of B / 2 images each, run side by
                           * side on their own HSA queues behind this one tamd_graph: launch boundaries and tile tails of one half
                           * overlap the other half's work (ResNet-50 b32 +8 %, MobileNet-v1 b64 +6 %, outputs identical:
                           * profiles/r06_split_batch_direct.txt).  Every entry point behaves as for one graph; host buffers are used
                           * as two contiguous halves.  0 (default): where it pays -- int8 graphs with direct_dispatch from batch 16
                           * on; 1: never (final: a caller that splits batches by itself, as the plugin does); 2: wherever it is possible.
                           * TAMD_SPLIT_BATCH=0|1|2 overrides 0 and 2 (0: never, 1: default rule, 2: wherever possible -- the values the
                           * plugin's switch of the same name takes).  tamd_graph_halves() tells which form a graph took. */
} tamd_options;

typedef struct tamd_graph tamd_graph;

/* ---- device: what struct interface.init / release_device and allocator.describe need
 * (source/device/device.h:40-84; the CUDA backend's equivalents: cuda_device.cc:47-83,212-238) ---- */
TAMD_API int tamd_device_count(void);
TAMD_API int tamd_init(int gpu_index);
TAMD_API int tamd_shutdown(void);
TAMD_API const char* tamd_last_error(void);
TAMD_API const char* tamd_version(void);
/* 1 if (op, dtype) can run on the device -- what allocator.describe publishes */
TAMD_API int tamd_op_supported(int op, int dtype);
/* 1 if this node with these parameters and tensors (descriptors only, const payloads are not read) is one the device
 * compiles; 0 -> leave it to the CPU device.  Replaces the per-op parameter tests a backend makes while it walks a
 * subgraph (source/device/cuda/cuda_executor.cc:72-134 fails Build() instead; tensorrt: trt_limit.hpp) */
TAMD_API int tamd_node_supported(const tamd_node_desc* node, const tamd_tensor_desc* inputs, int input_num,
                                 const tamd_tensor_desc* outputs, int output_num);

/* ---- graph construction: the plugin's pre_run walks `struct subgraph` (source/graph/subgraph.h, node.h:46-70,
 * tensor.h:43-102) and mirrors it through these calls, like CUDAEngine::Build does for its own IR
 * (source/device/cuda/cuda_executor.cc:72-134) ------------------------------------------------- */
TAMD_API tamd_graph* tamd_graph_create(void);
TAMD_API int tamd_graph_add_tensor(tamd_graph* g, const tamd_tensor_desc* desc);  /* -> tensor index or <0 */
TAMD_API int tamd_graph_add_node(tamd_graph* g, const tamd_node_desc* desc);      /* -> node index or <0   */
TAMD_API int tamd_graph_set_inputs(tamd_graph* g, int n, const int* tensor_ids);
TAMD_API int tamd_graph_set_outputs(tamd_graph* g, int n, const int* tensor_ids);
/* parse tmfile-v2 bytes (the reference's own model format); the buffer may be freed afterwards */
TAMD_API tamd_graph* tamd_graph_load_tm2(const void* mem, size_t size);
/* re-shape the batch dimension of every input (== set_tensor_shape + infer_shape) before prerun */
TAMD_API int tamd_graph_set_batch(tamd_graph* g, int batch);

/* ---- execution ----------------------------------------------------------------------------
 * Threading: the library may be used from several host threads, each graph by ONE thread at a time (a graph's run state --
 * I/O slots, runs in flight, its HSA queue, which is single-producer -- is not locked; the reference calls a subgraph's
 * interface from one scheduler thread as well, scheduler.c:95-213).  Different graphs never share run state.
 * Enforced since round 5: a call that arrives while ANOTHER thread is inside a call on the same graph returns -1
 * ("... one graph = one thread at a time"); calls from different threads one after the other are fine.
 * prerun  <- interface.pre_run   (device.h:46, called from scheduler.c:49-59; options may be NULL)
 * run     <- interface.run       (device.h:49, scheduler.c:134; must return with outputs complete)
 * destroy <- interface.post_run / release_graph (device.h:52-58, scheduler.c:201, subgraph.c:53-56) */
TAMD_API int tamd_graph_prerun(tamd_graph* g, const tamd_options* opt);
/* 2: the graph was compiled as two half-batch device graphs (tamd_options.split_batch); 0: one launch list */
TAMD_API int tamd_graph_halves(const tamd_graph* g);
TAMD_API int tamd_graph_input_num(const tamd_graph* g);
TAMD_API int tamd_graph_output_num(const tamd_graph* g);
/* dims/dtype of graph input / output `idx` (NCHW); returns dim_num */
TAMD_API int tamd_graph_input_desc(const tamd_graph* g, int idx, int* dims8, int* dtype);
TAMD_API int tamd_graph_output_desc(const tamd_graph* g, int idx, int* dims8, int* dtype, float* scale, int* zero_point);
/* host NCHW buffers; the input pointer is re-read at every run (tm_benchmark.cc:95-102 semantics) */
TAMD_API int tamd_graph_set_input(tamd_graph* g, int idx, const void* host_nchw, size_t bytes);
TAMD_API int tamd_graph_set_output(tamd_graph* g, int idx, void* host_nchw, size_t bytes);
/* H2D inputs -> kernels -> D2H outputs; returns when outputs are complete (scheduler is synchronous) */
TAMD_API int tamd_graph_run(tamd_graph* g);
/* asynchronous pair <- interface.async_run / async_wait (device.h:60-63; the reference's scheduler never issues them:
 * run_graph(graph, 0) is rejected, scheduler.c:75-79).  run_async stages the current input buffers, queues H2D -> kernels ->
 * D2H and returns; up to TWO runs may be in flight (a third submit fails); wait blocks for the OLDEST one and delivers its
 * outputs to the buffers that were set when it was submitted.  Same bytes as tamd_graph_run; with direct dispatch each run is one
 * burst on the graph's HSA queue (the second queues behind the first's closing packet), else both ride the graph's stream.
 * While runs are in flight the other entry points that touch the I/O buffers (run, upload_inputs, download_outputs) fail. */
TAMD_API int tamd_graph_run_async(tamd_graph* g);
TAMD_API int tamd_graph_wait(tamd_graph* g);
TAMD_API int tamd_graph_inflight(const tamd_graph* g);   /* runs submitted and not yet waited for */
/* HBM-resident variants (measurement, multi-GPU harness): stage inputs once, launch without copies */
TAMD_API int tamd_graph_upload_inputs(tamd_graph* g);
TAMD_API int tamd_graph_launch(tamd_graph* g);            /* async on the graph's stream          */
TAMD_API int tamd_graph_sync(tamd_graph* g);
/* packets per pass when tamd_graph_launch dispatches directly (tamd_options.direct_dispatch took effect), else 0 */
TAMD_API int tamd_graph_direct_packets(const tamd_graph* g);
/* of those packets, how many carry hidden arguments placed at the offsets the kernel's own code-object metadata lists (the rest
 * use the code-object-v5 default layout, vouched for by the prerun self-check) */
TAMD_API int tamd_graph_direct_meta_packets(const tamd_graph* g);
/* MEASUREMENT (round 6): `passes` back-to-back passes of the directly dispatched launch list with every packet stamped by the HSA
 * runtime's own dispatch profiling (hsa_amd_profiling_get_dispatch_time -- the timestamps a kernel trace reports, without a tool
 * intercepting the queue): dur_us[i] = mean duration of packet i, gap_us[i] = mean time from its end to the next packet's start
 * (the last packet: to the first packet of the next pass).  On this stack the stamps of consecutive barrier-bit packets ABUT (gap 0): a
 * packet's "duration" is its kernel plus the launch boundary in front of it, and the per-packet completion signals the stamps need cost
 * ~0.6 us per packet -- the sums equal the host's clock of an unstamped pass within 0.5 % on the batched configs, +18 % at batch 1.  Returns the packet count (<= max_packets), -1 when the graph does not
 * dispatch directly.  tamd_graph_direct_packet_name(g, i): kernel symbol of packet i. */
TAMD_API int tamd_graph_direct_timestamps(tamd_graph* g, int passes, double* dur_us, double* gap_us, int max_packets);
TAMD_API const char* tamd_graph_direct_packet_name(const tamd_graph* g, int i);
TAMD_API int tamd_graph_download_outputs(tamd_graph* g);
/* device pointer + byte size of graph output `idx` in the reference's NCHW order (for RCCL gather).  Valid after any pass: a
 * direct-dispatch tamd_graph_run / tamd_graph_wait leaves its outputs in the pinned host buffers only (zero-copy lists) and this
 * call refreshes the device copy from there first; it fails while asynchronous runs are in flight.  For a graph compiled as two
 * half-batch device graphs (tamd_options.split_batch) the pointer is a buffer of the pair into which THIS CALL gathers the two halves'
 * outputs (it waits for their passes first): call it after the pass whose outputs are wanted, not once before a loop of passes. */
TAMD_API int tamd_graph_output_device(tamd_graph* g, int idx, void** dptr, size_t* bytes);
TAMD_API void* tamd_graph_stream(tamd_graph* g);          /* hipStream_t                           */
/* wall time tamd_graph_prerun took (planning incl. the plan-time autotune, capture, direct-dispatch programs), milliseconds */
TAMD_API double tamd_graph_prerun_ms(const tamd_graph* g);
/* time `iters` back-to-back launches with HIP events on the graph's stream -> total ms */
TAMD_API int tamd_graph_time_launches(tamd_graph* g, int iters, float* total_ms);

/* ---- introspection / profiling: the TG_DEBUG_TIME / TG_DEBUG_DATA analogues of the CPU device
 * (source/device/cpu/cpu_define.h:41-43, cpu_dump.c:607-697) ---------------------------------- */
typedef struct tamd_kernel_info {
    char node[64];        /* graph node name                                     */
    char kernel[48];      /* device kernel family                                */
    double macs;          /* multiply-accumulates of this launch                 */
    double bytes;         /* algorithmic bytes: in + out + weights (+bias)       */
    float ms;             /* average duration (tamd_graph_profile)               */
} tamd_kernel_info;
TAMD_API int tamd_graph_kernel_num(const tamd_graph* g);
/* eager per-launch timing with hipEvent pairs on the graph's stream, averaged over `iters` */
TAMD_API int tamd_graph_profile(tamd_graph* g, int iters, tamd_kernel_info* out, int max_out);
/* copy any tensor back to host in NCHW (debug / layer-by-layer parity, cf. TG_DEBUG_DATA) */
TAMD_API int tamd_graph_read_tensor(tamd_graph* g, int tensor_idx, void* host_nchw, size_t bytes);
TAMD_API int tamd_graph_tensor_num(const tamd_graph* g);
TAMD_API int tamd_graph_tensor_desc(const tamd_graph* g, int tensor_idx, int* dims8, int* dtype);

TAMD_API void tamd_graph_destroy(tamd_graph* g);

#ifdef __cplusplus
}
#endif
#endif
