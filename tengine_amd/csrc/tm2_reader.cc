// Native tmfile-v2 loader: the SAME bytes the reference's serializer consumes
// (source/serializer/tmfile/tm2_serializer.c:157-466 load_graph_tensors, :468-760 load_graph_nodes,
// :915-936 load_mem) become a tamd_graph, so a model file -- or the RCCL-broadcast copy of it -- needs
// no Tengine host to run on the device.  Format restated from tm2_format.h:267-477 (little-endian,
// uint32 offsets from the buffer start, 0 == not set).  Only NCHW models (graph_layout 0) are accepted;
// every offset is bounds-checked because the bytes may come off the wire.
#include <stdint.h>
#include <string.h>

#include <string>
#include <vector>

#include "graph.h"

namespace {

struct Reader {
    const uint8_t* b;
    size_t n;
    bool ok = true;
    bool in(size_t off, size_t len) const { return off <= n && len <= n - off; }
    uint32_t u32(size_t off)
    {
        if (!in(off, 4)) { ok = false; return 0; }
        uint32_t v;
        memcpy(&v, b + off, 4);
        return v;
    }
    int32_t i32(size_t off) { return (int32_t)u32(off); }
    float f32(size_t off)
    {
        uint32_t v = u32(off);
        float f;
        memcpy(&f, &v, 4);
        return f;
    }
    std::vector<uint32_t> vec(size_t off)
    {
        std::vector<uint32_t> r;
        if (off == 0) return r;
        uint32_t cnt = u32(off);
        if (!ok || !in(off + 4, (size_t)cnt * 4)) { ok = false; return r; }
        r.resize(cnt);
        memcpy(r.data(), b + off + 4, (size_t)cnt * 4);
        return r;
    }
    std::string str(size_t off)
    {
        if (off == 0) return "";
        uint32_t size = u32(off), d = u32(off + 4);
        if (!ok || !in(d, size)) { ok = false; return ""; }
        std::string s((const char*)b + d, size);
        size_t z = s.find('\0');
        if (z != std::string::npos) s.resize(z);
        return s;
    }
};

// TM2_OPTYPE_* (tm2_format.h:157-264) -> TAMD_OP_*
int map_op(uint32_t t)
{
    switch (t) {
    case 3: return TAMD_OP_CONCAT;
    case 4: return TAMD_OP_CONST;
    case 5: return TAMD_OP_CONV;
    case 8: return TAMD_OP_DROPOUT;
    case 9: return TAMD_OP_ELTWISE;
    case 10: return TAMD_OP_FLATTEN;
    case 11: return TAMD_OP_FC;
    case 12: return TAMD_OP_INPUT;
    case 15: return TAMD_OP_PERMUTE;
    case 16: return TAMD_OP_POOL;
    case 18: return TAMD_OP_PRIORBOX;
    case 20: return TAMD_OP_RELU;
    case 21: return TAMD_OP_RELU6;
    case 23: return TAMD_OP_RESHAPE;
    case 28: return TAMD_OP_SOFTMAX;
    case 51: return TAMD_OP_UPSAMPLE;
    default: return -1;
    }
}

}  // namespace

extern "C" tamd_graph* tamd_graph_load_tm2(const void* mem, size_t size)
{
    using namespace tamd;
    Reader r{(const uint8_t*)mem, size};
    if (!mem || size < 16) { set_error("tm2: buffer too small"); return nullptr; }
    uint16_t ver_main;
    memcpy(&ver_main, r.b, 2);
    if (ver_main != 2) { set_error("tm2: unsupported file version %u", ver_main); return nullptr; }
    const uint32_t root = r.u32(8);                           // TM2_Header.offset_root
    const uint32_t vo_subgraphs = r.u32(root + 8);             // TM2_Model.offset_vo_subgraphs
    std::vector<uint32_t> subs = r.vec(vo_subgraphs);
    if (!r.ok || subs.empty()) { set_error("tm2: no subgraph"); return nullptr; }
    const size_t sg = subs[0];                                 // only subgraph 0 is read (tm2_serializer.c:93-99)
    const int graph_layout = r.i32(sg + 4);
    if (graph_layout != 0) { set_error("tm2: only NCHW models are supported"); return nullptr; }
    std::vector<uint32_t> in_nodes = r.vec(r.u32(sg + 12)), out_nodes = r.vec(r.u32(sg + 16));
    std::vector<uint32_t> nodes = r.vec(r.u32(sg + 20)), tensors = r.vec(r.u32(sg + 24)), buffers = r.vec(r.u32(sg + 28));
    if (!r.ok) { set_error("tm2: corrupt subgraph table"); return nullptr; }

    tamd_graph* g = tamd_graph_create();
    auto fail = [&](const char* msg) { set_error("tm2: %s", msg); tamd_graph_destroy(g); return (tamd_graph*)nullptr; };

    for (uint32_t to : tensors) {                              // TM2_Tensor
        const uint32_t buffer_id = r.u32(to + 4), dims_o = r.u32(to + 8), name_o = r.u32(to + 12), q_o = r.u32(to + 16);
        const int ttype = r.i32(to + 24), dtype = r.i32(to + 28);
        std::vector<uint32_t> dv = r.vec(dims_o);
        std::vector<uint32_t> qv = r.vec(q_o);
        std::string name = r.str(name_o);
        if (!r.ok || dv.size() > 8) return fail("corrupt tensor");
        tamd_tensor_desc d{};
        d.dtype = dtype; d.ttype = ttype; d.dim_num = (int)dv.size();
        size_t elems = 1;
        for (size_t i = 0; i < dv.size(); i++) { d.dims[i] = (int)dv[i]; elems *= (size_t)dv[i]; }
        std::vector<float> scales;
        std::vector<int> zps;
        for (uint32_t qo : qv) { zps.push_back(r.i32(qo)); scales.push_back(r.f32(qo + 4)); }   // TM2_QuantParam{zp, scale, width}
        d.quant_num = (int)scales.size(); d.scales = scales.data(); d.zero_points = zps.data();
        d.name = name.c_str();
        if (ttype == TAMD_TT_CONST) {
            if (buffer_id >= buffers.size()) return fail("bad buffer id");
            const uint32_t bsize = r.u32(buffers[buffer_id]), boff = r.u32(buffers[buffer_id] + 4);   // TM2_Buffer
            const size_t es = (dtype == TAMD_DT_FP32 || dtype == TAMD_DT_INT32) ? 4 : (dtype == TAMD_DT_FP16 ? 2 : 1);
            if (boff) {
                if (!r.in(boff, bsize) || elems * es > bsize) return fail("const tensor size in model is too small");
                d.data = r.b + boff;
            }                                                   // else: structure-only model, zero filled
        }
        if (!r.ok) return fail("corrupt tensor record");
        if (tamd_graph_add_tensor(g, &d) < 0) return fail("add_tensor failed");
    }

    for (uint32_t no : nodes) {                                // TM2_Node
        std::vector<uint32_t> vi = r.vec(r.u32(no + 4)), vo = r.vec(r.u32(no + 8));
        const uint32_t op_o = r.u32(no + 12);
        std::string name = r.str(r.u32(no + 16));
        const uint32_t optype = r.u32(op_o + 4), po = r.u32(op_o + 8);   // TM2_Operator
        if (!r.ok) return fail("corrupt node");
        const int op = map_op(optype);
        if (op < 0) { set_error("tm2: operator type %u (%s) is not supported by the device backend", optype, name.c_str()); tamd_graph_destroy(g); return nullptr; }
        NodeParam p{};
        bool have_param = po != 0;
        if (op == TAMD_OP_RESHAPE) {        // TM2_ReshapeParam carries the recipe (re_shape, is_mxnet, is_onnx: reshape.c:37-160); the
            // serializer also stores the RESOLVED shape with the output tensor, which is what the device needs
            if (vo.empty() || vo[0] >= g->tensors.size()) return fail("reshape without output tensor");
            const HTensor& ot = g->tensors[vo[0]];
            if (ot.dims.empty() || ot.dims.size() > 8) return fail("reshape output without a shape");
            p.reshape.dim_num = (int)ot.dims.size();
            for (size_t i = 0; i < ot.dims.size(); i++) p.reshape.dims[i] = ot.dims[i];
            have_param = true;
        } else if (po) {
            switch (op) {
            case TAMD_OP_CONV: {                               // TM2_ConvParam tm2_format.h:419-435 (tm2_conv.c:50-73)
                tamd_conv_param& c = p.conv;
                c.kernel_h = r.i32(po); c.kernel_w = r.i32(po + 4); c.stride_h = r.i32(po + 8); c.stride_w = r.i32(po + 12);
                c.dilation_h = r.i32(po + 16); c.dilation_w = r.i32(po + 20); c.input_channel = r.i32(po + 24);
                c.output_channel = r.i32(po + 28); c.group = r.i32(po + 32); c.activation = r.i32(po + 36);
                c.pad_h0 = r.i32(po + 40); c.pad_w0 = r.i32(po + 44); c.pad_h1 = r.i32(po + 48); c.pad_w1 = r.i32(po + 52);
                break;
            }
            case TAMD_OP_POOL: {                               // TM2_PoolParam :510-524 (tm2_pool.c)
                tamd_pool_param& q = p.pool;
                q.pool_method = (int)r.u32(po); q.kernel_h = r.i32(po + 4); q.kernel_w = r.i32(po + 8);
                q.stride_h = r.i32(po + 12); q.stride_w = r.i32(po + 16); q.global = r.i32(po + 20);
                q.caffe_flavor = r.i32(po + 24); q.pad_h0 = r.i32(po + 28); q.pad_w0 = r.i32(po + 32);
                q.pad_h1 = r.i32(po + 36); q.pad_w1 = r.i32(po + 40);
                break;
            }
            case TAMD_OP_FC: p.fc.num_output = r.i32(po); break;
            case TAMD_OP_RELU: p.relu.negative_slope = r.f32(po); break;
            case TAMD_OP_ELTWISE:
                p.elt.type = (int)r.u32(po); p.elt.caffe_flavor = r.i32(po + 4); p.elt.shift = r.f32(po + 8);
                p.elt.power = r.f32(po + 12); p.elt.scale = r.f32(po + 16);
                break;
            case TAMD_OP_CONCAT: p.concat.axis = r.i32(po); break;
            case TAMD_OP_SOFTMAX: p.softmax.axis = r.i32(po); break;   // TM2_SoftmaxParam {axis}
            case TAMD_OP_UPSAMPLE: p.ups.scale = r.f32(po); break;
            case TAMD_OP_PRIORBOX: {                           // TM2_PriorBoxParam :526-542 (tm2_priorbox.c:43-84)
                tamd_priorbox_param& q = p.priorbox;
                auto floats = [&](uint32_t vo_, float* dst, int cap, int* num) {      // TM2_Vector_floats {v_num, data[]}
                    const uint32_t n = r.u32(vo_);
                    if (!r.ok || n > (uint32_t)cap) { r.ok = false; return; }
                    for (uint32_t i = 0; i < n; i++) dst[i] = r.f32(vo_ + 4 + 4 * i);
                    if (num) *num = (int)n;
                };
                int nvar = 0;
                floats(r.u32(po), q.min_size, TAMD_PRIORBOX_MAX, &q.min_size_num);
                floats(r.u32(po + 4), q.max_size, TAMD_PRIORBOX_MAX, &q.max_size_num);
                floats(r.u32(po + 8), q.variance, 4, &nvar);
                floats(r.u32(po + 12), q.aspect_ratio, TAMD_PRIORBOX_MAX, &q.aspect_ratio_num);
                if (nvar != 4) r.ok = false;
                q.flip = r.i32(po + 16); q.clip = r.i32(po + 20); q.image_h = r.i32(po + 28); q.image_w = r.i32(po + 32);
                q.step_w = r.f32(po + 36); q.step_h = r.f32(po + 40); q.offset = r.f32(po + 44);
                break;
            }
            case TAMD_OP_PERMUTE:                              // TM2_PermuteParam {flag, order0..3} (tm2_permute.c)
                for (int i = 0; i < 4; i++) p.perm.order[i] = r.i32(po + 4 + 4 * i);
                break;
            default: break;
            }
        }
        if (!r.ok) return fail("corrupt operator params");
        std::vector<int> ins(vi.begin(), vi.end()), outs(vo.begin(), vo.end());
        tamd_node_desc d{};
        d.op = op; d.input_num = (int)ins.size(); d.inputs = ins.data(); d.output_num = (int)outs.size();
        d.outputs = outs.data(); d.param = have_param ? &p : nullptr; d.name = name.c_str();
        if (tamd_graph_add_node(g, &d) < 0) { tamd_graph_destroy(g); return nullptr; }
    }
    std::vector<int> gi, go;
    for (uint32_t ni : in_nodes) { if (ni >= g->nodes.size() || g->nodes[ni].out.empty()) return fail("bad input node"); gi.push_back(g->nodes[ni].out[0]); }
    for (uint32_t ni : out_nodes) { if (ni >= g->nodes.size() || g->nodes[ni].out.empty()) return fail("bad output node"); go.push_back(g->nodes[ni].out[0]); }
    tamd_graph_set_inputs(g, (int)gi.size(), gi.data());
    tamd_graph_set_outputs(g, (int)go.size(), go.data());
    return g;
}
