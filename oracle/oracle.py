"""TEST INFRASTRUCTURE ONLY -- Python face of the CPU oracle restatement (oracle/tg_oracle.c) and a
graph-level runner that executes a tengine_amd.tm2.Graph node by node in NCHW, choosing per conv
node the formula the reference's own kernel selection would use.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(HERE, "libtg_oracle.so")
_lib = None

DT_FP32, DT_INT8, DT_UINT8, DT_INT32 = 0, 2, 3, 4
CONV_HCL, CONV_REF = 0, 1


def build():
    subprocess.check_call(["make", "-s", "-C", HERE])


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(HERE, "tg_oracle.c")
        if not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
            build()
        _lib = C.CDLL(_LIB)
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def conv_variant(batch, p, cin, cout):
    g = p.get("group", 1)
    return lib().orc_conv_int8_variant(batch, g, cin // g, cout // g, p["kernel_h"], p["kernel_w"], p["stride_h"],
                                       p["stride_w"], p.get("pad_h0", 0), p.get("pad_h1", 0), p.get("pad_w0", 0),
                                       p.get("pad_w1", 0), p.get("dilation_h", 1), p.get("dilation_w", 1))


def conv_out_dims(xd, wd, p):
    """source/operator/prototype/convolution.c:35-145 (explicit-pad branch)."""
    n, _, h, w = xd
    oh = (h - p.get("dilation_h", 1) * (p["kernel_h"] - 1) - 1 + p.get("pad_h0", 0) + p.get("pad_h1", 0)) // p["stride_h"] + 1
    ow = (w - p.get("dilation_w", 1) * (p["kernel_w"] - 1) - 1 + p.get("pad_w0", 0) + p.get("pad_w1", 0)) // p["stride_w"] + 1
    return [n, wd[0], max(oh, 1), max(ow, 1)]


def conv2d_int8(x, w, bias, p, in_scale, w_scales, out_scale, variant=None):
    x = np.ascontiguousarray(x, np.int8)
    w = np.ascontiguousarray(w, np.int8)
    n, cin, h, wd = x.shape
    od = conv_out_dims(x.shape, w.shape, p)
    if variant is None:
        variant = conv_variant(n, p, cin, od[1])
    y = np.empty(od, np.int8)
    ws = np.ascontiguousarray(w_scales, np.float32)
    b = None if bias is None else np.ascontiguousarray(bias, np.int32)
    rc = lib().orc_conv2d_int8(_p(x), _p(w), _p(b), _p(y), n, cin, h, wd, od[1], od[2], od[3], p["kernel_h"],
                               p["kernel_w"], p["stride_h"], p["stride_w"], p.get("pad_h0", 0), p.get("pad_w0", 0),
                               p.get("dilation_h", 1), p.get("dilation_w", 1), p.get("group", 1),
                               p.get("activation", -1), C.c_float(in_scale), _p(ws), C.c_float(out_scale), variant)
    assert rc == 0
    return y


def conv2d_fp32(x, w, bias, p):
    x = np.ascontiguousarray(x, np.float32)
    w = np.ascontiguousarray(w, np.float32)
    n, cin, h, wd = x.shape
    od = conv_out_dims(x.shape, w.shape, p)
    y = np.empty(od, np.float32)
    b = None if bias is None else np.ascontiguousarray(bias, np.float32)
    rc = lib().orc_conv2d_fp32(_p(x), _p(w), _p(b), _p(y), n, cin, h, wd, od[1], od[2], od[3], p["kernel_h"],
                               p["kernel_w"], p["stride_h"], p["stride_w"], p.get("pad_h0", 0), p.get("pad_w0", 0),
                               p.get("dilation_h", 1), p.get("dilation_w", 1), p.get("group", 1),
                               p.get("activation", -1))
    assert rc == 0
    return y


def fc_int8(x, w, bias, in_scale, w_scales, out_scale):
    x = np.ascontiguousarray(x, np.int8).reshape(x.shape[0], -1)
    w = np.ascontiguousarray(w, np.int8)
    y = np.empty((x.shape[0], w.shape[0]), np.int8)
    ws = np.ascontiguousarray(w_scales, np.float32)
    b = None if bias is None else np.ascontiguousarray(bias, np.int32)
    lib().orc_fc_int8(_p(x), _p(w), _p(b), _p(y), x.shape[0], x.shape[1], w.shape[0], C.c_float(in_scale), _p(ws),
                      C.c_float(out_scale))
    return y


def fc_fp32(x, w, bias):
    x = np.ascontiguousarray(x, np.float32).reshape(x.shape[0], -1)
    w = np.ascontiguousarray(w, np.float32)
    y = np.empty((x.shape[0], w.shape[0]), np.float32)
    b = None if bias is None else np.ascontiguousarray(bias, np.float32)
    lib().orc_fc_fp32(_p(x), _p(w), _p(b), _p(y), x.shape[0], x.shape[1], w.shape[0])
    return y


def pool_resolve(xd, p):
    """infer_shape of Pooling (source/operator/prototype/pooling.c:36-100 + pooling_param.h:59-105):
    returns (out_h, out_w, kernel_h, kernel_w, stride_h, stride_w, pad_h0, pad_w0)."""
    from tengine_amd.models import pool_out
    n, c, h, w = xd
    kh, kw, sh, sw = p["kernel_h"], p["kernel_w"], p["stride_h"], p["stride_w"]
    ph0, pw0 = p.get("pad_h0", 0), p.get("pad_w0", 0)
    glob = p.get("global", 0)
    if kh == h and kw == w and ph0 == 0 and pw0 == 0 and p.get("pad_h1", 0) == 0 and p.get("pad_w1", 0) == 0:
        glob = 1
    if glob:
        return 1, 1, h, w, 1, 1, 0, 0
    caffe = p.get("caffe_flavor", 0)
    oh, rh0, _ = pool_out(h, kh, sh, ph0, caffe)
    ow, rw0, _ = pool_out(w, kw, sw, pw0, caffe)
    if caffe == 2:
        rh0, rw0 = ph0 // 2, pw0 // 2
    return oh, ow, kh, kw, sh, sw, rh0, rw0


def pool_int8(x, p, in_scale, out_scale):
    x = np.ascontiguousarray(x, np.int8)
    n, c, h, w = x.shape
    oh, ow, kh, kw, sh, sw, ph0, pw0 = pool_resolve(x.shape, p)
    y = np.empty((n, c, oh, ow), np.int8)
    rc = lib().orc_pool_int8(_p(x), _p(y), n, c, h, w, oh, ow, kh, kw, sh, sw, ph0, pw0, p["alg"],
                             p.get("caffe_flavor", 0), C.c_float(in_scale), C.c_float(out_scale))
    assert rc == 0
    return y


def pool_fp32(x, p):
    x = np.ascontiguousarray(x, np.float32)
    n, c, h, w = x.shape
    oh, ow, kh, kw, sh, sw, ph0, pw0 = pool_resolve(x.shape, p)
    y = np.empty((n, c, oh, ow), np.float32)
    rc = lib().orc_pool_fp32(_p(x), _p(y), n, c, h, w, oh, ow, kh, kw, sh, sw, ph0, pw0, p["alg"],
                             p.get("caffe_flavor", 0))
    assert rc == 0
    return y


def relu_int8(x, slope, in_scale, out_scale):
    x = np.ascontiguousarray(x, np.int8)
    y = np.empty_like(x)
    lib().orc_relu_int8(_p(x), _p(y), C.c_size_t(x.size), C.c_float(slope), C.c_float(in_scale), C.c_float(out_scale))
    return y


def eltwise_int8(a, b, etype, sa, sb, out_scale):
    a = np.ascontiguousarray(a, np.int8)
    b = np.ascontiguousarray(b, np.int8)
    assert a.shape == b.shape
    y = np.empty_like(a)
    rc = lib().orc_eltwise_int8(_p(a), _p(b), _p(y), C.c_size_t(a.size), etype, C.c_float(sa), C.c_float(sb),
                                C.c_float(out_scale))
    assert rc == 0
    return y


# ---- uint8 (per-tensor asymmetric; the reference simulates it in fp32) -----------------------------------
def conv2d_uint8(x, w, bias, p, in_q, w_q, out_q, variant=None):
    """*_q = (scale, zero_point).  variant None -> the reference's selection: group 1 -> hcl, else ref
    (conv_dw_hcl_x86.c:533 rejects uint8, so depthwise always lands on conv_ref)."""
    x = np.ascontiguousarray(x, np.uint8)
    w = np.ascontiguousarray(w, np.uint8)
    n, cin, h, wd = x.shape
    od = conv_out_dims(x.shape, w.shape, p)
    if variant is None:
        variant = CONV_HCL if p.get("group", 1) == 1 else CONV_REF
    y = np.empty(od, np.uint8)
    b = None if bias is None else np.ascontiguousarray(bias, np.int32)
    rc = lib().orc_conv2d_uint8(_p(x), _p(w), _p(b), _p(y), n, cin, h, wd, od[1], od[2], od[3], p["kernel_h"],
                                p["kernel_w"], p["stride_h"], p["stride_w"], p.get("pad_h0", 0), p.get("pad_w0", 0),
                                p.get("dilation_h", 1), p.get("dilation_w", 1), p.get("group", 1),
                                p.get("activation", -1), C.c_float(in_q[0]), int(in_q[1]), C.c_float(w_q[0]),
                                int(w_q[1]), C.c_float(out_q[0]), int(out_q[1]), variant)
    assert rc == 0
    return y


def fc_uint8(x, w, bias, in_q, w_q, bias_scale, out_q):
    x = np.ascontiguousarray(x, np.uint8).reshape(x.shape[0], -1)
    w = np.ascontiguousarray(w, np.uint8)
    y = np.empty((x.shape[0], w.shape[0]), np.uint8)
    b = None if bias is None else np.ascontiguousarray(bias, np.int32)
    lib().orc_fc_uint8(_p(x), _p(w), _p(b), _p(y), x.shape[0], x.shape[1], w.shape[0], C.c_float(in_q[0]), int(in_q[1]),
                       C.c_float(w_q[0]), int(w_q[1]), C.c_float(bias_scale), C.c_float(out_q[0]), int(out_q[1]))
    return y


def pool_uint8(x, p, in_q, out_q):
    x = np.ascontiguousarray(x, np.uint8)
    n, c, h, w = x.shape
    oh, ow, kh, kw, sh, sw, ph0, pw0 = pool_resolve(x.shape, p)
    y = np.empty((n, c, oh, ow), np.uint8)
    rc = lib().orc_pool_uint8(_p(x), _p(y), n, c, h, w, oh, ow, kh, kw, sh, sw, ph0, pw0, p["alg"],
                              p.get("caffe_flavor", 0), C.c_float(in_q[0]), int(in_q[1]), C.c_float(out_q[0]),
                              int(out_q[1]))
    assert rc == 0
    return y


def relu_uint8(x, slope, in_q, out_q):
    x = np.ascontiguousarray(x, np.uint8)
    y = np.empty_like(x)
    lib().orc_relu_uint8(_p(x), _p(y), C.c_size_t(x.size), C.c_float(slope), C.c_float(in_q[0]), int(in_q[1]),
                         C.c_float(out_q[0]), int(out_q[1]))
    return y


def requant_copy_uint8(x, in_q, out_q):
    x = np.ascontiguousarray(x, np.uint8)
    y = np.empty_like(x)
    lib().orc_requant_copy_uint8(_p(x), _p(y), C.c_size_t(x.size), C.c_float(in_q[0]), int(in_q[1]),
                                 C.c_float(out_q[0]), int(out_q[1]))
    return y


def upsample_uint8(x, scale, in_q, out_q):
    x = np.ascontiguousarray(x, np.uint8)
    n, c, h, w = x.shape
    y = np.empty((n, c, h * scale, w * scale), np.uint8)
    lib().orc_upsample_uint8(_p(x), _p(y), n, c, h, w, scale, C.c_float(in_q[0]), int(in_q[1]), C.c_float(out_q[0]),
                             int(out_q[1]))
    return y


def requant_copy_int8(x, in_scale, out_scale):
    x = np.ascontiguousarray(x, np.int8)
    y = np.empty_like(x)
    lib().orc_requant_copy_int8(_p(x), _p(y), C.c_size_t(x.size), C.c_float(in_scale), C.c_float(out_scale))
    return y


def softmax_uint8(x, axis, in_q, out_q):
    x = np.ascontiguousarray(x, np.uint8)
    axis = axis % x.ndim
    outer = int(np.prod(x.shape[:axis])) if axis else 1
    inner = int(np.prod(x.shape[axis + 1:])) if axis + 1 < x.ndim else 1
    y = np.empty_like(x)
    lib().orc_softmax_uint8(_p(x), _p(y), outer, x.shape[axis], inner, C.c_float(in_q[0]), int(in_q[1]),
                            C.c_float(out_q[0]), int(out_q[1]))
    return y


def softmax_int8(x, axis, in_scale, out_scale):
    x = np.ascontiguousarray(x, np.int8)
    axis = axis % x.ndim
    outer = int(np.prod(x.shape[:axis])) if axis else 1
    inner = int(np.prod(x.shape[axis + 1:])) if axis + 1 < x.ndim else 1
    y = np.empty_like(x)
    lib().orc_softmax_int8(_p(x), _p(y), outer, x.shape[axis], inner, C.c_float(in_scale), C.c_float(out_scale))
    return y


def priorbox(feat_dims, data_dims, p, dtype, out_q=None):
    """PriorBox node (priorbox_ref.c:53-213): [1, 2, out_dim, 1] in the output tensor's dtype."""
    fa = lambda v: np.ascontiguousarray(v, np.float32)
    mn, mx, ar, var = fa(p["min_size"]), fa(p.get("max_size", [])), fa(p.get("aspect_ratio", [])), fa(p["variance"])
    num = (len(ar) * (2 if p.get("flip", 0) else 1) + 1 + (1 if len(mx) else 0)) * len(mn)
    dim = feat_dims[2] * feat_dims[3] * num * 4
    f = np.zeros((1, 2, dim, 1), np.float32)
    fn = lib().orc_priorbox_f32
    fn.restype = C.c_int
    rc = fn(_p(f), feat_dims[2], feat_dims[3], data_dims[2], data_dims[3], int(p.get("img_h", 0)), int(p.get("img_w", 0)),
            C.c_float(p.get("step_h", 0.0)), C.c_float(p.get("step_w", 0.0)), C.c_float(p.get("offset", 0.5)), _p(mn),
            len(mn), _p(mx), len(mx), _p(ar), len(ar), _p(var), int(p.get("flip", 0)), int(p.get("clip", 0)))
    assert rc == num, rc
    if dtype == DT_FP32:
        return f
    if dtype == DT_UINT8:
        y = np.empty(f.shape, np.uint8)
        lib().orc_priorbox_quant_uint8(_p(f), _p(y), C.c_size_t(f.size), C.c_float(out_q[0]), int(out_q[1]))
        return y
    y = np.empty(f.shape, np.int8)
    lib().orc_priorbox_quant_int8(_p(f), _p(y), C.c_size_t(f.size), C.c_float(out_q[0]))
    return y


def eltwise_uint8(a, b, etype, qa, qb, out_q):
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    assert a.shape == b.shape
    y = np.empty_like(a)
    rc = lib().orc_eltwise_uint8(_p(a), _p(b), _p(y), C.c_size_t(a.size), etype, C.c_float(qa[0]), int(qa[1]),
                                 C.c_float(qb[0]), int(qb[1]), C.c_float(out_q[0]), int(out_q[1]))
    assert rc == 0
    return y


def run_graph(g, x, keep_all=False, teacher=None, report=None):
    """Execute a tm2.Graph on the oracle. Returns ({tensor idx: ndarray} if keep_all else list of outputs).
    `teacher` = {tensor idx: ndarray} (e.g. the real reference's intermediate tensors): every node output found
    there is compared (|diff| histogram appended to `report` as (node name, n, n_bad, max)) and then REPLACED by
    the teacher's value, so each node is checked in isolation on exactly the inputs the teacher saw."""
    from tengine_amd import tm2
    T = g.tensors
    vals = {i: t.data for i, t in enumerate(T) if t.ttype == tm2.TT_CONST}
    sc = lambda i: np.float32(T[i].scales[0])
    qp = lambda i: (np.float32(T[i].scales[0]), int(T[i].zero_points[0]) if T[i].zero_points else 0)
    for n in g.nodes:
        op, p = n.op, n.params
        if op == "Const":
            continue
        if op == "InputOp":
            vals[n.outputs[0]] = x
            continue
        i0, o0 = n.inputs[0], n.outputs[0]
        a = vals[i0]
        dt = T[o0].dtype
        if op == "PriorBox":
            if T[o0].dims[0] != 1:      # priorbox_ref.c fills image 0 only; the rest of the tensor is whatever malloc returned
                raise NotImplementedError("PriorBox with batch > 1 is undefined in the reference")
            y = priorbox(T[i0].dims, T[n.inputs[1]].dims, p, dt, qp(o0) if dt != DT_FP32 else None)
        elif op == "Convolution":
            w = vals[n.inputs[1]]
            b = vals[n.inputs[2]] if len(n.inputs) > 2 else None
            if dt == DT_INT8:
                y = conv2d_int8(a, w, b, p, sc(i0), T[n.inputs[1]].scales, sc(o0))
            elif dt == DT_FP32:
                y = conv2d_fp32(a, w, b, p)
            elif dt == DT_UINT8:
                y = conv2d_uint8(a, w, b, p, qp(i0), qp(n.inputs[1]), qp(o0))
            else:
                raise NotImplementedError("oracle conv dtype %d" % dt)
        elif op == "FullyConnected":
            w = vals[n.inputs[1]]
            b = vals[n.inputs[2]] if len(n.inputs) > 2 else None
            if dt == DT_UINT8:
                bs = sc(n.inputs[2]) if b is not None else np.float32(0)
                y = fc_uint8(a, w, b, qp(i0), qp(n.inputs[1]), bs, qp(o0))
            else:
                y = fc_int8(a, w, b, sc(i0), T[n.inputs[1]].scales, sc(o0)) if dt == DT_INT8 else fc_fp32(a, w, b)
        elif op == "Pooling":
            if dt == DT_UINT8:
                y = pool_uint8(a, p, qp(i0), qp(o0))
            else:
                y = pool_int8(a, p, sc(i0), sc(o0)) if dt == DT_INT8 else pool_fp32(a, p)
        elif op == "ReLU":
            if dt == DT_INT8:
                y = relu_int8(a, p.get("negative_slope", 0.0), sc(i0), sc(o0))
            elif dt == DT_UINT8:
                y = relu_uint8(a, p.get("negative_slope", 0.0), qp(i0), qp(o0))
            else:
                s = np.float32(p.get("negative_slope", 0.0))
                y = np.where(a < 0, a * s, a).astype(np.float32)
        elif op == "Eltwise":
            b2 = vals[n.inputs[1]]
            if dt == DT_INT8:
                y = eltwise_int8(a, b2, p["type"], sc(i0), sc(n.inputs[1]), sc(o0))
            elif dt == DT_UINT8:
                y = eltwise_uint8(a, b2, p["type"], qp(i0), qp(n.inputs[1]), qp(o0))
            else:
                y = {tm2.ELT_SUM: a + b2, tm2.ELT_PROD: a * b2, tm2.ELT_SUB: a - b2, tm2.ELT_MAX: np.maximum(a, b2)}[p["type"]]
        elif op == "Dropout":
            y = a
        elif op == "Permute":       # permute/permute_ref.c:203-296: a byte copy in the new order, no requantisation
            y = np.ascontiguousarray(np.transpose(a, p["order"]))
        elif op == "Reshape":       # reshape/reshape_ref.c: element copy into the inferred shape (reshape.c:37-160)
            y = a.reshape(g.tensors[o0].dims)
        elif op == "Softmax" and dt == DT_UINT8:
            y = softmax_uint8(a, p.get("axis", 1), qp(i0), qp(o0))
        elif op == "Softmax" and dt == DT_INT8:
            y = softmax_int8(a, p.get("axis", 1), sc(i0), sc(o0))
        elif op == "Flatten":       # flatten/flatten_ref.c:53-77: element copy; shape [n, prod(rest)] (flatten.c:34-61)
            y = a.reshape(a.shape[0], -1)
        elif op == "Concat" and dt in (DT_UINT8, DT_INT8) and len(n.inputs) == 1:
            # concat_kernel_ref_int8.c:47-57 / concat_kernel_ref_uint8.c:47-58: a SINGLE input is copied byte for byte, whatever the
            # two tensors' quantisation says (found in round 6 by tools/fuzz_heads.py --ref: the restatement used to rescale it)
            y = vals[n.inputs[0]].reshape(g.tensors[o0].dims).copy()
        elif op == "Concat" and dt == DT_UINT8:
            y = np.concatenate([requant_copy_uint8(vals[i], qp(i), qp(o0)) for i in n.inputs], axis=p.get("axis", 1))
        elif op == "Concat" and dt == DT_INT8:
            y = np.concatenate([requant_copy_int8(vals[i], sc(i), sc(o0)) for i in n.inputs], axis=p.get("axis", 1))
        elif op == "Concat" and dt == DT_FP32:
            y = np.concatenate([vals[i] for i in n.inputs], axis=p.get("axis", 1))
        elif op == "Upsample" and dt == DT_UINT8:
            y = upsample_uint8(a, int(p.get("scale", 2)), qp(i0), qp(o0))
        elif op == "Softmax" and dt == DT_FP32:
            ax = p.get("axis", 1)
            e = np.exp(a - a.max(axis=ax, keepdims=True))
            y = (e / e.sum(axis=ax, keepdims=True)).astype(np.float32)
        else:
            raise NotImplementedError("oracle op %s dtype %d" % (op, dt))
        if teacher is not None and o0 in teacher:
            tv = np.asarray(teacher[o0]).reshape(y.shape)
            if report is not None:
                d = np.abs(tv.astype(np.int64) - y.astype(np.int64)) if y.dtype != np.float32 else np.abs(tv - y)
                report.append((n.name, int(y.size), int(np.count_nonzero(d)), float(d.max())))
            y = tv
        vals[o0] = y
    if keep_all:
        return vals
    return [vals[g.nodes[ni].outputs[0]] for ni in g.output_nodes]
