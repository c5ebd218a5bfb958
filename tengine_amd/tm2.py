"""Tengine model file (tmfile v2) writer / reader and the tiny graph IR used to synthesise models.

The reference ships only structure-only benchmark tmfiles and no quantised tmfile at all
(SURVEY F7), so quantised models are synthesised here and written in the reference's own
on-disk format.  The SAME bytes are then consumed by
  * the reference loader (source/serializer/tmfile/tm2_serializer.c:157-466,865-936) -> CPU oracle,
  * this repo's native loader (tengine_amd/csrc/tm2_reader.cc)                      -> HIP backend,
  * the multi-GPU harness, which RCCL-broadcasts exactly these bytes.

Format facts restated from source/serializer/tmfile/tm2_format.h:267-477:
little-endian, every reference is a uint32 byte offset from the file start, 0 == "not set".
"""
import struct
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

# data types (source/api/c_api.h:58-63)
DT_FP32, DT_FP16, DT_INT8, DT_UINT8, DT_INT32, DT_INT16 = 0, 1, 2, 3, 4, 5
NP_OF_DT = {DT_FP32: np.float32, DT_FP16: np.float16, DT_INT8: np.int8, DT_UINT8: np.uint8,
            DT_INT32: np.int32, DT_INT16: np.int16}
# tensor types (c_api.h:70-74)
TT_VAR, TT_CONST, TT_INPUT = 1, 2, 3
LAYOUT_NCHW = 0

# tm2 operator type codes (tm2_format.h:157-264)
OPTYPE = {
    "Concat": 3, "Const": 4, "Convolution": 5, "Dropout": 8, "Eltwise": 9, "Flatten": 10,
    "DetectionOutput": 7, "FullyConnected": 11, "InputOp": 12, "Permute": 15, "Pooling": 16, "PriorBox": 18, "ReLU": 20, "ReLU6": 21,
    "Reshape": 23, "Softmax": 28, "Upsample": 51,
}
OPNAME = {v: k for k, v in OPTYPE.items()}

# eltwise types (source/operator/prototype/eltwise_param.h)
ELT_PROD, ELT_PROD_SCALAR, ELT_SUM, ELT_SUM_SCALAR, ELT_SUB, ELT_SUB_SCALAR, ELT_MAX = 0, 1, 2, 3, 4, 5, 6
POOL_MAX, POOL_AVG = 0, 1


@dataclass
class Tensor:
    name: str
    dims: List[int]
    dtype: int = DT_FP32
    ttype: int = TT_VAR
    data: Optional[np.ndarray] = None          # const payload (NCHW / OIHW), None for var/input
    scales: Optional[List[float]] = None       # 1 entry = per-tensor, N = per-channel
    zero_points: Optional[List[int]] = None


@dataclass
class Node:
    name: str
    op: str
    inputs: List[int]
    outputs: List[int]
    params: Dict = field(default_factory=dict)


@dataclass
class Graph:
    tensors: List[Tensor] = field(default_factory=list)
    nodes: List[Node] = field(default_factory=list)
    input_nodes: List[int] = field(default_factory=list)
    output_nodes: List[int] = field(default_factory=list)
    name: str = "tengine_amd_synth"

    # ---- builder helpers -------------------------------------------------------------
    def add_tensor(self, *a, **k) -> int:
        self.tensors.append(Tensor(*a, **k))
        return len(self.tensors) - 1

    def add_node(self, name, op, inputs, outputs, **params) -> int:
        self.nodes.append(Node(name, op, list(inputs), list(outputs), dict(params)))
        return len(self.nodes) - 1

    def add_const(self, name, array, dtype, scales=None, zero_points=None) -> int:
        """Const tensors are produced by a 'Const' node, as the converters emit them."""
        array = np.ascontiguousarray(array, dtype=NP_OF_DT[dtype])
        t = self.add_tensor(name, list(array.shape), dtype, TT_CONST, array, scales, zero_points)
        self.add_node(name, "Const", [], [t])
        return t

    def add_input(self, name, dims, dtype=DT_FP32, scales=None, zero_points=None) -> int:
        t = self.add_tensor(name, list(dims), dtype, TT_INPUT, None, scales, zero_points)
        n = self.add_node(name, "InputOp", [], [t])
        self.input_nodes.append(n)
        return t

    def producer(self, tidx) -> Optional[Node]:
        for n in self.nodes:
            if tidx in n.outputs:
                return n
        return None


# --------------------------------------------------------------------------------------
# operator parameter blobs (field order == TM2_*Param structs)
# --------------------------------------------------------------------------------------
def _pack_param(op: str, p: Dict) -> Optional[bytes]:
    if op == "Convolution":      # TM2_ConvParam tm2_format.h:419-435
        return struct.pack("<14i", p["kernel_h"], p["kernel_w"], p["stride_h"], p["stride_w"],
                           p.get("dilation_h", 1), p.get("dilation_w", 1), p["input_channel"],
                           p["output_channel"], p.get("group", 1), p.get("activation", -1),
                           p.get("pad_h0", 0), p.get("pad_w0", 0), p.get("pad_h1", 0), p.get("pad_w1", 0))
    if op == "Pooling":          # TM2_PoolParam tm2_format.h:510-524
        return struct.pack("<I10i", p["alg"], p["kernel_h"], p["kernel_w"], p["stride_h"], p["stride_w"],
                           p.get("global", 0), p.get("caffe_flavor", 0), p.get("pad_h0", 0),
                           p.get("pad_w0", 0), p.get("pad_h1", 0), p.get("pad_w1", 0))
    if op == "FullyConnected":   # TM2_FCParam :474-477
        return struct.pack("<i", p["num_output"])
    if op == "Eltwise":          # TM2_EltwiseParam :465-472
        return struct.pack("<Ii3f", p["type"], p.get("caffe_flavor", 1), p.get("shift", 0.0),
                           p.get("power", 1.0), p.get("scale", 1.0))
    if op == "ReLU":             # TM2_ReLuParam
        return struct.pack("<f", p.get("negative_slope", 0.0))
    if op == "Concat":           # TM2_ConcatParam
        return struct.pack("<i", p.get("axis", 1))
    if op == "Softmax":
        return struct.pack("<i", p.get("axis", 1))
    if op == "Flatten":          # TM2_FlattenParam {axis, end_axis}
        return struct.pack("<2i", p.get("axis", 1), p.get("end_axis", 3))
    if op == "Upsample":
        return struct.pack("<f", float(p.get("scale", 2)))
    if op == "Permute":          # TM2_PermuteParam {flag, order0..3}
        o = p["order"]
        return struct.pack("<5i", p.get("flag", 0), o[0], o[1], o[2], o[3])
    if op == "DetectionOutput":  # TM2_DetectionOutputParam tm2_format.h:455-463
        return struct.pack("<3i2f", p["num_classes"], p["keep_top_k"], p["nms_top_k"], p["confidence_threshold"], p["nms_threshold"])
    if op in ("Const", "InputOp", "Dropout", "ReLU6"):
        return None
    raise NotImplementedError("tm2 writer: op %s" % op)


def _unpack_param(op: str, b: bytes, off: int) -> Dict:
    if op == "Convolution":
        k = ["kernel_h", "kernel_w", "stride_h", "stride_w", "dilation_h", "dilation_w", "input_channel",
             "output_channel", "group", "activation", "pad_h0", "pad_w0", "pad_h1", "pad_w1"]
        return dict(zip(k, struct.unpack_from("<14i", b, off)))
    if op == "Pooling":
        k = ["alg", "kernel_h", "kernel_w", "stride_h", "stride_w", "global", "caffe_flavor", "pad_h0",
             "pad_w0", "pad_h1", "pad_w1"]
        return dict(zip(k, struct.unpack_from("<I10i", b, off)))
    if op == "FullyConnected":
        return {"num_output": struct.unpack_from("<i", b, off)[0]}
    if op == "Eltwise":
        k = ["type", "caffe_flavor", "shift", "power", "scale"]
        return dict(zip(k, struct.unpack_from("<Ii3f", b, off)))
    if op == "ReLU":
        return {"negative_slope": struct.unpack_from("<f", b, off)[0]}
    if op in ("Concat", "Softmax"):
        return {"axis": struct.unpack_from("<i", b, off)[0]}
    if op == "Flatten":
        return dict(zip(["axis", "end_axis"], struct.unpack_from("<2i", b, off)))
    if op == "Upsample":
        return {"scale": struct.unpack_from("<f", b, off)[0]}
    if op == "Permute":
        v = struct.unpack_from("<5i", b, off)
        return {"flag": v[0], "order": list(v[1:])}
    if op == "PriorBox":        # TM2_PriorBoxParam (tm2_format.h:526-542): four TM2_Vector_floats offsets, then scalars
        v = struct.unpack_from("<4I5i3f2i", b, off)

        def vf(o):
            n = struct.unpack_from("<I", b, o)[0]
            return list(struct.unpack_from("<%df" % n, b, o + 4)) if n else []

        return {"min_size": vf(v[0]), "max_size": vf(v[1]), "variance": vf(v[2]), "aspect_ratio": vf(v[3]), "flip": v[4],
                "clip": v[5], "img_size": v[6], "img_h": v[7], "img_w": v[8], "step_w": v[9], "step_h": v[10],
                "offset": v[11], "num_priors": v[12], "out_dim": v[13]}
    if op == "DetectionOutput":
        return dict(zip(["num_classes", "keep_top_k", "nms_top_k", "confidence_threshold", "nms_threshold"], struct.unpack_from("<3i2f", b, off)))
    if op == "Reshape":
        is_mx, rev, voff, is_onnx = struct.unpack_from("<iiIi", b, off)
        n = struct.unpack_from("<I", b, voff)[0] if voff else 0
        return {"is_mxnet": is_mx, "reverse": rev, "is_onnx": is_onnx,
                "re_shape": list(struct.unpack_from("<%di" % n, b, voff + 4)) if n else []}
    return {}


class _Blob:
    """Append-only byte arena; every object starts 4-byte aligned (const payloads 16-byte)."""

    def __init__(self):
        self.b = bytearray()

    def align(self, a):
        self.b.extend(b"\0" * ((-len(self.b)) % a))

    def put(self, data: bytes, align=4) -> int:
        self.align(align)
        off = len(self.b)
        self.b.extend(data)
        return off

    def string(self, s: str) -> int:
        raw = s.encode() + b"\0"
        d = self.put(raw)
        return self.put(struct.pack("<II", len(raw), d))        # TM2_String{size, offset_data}

    def vec_u32(self, vals) -> int:
        return self.put(struct.pack("<I%dI" % len(vals), len(vals), *vals))

    def vec_i32(self, vals) -> int:
        return self.put(struct.pack("<I%di" % len(vals), len(vals), *vals))

    def vec_f32(self, vals) -> int:
        return self.put(struct.pack("<I%df" % len(vals), len(vals), *vals))


def write_tm2(g: Graph) -> bytes:
    """Serialise `g` to tmfile v2 bytes loadable by the reference (`create_graph(ctx,"tengine",f)`)."""
    bl = _Blob()
    bl.put(b"\0" * 12)                                    # TM2_Header placeholder (10 bytes, padded)
    # buffers: one per const tensor (TM2_Buffer{size, offset_data})
    buf_offs, buf_id_of = [], {}
    for ti, t in enumerate(g.tensors):
        if t.ttype == TT_CONST:
            arr = np.ascontiguousarray(t.data, dtype=NP_OF_DT[t.dtype])
            d = bl.put(arr.tobytes(), align=16)
            buf_id_of[ti] = len(buf_offs)
            buf_offs.append(bl.put(struct.pack("<II", arr.nbytes, d)))
    # tensors
    ten_offs = []
    for ti, t in enumerate(g.tensors):
        dims = bl.vec_i32(t.dims) if t.dims else 0
        name = bl.string(t.name)
        q = 0
        if t.scales is not None:
            zps = t.zero_points if t.zero_points is not None else [0] * len(t.scales)
            qo = [bl.put(struct.pack("<ifi", int(z), float(s), 8)) for s, z in zip(t.scales, zps)]
            q = bl.vec_u32(qo)
        ten_offs.append(bl.put(struct.pack("<II3I3i", ti, buf_id_of.get(ti, 0), dims, name, q,
                                           LAYOUT_NCHW, t.ttype, t.dtype)))
    # nodes
    node_offs = []
    for ni, n in enumerate(g.nodes):
        if n.op == "Reshape":    # TM2_ReshapeParam {is_mxnet, reverse, offset_re_shape -> TM2_Vector_dims, is_onnx} (tm2_format.h:565-571)
            pb = struct.pack("<iiIi", n.params.get("is_mxnet", 0), n.params.get("reverse", 0),
                             bl.vec_i32(n.params["re_shape"]), n.params.get("is_onnx", 1))
        elif n.op == "PriorBox":  # TM2_PriorBoxParam; the loader dereferences all four vector offsets (tm2_priorbox.c:49-52): empty
            q = n.params        # vectors are written as {v_num = 0}
            pb = struct.pack("<4I5i3f2i", bl.vec_f32(q["min_size"]), bl.vec_f32(q.get("max_size", [])),
                             bl.vec_f32(q["variance"]), bl.vec_f32(q.get("aspect_ratio", [])), q.get("flip", 0),
                             q.get("clip", 0), q.get("img_size", 0), q.get("img_h", 0), q.get("img_w", 0),
                             q.get("step_w", 0.0), q.get("step_h", 0.0), q.get("offset", 0.5), q.get("num_priors", 0),
                             q.get("out_dim", 0))
        else:
            pb = _pack_param(n.op, n.params)
        po = bl.put(pb) if pb else 0
        op = bl.put(struct.pack("<III", 1, OPTYPE[n.op], po))               # TM2_Operator
        vi = bl.vec_u32(n.inputs) if n.inputs else 0
        vo = bl.vec_u32(n.outputs)
        nm = bl.string(n.name)
        node_offs.append(bl.put(struct.pack("<I5IB3x", ni, vi, vo, op, nm, 0, 0)))   # TM2_Node
    vin = bl.vec_u32(g.input_nodes)
    vout = bl.vec_u32(g.output_nodes)
    vnodes = bl.vec_u32(node_offs)
    vtens = bl.vec_u32(ten_offs)
    vbufs = bl.vec_u32(buf_offs)
    sname = bl.string(g.name)
    sub = bl.put(struct.pack("<Iii7I", 0, LAYOUT_NCHW, LAYOUT_NCHW, vin, vout, vnodes, vtens, vbufs, sname, 0))
    vsub = bl.vec_u32([sub])
    mname = bl.string(g.name)
    model = bl.put(struct.pack("<iiII", 1, 0, vsub, mname))                  # orig_format = TENGINE
    # TM2_Header {u16 ver_main=2, u16 ver_sub, u16 ver_compile, (pad), u32 offset_root}
    bl.b[0:12] = struct.pack("<HHHxxI", 2, 0, 0, model)
    return bytes(bl.b)


def read_tm2(b: bytes) -> Graph:
    """Parse tmfile bytes back into the IR (tests: round trip; tools: inspect reference models)."""
    u32 = lambda o: struct.unpack_from("<I", b, o)[0]

    def vec(o):
        n = u32(o)
        return list(struct.unpack_from("<%dI" % n, b, o + 4)) if n else []

    def string(o):
        if o == 0:
            return ""
        size, d = struct.unpack_from("<II", b, o)
        return b[d:d + size].split(b"\0")[0].decode()

    root = struct.unpack_from("<HHHxxI", b, 0)[3]
    _, _, vsub, _ = struct.unpack_from("<iiII", b, root)
    sub = vec(vsub)[0]
    (_, _, _, vin, vout, vnodes, vtens, vbufs, sname, _) = struct.unpack_from("<Iii7I", b, sub)
    g = Graph(name=string(sname))
    bufs = vec(vbufs)
    for to in vec(vtens):
        tid, bid, dims_o, name_o, q_o, _layout, ttype, dtype = struct.unpack_from("<II3I3i", b, to)
        dims = []
        if dims_o:
            n = u32(dims_o)
            dims = list(struct.unpack_from("<%di" % n, b, dims_o + 4))
        t = Tensor(string(name_o), dims, dtype, ttype)
        if q_o:
            qs = [struct.unpack_from("<ifi", b, o) for o in vec(q_o)]
            t.scales = [q[1] for q in qs]
            t.zero_points = [q[0] for q in qs]
        if ttype == TT_CONST:
            size, d = struct.unpack_from("<II", b, bufs[bid])
            cnt = int(np.prod(dims)) if dims else 0
            if d:
                t.data = np.frombuffer(b, dtype=NP_OF_DT[dtype], count=cnt, offset=d).reshape(dims).copy()
            else:           # structure-only benchmark model: loader zero-fills (tm2_serializer.c:240-246)
                t.data = np.zeros(dims, dtype=NP_OF_DT[dtype])
        g.tensors.append(t)
    for no in vec(vnodes):
        nid, vi, vo, op_o, nm, _attrs, _dyn = struct.unpack_from("<I5IB", b, no)
        _ver, optype, po = struct.unpack_from("<III", b, op_o)
        op = OPNAME.get(optype, "op%d" % optype)
        g.nodes.append(Node(string(nm), op, vec(vi) if vi else [], vec(vo),
                            _unpack_param(op, b, po) if po else {}))
    g.input_nodes = vec(vin)
    g.output_nodes = vec(vout)
    return g
