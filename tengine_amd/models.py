"""Synthetic config models + the quantiser restatement.

The reference ships structure-only `benchmark/models/*_benchmark.tmfile` (no weights)
and its quantisation tool needs OpenCV, so the quantised configs are synthesised:

  * topologies are written out here (node-for-node the same graphs the reference benchmarks:
    tests/test_models_structure.py diffs them against the reference files when those are present),
  * weights: fp32 N(0, sqrt(2/fan_in)), bias U(-0.1, 0.1), numpy PCG64 seeded 0x5EED0000+node index,
  * int8 quantisation mirrors tools/quantize/quant_save_graph.cpp:355-613 (min-max calibration):
    activations per-tensor symmetric (zp 0, scale = absmax/127), weights per-out-channel
    scale = max|w|/127 (:526-536), q = round(w/scale) clamp +-127 (:541-556), bias int32
    round(b/(in_scale*w_scale[c])) (:581-601); pass-through ops share scales (:440-470).

Calibration uses a plain fp32 forward in torch (CPU) -- it only decides the scales; it is not an
oracle and nothing in the product path depends on it.
"""
import json
import os

import numpy as np

from . import tm2
from .tm2 import DT_FP32, DT_INT8, DT_INT32, DT_UINT8, Graph

SEED0 = 0x5EED0000


# --------------------------------------------------------------------------------------
# fp32 graph construction helpers
# --------------------------------------------------------------------------------------
class _B:
    """fp32 graph builder with seeded synthetic weights."""

    def __init__(self, name, in_dims):
        self.g = Graph(name=name)
        self.cur = self.g.add_input("data", in_dims, DT_FP32)

    def _rng(self):
        return np.random.default_rng(SEED0 + len(self.g.nodes))

    def dims(self, t):
        return self.g.tensors[t].dims

    def conv(self, name, x, cout, k, s=1, p=0, group=1, act=-1, bias=True, dil=1):
        cin = self.dims(x)[1]
        rng = self._rng()
        fan_in = (cin // group) * k * k
        w = rng.normal(0.0, np.sqrt(2.0 / fan_in), size=(cout, cin // group, k, k)).astype(np.float32)
        ins = [x, self.g.add_const(name + "/weight", w, DT_FP32)]
        if bias:
            b = rng.uniform(-0.1, 0.1, size=(cout,)).astype(np.float32)
            ins.append(self.g.add_const(name + "/bias", b, DT_FP32))
        n, _, h, wd = self.dims(x)
        oh = (h - dil * (k - 1) - 1 + 2 * p) // s + 1
        ow = (wd - dil * (k - 1) - 1 + 2 * p) // s + 1
        y = self.g.add_tensor(name + "/0", [n, cout, oh, ow], DT_FP32)
        self.g.add_node(name, "Convolution", ins, [y], kernel_h=k, kernel_w=k, stride_h=s, stride_w=s,
                        dilation_h=dil, dilation_w=dil, input_channel=cin, output_channel=cout, group=group,
                        activation=act, pad_h0=p, pad_w0=p, pad_h1=p, pad_w1=p)
        return y

    def pool(self, name, x, alg, k, s, p=0, glob=0, caffe=0):
        n, c, h, w = self.dims(x)
        if glob:
            oh = ow = 1
        else:
            oh, _, _ = pool_out(h, k, s, p, caffe)
            ow, _, _ = pool_out(w, k, s, p, caffe)
        y = self.g.add_tensor(name + "/0", [n, c, oh, ow], DT_FP32)
        self.g.add_node(name, "Pooling", [x], [y], alg=alg, kernel_h=k, kernel_w=k, stride_h=s, stride_w=s,
                        **{"global": glob}, caffe_flavor=caffe, pad_h0=p, pad_w0=p, pad_h1=p, pad_w1=p)
        return y

    def fc(self, name, x, nout):
        d = self.dims(x)
        hidden = int(np.prod(d[1:]))
        rng = self._rng()
        w = rng.normal(0.0, np.sqrt(2.0 / hidden), size=(nout, hidden)).astype(np.float32)
        b = rng.uniform(-0.1, 0.1, size=(nout,)).astype(np.float32)
        ins = [x, self.g.add_const(name + "/weight", w, DT_FP32), self.g.add_const(name + "/bias", b, DT_FP32)]
        y = self.g.add_tensor(name + "/0", [d[0], nout], DT_FP32)
        self.g.add_node(name, "FullyConnected", ins, [y], num_output=nout)
        return y

    def relu(self, name, x, slope=0.0):
        y = self.g.add_tensor(name + "/0", list(self.dims(x)), DT_FP32)
        self.g.add_node(name, "ReLU", [x], [y], negative_slope=slope)
        return y

    def eltwise_sum(self, name, a, b):
        y = self.g.add_tensor(name + "/0", list(self.dims(a)), DT_FP32)
        self.g.add_node(name, "Eltwise", [a, b], [y], type=tm2.ELT_SUM, caffe_flavor=1)
        return y

    def concat(self, name, xs, axis=1):
        d = list(self.dims(xs[0]))
        d[axis] = sum(self.dims(x)[axis] for x in xs)
        y = self.g.add_tensor(name + "/0", d, DT_FP32)
        self.g.add_node(name, "Concat", xs, [y], axis=axis)
        return y

    def dropout(self, name, x):
        y = self.g.add_tensor(name + "/0", list(self.dims(x)), DT_FP32)
        self.g.add_node(name, "Dropout", [x], [y])
        return y

    def softmax(self, name, x, axis=1):
        y = self.g.add_tensor(name + "/0", list(self.dims(x)), DT_FP32)
        self.g.add_node(name, "Softmax", [x], [y], axis=axis)
        return y

    def upsample(self, name, x, scale=2):
        n, c, h, w = self.dims(x)
        y = self.g.add_tensor(name + "/0", [n, c, h * scale, w * scale], DT_FP32)
        self.g.add_node(name, "Upsample", [x], [y], scale=float(scale))
        return y

    def permute(self, name, x, order=(0, 2, 3, 1)):
        d = self.dims(x)
        y = self.g.add_tensor(name + "/0", [d[i] for i in order], DT_FP32)
        self.g.add_node(name, "Permute", [x], [y], flag=0, order=list(order))
        return y

    def flatten(self, name, x):
        d = self.dims(x)
        y = self.g.add_tensor(name + "/0", [d[0], int(np.prod(d[1:]))], DT_FP32)
        self.g.add_node(name, "Flatten", [x], [y], axis=1, end_axis=3)
        return y

    def reshape(self, name, x, re_shape):
        """reshape.c infer_shape with is_onnx set: 0 copies the input dim, -1 is inferred"""
        d = self.dims(x)
        out = [d[i] if v == 0 else v for i, v in enumerate(re_shape)]
        if -1 in out:
            out[out.index(-1)] = int(np.prod(d)) // int(-np.prod(out))
        y = self.g.add_tensor(name + "/0", out, DT_FP32)
        self.g.add_node(name, "Reshape", [x], [y], is_mxnet=0, reverse=0, is_onnx=1, re_shape=list(re_shape))
        return y

    def priorbox(self, name, feat, data, min_size, max_size, aspect_ratio, variance=(0.1, 0.1, 0.2, 0.2), flip=1, clip=0,
                 offset=0.5):
        """priorbox.c:33-75: [n, 2, feat_h * feat_w * num_priors * 4, 1]"""
        d = self.dims(feat)
        num = (len(aspect_ratio) * (2 if flip else 1) + 1 + (1 if max_size else 0)) * len(min_size)
        out_dim = d[2] * d[3] * num * 4
        y = self.g.add_tensor(name + "/0", [d[0], 2, out_dim, 1], DT_FP32)
        self.g.add_node(name, "PriorBox", [feat, data], [y], min_size=[float(v) for v in min_size],
                        max_size=[float(v) for v in max_size], aspect_ratio=[float(v) for v in aspect_ratio],
                        variance=[float(v) for v in variance], flip=flip, clip=clip, offset=offset, num_priors=num,
                        out_dim=out_dim)
        return y

    def finish(self, outs):
        for o in outs:
            for ni, n in enumerate(self.g.nodes):
                if o in n.outputs:
                    self.g.output_nodes.append(ni)
        return self.g


def _cdiv(a, b):
    """C integer division (truncates toward zero)."""
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b > 0) else -q


def pool_out(inp, k, s, pad, caffe):
    """Output size + real pads (source/operator/prototype/pooling_param.h:59-105)."""
    if pad >= 0:
        if caffe == 1:
            out = 2 + _cdiv(inp - k + 2 * pad - 1, s)
            if pad > 0 and (out - 1) * s >= inp + pad:
                out -= 1
        elif caffe == 2:
            out = 1 + _cdiv(inp - k + pad, s)
        else:
            out = 1 + _cdiv(inp - k + 2 * pad, s)
    else:
        out = 1 + _cdiv(inp - 1, s)
    if pad >= 0 and caffe == 2:      # darknet: the file's pad is the TOTAL, split low / high (pooling.c:81-88)
        return out, pad // 2, pad - pad // 2
    total = (out - 1) * s + k
    pad_num = max(total - inp, 0)
    p0 = pad_num // 2 if pad < 0 else pad
    p1 = pad_num - pad
    return out, p0, p1


# --------------------------------------------------------------------------------------
# topologies
# --------------------------------------------------------------------------------------
def mobilenet_v1_fp32(batch=1, res=224, classes=1000):
    """MobileNet-v1 1.0 (benchmark/models/mobilenet_benchmark.tmfile: 28 conv + global avgpool;
    BN/scale/ReLU folded into the convs, classifier `fc7` is a 1x1 conv)."""
    b = _B("mobilenet_v1", [batch, 3, res, res])
    x = b.conv("conv1", b.cur, 32, 3, 2, 1, act=0)
    cfg = [("2_1", 64, 1), ("2_2", 128, 2), ("3_1", 128, 1), ("3_2", 256, 2), ("4_1", 256, 1), ("4_2", 512, 2),
           ("5_1", 512, 1), ("5_2", 512, 1), ("5_3", 512, 1), ("5_4", 512, 1), ("5_5", 512, 1),
           ("5_6", 1024, 2), ("6", 1024, 1)]
    for tag, cout, s in cfg:
        cin = b.dims(x)[1]
        x = b.conv("conv%s/dw" % tag, x, cin, 3, s, 1, group=cin, act=0)
        x = b.conv("conv%s/sep" % tag, x, cout, 1, 1, 0, act=0)
    x = b.pool("pool6", x, tm2.POOL_AVG, 2, 1, 0, glob=1, caffe=1)
    x = b.conv("fc7", x, classes, 1, 1, 0, act=-1)
    return b.finish([x])


def resnet50_fp32(batch=1, res=224, classes=1000):
    """ResNet-50 v1 caffe style (benchmark/models/resnet50_benchmark.tmfile: 53 conv, 16 eltwise,
    16 standalone relu after the adds, max+avg pool, fc, softmax). Stride-2 sits on the first 1x1."""
    b = _B("resnet50", [batch, 3, res, res])
    x = b.conv("conv1", b.cur, 64, 7, 2, 3, act=0)
    x = b.pool("pool1", x, tm2.POOL_MAX, 3, 2, 0, caffe=1)
    stages = [(2, 64, 256, 3, 1), (3, 128, 512, 4, 2), (4, 256, 1024, 6, 2), (5, 512, 2048, 3, 2)]
    for st, mid, out, reps, stride in stages:
        for r in range(reps):
            tag = "res%d%s" % (st, "abcdef"[r])
            s = stride if r == 0 else 1
            y = b.conv(tag + "_branch2a", x, mid, 1, s, 0, act=0)
            y = b.conv(tag + "_branch2b", y, mid, 3, 1, 1, act=0)
            y = b.conv(tag + "_branch2c", y, out, 1, 1, 0, act=-1)
            sc = b.conv(tag + "_branch1", x, out, 1, s, 0, act=-1) if r == 0 else x
            x = b.eltwise_sum(tag, sc, y)
            x = b.relu(tag + "_relu", x)
    x = b.pool("pool5", x, tm2.POOL_AVG, 7, 1, 0, glob=0, caffe=1)
    x = b.fc("fc1000", x, classes)
    x = b.softmax("prob", x)
    return b.finish([x])


def squeezenet_v11_fp32(batch=1, res=227, classes=1000):
    """SqueezeNet v1.1 (benchmark/models/squeezenet_v1.1_benchmark.tmfile)."""
    b = _B("squeezenet_v1.1", [batch, 3, res, res])
    x = b.conv("conv1", b.cur, 64, 3, 2, 0, act=0)
    x = b.pool("pool1", x, tm2.POOL_MAX, 3, 2, 0, caffe=1)

    def fire(tag, x, sq, ex):
        s = b.conv(tag + "/squeeze1x1", x, sq, 1, act=0)
        e3 = b.conv(tag + "/expand3x3", s, ex, 3, 1, 1, act=0)      # node order as in the reference file
        e1 = b.conv(tag + "/expand1x1", s, ex, 1, act=0)
        return b.concat(tag + "/concat", [e1, e3])

    x = fire("fire2", x, 16, 64)
    x = fire("fire3", x, 16, 64)
    x = b.pool("pool3", x, tm2.POOL_MAX, 3, 2, 0, caffe=1)
    x = fire("fire4", x, 32, 128)
    x = fire("fire5", x, 32, 128)
    x = b.pool("pool5", x, tm2.POOL_MAX, 3, 2, 0, caffe=1)
    x = fire("fire6", x, 48, 192)
    x = fire("fire7", x, 48, 192)
    x = fire("fire8", x, 64, 256)
    x = fire("fire9", x, 64, 256)
    x = b.dropout("drop9", x)
    x = b.conv("conv10", x, classes, 1, act=0)
    x = b.pool("pool10", x, tm2.POOL_AVG, 2, 1, 0, glob=1, caffe=1)
    x = b.softmax("prob", x)
    return b.finish([x])


def yolov3_tiny_fp32(batch=1, res=416, nout=255):
    """YOLOv3-tiny, node for node benchmark/models/yolov3_tiny_benchmark.tmfile (35 compute nodes): 13 conv, leaky ReLUs kept as
    separate nodes, 6 maxpool in the darknet flavour (caffe_flavor 2, total pad 1: the last one is the stride-1 'same' pool), the
    single-input route (Concat) in front of the second head's 1x1, upsample + route, two 1x1 heads each behind the Dropout node the
    converter leaves for a yolo layer."""
    b = _B("yolov3_tiny", [batch, 3, res, res])
    x = b.cur

    def cbl(i, x, cout, k):
        y = b.conv("conv%d" % i, x, cout, k, 1, k // 2, act=-1)
        return b.relu("leaky%d" % i, y, 0.1)

    chans = [16, 32, 64, 128, 256, 512]
    route8 = None
    for i, c in enumerate(chans):
        x = cbl(i, x, c, 3)
        if i == 4:
            route8 = x
        x = b.pool("maxpool%d" % i, x, tm2.POOL_MAX, 2, 2 if i < 5 else 1, 1, caffe=2)     # darknet: pad = size - 1, split 0 | 1
    x = cbl(6, x, 1024, 3)
    x13 = cbl(7, x, 256, 1)
    y = cbl(8, x13, 512, 3)
    head1 = b.dropout("yolo1", b.conv("conv9", y, nout, 1, act=-1))
    z = cbl(10, b.concat("route0", [x13]), 128, 1)
    z = b.upsample("upsample", z, 2)
    z = b.concat("route", [z, route8])
    z = cbl(11, z, 256, 3)
    head2 = b.dropout("yolo2", b.conv("conv12", z, nout, 1, act=-1))
    return b.finish([head1, head2])


def mssd_fp32(batch=1, res=300, classes=21, tail=False, priorbox=False, detection=False):
    """MobileNet-v1-SSD 300x300 (benchmark/models/mssd_benchmark.tmfile, the BASELINE "MobileNet-SSD" stand-in,
    SURVEY §8d): 47 convs = conv0 + 13 (dw, pw) pairs + 4 (1x1, 3x3 s2) extra pairs + 6 loc and 6 conf 1x1 heads on
    conv11 (19x19, 3 priors), conv13 (10x10), conv14_2 (5x5), conv15_2 (3x3), conv16_2 (2x2), conv17_2 (1x1) (6 priors
    each); every head goes Permute(0,2,3,1) -> Flatten -> Concat(axis 1).  Outputs: mbox_loc [N, 1917*4] and
    mbox_conf [N, 1917*classes]; the Reshape/Softmax/PriorBox/DetectionOutput tail of the tmfile is host-side
    post-processing the splitter leaves on the CPU device.  `tail=True` appends the quantised part of that tail --
    Reshape(0,-1,classes) -> Softmax(axis 2) -> Flatten on mbox_conf -- for oracle-vs-reference tests of the next
    row; the device graphs are built without it.  `priorbox=True` adds the six PriorBox nodes of the
    MobileNet-SSD deploy prototxt (min / max sizes 60 | 105,150 | 150,195 | 195,240 | 240,285 | 285,300, aspect ratios
    2 | 2,3, flip, no clip, variances .1 .1 .2 .2, offset 0.5) and their Concat(axis 2) -> mbox_priorbox [1, 2, 7668, 1]:
    with both, the graph ends exactly at detection_output's three inputs; `detection=True` (with both) appends that node -- the
    whole benchmark file, 84 compute nodes (the HIP device leaves DetectionOutput to the CPU device)."""
    b = _B("mssd", [batch, 3, res, res])
    data = b.cur
    x = b.conv("conv0", b.cur, 32, 3, 2, 1, act=0)
    cfg = [(64, 1), (128, 2), (128, 1), (256, 2), (256, 1), (512, 2), (512, 1), (512, 1), (512, 1), (512, 1), (512, 1),
           (1024, 2), (1024, 1)]
    feats = []
    for i, (c, s) in enumerate(cfg):
        cin = b.dims(x)[1]
        x = b.conv("conv%d/dw" % (i + 1), x, cin, 3, s, 1, group=cin, act=0)
        x = b.conv("conv%d" % (i + 1), x, c, 1, act=0)
        if i + 1 in (11, 13):
            feats.append(("conv%d" % (i + 1), x, 3 if i + 1 == 11 else 6))
    for i, (c1, c2) in enumerate([(256, 512), (128, 256), (128, 256), (64, 128)]):
        x = b.conv("conv%d_1" % (14 + i), x, c1, 1, act=0)
        x = b.conv("conv%d_2" % (14 + i), x, c2, 3, 2, 1, act=0)
        feats.append(("conv%d_2" % (14 + i), x, 6))
    outs = []
    for kind, per in (("loc", 4), ("conf", classes)):
        flats = []
        for name, f, priors in feats:
            h = b.conv("%s_mbox_%s" % (name, kind), f, priors * per, 1, act=-1)
            h = b.permute("%s_mbox_%s_perm" % (name, kind), h)
            flats.append(b.flatten("%s_mbox_%s_flat" % (name, kind), h))
        outs.append(b.concat("mbox_%s" % kind, flats, axis=1))
    if tail:
        y = b.reshape("mbox_conf_reshape", outs[1], [0, -1, classes])
        y = b.softmax("mbox_conf_softmax", y, axis=2)
        outs[1] = b.flatten("mbox_conf_flatten", y)
    if priorbox:
        sizes = [(60, None), (105, 150), (150, 195), (195, 240), (240, 285), (285, 300)]
        pbs = [b.priorbox("%s_mbox_priorbox" % name, f, data, [mn], [mx] if mx else [], [2.0] if npri == 3 else [2.0, 3.0])
               for (name, f, npri), (mn, mx) in zip(feats, sizes)]
        outs.append(b.concat("mbox_priorbox", pbs, axis=2))
    if detection:
        assert tail and priorbox
        y = b.g.add_tensor("detection_out/0", [batch, 100, 6, 1], DT_FP32)         # detection_output.c infer_shape: keep_top_k rows of 6
        b.g.add_node("detection_out", "DetectionOutput", outs, [y], num_classes=classes, keep_top_k=100, nms_top_k=100,
                     confidence_threshold=0.25, nms_threshold=0.45)
        outs = [y]
    return b.finish(outs)


# --------------------------------------------------------------------------------------
# fp32 forward (calibration only)
# --------------------------------------------------------------------------------------
def fp32_forward(g: Graph, x: np.ndarray):
    """Plain fp32 forward of the IR; returns {tensor idx: ndarray}. Calibration aid, not an oracle."""
    import torch
    import torch.nn.functional as F

    vals = {}
    for ti, t in enumerate(g.tensors):
        if t.ttype == tm2.TT_CONST:
            vals[ti] = torch.from_numpy(np.asarray(t.data, dtype=np.float32))
    for n in g.nodes:
        op, p = n.op, n.params
        if op == "Const":
            continue
        if op == "InputOp":
            vals[n.outputs[0]] = torch.from_numpy(x.astype(np.float32))
            continue
        a = vals[n.inputs[0]]
        if op == "Convolution":
            w = vals[n.inputs[1]]
            bias = vals[n.inputs[2]] if len(n.inputs) > 2 else None
            a = F.pad(a, (p["pad_w0"], p["pad_w1"], p["pad_h0"], p["pad_h1"]))
            y = F.conv2d(a, w, bias, stride=(p["stride_h"], p["stride_w"]),
                         dilation=(p["dilation_h"], p["dilation_w"]), groups=p["group"])
            act = p["activation"]
            if act == 0:
                y = torch.relu(y)
            elif act > 0:
                y = torch.clamp(y, 0, 6)
        elif op == "Pooling":
            if p["global"]:
                y = a.mean(dim=(2, 3), keepdim=True) if p["alg"] == 1 else a.amax(dim=(2, 3), keepdim=True)
            else:
                od = g.tensors[n.outputs[0]].dims
                k, s = p["kernel_h"], p["stride_h"]
                _, p0h, p1h = pool_out(a.shape[2], k, s, p["pad_h0"], p["caffe_flavor"])
                _, p0w, p1w = pool_out(a.shape[3], k, s, p["pad_w0"], p["caffe_flavor"])
                if p["alg"] == 0:
                    ap = F.pad(a, (p0w, max(p1w, 0) + k, p0h, max(p1h, 0) + k), value=float("-inf"))
                    y = F.max_pool2d(ap, k, s)[:, :, :od[2], :od[3]]
                else:
                    ap = F.pad(a, (p0w, max(p1w, 0) + k, p0h, max(p1h, 0) + k))
                    ones = F.pad(torch.ones_like(a), (p0w, max(p1w, 0) + k, p0h, max(p1h, 0) + k))
                    y = (F.avg_pool2d(ap, k, s) / F.avg_pool2d(ones, k, s).clamp_min(1e-9))[:, :, :od[2], :od[3]]
        elif op == "FullyConnected":
            w = vals[n.inputs[1]]
            y = a.reshape(a.shape[0], -1) @ w.t()
            if len(n.inputs) > 2:
                y = y + vals[n.inputs[2]]
        elif op == "ReLU":
            y = torch.where(a < 0, a * p.get("negative_slope", 0.0), a)
        elif op == "ReLU6":
            y = torch.clamp(a, 0, 6)
        elif op == "Eltwise":
            bb = vals[n.inputs[1]]
            y = {tm2.ELT_SUM: a + bb, tm2.ELT_PROD: a * bb, tm2.ELT_MAX: torch.maximum(a, bb),
                 tm2.ELT_SUB: a - bb}[p["type"]]
        elif op == "Concat":
            y = torch.cat([vals[i] for i in n.inputs], dim=p.get("axis", 1))
        elif op == "Dropout":
            y = a
        elif op == "Softmax":
            y = torch.softmax(a, dim=p.get("axis", 1))
        elif op == "Upsample":
            s = int(p.get("scale", 2))
            y = a.repeat_interleave(s, dim=2).repeat_interleave(s, dim=3)
        elif op == "Flatten":
            y = a.reshape(a.shape[0], -1)
        elif op == "Permute":
            y = a.permute(*p["order"]).contiguous()
        elif op == "Reshape":
            y = a.reshape(g.tensors[n.outputs[0]].dims)
        elif op == "PriorBox":      # value range only (the table assigns these tensors an a-priori range, see calib_table)
            y = torch.zeros(g.tensors[n.outputs[0]].dims)
            y[:, 0] = 0.5
            y[:, 1] = 0.1
        elif op == "DetectionOutput":      # host-side post-processing: nothing to calibrate behind it
            y = torch.zeros(g.tensors[n.outputs[0]].dims)
        else:
            raise NotImplementedError(op)
        vals[n.outputs[0]] = y
    return {k: v.numpy() for k, v in vals.items()}


# --------------------------------------------------------------------------------------
# quantiser restatement
# --------------------------------------------------------------------------------------
PASS_THROUGH = ("Dropout", "Flatten", "Reshape", "Permute")


def synth_input(g: Graph, seed=1234, dtype=DT_INT8):
    """Seeded uniform input over the full quantised range (SURVEY §8d 'Inputs')."""
    dims = g.tensors[g.nodes[g.input_nodes[0]].outputs[0]].dims
    rng = np.random.default_rng(seed)
    if dtype == DT_INT8:
        return rng.integers(-127, 128, size=dims, dtype=np.int64).astype(np.int8)
    if dtype == DT_UINT8:
        return rng.integers(0, 256, size=dims, dtype=np.int64).astype(np.uint8)
    return rng.uniform(-1, 1, size=dims).astype(np.float32)


def calibrate_absmax(gf: Graph, calib_q: np.ndarray = None, in_scale=1.0 / 127.0):
    """min-max calibration pass: {tensor name: absmax} of one fp32 forward on the seeded input."""
    if calib_q is None:
        calib_q = synth_input(gf, 1234, DT_INT8)
    acts = fp32_forward(gf, calib_q.astype(np.float32) * np.float32(in_scale))
    return {t.name: float(np.abs(acts[ti]).max()) for ti, t in enumerate(gf.tensors)
            if t.ttype == tm2.TT_VAR}


def quantize_int8(gf: Graph, calib_q: np.ndarray = None, in_scale=1.0 / 127.0, table=None) -> Graph:
    """fp32 IR -> int8 IR following quant_save_graph.cpp:355-613 (see module docstring).
    `table` = {tensor name: absmax} from a previous calibration (the quant tool's .table file); with it
    the result is bit-reproducible on any host (no fp32 forward is run)."""
    if table is None:
        table = calibrate_absmax(gf, calib_q, in_scale)
    g = Graph(name=gf.name + "_int8")
    g.input_nodes, g.output_nodes = list(gf.input_nodes), list(gf.output_nodes)
    # activation scales
    scale = {}
    for ti, t in enumerate(gf.tensors):
        if t.ttype == tm2.TT_CONST:
            continue
        if t.ttype == tm2.TT_INPUT:
            scale[ti] = np.float32(in_scale)
        else:
            scale[ti] = np.float32(max(float(table[t.name]), 1e-6) / 127.0)
    # pass-through ops share the producer's scale; relu (slope 0) and max-pool too (:440-470)
    for n in gf.nodes:
        if n.op in PASS_THROUGH or (n.op == "ReLU" and n.params.get("negative_slope", 0.0) == 0.0) \
                or (n.op == "Pooling" and n.params["alg"] == 0):
            scale[n.outputs[0]] = scale[n.inputs[0]]
    # concat inputs share the output scale (:440-470): propagate backwards to the producers
    for n in reversed(gf.nodes):
        if n.op == "Concat":
            for i in n.inputs:
                scale[i] = scale[n.outputs[0]]
    for n in gf.nodes:      # re-share after concat adjustment
        if n.op in PASS_THROUGH or (n.op == "ReLU" and n.params.get("negative_slope", 0.0) == 0.0) \
                or (n.op == "Pooling" and n.params["alg"] == 0):
            scale[n.inputs[0]] = scale[n.outputs[0]]
    wq = {}
    for n in gf.nodes:
        if n.op in ("Convolution", "FullyConnected"):
            wt = gf.tensors[n.inputs[1]]
            w = np.asarray(wt.data, dtype=np.float32)
            w2 = w.reshape(w.shape[0], -1)
            ws = (np.abs(w2).max(axis=1) / np.float32(127.0)).astype(np.float32)
            ws = np.where(ws == 0, np.float32(1e-8), ws).astype(np.float32)
            q = np.clip(np.round(w2 / ws[:, None]), -127, 127).astype(np.int8).reshape(w.shape)
            wq[n.inputs[1]] = (q, ws)
            if len(n.inputs) > 2:
                bf = np.asarray(gf.tensors[n.inputs[2]].data, dtype=np.float32)
                bs = (scale[n.inputs[0]] * ws).astype(np.float32)
                bq = np.round(bf / bs).astype(np.int64).clip(-2 ** 31 + 1, 2 ** 31 - 1).astype(np.int32)
                wq[n.inputs[2]] = (bq, bs)
    for ti, t in enumerate(gf.tensors):
        if t.ttype == tm2.TT_CONST:
            if ti in wq:
                q, s = wq[ti]
                dt = DT_INT32 if q.dtype == np.int32 else DT_INT8
                g.tensors.append(tm2.Tensor(t.name, list(t.dims), dt, tm2.TT_CONST, q,
                                            [float(v) for v in s], [0] * len(s)))
            else:
                g.tensors.append(tm2.Tensor(t.name, list(t.dims), t.dtype, t.ttype, t.data))
        else:
            g.tensors.append(tm2.Tensor(t.name, list(t.dims), DT_INT8, t.ttype, None, [float(scale[ti])], [0]))
    for n in gf.nodes:
        g.nodes.append(tm2.Node(n.name, n.op, list(n.inputs), list(n.outputs), dict(n.params)))
    return g


U8_IN_SCALE, U8_IN_ZP = 2.0 / 255.0, 127      # synthetic uint8 input covers [-1, 1)


def calibrate_minmax(gf: Graph, calib_q: np.ndarray = None):
    """min-max calibration for the uint8 flow: {tensor name: [min, max]} of one fp32 forward."""
    if calib_q is None:
        calib_q = synth_input(gf, 1234, DT_UINT8)
    xin = (calib_q.astype(np.float32) - np.float32(U8_IN_ZP)) * np.float32(U8_IN_SCALE)
    acts = fp32_forward(gf, xin)
    return {t.name: [float(acts[ti].min()), float(acts[ti].max())] for ti, t in enumerate(gf.tensors)
            if t.ttype == tm2.TT_VAR}


def _u8_qparams(lo, hi):
    """tools/quantize/quant_tool_uint8.cpp:405-428 (min-max): scale, zero point of an activation."""
    lo, hi = np.float32(lo), np.float32(hi)
    if hi < 0:
        s = (np.float32(0) - lo) / np.float32(255)
        return s, int(-lo / s)
    if lo > 0:
        return hi / np.float32(255), 0
    s = (hi - lo) / np.float32(255)
    if s == 0:
        return np.float32(1e-6), 0
    return s, int(-lo / s)


def quantize_uint8(gf: Graph, table=None) -> Graph:
    """fp32 IR -> per-tensor asymmetric uint8 IR following save_graph_u8_perlayer
    (tools/quantize/quant_save_graph.cpp:82-353): activations scale=(max-min)/255, zp=int(-min/scale);
    relu(slope 0) / max-pool / reshape-like outputs hand their (scale, zp) back to a single-consumer input
    (:136-202); weights per TENSOR scale=(max-min)/255, zp=int(-min/scale), q=round(w/scale+zp) clip [0,255]
    (:228-275); bias int32 = roundf(b/(in_s*w_s)) (:277-304)."""
    if table is None:
        table = calibrate_minmax(gf)
    g = Graph(name=gf.name + "_uint8")
    g.input_nodes, g.output_nodes = list(gf.input_nodes), list(gf.output_nodes)
    qp = {}
    for ti, t in enumerate(gf.tensors):
        if t.ttype == tm2.TT_CONST:
            continue
        if t.ttype == tm2.TT_INPUT:
            qp[ti] = (np.float32(U8_IN_SCALE), U8_IN_ZP)
        else:
            qp[ti] = _u8_qparams(*table[t.name])
    used = {}
    for n in gf.nodes:
        for i in n.inputs:
            used[i] = used.get(i, 0) + 1
    for n in reversed(gf.nodes):
        share = n.op in PASS_THROUGH or (n.op == "ReLU" and n.params.get("negative_slope", 0.0) == 0.0) \
            or (n.op == "Pooling" and n.params["alg"] == 0)
        if share and gf.tensors[n.inputs[0]].ttype == tm2.TT_VAR and used.get(n.inputs[0], 0) == 1:
            qp[n.inputs[0]] = qp[n.outputs[0]]
    for n in gf.nodes:          # run-time pass-through ops never requantise: output carries the input's params
        if n.op in PASS_THROUGH:
            qp[n.outputs[0]] = qp[n.inputs[0]]
    wq = {}
    for n in gf.nodes:
        if n.op in ("Convolution", "FullyConnected"):
            w = np.asarray(gf.tensors[n.inputs[1]].data, dtype=np.float32)
            wmax, wmin = np.float32(w.max()), np.float32(w.min())
            ws = np.float32((wmax - wmin) / np.float32(255))
            if ws == 0:
                ws = np.float32(1e-8)
            wz = int(-wmin / ws)
            q = np.clip(np.round(w / ws + np.float32(wz)), 0, 255).astype(np.uint8)
            wq[n.inputs[1]] = (q, ws, wz)
            if len(n.inputs) > 2:
                bf = np.asarray(gf.tensors[n.inputs[2]].data, dtype=np.float32)
                bs = np.float32(qp[n.inputs[0]][0] * ws)
                bq = np.round(bf / bs).astype(np.int64).clip(-2 ** 31 + 1, 2 ** 31 - 1).astype(np.int32)
                wq[n.inputs[2]] = (bq, bs, 0)
    for ti, t in enumerate(gf.tensors):
        if t.ttype == tm2.TT_CONST:
            if ti in wq:
                q, s, z = wq[ti]
                dt = DT_INT32 if q.dtype == np.int32 else DT_UINT8
                g.tensors.append(tm2.Tensor(t.name, list(t.dims), dt, tm2.TT_CONST, q, [float(s)], [int(z)]))
            else:
                g.tensors.append(tm2.Tensor(t.name, list(t.dims), t.dtype, t.ttype, t.data))
        else:
            s, z = qp[ti]
            g.tensors.append(tm2.Tensor(t.name, list(t.dims), DT_UINT8, t.ttype, None, [float(s)], [int(z)]))
    for n in gf.nodes:
        g.nodes.append(tm2.Node(n.name, n.op, list(n.inputs), list(n.outputs), dict(n.params)))
        if n.op == "DetectionOutput":       # its rows (label, score, box) stay fp32: detection_output_ref.c:329-349 writes them as they are
            t = g.tensors[n.outputs[0]]
            t.dtype, t.scales, t.zero_points = DT_FP32, None, None
    return g


def set_batch(g: Graph, batch: int) -> Graph:
    """Re-shape every var/input tensor to a new batch (== set_tensor_shape + infer_shape)."""
    for t in g.tensors:
        if t.ttype != tm2.TT_CONST and t.dims:
            t.dims = [batch] + list(t.dims[1:])
    return g


BUILDERS = {
    "mobilenet_v1": mobilenet_v1_fp32,
    "resnet50": resnet50_fp32,
    "squeezenet_v1.1": squeezenet_v11_fp32,
    "yolov3_tiny": yolov3_tiny_fp32,
    "mssd": mssd_fp32,
}


CALIB_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "calib")


def calib_table(name, gf=None, write=False, dtype="int8"):
    """Committed calibration tables (tengine_amd/calib/<model>_<dtype>.json == the quant tool's scale
    table): make the synthetic quantised models bit-identical on every host."""
    path = os.path.join(CALIB_DIR, "%s_%s.json" % (name, dtype))
    if os.path.exists(path) and not write:
        table = json.load(open(path))
        if gf is not None and dtype == "uint8":      # optional tail tensors (mssd tail=True): ranges known a priori
            for t in gf.tensors:
                if t.ttype == tm2.TT_VAR and t.name not in table:
                    # prior boxes: corners of boxes centred inside the image, up to ~0.8 image sizes wide; variances 0.1 / 0.2
                    table[t.name] = table["mbox_conf/0"] if t.name == "mbox_conf_reshape/0" else \
                        ([-0.5, 1.5] if "priorbox" in t.name else [0.0, 1.0])
        return table
    gf = gf if gf is not None else BUILDERS[name]()
    table = calibrate_absmax(gf) if dtype == "int8" else calibrate_minmax(gf)
    if write:
        os.makedirs(CALIB_DIR, exist_ok=True)
        json.dump(table, open(path, "w"), indent=0, sort_keys=True)
    return table


def strip_tail(g: Graph, ops=("Softmax",)) -> Graph:
    """Cut trailing nodes of the given types (what Tengine's splitter hands to the CPU device,
    e.g. the Softmax after ResNet's fc1000): the graph output becomes their input."""
    changed = True
    while changed:
        changed = False
        for k, ni in enumerate(list(g.output_nodes)):
            n = g.nodes[ni]
            if n.op in ops:
                src = n.inputs[0]
                prod = [i for i, m in enumerate(g.nodes) if src in m.outputs][0]
                g.output_nodes[k] = prod
                g.nodes.pop(ni)
                g.input_nodes = [i - (i > ni) for i in g.input_nodes]
                g.output_nodes = [i - (i > ni) for i in g.output_nodes]
                changed = True
                break
    return g


def build(name, dtype="int8", batch=1, device_only=False, **kw) -> Graph:
    """`device_only`: without the classifier's trailing Softmax -- the graph output is the logits' tensor, which parity tests of
    the convolution stack want (a softmaxed output is mostly zeros).  The name is historical: until round 4 the int8 Softmax was
    the one tail op the HIP device left to the CPU subgraph; it runs on the device now (csrc/misc_kernels.hip softmax_i8)."""
    gf = BUILDERS[name](batch=1, **kw)
    if dtype == "fp32":
        g = set_batch(gf, batch)
    elif dtype == "int8":
        table = calib_table(name, gf) if not kw else None
        g = set_batch(quantize_int8(gf, table=table), batch)
    elif dtype == "uint8":
        table = calib_table(name, gf, dtype="uint8") if (not kw or set(kw) <= {"tail", "priorbox", "detection"}) else None
        g = set_batch(quantize_uint8(gf, table=table), batch)
    else:
        raise NotImplementedError(dtype)
    return strip_tail(g) if device_only else g


# --------------------------------------------------------------------------------------
# the reference's own benchmark files (benchmark/models/*_benchmark.tmfile: structure only, no weights)
# --------------------------------------------------------------------------------------
def _node_key(op, p):
    """what makes two nodes of one operator type the same computation (the fields the reference's kernels read)"""
    r = lambda v: round(float(v), 5)
    if op == "Convolution":
        return tuple(p.get(k) for k in ("kernel_h", "kernel_w", "stride_h", "stride_w", "dilation_h", "dilation_w", "group", "activation",
                                        "pad_h0", "pad_w0", "pad_h1", "pad_w1", "output_channel"))
    if op == "Pooling":
        if p.get("global"):
            return ("global", p.get("alg"))
        return tuple(p.get(k) for k in ("alg", "kernel_h", "kernel_w", "stride_h", "stride_w", "caffe_flavor", "pad_h0", "pad_w0"))
    if op == "ReLU":
        return (r(p.get("negative_slope", 0.0)),)
    if op in ("Concat", "Softmax"):
        return (p.get("axis", 1),)
    if op == "Flatten":
        return (p.get("axis", 1), p.get("end_axis", 3))
    if op == "Permute":
        return tuple(p.get("order", ()))
    if op == "Reshape":
        return tuple(p.get("re_shape", ()))
    if op == "Upsample":
        return (r(p.get("scale", 2)),)
    if op == "Eltwise":
        return (p.get("type"),)
    if op == "FullyConnected":
        return (p.get("num_output"),)
    if op == "PriorBox":
        return tuple(tuple(r(v) for v in p.get(k, [])) for k in ("min_size", "max_size", "aspect_ratio", "variance")) + \
            (p.get("flip"), p.get("clip"), r(p.get("offset", 0.5)))
    if op == "DetectionOutput":
        return (p.get("num_classes"), p.get("keep_top_k"), p.get("nms_top_k"), r(p.get("confidence_threshold")), r(p.get("nms_threshold")))
    return ()


def match_graphs(ga: Graph, gb: Graph):
    """Pairs the compute nodes of two graphs that are the same DAG up to node order and names: a node's signature is its
    operator, the parameters its kernel reads, the shapes of its constant inputs and -- recursively -- the signatures of the
    producers of its variable inputs, in input order.  Returns [(node index in ga, node index in gb)] or raises ValueError
    naming what has no partner.  (The converters emit nodes in their own topological order; tm_benchmark's files and the
    builders above agree on the graph, not on that order.)"""
    def sigs(g):
        prod = {}
        for ni, n in enumerate(g.nodes):
            for o in n.outputs:
                prod[o] = ni
        memo = {}

        def sig(ni):
            if ni in memo:
                return memo[ni]
            n = g.nodes[ni]
            if n.op == "InputOp":
                s = ("InputOp",)
            elif n.op == "Const":
                d = list(g.tensors[n.outputs[0]].dims)
                while len(d) > 1 and d[0] == 1:          # darknet's converter writes a bias as [1, 1, 1, C]
                    d.pop(0)
                s = ("Const", tuple(d))
            else:
                s = (n.op, _node_key(n.op, n.params), tuple(sig(prod[i]) for i in n.inputs))
            memo[ni] = hash(s)
            return memo[ni]

        out = {}
        for ni, n in enumerate(g.nodes):
            if n.op not in ("Const", "InputOp"):
                out.setdefault(sig(ni), []).append(ni)
        return out

    sa, sb = sigs(ga), sigs(gb)
    pairs = []
    for k, la in sa.items():
        lb = sb.get(k, [])
        if len(la) != len(lb):
            raise ValueError("no partner for node(s) %s" % [(ga.nodes[i].name, ga.nodes[i].op) for i in la[len(lb):]])
        pairs += list(zip(la, lb))
    for k, lb in sb.items():
        if k not in sa:
            raise ValueError("no partner for node(s) %s" % [(gb.nodes[i].name, gb.nodes[i].op) for i in lb])
    return sorted(pairs)


def graft_reference_file(ref_bytes: bytes, gq: Graph) -> bytes:
    """The reference's benchmark file with the tensors of `gq` (a built, possibly quantised, graph of the same topology) put in:
    constants get their data, every tensor its data type, shape and quantisation parameters -- nodes, names, parameter blobs
    and node order stay the file's own.  This is the "retag" SURVEY appendix E did in C for the files tm_benchmark ships
    without weights."""
    gr = tm2.read_tm2(ref_bytes)

    def put(tr, tq):
        tr.dims, tr.dtype, tr.scales, tr.zero_points = list(tq.dims), tq.dtype, tq.scales, tq.zero_points
        if tq.ttype == tm2.TT_CONST:
            tr.data = tq.data

    for nr, nq in match_graphs(gr, gq):
        a, b = gr.nodes[nr], gq.nodes[nq]
        assert len(a.inputs) == len(b.inputs) and len(a.outputs) == len(b.outputs), (a.name, b.name)
        for ti, tj in zip(a.inputs + a.outputs, b.inputs + b.outputs):
            put(gr.tensors[ti], gq.tensors[tj])
    return tm2.write_tm2(gr)


# tm_benchmark's model list (benchmark/tm_benchmark.cc:250-289) -> the builder and options that reproduce the file's graph
REFERENCE_BENCHMARKS = {
    "squeezenet_v1.1": ("squeezenet_v1.1", "fp32", {}),
    "mobilenet": ("mobilenet_v1", "int8", {}),
    "resnet50": ("resnet50", "int8", {}),
    "yolov3_tiny": ("yolov3_tiny", "uint8", {}),
    "mssd": ("mssd", "uint8", {"tail": True, "priorbox": True, "detection": True}),
}
