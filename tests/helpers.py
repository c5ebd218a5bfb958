"""Seeded single-op and small-graph builders shared by the CPU and GPU parity tests.
Every case is expressed as tmfile bytes so the reference, the oracle and the HIP backend all
consume the same model."""
import numpy as np

from tengine_amd import tm2
from tengine_amd.tm2 import DT_INT8, DT_INT32, Graph


def _scales(rng, n, lo=0.002, hi=0.02):
    return [float(np.float32(v)) for v in rng.uniform(lo, hi, size=n)]


def conv_graph(seed, n, cin, h, w, cout, k, s=1, p=0, group=1, act=0, bias=True, dil=1, kw=None, pw=None):
    rng = np.random.default_rng(seed)
    kh, kw = k, (k if kw is None else kw)
    ph, pw = p, (p if pw is None else pw)
    g = Graph(name="conv_case")
    xs = float(np.float32(rng.uniform(0.01, 0.05)))
    x = g.add_input("data", [n, cin, h, w], DT_INT8, [xs], [0])
    wq = rng.integers(-127, 128, size=(cout, cin // group, kh, kw)).astype(np.int8)
    ws = _scales(rng, cout)
    ins = [x, g.add_const("w", wq, DT_INT8, ws, [0] * cout)]
    if bias:
        bq = rng.integers(-2000, 2000, size=(cout,)).astype(np.int32)
        ins.append(g.add_const("b", bq, DT_INT32, [1.0], [0]))
    oh = (h - dil * (kh - 1) - 1 + 2 * ph) // s + 1
    ow = (w - dil * (kw - 1) - 1 + 2 * pw) // s + 1
    # output scale sized so results spread over the int8 range without saturating everywhere
    fan = (cin // group) * kh * kw
    os_ = float(np.float32(xs * np.mean(ws) * 73.0 * np.sqrt(fan) * 73.0 / 60.0))
    y = g.add_tensor("out", [n, cout, oh, ow], DT_INT8, tm2.TT_VAR, None, [os_], [0])
    ni = g.add_node("conv", "Convolution", ins, [y], kernel_h=kh, kernel_w=kw, stride_h=s, stride_w=s,
                    dilation_h=dil, dilation_w=dil, input_channel=cin, output_channel=cout, group=group,
                    activation=act, pad_h0=ph, pad_w0=pw, pad_h1=ph, pad_w1=pw)
    g.output_nodes = [ni]
    xin = rng.integers(-127, 128, size=(n, cin, h, w)).astype(np.int8)
    return g, xin


def fc_graph(seed, n, hidden_dims, nout, bias=True):
    rng = np.random.default_rng(seed)
    g = Graph(name="fc_case")
    xs = float(np.float32(rng.uniform(0.01, 0.05)))
    x = g.add_input("data", [n] + list(hidden_dims), DT_INT8, [xs], [0])
    hidden = int(np.prod(hidden_dims))
    wq = rng.integers(-127, 128, size=(nout, hidden)).astype(np.int8)
    ws = _scales(rng, nout)
    ins = [x, g.add_const("w", wq, DT_INT8, ws, [0] * nout)]
    if bias:
        ins.append(g.add_const("b", rng.integers(-2000, 2000, size=(nout,)).astype(np.int32), DT_INT32, [1.0], [0]))
    os_ = float(np.float32(xs * np.mean(ws) * 73.0 * np.sqrt(hidden) * 73.0 / 60.0))
    y = g.add_tensor("out", [n, nout], DT_INT8, tm2.TT_VAR, None, [os_], [0])
    ni = g.add_node("fc", "FullyConnected", ins, [y], num_output=nout)
    g.output_nodes = [ni]
    return g, rng.integers(-127, 128, size=[n] + list(hidden_dims)).astype(np.int8)


def pool_graph(seed, n, c, h, w, alg, k, s, p=0, glob=0, caffe=0, same_scale=False):
    from tengine_amd.models import pool_out
    rng = np.random.default_rng(seed)
    g = Graph(name="pool_case")
    xs = float(np.float32(rng.uniform(0.01, 0.05)))
    x = g.add_input("data", [n, c, h, w], DT_INT8, [xs], [0])
    if glob:
        oh = ow = 1
    else:
        oh, _, _ = pool_out(h, k, s, p, caffe)
        ow, _, _ = pool_out(w, k, s, p, caffe)
    os_ = xs if same_scale else float(np.float32(xs * rng.uniform(0.4, 1.3)))
    y = g.add_tensor("out", [n, c, oh, ow], DT_INT8, tm2.TT_VAR, None, [os_], [0])
    ni = g.add_node("pool", "Pooling", [x], [y], alg=alg, kernel_h=k, kernel_w=k, stride_h=s, stride_w=s,
                    **{"global": glob}, caffe_flavor=caffe, pad_h0=p, pad_w0=p, pad_h1=p, pad_w1=p)
    g.output_nodes = [ni]
    return g, rng.integers(-127, 128, size=(n, c, h, w)).astype(np.int8)


def eltwise_relu_graph(seed, n, c, h, w, with_relu=True, etype=tm2.ELT_SUM, k=1):
    """two k x k convs (same-size output) -> eltwise -> [relu]; mirrors a ResNet block tail (k = 3: a basic block's)."""
    rng = np.random.default_rng(seed)
    g, xin = conv_graph(seed, n, c, h, w, c, k, 1, k // 2, act=-1)
    g.output_nodes = []
    x = g.nodes[g.input_nodes[0]].outputs[0]
    a = g.nodes[-1].outputs[0]
    wq = rng.integers(-127, 128, size=(c, c, k, k)).astype(np.int8)
    ws = _scales(rng, c)
    wt = g.add_const("w2", wq, DT_INT8, ws, [0] * c)
    sb = float(np.float32(g.tensors[a].scales[0] * 1.37))
    b = g.add_tensor("out2", [n, c, h, w], DT_INT8, tm2.TT_VAR, None, [sb], [0])
    g.add_node("conv2", "Convolution", [x, wt], [b], kernel_h=k, kernel_w=k, stride_h=1, stride_w=1, dilation_h=1,
               dilation_w=1, input_channel=c, output_channel=c, group=1, activation=-1, pad_h0=k // 2, pad_w0=k // 2,
               pad_h1=k // 2, pad_w1=k // 2)
    so = float(np.float32(g.tensors[a].scales[0] * 1.9))
    e = g.add_tensor("sum", [n, c, h, w], DT_INT8, tm2.TT_VAR, None, [so], [0])
    ni = g.add_node("elt", "Eltwise", [a, b], [e], type=etype, caffe_flavor=1)
    if with_relu:
        r = g.add_tensor("relu", [n, c, h, w], DT_INT8, tm2.TT_VAR, None, [so], [0])
        ni = g.add_node("relu", "ReLU", [e], [r], negative_slope=0.0)
    g.output_nodes = [ni]
    return g, xin


# ---- uint8 (per-tensor asymmetric) single-op graphs --------------------------------------------------------
from tengine_amd.tm2 import DT_UINT8  # noqa: E402


def i8_unary_graph(seed, op, dims, out_scale=None, **params):
    """one int8 node (Softmax, ..) on a seeded input; out_scale None: a scale that spreads a softmax's (0, 1] over the int8 range"""
    rng = np.random.default_rng(seed)
    g = Graph(name="i8_%s_case" % op)
    xs = float(np.float32(rng.uniform(0.01, 0.06)))
    x = g.add_input("data", list(dims), DT_INT8, [xs], [0])
    os_ = float(np.float32(out_scale if out_scale is not None else rng.uniform(0.5, 1.5) / 127.0))
    y = g.add_tensor("out", list(dims), DT_INT8, tm2.TT_VAR, None, [os_], [0])
    ni = g.add_node(op.lower(), op, [x], [y], **params)
    g.output_nodes = [ni]
    return g, rng.integers(-127, 128, size=dims).astype(np.int8)


def _u8q(rng, lo=0.01, hi=0.05):
    # zero points near mid-range keep the accumulators centred (a random zp pair saturates every output);
    # extreme zero points are exercised through the explicit in_zp / w_zp / out_zp arguments
    return float(np.float32(rng.uniform(lo, hi))), int(rng.integers(112, 144))


def u8_conv_graph(seed, n, cin, h, w, cout, k, s=1, p=0, group=1, act=0, bias=True, dil=1, in_zp=None, w_zp=None,
                  out_zp=None):
    rng = np.random.default_rng(seed)
    g = Graph(name="u8conv_case")
    xs, xz = _u8q(rng)
    if in_zp is not None:
        xz = in_zp
    x = g.add_input("data", [n, cin, h, w], DT_UINT8, [xs], [xz])
    wq = rng.integers(0, 256, size=(cout, cin // group, k, k)).astype(np.uint8)
    ws, wz = _u8q(rng, 0.002, 0.02)
    if w_zp is not None:
        wz = w_zp
    ins = [x, g.add_const("w", wq, DT_UINT8, [ws], [wz])]
    if bias:
        bq = rng.integers(-4000, 4000, size=(cout,)).astype(np.int32)
        ins.append(g.add_const("b", bq, DT_INT32, [float(np.float32(xs) * np.float32(ws))], [0]))
    oh = (h - dil * (k - 1) - 1 + 2 * p) // s + 1
    ow = (w - dil * (k - 1) - 1 + 2 * p) // s + 1
    fan = (cin // group) * k * k
    os_ = float(np.float32(xs * ws * 73.0 * np.sqrt(fan) * 73.0 / 40.0))
    oz = int(rng.integers(0, 256)) if out_zp is None else out_zp
    if act >= 0 and out_zp is None:
        oz = int(rng.integers(0, 40))
    if act > 0:
        os_ = min(os_, float(np.float32(6.3 / (255 - oz))))
    y = g.add_tensor("out", [n, cout, oh, ow], DT_UINT8, tm2.TT_VAR, None, [os_], [oz])
    ni = g.add_node("conv", "Convolution", ins, [y], kernel_h=k, kernel_w=k, stride_h=s, stride_w=s,
                    dilation_h=dil, dilation_w=dil, input_channel=cin, output_channel=cout, group=group,
                    activation=act, pad_h0=p, pad_w0=p, pad_h1=p, pad_w1=p)
    g.output_nodes = [ni]
    return g, rng.integers(0, 256, size=(n, cin, h, w)).astype(np.uint8)


def u8_unary_graph(seed, op, dims, out_dims=None, same_q=False, **params):
    rng = np.random.default_rng(seed)
    g = Graph(name="u8_%s_case" % op)
    xs, xz = _u8q(rng)
    x = g.add_input("data", list(dims), DT_UINT8, [xs], [xz])
    if same_q:
        os_, oz = xs, xz
    else:
        os_, oz = float(np.float32(xs * rng.uniform(0.5, 1.4))), int(rng.integers(0, 256))
    y = g.add_tensor("out", list(out_dims or dims), DT_UINT8, tm2.TT_VAR, None, [os_], [oz])
    ni = g.add_node(op.lower(), op, [x], [y], **params)
    g.output_nodes = [ni]
    return g, rng.integers(0, 256, size=dims).astype(np.uint8)


def u8_pool_graph(seed, n, c, h, w, alg, k, s, p=0, glob=0, caffe=0, same_q=False):
    from tengine_amd.models import pool_out
    if glob:
        oh = ow = 1
    else:
        oh, _, _ = pool_out(h, k, s, p, caffe)
        ow, _, _ = pool_out(w, k, s, p, caffe)
    return u8_unary_graph(seed, "Pooling", [n, c, h, w], [n, c, oh, ow], same_q, alg=alg, kernel_h=k, kernel_w=k,
                          stride_h=s, stride_w=s, **{"global": glob}, caffe_flavor=caffe, pad_h0=p, pad_w0=p,
                          pad_h1=p, pad_w1=p)


def u8_fc_graph(seed, n, hidden_dims, nout, bias=True):
    rng = np.random.default_rng(seed)
    g = Graph(name="u8fc_case")
    xs, xz = _u8q(rng)
    x = g.add_input("data", [n] + list(hidden_dims), DT_UINT8, [xs], [xz])
    hidden = int(np.prod(hidden_dims))
    wq = rng.integers(0, 256, size=(nout, hidden)).astype(np.uint8)
    ws, wz = _u8q(rng, 0.002, 0.02)
    ins = [x, g.add_const("w", wq, DT_UINT8, [ws], [wz])]
    if bias:
        ins.append(g.add_const("b", rng.integers(-4000, 4000, size=(nout,)).astype(np.int32), DT_INT32,
                               [float(np.float32(xs) * np.float32(ws))], [0]))
    os_ = float(np.float32(xs * ws * 73.0 * np.sqrt(hidden) * 73.0 / 40.0))
    y = g.add_tensor("out", [n, nout], DT_UINT8, tm2.TT_VAR, None, [os_], [int(rng.integers(0, 256))])
    ni = g.add_node("fc", "FullyConnected", ins, [y], num_output=nout)
    g.output_nodes = [ni]
    return g, rng.integers(0, 256, size=[n] + list(hidden_dims)).astype(np.uint8)


def u8_route_graph(seed, n, c, h, w):
    """conv -> leaky -> (upsample x2 ; maxpool) mixed through a concat with per-input rescale, like the
    YOLOv3-tiny route: data -> leaky -> upsample --\\
                        data2(=maxpool of a 2x larger leaky) ----> concat -> leaky"""
    rng = np.random.default_rng(seed)
    g = Graph(name="u8route_case")
    xs, xz = _u8q(rng)
    x = g.add_input("data", [n, c, 2 * h, 2 * w], DT_UINT8, [xs], [xz])

    def var(name, dims, scale_mul):
        return g.add_tensor(name, dims, DT_UINT8, tm2.TT_VAR, None, [float(np.float32(xs * scale_mul))],
                            [int(rng.integers(0, 256))])
    lk = var("lk", [n, c, 2 * h, 2 * w], 0.8)
    g.add_node("leaky", "ReLU", [x], [lk], negative_slope=0.1)
    mp = var("mp", [n, c, h, w], 0.9)
    g.add_node("maxpool", "Pooling", [lk], [mp], alg=0, kernel_h=2, kernel_w=2, stride_h=2, stride_w=2,
               **{"global": 0}, caffe_flavor=0, pad_h0=0, pad_w0=0, pad_h1=0, pad_w1=0)
    up = var("up", [n, c, 2 * h, 2 * w], 1.1)
    g.add_node("upsample", "Upsample", [mp], [up], scale=2)
    cc = var("cat", [n, 2 * c, 2 * h, 2 * w], 1.2)
    ni = g.add_node("route", "Concat", [up, lk], [cc], axis=1)
    g.output_nodes = [ni]
    return g, rng.integers(0, 256, size=(n, c, 2 * h, 2 * w)).astype(np.uint8)


def u8_ssd_head_graph(seed, n, c, h, w, priors=(3, 6), per=4, same_q=False, standalone_permute=False):
    """SSD head plumbing on two feature maps (data at h x w, its 2x2 max-pool): per map a 1x1 conv with priors*per
    channels -> Permute(0,2,3,1) -> Flatten -> one Concat on axis 1 with per-input rescale (`same_q`: every head
    carries the concat's scale / zero point, the copy case).  `standalone_permute`: the graph ends at the first
    head's Permute (its own launch on the device)."""
    rng = np.random.default_rng(seed)
    g = Graph(name="u8_ssd_head_case")
    xs, xz = _u8q(rng)
    x = g.add_input("data", [n, c, h, w], DT_UINT8, [xs], [xz])
    mp = g.add_tensor("mp", [n, c, h // 2, w // 2], DT_UINT8, tm2.TT_VAR, None, [xs], [xz])
    g.add_node("maxpool", "Pooling", [x], [mp], alg=0, kernel_h=2, kernel_w=2, stride_h=2, stride_w=2,
               **{"global": 0}, caffe_flavor=0, pad_h0=0, pad_w0=0, pad_h1=0, pad_w1=0)
    cat_s, cat_z = float(np.float32(xs * 9.0)), int(rng.integers(60, 200))
    flats = []
    for i, (feat, fh, fw) in enumerate(((x, h, w), (mp, h // 2, w // 2))):
        cout = priors[i] * per
        wq = rng.integers(0, 256, size=(cout, c, 1, 1)).astype(np.uint8)
        ws, wz = _u8q(rng, 0.002, 0.02)
        bq = rng.integers(-4000, 4000, size=(cout,)).astype(np.int32)
        ins = [feat, g.add_const("w%d" % i, wq, DT_UINT8, [ws], [wz]),
               g.add_const("b%d" % i, bq, DT_INT32, [float(np.float32(xs) * np.float32(ws))], [0])]
        if same_q:
            os_, oz = cat_s, cat_z
        else:
            os_, oz = float(np.float32(xs * ws * 73.0 * np.sqrt(c) * 73.0 / 40.0 * (1.0 + 0.3 * i))), int(rng.integers(60, 200))
        y = g.add_tensor("head%d" % i, [n, cout, fh, fw], DT_UINT8, tm2.TT_VAR, None, [os_], [oz])
        g.add_node("head%d" % i, "Convolution", ins, [y], kernel_h=1, kernel_w=1, stride_h=1, stride_w=1, dilation_h=1,
                   dilation_w=1, input_channel=c, output_channel=cout, group=1, activation=-1, pad_h0=0, pad_w0=0,
                   pad_h1=0, pad_w1=0)
        pm = g.add_tensor("perm%d" % i, [n, fh, fw, cout], DT_UINT8, tm2.TT_VAR, None, [os_], [oz])
        pi = g.add_node("perm%d" % i, "Permute", [y], [pm], flag=0, order=[0, 2, 3, 1])
        if standalone_permute:
            g.output_nodes = [pi]
            return g, rng.integers(0, 256, size=(n, c, h, w)).astype(np.uint8)
        fl = g.add_tensor("flat%d" % i, [n, fh * fw * cout], DT_UINT8, tm2.TT_VAR, None, [os_], [oz])
        g.add_node("flat%d" % i, "Flatten", [pm], [fl], axis=1, end_axis=3)
        flats.append(fl)
    total = sum(g.tensors[f].dims[1] for f in flats)
    cc = g.add_tensor("mbox", [n, total], DT_UINT8, tm2.TT_VAR, None, [cat_s], [cat_z])
    ni = g.add_node("mbox", "Concat", flats, [cc], axis=1)
    g.output_nodes = [ni]
    return g, rng.integers(0, 256, size=(n, c, h, w)).astype(np.uint8)


def i8_concat_graph(seed, n, c, h, w, axis=1, shrink=False):
    """int8: data -> ReLU (scale s1) and data -> leaky ReLU (scale s2) -> Concat (scale s3) on `axis`.  `shrink`: the
    output scale is smaller than an input scale, so rescaled values leave [-127, 127] (the reference's clamp path)."""
    rng = np.random.default_rng(seed)
    g = Graph(name="i8concat_case")
    xs = float(np.float32(rng.uniform(0.01, 0.05)))
    x = g.add_input("data", [n, c, h, w], DT_INT8, [xs], [0])
    a = g.add_tensor("a", [n, c, h, w], DT_INT8, tm2.TT_VAR, None, [float(np.float32(xs * 0.9))], [0])
    g.add_node("relu", "ReLU", [x], [a], negative_slope=0.0)
    b = g.add_tensor("b", [n, c, h, w], DT_INT8, tm2.TT_VAR, None, [float(np.float32(xs * 1.1))], [0])
    g.add_node("leaky", "ReLU", [x], [b], negative_slope=0.9)
    d = [n, c, h, w]
    d[axis] *= 2
    so = float(np.float32(xs * (0.7 if shrink else 1.3)))
    y = g.add_tensor("cat", d, DT_INT8, tm2.TT_VAR, None, [so], [0])
    ni = g.add_node("cat", "Concat", [a, b], [y], axis=axis)
    g.output_nodes = [ni]
    return g, rng.integers(-127, 128, size=(n, c, h, w)).astype(np.int8)


def pwdw_graph(seed, n, cin, h, w, c, s=1, p=1, act_pw=0, act_dw=0, tail="dw", pool_alg=1, bias=True, first=None):
    """int8: pointwise 1x1 conv (cin -> c) -> depthwise 3x3 (stride s, pad p) | global pooling: the pair pwdw.hip fuses.
    `first` = (k, stride, pad[, dilation]): the producer is a k x k conv on the graph input instead (network's first layer)"""
    rng = np.random.default_rng(seed)
    fk, fs, fp, fd = (list(first) + [1])[:4] if first else (1, 1, 0, 1)
    g = Graph(name="pwdw_case")
    xs = float(np.float32(rng.uniform(0.01, 0.05)))
    x = g.add_input("data", [n, cin, h, w], DT_INT8, [xs], [0])
    wq = rng.integers(-127, 128, size=(c, cin, fk, fk)).astype(np.int8)
    ws = _scales(rng, c)
    ins = [x, g.add_const("w_pw", wq, DT_INT8, ws, [0] * c)]
    if bias:
        ins.append(g.add_const("b_pw", rng.integers(-2000, 2000, size=(c,)).astype(np.int32), DT_INT32, [1.0], [0]))
    ms = float(np.float32(xs * np.mean(ws) * 73.0 * np.sqrt(cin * fk * fk) * 73.0 / 60.0))
    h = (h - fd * (fk - 1) - 1 + 2 * fp) // fs + 1          # from here on: the producer's output map
    w = (w - fd * (fk - 1) - 1 + 2 * fp) // fs + 1
    mid = g.add_tensor("mid", [n, c, h, w], DT_INT8, tm2.TT_VAR, None, [ms], [0])
    g.add_node("pw", "Convolution", ins, [mid], kernel_h=fk, kernel_w=fk, stride_h=fs, stride_w=fs, dilation_h=fd, dilation_w=fd,
               input_channel=cin, output_channel=c, group=1, activation=act_pw, pad_h0=fp, pad_w0=fp, pad_h1=fp, pad_w1=fp)
    if tail == "pool":
        os_ = float(np.float32(ms * rng.uniform(0.3, 0.9)))
        y = g.add_tensor("out", [n, c, 1, 1], DT_INT8, tm2.TT_VAR, None, [os_], [0])
        ni = g.add_node("pool", "Pooling", [mid], [y], alg=pool_alg, kernel_h=h, kernel_w=w, stride_h=1, stride_w=1,
                        **{"global": 1}, caffe_flavor=0, pad_h0=0, pad_w0=0, pad_h1=0, pad_w1=0)
    else:
        dq = rng.integers(-127, 128, size=(c, 1, 3, 3)).astype(np.int8)
        dsc = _scales(rng, c)
        dins = [mid, g.add_const("w_dw", dq, DT_INT8, dsc, [0] * c)]
        if bias:
            dins.append(g.add_const("b_dw", rng.integers(-2000, 2000, size=(c,)).astype(np.int32), DT_INT32, [1.0], [0]))
        oh = (h - 3 + 2 * p) // s + 1
        ow = (w - 3 + 2 * p) // s + 1
        os_ = float(np.float32(ms * np.mean(dsc) * 73.0 * 3.0 * 73.0 / 60.0))
        y = g.add_tensor("out", [n, c, oh, ow], DT_INT8, tm2.TT_VAR, None, [os_], [0])
        ni = g.add_node("dw", "Convolution", dins, [y], kernel_h=3, kernel_w=3, stride_h=s, stride_w=s, dilation_h=1,
                        dilation_w=1, input_channel=c, output_channel=c, group=c, activation=act_dw, pad_h0=p, pad_w0=p,
                        pad_h1=p, pad_w1=p)
    g.output_nodes = [ni]
    return g, rng.integers(-127, 128, size=g.tensors[x].dims).astype(np.int8)



def stem_graph(seed, n, h, w, c, kw=7, pad=3, act=0, caffe=1, pool_k=3, pool_s=2, same_scale=True, cin=3, tail_conv=False):
    """int8 ResNet-style stem: 7 x kw / stride-2 conv on the NCHW graph input -> MAX pool (3x3 / 2 / pad 0 by default): the pair
    conv_first_pool.hip runs as one launch.  `tail_conv`: a 1x1 conv behind the pool, so that the pooled map is an inner tensor."""
    from tengine_amd.models import pool_out
    rng = np.random.default_rng(seed)
    g = Graph(name="stem_case")
    xs = float(np.float32(rng.uniform(0.01, 0.05)))
    x = g.add_input("data", [n, cin, h, w], DT_INT8, [xs], [0])
    wq = rng.integers(-127, 128, size=(c, cin, 7, kw)).astype(np.int8)
    ws = _scales(rng, c)
    ins = [x, g.add_const("w1", wq, DT_INT8, ws, [0] * c), g.add_const("b1", rng.integers(-2000, 2000, size=(c,)).astype(np.int32), DT_INT32, [1.0], [0])]
    ms = float(np.float32(xs * np.mean(ws) * 73.0 * np.sqrt(cin * 7 * kw) * 73.0 / 60.0))
    ch = (h - 7 + 2 * pad) // 2 + 1
    cw = (w - kw + 2 * pad) // 2 + 1
    mid = g.add_tensor("mid", [n, c, ch, cw], DT_INT8, tm2.TT_VAR, None, [ms], [0])
    g.add_node("conv1", "Convolution", ins, [mid], kernel_h=7, kernel_w=kw, stride_h=2, stride_w=2, dilation_h=1, dilation_w=1,
               input_channel=cin, output_channel=c, group=1, activation=act, pad_h0=pad, pad_w0=pad, pad_h1=pad, pad_w1=pad)
    oh, _, _ = pool_out(ch, pool_k, pool_s, 0, caffe)
    ow, _, _ = pool_out(cw, pool_k, pool_s, 0, caffe)
    ps = ms if same_scale else float(np.float32(ms * rng.uniform(0.5, 1.2)))
    y = g.add_tensor("pooled", [n, c, oh, ow], DT_INT8, tm2.TT_VAR, None, [ps], [0])
    ni = g.add_node("pool1", "Pooling", [mid], [y], alg=tm2.POOL_MAX, kernel_h=pool_k, kernel_w=pool_k, stride_h=pool_s, stride_w=pool_s,
                    **{"global": 0}, caffe_flavor=caffe, pad_h0=0, pad_w0=0, pad_h1=0, pad_w1=0)
    if tail_conv:
        w2 = rng.integers(-127, 128, size=(32, c, 1, 1)).astype(np.int8)
        s2 = _scales(rng, 32)
        os_ = float(np.float32(ps * np.mean(s2) * 73.0 * np.sqrt(c) * 73.0 / 60.0))
        z = g.add_tensor("out", [n, 32, oh, ow], DT_INT8, tm2.TT_VAR, None, [os_], [0])
        ni = g.add_node("conv2", "Convolution", [y, g.add_const("w2", w2, DT_INT8, s2, [0] * 32)], [z], kernel_h=1, kernel_w=1, stride_h=1,
                        stride_w=1, dilation_h=1, dilation_w=1, input_channel=c, output_channel=32, group=1, activation=0, pad_h0=0,
                        pad_w0=0, pad_h1=0, pad_w1=0)
    g.output_nodes = [ni]
    return g, rng.integers(-127, 128, size=(n, cin, h, w)).astype(np.int8)


def dwpw_graph(seed, n, c, h, w, cout, p=1, act_dw=0, act_pw=0, bias=True):
    """int8: depthwise 3x3 stride 1 (pad p) -> pointwise 1x1 (c -> cout): the pair dwpw.hip runs as one launch"""
    rng = np.random.default_rng(seed)
    g = Graph(name="dwpw_case")
    xs = float(np.float32(rng.uniform(0.01, 0.05)))
    x = g.add_input("data", [n, c, h, w], DT_INT8, [xs], [0])
    dq = rng.integers(-127, 128, size=(c, 1, 3, 3)).astype(np.int8)
    dsc = _scales(rng, c)
    dins = [x, g.add_const("w_dw", dq, DT_INT8, dsc, [0] * c)]
    if bias:
        dins.append(g.add_const("b_dw", rng.integers(-2000, 2000, size=(c,)).astype(np.int32), DT_INT32, [1.0], [0]))
    oh, ow = h - 3 + 2 * p + 1, w - 3 + 2 * p + 1
    ms = float(np.float32(xs * np.mean(dsc) * 73.0 * 3.0 * 73.0 / 60.0))
    mid = g.add_tensor("mid", [n, c, oh, ow], DT_INT8, tm2.TT_VAR, None, [ms], [0])
    g.add_node("dw", "Convolution", dins, [mid], kernel_h=3, kernel_w=3, stride_h=1, stride_w=1, dilation_h=1, dilation_w=1,
               input_channel=c, output_channel=c, group=c, activation=act_dw, pad_h0=p, pad_w0=p, pad_h1=p, pad_w1=p)
    wq = rng.integers(-127, 128, size=(cout, c, 1, 1)).astype(np.int8)
    ws = _scales(rng, cout)
    pins = [mid, g.add_const("w_pw", wq, DT_INT8, ws, [0] * cout)]
    if bias:
        pins.append(g.add_const("b_pw", rng.integers(-2000, 2000, size=(cout,)).astype(np.int32), DT_INT32, [1.0], [0]))
    os_ = float(np.float32(ms * np.mean(ws) * 73.0 * np.sqrt(c) * 73.0 / 60.0))
    y = g.add_tensor("out", [n, cout, oh, ow], DT_INT8, tm2.TT_VAR, None, [os_], [0])
    ni = g.add_node("pw", "Convolution", pins, [y], kernel_h=1, kernel_w=1, stride_h=1, stride_w=1, dilation_h=1, dilation_w=1,
                    input_channel=c, output_channel=cout, group=1, activation=act_pw, pad_h0=0, pad_w0=0, pad_h1=0, pad_w1=0)
    g.output_nodes = [ni]
    return g, rng.integers(-127, 128, size=(n, c, h, w)).astype(np.int8)


def priorbox_graph(seed, dtype, img_h, img_w, feats, min_sizes, max_sizes, ratios, flip=1, clip=0, offset=0.5, step=0.0,
                   img_param=0, variance=(0.1, 0.1, 0.2, 0.2), q=(2.0 / 255, 63)):
    """data -> chain of ReLU + max-pool nodes down to each (h, w) of `feats` -> one PriorBox per feature map -> Concat(axis 2):
    the priors part of an SSD tail with arbitrary (also non-square) image / map sizes.  Outputs: the concat, and every
    PriorBox alone when there is only one.  dtype: tm2.DT_UINT8 | tm2.DT_FP32."""
    rng = np.random.default_rng(seed)
    g = Graph(name="priorbox_case")
    u8 = dtype == tm2.DT_UINT8
    if dtype == tm2.DT_INT8:         # round 6: symmetric int8 (priorbox_ref.c:195-210 rounds, the uint8 form truncates); q[0] doubles as its scale
        u8, q = True, (float(np.float32(1.0 / 127.0)) * (q[0] * 255.0 / 2.0), 0)
    qa = dict(scales=[float(np.float32(0.02))], zps=[int(rng.integers(100, 150)) if dtype != tm2.DT_INT8 else 0]) if u8 else dict(scales=None, zps=None)
    x = g.add_input("data", [1, 3, img_h, img_w], dtype, qa["scales"], qa["zps"])
    pbs = []
    for i, (fh, fw) in enumerate(feats):
        kh, kw = img_h // fh, img_w // fw
        f = g.add_tensor("feat%d" % i, [1, 3, fh, fw], dtype, tm2.TT_VAR, None, qa["scales"], qa["zps"])
        g.add_node("pool%d" % i, "Pooling", [x], [f], alg=0, kernel_h=kh, kernel_w=kw, stride_h=kh, stride_w=kw,
                   **{"global": 0}, caffe_flavor=0, pad_h0=0, pad_w0=0, pad_h1=0, pad_w1=0)
        mn, mx = [float(v) for v in min_sizes[i]], [float(v) for v in max_sizes[i]]
        num = (len(ratios[i]) * (2 if flip else 1) + 1 + (1 if mx else 0)) * len(mn)
        y = g.add_tensor("pb%d" % i, [1, 2, fh * fw * num * 4, 1], dtype, tm2.TT_VAR, None,
                         [float(np.float32(q[0]))] if u8 else None, [int(q[1])] if u8 else None)
        g.add_node("priorbox%d" % i, "PriorBox", [f, x], [y], min_size=mn, max_size=mx,
                   aspect_ratio=[float(v) for v in ratios[i]], variance=[float(v) for v in variance], flip=flip, clip=clip,
                   offset=offset, step_w=float(step), step_h=float(step), img_h=img_param, img_w=img_param, num_priors=num,
                   out_dim=fh * fw * num * 4)
        pbs.append(y)
    if len(pbs) == 1:
        g.output_nodes = [len(g.nodes) - 1]
    else:
        total = sum(g.tensors[t].dims[2] for t in pbs)
        # a different output scale on purpose: the (once-only) concat launch requantises like any other uint8 concat
        cat = g.add_tensor("mbox_priorbox", [1, 2, total, 1], dtype, tm2.TT_VAR, None,
                           [float(np.float32(q[0] * 1.25))] if u8 else None, [int(q[1]) - 9 if dtype != tm2.DT_INT8 else 0] if u8 else None)
        ni = g.add_node("mbox_priorbox", "Concat", pbs, [cat], axis=2)
        g.output_nodes = [ni]
    if dtype == tm2.DT_INT8:
        return g, rng.integers(-127, 128, size=(1, 3, img_h, img_w)).astype(np.int8)
    xin = rng.integers(0, 256, size=(1, 3, img_h, img_w)).astype(np.uint8) if u8 else \
        rng.uniform(-1, 1, size=(1, 3, img_h, img_w)).astype(np.float32)
    return g, xin


def axis_concat_graph(seed, dtype, dims, axis, branches=3):
    """data -> `branches` leaky ReLUs (own slope, own output quantisation) -> Concat on `axis` (2, 3, -1 ...): a concat of
    run-time tensors on an axis other than the channels -- dense NCHW device tensors, uint8 (per-input rescale) or fp32."""
    rng = np.random.default_rng(seed)
    g = Graph(name="axis_concat_case")
    u8 = dtype == tm2.DT_UINT8
    x = g.add_input("data", list(dims), dtype, [0.03] if u8 else None, [120] if u8 else None)
    parts = []
    for i in range(branches):
        qs = ([float(np.float32(0.03 * rng.uniform(0.6, 1.5)))], [int(rng.integers(90, 160))]) if u8 else (None, None)
        r = g.add_tensor("r%d" % i, list(dims), dtype, tm2.TT_VAR, None, qs[0], qs[1])
        g.add_node("relu%d" % i, "ReLU", [x], [r], negative_slope=0.1 * (i + 1))
        parts.append(r)
    od = list(dims)
    od[axis] = dims[axis] * branches
    y = g.add_tensor("cat", od, dtype, tm2.TT_VAR, None, [0.035] if u8 else None, [131] if u8 else None)
    g.output_nodes = [g.add_node("cat", "Concat", parts, [y], axis=axis)]
    xin = rng.integers(0, 256, size=dims).astype(np.uint8) if u8 else rng.uniform(-1, 1, size=dims).astype(np.float32)
    return g, xin


# PriorBox parameter sets shared by the CPU (oracle vs reference / golden) and GPU (device vs oracle / golden) tests
PRIORBOX_CASES = {
    "ssd304_conv11": dict(seed=1, img_h=304, img_w=304, feats=[(19, 19)], min_sizes=[[60]], max_sizes=[[]], ratios=[[2]]),
    "ssd_three_maps_concat": dict(seed=2, img_h=96, img_w=96, feats=[(12, 12), (6, 6), (3, 3)], min_sizes=[[20], [35], [50]],
                                  max_sizes=[[35], [50], [70]], ratios=[[2], [2, 3], [2, 3]]),
    "non_square_fractional_sizes_clip": dict(seed=3, img_h=48, img_w=64, feats=[(6, 8), (3, 4), (1, 2)],
                                             min_sizes=[[20], [30, 35.7], [40]], max_sizes=[[], [44, 50.2], [60]],
                                             ratios=[[2], [2, 3], [1.5, 2.5, 3]], clip=1),
    "no_flip_own_step_and_image": dict(seed=4, img_h=64, img_w=64, feats=[(8, 8), (4, 4)], min_sizes=[[16], [32]],
                                       max_sizes=[[24], [48]], ratios=[[2, 3], [2]], flip=0, step=7.5, img_param=60, offset=0.25,
                                       variance=(0.1, 0.15, 0.2, 0.3)),
    "no_ratios": dict(seed=5, img_h=32, img_w=32, feats=[(4, 4)], min_sizes=[[8, 12]], max_sizes=[[12, 20]], ratios=[[]]),
}


def u8_conv_pool_graph(seed, n, cin, h, w, cout, k=3, p=1, slope=0.1, relu=True, second_reader=False, pool_k=2, pool_s=2,
                       same_q=False):
    """conv (-> leaky ReLU) -> max-pool, the YOLOv3-tiny stage; `second_reader`: the unpooled tensor also feeds
    a ReLU that is a graph output (the fused kernel must still store it); pool input / output quantisation differ unless
    same_q (the quantiser normally hands the pool's parameters back to its input: both cases occur)."""
    g, x = u8_conv_graph(seed, n, cin, h, w, cout, k, 1, p, act=-1)
    rng = np.random.default_rng(seed + 1000)
    t = g.tensors[g.nodes[-1].outputs[0]]
    oh, ow = t.dims[2], t.dims[3]
    cur = g.nodes[-1].outputs[0]
    outs = []
    if relu:
        r = g.add_tensor("lk", [n, cout, oh, ow], DT_UINT8, tm2.TT_VAR, None, [float(np.float32(t.scales[0] * 0.9))], [int(rng.integers(20, 60))])
        g.add_node("leaky", "ReLU", [cur], [r], negative_slope=slope)
        cur = r
    ct = g.tensors[cur]
    qs = (ct.scales, ct.zero_points) if same_q else ([float(np.float32(ct.scales[0] * 1.07))], [int(rng.integers(0, 40))])
    ph, pw = (oh - pool_k) // pool_s + 1, (ow - pool_k) // pool_s + 1
    po = g.add_tensor("pooled", [n, cout, ph, pw], DT_UINT8, tm2.TT_VAR, None, list(qs[0]), list(qs[1]))
    outs.append(g.add_node("maxpool", "Pooling", [cur], [po], alg=0, kernel_h=pool_k, kernel_w=pool_k, stride_h=pool_s,
                           stride_w=pool_s, **{"global": 0}, caffe_flavor=0, pad_h0=0, pad_w0=0, pad_h1=0, pad_w1=0))
    if second_reader:
        r2 = g.add_tensor("side", [n, cout, oh, ow], DT_UINT8, tm2.TT_VAR, None, [float(np.float32(ct.scales[0] * 1.2))], [17])
        outs.append(g.add_node("side_relu", "ReLU", [cur], [r2], negative_slope=0.0))
    g.output_nodes = outs
    return g, x


# ---- the opt-in integer uint8 path (tamd_options.u8_integer; csrc/u8i_kernels.hip) ---------------------------------------------
def u8_conv_int_model(g, x):
    """What the integer path computes for a single group-1 uint8 Convolution graph (u8_conv_graph), operation for operation
    (csrc/u8_epilogue.h: u8i_requant): t = the exact integer sum of (x - zx)(w - zw) + bias; y = fma((float)t, M, copysign(0.5, t))
    in binary32 with M = fl(fl(in_scale * w_scale) / out_scale); q = clamp(trunc(clamp(y, +-512)) + zp, lo, hi), the conv's own
    activation being the clamp window (relu: lo = zp; relu6: also hi = round(fl(6 / out_scale)) + zp).  Not the reference's bytes
    (its fp32 simulation rounds K times, divides and rounds): the bar against the reference is one quantisation step."""
    node = g.nodes[-1]
    assert node.op == "Convolution"
    p = node.params
    xt = g.tensors[node.inputs[0]]
    wt = g.tensors[node.inputs[1]]
    yt = g.tensors[node.outputs[0]]
    w = np.asarray(wt.data).astype(np.int64) - int(wt.zero_points[0])
    xv = x.astype(np.int64) - int(xt.zero_points[0])
    n, cin, h, wd = x.shape
    cout, _, kh, kw = w.shape
    s, dil, pad = p["stride_h"], p["dilation_h"], p["pad_h0"]
    oh, ow = yt.dims[2], yt.dims[3]
    xp = np.zeros((n, cin, h + 2 * pad, wd + 2 * pad), np.int64)       # out-of-image taps contribute (zx - zx) = 0
    xp[:, :, pad:pad + h, pad:pad + wd] = xv
    acc = np.zeros((n, cout, oh, ow), np.int64)
    for ky in range(kh):
        for kx in range(kw):
            patch = xp[:, :, ky * dil:ky * dil + (oh - 1) * s + 1:s, kx * dil:kx * dil + (ow - 1) * s + 1:s]
            acc += np.einsum("nchw,oc->nohw", patch, w[:, :, ky, kx])
    if len(node.inputs) > 2:
        acc += np.asarray(g.tensors[node.inputs[2]].data).astype(np.int64).reshape(1, cout, 1, 1)
    assert np.abs(acc).max() < 2 ** 31
    os_, zp = np.float32(yt.scales[0]), int(yt.zero_points[0])
    m = np.float32(np.float32(np.float32(xt.scales[0]) * np.float32(wt.scales[0])) / os_)
    tf = acc.astype(np.float32)                                          # int32 -> binary32, round to nearest even
    # the fused multiply-add, exactly: a 24 x 24 bit product and +-0.5 fit binary64 without rounding; ONE rounding to binary32
    y = (tf.astype(np.float64) * np.float64(m) + np.copysign(0.5, tf.astype(np.float64))).astype(np.float32)
    y = np.clip(y, np.float32(-512), np.float32(512))
    q = np.trunc(y).astype(np.int64) + zp
    act = p["activation"]
    lo, hi = 0, 255
    if act >= 0:
        lo = min(max(zp, 0), 255)
    if act > 0:
        r6 = np.float32(6.0) / os_
        hi = min(255, max(lo, int(np.sign(r6) * np.floor(np.abs(np.float64(r6)) + 0.5)) + zp))
    return np.clip(q, lo, hi).astype(np.uint8)


# ---- TAMD_PIN (tengine_amd/csrc/env.h): the one variable tests use to pin a plan-time choice or switch a live optimisation off ----
def pin(**kv):
    """merge key=value pairs into TAMD_PIN (value None removes the key); returns the previous string for restore_pins()"""
    import os
    old = os.environ.get("TAMD_PIN")
    cur = dict(p.split("=", 1) for p in old.split(",") if p) if old else {}
    for k, v in kv.items():
        if v is None:
            cur.pop(k, None)
        else:
            cur[k] = str(v)
    if cur:
        os.environ["TAMD_PIN"] = ",".join("%s=%s" % kv2 for kv2 in cur.items())
    else:
        os.environ.pop("TAMD_PIN", None)
    return old


def restore_pins(old):
    import os
    if old is None:
        os.environ.pop("TAMD_PIN", None)
    else:
        os.environ["TAMD_PIN"] = old


class pinned:
    """with pinned(u8_patch=1, u8_patch_cfg=4): ..."""

    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.old = pin(**self.kv)

    def __exit__(self, *a):
        restore_pins(self.old)


# ---- int8 head plumbing (round 6): Permute / Flatten / Reshape / Concat / Softmax behind a convolution ----------------------------
def i8_head_graph(seed, n, cin, h, w, couts=(12, 21), tail="concat", same_q=False, softmax_axis=None):
    """int8: data -> len(couts) 1x1 convolutions (channel counts that are NOT multiples of 16, like the SSD heads) -> per head
    `tail`:
      "concat"   Permute(0,2,3,1) -> Flatten -> ONE Concat on axis 1 (per-input rescale unless same_q)      [the SSD head]
      "permute"  the graph ends at the first head's Permute
      "flatcat"  Flatten (NCHW order, no Permute) -> Concat on axis 1
      "reshape"  Reshape of the first head to [n, h*w, cout] (after Permute) -> Softmax over `softmax_axis` (1 | 2) -> Flatten
      "softmax4" Softmax over axis `softmax_axis` (1 | 2 | 3) of the first head's 4-D map
      "reshape_map" Reshape of the first head's NCHW map (no Permute) to [n, cout*h, w] -> Softmax over `softmax_axis`"""
    rng = np.random.default_rng(seed)
    g = Graph(name="i8_head_case")
    xs = float(np.float32(rng.uniform(0.01, 0.05)))
    x = g.add_input("data", [n, cin, h, w], DT_INT8, [xs], [0])
    cat_s = float(np.float32(xs * 7.0))
    flats = []
    for i, cout in enumerate(couts):
        wq = rng.integers(-127, 128, size=(cout, cin, 1, 1)).astype(np.int8)
        ws = _scales(rng, cout)
        bq = rng.integers(-2000, 2000, size=(cout,)).astype(np.int32)
        ins = [x, g.add_const("w%d" % i, wq, DT_INT8, ws, [0] * cout), g.add_const("b%d" % i, bq, DT_INT32, [1.0], [0])]
        os_ = cat_s if same_q else float(np.float32(xs * np.mean(ws) * 73.0 * np.sqrt(cin) * 73.0 / 60.0 * (1.0 + 0.4 * i)))
        y = g.add_tensor("head%d" % i, [n, cout, h, w], DT_INT8, tm2.TT_VAR, None, [os_], [0])
        g.add_node("head%d" % i, "Convolution", ins, [y], kernel_h=1, kernel_w=1, stride_h=1, stride_w=1, dilation_h=1, dilation_w=1,
                   input_channel=cin, output_channel=cout, group=1, activation=-1, pad_h0=0, pad_w0=0, pad_h1=0, pad_w1=0)
        if tail == "softmax4":
            sm = g.add_tensor("prob", [n, cout, h, w], DT_INT8, tm2.TT_VAR, None, [float(np.float32(1.0 / 127.0))], [0])
            g.output_nodes = [g.add_node("prob", "Softmax", [y], [sm], axis=softmax_axis)]
            return g, rng.integers(-127, 128, size=(n, cin, h, w)).astype(np.int8)
        if tail == "flatcat":
            fl = g.add_tensor("flat%d" % i, [n, cout * h * w], DT_INT8, tm2.TT_VAR, None, [os_], [0])
            g.add_node("flat%d" % i, "Flatten", [y], [fl], axis=1, end_axis=3)
            flats.append(fl)
            continue
        if tail == "reshape_map":       # Reshape of the NCHW map itself (no Permute): [n, cout * h, w] -> Softmax over the last axis -> graph output
            rs = g.add_tensor("rs", [n, cout * h, w], DT_INT8, tm2.TT_VAR, None, [os_], [0])
            g.add_node("rs", "Reshape", [y], [rs], is_mxnet=0, reverse=0, is_onnx=1, re_shape=[0, -1, w])
            sm = g.add_tensor("prob", [n, cout * h, w], DT_INT8, tm2.TT_VAR, None, [float(np.float32(1.0 / 127.0))], [0])
            g.output_nodes = [g.add_node("prob", "Softmax", [rs], [sm], axis=softmax_axis)]
            return g, rng.integers(-127, 128, size=(n, cin, h, w)).astype(np.int8)
        pm = g.add_tensor("perm%d" % i, [n, h, w, cout], DT_INT8, tm2.TT_VAR, None, [os_], [0])
        pi = g.add_node("perm%d" % i, "Permute", [y], [pm], flag=0, order=[0, 2, 3, 1])
        if tail == "permute":
            g.output_nodes = [pi]
            return g, rng.integers(-127, 128, size=(n, cin, h, w)).astype(np.int8)
        if tail == "reshape":
            rs = g.add_tensor("rs", [n, h * w, cout], DT_INT8, tm2.TT_VAR, None, [os_], [0])
            g.add_node("rs", "Reshape", [pm], [rs], is_mxnet=0, reverse=0, is_onnx=1, re_shape=[0, -1, cout])
            sm = g.add_tensor("prob", [n, h * w, cout], DT_INT8, tm2.TT_VAR, None, [float(np.float32(1.0 / 127.0))], [0])
            g.add_node("prob", "Softmax", [rs], [sm], axis=softmax_axis)
            fl = g.add_tensor("prob_flat", [n, h * w * cout], DT_INT8, tm2.TT_VAR, None, [float(np.float32(1.0 / 127.0))], [0])
            g.output_nodes = [g.add_node("prob_flat", "Flatten", [sm], [fl], axis=1, end_axis=2)]
            return g, rng.integers(-127, 128, size=(n, cin, h, w)).astype(np.int8)
        fl = g.add_tensor("flat%d" % i, [n, h * w * cout], DT_INT8, tm2.TT_VAR, None, [os_], [0])
        g.add_node("flat%d" % i, "Flatten", [pm], [fl], axis=1, end_axis=3)
        flats.append(fl)
    total = sum(g.tensors[f].dims[1] for f in flats)
    cc = g.add_tensor("mbox", [n, total], DT_INT8, tm2.TT_VAR, None, [cat_s], [0])
    g.output_nodes = [g.add_node("mbox", "Concat", flats, [cc], axis=1)]
    return g, rng.integers(-127, 128, size=(n, cin, h, w)).astype(np.int8)


I8_HEAD_CASES = {
    "ssd_head_two_maps": dict(seed=1, n=2, cin=24, h=5, w=7, couts=(12, 21), tail="concat"),
    "ssd_head_same_scale_ten_inputs": dict(seed=2, n=1, cin=16, h=3, w=3, couts=(4, 6, 12, 5, 7, 9, 3, 8, 10, 11), tail="concat", same_q=True),
    "ssd_head_rescaled_ten_inputs": dict(seed=3, n=3, cin=16, h=2, w=3, couts=(4, 6, 12, 5, 7, 9, 3, 8, 10, 11), tail="concat"),
    "standalone_permute": dict(seed=4, n=2, cin=8, h=4, w=6, couts=(20,), tail="permute"),
    "flatten_maps_then_concat": dict(seed=5, n=2, cin=8, h=3, w=5, couts=(12, 32), tail="flatcat"),
    "reshape_softmax_last_axis": dict(seed=6, n=2, cin=8, h=4, w=5, couts=(21,), tail="reshape", softmax_axis=2),
    "reshape_softmax_middle_axis": dict(seed=7, n=1, cin=8, h=3, w=4, couts=(6,), tail="reshape", softmax_axis=1),
    "softmax_over_h": dict(seed=8, n=2, cin=8, h=6, w=5, couts=(12,), tail="softmax4", softmax_axis=2),
    "softmax_over_w": dict(seed=9, n=2, cin=8, h=4, w=9, couts=(32,), tail="softmax4", softmax_axis=3),
    "softmax_over_c_padded": dict(seed=10, n=2, cin=8, h=4, w=3, couts=(21,), tail="softmax4", softmax_axis=1),
}


def single_input_concat_graph(seed, dtype, dims=(2, 12, 5, 4)):
    """data -> leaky ReLU (own quantisation) -> Concat of that ONE tensor into an output with ANOTHER scale / zero point: the reference
    copies the bytes as they are (concat_kernel_ref_int8.c:47-57, concat_kernel_ref_uint8.c:47-58) -- it does not rescale a lone input"""
    rng = np.random.default_rng(seed)
    u8 = dtype == tm2.DT_UINT8
    g = Graph(name="single_input_concat_case")
    x = g.add_input("data", list(dims), dtype, [0.03], [120 if u8 else 0])
    r = g.add_tensor("r", list(dims), dtype, tm2.TT_VAR, None, [0.041], [97 if u8 else 0])
    g.add_node("relu", "ReLU", [x], [r], negative_slope=0.25)
    y = g.add_tensor("cat", list(dims), dtype, tm2.TT_VAR, None, [0.07], [131 if u8 else 0])
    g.output_nodes = [g.add_node("cat", "Concat", [r], [y], axis=1)]
    xin = rng.integers(0, 256, size=dims).astype(np.uint8) if u8 else rng.integers(-127, 128, size=dims).astype(np.int8)
    return g, xin


I8_HEAD_CASES["single_head_concat_is_a_copy"] = dict(seed=11, n=2, cin=16, h=3, w=4, couts=(21,), tail="concat")
I8_HEAD_CASES["single_flattened_map_concat_is_a_copy"] = dict(seed=12, n=2, cin=16, h=3, w=4, couts=(24,), tail="flatcat")
I8_HEAD_CASES["reshape_of_a_map_softmax"] = dict(seed=13, n=2, cin=8, h=3, w=7, couts=(12,), tail="reshape_map", softmax_axis=2)
I8_HEAD_CASES["reshape_of_a_map_softmax_middle"] = dict(seed=14, n=2, cin=8, h=2, w=5, couts=(20,), tail="reshape_map", softmax_axis=1)
