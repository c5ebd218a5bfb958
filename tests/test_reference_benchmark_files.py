"""The reference's OWN model files (benchmark/models/*_benchmark.tmfile, the list tm_benchmark.cc:250-289 runs) through the HIP
device.  The files carry structure only (no weights: tm2_serializer.c:240-246 zero-fills), so the five BASELINE graphs get the
tensors of the seeded synthetic models grafted onto the file's own nodes (models.graft_reference_file: data, data types and
quantisation parameters go in; nodes, names, parameter blobs and node order stay the file's) -- the "retag" SURVEY appendix E did
in C.  CPU part: every BASELINE builder is the file's graph node for node (structural match, the converters' node order differs);
the split of all fourteen files on device "HIP" in fp32.  GPU part: the grafted files on "HIP" against the reference CPU device
on the same bytes, with the placement asserted."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

from tengine_amd import capi, models, tm2

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NPD = {"fp32": tm2.DT_FP32, "int8": tm2.DT_INT8, "uint8": tm2.DT_UINT8}


def _model_file(name):
    for d in ("/root/reference/benchmark/models", os.path.join(ROOT, "oracle", "_ref", "models")):
        p = os.path.join(d, "%s_benchmark.tmfile" % name)
        if os.path.exists(p):
            return p
    pytest.skip("tm_benchmark's model files are not here (oracle/build_ref.py copies them to oracle/_ref/models)")


@pytest.mark.parametrize("fname", sorted(models.REFERENCE_BENCHMARKS))
def test_builders_are_the_reference_files_node_for_node(fname):
    """every compute node of the file has exactly one partner in the builder's graph: same operator, same parameters (the ones the
    kernels read), same constant shapes, same producers -- and vice versa"""
    name, dtype, kw = models.REFERENCE_BENCHMARKS[fname]
    gr = tm2.read_tm2(open(_model_file(fname), "rb").read())
    gm = models.BUILDERS[name](**kw)
    pairs = models.match_graphs(gr, gm)
    nr = [n for n in gr.nodes if n.op not in ("Const", "InputOp")]
    nm = [n for n in gm.nodes if n.op not in ("Const", "InputOp")]
    assert len(pairs) == len(nr) == len(nm)
    assert len({a for a, _ in pairs}) == len(pairs) == len({b for _, b in pairs})
    for a, b in pairs:
        assert gr.nodes[a].op == gm.nodes[b].op
    # the census of SURVEY appendix C
    want = {"squeezenet_v1.1": 40, "mobilenet": 29, "resnet50": 89, "yolov3_tiny": 35, "mssd": 84}[fname]
    assert len(pairs) == want


def test_a_differing_graph_is_not_matched():
    gr = tm2.read_tm2(open(_model_file("yolov3_tiny"), "rb").read())
    gm = models.yolov3_tiny_fp32()
    [n for n in gm.nodes if n.op == "Pooling"][2].params["stride_h"] = 1
    with pytest.raises(ValueError):
        models.match_graphs(gr, gm)


def test_grafted_files_run_on_the_reference_cpu_like_the_builders_own_files(ref):
    """the graft changes nothing the kernels see: the file's graph with our tensors == our graph, on the reference's CPU device"""
    for fname in ("mobilenet", "yolov3_tiny", "mssd"):
        name, dtype, kw = models.REFERENCE_BENCHMARKS[fname]
        gq = models.build(name, dtype, 1, **kw)
        b = models.graft_reference_file(open(_model_file(fname), "rb").read(), gq)
        x = models.synth_input(gq, 5, NPD[dtype])
        mode = {"fp32": ref.MODE_FP32, "int8": ref.MODE_INT8, "uint8": ref.MODE_UINT8}[dtype]
        got, want = ref.run_model(b, x, mode, 4), ref.run_model(tm2.write_tm2(gq), x, mode, 4)
        assert len(got) == len(want) >= 1
        for a, c in zip(got, want):
            assert np.array_equal(a, c)
        names = [n.name for n in tm2.read_tm2(b).nodes]
        assert names == [n.name for n in tm2.read_tm2(open(_model_file(fname), "rb").read()).nodes]      # the file's own nodes, in its order


def _plugin(ref):
    import test_plugin_dropin as tp
    tp._load_plugin(ref)
    return tp


# what tm_benchmark lists (tm_benchmark.cc:250-289) and what keeps a graph from being ONE "HIP" subgraph in fp32
ALL_FILES = {"squeezenet_v1.1": (227, 227), "mobilenet": (224, 224), "mobilenet_v2": (224, 224), "mobilenet_v3": (224, 224),
             "shufflenet_v2": (224, 224), "resnet18": (224, 224), "resnet50": (224, 224), "googlenet": (224, 224), "inception_v3": (299, 299),
             "vgg16": (224, 224), "mssd": (300, 300), "retinaface": (320, 240), "yolov3_tiny": (416, 416), "mobilefacenets": (112, 112)}     # img_h, img_w


def split_table(ref):
    """[(file, compute nodes, HIP subgraphs, nodes on HIP, sorted operator names left to the CPU device)] for the fourteen files, fp32"""
    tp = _plugin(ref)
    rows = []
    for f, (h, w) in ALL_FILES.items():
        b = open(_model_file(f), "rb").read()
        x = np.zeros((1, 3, h, w), np.float32)
        rg = ref.RefGraph(b, ref.MODE_FP32, 1, device="HIP", dev_opt=tp.HipOpt(b"HIP", C.sizeof(tp.HipOpt), 0, 1, 0))
        rg.set_input(x)
        try:
            rg.prerun()
        except RuntimeError:
            assert capi.device_count() == 0
        pl = tp.placement(rg)
        rg.close()
        hip = [p for p in pl if p[0] == "HIP" and p[2]]
        cpu_ops = sorted({o for p in pl if p[0] != "HIP" for o in p[3] if o not in ("InputOp", "Const")})
        rows.append((f, sum(p[2] for p in pl), len(hip), sum(p[2] for p in hip), cpu_ops))
    return rows


# the table of INTEGRATION.md section G: file -> operators the fp32 split leaves to the CPU device ([] = ONE "HIP" subgraph)
FP32_SPLIT = {
    "squeezenet_v1.1": [], "mobilenet": [], "mobilenet_v2": [], "resnet18": [], "resnet50": [], "inception_v3": [], "vgg16": [], "yolov3_tiny": [],
    "mobilenet_v3": ["BroadMul", "Eltwise", "Flatten"],           # squeeze-excite gates (BroadMul) and the hard-swish Eltwise forms
    "shufflenet_v2": ["ShuffleChannel", "Slice"],
    "googlenet": ["Lrn"],
    "mssd": ["Concat", "DetectionOutput", "Flatten", "Permute", "PriorBox", "Reshape", "Softmax"],      # the SSD head is a uint8 device path (below), not an fp32 one
    "retinaface": ["Convolution", "Crop", "Interp", "Reshape"],
    "mobilefacenets": ["BatchNormalize", "Eltwise", "PReLU"],
}


def test_split_of_all_fourteen_benchmark_files_in_fp32(ref):
    """which of tm_benchmark's graphs the device takes whole, and which operators keep the rest on the CPU device
    (tools/ref_benchmark_split.py prints the table of INTEGRATION.md section G from the same function)"""
    rows = {r[0]: r for r in split_table(ref)}
    assert sorted(rows) == sorted(FP32_SPLIT)
    for f, (_, nodes, hip_subgraphs, on_hip, cpu_ops) in rows.items():
        assert cpu_ops == FP32_SPLIT[f], (f, cpu_ops)
        if not cpu_ops:
            assert hip_subgraphs == 1 and on_hip == nodes, rows[f]
        else:
            assert 1 <= on_hip < nodes, rows[f]          # the convolution stack is on the device, the rest cut around (no surrender of the whole graph)


def test_mssd_file_in_uint8_leaves_only_detection_output_to_the_cpu(ref):
    """BASELINE configs[4]'s graph as the reference ships it: 83 nodes in ONE "HIP" subgraph, DetectionOutput alone on the CPU device"""
    tp = _plugin(ref)
    name, dtype, kw = models.REFERENCE_BENCHMARKS["mssd"]
    gq = models.build(name, dtype, 1, **kw)
    b = models.graft_reference_file(open(_model_file("mssd"), "rb").read(), gq)
    rg = ref.RefGraph(b, ref.MODE_UINT8, 1, device="HIP", dev_opt=tp.HipOpt(b"HIP", C.sizeof(tp.HipOpt), 0, 1, 0))
    rg.set_input(models.synth_input(gq, 5, tm2.DT_UINT8))
    try:
        rg.prerun()
    except RuntimeError:
        assert capi.device_count() == 0
    pl = tp.placement(rg)
    rg.close()
    real = [(dev, [o for o in ops if o not in ("InputOp", "Const")]) for dev, _, r, ops in pl if r]
    assert len(real) == 2 and real[0][0] == "HIP" and len(real[0][1]) == 83 and real[1][0] != "HIP" and real[1][1] == ["DetectionOutput"], pl


@pytest.mark.gpu
@pytest.mark.parametrize("fname", sorted(models.REFERENCE_BENCHMARKS))
def test_reference_benchmark_file_on_hip_equals_reference_cpu(ref, fname):
    """the reference's file (grafted tensors) through create_graph / prerun / run_graph on device "HIP": placement as claimed, bytes
    of the reference's CPU device (fp32: 1e-4)"""
    tp = _plugin(ref)
    name, dtype, kw = models.REFERENCE_BENCHMARKS[fname]
    gq = models.build(name, dtype, 1, **kw)
    b = models.graft_reference_file(open(_model_file(fname), "rb").read(), gq)
    x = models.synth_input(gq, 11, NPD[dtype])
    mode = {"fp32": ref.MODE_FP32, "int8": ref.MODE_INT8, "uint8": ref.MODE_UINT8}[dtype]
    want = ref.run_model(b, x, mode, 8)
    rg = ref.RefGraph(b, mode, 1, device="HIP", dev_opt=tp.HipOpt(b"HIP", C.sizeof(tp.HipOpt), 0, 1, 0))
    rg.set_input(x)
    rg.run()
    pl = tp.placement(rg)
    got = rg.outputs()
    rg.run()
    again = rg.outputs()
    rg.close()
    real = [(dev, ops) for dev, _, r, ops in pl if r]
    if fname == "mssd":          # the 83 nodes in front of DetectionOutput are ONE HIP subgraph, the host-side post-processing one CPU node
        assert [d for d, _ in real] == ["HIP", real[1][0]] and real[1][0] != "HIP" and real[1][1] == ["DetectionOutput"], pl
        assert len([o for o in real[0][1] if o not in ("InputOp", "Const")]) == 83, pl
    else:
        assert len(real) == 1 and real[0][0] == "HIP", pl
    assert len(want) == len(got)
    for w, o, a in zip(want, got, again):
        if dtype == "fp32" or w.dtype == np.float32:
            assert w.shape == o.shape and np.abs(w - o).max() <= 1e-4 and np.abs(w - a).max() <= 1e-4
        else:
            assert np.array_equal(w, o) and np.array_equal(w, a)
