"""The C-ABI library loads here (no GPU) and exports every symbol include/tengine_amd.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "tengine_amd.h")).read()
    return sorted(set(re.findall(r"TAMD_API\s+[\w\s\*]+?\b(tamd_\w+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from tengine_amd import capi
    assert os.path.exists(capi.LIB_PATH), "run __graft_entry__.build() first"
    L = ctypes.CDLL(capi.LIB_PATH)
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), "missing export %s" % n
    assert sorted(capi.EXPORTS) == names


def test_no_device_fails_loudly():
    """Without a GPU prerun must fail with an error, never fall back to a CPU path."""
    from helpers import conv_graph
    from tengine_amd import capi, tm2
    if capi.device_count() > 0:
        pytest.skip("GPU present")
    g, _ = conv_graph(0, 1, 16, 8, 8, 16, 1)
    with pytest.raises(capi.TamdError):
        capi.Graph(tm2.write_tm2(g))


def test_tm2_loader_rejects_garbage():
    from tengine_amd import capi
    L = capi.lib()
    assert not L.tamd_graph_load_tm2(b"\x02\x00" + b"\xff" * 64, 66)
    assert b"tm2" in L.tamd_last_error()


def test_product_never_imports_oracle():
    """tengine_amd/ must not reference oracle/ (the oracle is test infrastructure only)."""
    for dp, _, fs in os.walk(os.path.join(ROOT, "tengine_amd")):
        for f in fs:
            if f.endswith((".py", ".cc", ".hip", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "tg_oracle" not in src, f


# ---- host logic that needs no GPU: graph validation and the per-node support query -------------------------------------
def _prerun_error(g):
    from tengine_amd import capi, tm2
    with pytest.raises(capi.TamdError) as e:
        capi.Graph(tm2.write_tm2(g))
    return str(e.value)


def test_malformed_graphs_are_refused_at_prerun_not_over_read():
    """operator parameters that disagree with the constant payloads (a corrupt / hostile tmfile) fail prerun with a
    message naming the node -- before the planner indexes the payloads, and before the device is even touched"""
    from helpers import conv_graph, fc_graph
    g, _ = conv_graph(1, 1, 16, 8, 8, 32, 3, 1, 1)
    g.nodes[-1].params["group"] = 0
    assert "group" in _prerun_error(g)
    g, _ = conv_graph(1, 1, 16, 8, 8, 32, 3, 1, 1)
    g.nodes[-1].params["group"] = 3                           # does not divide 16 / 32
    assert "group" in _prerun_error(g)
    g, _ = conv_graph(1, 1, 16, 8, 8, 32, 3, 1, 1)
    g.nodes[-1].params["kernel_h"] = 5                         # weights are 3x3
    assert "weight size" in _prerun_error(g)
    g, _ = conv_graph(1, 1, 16, 8, 8, 32, 3, 1, 1)
    b = [t for t in g.tensors if t.name == "b"][0]
    b.data = b.data[:7]                                        # bias shorter than the output channels
    b.dims = [7]
    assert "bias" in _prerun_error(g)
    g, _ = conv_graph(1, 1, 16, 8, 8, 32, 3, 1, 1)
    g.nodes[-1].inputs = g.nodes[-1].inputs[:1]                # no weight tensor at all
    assert "weight" in _prerun_error(g)
    g, _ = fc_graph(1, 2, (24,), 10)
    w = [t for t in g.tensors if t.name == "w"][0]
    w.data = w.data.T.copy()                                   # [hidden][out]: the reference's own infer_shape refuses it too
    w.dims = [24, 10]
    assert "fc weight" in _prerun_error(g) or "weight" in _prerun_error(g)


def test_node_supported_query():
    """tamd_node_supported: what the Tengine plugin asks before it claims a subgraph (no payloads, no device)"""
    import ctypes as C
    from tengine_amd import capi
    L = capi.lib()

    class T(C.Structure):          # tamd_tensor_desc
        _fields_ = [("dtype", C.c_int), ("ttype", C.c_int), ("dim_num", C.c_int), ("dims", C.c_int * 8), ("data", C.c_void_p),
                    ("quant_num", C.c_int), ("scales", C.c_void_p), ("zero_points", C.c_void_p), ("name", C.c_char_p)]

    class N(C.Structure):          # tamd_node_desc
        _fields_ = [("op", C.c_int), ("input_num", C.c_int), ("inputs", C.c_void_p), ("output_num", C.c_int), ("outputs", C.c_void_p),
                    ("param", C.c_void_p), ("name", C.c_char_p)]

    def t(dtype, ttype, dims, q=1):
        d = T()
        d.dtype, d.ttype, d.dim_num, d.quant_num = dtype, ttype, len(dims), q
        for i, v in enumerate(dims):
            d.dims[i] = v
        return d

    def ask(op, ins, outs, param=None):
        n = N()
        n.op, n.input_num, n.output_num = op, len(ins), len(outs)
        n.param = C.cast(C.pointer(param), C.c_void_p) if param is not None else None
        return L.tamd_node_supported(C.byref(n), (T * len(ins))(*ins), len(ins), (T * len(outs))(*outs), len(outs))

    I8, U8, F32, VAR, CONST = 2, 3, 0, 1, 2
    conv = (C.c_int * 14)(3, 3, 1, 1, 1, 1, 1, 1, 1, 1, 16, 32, 1, 0)          # tamd_conv_param
    x, y = t(I8, VAR, [1, 16, 8, 8]), t(I8, VAR, [1, 32, 8, 8])
    assert ask(2, [x, t(I8, CONST, [32, 16, 3, 3], 32)], [y], conv) == 1
    assert ask(2, [x, t(I8, CONST, [32, 16, 5, 5], 32)], [y], conv) == 0       # weights disagree with the kernel size
    assert ask(2, [x, t(I8, CONST, [32, 16, 3, 3], 7)], [y], conv) == 0        # 7 scales for 32 channels
    grp = (C.c_int * 14)(3, 3, 1, 1, 1, 1, 1, 1, 1, 1, 16, 32, 3, 0)
    assert ask(2, [x, t(I8, CONST, [32, 16, 3, 3], 32)], [y], grp) == 0        # group 3 does not divide the channels
    big = (C.c_int * 14)(12, 12, 1, 1, 0, 0, 0, 0, 1, 1, 16, 32, 1, 0)
    assert ask(2, [t(I8, VAR, [1, 16, 20, 20]), t(I8, CONST, [32, 16, 12, 12], 32)], [t(I8, VAR, [1, 32, 9, 9])], big) == 0   # 144 taps
    assert ask(2, [t(U8, VAR, [1, 16, 8, 8]), t(U8, CONST, [32, 16, 3, 3], 32)], [t(U8, VAR, [1, 32, 8, 8])], conv) == 0     # uint8: per-tensor weights only
    fc = (C.c_int * 1)(10)
    assert ask(3, [t(I8, VAR, [2, 24]), t(I8, CONST, [10, 24], 10)], [t(I8, VAR, [2, 10])], fc) == 1
    assert ask(3, [t(I8, VAR, [2, 24]), t(I8, CONST, [24, 10], 10)], [t(I8, VAR, [2, 10])], fc) == 0      # transposed layout
    cat0 = (C.c_int * 1)(0)
    cat1 = (C.c_int * 1)(1)
    a, b = t(I8, VAR, [1, 8, 4, 4]), t(I8, VAR, [1, 24, 4, 4])
    assert ask(7, [a, b], [t(I8, VAR, [1, 32, 4, 4])], cat1) == 1              # any channel counts, any scales: rescaling copy
    assert ask(7, [a, a], [t(I8, VAR, [2, 8, 4, 4])], cat0) == 0               # batch axis
    elt = (C.c_int * 5)(9, 0, 0, 0, 0)                                          # an eltwise type the device does not run
    assert ask(6, [a, a], [a], elt) == 0
    assert ask(12, [a], [a]) == 1                                               # softmax int8: over the channel axis (NHWC on the device)
    assert ask(12, [a], [a], (C.c_int * 1)(2)) == 1                             # .. the spatial axes too since round 6 (strided addressing)
    assert ask(12, [a], [a], (C.c_int * 1)(0)) == 0                             # .. the batch axis stays on the CPU
    assert ask(12, [a], [a], (C.c_int * 1)(-3)) == 1
    assert ask(12, [t(I8, VAR, [2, 1000])], [t(I8, VAR, [2, 1000])], (C.c_int * 1)(-1)) == 1
    assert ask(12, [t(I8, VAR, [2, 5, 1000])], [t(I8, VAR, [2, 5, 1000])], (C.c_int * 1)(1)) == 1     # 3-D (a Reshape result: dense on the device), round 6
    assert ask(12, [t(I8, VAR, [2, 5, 1000])], [t(I8, VAR, [2, 5, 1000])], (C.c_int * 1)(2)) == 1
    assert ask(12, [t(I8, VAR, [1, 20000])], [t(I8, VAR, [1, 20000])]) == 0     # the axis must fit a wave's LDS slice
    assert ask(12, [t(F32, VAR, [1, 8], 0)], [t(F32, VAR, [1, 8], 0)]) == 1
    # uint8 / fp32 device tensors are dense NCHW: concat on any axis; int8 (NHWC blocks): channels -- and, since round 6, the PriorBox layout
    # [1][2][K][1] on axis 2 and tensors of other ranks (dense on the device)
    cat2 = (C.c_int * 1)(2)
    ua = t(U8, VAR, [1, 2, 100, 1])
    assert ask(7, [ua, ua], [t(U8, VAR, [1, 2, 200, 1])], cat2) == 1
    assert ask(7, [t(I8, VAR, [1, 2, 100, 1])] * 2, [t(I8, VAR, [1, 2, 200, 1])], cat2) == 1
    assert ask(7, [t(I8, VAR, [2, 8, 4, 4])] * 2, [t(I8, VAR, [2, 8, 8, 4])], cat2) == 0                  # rows of a convolution-stack map: CPU
    assert ask(7, [t(I8, VAR, [2, 30, 7])] * 2, [t(I8, VAR, [2, 30, 14])], cat2) == 1

    class PB(C.Structure):         # tamd_priorbox_param
        _fields_ = [("min_size_num", C.c_int), ("max_size_num", C.c_int), ("aspect_ratio_num", C.c_int), ("min_size", C.c_float * 8),
                    ("max_size", C.c_float * 8), ("aspect_ratio", C.c_float * 8), ("variance", C.c_float * 4), ("flip", C.c_int),
                    ("clip", C.c_int), ("image_h", C.c_int), ("image_w", C.c_int), ("step_h", C.c_float), ("step_w", C.c_float),
                    ("offset", C.c_float)]
    pb = PB()
    pb.min_size_num, pb.max_size_num, pb.aspect_ratio_num, pb.flip, pb.offset = 1, 1, 2, 1, 0.5
    feat, img = t(U8, VAR, [1, 64, 5, 5]), t(U8, VAR, [1, 3, 80, 80])
    assert ask(15, [feat, img], [t(U8, VAR, [1, 2, 600, 1])], pb) == 1
    assert ask(15, [feat, img], [t(U8, VAR, [2, 2, 600, 1])], pb) == 0           # batch > 1: undefined in the reference
    assert ask(15, [feat], [t(U8, VAR, [1, 2, 600, 1])], pb) == 0                # needs the image tensor too
    assert ask(15, [t(I8, VAR, [1, 64, 5, 5]), t(I8, VAR, [1, 3, 80, 80])], [t(I8, VAR, [1, 2, 600, 1])], pb) == 1   # int8 graphs too since round 6
    perm = (C.c_int * 4)(0, 2, 3, 1)
    assert ask(13, [t(I8, VAR, [1, 12, 5, 5])], [t(I8, VAR, [1, 5, 5, 12])], perm) == 1                              # the SSD head permute, int8 since round 6
    assert ask(13, [t(I8, VAR, [1, 12, 5, 5])], [t(I8, VAR, [1, 5, 12, 5])], (C.c_int * 4)(0, 3, 1, 2)) == 0
    assert ask(13, [t(F32, VAR, [1, 12, 5, 5], 0)], [t(F32, VAR, [1, 5, 5, 12], 0)], perm) == 0
    pb.max_size_num = 2
    assert ask(15, [feat, img], [t(U8, VAR, [1, 2, 600, 1])], pb) == 0           # max sizes must pair with min sizes (priorbox.c:48-61)


def test_launch_recorder_packs_arguments_like_the_code_object(tmp_path):
    """Direct dispatch replays recorded launches as AQL packets: the explicit argument segment the recorder packs
    (tengine_amd/csrc/launch_rec.h: rec_pack) must put every by-value argument at its natural alignment, in order -- checked on
    the host against a struct of the same members (tests/csrc/rec_pack_check.cc; no device involved)."""
    import shutil
    import subprocess
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.exists(hipcc) or shutil.which(hipcc)):
        pytest.skip("hipcc not available")
    exe = str(tmp_path / "rec_pack_check")
    subprocess.check_call([hipcc, "-std=c++17", "-I", os.path.join(ROOT, "tengine_amd", "csrc"),
                           os.path.join(ROOT, "tests", "csrc", "rec_pack_check.cc"), "-o", exe], stderr=subprocess.DEVNULL)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "mismatches 0" in out.stdout, out.stdout


def test_hidden_argument_offsets_are_read_from_code_object_metadata(tmp_path):
    """Direct dispatch places the hidden arguments (block counts, group sizes, grid dimensionality) where the kernel's own
    NT_AMDGPU_METADATA note lists them (tengine_amd/csrc/codeobj_meta.h: ELF note walk + MessagePack reader).
    Checked on the host: hipcc compiles two kernels for gfx950, the parser's offsets must be the ones `llvm-readelf --notes`
    prints, and truncated objects are refused without reading out of bounds (the check is built with the address sanitizer)."""
    import shutil
    import subprocess
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not (os.path.exists(hipcc) or shutil.which(hipcc)) or not os.path.exists(readelf) or not shutil.which("g++"):
        pytest.skip("hipcc / llvm-readelf / g++ not available")
    src = tmp_path / "k.hip"
    src.write_text("#include <hip/hip_runtime.h>\n"
                   "struct A { const float* x; float* y; int n; short s; };\n"
                   "__global__ void k1(A a) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < a.n) a.y[i] = a.x[i] * a.s; }\n"
                   "__global__ void k2(const int* p, int* q, int n, float f, double d, char c) { int i = blockIdx.y * gridDim.x * blockDim.x + blockIdx.x * blockDim.x"
                   " + threadIdx.x; if (i < n) q[i] = p[i] + (int)f + (int)d + c; }\n")
    co = str(tmp_path / "k.co")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "--cuda-device-only", "--no-gpu-bundle-output", "-c", str(src), "-o", co],
                          stderr=subprocess.DEVNULL)
    exe = str(tmp_path / "codeobj_meta_test")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fsanitize=address,undefined", os.path.join(ROOT, "tests", "csrc", "codeobj_meta_test.cc"), "-o", exe])
    out = subprocess.run([exe, co], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    got = {}
    for line in out.stdout.splitlines():
        w = line.split()
        got[w[0]] = {w[i]: [int(v) for v in w[i + 1:i + (4 if w[i] in ("block_count", "group_size", "remainder") else 2)]] for i in range(1, len(w)) if not w[i].lstrip("-").isdigit()}
    notes = subprocess.run([readelf, "--notes", co], capture_output=True, text=True).stdout
    # the note, kernel by kernel: offsets by value_kind
    want, cur, off = {}, None, None
    for line in notes.splitlines():
        t = line.strip()
        if t.startswith("- .offset:") or t.startswith(".offset:"):
            off = int(t.split()[-1])
        elif t.startswith(".value_kind:"):
            cur = cur if cur is not None else {}
            cur.setdefault(t.split()[-1], []).append(off)
        elif t.startswith(".symbol:"):
            want[t.split()[-1]] = cur or {}
            cur = None
    assert len(got) == 2 and set(got) == set(want), (got.keys(), want.keys())
    for sym, g_ in got.items():
        w_ = want[sym]
        assert g_["block_count"] == [w_["hidden_block_count_" + a][0] for a in "xyz"]
        assert g_["group_size"] == [w_["hidden_group_size_" + a][0] for a in "xyz"]
        assert g_["remainder"] == [w_["hidden_remainder_" + a][0] for a in "xyz"]
        assert g_["grid_dims"] == w_["hidden_grid_dims"]
        assert g_["unknown_pointer"] == [0]


def test_dependency_test_of_the_barrier_free_launches(tmp_path):
    """which launch may run beside which (tengine_amd/csrc/graph.h: access_of / access_overlap / step_conflict -- NCHW tensors,
    uint8 concat slices, int8 NHWC channel views, concat slots; RAW / WAR / WAW): checked on the host, no device involved
    (tests/csrc/step_conflict_check.cc)"""
    import shutil
    import subprocess
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.exists(hipcc) or shutil.which(hipcc)):
        pytest.skip("hipcc not available")
    exe = str(tmp_path / "step_conflict_check")
    subprocess.check_call([hipcc, "-std=c++17", "-I", os.path.join(ROOT, "tengine_amd", "csrc"), "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "csrc", "step_conflict_check.cc"), "-o", exe], stderr=subprocess.DEVNULL)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "0 failures" in out.stdout, out.stdout


def test_direct_dispatch_option_is_in_the_abi_and_size_guarded():
    """tamd_options grew a field (direct_dispatch): a caller compiled against the older, shorter struct still passes its own
    `size`, and the binding's struct matches the header's field order."""
    from tengine_amd import capi
    hdr = open(os.path.join(ROOT, "include", "tengine_amd.h")).read()
    body = hdr[hdr.index("typedef struct tamd_options {"):hdr.index("} tamd_options;")]
    fields = re.findall(r"^\s+(?:const\s+)?\w+\*?\s+(\w+);", body, re.M)
    assert fields == [f[0] for f in capi.Options._fields_], (fields, capi.Options._fields_)
    assert fields[-4:] == ["direct_dispatch", "keep_tensors", "u8_integer", "split_batch"] and ctypes.sizeof(capi.Options) == 40   # (8 + 8 ints: no padding added)


def test_environment_surface_is_the_documented_one():
    """the library reads 23 named TAMD_* variables + TAMD_PIN (csrc/env.h); every name and every TAMD_PIN key is
    listed in INTEGRATION.md section E AND set by some test or tool of this repository; experiment switches go through exp_env()
    only, which is getenv in -DTAMD_EXPERIMENTS builds alone."""
    import glob
    import re
    src = ""
    for f in glob.glob(os.path.join(ROOT, "tengine_amd", "csrc", "*")) + glob.glob(os.path.join(ROOT, "tengine_amd", "device", "*")):
        if f.endswith((".hip", ".cc", ".h")) and not f.endswith("env.h"):
            src += open(f).read()
    named = set(re.findall(r'(?:getenv|pg_env)\("(TAMD_[A-Z0-9_]+)"', src))
    pins = set(re.findall(r'tamd_pin(?:_int)?\("([a-z0-9_]+)"', src))
    exps = set(re.findall(r'(?:exp_env|env_int)\("(TAMD_[A-Z0-9_]+)"', src))
    assert len(named) <= 30 and "TAMD_PIN" not in named, sorted(named)      # (TAMD_PIN itself is read in env.h)
    assert not (named & exps), named & exps
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = doc[doc.index("## E. Environment"):doc.index("## F. ")]
    users = ""
    for f in glob.glob(os.path.join(ROOT, "tests", "*.py")) + glob.glob(os.path.join(ROOT, "tools", "*.py")) + [os.path.join(ROOT, "bench.py")] + \
            glob.glob(os.path.join(ROOT, "tengine_amd", "*.py")):
        if not f.endswith("test_abi.py"):
            users += open(f).read()
    for v in sorted(named):
        assert v in sec, "%s is read by the library but not documented in INTEGRATION.md section E" % v
        assert v in users, "%s is read by the library but no test / tool sets it" % v
    for k in sorted(pins):
        assert re.search(r"`%s[=`]" % k, sec), "TAMD_PIN key %s is not documented" % k
        assert re.search(r"\b%s\b" % k, users), "TAMD_PIN key %s is not set by any test / tool" % k
    for v in sorted(exps):
        assert v in sec, "experiment switch %s is not listed" % v
    # nothing else in the library looks at the environment
    other = set(re.findall(r'getenv\("([A-Z0-9_]+)"', src)) - named
    assert other <= {"TG_HIP_DEVICE", "TG_HIP_SCHEDULER", "TG_DEBUG_TIME", "ROCPROFILER_LIBRARY", "HSA_TOOLS_LIB", "ROCP_TOOL_LIBRARIES", "LD_PRELOAD"}, other
