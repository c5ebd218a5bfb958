"""TEST INFRASTRUCTURE ONLY -- ctypes driver for the real reference library built by
oracle/build_ref.py (oracle/_ref/libtengine-lite.so).  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this module; the product (tengine_amd/) never does.

Drives the reference exactly like its own examples do (examples/tm_classification_int8.c:68-72,
115-123,154-164): create_graph(NULL,"tengine:m",buf,size) -> set_tensor_shape/buffer ->
prerun_graph_multithread(opt{num_thread,cluster,precision,affinity}) -> run_graph(graph,1).
`precision` must match the model dtype or depthwise int8 silently runs the
fp32 kernel (conv_dw_hcl_x86.c:485-493).
"""
import ctypes as C
import os
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_ref", "libtengine-lite.so")

MODE_FP32, MODE_UINT8, MODE_INT8 = 0, 3, 4
DT_FP32, DT_FP16, DT_INT8, DT_UINT8, DT_INT32 = 0, 1, 2, 3, 4
_NP = {DT_FP32: np.float32, DT_INT8: np.int8, DT_UINT8: np.uint8, DT_INT32: np.int32, DT_FP16: np.float16}


class _Opt(C.Structure):   # struct options, source/api/c_api.h:153-159
    _fields_ = [("num_thread", C.c_int), ("cluster", C.c_int), ("precision", C.c_int), ("affinity", C.c_uint64)]


_lib = None


def available():
    return os.path.exists(LIB_PATH)


def lib():
    global _lib
    if _lib is None:
        if not available():
            raise FileNotFoundError("%s missing: run `python oracle/build_ref.py` where /root/reference exists" % LIB_PATH)
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        vp, ci = C.c_void_p, C.c_int
        L.init_tengine.restype = ci
        L.create_graph.restype = vp
        L.create_context.restype = vp
        L.create_context.argtypes = [C.c_char_p, ci]
        L.set_context_device.restype = ci
        L.set_context_device.argtypes = [vp, C.c_char_p, vp, C.c_size_t]
        L.load_tengine_plugin.restype = ci
        L.load_tengine_plugin.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p]
        L.prerun_graph_multithread.restype = ci
        L.prerun_graph_multithread.argtypes = [vp, _Opt]
        L.run_graph.restype = ci
        L.run_graph.argtypes = [vp, ci]
        L.postrun_graph.restype = ci
        L.postrun_graph.argtypes = [vp]
        L.destroy_graph.restype = ci
        L.destroy_graph.argtypes = [vp]
        L.get_graph_input_tensor.restype = vp
        L.get_graph_input_tensor.argtypes = [vp, ci, ci]
        L.get_graph_output_tensor.restype = vp
        L.get_graph_output_tensor.argtypes = [vp, ci, ci]
        L.get_graph_output_node_number.restype = ci
        L.get_graph_output_node_number.argtypes = [vp]
        L.get_graph_tensor.restype = vp
        L.get_graph_tensor.argtypes = [vp, C.c_char_p]
        L.set_tensor_shape.restype = ci
        L.set_tensor_shape.argtypes = [vp, C.POINTER(ci), ci]
        L.get_tensor_shape.restype = ci
        L.get_tensor_shape.argtypes = [vp, C.POINTER(ci), ci]
        L.set_tensor_buffer.restype = ci
        L.set_tensor_buffer.argtypes = [vp, vp, ci]
        L.get_tensor_buffer.restype = vp
        L.get_tensor_buffer.argtypes = [vp]
        L.get_tensor_buffer_size.restype = ci
        L.get_tensor_buffer_size.argtypes = [vp]
        L.get_tensor_data_type.restype = ci
        L.get_tensor_data_type.argtypes = [vp]
        if L.init_tengine() != 0:
            raise RuntimeError("init_tengine failed")
        _lib = L
    return _lib


class RefGraph:
    """One reference graph created from tmfile bytes (`tengine:m`), CPU device unless `device` given."""

    def __init__(self, tm_bytes: bytes, mode=MODE_INT8, threads=1, device=None, dev_opt=None):
        L = lib()
        # load_mem aliases the caller's buffer for const tensors AND unload_graph() sys_free()s it
        # (tm2_serializer.c:915-936, :938-960), so hand it a malloc'd copy that the graph then owns.
        libc = C.CDLL(None)
        libc.malloc.restype = C.c_void_p
        libc.malloc.argtypes = [C.c_size_t]
        self._buf = libc.malloc(len(tm_bytes))
        C.memmove(self._buf, tm_bytes, len(tm_bytes))
        self.ctx = None
        if device:
            self.ctx = L.create_context(b"tamd", 1)
            self._dev_opt = dev_opt      # keep alive: the context stores the pointer
            rc = L.set_context_device(self.ctx, device.encode(),
                                      None if dev_opt is None else C.cast(C.pointer(dev_opt), C.c_void_p),
                                      0 if dev_opt is None else C.sizeof(dev_opt))
            if rc != 0:
                raise RuntimeError("set_context_device(%s) failed" % device)
        self.g = self._create(L, len(tm_bytes))
        if not self.g:
            raise RuntimeError("create_graph failed")
        self.mode, self.threads = mode, threads
        self._inputs = []
        self.prerun_done = False

    def _create(self, L, size):
        # create_graph(ctx, "tengine:m", const void* buf, int size) -- variadic tail (c_api.c:399-421)
        f = L.create_graph
        f.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int]
        return f(self.ctx, b"tengine:m", C.c_void_p(self._buf), size)

    def set_input(self, arr: np.ndarray, idx=0):
        L = lib()
        t = L.get_graph_input_tensor(self.g, idx, 0)
        dims = (C.c_int * arr.ndim)(*arr.shape)
        if L.set_tensor_shape(t, dims, arr.ndim) != 0:
            raise RuntimeError("set_tensor_shape failed")
        arr = np.ascontiguousarray(arr)
        self._inputs.append(arr)
        if L.set_tensor_buffer(t, arr.ctypes.data, arr.nbytes) != 0:
            raise RuntimeError("set_tensor_buffer failed")

    def prerun(self):
        L = lib()
        rc = L.prerun_graph_multithread(self.g, _Opt(self.threads, 0, self.mode, 0))
        if rc != 0:
            raise RuntimeError("prerun_graph_multithread failed (%d)" % rc)
        self.prerun_done = True

    def run(self):
        if not self.prerun_done:
            self.prerun()
        rc = lib().run_graph(self.g, 1)
        if rc != 0:
            raise RuntimeError("run_graph failed (%d)" % rc)

    def _tensor_np(self, t):
        L = lib()
        dims = (C.c_int * 8)()
        nd = L.get_tensor_shape(t, dims, 8)
        shape = [dims[i] for i in range(nd)]
        dt = _NP[L.get_tensor_data_type(t)]
        n = L.get_tensor_buffer_size(t)
        p = L.get_tensor_buffer(t)
        raw = (C.c_char * n).from_address(p)
        return np.frombuffer(raw, dtype=dt, count=int(np.prod(shape))).reshape(shape).copy()

    def outputs(self):
        L = lib()
        n = L.get_graph_output_node_number(self.g)
        return [self._tensor_np(L.get_graph_output_tensor(self.g, i, 0)) for i in range(n)]

    def tensor(self, name: str):
        t = lib().get_graph_tensor(self.g, name.encode())
        return None if not t else self._tensor_np(t)

    def time_run(self, iters=5, warmup=1):
        for _ in range(warmup):
            self.run()
        ts = []
        for _ in range(iters):
            t0 = time.perf_counter()
            self.run()
            ts.append(time.perf_counter() - t0)
        return ts

    def close(self):
        L = lib()
        if self.g:
            if self.prerun_done:
                L.postrun_graph(self.g)
            L.destroy_graph(self.g)
            self.g = None


def run_model(tm_bytes, x, mode=MODE_INT8, threads=1):
    g = RefGraph(tm_bytes, mode, threads)
    g.set_input(x)
    g.run()
    out = g.outputs()
    g.close()
    return out
