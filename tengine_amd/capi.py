"""ctypes binding of the C ABI (include/tengine_amd.h) -- the only way Python reaches the device.

There is NO CPU fallback: if the native library is missing, or no HIP device is visible at prerun,
the calls raise (the CPU oracle lives in oracle/ and is test infrastructure only).
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libtengine_amd.so")

DT_FP32, DT_FP16, DT_INT8, DT_UINT8, DT_INT32 = 0, 1, 2, 3, 4
_NP = {DT_FP32: np.float32, DT_FP16: np.float16, DT_INT8: np.int8, DT_UINT8: np.uint8, DT_INT32: np.int32}


class Options(C.Structure):           # tamd_options
    _fields_ = [("dev_name", C.c_char_p), ("size", C.c_int), ("gpu_index", C.c_int), ("use_hip_graph", C.c_int), ("profile", C.c_int),
                ("direct_dispatch", C.c_int), ("keep_tensors", C.c_int), ("u8_integer", C.c_int), ("split_batch", C.c_int)]


class KernelInfo(C.Structure):        # tamd_kernel_info
    _fields_ = [("node", C.c_char * 64), ("kernel", C.c_char * 48), ("macs", C.c_double), ("bytes", C.c_double),
                ("ms", C.c_float)]


EXPORTS = [
    "tamd_device_count", "tamd_init", "tamd_shutdown", "tamd_last_error", "tamd_version", "tamd_op_supported", "tamd_node_supported",
    "tamd_graph_create", "tamd_graph_add_tensor", "tamd_graph_add_node", "tamd_graph_set_inputs",
    "tamd_graph_set_outputs", "tamd_graph_load_tm2", "tamd_graph_set_batch", "tamd_graph_prerun", "tamd_graph_halves",
    "tamd_graph_input_num", "tamd_graph_output_num", "tamd_graph_input_desc", "tamd_graph_output_desc",
    "tamd_graph_set_input", "tamd_graph_set_output", "tamd_graph_run", "tamd_graph_run_async", "tamd_graph_wait", "tamd_graph_inflight", "tamd_graph_upload_inputs",
    "tamd_graph_launch", "tamd_graph_sync", "tamd_graph_direct_packets", "tamd_graph_direct_meta_packets", "tamd_graph_download_outputs", "tamd_graph_output_device",
    "tamd_graph_direct_timestamps", "tamd_graph_direct_packet_name", "tamd_graph_stream", "tamd_graph_time_launches", "tamd_graph_prerun_ms", "tamd_graph_kernel_num", "tamd_graph_profile",
    "tamd_graph_read_tensor", "tamd_graph_tensor_num", "tamd_graph_tensor_desc", "tamd_graph_destroy",
]

_lib = None


class TamdError(RuntimeError):
    pass


def version():
    return lib().tamd_version().decode()


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TamdError("native library %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(tengine_amd has no CPU fallback)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        vp, ci = C.c_void_p, C.c_int
        L.tamd_last_error.restype = C.c_char_p
        L.tamd_version.restype = C.c_char_p
        L.tamd_graph_create.restype = vp
        L.tamd_graph_load_tm2.restype = vp
        L.tamd_graph_load_tm2.argtypes = [vp, C.c_size_t]
        L.tamd_graph_stream.restype = vp
        L.tamd_graph_direct_packet_name.restype = C.c_char_p
        L.tamd_graph_direct_packet_name.argtypes = [vp, ci]
        L.tamd_graph_direct_timestamps.argtypes = [vp, ci, C.POINTER(C.c_double), C.POINTER(C.c_double), ci]
        L.tamd_graph_prerun_ms.restype = C.c_double
        L.tamd_graph_prerun_ms.argtypes = [vp]
        for name, args in {
            "tamd_graph_set_batch": [vp, ci], "tamd_graph_prerun": [vp, C.POINTER(Options)],
            "tamd_graph_input_num": [vp], "tamd_graph_output_num": [vp], "tamd_graph_halves": [vp],
            "tamd_graph_input_desc": [vp, ci, C.POINTER(ci), C.POINTER(ci)],
            "tamd_graph_output_desc": [vp, ci, C.POINTER(ci), C.POINTER(ci), C.POINTER(C.c_float), C.POINTER(ci)],
            "tamd_graph_set_input": [vp, ci, vp, C.c_size_t], "tamd_graph_set_output": [vp, ci, vp, C.c_size_t],
            "tamd_graph_run": [vp], "tamd_graph_run_async": [vp], "tamd_graph_wait": [vp], "tamd_graph_inflight": [vp],
            "tamd_graph_upload_inputs": [vp], "tamd_graph_launch": [vp],
            "tamd_graph_sync": [vp], "tamd_graph_direct_packets": [vp], "tamd_graph_direct_meta_packets": [vp], "tamd_graph_download_outputs": [vp],
            "tamd_graph_output_device": [vp, ci, C.POINTER(vp), C.POINTER(C.c_size_t)],
            "tamd_graph_stream": [vp], "tamd_graph_time_launches": [vp, ci, C.POINTER(C.c_float)],
            "tamd_graph_kernel_num": [vp], "tamd_graph_profile": [vp, ci, C.POINTER(KernelInfo), ci],
            "tamd_graph_read_tensor": [vp, ci, vp, C.c_size_t], "tamd_graph_tensor_num": [vp],
            "tamd_graph_tensor_desc": [vp, ci, C.POINTER(ci), C.POINTER(ci)], "tamd_graph_destroy": [vp],
            "tamd_init": [ci], "tamd_op_supported": [ci, ci],
        }.items():
            getattr(L, name).argtypes = args
        L.tamd_graph_destroy.restype = None
        _lib = L
    return _lib


def _check(rc, what):
    if rc is None or (isinstance(rc, int) and rc < 0):
        raise TamdError("%s failed: %s" % (what, lib().tamd_last_error().decode()))
    return rc


def device_count():
    return lib().tamd_device_count()


class Graph:
    """A device graph loaded from tmfile bytes (same bytes the reference's `tengine:m` loader takes)."""

    def __init__(self, tm_bytes: bytes, batch=None, gpu_index=0, use_hip_graph=True, profile=False, direct_dispatch=False, keep_tensors=False, u8_integer=False, split_batch=0):
        L = lib()
        self._h = L.tamd_graph_load_tm2(tm_bytes, len(tm_bytes))
        if not self._h:
            raise TamdError("tamd_graph_load_tm2 failed: %s" % L.tamd_last_error().decode())
        if batch is not None:
            _check(L.tamd_graph_set_batch(self._h, batch), "set_batch")
        opt = Options(b"HIP", C.sizeof(Options), gpu_index, 1 if use_hip_graph else 0, 1 if profile else 0, 1 if direct_dispatch else 0,
                      1 if keep_tensors else 0, 1 if u8_integer else 0, int(split_batch))
        _check(L.tamd_graph_prerun(self._h, C.byref(opt)), "prerun")
        self._in, self._out = [], []
        for i in range(L.tamd_graph_output_num(self._h)):
            dims = (C.c_int * 8)()
            dt, sc, zp = C.c_int(), C.c_float(), C.c_int()
            nd = L.tamd_graph_output_desc(self._h, i, dims, C.byref(dt), C.byref(sc), C.byref(zp))
            arr = np.zeros([dims[k] for k in range(nd)], _NP[dt.value])
            self._out.append(arr)
            _check(L.tamd_graph_set_output(self._h, i, arr.ctypes.data, arr.nbytes), "set_output")

    def halves(self):
        """2: compiled as two half-batch device graphs behind this one handle (tamd_options.split_batch); 0: one launch list"""
        return lib().tamd_graph_halves(self._h)

    def input_desc(self, idx=0):
        dims = (C.c_int * 8)()
        dt = C.c_int()
        nd = lib().tamd_graph_input_desc(self._h, idx, dims, C.byref(dt))
        return [dims[k] for k in range(nd)], dt.value

    def set_input(self, arr, idx=0):
        arr = np.ascontiguousarray(arr)
        while len(self._in) <= idx:
            self._in.append(None)
        self._in[idx] = arr       # keep alive: the pointer is re-read at every run
        _check(lib().tamd_graph_set_input(self._h, idx, arr.ctypes.data, arr.nbytes), "set_input")

    def run(self):
        _check(lib().tamd_graph_run(self._h), "run")
        return [o.copy() for o in self._out]

    def run_noreturn(self):
        """tamd_graph_run() without copying the outputs again (they are in the arrays handed to set_output)"""
        _check(lib().tamd_graph_run(self._h), "run")

    def run_async(self, out_arrays=None):
        """tamd_graph_run_async(): submit the current input; `out_arrays` (one per output) receive THIS run's results at wait()"""
        if out_arrays is not None:
            for i, a in enumerate(out_arrays):
                _check(lib().tamd_graph_set_output(self._h, i, a.ctypes.data, a.nbytes), "set_output")
        _check(lib().tamd_graph_run_async(self._h), "run_async")

    def wait(self):
        _check(lib().tamd_graph_wait(self._h), "wait")

    def bind_default_outputs(self):
        """point the graph's output buffers back at the arrays run() / download() return"""
        for i, a in enumerate(self._out):
            _check(lib().tamd_graph_set_output(self._h, i, a.ctypes.data, a.nbytes), "set_output")

    def output_like(self):
        return [np.zeros_like(o) for o in self._out]

    def output_num(self):
        return lib().tamd_graph_output_num(self._h)

    def upload(self):
        _check(lib().tamd_graph_upload_inputs(self._h), "upload_inputs")

    def launch(self):
        _check(lib().tamd_graph_launch(self._h), "launch")

    def sync(self):
        _check(lib().tamd_graph_sync(self._h), "sync")

    def download(self):
        _check(lib().tamd_graph_download_outputs(self._h), "download_outputs")
        return [o.copy() for o in self._out]

    def time_launches(self, iters):
        ms = C.c_float()
        _check(lib().tamd_graph_time_launches(self._h, iters, C.byref(ms)), "time_launches")
        return ms.value

    def output_device(self, idx=0):
        p, n = C.c_void_p(), C.c_size_t()
        _check(lib().tamd_graph_output_device(self._h, idx, C.byref(p), C.byref(n)), "output_device")
        return p.value, n.value

    def stream(self):
        return lib().tamd_graph_stream(self._h)

    def prerun_ms(self):
        """wall milliseconds tamd_graph_prerun took (planning incl. autotune, capture, direct-dispatch programs)"""
        return lib().tamd_graph_prerun_ms(self._h)

    def direct_packets(self):
        """AQL packets per launch() when direct dispatch is active, else 0"""
        return lib().tamd_graph_direct_packets(self._h)

    def direct_timestamps(self, passes=200):
        """the directly dispatched pass under the HSA runtime's dispatch profiling: [(kernel symbol, mean us, mean gap to the next packet us)]"""
        n = lib().tamd_graph_direct_packets(self._h)
        dur, gap = (C.c_double * n)(), (C.c_double * n)()
        _check(lib().tamd_graph_direct_timestamps(self._h, passes, dur, gap, n), "direct_timestamps")
        return [(lib().tamd_graph_direct_packet_name(self._h, i).decode(), dur[i], gap[i]) for i in range(n)]

    def direct_meta_packets(self):
        return lib().tamd_graph_direct_meta_packets(self._h)

    def kernel_num(self):
        """number of compute launches of one forward pass"""
        return lib().tamd_graph_kernel_num(self._h)

    def profile(self, iters=10):
        n = lib().tamd_graph_kernel_num(self._h)
        arr = (KernelInfo * n)()
        _check(lib().tamd_graph_profile(self._h, iters, arr, n), "profile")
        return [dict(node=k.node.decode(), kernel=k.kernel.decode(), macs=k.macs, bytes=k.bytes, ms=k.ms) for k in arr]

    def read_tensor(self, idx):
        dims = (C.c_int * 8)()
        dt = C.c_int()
        nd = _check(lib().tamd_graph_tensor_desc(self._h, idx, dims, C.byref(dt)), "tensor_desc")
        arr = np.zeros([dims[k] for k in range(nd)], _NP[dt.value])
        _check(lib().tamd_graph_read_tensor(self._h, idx, arr.ctypes.data, arr.nbytes), "read_tensor")
        return arr

    def close(self):
        if self._h:
            lib().tamd_graph_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
