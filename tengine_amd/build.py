"""Builds the in-tree native artefacts with hipcc for gfx950 (cross-compiles without a GPU):

  tengine_amd/lib/libtengine_amd.so      HIP kernels + planner/executor + tm2 loader, C ABI of include/tengine_amd.h
  tengine_amd/lib/rccl_gather.bin        the C multi-GPU harness of INTEGRATION.md section F (tengine_amd/harness/rccl_gather.cpp)
  tengine_amd/lib/libtengine_hip_device.so   the Tengine device plugin (register_hip_device), only where the
                                         reference headers are present (it compiles against source/*.h; the
                                         prebuilt .so travels to the GPU box)

-ffp-contract=off and no fast-math are REQUIRED: the requantising epilogues must round exactly like the
reference's scalar C.
"""
import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

SOURCES = ["conv_igemm.hip", "conv_igemm2.hip", "conv_pgemm.hip", "conv_pgemm_w.hip", "gemm_direct.hip", "pw_stream.hip", "pw_rows.hip", "conv_first.hip", "conv_first_pool.hip", "dwconv.hip", "dwpw.hip", "pwdw.hip", "pwdw_slices.hip", "conv_direct.hip", "misc_kernels.hip", "u8_kernels.hip", "u8_conv_gemm.hip", "u8_conv_patch.hip", "u8_conv_small.hip", "u8i_kernels.hip", "conv_f32_mfma.hip", "winograd_f32.hip", "f32_kernels.hip", "graph.hip", "graph_infer.hip", "graph_plan.hip", "graph_plan_pairs.hip", "plan_cache.hip", "graph_exec.hip", "graph_pair.hip", "graph_u8.hip", "graph_f32.hip", "tm2_reader.cc", "direct.cc"]
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-fvisibility=hidden", "-Wall", "-Wno-unused-function", "-Wno-unused-variable", "-Wno-unused-value"]
# int8 GEMM kernels: MFMA accumulators in architectural VGPRs.  hipcc's heuristic keeps them in AccVGPRs and every value
# of the requantising epilogue then costs a v_accvgpr_read on top of its ~8 VALU instructions; the VGPR form also lowers
# the total register count of these kernels (pw_stream<2,4>: 112 -> 90, conv_igemm2 128x128: 232 -> 175)
VGPR_FORM = {"pw_stream.hip", "pw_rows.hip", "conv_igemm.hip", "conv_igemm2.hip", "conv_pgemm.hip", "conv_pgemm_w.hip", "conv_first.hip", "conv_first_pool.hip", "dwpw.hip", "gemm_direct.hip", "u8i_kernels.hip"}


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_core(verbose=False, force=False):
    os.makedirs(OBJDIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + \
              [os.path.join(ROOT, "include", "tengine_amd.h")]

    def one(src):
        path = os.path.join(CSRC, src)
        obj = os.path.join(OBJDIR, src + ".o")
        included = [os.path.join(CSRC, "pwdw.hip")] if src == "pwdw_slices.hip" else []      # (a source that includes another source)
        if not force and not _newer(obj, [path, os.path.abspath(__file__)] + headers + included):
            return obj, None
        cmd = [HIPCC] + FLAGS + (["-mllvm", "--amdgpu-mfma-vgpr-form"] if src in VGPR_FORM else []) + \
              (["-x", "hip"] if src.endswith(".cc") else []) + ["-c", path, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if verbose and r.stderr:
            sys.stderr.write(r.stderr)
        return obj, (r.stderr if r.returncode else None)

    objs, errs = [], []
    with cf.ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        for obj, err in ex.map(one, SOURCES):
            objs.append(obj)
            if err:
                errs.append(err)
    if errs:
        raise RuntimeError("hipcc failed:\n" + "\n".join(errs))
    lib = os.path.join(LIBDIR, "libtengine_amd.so")
    if force or _newer(lib, objs):
        subprocess.check_call([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib] + objs + ["-lhsa-runtime64"])
    return lib


def build_plugin(ref="/root/reference", verbose=False, force=False):
    """Tengine device plugin; needs the reference headers at build time only."""
    src = os.path.join(HERE, "device", "hip_device.cc")
    if not os.path.isdir(os.path.join(ref, "source")) or not os.path.exists(src):
        return None
    lib = os.path.join(LIBDIR, "libtengine_hip_device.so")
    if not force and not _newer(lib, [src, os.path.join(ROOT, "include", "tengine_amd.h")]):
        return lib
    # the one configured header the reference's internal headers include (source/defines.h.in)
    import re
    gen = os.path.join(LIBDIR, "gen")
    os.makedirs(gen, exist_ok=True)
    defines = re.sub(r"#cmakedefine (\w+)", r"#define \1", open(os.path.join(ref, "source", "defines.h.in")).read())
    open(os.path.join(gen, "defines.h"), "w").write(defines)
    cmd = ["g++", "-O2", "-std=c++14", "-fPIC", "-shared", "-fvisibility=hidden", "-I" + os.path.join(ref, "source"),
           "-I" + os.path.join(ref, "source", "operator", "prototype"), "-I" + gen,
           "-I" + os.path.join(ROOT, "include"), src, "-o", lib,
           "-L" + LIBDIR, "-ltengine_amd", "-Wl,-rpath,$ORIGIN"]
    subprocess.check_call(cmd)
    return lib


def build_harness(force=False):
    """tengine_amd/harness/rccl_gather.cpp -> tengine_amd/lib/rccl_gather.bin: the C-level multi-GPU program of INTEGRATION.md section F
    (RCCL broadcast of the tmfile bytes -> native loader -> all-gather of every output), product code since round 6 (it used to live
    under tools/exp although build() compiled it and tests/test_gpu_rccl_c.py ran it)."""
    src = os.path.join(HERE, "harness", "rccl_gather.cpp")
    out = os.path.join(LIBDIR, "rccl_gather.bin")
    core = os.path.join(LIBDIR, "libtengine_amd.so")
    if force or _newer(out, [src, core, os.path.join(ROOT, "include", "tengine_amd.h")]):
        subprocess.check_call([HIPCC, "--offload-arch=" + ARCH, "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-o", out, src,
                               "-L" + LIBDIR, "-ltengine_amd", "-lrccl", "-Wl,-rpath,$ORIGIN"])
    return out


def build_all(verbose=False, force=False):
    core = build_core(verbose, force)
    plugin = build_plugin(verbose=verbose, force=force)
    build_harness(force)
    return core, plugin


if __name__ == "__main__":
    print(build_all(verbose=True, force="--force" in sys.argv))
