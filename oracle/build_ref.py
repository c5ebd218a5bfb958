#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- builds the *unmodified* reference (OAID/Tengine, tengine-lite)
CPU library from the sources where they lie (default /root/reference) into oracle/_ref/.

Why a hand-written recipe: the task forbids running the reference's own build system (CMake).
The reference's CMake does exactly three things for the CPU path, all restated here:
  1. globs C sources (source/CMakeLists.txt:136-177, source/device/cpu/CMakeLists.txt:93-146),
  2. derives five registry headers from source-file *names*
     (cmake/registry.cmake:2-40: register_<basename>[_op]() lists) plus defines.h
     (source/defines.h.in),
  3. compiles with `-O3 -DNDEBUG -std=gnu99 -mfma -mf16c -fopenmp`
     (source/device/cpu/CMakeLists.txt:296-299; FMA contraction matters for the fp32/uint8 paths).
Outputs go ONLY to oracle/_ref/ (git-ignored, shipped to the GPU box as a prebuilt .so):
  oracle/_ref/libtengine-lite.so   all symbols visible (== -DTENGINE_ENABLE_ALL_SYMBOL=ON,
                                   source/CMakeLists.txt:324-327) so a device plugin can be loaded
                                   with load_tengine_plugin() (source/api/plugin.c:88-159)
  oracle/_ref/gen/                 generated registries + defines.h
  oracle/_ref/models/              the structure-only model files tm_benchmark runs (benchmark/models/*_benchmark.tmfile,
                                   10 - 80 KB each, no weights): test data for tests/test_reference_benchmark_files.py, which
                                   must also run on the GPU box where /root/reference does not exist
No reference source is copied into this repository.
"""
import concurrent.futures as cf
import glob
import hashlib
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
GEN = os.path.join(OUT, "gen")
OBJ = os.path.join(OUT, "obj")

CFLAGS = ["-O3", "-DNDEBUG", "-std=gnu99", "-mfma", "-mf16c", "-fopenmp", "-fPIC",
          "-fdata-sections", "-ffunction-sections", "-w"]


def _registry(template, target, reg_lead, del_lead, back, files):
    names = [os.path.splitext(os.path.basename(f))[0] for f in files]
    reg_def = "".join("extern int %s%s%s();\n" % (reg_lead, n, back) for n in names)
    del_def = "".join("extern int %s%s%s();\n" % (del_lead, n, back) for n in names)
    reg_cal = "".join("    ret = %s%s%s();\n    if(0 != ret) { TLOG_ERR(\"Tengine FATAL: Call %%s failed(%%d).\\n\", \"%s%s%s()\", ret); }\n"
                      % (reg_lead, n, back, reg_lead, n, back) for n in names)
    del_cal = "".join("    ret = %s%s%s();\n" % (del_lead, n, back) for n in names)
    text = open(template).read()
    text = (text.replace("@_GEN_REG_DEF_STR@", reg_def).replace("@_GEN_DEL_DEF_STR@", del_def)
            .replace("@_GEN_REG_CAL_STR@", reg_cal).replace("@_GEN_DEL_CAL_STR@", del_cal))
    os.makedirs(os.path.dirname(target), exist_ok=True)
    if not os.path.exists(target) or open(target).read() != text:
        open(target, "w").write(text)


def build(ref="/root/reference", jobs=None, verbose=False):
    src = os.path.join(ref, "source")
    if not os.path.isdir(src):
        raise FileNotFoundError("reference tree not found at %s" % ref)
    os.makedirs(GEN, exist_ok=True)
    os.makedirs(OBJ, exist_ok=True)
    g = lambda pat: sorted(glob.glob(os.path.join(src, pat)))

    # ---- 1. source lists (same globs as the reference CMake, x86 target) ----
    op_dirs = sorted(d for d in os.listdir(os.path.join(src, "device/cpu/op"))
                     if os.path.isdir(os.path.join(src, "device/cpu/op", d)))
    cpu_ref, cpu_x86, cpu_reg = [], [], []
    for d in op_dirs:
        cpu_ref += g("device/cpu/op/%s/*.c" % d)
        cpu_x86 += g("device/cpu/op/%s/x86/*.c" % d)
        cpu_reg += g("device/cpu/op/%s/%s_ref.c" % (d, d))
        cpu_reg += g("device/cpu/op/%s/x86/*_hcl_x86.c" % d)
    proto = g("operator/prototype/*.c")
    tm2_srl = g("serializer/tmfile/*.c")
    tm2_ops = g("serializer/tmfile/op/*.c")
    sources = (g("api/*.c") + g("device/*.c") + g("device/cpu/*.c") + cpu_ref + cpu_x86
               + g("operator/*.c") + proto + g("scheduler/*.c") + g("serializer/*.c") + tm2_srl
               + tm2_ops + g("executer/*.c") + g("graph/*.c") + g("module/*.c")
               + g("optimizer/*.c") + g("system/*.c") + g("utility/*.c"))

    # ---- 2. generated registries (cmake/registry.cmake restated) ----
    defines = open(os.path.join(src, "defines.h.in")).read()
    defines = re.sub(r"#cmakedefine (\w+)", r"#define \1", defines)
    dpath = os.path.join(GEN, "defines.h")
    if not os.path.exists(dpath) or open(dpath).read() != defines:
        open(dpath, "w").write(defines)
    _registry(os.path.join(src, "device/register.h.in"), os.path.join(GEN, "device/register.h"),
              "register_", "unregister_", "", [os.path.join(src, "device/cpu/cpu_device.c")])
    _registry(os.path.join(src, "device/cpu/cpu_ops.h.in"), os.path.join(GEN, "device/cpu/cpu_ops.h"),
              "register_", "unregister_", "_op", cpu_reg)
    _registry(os.path.join(src, "operator/prototype.h.in"), os.path.join(GEN, "operator/prototype.h"),
              "register_", "unregister_", "_op", proto)
    _registry(os.path.join(src, "serializer/register.h.in"), os.path.join(GEN, "serializer/register.h"),
              "register_", "unregister_", "", tm2_srl)
    _registry(os.path.join(src, "serializer/tmfile/tm2_ops.h.in"),
              os.path.join(GEN, "serializer/tmfile/tm2_ops.h"), "register_", "unregister_", "_op", tm2_ops)

    inc = [src, GEN, os.path.join(src, "device"), os.path.join(GEN, "device/cpu"),
           os.path.join(src, "device/cpu"), os.path.join(src, "operator/prototype"),
           os.path.join(src, "serializer"), os.path.join(GEN, "serializer")]
    iflags = ["-I" + p for p in inc]

    # ---- 3. compile (incremental by mtime) + link ----
    def one(path):
        rel = os.path.relpath(path, src)
        o = os.path.join(OBJ, hashlib.md5(rel.encode()).hexdigest()[:10] + "_" + os.path.basename(path) + ".o")
        if os.path.exists(o) and os.path.getmtime(o) >= os.path.getmtime(path):
            return o, None
        r = subprocess.run(["gcc"] + CFLAGS + iflags + ["-c", path, "-o", o], capture_output=True, text=True)
        return o, (r.stderr if r.returncode else None)

    objs, errs = [], []
    with cf.ThreadPoolExecutor(max_workers=jobs or os.cpu_count() or 4) as ex:
        for o, err in ex.map(one, sources):
            objs.append(o)
            if err:
                errs.append(err)
    if errs:
        raise RuntimeError("reference compile failed:\n" + "\n".join(errs[:5]))
    lib = os.path.join(OUT, "libtengine-lite.so")
    stale = (not os.path.exists(lib)) or any(os.path.getmtime(o) > os.path.getmtime(lib) for o in objs)
    if stale:
        rsp = os.path.join(OBJ, "link.rsp")
        open(rsp, "w").write("\n".join(objs))
        subprocess.check_call(["gcc", "-shared", "-fopenmp", "-o", lib, "@" + rsp, "-ldl", "-lm", "-lpthread"])
    if verbose:
        print("reference lib: %s (%d objects)" % (lib, len(objs)))
    copy_benchmark_models(ref)
    return lib


def copy_benchmark_models(ref="/root/reference"):
    """tm_benchmark's model files (benchmark/tm_benchmark.cc:250-289) -> oracle/_ref/models/ (git-ignored, travels with the prebuilt library)"""
    import shutil
    dst = os.path.join(OUT, "models")
    os.makedirs(dst, exist_ok=True)
    for f in sorted(glob.glob(os.path.join(ref, "benchmark", "models", "*_benchmark.tmfile"))):
        t = os.path.join(dst, os.path.basename(f))
        if not os.path.exists(t) or os.path.getmtime(t) < os.path.getmtime(f):
            shutil.copyfile(f, t)
    return dst


def build_tm_benchmark(ref="/root/reference", verbose=False):
    """INTEGRATION.md route B made literal: the reference's own `benchmark/tm_benchmark.cc`, UNMODIFIED, linked against the
    unmodified reference objects with our device compiled IN-TREE -- i.e. what adding `source/device/hip/` to the
    reference's CMake would produce: `device/register.h` regenerated from the two file names cpu_device.c + hip_device.cc
    (cmake/registry.cmake:11-31), api/c_api.c recompiled against it, hip_device.cc compiled as part of the library.
    Output: oracle/_ref/tm_benchmark_hip (`-d HIP` selects the device, `-m file.tmfile -f 2` runs a quantised tmfile)."""
    src = os.path.join(ref, "source")
    repo = os.path.dirname(HERE)
    dev_src = os.path.join(repo, "tengine_amd", "device", "hip_device.cc")
    core = os.path.join(repo, "tengine_amd", "lib", "libtengine_amd.so")
    lib = build(ref, verbose=verbose)
    if not (os.path.exists(dev_src) and os.path.exists(core)):
        return None
    gen2 = os.path.join(OUT, "gen_hip")
    _registry(os.path.join(src, "device/register.h.in"), os.path.join(gen2, "device/register.h"), "register_", "unregister_", "",
              [os.path.join(src, "device/cpu/cpu_device.c"), dev_src])
    inc = [gen2, src, GEN, os.path.join(src, "device"), os.path.join(GEN, "device/cpu"), os.path.join(src, "device/cpu"),
           os.path.join(src, "operator/prototype"), os.path.join(src, "serializer"), os.path.join(GEN, "serializer")]
    iflags = ["-I" + p for p in inc]
    capi_o = os.path.join(OBJ, "hiptree_c_api.o")
    subprocess.check_call(["gcc"] + CFLAGS + iflags + ["-c", os.path.join(src, "api/c_api.c"), "-o", capi_o])
    dev_o = os.path.join(OBJ, "hiptree_hip_device.o")
    subprocess.check_call(["g++", "-O2", "-std=c++14", "-fPIC", "-w", "-I" + src, "-I" + os.path.join(src, "operator/prototype"), "-I" + GEN,
                           "-I" + os.path.join(repo, "include"), "-c", dev_src, "-o", dev_o])
    objs = [o for o in open(os.path.join(OBJ, "link.rsp")).read().split("\n") if o and not o.endswith("_c_api.c.o")]
    exe = os.path.join(OUT, "tm_benchmark_hip")
    rsp = os.path.join(OBJ, "link_bench.rsp")
    open(rsp, "w").write("\n".join(objs + [capi_o, dev_o]))
    bench = os.path.join(ref, "benchmark")
    # tm_benchmark.cc includes "tengine/c_api.h" (the installed layout, source/CMakeLists.txt install rules): a symlink farm
    # under _ref/ stands in for the install step
    incdir = os.path.join(gen2, "include", "tengine")
    os.makedirs(incdir, exist_ok=True)
    for h in ("c_api.h", "defines.h"):
        link = os.path.join(incdir, h)
        target = os.path.join(src, "api", h) if h == "c_api.h" else os.path.join(GEN, h)
        if os.path.lexists(link):
            os.remove(link)
        os.symlink(target, link)
    subprocess.check_call(["g++", "-O2", "-std=c++11", "-w", "-fopenmp", "-I" + os.path.join(gen2, "include"), "-I" + src, "-I" + os.path.join(bench, "common"),
                           os.path.join(bench, "tm_benchmark.cc"), os.path.join(bench, "common", "timer.cc"), "@" + rsp,
                           "-L" + os.path.dirname(core), "-ltengine_amd", "-Wl,-rpath,$ORIGIN/../../tengine_amd/lib", "-ldl", "-lm", "-lpthread",
                           "-o", exe])
    if verbose:
        print("unmodified tm_benchmark with the HIP device in-tree: %s" % exe)
    return exe


if __name__ == "__main__":
    build(ref=sys.argv[1] if len(sys.argv) > 1 else "/root/reference", verbose=True)
    build_tm_benchmark(ref=sys.argv[1] if len(sys.argv) > 1 else "/root/reference", verbose=True)
