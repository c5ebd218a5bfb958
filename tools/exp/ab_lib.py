#!/usr/bin/env python3
"""A/B of two BUILDS of libtengine_amd.so on one config's device-resident step, interleaved inside one box: each measurement is a
fresh process that imports a private copy of the tengine_amd package holding the build under test (a process can load the
library only once), plans from its own plan file, checks its bytes against the other build's and times the step as bench.py does.

usage: ab_lib.py model batch dtype iters rounds  NAME=/path/to/libtengine_amd.so  NAME=...   (NAME=product: the tree's own build)
       [env TAMD_U8_INT=1 etc. is inherited; AB_LAYERS=1 adds a per-launch table (isolated launches, HIP events) of two builds side by side]"""
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

WORKER = r'''
import hashlib, json, os, sys
sys.path.insert(0, os.environ["AB_PKG"])
from tengine_amd import capi, models, tm2
name, batch, dtype, iters = sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
g = models.build(name, dtype, batch, device_only=True)
x = models.synth_input(g, 3, {"uint8": tm2.DT_UINT8, "fp32": tm2.DT_FP32}.get(dtype, tm2.DT_INT8))
gr = capi.Graph(tm2.write_tm2(g), batch=batch, direct_dispatch=True)
gr.set_input(x)
out = gr.run()
h = hashlib.sha256(b"".join(o.tobytes() for o in out)).hexdigest()[:16]
gr.upload(); gr.sync()
gr.time_launches(max(3, iters // 10))
ts = [1e3 * gr.time_launches(iters) / iters for _ in range(3)]
layers = [(k["node"], k["kernel"], k["ms"] * 1e3) for k in gr.profile(max(3, iters // 10))] if os.environ.get("AB_LAYERS") else []
print(json.dumps({"us": min(ts), "sha": h, "launches": gr.kernel_num(), "prerun_ms": gr.prerun_ms(), "layers": layers}))
'''


def main():
    name, batch, dtype, iters, rounds = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5])
    tmp = tempfile.mkdtemp(prefix="ab_lib_")
    variants = []
    for spec in sys.argv[6:]:
        nm, _, path = spec.partition("=")
        pkg = os.path.join(tmp, nm)
        shutil.copytree(os.path.join(ROOT, "tengine_amd"), os.path.join(pkg, "tengine_amd"), ignore=shutil.ignore_patterns("obj", "__pycache__"))
        if path and path != "product":
            shutil.copyfile(path, os.path.join(pkg, "tengine_amd", "lib", "libtengine_amd.so"))
        variants.append((nm, pkg, os.path.join(tmp, nm + "_plan.txt")))
    res = {nm: [] for nm, _, _ in variants}
    sha = {}
    layers = {}
    for r in range(rounds):
        for nm, pkg, plan in variants:
            env = dict(os.environ, AB_PKG=pkg, TAMD_PLAN_CACHE=plan)
            p = subprocess.run([sys.executable, "-c", WORKER, name, batch, dtype, iters], env=env, capture_output=True, text=True, timeout=900)
            line = [l for l in p.stdout.splitlines() if l.startswith("{")]
            if p.returncode or not line:
                print("  %s: FAILED\n%s" % (nm, p.stderr[-800:]))
                continue
            j = json.loads(line[-1])
            res[nm].append(j["us"])
            sha[nm] = j["sha"]
            for node, kern, us in j.get("layers", []):
                key = (node, kern)
                layers.setdefault(nm, {})
                layers[nm][key] = min(us, layers[nm].get(key, 1e30))
    print("== %s %s b%s: us per step (direct dispatch), %d fresh processes per build, interleaved; min of 3 x %s steps each" % (name, dtype, batch, rounds, iters))
    for nm, _, _ in variants:
        v = sorted(res[nm])
        if v:
            print("  %-28s min %9.2f  median %9.2f  max %9.2f | output sha %s%s" % (nm, v[0], v[len(v) // 2], v[-1], sha.get(nm), "" if len(set(sha.values())) == 1 else "  (DIFFERS between builds)"))
    if layers and len(variants) == 2:          # AB_LAYERS=1: isolated launches (HIP events), min over the rounds, side by side
        a, b = variants[0][0], variants[1][0]
        print("  per launch, isolated (us): %-24s %-34s %8s | %-34s %8s" % ("node", a, "", b, ""))
        nodes = []
        for nm in (a, b):
            for (node, kern) in layers.get(nm, {}):
                if node not in nodes:
                    nodes.append(node)
        for node in nodes:
            ea = sorted((us, k) for (n, k), us in layers.get(a, {}).items() if n == node)
            eb = sorted((us, k) for (n, k), us in layers.get(b, {}).items() if n == node)
            if ea and eb:
                print("    %-38s %-34s %8.2f | %-34s %8.2f  %+6.1f%%" % (node[:38], ea[0][1][:34], ea[0][0], eb[0][1][:34], eb[0][0], 100 * (eb[0][0] / ea[0][0] - 1)))
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
