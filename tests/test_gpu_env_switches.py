"""The named runtime switches no other test sets (class 1; tests/test_abi.py checks that every variable the
library reads is set somewhere): each is flipped on a small graph inside a fresh process and must leave the bytes alone."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
from helpers import conv_graph, eltwise_relu_graph
from oracle import oracle
from tengine_amd import capi, tm2
g, x = eltwise_relu_graph(9, 2, 64, 14, 14, True)
want = oracle.run_graph(g, x)[0]
gr = capi.Graph(tm2.write_tm2(g), direct_dispatch=True)
gr.set_input(x)
got = gr.run()[0].reshape(want.shape)
print("KERNELS", " ".join(k["kernel"] for k in gr.profile(1)))
print("PACKETS", gr.direct_packets())
for t in range(len(g.tensors)):
    pass
gr.close()
print("SAME", bool(np.array_equal(got, want)))
try:
    capi.Graph(b"not a tmfile")
except Exception as e:
    print("REFUSED")
'''


def run(env):
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, "-c", SCRIPT % (ROOT, os.path.join(ROOT, "tests"))], env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "SAME True" in r.stdout, r.stdout
    return r.stdout, r.stderr


def test_debug_and_verbose_only_talk():
    out0, err0 = run({})
    out, err = run({"TAMD_DEBUG": "1", "TAMD_VERBOSE": "1"})
    assert "[tamd]" in err and "[tamd]" not in err0            # the plan-time narration
    assert "tengine_amd:" in err                                # the refused garbage file's error, printed as it is recorded
    assert "REFUSED" in out and "REFUSED" in out0


def test_fusion_opt_out_and_private_buffers_keep_the_bytes():
    out0, _ = run({})
    assert "+eltwise" in out0
    out, _ = run({"TAMD_FUSE_ELTWISE": "0"})                    # conv, eltwise, relu as separate launches
    assert "+eltwise" not in out and "eltwise" in out
    run({"TAMD_POOL": "0"})                                     # every intermediate tensor keeps its own buffer


def test_direct_dispatch_guards():
    out0, _ = run({})
    assert int(out0.split("PACKETS")[1].split()[0]) > 0
    # a profiler's environment sends the graph back to hipGraph replay; TAMD_DIRECT_UNDER_TOOLS=1 keeps the direct path
    out, _ = run({"LD_PRELOAD": "/nonexistent/librocprof-not-there.so"})      # (ld.so ignores a preload it cannot find; the name is what the guard looks at)
    assert int(out.split("PACKETS")[1].split()[0]) == 0
    out, _ = run({"LD_PRELOAD": "/nonexistent/librocprof-not-there.so", "TAMD_DIRECT_UNDER_TOOLS": "1"})
    assert int(out.split("PACKETS")[1].split()[0]) > 0
    run({"TAMD_DIRECT_TIMEOUT_S": "5"})                         # a shorter deadline changes nothing on a healthy queue
    run({"TAMD_H2H_TRACE": "1"})                                # the host-side anatomy counters of blocking runs
