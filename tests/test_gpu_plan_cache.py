"""TAMD_PLAN_CACHE (graph.hip / graph_u8.hip): the first prerun of a model measures its candidates and writes what it chose,
later preruns take the recorded choices without launching anything -- same kernels, same bytes, a shorter prerun; a line that
names nothing known is ignored (the site is measured again)."""
import numpy as np
import pytest

from tengine_amd import capi, models, tm2

pytestmark = pytest.mark.gpu


def plan(g, x):
    gr = capi.Graph(tm2.write_tm2(g))
    gr.set_input(x)
    out = [o.copy() for o in gr.run()]
    kernels = [k["kernel"] for k in gr.profile(1)]
    ms = gr.prerun_ms()
    gr.close()
    return out, kernels, ms


@pytest.mark.parametrize("name,dtype,kw,site", [("yolov3_tiny", "uint8", dict(res=160), "u8conv|"), ("mobilenet_v1", "int8", dict(), "pwdw|"),
                                                ("resnet50", "int8", dict(device_only=True), "gemm|")])
def test_second_prerun_takes_the_recorded_choices(name, dtype, kw, site, tmp_path, monkeypatch):
    cache = tmp_path / "plan.txt"
    monkeypatch.setenv("TAMD_PLAN_CACHE", str(cache))
    g = models.build(name, dtype, 2, **kw)
    x = models.synth_input(g, 9, tm2.DT_UINT8 if dtype == "uint8" else tm2.DT_INT8)
    out1, k1, ms1 = plan(g, x)
    lines = cache.read_text().splitlines()
    assert any(ln.startswith(site) for ln in lines), lines[:5]
    assert all("\t" in ln for ln in lines)
    out2, k2, ms2 = plan(g, x)
    assert k2 == k1
    for a, b in zip(out1, out2):
        assert np.array_equal(a, b)
    assert ms2 < 0.95 * ms1, (ms1, ms2)                  # nothing was timed the second time
    # a line whose value names no candidate: that site is measured again, the rest still comes from the file
    bad = [ln.split("\t")[0] + "\tno_such_kernel" if i == 0 else ln for i, ln in enumerate(lines)]
    cache.write_text("\n".join(bad) + "\n")
    out3, k3, _ = plan(g, x)
    assert len(k3) == len(k1)
    for a, b in zip(out1, out3):
        assert np.array_equal(a, b)
