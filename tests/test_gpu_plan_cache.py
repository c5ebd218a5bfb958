"""TAMD_PLAN_CACHE (plan_cache.hip; used by graph_plan*.hip / graph_u8.hip): the first prerun of a model measures its candidates and writes what it chose,
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
    # first line: what the choices were made for (architecture, library version, candidate counts); then "key <tab> choice"
    assert lines[0].startswith("#tamd-plan v2 gfx950 "), lines[0]
    assert any(ln.startswith(site) for ln in lines), lines[:5]
    assert all("\t" in ln for ln in lines[1:])
    out2, k2, ms2 = plan(g, x)
    assert k2 == k1
    for a, b in zip(out1, out2):
        assert np.array_equal(a, b)
    assert ms2 < 0.95 * ms1, (ms1, ms2)                  # nothing was timed the second time
    # a line whose value names no candidate: that site is measured again, the rest still comes from the file
    bad = [ln.split("\t")[0] + "\tno_such_kernel" if i == 1 else ln for i, ln in enumerate(lines)]
    cache.write_text("\n".join(bad) + "\n")
    out3, k3, _ = plan(g, x)
    assert len(k3) == len(k1)
    for a, b in zip(out1, out3):
        assert np.array_equal(a, b)


def test_plan_file_of_another_build_is_ignored_and_foreign_entries_survive_a_flush(tmp_path, monkeypatch):
    """a file whose header names another library version / candidate list is not trusted at all (the sites are measured
    again and the file is rewritten with this build's header); entries another process wrote into a file of THIS build are merged,
    not overwritten, and the file is replaced atomically (no temporary left behind)."""
    cache = tmp_path / "plan.txt"
    monkeypatch.setenv("TAMD_PLAN_CACHE", str(cache))
    g = models.build("mobilenet_v1", "int8", 1)
    x = models.synth_input(g, 3)
    cache.write_text("#tamd-plan v1 gfx942 something else\npwdw|conv2_1/sep|n1 112x112 k32 m1 f1 c8\t1,99\n")
    out1, k1, _ = plan(g, x)
    lines = cache.read_text().splitlines()
    assert lines[0].startswith("#tamd-plan v2 gfx950 ") and not any("1,99" in ln for ln in lines)
    cache.write_text("\n".join(lines + ["gemm|some_other_model_layer|1x8x8x8>8 k1x1 s1\tgemm_direct_i8"]) + "\n")
    g2 = models.build("resnet50", "int8", 1, device_only=True)
    out2, _, _ = plan(g2, models.synth_input(g2, 4))
    after = cache.read_text().splitlines()
    assert any(ln.startswith("gemm|some_other_model_layer|") for ln in after) and any(ln.startswith("gemm|res") for ln in after)
    assert not [f for f in tmp_path.iterdir() if ".tmp." in f.name]


def test_autotune_switch_is_read_at_every_prerun(tmp_path, monkeypatch):
    """round 6 (found by the suite's seed-7 file order): TAMD_AUTOTUNE / TAMD_FUSE_ELTWISE were read ONCE per process (function-local
    statics), so whichever test planned first decided for every later graph -- after tests/test_gpu_pgemm.py's TAMD_AUTOTUNE=0 case no plan
    file was written any more.  Both orders in one process: off (nothing timed, nothing recorded), then on (the file appears), then off."""
    g = models.build("mobilenet_v1", "int8", 2)
    x = models.synth_input(g, 5)
    outs = []
    for i, off in enumerate([True, False, True]):
        cache = tmp_path / ("plan%d.txt" % i)
        monkeypatch.setenv("TAMD_PLAN_CACHE", str(cache))
        if off:
            monkeypatch.setenv("TAMD_AUTOTUNE", "0")
        else:
            monkeypatch.delenv("TAMD_AUTOTUNE", raising=False)
        out, _, _ = plan(g, x)
        outs.append(out)
        assert cache.exists() == (not off), (i, off)
    for o in outs[1:]:
        for a, b in zip(outs[0], o):
            assert np.array_equal(a, b)
