"""Host-side pieces of bench.py and the evidence tooling that need no GPU: how launches are grouped into kernel families for the
`roofline` object, how the committed PMC summaries are matched to a family, and that the per-config table of profiles/README.md
can be re-derived from the committed files."""
import importlib.util
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_kernel_families():
    b = _bench()
    assert b.kernel_family("conv_u8_patch_32x64<3x3>+relu") == "conv_u8_patch"
    assert b.kernel_family("conv_u8_patch_128x64<1x1>") == "conv_u8_patch"
    assert b.kernel_family("conv_u8_mfma_64x64k64+relu+maxpool") == "conv_u8_mfma"
    assert b.kernel_family("conv_u8_rgb3x3+relu+maxpool") == "conv_u8_rgb3x3"
    assert b.kernel_family("pwdw_i8<s1,7x14,512>") == "pwdw_i8"
    assert b.kernel_family("conv_pgemm_i8<128x64,3x3,w4b3>") == "conv_pgemm_i8"
    assert b.kernel_family("conv_igemm_i8<64x64x64>") == "conv_igemm_i8"
    assert b.kernel_family("permute_concat_u8<x6>") == "permute_concat_u8"


def test_pmc_traffic_reads_the_newest_committed_summary():
    b = _bench()
    # the headline workload has a committed PMC pass; bytes per launch of the dominant family are a sane number
    t = b.pmc_traffic("mobilenet_v1", "int8", 1, "pwdw_i8")
    assert t is not None and 2e5 < t < 2e7
    # the uint8 configs of round 3 as well (step name conv_u8_patch -> kernel symbol conv_u8_patch_k)
    t = b.pmc_traffic("yolov3_tiny", "uint8", 8, "conv_u8_patch")
    assert t is not None and t > 1e5
    assert b.pmc_traffic("no_such_model", "int8", 1, "pwdw_i8") is None


def test_evidence_table_is_rederivable_from_profiles():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "evidence_table.py"), "r03"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-500:]
    rows = [ln for ln in r.stdout.splitlines() if ln.startswith("| ") and "ms / step" not in ln]
    assert len(rows) == 5, r.stdout
    readme = open(os.path.join(ROOT, "profiles", "README.md")).read()
    for ln in rows:
        assert ln in readme, "profiles/README.md is out of date with the committed evidence: " + ln


def test_committed_bench_lines_carry_the_contract_fields():
    for f in ("r03_bench_b1.json", "r03_bench_b1_driver_invocation.json"):
        j = json.loads(open(os.path.join(ROOT, "profiles", f)).read().strip().splitlines()[-1])
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                  "config", "roofline", "cpu_baseline", "host_to_host_images_per_s", "prerun_ms", "value_definition"):
            assert k in j, (f, k)
        assert j["vs_baseline"] is None and j["n_gpus"] == 1 and j["dtype"] == "int8" and "workload" in j["config"]
        r = j["roofline"]
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert k in r
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
        c = j["cpu_baseline"]
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in c


def test_reference_thread_cap_defaults_to_the_smallest_allowance(monkeypatch):
    """the reference CPU baseline must be its real multi-thread number.  The reference breaks on >= 64 logical
    CPUs (source/system/cpu.c:110-121,269: zero core mask); bench.py limits OpenMP BEFORE the reference's first graph to
    min(cgroup allowance, physical cores, affinity, 63); TAMD_BENCH_REF_THREADS pins or disables the cap."""
    import bench
    monkeypatch.setattr(bench, "cgroup_cpu_max", lambda: "1600000 100000")
    monkeypatch.setattr(bench.os, "sched_getaffinity", lambda pid: set(range(256)), raising=False)
    assert bench.cgroup_cpus() == 16
    assert bench.ref_thread_cap(256, 128, env="") == (16, "cgroup cpu.max allowance")
    monkeypatch.setattr(bench, "cgroup_cpu_max", lambda: "max 100000")
    assert bench.cgroup_cpus() is None
    assert bench.ref_thread_cap(256, 128, env="") == (63, "the reference's 63-core limit")
    assert bench.ref_thread_cap(256, 32, env="") == (32, "physical cores")
    assert bench.ref_thread_cap(256, 128, env="0")[0] == 256
    assert bench.ref_thread_cap(256, 128, env="40") == (40, "TAMD_BENCH_REF_THREADS")
    assert bench.ref_thread_cap(256, 128, env="200")[0] == 63
    assert bench.ref_thread_cap(8, 4, env="") == (4, "physical cores")


def test_shipped_plan_seeds_a_copy_and_never_overwrites(tmp_path, monkeypatch):
    """tengine_amd/plans.py: the job's plan file starts as a COPY of the shipped one; an existing file is left alone; no shipped plan -> None"""
    from tengine_amd import plans
    monkeypatch.setattr(plans, "PLAN_DIR", str(tmp_path / "plans"))
    os.makedirs(plans.PLAN_DIR)
    src = plans.path_for("mobilenet_v1", "int8", 1)
    open(src, "w").write("#tamd-plan v2 test\nkey\tvalue\n")
    dst = str(tmp_path / "job_plan.txt")
    assert plans.seed(dst, "mobilenet_v1", "int8", 1) is not None
    assert open(dst).read() == open(src).read()
    open(dst, "a").write("mine\t1\n")
    assert plans.seed(dst, "mobilenet_v1", "int8", 1) is None              # the job's own file wins
    assert "mine" in open(dst).read() and "mine" not in open(src).read()
    assert plans.seed(str(tmp_path / "other.txt"), "resnet50", "int8", 32) is None
    assert not os.path.exists(str(tmp_path / "other.txt"))


def test_shipped_plan_of_the_half_batch_is_merged_in(tmp_path, monkeypatch):
    """the library may compile a batch as two graphs of half the batch (tamd_options.split_batch): the job's plan file then also holds the
    shipped plan of batch / 2 -- entries only (every key carries its batch), and only from a file of the same header"""
    from tengine_amd import plans
    monkeypatch.setattr(plans, "PLAN_DIR", str(tmp_path / "plans"))
    os.makedirs(plans.PLAN_DIR)
    open(plans.path_for("resnet50", "int8", 32), "w").write("#tamd-plan v2 test\ngemm|a|32x64\tA\n# make_plans: note\n")
    open(plans.path_for("resnet50", "int8", 16), "w").write("#tamd-plan v2 test\ngemm|a|16x64\tB\n# make_plans: other note\n")
    dst = str(tmp_path / "job.txt")
    assert plans.seed(dst, "resnet50", "int8", 32) is not None
    lines = open(dst).read().splitlines()
    assert lines[0] == "#tamd-plan v2 test" and "gemm|a|32x64\tA" in lines and "gemm|a|16x64\tB" in lines
    assert sum(1 for ln in lines if ln.startswith("#tamd-plan")) == 1 and not any("other note" in ln for ln in lines)
    open(plans.path_for("resnet50", "int8", 16), "w").write("#tamd-plan v1 older build\ngemm|a|16x64\tB\n")
    dst2 = str(tmp_path / "job2.txt")
    assert plans.seed(dst2, "resnet50", "int8", 32) is not None
    assert "16x64" not in open(dst2).read()                        # a void half-batch file is not merged
    # the shipped files of this repository: every batched configuration has its half
    monkeypatch.undo()
    for model, dtype, batch in (("mobilenet_v1", "int8", 64), ("resnet50", "int8", 32)):
        assert os.path.isfile(plans.path_for(model, dtype, batch)) and os.path.isfile(plans.path_for(model, dtype, batch // 2))
        d = str(tmp_path / ("%s.txt" % model))
        plans.seed(d, model, dtype, batch)
        body = open(d).read()
        assert ("n%d " % batch in body or "|%dx" % batch in body) and ("n%d " % (batch // 2) in body or "|%dx" % (batch // 2) in body)


def test_timed_region_repeats():
    """round 6: the region of exactly K steps is repeated until the regions add up to --min-seconds (the driver's K = 20 is a 1 ms region)"""
    b = _bench()
    assert b.repeats_for(0.00106, 0.05, 400) == 48          # the driver's invocation at batch 1
    assert b.repeats_for(0.2, 0.05, 400) == 1               # a region that is long enough is timed once
    assert b.repeats_for(1e-7, 0.05, 400) == 400            # capped
    assert b.repeats_for(0.0, 0.05, 400) == 400
    assert b.repeats_for(0.025, 0.05, 400) == 2


def test_side_configs_have_a_reference_golden():
    """every configuration bench.py times (headline + `configs`) has the real reference's output hashes committed, one per graph output"""
    b = _bench()
    gold = json.load(open(b.GOLDEN_SHA))
    want_outputs = {"mobilenet_v1_int8_b1": 1, "mobilenet_v1_int8_b64": 1, "resnet50_int8_b32": 1, "yolov3_tiny_uint8_b8": 2, "mssd_uint8_b16": 2}
    for name, dtype, batch, _ in b.SIDE_CONFIGS:
        key = "%s_%s_b%d" % (name, dtype, batch)
        sha = b.golden_sha(name, dtype, batch)
        assert sha is not None and len(sha) == want_outputs[key] and all(len(h) == 64 for h in sha), key
        assert gold[key]["seed"] == 1000 and gold[key]["outputs"][0]["shape"][0] == batch
    assert b.golden_sha("no_such_model", "int8", 1) is None
