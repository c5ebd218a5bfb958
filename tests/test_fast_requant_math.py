"""The division-free requantisation fast path of tengine_amd/csrc/epilogue.h (round_div_sat) is exact:
host replica vs the reference expression on random + boundary-hugging inputs (tests/csrc/fast_requant_check.c)."""
import os
import subprocess


def test_round_div_sat_is_exact(tmp_path):
    src = os.path.join(os.path.dirname(__file__), "csrc", "fast_requant_check.c")
    exe = str(tmp_path / "frc")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", src, "-o", exe, "-lm"])
    out = subprocess.run([exe, "4000000"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    assert "mismatches 0" in out.stdout


def test_one_fma_requant_is_exact(tmp_path):
    """requant4 / requant1 of epilogue.h: y = fma(acc, M[c], 128.5 + e), clamp window, truncate -- equal to the reference chain
    (two multiplications, clamp, division, round half away, saturate) on every value the fast path keeps; random layers
    (scales over six decades, every activation window) x random and boundary-hugging accumulators
    (tests/csrc/fold_requant_check.c, same IEEE operations as the device)."""
    src = os.path.join(os.path.dirname(__file__), "csrc", "fold_requant_check.c")
    exe = str(tmp_path / "foldc")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", src, "-o", exe, "-lm"])
    out = subprocess.run([exe, "20000"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    assert "mismatches 0" in out.stdout


def test_uint8_saturating_round_is_exact(tmp_path):
    """quant_round_sat_u8 of u8_kernels.hip (uint8 conv / depthwise / FC / eltwise epilogues): the division-free rounding against
    sat_u8((int)clamp(roundf(s / scale)) + zp) on random, far-out-of-range and boundary-hugging values (tests/csrc/u8_round_check.c)."""
    src = os.path.join(os.path.dirname(__file__), "csrc", "u8_round_check.c")
    exe = str(tmp_path / "u8c")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", src, "-o", exe, "-lm"])
    out = subprocess.run([exe, "6000000"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    assert "mismatches 0" in out.stdout


def test_one_fma_requant_exhaustive_on_model_constants(tmp_path):
    """EVERY accumulator value (not a sample) for the real quantisation constants of MobileNet-v1's first layers and of ResNet-50's FC, in the
    three reference formulas (tools/exhaustive_requant.py + tests/csrc/exhaustive_requant.c; the whole of MobileNet-v1 and
    ResNet-50 -- 1.4e10 accumulators, 0 mismatches -- is in profiles/r02_exhaustive_requant.txt)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import exhaustive_requant as ex
    rec = ex.records("mobilenet_v1")
    sub = rec[:256]                                  # conv1, conv2_1/dw, conv2_1/sep x (A1, A2)
    rn = ex.records("resnet50")
    fc = rn[rn[:, 4] == 1.0][:64]                    # ResNet-50's fc1000 in formula A5 (out_scale folded to 1)
    assert len(fc) == 64
    path = str(tmp_path / "rec.bin")
    import numpy as np
    np.concatenate([sub, fc]).astype(np.float32).tofile(path)
    exe = str(tmp_path / "exh")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", os.path.join(os.path.dirname(__file__), "csrc", "exhaustive_requant.c"), "-o", exe, "-lm"])
    out = subprocess.run([exe, path], capture_output=True, text=True)
    assert out.returncode == 0 and "mismatches 0" in out.stdout and "records 320" in out.stdout, out.stdout
