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
