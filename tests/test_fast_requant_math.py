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
