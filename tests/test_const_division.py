"""CPU: the constant-divisor quotient of the stem pre-pass (csrc/stem_bf16.hip::div_const) is the IEEE quotient — exhaustively, for
every float in [2^-40, 512] and the four divisors in use (255 and the three Normalize standard deviations). The reference divides
(x / 255, then (x - mean) / std: /root/reference/r3m/models/models_r3m.py:97-98); the kernel must produce the same bits."""
import os
import shutil
import subprocess

import pytest


def _has_fma():
    try:
        return " fma " in open("/proc/cpuinfo").read()
    except OSError:
        return False


@pytest.mark.skipif(shutil.which("gcc") is None or not _has_fma(), reason="needs gcc and a CPU with hardware fma")
def test_div_const_equals_ieee_division_for_every_float(tmp_path):
    src = os.path.join(os.path.dirname(__file__), "const_division_check.c")
    exe = str(tmp_path / "divcheck")
    subprocess.run(["gcc", "-O2", "-mfma", "-ffp-contract=off", src, "-o", exe, "-lm"], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    lines = out.stdout.strip().splitlines()
    assert out.returncode == 0 and len(lines) == 4, out.stdout + out.stderr
    for ln in lines:
        assert ln.endswith("mismatches=0") and "values=411041793" in ln, ln
