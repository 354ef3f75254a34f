"""The division-free quotient of the fused Jacobi kernel (csrc/kernels_pressure_fused.h:
div_small_int) against IEEE division on the CPU — sampled here, exhaustive with `full`
(tests/divide_small_int_check.c; the exhaustive run is recorded in DESIGN.md)."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _has_fma():
    try:
        return " fma " in open("/proc/cpuinfo").read().replace("\n", " ")
    except OSError:
        return False


@pytest.mark.skipif(shutil.which("gcc") is None or not _has_fma(), reason="needs gcc and an FMA CPU")
def test_three_op_quotient_equals_ieee_division(tmp_path):
    exe = str(tmp_path / "divide_small_int_check")
    subprocess.run(["gcc", "-O2", "-mfma", "-fopenmp", "-ffp-contract=off", "-o", exe,
                    os.path.join(HERE, "divide_small_int_check.c"), "-lm"], check=True)
    res = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "0 mismatches above the guard" in res.stdout
