"""The three-sweeps-per-pass Jacobi kernel takes the reciprocals of its quotient n / aii from v_rcp_f32
(kernels_pressure_fused3.h: div_pairs_rcp).  tests/divide_small_int_check.c proves the quotient chain exact for
RN(1 / aii) and for one ulp less; this is the check that the instruction returns one of the two on the GPU at
hand (tools/micro/rcp_small_int.hip: exit status 0)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_v_rcp_f32_of_the_small_integers_is_a_proven_reciprocal():
    exe = os.path.join(ROOT, "tools", "micro", "rcp_small_int")
    if not os.path.exists(exe):
        subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "--offload-arch=gfx950", "-o", exe, exe + ".hip"], check=True)
    res = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    lines = {int(ln.split()[2]): ln for ln in res.stdout.splitlines() if ln.startswith("a =")}
    for a in (1, 2, 4):
        assert lines[a].rstrip().endswith("same"), lines[a]
    for a in (3, 5, 6):
        assert lines[a].rstrip().endswith(("same", "one ulp less")), lines[a]
