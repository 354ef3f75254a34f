"""bench.py's multi-GPU branch (slab Jacobi benchmark + slab full step) run with an RCCL group of ONE
rank on the test box's single GPU: everything except the wire is the code the driver launches with
torch.distributed.run at N = 2, 4, 8."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_multi_gpu_branch_with_one_rank():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0",
               WORLD_SIZE="1", LOCAL_RANK="0", FLUID_BENCH_FORCE_SLAB="1")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--grid",
                          "128", "--iters", "20", "--steps", "2", "--warmup", "1",
                          "--full-step-steps", "2"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout + res.stderr
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["metric"] == "pressure_jacobi_iterations_per_sec" and out["value"] > 0
    assert out["roofline"]["frac"] > 0
    assert "error" not in out["full_step"], out["full_step"]
    assert out["full_step"]["steps_per_sec"] > 0


def _bare_env(**kw):
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(kw)
    return env


def test_self_launched_rank_on_the_gpu():
    """`bench.py --gpus 1 --spawn`: the launcher the driver's `--gpus N` call goes through (parent without
    HIP, child processes with RANK / WORLD_SIZE set, rank 0's line relayed), with the one rank this box
    can hold, through the slab branch."""
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--spawn", "--grid",
                          "128", "--iters", "20", "--steps", "2", "--warmup", "1", "--no-full-step"],
                         env=_bare_env(FLUID_BENCH_FORCE_SLAB="1"), capture_output=True, text=True,
                         timeout=900)
    assert res.returncode == 0, res.stdout + res.stderr
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["roofline"]["frac"] > 0
    assert out["checksum"]["matches_one_rank_run"] is True, out["checksum"]
    assert out["checksum"]["words"] == 128 ** 3
    assert out["overlap_mode"] == "inline" and out["exchange_ms_per_sweep"] is not None


def test_bare_gpus_2_on_a_one_gpu_box_fails_with_one_line():
    import ctypes
    n = ctypes.c_int(0)
    # (counting devices in a child would be cleaner still; this process has the GPU anyway: pytest -m gpu)
    if ctypes.CDLL("libamdhip64.so").hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value >= 2:
        pytest.skip("this box has two GPUs")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"],
                         env=_bare_env(), capture_output=True, text=True, timeout=300)
    assert res.returncode == 2 and res.stdout == ""
    lines = [ln for ln in res.stderr.splitlines() if ln.strip()]
    assert len(lines) == 1 and "1 GPU(s) visible" in lines[0] and "needs 2" in lines[0]
