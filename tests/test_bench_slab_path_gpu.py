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
