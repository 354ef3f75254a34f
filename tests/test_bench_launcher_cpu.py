"""`python bench.py --gpus N` starts its own ranks.  Here without a GPU: the
parent's launcher is the product's; the ranks are tests/bench_child_standin.py — bench.py's rank code over
the C++ slab driver with the oracle stand-in as compute and gloo as transport."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
CHILD = os.path.join(ROOT, "tests", "bench_child_standin.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(kw)
    return env


def test_bare_gpus_2_spawns_two_ranks_and_relays_rank_0s_line():
    res = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--grid", "16", "12", "20", "--iters", "10",
                          "--steps", "2", "--warmup", "1", "--no-full-step"],
                         env=_env(FLUID_BENCH_CHILD=CHILD), capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout      # ONE JSON line on stdout, everything else on stderr
    out = json.loads(lines[0])
    assert out["metric"] == "pressure_jacobi_iterations_per_sec" and out["n_gpus"] == 2
    assert out["value"] > 0 and out["scaling"] == "strong" and out["steps"] == 2 and out["warmup"] == 1
    assert out["config"]["parallelism"] == "zslab2" and "workload" in out["config"]
    assert out["n_ranks_rccl"] == 0          # gloo here; on GPUs this is ncclCommCount
    assert out["exchange_ms_per_sweep"] is not None and out["halo_depth"] >= 1
    assert out["overlap_mode"] in ("inline", "split_pass_before_exchange", "split_passes_before_and_after")
    assert out["exchanges_per_step"] >= 1
    c = out["checksum"]
    assert c["words"] == 16 * 12 * 20 and c["matches_one_rank_run"] is True, c


def test_a_failing_rank_fails_the_launcher_with_one_line():
    res = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--grid", "16", "--iters", "4", "--steps", "1",
                          "--warmup", "0", "--no-full-step", "--launch-timeout", "120"],
                         env=_env(FLUID_BENCH_CHILD=CHILD, FLUID_BENCH_CHILD_FAIL_RANK="1"),
                         capture_output=True, text=True, timeout=600)
    assert res.returncode != 0
    assert not [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert "bench.py --gpus 2: rank(s) failed: rank 1 exit code 7" in res.stderr


def test_without_enough_gpus_the_launcher_says_so_in_one_line():
    if os.path.exists("/dev/kfd"):
        import pytest
        pytest.skip("this box has a GPU: covered by the gpu-marked twin")
    res = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1"], env=_env(),
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 2 and res.stdout == ""
    lines = [ln for ln in res.stderr.splitlines() if ln.strip()]
    assert len(lines) == 1 and "0 GPU(s) visible" in lines[0] and "needs 2" in lines[0]


def test_gpus_and_world_size_must_agree():
    res = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1"],
                         env=_env(WORLD_SIZE="1", RANK="0"), capture_output=True, text=True, timeout=300)
    assert res.returncode != 0 and "--gpus 2 but WORLD_SIZE=1" in res.stderr
