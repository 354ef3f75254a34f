"""bench.py pieces that need no GPU: defaults of the driver contract, the recorded PMC traffic lookup."""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_defaults_follow_the_driver_contract(monkeypatch):
    b = _bench()
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = b.parse_args()
    assert a.gpus == 1 and a.steps > 0 and a.warmup >= 0
    assert b.grid_dims(a.grid) == (512, 512, 512) and a.iters == 200   # BASELINE.json: 512^3 / 200
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7", "--warmup", "3"])
    a = b.parse_args()
    assert (a.gpus, a.steps, a.warmup) == (4, 7, 3)
    assert b.grid_dims([1024, 1024, 512]) == (1024, 1024, 512)


def test_recorded_traffic_matches_the_committed_profile():
    b = _bench()
    traffic, src = b.recorded_traffic("k12_canon2", (512, 512, 512))
    rec = json.load(open(os.path.join(ROOT, src)))
    assert traffic == rec["traffic_bytes_per_launch"] == (2 * rec["FETCH_SIZE_KiB"] + rec["WRITE_SIZE_KiB"]) * 1024
    assert 1.7e9 < traffic < 3.49e9        # between one streamed grid and the two-sweep algorithmic bytes
    assert b.recorded_traffic("k12_canon2", (256, 256, 256)) == (None, None)
    assert b.HBM_PEAK_GBS == 8000.0 and b.JACOBI_BYTES_PER_CELL == 13.0


def test_committed_bench_line_has_the_contract_keys():
    line = json.load(open(os.path.join(ROOT, "profiles", "round01", "bench_512_default.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["metric"] == "pressure_jacobi_iterations_per_sec" and line["n_gpus"] == 1
    assert line["dtype"] == "f32" and line["data"] == "synthetic" and line["vs_baseline"] is None
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(line["roofline"])
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(line["cpu_baseline"])
    assert "workload" in line["config"] and "model" not in line["config"]
