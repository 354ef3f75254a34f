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


def test_recorded_traffic_is_taken_only_from_a_record_of_the_current_kernel_sources(tmp_path, monkeypatch):
    b = _bench()
    path = os.path.join(ROOT, "profiles", "round02", "pmc_traffic_k12_canon2_512x512x512.json")
    rec = json.load(open(path))
    # calibrated rule (profiles/round02/calibration_read_request_size.txt): a read request is 128 B,
    # FETCH_SIZE counts 64 B per request, WRITE_SIZE is exact
    assert abs(rec["read_bytes_per_launch"] - rec["TCC_EA0_RDREQ_sum"] * 128) < 1.0
    assert abs(rec["read_bytes_per_launch"] - 2 * rec["FETCH_SIZE_KiB"] * 1024) < 1e-3 * rec["read_bytes_per_launch"]
    assert rec["write_bytes_per_launch"] == rec["WRITE_SIZE_KiB"] * 1024
    assert rec["traffic_bytes_per_launch"] == rec["read_bytes_per_launch"] + rec["write_bytes_per_launch"]
    assert rec["single_pass_min_bytes"] == 13.0 * 512 ** 3
    assert 1.0 <= rec["traffic_bytes_per_launch"] / rec["single_pass_min_bytes"] < 2.0   # two sweeps, < 2 passes
    traffic, src = b.recorded_traffic("k12_canon2", (512, 512, 512))
    if rec["kernel_sources_sha16"] == b.kernel_sources_sha16():
        assert traffic == rec["traffic_bytes_per_launch"] and src.endswith(os.path.basename(path))
    else:   # kernel edited after the counters were collected: no traffic claim, and the reason is given
        assert traffic is None and "recorded for kernel sources" in src
    assert b.recorded_traffic("k12_canon2", (256, 256, 256)) == (None, None)
    assert b.HBM_PEAK_GBS == 8000.0 and b.JACOBI_BYTES_PER_CELL == 13.0
    # a record of other sources is refused even if it is the only one
    fake = dict(rec, kernel_sources_sha16="0" * 16)
    os.makedirs(tmp_path / "profiles" / "roundXX")
    json.dump(fake, open(tmp_path / "profiles" / "roundXX" / "pmc_traffic_x.json", "w"))
    monkeypatch.setattr(b, "ROOT", str(tmp_path))
    monkeypatch.setattr(b, "kernel_sources_sha16", lambda: rec["kernel_sources_sha16"])
    t2, why = b.recorded_traffic("k12_canon2", (512, 512, 512))
    assert t2 is None and "0000000000000000" in why


def _bench_line_ok(line):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["metric"] == "pressure_jacobi_iterations_per_sec" and line["n_gpus"] == 1
    assert line["dtype"] == "f32" and line["data"] == "synthetic" and line["vs_baseline"] is None
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(line["roofline"])
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(line["cpu_baseline"])
    assert "workload" in line["config"] and "model" not in line["config"]


def test_committed_bench_lines_have_the_contract_keys():
    import glob
    lines = sorted(glob.glob(os.path.join(ROOT, "profiles", "round*", "bench_512_default.json")))
    assert lines
    for path in lines:
        _bench_line_ok(json.load(open(path)))
    newest = json.load(open(lines[-1]))
    if "round01" not in lines[-1]:
        r = newest["roofline"]
        assert "kernel_sources_sha16" in r
        if r["traffic"] is not None:   # the physical fraction never exceeds the peak, whatever the algorithmic one says
            assert 0.0 < r["frac_traffic"] <= 1.0 and r["traffic_over_single_pass_min"] >= 1.0
