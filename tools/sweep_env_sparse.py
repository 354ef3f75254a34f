"""Dev tool: like sweep_env.py, on the dam-break scene (sparse water) after a few whole steps."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fluid_amd
from fluid_amd import engine as E

var, values = sys.argv[1], sys.argv[2].split(",")
n = int(sys.argv[3]) if len(sys.argv) > 3 else 512
iters = 200
p, cap = fluid_amd.dam_break_params(n, n, n)
eng = fluid_amd.FluidEngine(p, particle_capacity=cap, pressure_iterations=iters)
eng.run_init()
for _ in range(4):
    eng.run_step()
eng.enable_timing(True)
res = {v: [] for v in values}
for r in range(5):
    for v in values:
        if v == "default":
            os.environ.pop(var, None)
        else:
            os.environ[var] = v
        eng.reset_timing()
        eng.run_section("12a_clear_pressures_1"); eng.run_section("12b_clear_pressures_2")
        eng.solve_pressure(iters)
        ms, _ = eng.section_time_ms("12_solve_pressure")
        if r: res[v].append(ms / iters)
for v in values:
    print(f"{var}={v:8s} median {1e3 * statistics.median(res[v]):.2f} us/sweep  min {1e3 * min(res[v]):.2f}")
