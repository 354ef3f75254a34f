"""Dev tool: time the Jacobi loop of the current build for several values of an environment variable
(read by the launcher at every launch), one process, interleaved rounds."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import fluid_amd
from fluid_amd import engine as E, scenes

var, values = sys.argv[1], sys.argv[2].split(",")
n = int(sys.argv[3]) if len(sys.argv) > 3 else 512
iters = 100
p = fluid_amd.default_params(n, n, n, 0)
eng = fluid_amd.FluidEngine(p, particle_capacity=0)
eng.upload_image(E.CELL_TYPES, scenes.full_fluid_types((n, n, n)))
div = scenes.full_fluid_divergence((min(n, 64), n, n))
eng.upload_image(E.DIVERGENCES, np.tile(div, (n // div.shape[0], 1, 1)))
eng.enable_timing(True)
res = {v: [] for v in values}
for r in range(6):
    for v in values:
        os.environ[var] = v
        eng.reset_timing()
        eng.run_section("12a_clear_pressures_1"); eng.run_section("12b_clear_pressures_2")
        eng.solve_pressure(iters)
        ms, _ = eng.section_time_ms("12_solve_pressure")
        if r: res[v].append(ms / iters)
for v in values:
    print(f"{var}={v:6s} median {statistics.median(res[v]):.4f} ms/sweep  min {min(res[v]):.4f}")
