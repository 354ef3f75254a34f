"""Dev tool: time one red-black SOR iteration (two launches) against two Jacobi sweeps on the full-fluid grid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fluid_amd
from fluid_amd import engine as E, scenes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
p = fluid_amd.default_params(n, n, n, 0)
with fluid_amd.FluidEngine(p, particle_capacity=0) as eng:
    eng.upload_image(E.CELL_TYPES, scenes.full_fluid_types((n, n, n)))
    div = scenes.full_fluid_divergence((min(n, 64), n, n))
    eng.upload_image(E.DIVERGENCES, np.tile(div, (n // div.shape[0], 1, 1)))
    for name, solver, its in (("jacobi", 0, 100), ("sor", 1, 50)):
        eng.set_pressure_solver(solver, 1.8)
        eng.solve_pressure(4)
        eng.enable_timing(True)
        eng.reset_timing()
        eng.solve_pressure(its)
        ms, calls = eng.section_time_ms("12_solve_pressure")
        eng.enable_timing(False)
        print(f"{name}: {ms / its:.4f} ms per {'iteration (both colours)' if solver else 'sweep'} at {n}^3")
