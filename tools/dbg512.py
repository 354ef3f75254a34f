"""dev: the default Jacobi loop against the one-thread-per-cell kernel, many times over (a race would show as a
run that differs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fluid_amd
from fluid_amd import engine as E, scenes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 8
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
p = fluid_amd.default_params(n, n, n, 0)
t = scenes.full_fluid_types((n, n, n)); div = scenes.full_fluid_divergence((n, n, n))
bad = 0
with fluid_amd.FluidEngine(p, particle_capacity=0) as eng:
    eng.upload_image(E.CELL_TYPES, t); eng.upload_image(E.DIVERGENCES, div)
    eng.set_option(E.OPT_PRESSURE_KERNEL, 1)
    eng.run_section("12a_clear_pressures_1"); eng.run_section("12b_clear_pressures_2"); eng.solve_pressure(iters)
    ref = eng.download_image(E.PRESSURES_1)
    eng.set_option(E.OPT_PRESSURE_KERNEL, 0)
    for rep in range(reps):
        fuse = 2 if rep % 4 == 3 else 0
        eng.set_option(E.OPT_JACOBI_FUSE, fuse)
        eng.run_section("12a_clear_pressures_1"); eng.run_section("12b_clear_pressures_2"); eng.solve_pressure(iters)
        p1 = eng.download_image(E.PRESSURES_1)
        d = p1.view(np.uint32) != ref.view(np.uint32)
        if d.any():
            bad += 1
            idx = np.argwhere(d)
            print(f"rep {rep} fuse {fuse}: {int(d.sum())} differ z", idx[:, 0].min(), idx[:, 0].max(), "y", idx[:, 1].min(), idx[:, 1].max(),
                  "x", idx[:, 2].min(), idx[:, 2].max(), "planes", np.unique(idx[:, 0])[:8], "rows", np.unique(idx[:, 1])[:16], flush=True)
print(f"grid {n}^3, {iters} sweeps, {reps} runs: {bad} differ")
