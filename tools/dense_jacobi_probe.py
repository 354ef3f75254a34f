"""Dev tool: why does the fused Jacobi launch take 10 % longer inside the full-tank step than in the
micro-benchmark?  Times the loop on the tank's own state, then with the benchmark's divergence and / or
cell types swapped in."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fluid_amd
from fluid_amd import engine as E, scenes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
size = (n - 4.0,) * 3
res = tuple(int(round(2.0 * s)) for s in size)
vol = res[0] * res[1] * res[2]
p = fluid_amd.default_params(n, n, n, vol)
p.particle_spawn_cube_resolution[:] = res
p.particle_spawn_cube_volume = vol
p.particle_spawn_cube_offset[:] = (2.0, 2.0, 2.0)
p.particle_spawn_cube_size[:] = size


def timed(eng, label, its=400):
    eng.solve_pressure(40)
    vals = []
    for _ in range(3):
        eng.enable_timing(True)
        eng.reset_timing()
        eng.solve_pressure(its)
        ms, _ = eng.section_time_ms("12_solve_pressure")
        eng.enable_timing(False)
        vals.append(ms / its)
    print(f"{label:60s} " + "  ".join(f"{v:.4f}" for v in vals) + " ms per sweep (passes included)", flush=True)


with fluid_amd.FluidEngine(p, particle_capacity=vol, pressure_iterations=20) as eng:
    eng.run_init()
    for _ in range(3):
        eng.run_step()
    types = eng.download_image(E.CELL_TYPES)
    div = eng.download_image(E.DIVERGENCES)
    print("tank: water cells", int((types == 1).sum()), "types", np.unique(types, return_counts=True),
          "div: zeros", int((div == 0).sum()), "denormal", int(((div != 0) & (np.abs(div) < 1.2e-38)).sum()),
          "nonfinite", int((~np.isfinite(div)).sum()), "max|div|", float(np.nanmax(np.abs(div))), flush=True)
    timed(eng, "tank types, tank divergence")
    syn = scenes.full_fluid_divergence((min(n, 64), n, n))
    syn = np.tile(syn, (n // syn.shape[0], 1, 1))
    eng.upload_image(E.DIVERGENCES, syn)
    timed(eng, "tank types, benchmark divergence")
    eng.upload_image(E.CELL_TYPES, scenes.full_fluid_types((n, n, n)))
    timed(eng, "benchmark types, benchmark divergence")
    eng.upload_image(E.DIVERGENCES, div)
    timed(eng, "benchmark types, tank divergence")
    eng.upload_image(E.DIVERGENCES, np.zeros_like(div))
    timed(eng, "benchmark types, zero divergence")
p0 = fluid_amd.default_params(n, n, n, 0)
with fluid_amd.FluidEngine(p0, particle_capacity=0) as eng:
    eng.upload_image(E.CELL_TYPES, scenes.full_fluid_types((n, n, n)))
    eng.upload_image(E.DIVERGENCES, syn)
    timed(eng, "no particles: benchmark types, benchmark divergence")
    eng.upload_image(E.CELL_TYPES, types)
    eng.upload_image(E.DIVERGENCES, div)
    timed(eng, "no particles: tank types, tank divergence")
