"""dev: ONE three-sweep launch (explicit loop API), without / with the kept iterate, against the plain kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fluid_amd
from fluid_amd import engine as E, scenes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
sweeps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
p = fluid_amd.default_params(n, n, n, 0)
t = scenes.full_fluid_types((n, n, n)); div = scenes.full_fluid_divergence((n, n, n))
with fluid_amd.FluidEngine(p, particle_capacity=0) as eng:
    eng.upload_image(E.CELL_TYPES, t); eng.upload_image(E.DIVERGENCES, div)
    eng.set_option(E.OPT_PRESSURE_KERNEL, 1)
    eng.run_section("12a_clear_pressures_1"); eng.run_section("12b_clear_pressures_2"); eng.solve_pressure(sweeps)
    ref = eng.download_image(E.PRESSURES_2 if sweeps % 2 else E.PRESSURES_1)
    eng.set_option(E.OPT_PRESSURE_KERNEL, 0)
    for keep in (False, True):
        bad = 0
        for rep in range(reps):
            eng.run_section("12a_clear_pressures_1"); eng.run_section("12b_clear_pressures_2")
            eng.pressure_loop_begin(); eng.pressure_loop_advance(sweeps, keep); eng.pressure_loop_end()
            got = eng.download_image(E.PRESSURES_2 if sweeps % 2 else E.PRESSURES_1)
            d = got.view(np.uint32) != ref.view(np.uint32)
            if d.any():
                bad += 1
                idx = np.argwhere(d)
                if bad <= 4:
                    print(f"keep {keep} rep {rep}: {int(d.sum())} differ planes", np.unique(idx[:, 0])[:8], "rows", np.unique(idx[:, 1])[:12],
                          "x", idx[:, 2].tolist()[:40], flush=True)
                    z, y, x = idx[0]
                    print("   got", got[z, y, x:x + 4], "ref", ref[z, y, x:x + 4], "types", t[z, y, x - 1:x + 5], flush=True)
        print(f"grid {n}^3, one launch of {sweeps}, keep {keep}: {bad} of {reps} runs differ", flush=True)
