"""Dev tool: does storing the particles sorted pay over a long run in which the water moves?  The 512^3 dam
break for N steps with FLUID_OPT_PARTICLE_SORT off and on (default policy), total time and the sort count.
Usage: particle_sort_longrun.py [grid=512] [steps=300] [iters=50]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fluid_amd
from fluid_amd import engine as E

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 50
p, cap = fluid_amd.dam_break_params(n, n, n)
for mode in (1, 0):
    with fluid_amd.FluidEngine(p, particle_capacity=cap, pressure_iterations=iters) as eng:
        eng.set_option(E.OPT_PARTICLE_SORT, mode)
        eng.run_init()
        eng.run_step()
        eng.sync()
        eng.enable_timing(True)
        eng.reset_timing()
        t0 = time.perf_counter()
        marks = []
        for k in range(steps):
            eng.run_step()
            if (k + 1) % 50 == 0:
                marks.append((k + 1, eng.get_stat(E.STAT_PARTICLE_SORTS), eng.get_stat(E.STAT_PARTICLE_STRAYS)))
        eng.sync()
        dt = time.perf_counter() - t0
        t = eng.section_times()
        print(f"sort option {mode}: {1e3 * dt / steps:7.3f} ms/step over {steps} steps   01 {t['01_update_densities'][0] / steps:6.3f} ms"
              f"   14 {t['14_particles'][0] / steps:6.3f} ms   (step, sorts, strays) {marks}", flush=True)
