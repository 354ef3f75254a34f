"""Dev tool: the particle passes with the storage in slot order and sorted by bin (FLUID_OPT_PARTICLE_SORT),
on the full tank and on the dam break.  Usage: particle_sort_ab.py [grid=512] [iters=20]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fluid_amd
from fluid_amd import engine as E

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20


def tank():
    size = (n - 4.0,) * 3
    res = tuple(int(round(2.0 * s)) for s in size)
    vol = res[0] * res[1] * res[2]
    p = fluid_amd.default_params(n, n, n, vol)
    p.particle_spawn_cube_resolution[:] = res
    p.particle_spawn_cube_volume = vol
    p.particle_spawn_cube_offset[:] = (2.0, 2.0, 2.0)
    p.particle_spawn_cube_size[:] = size
    return p, vol


for scene, (p, cap) in (("dam break", fluid_amd.dam_break_params(n, n, n)), ("full tank", tank())):
    for mode in (1, 2):
        with fluid_amd.FluidEngine(p, particle_capacity=cap, pressure_iterations=iters) as eng:
            eng.set_option(E.OPT_PARTICLE_SORT, mode)
            eng.run_init()
            for _ in range(4):
                eng.run_step()
            eng.sync()
            eng.enable_timing(True)
            eng.reset_timing()
            steps = 6
            t0 = time.perf_counter()
            for _ in range(steps):
                eng.run_step()
            eng.sync()
            dt = time.perf_counter() - t0
            t = eng.section_times()
            print(f"{scene} {n}^3, {cap} particles, sort mode {mode}: {1e3 * dt / steps:8.3f} ms/step   "
                  f"01 {t['01_update_densities'][0] / steps:7.3f} ms   14 {t['14_particles'][0] / steps:7.3f} ms   "
                  f"sorts {eng.get_stat(E.STAT_PARTICLE_SORTS)} strays {eng.get_stat(E.STAT_PARTICLE_STRAYS)}",
                  flush=True)
