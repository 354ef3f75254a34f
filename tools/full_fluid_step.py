"""Dev tool: whole steps of a tank that is almost completely full of water (every section works on nearly
every cell), per-section times.  Usage: full_fluid_step.py [grid=256] [iters=50]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fluid_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
size = (n - 4.0,) * 3
res = tuple(int(round(2.0 * s)) for s in size)
vol = res[0] * res[1] * res[2]
p = fluid_amd.default_params(n, n, n, vol)
p.particle_spawn_cube_resolution[:] = res
p.particle_spawn_cube_volume = vol
p.particle_spawn_cube_offset[:] = (2.0, 2.0, 2.0)
p.particle_spawn_cube_size[:] = size
with fluid_amd.FluidEngine(p, particle_capacity=vol, pressure_iterations=iters) as eng:
    eng.run_init()
    for _ in range(3):
        eng.run_step()
    eng.sync()
    eng.enable_timing(True)
    eng.reset_timing()
    t0 = time.perf_counter()
    steps = 5
    for _ in range(steps):
        eng.run_step()
    eng.sync()
    dt = time.perf_counter() - t0
    print(f"full tank {n}^3, {vol} particles, {iters} iters: {1e3 * dt / steps:.3f} ms/step")
    for k, v in eng.section_times().items():
        if v[1]:
            print(f"  {k:40s} {v[0] / steps:8.3f} ms")
