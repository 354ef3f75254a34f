"""Dev tool: per-section times of the dam-break step (sparse scene).  sparse_sections.py [grid=512] [iters=200]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fluid_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
p, cap = fluid_amd.dam_break_params(n, n, n)
with fluid_amd.FluidEngine(p, particle_capacity=cap, pressure_iterations=iters) as eng:
    eng.run_init()
    for _ in range(10):
        eng.run_step()
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(20):
        eng.run_step()
    eng.sync()
    print(f"dam break {n}^3: {1e3 * (time.perf_counter() - t0) / 20:.3f} ms/step")
    eng.enable_timing(True); eng.reset_timing()
    for _ in range(5):
        eng.run_step()
    eng.sync()
    for k, v in eng.section_times().items():
        if v[1]:
            print(f"  {k:40s} {v[0] / 5:8.3f} ms")
