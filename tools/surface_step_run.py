"""Dev tool: whole steps of the dam-break scene with the surface-prep passes (detailed grid = 5x per axis),
per-section times of the tail.  Usage: surface_step_run.py [grid=128] [iters=80]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fluid_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 80
p, cap = fluid_amd.dam_break_params(n, n, n)
with fluid_amd.FluidEngine(p, particle_capacity=cap, pressure_iterations=iters, surface_prep=True) as eng:
    eng.run_init()
    for _ in range(4):
        eng.run_step()
    eng.sync()
    eng.enable_timing(True)
    eng.reset_timing()
    steps = 5
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.run_step()
    eng.sync()
    dt = time.perf_counter() - t0
    d = eng.detailed_shape
    cells = d[0] * d[1] * d[2]
    print(f"{n}^3 + detailed {d[2]}x{d[1]}x{d[0]} ({cells / 1e6:.0f} M cells): {1e3 * dt / steps:.3f} ms/step")
    tail = 0.0
    for k, v in eng.section_times().items():
        if v[1] and k[:2] in ("14", "15", "16", "17", "18"):
            print(f"  {k:42s} {v[0] / steps:8.3f} ms")
            if k != "14_particles":
                tail += v[0] / steps
    print(f"  surface-prep tail {tail:.3f} ms = {56.0 * cells / (tail * 1e-3) / 1e9:.0f} GB/s of the 56 B/cell it moves")
