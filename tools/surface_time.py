"""Dev tool: the surface-prep tail (14a, 15, 16+17, 18 x4) on the dam break with a detailed grid of 5^3 cells
per simulation cell.  Usage: surface_time.py [grid=128] [kernel option=0]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fluid_amd
from fluid_amd import engine as E

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
opt = int(sys.argv[2]) if len(sys.argv) > 2 else 0
p, cap = fluid_amd.dam_break_params(n, n, n)
with fluid_amd.FluidEngine(p, particle_capacity=cap, pressure_iterations=40, surface_prep=True) as eng:
    eng.set_option(E.OPT_SURFACE_KERNEL, opt)
    eng.run_init()
    for _ in range(4):
        eng.run_step()
    eng.enable_timing(True)
    eng.reset_timing()
    steps = 10
    for _ in range(steps):
        eng.run_step()
    t = eng.section_times()
    for k in ("14a_clear_detailed_densities", "15_update_detailed_densities",
              "16_compute_detailed_densities_inertia", "18_diffuse_float_densities"):
        print(f"{k:44s} {t[k][0] / steps:7.3f} ms per step ({t[k][1] // steps} dispatches)")
