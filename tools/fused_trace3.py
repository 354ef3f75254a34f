"""Dev tool: where do the cycles of a plane step of the three-sweeps-per-pass Jacobi kernel go?  Needs
`make -C vulkan-3d-fluid-simulation_amd/csrc trace` (libfluid_engine_trace.so: s_memtime stamps).  Runs the
Jacobi loop and prints, per phase, the mean cycles per step over the wavefronts of the first 64 workgroups of
the LAST three-sweep launch.      python tools/fused_trace3.py [--grid 512 512 512] [--iters 9]"""
import argparse, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import fluid_amd
from fluid_amd import engine as E, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--grid", type=int, nargs=3, default=[512, 512, 512])
ap.add_argument("--iters", type=int, default=9)
a = ap.parse_args()
lib_path = os.path.join(ROOT, "vulkan-3d-fluid-simulation_amd", "libfluid_engine_trace.so")
w, h, d = a.grid
p = fluid_amd.default_params(w, h, d, 0)
eng = fluid_amd.FluidEngine(p, particle_capacity=0, lib_path=lib_path)
eng.upload_image(E.CELL_TYPES, scenes.full_fluid_types((d, h, w)))
eng.upload_image(E.DIVERGENCES, scenes.full_fluid_divergence((min(d, 64), h, w)).repeat(max(d // 64, 1), axis=0)[:d])
for _ in range(2):
    eng.run_section("12a_clear_pressures_1")
    eng.run_section("12b_clear_pressures_2")
    eng.solve_pressure(a.iters)
eng.sync()
PH = 7
buf = (C.c_ulonglong * (64 * 16 * (PH + 1)))()
lib = eng._lib
lib.fluid_dev_fused_trace3.argtypes = [C.c_void_p, C.c_int]
rc = lib.fluid_dev_fused_trace3(buf, len(buf))
assert rc == 0, rc
t = np.frombuffer(buf, dtype=np.uint64).reshape(64, 16, PH + 1).astype(np.float64)
steps = t[..., PH]
ok = steps > 0
names = ["LDS reads issued, wait for loads, fix-ups", "stage 1", "stage 2 (+ issue of the global loads)",
         "stage 3 + stores", "publish", "barrier", "whole march"]
per = t[..., :PH] / np.maximum(steps[..., None], 1)
print(f"grid {w}x{h}x{d}: {int(ok.sum())} wavefronts traced, {steps[ok].mean():.1f} steps each; s_memtime ticks per step")
for i, n in enumerate(names):
    v = per[..., i][ok]
    print(f"  {n:45s} mean {v.mean():9.1f}   min {v.min():9.1f}   max {v.max():9.1f}")
print("  by wavefront, mean ticks per step (the phases above, without the whole march):")
for wv in range(16):
    v = per[:, wv, :6][ok[:, wv]]
    if len(v):
        print(f"    wave {wv:2d}: " + "  ".join(f"{x:8.1f}" for x in v.mean(0)))
eng.close()
