"""Dev tool: fused_trace.py for a small box of water — the dam break on a 256 x 512 x 512 grid (one x tile, no x
window: the traced translation unit), whose Jacobi launches are one round of workgroups marching a few planes.
Needs `make -C vulkan-3d-fluid-simulation_amd/csrc trace`.   python tools/fused_trace_sparse.py [iters=40]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import fluid_amd
from fluid_amd import engine as E

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
lib_path = os.path.join(ROOT, "vulkan-3d-fluid-simulation_amd", "libfluid_engine_trace.so")
p, cap = fluid_amd.dam_break_params(256, 512, 512)
eng = fluid_amd.FluidEngine(p, particle_capacity=cap, pressure_iterations=iters, lib_path=lib_path)
eng.run_init()
for _ in range(5):
    eng.run_step()
eng.sync()
eng.enable_timing(True)
eng.reset_timing()
eng.run_step()
ms, n = eng.section_time_ms("12_solve_pressure")
print(f"12_solve_pressure: {ms:.3f} ms for {iters} sweeps = {1e3 * ms / (iters / 2):.1f} us per two-sweep launch (passes included)")
PH = 6
buf = (C.c_ulonglong * (64 * 16 * (PH + 1)))()
lib = eng._lib
lib.fluid_dev_fused_trace.argtypes = [C.c_void_p, C.c_int]
rc = lib.fluid_dev_fused_trace(buf, len(buf))
assert rc == 0, rc
t = np.frombuffer(buf, dtype=np.uint64).reshape(64, 16, PH + 1).astype(np.float64)
steps = t[..., PH]
ok = steps > 0
names = ["issue loads + wait for the previous step's", "table look-ups + publish", "barrier",
         "LDS round trip + stage 1", "stage 2 + stores", "whole march"]
per = t[..., :PH] / np.maximum(steps[..., None], 1)
print(f"{int(ok.sum())} wavefronts traced, {steps[ok].mean():.1f} steps each (min {steps[ok].min():.0f}, max {steps[ok].max():.0f}); s_memtime ticks (100 MHz) per step")
for i, nme in enumerate(names):
    v = per[..., i][ok]
    print(f"  {nme:45s} mean {v.mean():9.1f}   min {v.min():9.1f}   max {v.max():9.1f}")
tot = t[..., PH - 1][ok]
print(f"  whole march per wavefront: mean {tot.mean():.0f} ticks, max {tot.max():.0f}")
eng.close()
