#!/usr/bin/env python3
"""Run the whole step (fluid_run_step) of the dam-break scene a few times; meant to sit under
rocprofv3 --kernel-trace --stats.  Usage: full_step_run.py [grid=512] [steps=10] [iters=200] [warmup=3]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fluid_amd  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    warm = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    p, cap = fluid_amd.dam_break_params(n, n, n)
    with fluid_amd.FluidEngine(p, particle_capacity=cap, pressure_iterations=iters) as eng:
        eng.run_init()
        for _ in range(warm):
            eng.run_step()
        eng.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.run_step()
        eng.sync()
        dt = time.perf_counter() - t0
        print(f"grid {n}^3 iters {iters}: {steps / dt:.2f} steps/s ({1e3 * dt / steps:.3f} ms/step)")
        if os.environ.get("FLUID_SECTIONS"):
            eng.enable_timing(True)
            eng.reset_timing()
            for _ in range(3):
                eng.run_step()
            eng.sync()
            print("  ", {k[:20]: round(v[0] / 3, 3) for k, v in eng.section_times().items() if v[1] and v[0] / 3 > 0.02})


if __name__ == "__main__":
    main()
