"""Dev tool: A/B the Jacobi loop of several builds of libfluid_engine.so in ONE process on ONE
device, interleaved rounds (cdna_hip_programming.md §5.4 rule 24).

    python tools/ab_libs.py [--grid 512] [--iters 100] [--rounds 7] name=path[:opt=val,...] ...
e.g. python tools/ab_libs.py base=gpurun_out/lib_base.so new=vulkan-3d-fluid-simulation_amd/libfluid_engine.so:fuse=1
"""
import argparse
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import fluid_amd  # noqa: E402
from fluid_amd import engine as E  # noqa: E402
from fluid_amd import scenes  # noqa: E402

OPTS = {"kernel": E.OPT_PRESSURE_KERNEL, "fuse": E.OPT_JACOBI_FUSE}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=512)
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("specs", nargs="+")
    a = ap.parse_args()
    n = a.grid
    p = fluid_amd.default_params(n, n, n, 0)
    t = scenes.full_fluid_types((n, n, n))
    div = scenes.full_fluid_divergence((min(n, 64), n, n))
    div = np.tile(div, (n // div.shape[0], 1, 1))
    engines = []
    for spec in a.specs:
        name, rest = spec.split("=", 1)
        path, _, optstr = rest.partition(":")
        eng = fluid_amd.FluidEngine(p, particle_capacity=0, lib_path=os.path.abspath(path))
        for kv in filter(None, optstr.split(",")):
            k, v = kv.split("=")
            eng.set_option(OPTS[k], int(v))
        eng.upload_image(E.CELL_TYPES, t)
        eng.upload_image(E.DIVERGENCES, div)
        eng.run_section("12a_clear_pressures_1")
        eng.run_section("12b_clear_pressures_2")
        eng.solve_pressure(a.iters)  # warm-up, builds mask etc.
        eng.sync()
        eng.enable_timing(True)
        engines.append((name, eng, []))
    for _ in range(a.rounds):
        for name, eng, samples in engines:
            eng.reset_timing()
            eng.run_section("12a_clear_pressures_1")
            eng.run_section("12b_clear_pressures_2")
            eng.solve_pressure(a.iters)
            ms, calls = eng.section_time_ms("12_solve_pressure")
            samples.append(ms / a.iters)
    ref = None
    for name, eng, samples in engines:
        med, lo = statistics.median(samples), min(samples)
        ref = ref or med
        gbs = 13.0 * n ** 3 / (med * 1e-3) / 1e9
        print(f"{name:16s} median {med:.4f} ms/sweep  min {lo:.4f}  {gbs:7.1f} GB/s alg  "
              f"x{ref / med:.3f} vs first")
        eng.close()


if __name__ == "__main__":
    main()
