"""Dev tool: rehearse ONE rank of an N-GPU Z-slab run on a single GPU.  The rank owns an interior slab
(neighbours on both sides); halo exchanges are replaced by device copies of the same size
(fluid_slab_attach_loopback), so kernel time, redundant ghost-region compute and host overhead of the C++
schedule are real and only the wire / RCCL latency is missing.  Prints ms per sweep and the implied
aggregate iterations/s.

    python tools/slab_rank_sim.py [--grid 512] [--ranks 8] [--iters 200] [--halo 8]
"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import fluid_amd
from fluid_amd import engine as E, scenes, slab as S

ap = argparse.ArgumentParser()
ap.add_argument("--grid", type=int, nargs="+", default=[512])
ap.add_argument("--ranks", type=int, nargs="+", default=[2, 4, 8])
ap.add_argument("--iters", type=int, default=200)
ap.add_argument("--halo", type=int, nargs="+", default=[2, 4, 8])
ap.add_argument("--overlap", type=int, nargs="+", default=[0, 1, 2])
a = ap.parse_args()
w, h, d = (a.grid * 3)[:3] if len(a.grid) == 1 else a.grid
for ranks in a.ranks:
    rank = ranks // 2 if ranks > 2 else 0   # an interior rank where there is one
    p = fluid_amd.default_params(w, h, d, 0)
    for halo in a.halo:
        for overlap in a.overlap:
            drv = S.SlabDriver(p, rank, ranks, pressure_iterations=a.iters, device=0, halo_depth=halo,
                               overlap=overlap)
            drv.attach_loopback(rank > 0 or ranks > 2, True)
            z0, n = drv.slab
            drv.upload_image(E.CELL_TYPES, scenes.full_fluid_types((n, h, w), z0, d))
            div = scenes.full_fluid_divergence((min(n, 32), h, w))
            drv.engine.upload_image(E.DIVERGENCES, np.tile(div, (n // div.shape[0], 1, 1)))
            drv.pressure_step()
            drv.engine.sync()
            x0 = drv.stat(S.STAT_EXCHANGES)
            drv.engine.enable_timing(True)
            drv.engine.reset_timing()
            reps = 3
            t0 = time.perf_counter()
            for _ in range(reps):
                drv.pressure_step()
            drv.engine.sync()
            dt = (time.perf_counter() - t0) / reps
            gpu_ms, calls = drv.engine.section_time_ms("12_solve_pressure")
            print(f"ranks {ranks} (slab {n} planes) halo {drv.stat(S.STAT_EFFECTIVE_HALO)} overlap {overlap}: "
                  f"{1e3 * dt / a.iters:.4f} ms/sweep -> {a.iters / dt:9.1f} iterations/s aggregate, "
                  f"kernel {gpu_ms / reps / a.iters:.4f} ms/sweep, "
                  f"{(drv.stat(S.STAT_EXCHANGES) - x0) // reps} exchanges per {a.iters} sweeps", flush=True)
            drv.close()
