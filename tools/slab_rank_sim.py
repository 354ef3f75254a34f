"""Dev tool: rehearse ONE rank of an N-GPU Z-slab run on a single GPU.  The rank owns an interior slab
(neighbours on both sides); halo exchanges are replaced by device copies of the same size
(transport "loopback"), so kernel time, redundant ghost-region compute and host overhead are real and
only the wire/NCCL latency is missing.  Prints ms per sweep and the implied aggregate iterations/s.

    python tools/slab_rank_sim.py [--grid 512] [--ranks 8] [--iters 200] [--halo 8]
"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import fluid_amd
from fluid_amd import engine as E, scenes
from fluid_amd.slab import DistContext, GpuSlabCompute, SlabPressureSolver, partition_z

ap = argparse.ArgumentParser()
ap.add_argument("--grid", type=int, default=512)
ap.add_argument("--ranks", type=int, nargs="+", default=[2, 4, 8])
ap.add_argument("--iters", type=int, default=200)
ap.add_argument("--halo", type=int, nargs="+", default=[2, 4, 8])
a = ap.parse_args()
n = a.grid
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
for ranks in a.ranks:
    rank = ranks // 2 if ranks > 2 else 0   # an interior rank where there is one
    slab = partition_z(n, ranks)[rank]
    p = fluid_amd.default_params(n, n, n, 0)
    comp = GpuSlabCompute(p, slab, dev)
    shape = (slab[1], n, n)
    comp.upload(E.CELL_TYPES, scenes.full_fluid_types(shape, slab[0], n))
    div = scenes.full_fluid_divergence((min(slab[1], 32), n, n))
    comp.upload(E.DIVERGENCES, np.tile(div, (slab[1] // div.shape[0], 1, 1)))
    for halo in a.halo:
        ctx = DistContext(rank, ranks, dev, "none")
        solver = SlabPressureSolver((n, n, n), a.iters, ctx, comp, slab, transport="loopback",
                                    halo_depth=halo)
        solver.step()
        comp.sync()
        solver.exchanges = 0
        comp.engine.enable_timing(True)
        comp.engine.reset_timing()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            solver.step()
        comp.sync()
        dt = (time.perf_counter() - t0) / reps
        gpu_ms, calls = comp.engine.section_time_ms("12_solve_pressure")
        comp.engine.enable_timing(False)
        print(f"   kernel time {gpu_ms / reps / a.iters:.4f} ms/sweep ({calls // reps} sweeps counted)")
        print(f"ranks {ranks} (slab {slab[1]} planes) halo {solver.effective_halo()} "
              f"overlapped {solver.overlapped // (reps + 1)}: "
              f"{1e3 * dt / a.iters:.4f} ms/sweep -> {a.iters / dt:9.1f} iterations/s, "
              f"{solver.exchanges // reps} exchanges per {a.iters} sweeps")
    comp.close()
