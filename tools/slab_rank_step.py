"""Dev tool: the dam-break step of ONE rank of an N-way Z-slab run (loopback in place of the wire), for a kernel
timeline (tools/step_timeline.sh SCRIPT=tools/slab_rank_step.py).   python tools/slab_rank_step.py [grid=512] [ranks=2] [rank=0]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fluid_amd
from fluid_amd import engine as E, slab as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ranks = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rank = int(sys.argv[3]) if len(sys.argv) > 3 else 0
p, cap = fluid_amd.dam_break_params(n, n, n)
with S.SlabDriver(p, rank, ranks, particle_capacity=cap, pressure_iterations=200, device=0) as drv:
    drv.attach_loopback(rank > 0, rank < ranks - 1)
    drv.run_init()
    for _ in range(6):
        drv.run_step()
    drv.engine.sync()
    t0 = time.perf_counter()
    for _ in range(5):
        drv.run_step()
    drv.engine.sync()
    print(f"rank {rank} of {ranks}: {1e3 * (time.perf_counter() - t0) / 5:.3f} ms/step, dry-face skips {drv.stat(S.STAT_DRY_FACE_SKIPS)}, exchanges {drv.stat(S.STAT_EXCHANGES)}")
