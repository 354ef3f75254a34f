"""Dev tool: the dam-break full step through the slab driver with ONE rank (whole grid) against
fluid_run_step on the same scene: what the driver's section-by-section schedule costs.
    python tools/slab_one_rank_step.py [grid=512] [iters=200]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fluid_amd
from fluid_amd import engine as E, slab as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
p, cap = fluid_amd.dam_break_params(n, n, n)
def timed(step, sync, warm=10, steps=20):
    for _ in range(warm): step()
    sync(); t0 = time.perf_counter()
    for _ in range(steps): step()
    sync(); return 1e3 * (time.perf_counter() - t0) / steps
with fluid_amd.FluidEngine(p, particle_capacity=cap, pressure_iterations=iters) as eng:
    eng.run_init()
    print(f"fluid_run_step        : {timed(eng.run_step, eng.sync):8.3f} ms/step")
with S.SlabDriver(p, 0, 1, particle_capacity=cap, pressure_iterations=iters, device=0) as drv:
    drv.run_init()
    print(f"slab driver, 1 rank   : {timed(drv.run_step, drv.engine.sync):8.3f} ms/step   quiet bricks {drv.engine.get_stat(E.STAT_QUIET_BRICKS)} of {drv.engine.get_stat(E.STAT_BRICKS)}")
for ranks, rank in ((2, 0), (8, 0)):
    with S.SlabDriver(p, rank, ranks, particle_capacity=cap, pressure_iterations=iters, device=0) as drv:
        drv.attach_loopback(rank > 0, True)
        drv.run_init()
        print(f"rank {rank} of {ranks} (loopback): {timed(drv.run_step, drv.engine.sync):8.3f} ms/step   quiet bricks {drv.engine.get_stat(E.STAT_QUIET_BRICKS)} of {drv.engine.get_stat(E.STAT_BRICKS)}")
