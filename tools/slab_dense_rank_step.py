"""Dev tool: one rank of an N-way Z-slab run of the FULL TANK (8 particles per cell everywhere), neighbours
played by itself (loopback): the step of a middle rank, section by section, and the size of its particle storage.
    python tools/slab_dense_rank_step.py [grid=512] [ranks=8] [iters=200]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fluid_amd
from fluid_amd import engine as E, slab as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ranks = int(sys.argv[2]) if len(sys.argv) > 2 else 8
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 200
size = (n - 4.0,) * 3
res = tuple(int(round(2.0 * v)) for v in size)
cap = res[0] * res[1] * res[2]
p = fluid_amd.default_params(n, n, n, cap)
p.particle_spawn_cube_resolution[:] = res
p.particle_spawn_cube_volume = cap
p.particle_spawn_cube_offset[:] = (2.0, 2.0, 2.0)
p.particle_spawn_cube_size[:] = size
rank = ranks // 2
# the loop's exchange schedule: the driver's default (8 planes per exchange, both passes around an exchange split so
# that it runs beside them — what a real wire wants) and what bench.py's probe picks in loopback, where an
# exchange is a device copy (3 planes per exchange, in line)
for mode, name, halo, overlap in ((0, "compact storage, h = 8, split passes", 8, S.OVERLAP_BOTH),
                                  (0, "compact storage, h = 3, exchanges in line", 3, S.OVERLAP_NONE)):
    with S.SlabDriver(p, rank, ranks, particle_capacity=cap, pressure_iterations=iters, device=0) as drv:
        drv.engine.set_option(E.OPT_PARTICLE_SORT, mode)
        drv.set_option(S.OPT_HALO_DEPTH, halo)
        drv.set_option(S.OPT_OVERLAP, overlap)
        drv.attach_loopback(True, True)
        drv.run_init()
        for _ in range(3):
            drv.run_step()
        drv.engine.sync()
        t0 = time.perf_counter()
        for _ in range(5):
            drv.run_step()
        drv.engine.sync()
        ms = 1e3 * (time.perf_counter() - t0) / 5
        drv.engine.enable_timing(True)
        drv.engine.reset_timing()
        for _ in range(2):
            drv.run_step()
        drv.engine.sync()
        t = drv.engine.section_times()
        sec = {k[:24]: round(v[0] / 2, 3) for k, v in sorted(t.items(), key=lambda kv: -kv[1][0]) if v[1] and v[0] / 2 >= 0.05}
        print(f"rank {rank} of {ranks}, full tank {n}^3, {cap} slots, {name}: {ms:8.3f} ms/step   "
              f"entries {drv.engine.get_stat(E.STAT_PARTICLE_ENTRIES)}   {sec}")
