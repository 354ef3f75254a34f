"""The whole simulation step on Z slabs (the C++ driver of include/fluid_slab.h on the HIP engine) with 2
and 3 ranks on the one GPU of the test box (planes and particle lists travel over gloo through host
staging), against the single-domain oracle: cell types, velocities, pressures and the particle buffer bit for bit after several steps of a
scene whose water and particles cross the slab faces."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def scene_params(size, intended=False, zspan=None):
    """Dam-break block placed across the slab faces: 8 particles per cell, z from 25 % to 75 %
    (zspan = (first, extent) as fractions of the depth puts it elsewhere).
    intended: a diffusion coefficient for 09_diffuse in FLUID_DIFFUSE_INTENDED mode."""
    import fluid_amd
    w, h, d = size
    p, _ = fluid_amd.dam_break_params(w, h, d)
    z_first, z_ext = zspan if zspan else (0.25, 0.5)
    ext = (0.5 * w, 0.5 * h, z_ext * d)
    res = tuple(max(1, int(round(2 * e))) for e in ext)
    p.particle_spawn_cube_resolution[:] = res
    p.particle_spawn_cube_volume = res[0] * res[1] * res[2]
    p.particle_spawn_cube_offset[:] = (0.25 * w, 0.15 * h, z_first * d)
    p.particle_spawn_cube_size[:] = ext
    cap = res[0] * res[1] * res[2] + 37  # a few inactive slots at the end
    p.particle_compute_size[:] = (cap, 1)
    if intended:
        p.diffuse_k = 1.5
    p.time_delta = 0.04                  # bigger steps: particles cross slab faces within a few
    return p, cap


def drift(shape, fast=0.0):
    """Initial velocity field uploaded after init: a steady drift along +z (and a little -x).
    fast: the z speed, in cells per time unit (default 5 = 0.2 cells per step of this scene)."""
    v = np.zeros(shape + (4,), np.float32)
    v[..., 2] = fast if fast else 5.0
    v[..., 0] = -1.5
    return v


def _worker(rank, world, port, size, iters, steps, grouped, out_dir, intended=False, fast=0.0, plain_slots=False,
            zspan=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK="0")
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist

    import fluid_amd  # noqa: F401
    from fluid_amd import engine as E
    from fluid_amd import slab as S
    from test_slab_step_gpu import drift, scene_params

    dist.init_process_group(backend="gloo")
    params, cap = scene_params(size, intended, zspan)
    # the product's C++ driver on the HIP engine; only the wire differs from an N-GPU run: the ranks
    # share GPU 0, where RCCL refuses to run, so the planes are staged through the host over gloo
    sim = S.SlabDriver(params, rank, world, particle_capacity=cap, pressure_iterations=iters, device=0,
                       grouped=grouped,
                       diffuse_mode=E.DIFFUSE_INTENDED if intended else E.DIFFUSE_REFERENCE_EXACT)
    sim.attach_torch_transport(device_memory=True)
    if world == 3:   # one workgroup per brick layer in the skipping passes, as on grids of >= 4096 bricks
        sim.engine.set_option(E.OPT_QUIET_BRICKS, 2)
        # ... and the compact particle storage squeezed whenever it has a hole (by itself: from 65536 holes on)
        sim.engine.set_option(E.OPT_PARTICLE_SORT, 3)
    if plain_slots:  # FLUID_OPT_PARTICLE_SORT: 1 = the compact storage is never sorted by bin; 2 = sorted whatever its
        # size (by itself: from 4 M entries); 4 = sorted once and never again (strays and adopted particles pile up)
        sim.engine.set_option(E.OPT_PARTICLE_SORT, int(plain_slots))
    sim.run_init()
    sim.run_step()  # cells become active first: velocities of newly active faces are replaced (05)
    sim.upload_image_global(E.VELOCITIES_1, drift((size[2], size[1], size[0]), fast))
    for _ in range(steps):
        sim.run_step()
    out = {name: sim.gather_image(img) for name, img in [
        ("velocities_1", E.VELOCITIES_1), ("cell_types", E.CELL_TYPES),
        ("pressures_1", E.PRESSURES_1), ("pressures_2", E.PRESSURES_2),
        ("divergences", E.DIVERGENCES), ("particle_densities", E.PARTICLE_DENSITIES_IMG)]}
    out["particles"] = sim.gather_particles()
    t = torch.tensor([sim.stat(i) for i in range(8)] + [sim.engine.get_stat(E.STAT_QUIET_BRICKS),
                                                        sim.engine.get_stat(E.STAT_PARTICLE_ENTRIES),
                                                        sim.engine.get_stat(E.STAT_OWNED_SQUEEZES),
                                                        sim.engine.get_stat(E.STAT_PARTICLE_SORTS),
                                                        sim.stat(S.STAT_DRY_FACE_SKIPS)],
                     dtype=torch.int64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        np.savez(os.path.join(out_dir, "result.npz"), stats=t.numpy(), **out)
    dist.barrier()
    sim.close()
    dist.destroy_process_group()


def _check(got, world, size, iters, steps, intended=False, fast=0.0, zspan=None):
    from helpers import assert_bit_equal
    from oracle_binding import OracleState

    params, cap = scene_params(size, intended, zspan)
    st = OracleState(params, cap, iters, diffuse_mode=1 if intended else 0)
    st.run_init()
    st.run_step()
    st.velocities_1[...] = drift(st.shape, fast)
    for _ in range(steps):
        st.run_step()
    for name in ("cell_types", "particle_densities", "divergences", "pressures_1", "pressures_2",
                 "velocities_1", "particles"):
        assert_bit_equal(got[name], getattr(st, name), f"{world} slabs, {name}")
    return st


@pytest.mark.parametrize("world,size,iters,steps,grouped,intended,plain_slots", [
    (2, (32, 24, 16), 12, 6, True, False, False), (3, (64, 16, 24), 9, 5, True, False, False),
    (2, (32, 24, 16), 12, 4, False, False, False),  # the section list, one kernel per section
    (2, (30, 24, 16), 12, 4, True, False, False),   # width not a multiple of 4
    (2, (32, 24, 16), 12, 4, True, True, False),   # 09_diffuse in intended mode: V2 ghost planes, no 09+10+11 group
    (3, (64, 32, 48), 8, 60, True, False, False),  # a long run: the block collapses across both faces and spreads
    (2, (32, 24, 16), 12, 6, True, False, 1),   # FLUID_OPT_PARTICLE_SORT = 1: never sorted
    (2, (32, 24, 16), 12, 6, True, False, 2),   # sorted by bin when the policy says so
    (2, (32, 24, 16), 12, 8, True, False, 4),   # sorted once: leavers leave holes, the adopted sit behind the bins
    (3, (64, 32, 48), 8, 30, True, False, 4),   # ... over a run in which the block collapses across both faces
])
def test_slab_simulation_matches_oracle(world, size, iters, steps, grouped, intended, plain_slots, tmp_path):
    import torch.multiprocessing as mp

    from fluid_amd import slab as S

    mp.start_processes(_worker, args=(world, _free_port(), size, iters, steps, grouped, str(tmp_path), intended,
                                      0.0, plain_slots),
                       nprocs=world, join=True, start_method="spawn")
    got = np.load(os.path.join(str(tmp_path), "result.npz"))
    st = _check(got, world, size, iters, steps, intended)
    # a slab stores (and 01, 14 and the search for leavers look at) the particles it owns, not a slot per particle of
    # the run: the largest storage of the ranks holds fewer entries than the run has particles
    _, cap = scene_params(size, intended)
    entries, squeezes = int(got["stats"][9]), int(got["stats"][10])
    assert 0 < entries < cap, (entries, cap)
    sorts = int(got["stats"][11])
    if world == 3 and not plain_slots:   # mode 3: sorted before every 01 (a sort drops the holes: no squeeze is needed)
        assert sorts >= steps and squeezes == 0
    if plain_slots == 1:
        assert sorts == 0
    if plain_slots in (2, 4):
        assert sorts >= 1
    # the scene did what the test is for: water on both sides of a face, particles changed owner
    d = size[2]
    face = d // world
    assert np.any(st.cell_types[face - 1] == 2) and np.any(st.cell_types[face] == 2)
    assert int(got["stats"][S.STAT_MIGRATED]) > 0
    assert int(got["stats"][S.STAT_SAMPLER_RERUNS]) == 0


@pytest.mark.parametrize("world,steps", [(2, 8), (2, 40), (3, 40)])
def test_slab_loop_leaves_dry_faces_out(world, steps, tmp_path):
    """A block of water well inside the lowest slab, drifting towards the face: while no water lies within eight
    planes of a face on either side, the Jacobi loop exchanges nothing there (the planes it would move hold the
    constants of non-water cells, primed once per loop in all three working buffers) — the step's table of boxes
    tells both ranks of a face the same thing; when the water comes near, the exchanges are back.  Bit-identical
    to the oracle throughout."""
    import torch.multiprocessing as mp

    from fluid_amd import slab as S

    size, iters, zspan = (32, 24, 64 * world), 12, (0.05 * 2 / world, 0.30 * 2 / world)   # z 6.4 .. 44.8
    mp.start_processes(_worker, args=(world, _free_port(), size, iters, steps, True, str(tmp_path), False, 0.0, False,
                                      zspan),
                       nprocs=world, join=True, start_method="spawn")
    got = np.load(os.path.join(str(tmp_path), "result.npz"))
    st = _check(got, world, size, iters, steps, zspan=zspan)
    skips, exchanges = int(got["stats"][12]), int(got["stats"][S.STAT_EXCHANGES])
    assert skips > 0, (skips, exchanges)
    print("dry-face skips", skips, "exchanges", exchanges, "water in the last layer of slab 0:",
          bool(np.any(st.cell_types[48:64] == 2)))
    if steps >= 40:   # the water has reached the last brick layer of the lowest slab: its upper face is wet again
        assert np.any(st.cell_types[48:64] == 2)


@pytest.mark.parametrize("world,size", [(2, (64, 64, 96)), (3, (256, 48, 96))])
def test_slab_simulation_skips_quiet_bricks_like_the_single_gpu_step(world, size, tmp_path):
    """Ten steps of a block of water that straddles the slab faces in a tank with room around it: from the
    third step on every slab skips the bricks far from the water (quiet_bricks.h) — across a face by the
    neighbouring slab's edge layer of activity bricks, exchanged after 06 — and launches its pressure loop
    over the union of its box with its neighbours'.  Bit-identical to the oracle, and bricks WERE skipped."""
    import torch.multiprocessing as mp

    iters, steps = 6, 10
    mp.start_processes(_worker, args=(world, _free_port(), size, iters, steps, True, str(tmp_path), False),
                       nprocs=world, join=True, start_method="spawn")
    got = np.load(os.path.join(str(tmp_path), "result.npz"))
    _check(got, world, size, iters, steps)
    assert int(got["stats"][8]) > 0    # FLUID_STAT_QUIET_BRICKS of the last step, max over the ranks


@pytest.mark.parametrize("world,size,fast,expect_wide", [
    (2, (32, 24, 16), 55.0, False),    # 2.2 cells per step: more than the default two ghost planes serve
    (3, (64, 16, 24), 260.0, True),    # 10.4 cells per step across slabs of 8 planes: the wide source
])
def test_fast_flow_widens_the_sampler_halo_and_never_fails(world, size, fast, expect_wide, tmp_path):
    """SURVEY.md F6 on the engine: a z-drift of more than a cell per step.  One rank's 07 kernel flags a
    tap beyond its ghost planes; every rank redoes the pass with the halo max |v.z| dt calls for — from
    the image's own ghost planes, or from the wide source filled by whoever owns the planes — and
    particles that cross more than one slab in a step travel from neighbour to neighbour.  Bit-identical
    to the single-domain oracle."""
    import torch.multiprocessing as mp

    from fluid_amd import slab as S

    iters, steps = 6, 3
    mp.start_processes(_worker, args=(world, _free_port(), size, iters, steps, True, str(tmp_path), False,
                                      fast),
                       nprocs=world, join=True, start_method="spawn")
    got = np.load(os.path.join(str(tmp_path), "result.npz"))
    _check(got, world, size, iters, steps, fast=fast)
    stats = got["stats"]
    assert int(stats[S.STAT_SAMPLER_RERUNS]) > 0
    assert (int(stats[S.STAT_SAMPLER_WIDE]) > 0) == expect_wide
    assert int(stats[S.STAT_MIGRATED]) > 0


@pytest.mark.parametrize("grouped", [True, False])
def test_whole_step_rehearsal_by_copies_and_through_rccl_are_identical(grouped):
    """One interior rank of a 3-way run, both neighbours played by itself: the whole step's exchanges (ghost
    planes of every image, activity-brick layers, the status reduction) once as device copies and once as
    ncclSend / ncclRecv / ncclAllReduce on a communicator of one.  Every image and the particle buffer end
    bit-identical: the RCCL calls move the same bytes in the same stream order."""
    import fluid_amd  # noqa: F401
    from fluid_amd import engine as E
    from fluid_amd import slab as S
    from helpers import assert_bit_equal

    size, iters = (64, 24, 48), 12
    params, cap = scene_params(size)
    got = {}
    for wire in ("copies", "rccl"):
        with S.SlabDriver(params, 1, 3, particle_capacity=cap, pressure_iterations=iters, device=0,
                          grouped=grouped) as drv:
            (drv.attach_loopback if wire == "copies" else drv.attach_rccl_self)(True, True)
            drv.run_init()
            drv.run_step()
            z0, n = drv.slab
            drv.engine.upload_image(E.VELOCITIES_1, drift((n, size[1], size[0])))
            for _ in range(4):
                drv.run_step()
            drv.engine.sync()
            assert drv.stat(S.STAT_EXCHANGES) > 0
            got[wire] = {img: drv.engine.download_image(img) for img in
                         (E.VELOCITIES_1, E.CELL_TYPES, E.PRESSURES_1, E.PRESSURES_2, E.DIVERGENCES,
                          E.PARTICLE_DENSITIES_IMG)}
            got[wire]["particles"] = drv.engine.download_particles()
    for k in got["copies"]:
        assert_bit_equal(got["copies"][k], got["rccl"][k], f"whole step, copies vs RCCL, {k}")
    assert np.isfinite(got["rccl"][E.PRESSURES_1]).all()
