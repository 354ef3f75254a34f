"""The whole simulation step on Z slabs (slab.SlabSimulation: section order, ghost-plane exchanges,
particle hand-over) with world_size 2 and 3 over gloo on CPU.  The per-slab compute is the oracle on
poisoned global arrays (tests/host_standin.py), so a missing or too-shallow exchange cannot pass; the
result must equal the single-domain oracle bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, size, iters, steps, grouped, out_dir, intended=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist

    import fluid_amd  # noqa: F401
    from fluid_amd import engine as E
    from fluid_amd.slab import SlabSimulation, init_distributed, partition_z
    from host_standin import HostGlobalCompute
    from test_slab_step_gpu import drift, scene_params

    ctx = init_distributed(rank, backend="gloo")
    params, cap = scene_params(size, intended)
    slab = partition_z(size[2], world)[rank]
    comp = HostGlobalCompute(params, slab, cap, iters)
    sim = SlabSimulation(params, cap, iters, ctx, compute=comp, grouped=grouped,
                         diffuse_mode=E.DIFFUSE_INTENDED if intended else E.DIFFUSE_REFERENCE_EXACT)
    sim.run_init()
    sim.run_step()  # cells become active first: velocities of newly active faces are replaced (05)
    sim.upload_image_global(E.VELOCITIES_1, drift((size[2], size[1], size[0])))
    for _ in range(steps):
        sim.run_step()
    out = {name: sim.gather_image(img) for name, img in [
        ("velocities_1", E.VELOCITIES_1), ("cell_types", E.CELL_TYPES),
        ("pressures_1", E.PRESSURES_1), ("pressures_2", E.PRESSURES_2),
        ("divergences", E.DIVERGENCES), ("particle_densities", E.PARTICLE_DENSITIES_IMG)]}
    out["particles"] = sim.gather_particles()
    if rank == 0:
        np.savez(os.path.join(out_dir, "result.npz"), migrated=sim.migrated, **out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,size,iters,steps,grouped,intended", [
    (2, (16, 12, 16), 6, 4, True, False), (3, (12, 10, 18), 5, 4, True, False),
    (2, (16, 12, 16), 6, 4, False, False),   # the section list, one call per section
    (2, (14, 12, 16), 6, 3, True, False),    # width not a multiple of 4: 09, 10, 11 stay separate
    (2, (16, 12, 16), 6, 3, True, True),    # 09_diffuse in intended mode: V2 ghost planes, no 09+10+11 group
])
def test_slab_simulation_over_gloo_matches_oracle(world, size, iters, steps, grouped, intended, tmp_path):
    import torch.multiprocessing as mp

    from helpers import assert_bit_equal
    from oracle_binding import OracleState
    from test_slab_step_gpu import drift, scene_params

    mp.start_processes(_worker, args=(world, _free_port(), size, iters, steps, grouped, str(tmp_path), intended),
                       nprocs=world, join=True, start_method="spawn")
    got = np.load(os.path.join(str(tmp_path), "result.npz"))
    params, cap = scene_params(size, intended)
    st = OracleState(params, cap, iters, diffuse_mode=1 if intended else 0)
    st.run_init()
    st.run_step()
    st.velocities_1[...] = drift(st.shape)
    for _ in range(steps):
        st.run_step()
    for name in ("cell_types", "particle_densities", "divergences", "pressures_1", "pressures_2",
                 "velocities_1", "particles"):
        assert_bit_equal(got[name], getattr(st, name), f"{world} slabs, {name}")
    assert int(got["migrated"]) > 0
