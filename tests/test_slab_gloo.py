"""Multi-process test of the Z-slab pressure solve on CPU: world_size 2 and 3 over the gloo backend.
The slab decomposition, the halo-exchange schedule (deep halos, split passes) and the ping-pong parity
are the product's C++ driver (csrc/slab_driver.hip through include/fluid_slab.h); the per-slab sweep is
the CPU oracle behind the driver's compute callbacks (host_standin.HostSlabCompute) and the planes
travel through its transport callbacks over gloo.  The result must equal the oracle's single-domain
result bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from fluid_amd.slab import partition_z  # noqa: E402


def test_partition_z_is_balanced_contiguous_and_complete():
    for depth, world in [(512, 8), (512, 3), (10, 4), (7, 7), (64, 1)]:
        parts = partition_z(depth, world)
        assert len(parts) == world
        assert parts[0][0] == 0
        for (a, n), (b, _) in zip(parts, parts[1:]):
            assert a + n == b
        assert parts[-1][0] + parts[-1][1] == depth
        counts = [n for _, n in parts]
        assert max(counts) - min(counts) <= 1 and min(counts) >= 1
    with pytest.raises(ValueError):
        partition_z(3, 4)
    with pytest.raises(ValueError):
        partition_z(8, 0)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, size, iters, seed, max_sweeps, halo, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist

    import fluid_amd  # noqa: F401
    from fluid_amd import engine as E
    from fluid_amd import slab as S
    from helpers import random_state
    from host_standin import HostSlabCompute

    S.init_distributed(rank, backend="gloo")
    w, h, d = size
    st = random_state(size, seed=seed, iters=iters)  # every rank builds the same global scene
    slab = S.partition_z(d, world)[rank]
    comp = HostSlabCompute(st.params, slab, max_sweeps=max_sweeps)
    # the C++ schedule (csrc/slab_driver.hip) over the oracle stand-in and gloo
    drv = S.SlabDriver(st.params, rank, world, pressure_iterations=iters, halo_depth=halo, compute=comp)
    assert drv.slab == slab
    drv.attach_torch_transport()
    z0, n = slab
    comp.upload(E.CELL_TYPES, st.cell_types[z0:z0 + n])
    comp.upload(E.DIVERGENCES, st.divergences[z0:z0 + n])
    drv.exchange_image(E.CELL_TYPES, 1)

    def pressures():
        return drv.gather_image(E.PRESSURES_1), drv.gather_image(E.PRESSURES_2)

    # case A: the step as the bench runs it (clears + loop)
    drv.pressure_step()
    a1, a2 = pressures()
    # case B: arbitrary uploaded pressures, odd iteration count
    comp.upload(E.PRESSURES_1, st.pressures_1[z0:z0 + n])
    comp.upload(E.PRESSURES_2, st.pressures_2[z0:z0 + n])
    drv.solve(iters + 1)
    b1, b2 = pressures()
    thinnest = min(m for _, m in S.partition_z(d, world))
    eff = drv.stat(S.STAT_EFFECTIVE_HALO)
    assert eff == max(1, min(halo, 8, thinnest) - (min(halo, 8, thinnest) % 2 if max_sweeps == 2 and
                                                    min(halo, 8, thinnest) >= 2 else 0))
    if max_sweeps >= 2 and eff >= 4 and thinnest > 2 * eff:
        assert drv.stat(S.STAT_OVERLAPPED) > 0  # the split-pass schedule ran
    else:
        assert drv.stat(S.STAT_OVERLAPPED) == 0
    # the measurement loop of bench.py --gpus N, including its inline-vs-overlapped probe
    res = drv.benchmark(1, 0)
    assert res["wall_s"] > 0 and res["halo_overlap"]["probed"] and "step_ms_inline" in res["halo_overlap"]
    # the other two schedules (exchanges in line; only the pass before an exchange split): same iterates
    for mode in (S.OVERLAP_NONE, S.OVERLAP_BEFORE):
        drv.set_option(S.OPT_OVERLAP, mode)
        comp.upload(E.PRESSURES_1, st.pressures_1[z0:z0 + n])
        comp.upload(E.PRESSURES_2, st.pressures_2[z0:z0 + n])
        drv.solve(iters + 1)
        c1, c2 = pressures()
        if rank == 0:
            assert np.array_equal(c1.view(np.uint32), b1.view(np.uint32))
            assert np.array_equal(c2.view(np.uint32), b2.view(np.uint32))
    if rank == 0:
        np.savez(os.path.join(out_dir, "result.npz"), a1=a1, a2=a2, b1=b1, b2=b2)
    dist.barrier()
    drv.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,size,iters,max_sweeps,halo", [
    (2, (12, 10, 16), 6, 2, 2),    # two sweeps per exchange
    (2, (12, 10, 16), 13, 2, 8),   # eight sweeps per exchange, odd tail
    (3, (8, 9, 11), 5, 2, 8),      # slabs of 4/4/3 planes clip the halo to 2
    (3, (8, 9, 20), 9, 2, 4),      # 7/7/6 planes, halo 4
    (2, (10, 8, 9), 4, 1, 8),      # one sweep per launch: one plane per sweep
    (2, (12, 10, 20), 13, 2, 4),   # slabs of 10 planes, halo 4: exchanges overlap split passes
    (3, (8, 9, 30), 18, 2, 4),     # three ranks (the middle one has two neighbours), overlapped
    (2, (8, 8, 40), 20, 2, 8),     # halo 8 on slabs of 20 planes, overlapped
    # three sweeps per pass (kernels_pressure_fused3.h): launches of 3 + 3 + 2 between two exchanges of 8
    (2, (12, 10, 16), 13, 3, 8),   # slabs of 8 planes, no split passes; 13 = 3 + 3 + 2 | 3 + 2, then 14
    (2, (8, 8, 40), 20, 3, 8),     # overlapped: the pass before an exchange is a pair, the one after a triple
    (3, (8, 9, 30), 18, 3, 6),     # halo 6: two triples per exchange, both split
    (3, (8, 9, 20), 9, 3, 3),      # halo 3: one triple per exchange
    (2, (8, 8, 40), 7, 3, 7),      # an odd halo: 3 + 3, one plane left over
])
def test_slab_solver_equals_single_domain_oracle(world, size, iters, max_sweeps, halo, tmp_path):
    import torch.multiprocessing as mp

    from helpers import assert_bit_equal, random_state

    seed = 21
    port = _free_port()
    mp.start_processes(_worker, args=(world, port, size, iters, seed, max_sweeps, halo, str(tmp_path)),
                       nprocs=world, join=True, start_method="spawn")
    got = np.load(os.path.join(str(tmp_path), "result.npz"))
    st = random_state(size, seed=seed, iters=iters)
    ref = st.copy()
    ref.run_section("12a_clear_pressures_1")
    ref.run_section("12b_clear_pressures_2")
    ref.solve_pressure(iters)
    assert_bit_equal(got["a1"], ref.pressures_1, "step P1")
    assert_bit_equal(got["a2"], ref.pressures_2, "step P2")
    ref = st.copy()
    ref.solve_pressure(iters + 1)
    assert_bit_equal(got["b1"], ref.pressures_1, "odd loop P1")
    assert_bit_equal(got["b2"], ref.pressures_2, "odd loop P2")
    assert np.any(ref.pressures_1 != st.pressures_1)
