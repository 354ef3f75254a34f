"""The GPU slab path (GpuSlabCompute: torch-owned arena, engine on torch's stream, plane views,
slab kernels with real neighbour data in the ghost planes) rehearsed with 2 ranks on the ONE GPU
of the test box.  RCCL refuses two ranks on one device, so the planes travel over gloo through
host staging (transport="staged"); the schedule and every device-side piece are the product's.
Result must equal the single-domain oracle bit for bit."""
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


def _worker(rank, world, port, size, iters, seed, variant, halo, out_dir, edge=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK="0")
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist

    import fluid_amd  # noqa: F401
    from fluid_amd import engine as E
    from fluid_amd.slab import (DistContext, GpuSlabCompute, SlabPressureSolver, partition_z)
    from helpers import random_state

    dist.init_process_group(backend="gloo")
    torch.cuda.set_device(0)
    ctx = DistContext(rank, world, torch.device("cuda", 0), "gloo")
    w, h, d = size
    st = random_state(size, seed=seed, iters=iters)
    slab = partition_z(d, world)[rank]
    comp = GpuSlabCompute(st.params, slab, ctx.device, pressure_kernel=variant, edge_stream=edge is True)
    solver = SlabPressureSolver(size, iters, ctx, comp, slab, transport="staged", halo_depth=halo)
    if edge == "before":   # only the pass before an exchange is split
        solver.overlap = "before"
    z0, n = slab
    comp.upload(E.CELL_TYPES, st.cell_types[z0:z0 + n])
    comp.upload(E.DIVERGENCES, st.divergences[z0:z0 + n])
    solver.exchange(E.CELL_TYPES)
    solver.step()
    a1, a2 = solver.gather_pressures()
    comp.upload(E.PRESSURES_1, st.pressures_1[z0:z0 + n])
    comp.upload(E.PRESSURES_2, st.pressures_2[z0:z0 + n])
    solver.solve(iters + 1)
    b1, b2 = solver.gather_pressures()
    if rank == 0:
        np.savez(os.path.join(out_dir, "result.npz"), a1=a1, a2=a2, b1=b1, b2=b2)
    dist.barrier()
    solver.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,size,iters,variant,halo,edge", [
    (2, (64, 24, 20), 6, 0, 2, False),     # working-buffer loop, two sweeps per exchange
    (2, (64, 24, 20), 21, 0, 8, False),    # eight sweeps per exchange (redundant ghost-region compute), odd tail
    (2, (64, 24, 40), 12, 0, 6, False),    # six
    (3, (260, 9, 11), 5, 0, 8, False),     # slabs of 4/4/3 planes clip the halo to 2
    (3, (256, 12, 13), 8, 7, 4, False),    # explicit fast-path kernel option
    (2, (64, 24, 20), 7, 2, 8, False),     # general kernel on the images: one plane per sweep
    (2, (17, 9, 8), 4, 0, 8, False),       # width not a multiple of 4: falls back to the images as well
    (2, (64, 24, 40), 21, 0, 8, False),    # slabs of 20 planes, halo 8: split passes around the exchanges
    (3, (256, 12, 30), 14, 0, 4, False),   # three ranks, halo 4 on slabs of 10 planes, split passes
    (2, (512, 7, 24), 12, 0, 4, False),    # two x tiles per row
    (2, (64, 24, 40), 21, 0, 8, True),    # the same with the EDGES launches on the second engine stream
    (3, (256, 12, 30), 14, 0, 4, True),
    (2, (64, 24, 40), 21, 0, 8, "before"),  # half-overlapped schedule
])
def test_gpu_slab_solver_equals_single_domain_oracle(world, size, iters, variant, halo, edge, tmp_path):
    import torch.multiprocessing as mp

    from helpers import assert_bit_equal, random_state

    seed = 33
    mp.start_processes(_worker, args=(world, _free_port(), size, iters, seed, variant, halo,
                                      str(tmp_path), edge),
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


def test_single_rank_slab_bench_path_runs():
    """world_size 1 through the same solver + benchmark code the multi-GPU bench uses (RCCL group
    of one)."""
    import subprocess
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0",
               WORLD_SIZE="1", LOCAL_RANK="0")
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import fluid_amd\n"
        "from fluid_amd.slab import SlabPressureSolver, init_distributed\n"
        "ctx = init_distributed(0)\n"
        "s = SlabPressureSolver.create_gpu((128, 128, 64), 20, ctx)\n"
        "r = s.benchmark(2, 1)\n"
        "assert r['wall_s'] > 0 and r['kernel_ms_per_sweep'] > 0, r\n"
        "print('ok', r)\n" % (ROOT, os.path.join(ROOT, "tests")))
    res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True,
                         timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "ok" in res.stdout
