"""Z-slab decomposition of the pressure solve across GPUs — one process per GPU.

The reference is single-GPU (main.cpp:43-48); this is the multi-GPU form of its hot loop
(`12_solve_pressure x N`, fluid_flow_sections.h:300-313).  The global grid is cut into contiguous
Z slabs (z is the slowest index, so a slab and each XY plane are contiguous in memory); every rank
owns `z_count` planes plus one ghost plane per side.  A 7-point sweep reads z±1, so after every
sweep each rank sends its first/last owned plane of the buffer just written to its lower/upper
neighbour and receives their planes into its ghost planes: point-to-point Send/Recv with the two
Z-neighbours only (2 of the 7 xGMI links per GPU), W*H*4 bytes per message — no collective on the
data path.  `torch.distributed` is the transport (backend "nccl" = RCCL on GPU tensors that alias
the engine's device memory; backend "gloo" on CPU tensors in the tests) — the decomposition and the
exchange schedule below are the same code in both cases, only the per-slab compute differs.

Ghost planes at a domain face are never written and stay 0 (= the reference's out-of-bounds load).
"""
import os
import time
from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np

from . import engine as E
from .params import FluidParams, default_params


def partition_z(depth: int, world: int) -> List[Tuple[int, int]]:
    """Balanced contiguous split of `depth` planes over `world` ranks: (z_begin, z_count) per rank.
    The first depth % world ranks get one extra plane.  Every rank must own at least one plane."""
    if world < 1 or depth < world:
        raise ValueError(f"cannot split {depth} planes over {world} ranks")
    base, extra = divmod(depth, world)
    out, z = [], 0
    for r in range(world):
        n = base + (1 if r < extra else 0)
        out.append((z, n))
        z += n
    return out


@dataclass
class DistContext:
    rank: int
    world: int
    device: object  # torch.device
    backend: str
    group: object = None


def init_distributed(local_rank: int = 0, backend: Optional[str] = None) -> DistContext:
    """Join the process group described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT."""
    import torch
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    use_gpu = backend in (None, "nccl") and torch.cuda.is_available()
    if backend is None:
        backend = "nccl" if use_gpu else "gloo"
    if use_gpu:
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
    else:
        device = torch.device("cpu")
    if not dist.is_initialized():
        kw = {"device_id": device} if use_gpu else {}
        dist.init_process_group(backend=backend, **kw)
    return DistContext(dist.get_rank(), dist.get_world_size(), device, backend)


# ---- per-slab compute backends ---------------------------------------------------------------------
class GpuSlabCompute:
    """The HIP engine on one slab.  Device memory is a torch tensor (so RCCL can address the halo
    planes as tensor views) handed to the engine as its arena; kernels run on torch's current
    stream, which is also the stream the NCCL ops synchronise with."""

    def __init__(self, params: FluidParams, slab: Tuple[int, int], device, pressure_kernel: int = 0):
        import torch

        self.torch = torch
        self.device = device
        nbytes = E.FluidEngine.required_arena_bytes(params, 0, slab=slab)
        if nbytes == 0:
            raise RuntimeError("invalid slab geometry")
        # One explicit side stream, made torch's current stream for this process: the engine's
        # kernels and the communicator's stream-ordering both follow it.  (The legacy null stream
        # has handle 0, which the C ABI reads as "create your own".)
        self.stream = torch.cuda.Stream(device=device)
        torch.cuda.set_stream(self.stream)
        self.arena = torch.zeros(nbytes + 256, dtype=torch.uint8, device=device)
        base = self.arena.data_ptr()
        self._pad = (-base) % 256
        assert self.stream.cuda_stream != 0
        self.engine = E.FluidEngine(
            params, particle_capacity=0, device=device.index if device.index is not None else -1,
            slab=slab, stream=self.stream.cuda_stream,
            arena=base + self._pad, arena_bytes=nbytes)
        self.engine.set_option(E.OPT_PRESSURE_KERNEL, pressure_kernel)
        self._base = base
        # working-buffer loop (fluid_pressure_loop_*) when the engine offers it for this grid
        self.fast = params.size[0] % 4 == 0 and pressure_kernel in (0, 5, 6, 7)

    def plane(self, image_id: int, local_z: int):
        ptr, nbytes = self.engine.image_plane_ptr(image_id, local_z)
        off = ptr - self._base
        view = self.arena[off:off + nbytes]
        dtype, _ = E.IMAGE_DTYPES[image_id]
        return view.view(self.torch.float32) if dtype == np.float32 else view

    def _view(self, ptr: int, nbytes: int, as_float: bool):
        off = ptr - self._base
        view = self.arena[off:off + nbytes]
        return view.view(self.torch.float32) if as_float else view

    def work_plane(self, which: int, local_z: int):
        """Plane of the buffer that holds the iterates of parity `which` during a loop: the
        engine's working buffer (fast path) or the pressure image itself."""
        if self.fast:
            return self._view(*self.engine.pressure_work_plane_ptr(which, local_z), True)
        return self.plane(E.PRESSURES_1 if which == 0 else E.PRESSURES_2, local_z)

    def upload(self, image_id: int, array: np.ndarray):
        self.engine.upload_image(image_id, array)

    def download(self, image_id: int) -> np.ndarray:
        return self.engine.download_image(image_id)

    def clear_pressures(self):
        self.engine.run_section("12a_clear_pressures_1")
        self.engine.run_section("12b_clear_pressures_2")

    # the loop section in explicit form (include/fluid_engine.h: fluid_pressure_loop_*)
    def loop_begin(self):
        if self.fast:
            self.engine.pressure_loop_begin()

    def loop_sweep(self, k: int):
        if self.fast:
            self.engine.pressure_loop_sweep(k)
        else:
            self.engine.run_pressure_dispatch(1 if k % 2 == 0 else 0)

    def loop_end(self, iterations: int):
        if self.fast:
            self.engine.pressure_loop_end(iterations)

    def sync(self):
        self.torch.cuda.synchronize(self.device)

    def halo_written(self, image_id: int):
        self.engine.notify_image_written(image_id)

    def close(self):
        self.engine.close()


class HostSlabCompute:
    """CPU stand-in with the same interface, for the multi-process tests: numpy arrays with ghost
    planes, the sweep supplied by the caller (the tests pass the CPU oracle).  Not a product path."""

    def __init__(self, params: FluidParams, slab: Tuple[int, int], sweep_fn):
        import torch

        self.torch = torch
        w, h, _ = params.size
        self.params = params
        self.z0, self.dl = slab
        self.sweep_fn = sweep_fn
        shape = (self.dl + 2, h, w)
        self.arr = {
            E.CELL_TYPES: torch.zeros(shape, dtype=torch.uint8),
            E.DIVERGENCES: torch.zeros(shape, dtype=torch.float32),
            E.PRESSURES_1: torch.zeros(shape, dtype=torch.float32),
            E.PRESSURES_2: torch.zeros(shape, dtype=torch.float32),
        }

    def plane(self, image_id: int, local_z: int):
        return self.arr[image_id][local_z + 1].view(-1)

    def upload(self, image_id: int, array: np.ndarray):
        self.arr[image_id][1:-1] = self.torch.from_numpy(np.ascontiguousarray(array))

    def download(self, image_id: int) -> np.ndarray:
        return self.arr[image_id][1:-1].numpy().copy()

    def work_plane(self, which: int, local_z: int):
        return self.plane(E.PRESSURES_1 if which == 0 else E.PRESSURES_2, local_z)

    def clear_pressures(self):
        self.arr[E.PRESSURES_1][1:-1] = float(self.params.pressure_air)
        self.arr[E.PRESSURES_2][1:-1] = float(self.params.pressure_air)

    def loop_begin(self):
        pass

    def loop_end(self, iterations: int):
        pass

    def loop_sweep(self, k: int):
        is_even_iteration = 1 if k % 2 == 0 else 0
        src = E.PRESSURES_1 if is_even_iteration == 1 else E.PRESSURES_2
        dst = E.PRESSURES_2 if is_even_iteration == 1 else E.PRESSURES_1
        out = self.arr[dst].numpy()
        ghosts = out[0].copy(), out[-1].copy()
        # one sweep over the slab INCLUDING its ghost planes as if they were cells, then put the
        # ghost planes of the output back: owned planes only depend on z±1, so they are exact.
        self.sweep_fn(self.params, self.arr[E.CELL_TYPES].numpy(), self.arr[E.DIVERGENCES].numpy(),
                      self.arr[src].numpy(), out)
        out[0], out[-1] = ghosts

    def sync(self):
        pass

    def halo_written(self, image_id: int):
        pass

    def close(self):
        pass


# ---- the solver ------------------------------------------------------------------------------------------
class SlabPressureSolver:
    def __init__(self, size, iterations: int, ctx: DistContext, compute, slab: Tuple[int, int],
                 transport: str = "direct"):
        # transport "direct": the communicator addresses the planes where they live (RCCL on
        # device memory, gloo on host memory).  "staged": bounce through host tensors — only for
        # rehearsing the GPU slab path over gloo on a box with a single GPU (tests).
        self.transport = transport
        self.size = tuple(size)
        self.iterations = iterations
        self.ctx = ctx
        self.compute = compute
        self.z_begin, self.z_count = slab
        self.lo = ctx.rank - 1 if ctx.rank > 0 else None
        self.hi = ctx.rank + 1 if ctx.rank < ctx.world - 1 else None
        self._plans = {}
        self._ops = {}

    @classmethod
    def create_gpu(cls, size, iterations: int, ctx: DistContext, pressure_kernel: int = 0,
                   seed: Optional[int] = None, params: Optional[FluidParams] = None):
        """Full-fluid benchmark scene (scenes.py) on this rank's slab of the global grid."""
        from . import scenes

        w, h, d = size
        params = params or default_params(w, h, d, 0)
        slab = partition_z(d, ctx.world)[ctx.rank]
        comp = GpuSlabCompute(params, slab, ctx.device, pressure_kernel)
        self = cls(size, iterations, ctx, comp, slab)
        shape = (slab[1], h, w)
        comp.upload(E.CELL_TYPES, scenes.full_fluid_types(shape, slab[0], d))
        comp.upload(E.DIVERGENCES, scenes.full_fluid_divergence(
            shape, scenes.SEED_JACOBI if seed is None else seed, slab[0]))
        self.exchange(E.CELL_TYPES)
        return self

    # -- halo exchange ---------------------------------------------------------------------------------
    def _run_plan(self, key, make_plane):
        """Send the first/last owned plane to the lower/upper neighbour, receive their last/first
        owned plane into the ghost planes.  Grouped point-to-point, both directions at once.  The
        tensor views and P2POps are built once per buffer and reused (the planes never move)."""
        import torch.distributed as dist

        if self.ctx.world == 1:
            return
        plan = self._plans.get(key)
        if plan is None:
            plan = []  # (is_send, plane tensor, peer)
            if self.lo is not None:
                plan.append((True, make_plane(0), self.lo))
                plan.append((False, make_plane(-1), self.lo))
            if self.hi is not None:
                plan.append((True, make_plane(self.z_count - 1), self.hi))
                plan.append((False, make_plane(self.z_count), self.hi))
            self._plans[key] = plan
        if self.transport == "staged":
            staged = [(snd, t.cpu() if snd else t.new_empty(t.shape, device="cpu"), t, peer)
                      for snd, t, peer in plan]
            ops = [dist.P2POp(dist.isend if snd else dist.irecv, h, peer)
                   for snd, h, _, peer in staged]
            for work in dist.batch_isend_irecv(ops):
                work.wait()
            for snd, h, t, _ in staged:
                if not snd:
                    t.copy_(h)
            return
        ops = self._ops.get(key)
        if ops is None:
            ops = [dist.P2POp(dist.isend if snd else dist.irecv, t, peer) for snd, t, peer in plan]
            self._ops[key] = ops
        for work in dist.batch_isend_irecv(ops):
            work.wait()

    def exchange(self, image_id: int):
        """Halo exchange of an image (cell types at set-up)."""
        self._run_plan(("img", image_id), lambda z: self.compute.plane(image_id, z))
        self.compute.halo_written(image_id)  # derived data (the neighbour mask) is rebuilt

    def exchange_work(self, which: int):
        """Halo exchange of the buffer holding the iterates of parity `which`."""
        self._run_plan(("work", which), lambda z: self.compute.work_plane(which, z))

    # -- the loop section ---------------------------------------------------------------------------------
    def clear_pressures(self):
        self.compute.clear_pressures()

    def solve(self, iterations: Optional[int] = None):
        """FlowLoopPushConstantSection semantics: dispatch k maps iterate k (parity
        k%2; PRESSURES_1 holds the even ones) to iterate k+1.  After every dispatch the boundary
        planes of the buffer just written are exchanged so the next dispatch sees the neighbours'
        new values."""
        n = self.iterations if iterations is None else iterations
        c = self.compute
        c.loop_begin()
        self.exchange_work(0)           # iterate 0
        for k in range(n):
            c.loop_sweep(k)
            self.exchange_work((k + 1) % 2)
        c.loop_end(n)

    def step(self):
        self.clear_pressures()
        self.solve()

    # -- measurement -----------------------------------------------------------------------------------------
    def benchmark(self, steps: int, warmup: int) -> dict:
        import torch
        import torch.distributed as dist

        for _ in range(warmup):
            self.step()
        eng = getattr(self.compute, "engine", None)
        self.compute.sync()
        dist.barrier()
        self.compute.sync()
        if eng is not None:
            eng.enable_timing(True)
            eng.reset_timing()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        self.compute.sync()
        dist.barrier()
        self.compute.sync()
        wall = time.perf_counter() - t0
        t = torch.tensor([wall], dtype=torch.float64, device=self.ctx.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        out = {"wall_s": float(t.item()),
               "local_cells": self.size[0] * self.size[1] * self.z_count}
        if eng is not None:
            ms, calls = eng.section_time_ms("12_solve_pressure")
            eng.enable_timing(False)
            k = torch.tensor([ms / max(calls, 1)], dtype=torch.float64, device=self.ctx.device)
            dist.all_reduce(k, op=dist.ReduceOp.MAX)
            out["kernel_ms_per_sweep"] = float(k.item())
            out["exchange_ms_per_sweep"] = max(
                0.0, 1e3 * out["wall_s"] / (steps * self.iterations) - out["kernel_ms_per_sweep"])
        return out

    def gather_pressures(self):
        """Rank 0 gets the global PRESSURES_1 / PRESSURES_2 arrays (tests)."""
        import torch
        import torch.distributed as dist

        res = []
        for img in (E.PRESSURES_1, E.PRESSURES_2):
            local = torch.from_numpy(self.compute.download(img))
            parts = [None] * self.ctx.world if self.ctx.rank == 0 else None
            dist.gather_object(local.numpy(), parts, dst=0)
            res.append(np.concatenate(parts, axis=0) if self.ctx.rank == 0 else None)
        return res

    def close(self):
        self.compute.close()
