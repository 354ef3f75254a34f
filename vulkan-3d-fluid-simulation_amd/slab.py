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

    def __init__(self, params: FluidParams, slab: Tuple[int, int], device, pressure_kernel: int = 0,
                 particle_capacity: int = 0, pressure_iterations: int = 200, edge_stream: bool = False):
        import torch

        self.torch = torch
        self.device = device
        nbytes = E.FluidEngine.required_arena_bytes(params, particle_capacity, slab=slab)
        if nbytes == 0:
            raise RuntimeError("invalid slab geometry")
        # One explicit side stream, made torch's current stream for this process: the engine's
        # kernels and the communicator's stream-ordering both follow it.  (The legacy null stream
        # has handle 0, which the C ABI reads as "create your own".)
        self.stream = torch.cuda.Stream(device=device)
        self.comm_stream = None  # created by the first overlapped exchange
        torch.cuda.set_stream(self.stream)
        self.arena = torch.zeros(nbytes + 256, dtype=torch.uint8, device=device)
        base = self.arena.data_ptr()
        self._pad = (-base) % 256
        assert self.stream.cuda_stream != 0
        self.engine = E.FluidEngine(
            params, particle_capacity=particle_capacity, pressure_iterations=pressure_iterations,
            device=device.index if device.index is not None else -1,
            slab=slab, stream=self.stream.cuda_stream,
            arena=base + self._pad, arena_bytes=nbytes)
        self.engine.set_option(E.OPT_PRESSURE_KERNEL, pressure_kernel)
        # FLUID_OPT_EDGE_STREAM: the EDGES launches of split passes run on a second engine stream beside
        # the INTERIOR launches; exchanges are then ordered against that stream (comm_scope / comm_join)
        self.edge_ext = None
        if edge_stream or os.environ.get("FLUID_SLAB_EDGE_STREAM") == "1":
            self.engine.set_option(E.OPT_EDGE_STREAM, 1)
            self.edge_ext = torch.cuda.ExternalStream(self.engine.pressure_loop_edge_stream(),
                                                      device=device)
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

    def upload(self, image_id: int, array: np.ndarray):
        self.engine.upload_image(image_id, array)

    def download(self, image_id: int) -> np.ndarray:
        return self.engine.download_image(image_id)

    def clear_pressures(self):
        self.engine.run_section("12a_clear_pressures_1")
        self.engine.run_section("12b_clear_pressures_2")

    # ---- the loop section in explicit form (include/fluid_engine.h: fluid_pressure_loop_*) ----
    # A "loop buffer" is anything the slabs must exchange boundary planes of.  planes(buf, first, n)
    # returns n consecutive local planes starting at `first` as one flat tensor (contiguous memory).
    def max_halo(self) -> int:
        """Deepest halo this backend can use (planes per exchange = sweeps between exchanges)."""
        return self.engine.LOOP_MAX_HALO if self.fast else 1

    def loop_begin(self, halo: int):
        """Returns [(buffer, planes)] to exchange once before the first advance; the newest
        iterate is then in buffer 0 with `halo` valid ghost planes (after loop_halo_exchanged)."""
        if self.fast:
            self.engine.pressure_loop_begin()
            aux = max(halo - 1, 1)
            return [(self.engine.LOOP_MASK, aux), (self.engine.LOOP_RHS, aux), (0, halo)]
        return [(0, 1), (1, 1)]  # the two pressure images themselves

    def loop_halo_exchanged(self, halo: int, first: bool):
        if self.fast:
            self.engine.pressure_loop_halo_exchanged(halo, max(halo - 1, 1) if first else 0)

    def loop_max_sweeps(self) -> int:
        return self.engine.pressure_loop_max_sweeps() if self.fast else 1

    def loop_advance(self, k: int, sweeps: int, keep_mid: bool, part: Optional[str] = None,
                     interior: Optional[Tuple[int, int]] = None) -> int:
        """Sweeps k .. k+sweeps-1.  Returns the buffer that now holds the newest iterate.
        `part` = "edges" / "interior": one of the two launches of a split two-sweep pass; the output
        planes [interior[0], interior[1]) are the interior (include/fluid_engine.h)."""
        if self.fast:
            if part is not None:
                assert sweeps == 2
                which = (self.engine.LOOP_PART_EDGES if part == "edges"
                         else self.engine.LOOP_PART_INTERIOR)
                return self.engine.pressure_loop_advance_part(keep_mid, which, *interior)
            return self.engine.pressure_loop_advance(sweeps, keep_mid)
        assert sweeps == 1 and part is None
        self.engine.run_pressure_dispatch(1 if k % 2 == 0 else 0)
        return (k + 1) % 2

    # ---- a halo exchange that runs beside compute: a second stream, ordered by events ----
    can_overlap = True

    def comm_scope(self):
        """Context in which the exchange is issued: the communication stream, which first waits for
        everything launched on the compute stream so far."""
        torch = self.torch
        if self.comm_stream is None:
            self.comm_stream = torch.cuda.Stream(device=self.device)
        ev = torch.cuda.Event()
        ev.record(self.stream)
        self.comm_stream.wait_event(ev)
        if self.edge_ext is not None:  # ... and on the edge stream (the planes to send come from there)
            ev2 = torch.cuda.Event()
            ev2.record(self.edge_ext)
            self.comm_stream.wait_event(ev2)
        return torch.cuda.stream(self.comm_stream)

    def comm_mark(self):
        """Call inside comm_scope() after the exchange has been issued (and stream-waited for)."""
        ev = self.torch.cuda.Event()
        ev.record(self.comm_stream)
        return ev

    def comm_join(self, mark):
        """Launches on the compute stream (and the edge stream) from here on come after the exchange."""
        self.stream.wait_event(mark)
        if self.edge_ext is not None:
            self.edge_ext.wait_event(mark)

    def loop_end(self):
        if self.fast:
            self.engine.pressure_loop_end()

    def planes(self, buf: int, first: int, count: int):
        if self.fast:
            ptr, nbytes = self.engine.pressure_loop_plane_ptr(buf, first)
            return self._view(ptr, nbytes * count, buf != self.engine.LOOP_MASK)
        ptr, nbytes = self.engine.image_plane_ptr(E.PRESSURES_1 if buf == 0 else E.PRESSURES_2, first)
        return self._view(ptr, nbytes * count, True)

    def sync(self):
        self.torch.cuda.synchronize(self.device)

    def halo_written(self, image_id: int):
        self.engine.notify_ghost_planes_written(image_id)

    def run_section_group(self, first: str, count: int):
        self.engine.run_section_group(first, count)

    def set_diffuse_mode(self, mode: int):
        self.engine.set_diffuse_mode(mode)

    # ---- full step on slabs -----------------------------------------------------------------------
    IMAGE_GHOST = E.FluidEngine.IMAGE_GHOST_PLANES

    def run_section(self, name: str):
        self.engine.run_section(name)

    def image_planes(self, image_id: int, first: int, count: int):
        """`count` consecutive local planes of an image as one flat tensor."""
        ptr, nbytes = self.engine.image_plane_ptr(image_id, first)
        dtype, _ = E.IMAGE_DTYPES[image_id]
        return self._view(ptr, nbytes * count, dtype == np.float32)

    def collect_leavers(self):
        """Particles that left this slab in 14_particles: (uint8 tensor of 32-byte entries, count)."""
        ptr, n = self.engine.particles_collect_leavers()
        if n == 0:
            return self.arena[:0], 0
        return self._view(ptr, n * self.engine.LEAVER_BYTES, False), n

    def adopt(self, entries, count: int):
        if count:
            self.engine.particles_adopt(entries.data_ptr(), count)

    def halo_violation(self) -> bool:
        return self.engine.slab_halo_violation()

    def download_particles(self) -> np.ndarray:
        return self.engine.download_particles()

    def upload_particles(self, particles: np.ndarray):
        self.engine.upload_particles(particles)

    def close(self):
        self.engine.close()


class HostSlabCompute:
    """CPU stand-in with the same interface, for the multi-process tests: numpy arrays with GW ghost
    planes per side, the sweep supplied by the caller (the tests pass the CPU oracle).  Not a product
    path.  Loop buffers: 0..2 working pressures, 3 cell types, 4 divergence."""

    GW = 8
    TYPES, DIV = 3, 4

    def __init__(self, params: FluidParams, slab: Tuple[int, int], sweep_fn, max_sweeps: int = 2):
        import torch

        self.torch = torch
        w, h, _ = params.size
        self.params = params
        self.z0, self.dl = slab
        self.sweep_fn = sweep_fn
        self.max_sweeps = max_sweeps
        shape = (self.dl + 2 * self.GW, h, w)
        self.arr = {
            E.CELL_TYPES: torch.zeros(shape, dtype=torch.uint8),
            E.DIVERGENCES: torch.zeros(shape, dtype=torch.float32),
            E.PRESSURES_1: torch.zeros(shape, dtype=torch.float32),
            E.PRESSURES_2: torch.zeros(shape, dtype=torch.float32),
        }
        self.work = [torch.zeros(shape, dtype=torch.float32) for _ in range(3)]
        self.cur, self.prev, self.k = 0, -1, 0

    def _owned(self, t):
        return t[self.GW:self.GW + self.dl]

    def plane(self, image_id: int, local_z: int):
        return self.arr[image_id][local_z + self.GW].view(-1)

    def upload(self, image_id: int, array: np.ndarray):
        self._owned(self.arr[image_id])[...] = self.torch.from_numpy(np.ascontiguousarray(array))

    def download(self, image_id: int) -> np.ndarray:
        return self._owned(self.arr[image_id]).numpy().copy()

    def clear_pressures(self):
        self._owned(self.arr[E.PRESSURES_1])[...] = float(self.params.pressure_air)
        self._owned(self.arr[E.PRESSURES_2])[...] = float(self.params.pressure_air)

    def max_halo(self) -> int:
        return self.GW

    def loop_begin(self, halo: int):
        self._owned(self.work[0])[...] = self._owned(self.arr[E.PRESSURES_1])
        self.cur, self.prev, self.k = 0, -1, 0
        return [(self.TYPES, halo), (self.DIV, max(halo - 1, 1)), (0, halo)]

    def loop_halo_exchanged(self, halo: int, first: bool):
        pass

    def loop_max_sweeps(self) -> int:
        return self.max_sweeps

    def _other(self, a, b):
        return next(i for i in range(3) if i not in (a, b))

    def _sweep(self, src, dst):
        # one sweep over the slab INCLUDING its ghost planes as if they were cells: with valid data g
        # planes deep in the ghost region the result is exact g-1 planes deep (and on all owned planes)
        self.sweep_fn(self.params, self.arr[E.CELL_TYPES].numpy(), self.arr[E.DIVERGENCES].numpy(),
                      self.work[src].numpy(), self.work[dst].numpy())

    can_overlap = True

    def comm_scope(self):
        import contextlib

        return contextlib.nullcontext()

    def comm_mark(self):
        return None

    def comm_join(self, mark):
        pass

    def _split_pass(self, keep_mid: bool, part: str, interior):
        """One launch of a split pass: two sweeps from `cur` into scratch copies, of which only this
        part's planes are stored — so a part that (wrongly) depended on ghost planes still in flight
        would show up as a mismatch."""
        dst = self._other(self.cur, self.cur)
        mid = self._other(self.cur, dst)
        t_mid, t_dst = self.work[mid].clone(), self.work[dst].clone()
        self.sweep_fn(self.params, self.arr[E.CELL_TYPES].numpy(), self.arr[E.DIVERGENCES].numpy(),
                      self.work[self.cur].numpy(), t_mid.numpy())
        self.sweep_fn(self.params, self.arr[E.CELL_TYPES].numpy(), self.arr[E.DIVERGENCES].numpy(),
                      t_mid.numpy(), t_dst.numpy())
        n = self.dl + 2 * self.GW
        a = min(max(interior[0] + self.GW, 0), n)
        b = min(max(interior[1] + self.GW, a), n)
        ranges = [(a, b)] if part == "interior" else [(0, a), (b, n)]
        for lo, hi in ranges:
            self.work[dst][lo:hi] = t_dst[lo:hi]
            self.work[mid][lo:hi] = t_mid[lo:hi]
        if self._part_done is None:
            self._part_done = part
            return dst
        assert self._part_done != part
        self._part_done = None
        self.prev = mid if keep_mid else -1
        self.cur = dst
        self.k += 2
        return dst

    _part_done = None

    def loop_advance(self, k: int, sweeps: int, keep_mid: bool, part=None, interior=None):
        assert k == self.k
        if part is not None:
            assert sweeps == 2
            return self._split_pass(keep_mid, part, interior)
        assert self._part_done is None
        if sweeps == 2:
            dst = self._other(self.cur, self.cur)
            mid = self._other(self.cur, dst)
            self._sweep(self.cur, mid)
            self._sweep(mid, dst)
            self.prev = mid if keep_mid else -1
            self.cur = dst
        else:
            dst = self._other(self.cur, self.prev if self.prev >= 0 else self.cur)
            self._sweep(self.cur, dst)
            self.prev, self.cur = self.cur, dst
        self.k += sweeps
        return self.cur

    def loop_end(self):
        if self.k == 0:
            return
        water = self._owned(self.arr[E.CELL_TYPES]) == int(self.params.cell_type_water)
        even, odd = (self.cur, self.prev) if self.k % 2 == 0 else (self.prev, self.cur)
        for img, buf in ((E.PRESSURES_1, even), (E.PRESSURES_2, odd)):
            if buf >= 0:
                dst = self._owned(self.arr[img])
                dst[water] = self._owned(self.work[buf])[water]

    def planes(self, buf: int, first: int, count: int):
        t = self.work[buf] if buf < 3 else self.arr[E.CELL_TYPES if buf == self.TYPES
                                                     else E.DIVERGENCES]
        return t[first + self.GW:first + self.GW + count].view(-1)

    def sync(self):
        pass

    def halo_written(self, image_id: int):
        pass

    def close(self):
        pass


# ---- the solver ------------------------------------------------------------------------------------------
class SlabPressureSolver:
    def __init__(self, size, iterations: int, ctx: DistContext, compute, slab: Tuple[int, int],
                 transport: str = "direct", halo_depth: int = 8):
        # halo_depth h: the slabs exchange h boundary planes at a time and then run h sweeps without
        # communication, recomputing the shrinking ghost region (same bytes on the wire as one plane
        # per sweep, h times fewer messages and host round trips).  Clipped to what the compute
        # backend and the slab thickness allow; even, so sweeps can go in pairs.
        self.halo_depth = halo_depth
        # run halo exchanges beside the passes that do not need them (solve()): True = split the pass
        # before and the pass after each exchange, "before" = only the pass before it, False = exchanges
        # in line.  FLUID_SLAB_OVERLAP=0 / before / 1 forces one; bench.py --gpus N measures all three.
        env = os.environ.get("FLUID_SLAB_OVERLAP", "1")
        self.overlap = False if env == "0" else ("before" if env == "before" else True)
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
        self.exchanges = 0  # halo exchanges performed (diagnostics)
        self.overlapped = 0  # ... of which started beside a split pass

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
    def _run_plan(self, key, make_planes, width: int):
        """Send the first/last `width` owned planes to the lower/upper neighbour, receive their
        last/first owned planes into the ghost planes.  Grouped point-to-point, both directions at
        once.  Tensor views and P2POps are built once per buffer and reused (planes never move)."""
        self._finish_plan(self._start_plan(key, make_planes, width, overlapped=False))

    def _start_plan(self, key, make_planes, width: int, overlapped: bool):
        """Issue the exchange; `_finish_plan(handle)` before anything may touch the received planes.
        overlapped: issue it on the compute backend's communication stream (ordered after what has
        been launched so far), so that launches made between start and finish run beside it."""
        import torch.distributed as dist

        if self.ctx.world == 1 and self.transport != "loopback":
            return None
        self.exchanges += 1
        if self.z_count < width:
            raise RuntimeError(f"slab of {self.z_count} planes is thinner than the halo ({width})")
        plan = self._plans.get(key)
        if plan is None:
            plan = []  # (is_send, flat tensor of `width` planes, peer)
            n = self.z_count
            if self.lo is not None:
                plan.append((True, make_planes(0, width), self.lo))
                plan.append((False, make_planes(-width, width), self.lo))
            if self.hi is not None:
                plan.append((True, make_planes(n - width, width), self.hi))
                plan.append((False, make_planes(n, width), self.hi))
            self._plans[key] = plan
        scope = self.compute.comm_scope() if overlapped else None
        if scope is not None:
            scope.__enter__()
        try:
            if self.transport == "staged":
                # through host tensors, synchronously (t.cpu() waits for the current stream, which inside
                # the scope is the communication stream, ordered after the launches that produce the
                # planes): rehearses the schedule of the GPU path over gloo on a box with one GPU
                staged = [(snd, t.cpu() if snd else t.new_empty(t.shape, device="cpu"), t, peer)
                          for snd, t, peer in plan]
                ops = [dist.P2POp(dist.isend if snd else dist.irecv, h, peer)
                       for snd, h, _, peer in staged]
                for work in dist.batch_isend_irecv(ops):
                    work.wait()
                for snd, h, t, _ in staged:
                    if not snd:
                        t.copy_(h)
                if not overlapped:
                    return None
                return ("overlapped", [], self.compute.comm_mark())
            if self.transport == "loopback":
                # single-process rehearsal of one rank's work (tools/slab_rank_sim.py): every receive
                # is filled by a device copy of a send buffer of the same size; no communicator
                sends = [t for snd, t, _ in plan if snd]
                recvs = [t for snd, t, _ in plan if not snd]
                for i, r in enumerate(recvs):
                    r.copy_(sends[(i + 1) % len(sends)])
                works = []
            else:
                ops = self._ops.get(key)
                if ops is None:
                    ops = [dist.P2POp(dist.isend if snd else dist.irecv, t, peer)
                           for snd, t, peer in plan]
                    self._ops[key] = ops
                works = dist.batch_isend_irecv(ops)
            if not overlapped:
                return ("works", works)
            if self.ctx.backend == "nccl":
                # stream-level wait (the host does not block): the communication stream now ends
                # with the exchange, and the mark below is what the compute stream joins on
                for work in works:
                    work.wait()
                works = []
            return ("overlapped", works, self.compute.comm_mark())
        finally:
            if scope is not None:
                scope.__exit__(None, None, None)

    def _finish_plan(self, handle):
        if handle is None:
            return
        for work in handle[1]:
            work.wait()
        if handle[0] == "overlapped":
            self.compute.comm_join(handle[2])

    def exchange(self, image_id: int):
        """One-plane halo exchange of an image (cell types at set-up)."""
        self._run_plan(("img", image_id), lambda z, n: self.compute.plane(image_id, z), 1)
        self.compute.halo_written(image_id)  # derived data (the neighbour mask) is rebuilt

    def exchange_loop_buffer(self, buf: int, width: int):
        """Halo exchange of a buffer of the running loop, `width` planes deep."""
        self._run_plan(("loop", buf, width), lambda z, n: self.compute.planes(buf, z, n), width)

    # -- the loop section ---------------------------------------------------------------------------------
    def clear_pressures(self):
        self.compute.clear_pressures()

    def effective_halo(self) -> int:
        # every rank must come to the same depth: limit by the thinnest slab of the partition
        thinnest = min(n for _, n in partition_z(self.size[2], self.ctx.world))
        h = min(self.halo_depth, self.compute.max_halo(), thinnest)
        if self.compute.loop_max_sweeps() >= 2 and h >= 2:
            h -= h % 2
        return max(h, 1)

    def solve(self, iterations: Optional[int] = None):
        """FlowLoopPushConstantSection semantics: dispatch k maps iterate k to iterate
        k+1; after N dispatches PRESSURES_1 holds the last even iterate, PRESSURES_2 the last odd one.
        Every sweep consumes one valid ghost plane per side; when fewer are left than the next
        launch needs (2 for a two-sweeps-per-pass launch), `h` boundary planes of the newest iterate
        are exchanged with the two Z-neighbours.

        Overlap (self.overlap, h >= 4, slabs thicker than 2h): the pass before an exchange is split —
        the h planes per face that will be sent are computed first, the exchange starts on the
        communication stream, the planes in between follow —, and so is the pass after it: the
        output planes that depend on owned planes only are computed while the exchange is in flight,
        the rest once it has landed.  Same arithmetic, same iterates."""
        n = self.iterations if iterations is None else iterations
        c = self.compute
        h = self.effective_halo()
        for buf, width in c.loop_begin(h):
            self.exchange_loop_buffer(buf, width)
        c.loop_halo_exchanged(h, True)
        valid = h          # valid ghost planes of the newest iterate
        cur = 0            # buffer holding it
        pair = c.loop_max_sweeps() >= 2 and h >= 2
        thinnest = min(m for _, m in partition_z(self.size[2], self.ctx.world))
        split = (self.overlap and pair and h >= 4 and thinnest > 2 * h
                 and getattr(c, "can_overlap", False)
                 and (self.ctx.world > 1 or self.transport == "loopback"))
        big = 1 << 30
        dl = self.z_count
        # interior of the pass before an exchange / of the pass after it (local output planes)
        before = (h if self.lo is not None else -big, dl - h if self.hi is not None else big)
        after = (2 if self.lo is not None else -big, dl - 2 if self.hi is not None else big)
        pending = None     # exchange in flight: finish before launching anything that reads ghosts
        k = 0
        while k < n:
            sweeps = 2 if (pair and n - k >= 2) else 1
            keep = sweeps == 2 and n - k == 2
            if pending is None and valid < sweeps:
                self.exchange_loop_buffer(cur, h)
                c.loop_halo_exchanged(h, False)
                valid = h
            left = n - k - sweeps
            next_sweeps = 2 if (pair and left >= 2) else min(left, 1)
            valid_after = valid - 2 if sweeps == 2 else 0
            if pending is not None and self.overlap == "before":
                # half the overlap: only the pass before the exchange was split; wait, then a whole pass
                self._finish_plan(pending)
                pending = None
                cur = c.loop_advance(k, sweeps, keep)
            elif pending is not None:
                # first pass after the exchange started: valid == h here (reported at the start)
                c.loop_advance(k, 2, keep, "interior", after)
                self._finish_plan(pending)
                pending = None
                cur = c.loop_advance(k, 2, keep, "edges", after)
            elif split and sweeps == 2 and next_sweeps == 2 and valid_after < 2:
                dst = c.loop_advance(k, 2, keep, "edges", before)
                self.overlapped += 1
                pending = self._start_plan(("loop", dst, h),
                                           lambda z, m, b=dst: c.planes(b, z, m), h, overlapped=True)
                cur = c.loop_advance(k, 2, keep, "interior", before)
                assert cur == dst
                c.loop_halo_exchanged(h, False)  # started; the next pass orders itself behind it
                valid_after = h
            else:
                cur = c.loop_advance(k, sweeps, keep)
            valid = valid_after
            k += sweeps
        assert pending is None
        c.loop_end()

    def step(self):
        self.clear_pressures()
        self.solve()

    # -- measurement -----------------------------------------------------------------------------------------
    def benchmark(self, steps: int, warmup: int) -> dict:
        import torch
        import torch.distributed as dist

        probe = self._probe_overlap()
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
               "local_cells": self.size[0] * self.size[1] * self.z_count,
               "halo_overlap": probe}
        if eng is not None:
            ms, calls = eng.section_time_ms("12_solve_pressure")
            eng.enable_timing(False)
            k = torch.tensor([ms / max(calls, 1)], dtype=torch.float64, device=self.ctx.device)
            dist.all_reduce(k, op=dist.ReduceOp.MAX)
            out["kernel_ms_per_sweep"] = float(k.item())
            out["exchange_ms_per_sweep"] = max(
                0.0, 1e3 * out["wall_s"] / (steps * self.iterations) - out["kernel_ms_per_sweep"])
        return out

    def _probe_overlap(self) -> dict:
        """Untimed, before the warm-up: one step of the loop with the exchanges issued in line and one
        with the split-pass overlap (solve()), each after a step of its own to set up streams and
        communicators; every rank adopts the faster schedule (MAX over ranks).  Whether hiding an
        8-MiB exchange is worth two extra launches per exchange depends on the link, so it is measured
        where it runs."""
        import torch
        import torch.distributed as dist

        if not self.overlap or self.ctx.world == 1 or "FLUID_SLAB_OVERLAP" in os.environ:
            return {"used": self.overlap if self.ctx.world > 1 else False, "probed": False}
        times = {}
        for mode in (False, "before", True):
            self.overlap = mode
            self.step()
            self.compute.sync()
            dist.barrier()
            t0 = time.perf_counter()
            self.step()
            self.compute.sync()
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=self.ctx.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            times[mode] = float(t.item())
        self.overlap = min(times, key=times.get)  # the same on every rank: the times are the all-reduced MAX
        return {"used": self.overlap if self.overlap else False, "probed": True,
                "step_ms_inline": 1e3 * times[False], "step_ms_overlap_before": 1e3 * times["before"],
                "step_ms_overlapped": 1e3 * times[True]}

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


# ---- the whole simulation step on Z slabs ---------------------------------------------------------------
class SlabSimulation:
    """SimulationInitializationSections / SimulationStepSections (fluid_flow_sections.h:136-338) with
    the grid cut into Z slabs, one rank per slab: the reference's section order, plus the ghost-plane
    exchanges the stencils and the velocity sampler need and the hand-over of particles that cross a
    slab face (include/fluid_engine.h, "Z-slab contexts, full step").  Results equal the single-GPU
    (and the oracle's) step bit for bit as long as no sample reaches further than the ghost planes
    (checked every step)."""

    def __init__(self, params: FluidParams, particle_capacity: int, iterations: int, ctx: DistContext,
                 compute=None, transport: str = "direct", halo_depth: int = 8, grouped: bool = True,
                 overlap: Optional[bool] = None, diffuse_mode: int = E.DIFFUSE_REFERENCE_EXACT):
        # grouped: 04+05, 07+08 and 09+10+11 as single passes (include/fluid_engine.h:
        # fluid_run_section_group); 09+10+11 needs fluid_size.x % 4 == 0
        self.grouped = grouped
        # E.DIFFUSE_INTENDED: 09 is the 7-point diffusion the shader meant, which needs a
        # ghost plane of VELOCITIES_2 per side and runs 09, 10, 11 one by one
        self.diffuse_mode = diffuse_mode
        self.params = params
        self.ctx = ctx
        self.size = params.size
        self.capacity = particle_capacity
        slab = partition_z(self.size[2], ctx.world)[ctx.rank]
        self.slab = slab
        self.compute = compute or GpuSlabCompute(params, slab, ctx.device,
                                                 particle_capacity=particle_capacity,
                                                 pressure_iterations=iterations)
        if diffuse_mode != E.DIFFUSE_REFERENCE_EXACT:
            self.compute.set_diffuse_mode(diffuse_mode)
        self.pressure = SlabPressureSolver(self.size, iterations, ctx, self.compute, slab,
                                           transport=transport, halo_depth=halo_depth)
        if overlap is not None:  # else SlabPressureSolver's default (FLUID_SLAB_OVERLAP)
            self.pressure.overlap = overlap
        self.transport = transport
        self.ghost = min(self.compute.IMAGE_GHOST, min(n for _, n in partition_z(self.size[2],
                                                                                  ctx.world)))
        self.migrated = 0  # particles handed over so far (diagnostics)

    # -- halo exchange of an image, `width` planes deep (reuses the solver's plan machinery)
    def exchange_image(self, image_id: int, width: int):
        self.pressure._run_plan(("image", image_id, width),
                                lambda z, n: self.compute.image_planes(image_id, z, n), width)
        self.compute.halo_written(image_id)

    def run_init(self):
        for name in ("init_clear_velocities_1", "init_clear_cell_types", "00_init_particles"):
            self.compute.run_section(name)
        # cleared images are uniform, but their value need not be the ghost planes' zero
        self.exchange_image(E.CELL_TYPES, 1)
        self.exchange_image(E.VELOCITIES_1, self.ghost)

    def run_step(self):
        c, x = self.compute, self.exchange_image
        c.run_section("01a_clear_particle_densities")
        c.run_section("01_update_densities")          # owned particles only, into owned planes
        c.run_section("02_update_water")
        x(E.NEW_CELL_TYPES, 1)                         # 03 looks at z-1 / z+1
        c.run_section("03_update_air")
        x(E.NEW_CELL_TYPES, 1)                         # 05 reads the final new types at z-1
        if self.grouped:                               # old types / V1 at z+-1: still current
            c.run_section_group("04_compute_extrapolated_velocities", 2)
        else:
            c.run_section("04_compute_extrapolated_velocities")
            c.run_section("05_set_extrapolated_velocities")
        x(E.VELOCITIES_1, self.ghost)                  # 07 samples V1 around each cell
        c.run_section("06_update_cell_types")          # carries one ghost plane per side along
        if self.grouped:
            c.run_section_group("07_advect", 2)
        else:
            c.run_section("07_advect")
            c.run_section("08_forces")
        intended = self.diffuse_mode == E.DIFFUSE_INTENDED
        if self.grouped and self.size[0] % 4 == 0 and not intended:
            x(E.VELOCITIES_2, 1)                       # 11, on what 10 makes of V2 at z+1
            c.run_section_group("09_diffuse", 3)
        else:
            if intended:
                x(E.VELOCITIES_2, 1)                   # the diffusion stencil reads V2 at z-1, z+1
            c.run_section("09_diffuse")
            c.run_section("10_solids")
            x(E.VELOCITIES_1, 1)                       # 11 reads V1 at z+1
            c.run_section("11_compute_divergence")
        self.pressure.step()                           # 12a, 12b, the 12_solve_pressure loop
        x(E.PRESSURES_2, 1)                            # 13 reads P2 at z-1
        c.run_section("13_fix_divergence")
        x(E.VELOCITIES_1, self.ghost)                  # 14 samples V1; 04 of the next step reads z+-1
        c.run_section("14_particles")
        self.migrate_particles()
        if c.halo_violation():
            raise RuntimeError("a velocity sample reached beyond the ghost planes of this slab "
                               f"({self.ghost} planes): the fluid moves too fast for the slab halo")

    # -- particles that crossed a slab face change owner
    def migrate_particles(self):
        import torch
        import torch.distributed as dist

        if self.ctx.world == 1 or self.capacity == 0:
            return
        entries, n = self.compute.collect_leavers()
        staged = self.transport == "staged" or self.ctx.backend == "gloo"
        dev = torch.device("cpu") if staged else self.ctx.device
        counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(self.ctx.world)]
        dist.all_gather(counts, torch.tensor([n], dtype=torch.int64, device=dev))
        counts = [int(t.item()) for t in counts]
        most = max(counts)
        if most == 0:
            return
        width = most * E.FluidEngine.LEAVER_BYTES
        send = torch.zeros(width, dtype=torch.uint8, device=dev)
        if n:
            send[:n * E.FluidEngine.LEAVER_BYTES] = entries.to(dev)
        gathered = [torch.empty(width, dtype=torch.uint8, device=dev) for _ in range(self.ctx.world)]
        dist.all_gather(gathered, send)
        for r, cnt in enumerate(counts):
            if cnt and r != self.ctx.rank:
                self.compute.adopt(gathered[r].to(self.ctx.device) if staged else gathered[r], cnt)
        self.migrated += sum(counts)
        self.compute.sync()  # the gathered buffers are released when this returns

    # -- global state in (checkpoint restore): every rank passes the same global array
    def upload_image_global(self, image_id: int, array: np.ndarray):
        z0, n = self.slab
        self.compute.upload(image_id, np.ascontiguousarray(array[z0:z0 + n]))
        self.exchange_image(image_id, self.ghost)

    def upload_particles_global(self, particles: np.ndarray):
        self.compute.upload_particles(particles)  # the backend keeps the slots this slab owns

    # -- global views (tests, checkpoints): rank 0 gets the arrays, the others None
    def gather_image(self, image_id: int):
        import torch.distributed as dist

        local = self.compute.download(image_id)
        parts = [None] * self.ctx.world if self.ctx.rank == 0 else None
        dist.gather_object(local, parts, dst=0)
        return np.concatenate(parts, axis=0) if self.ctx.rank == 0 else None

    def gather_particles(self):
        import torch.distributed as dist

        local = self.compute.download_particles()
        parts = [None] * self.ctx.world if self.ctx.rank == 0 else None
        dist.gather_object(local, parts, dst=0)
        if self.ctx.rank != 0:
            return None
        out = np.zeros_like(parts[0])
        owners = np.zeros(out.shape[0], np.int32)
        for arr in parts:
            real = arr.view(np.uint32)[:, 3] != E.FluidEngine.TOMBSTONE_BITS
            out[real] = arr[real]
            owners += real
        if not np.all(owners == 1):
            raise RuntimeError(f"{int(np.sum(owners != 1))} particle slots do not have exactly one owner")
        return out

    def close(self):
        self.compute.close()
