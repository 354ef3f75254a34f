"""ctypes binding of include/fluid_slab.h — the multi-GPU driver (one process per GPU, Z slabs).

The schedule (which ghost planes travel when, deep halos and split passes of the Jacobi loop, particle
hand-over, the wider sampler halo for fast flows) is C++ in csrc/slab_driver.hip; planes travel by RCCL
Send/Recv issued from there.  This module is plumbing: it loads the symbols, wraps a driver in a class
whose methods follow main.cpp's frame loop (run_init once, run_step per frame), and offers
`torch.distributed` as

  * the bootstrap of the RCCL communicator (rank 0's unique id is broadcast over the process group the
    launcher set up; the data path never touches torch), and
  * a callback transport (fluid_slab_transport) for runs without RCCL: the multi-process CPU tests drive
    the same C++ schedule over gloo on host memory, and several ranks can rehearse on ONE GPU with the
    planes staged through the host.

The reference is single-GPU (main.cpp:43-48).
"""
import ctypes as C
import os
import time
from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np

from . import engine as E
from .params import PARAMS_BYTES, FluidParams, default_params

OVERLAP_NONE, OVERLAP_BEFORE, OVERLAP_BOTH = 0, 1, 2
OPT_OVERLAP, OPT_HALO_DEPTH, OPT_SAMPLER_HALO = 0, 1, 2
(STAT_EXCHANGES, STAT_OVERLAPPED, STAT_MIGRATED, STAT_SAMPLER_RERUNS, STAT_SAMPLER_WIDE,
 STAT_EFFECTIVE_HALO, STAT_SAMPLER_HALO, STAT_MIGRATE_ROUNDS, STAT_RCCL_RANKS, STAT_DRY_FACE_SKIPS) = range(10)
XFER_SEND, XFER_HOST_MEMORY = 1, 2
RCCL_ID_BYTES = 128
LOOP_PART_EDGES, LOOP_PART_INTERIOR = 1, 2

# every symbol include/fluid_slab.h declares
EXPORTED_SYMBOLS = [
    "fluid_slab_partition", "fluid_slab_create", "fluid_slab_create_custom", "fluid_slab_destroy",
    "fluid_slab_last_error", "fluid_slab_engine", "fluid_slab_get_slab", "fluid_slab_rccl_unique_id",
    "fluid_slab_attach_rccl", "fluid_slab_attach_transport", "fluid_slab_attach_loopback",
    "fluid_slab_attach_rccl_self",
    "fluid_slab_run_init", "fluid_slab_run_step", "fluid_slab_pressure_step", "fluid_slab_solve",
    "fluid_slab_exchange_image", "fluid_slab_set_option", "fluid_slab_get_stat", "fluid_slab_tune_exchange",
]


class SlabCreateInfo(C.Structure):
    _fields_ = [
        ("struct_bytes", C.c_uint32), ("rank", C.c_uint32), ("world", C.c_uint32),
        ("device", C.c_int32), ("params_blob", C.c_void_p), ("particle_capacity", C.c_uint64),
        ("pressure_iterations", C.c_uint32), ("halo_depth", C.c_uint32), ("overlap", C.c_int32),
        ("section_list", C.c_uint32), ("diffuse_mode", C.c_int32), ("sampler_halo", C.c_uint32),
    ]


class TuneResult(C.Structure):
    _fields_ = [("halo_depth", C.c_uint32), ("overlap", C.c_uint32), ("times_us", C.c_uint32 * 9)]


class Xfer(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("bytes", C.c_uint64), ("peer", C.c_int32), ("flags", C.c_uint32)]


class LoopBuffer(C.Structure):
    _fields_ = [("which", C.c_int32), ("planes", C.c_uint32)]


_EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(Xfer), C.c_uint32)
_ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint32), C.c_uint32)


class TransportTable(C.Structure):
    _fields_ = [("struct_bytes", C.c_uint32), ("reserved", C.c_uint32), ("user", C.c_void_p),
                ("exchange", _EXCHANGE_FN), ("allreduce_max_u32", _ALLREDUCE_FN)]


_vp, _i32, _u32, _u64 = C.c_void_p, C.c_int32, C.c_uint32, C.c_uint64
_pvp, _pu32, _pu64 = C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
_BACKEND_FIELDS = [   # order = include/fluid_slab.h: fluid_slab_backend
    ("run_section", [C.c_int]),
    ("run_section_group", [C.c_int, _u32]),
    ("image_planes", [C.c_int, _i32, _u32, _pvp, _pu64]),
    ("ghost_planes_written", [C.c_int]),
    ("loop_limits", [_pu32, _pu32]),
    ("loop_begin", [_u32, C.POINTER(LoopBuffer), _pu32]),
    ("loop_halo_exchanged", [_u32, C.c_int]),
    ("loop_advance", [_u32, _u32, C.c_int, C.c_int, _i32, _i32, C.POINTER(C.c_int)]),
    ("loop_end", []),
    ("loop_planes", [C.c_int, _i32, _u32, _pvp, _pu64]),
    ("slab_status", [_pu32]),
    ("set_sampler_halo", [_u32]),
    ("sampler_reach", [_pu32]),
    ("sampler_wide_begin", [_u32, _u32]),
    ("sampler_wide_planes", [_i32, _u32, _pvp, _pu64]),
    ("run_advect_wide", [C.c_int]),
    ("migrate_list", [C.c_int, _pvp, _pu32]),
    ("collect", [C.c_int, _pu32, _pu32]),
    ("adopt_received", [_u32, _u32, _pu32]),
    ("sync", []),
    ("step_begin", [C.c_int]),
    ("step_end", []),
    ("build_activity", []),
    ("activity_layer", [C.c_int, _pvp, _pu64]),
    ("step_status", [_pu32]),
    ("set_box", [C.c_int, _u32, _u32, _u32, _u32, _u32]),
]
_BACKEND_FN = {name: C.CFUNCTYPE(C.c_int, C.c_void_p, *args) for name, args in _BACKEND_FIELDS}


class BackendTable(C.Structure):
    _fields_ = ([("struct_bytes", C.c_uint32), ("reserved", C.c_uint32), ("user", C.c_void_p)]
                + [(name, _BACKEND_FN[name]) for name, _ in _BACKEND_FIELDS])


_declared = False


def _lib():
    """libfluid_engine.so with the fluid_slab_* signatures declared."""
    global _declared
    lib = E.load_library()
    if not _declared:
        vp = C.c_void_p
        sig = {
            "fluid_slab_partition": (C.c_int, [_u32, _u32, _u32, _pu32, _pu32]),
            "fluid_slab_create": (C.c_int, [_pvp, C.POINTER(SlabCreateInfo)]),
            "fluid_slab_create_custom": (C.c_int, [_pvp, C.POINTER(SlabCreateInfo),
                                                   C.POINTER(BackendTable)]),
            "fluid_slab_destroy": (None, [vp]),
            "fluid_slab_last_error": (C.c_char_p, [vp]),
            "fluid_slab_engine": (vp, [vp]),
            "fluid_slab_get_slab": (C.c_int, [vp, _pu32, _pu32]),
            "fluid_slab_rccl_unique_id": (C.c_int, [vp]),
            "fluid_slab_attach_rccl": (C.c_int, [vp, vp]),
            "fluid_slab_attach_transport": (C.c_int, [vp, C.POINTER(TransportTable)]),
            "fluid_slab_attach_loopback": (C.c_int, [vp, C.c_int, C.c_int]),
            "fluid_slab_attach_rccl_self": (C.c_int, [vp, C.c_int, C.c_int]),
            "fluid_slab_run_init": (C.c_int, [vp]),
            "fluid_slab_run_step": (C.c_int, [vp]),
            "fluid_slab_pressure_step": (C.c_int, [vp]),
            "fluid_slab_tune_exchange": (C.c_int, [vp, C.c_void_p]),
            "fluid_slab_solve": (C.c_int, [vp, _u32]),
            "fluid_slab_exchange_image": (C.c_int, [vp, C.c_int, _u32]),
            "fluid_slab_set_option": (C.c_int, [vp, C.c_int, C.c_int64]),
            "fluid_slab_get_stat": (C.c_int, [vp, C.c_int, _pu64]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _declared = True
    return lib


def partition_z(depth: int, world: int) -> List[Tuple[int, int]]:
    """fluid_slab_partition for every rank: (z_begin, z_count); the first depth % world ranks get one
    plane more."""
    if world < 1 or depth < world:
        raise ValueError(f"cannot split {depth} planes over {world} ranks")
    lib = _lib()
    out = []
    for r in range(world):
        z0, n = C.c_uint32(), C.c_uint32()
        if lib.fluid_slab_partition(depth, world, r, C.byref(z0), C.byref(n)) != 0:
            raise ValueError(f"cannot split {depth} planes over {world} ranks")
        out.append((int(z0.value), int(n.value)))
    return out


@dataclass
class DistContext:
    rank: int
    world: int
    device: int   # HIP device ordinal of this process (LOCAL_RANK)
    backend: str


def init_distributed(local_rank: int = 0, backend: Optional[str] = None) -> DistContext:
    """Join the process group described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT.  The group is
    used for bootstrap and measurement only (unique id broadcast, barriers, MAX of the timings) and runs
    over gloo on host tensors: torch never touches the GPU here.  The engine library is loaded first, so
    that it binds the ROCm installation's HIP runtime rather than the one a PyTorch wheel bundles (the
    driver then takes the librccl that belongs to that runtime, csrc/slab_driver.hip: rccl_api)."""
    E.load_library()
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not dist.is_initialized():
        dist.init_process_group(backend=backend or "gloo")
    return DistContext(dist.get_rank(), dist.get_world_size(), local_rank, dist.get_backend())


# ---- callback transport over torch.distributed --------------------------------------------------------
class TorchDistTransport:
    """fluid_slab_transport over a torch.distributed process group (gloo): point-to-point between the
    ranks the driver names, a MAX all-reduce for its 4-byte flags.  Host memory is addressed in place;
    device memory (`device_memory=True`: several ranks of a rehearsal sharing one GPU, where RCCL refuses
    to run) is staged through host buffers with hipMemcpy."""

    def __init__(self, device_memory: bool = False):
        import torch
        import torch.distributed as dist

        self.torch, self.dist = torch, dist
        self.device_memory = device_memory
        self.error = None
        self._hip = None
        if device_memory:
            self._hip = C.CDLL("libamdhip64.so.7")  # by soname: the runtime the engine library is bound to
            self._hip.hipMemcpy.restype = C.c_int
            self._hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.table = TransportTable()
        self.table.struct_bytes = C.sizeof(TransportTable)
        self._cb = (_EXCHANGE_FN(self._exchange), _ALLREDUCE_FN(self._allreduce))  # keep alive
        self.table.exchange, self.table.allreduce_max_u32 = self._cb

    def _host_view(self, ptr: int, nbytes: int):
        arr = np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(ptr))
        return self.torch.from_numpy(arr)

    def _exchange(self, _user, ops, count):
        try:
            torch, dist = self.torch, self.dist
            p2p, landings = [], []
            for i in range(count):
                x = ops[i]
                send = bool(x.flags & XFER_SEND)
                on_device = self.device_memory and not (x.flags & XFER_HOST_MEMORY)
                if on_device:
                    host = torch.empty(int(x.bytes), dtype=torch.uint8)
                    if send:
                        rc = self._hip.hipMemcpy(host.data_ptr(), x.ptr, int(x.bytes), 2)  # D2H
                        if rc:
                            raise RuntimeError(f"hipMemcpy D2H failed ({rc})")
                    else:
                        landings.append((host, int(x.ptr), int(x.bytes)))
                else:
                    host = self._host_view(int(x.ptr), int(x.bytes))
                p2p.append(dist.P2POp(dist.isend if send else dist.irecv, host, int(x.peer)))
            for work in dist.batch_isend_irecv(p2p):
                work.wait()
            for host, ptr, nbytes in landings:
                rc = self._hip.hipMemcpy(ptr, host.data_ptr(), nbytes, 1)  # H2D
                if rc:
                    raise RuntimeError(f"hipMemcpy H2D failed ({rc})")
            return 0
        except Exception as e:  # never let an exception cross the C frame
            self.error = e
            return -3

    def _allreduce(self, _user, values, count):
        try:
            t = self.torch.tensor([int(values[i]) for i in range(count)], dtype=self.torch.int64)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            for i in range(count):
                values[i] = int(t[i])
            return 0
        except Exception as e:
            self.error = e
            return -3


# ---- per-slab compute by callbacks (tests/host_standin.py puts the CPU oracle behind this) -------------
class CallbackBackend:
    """fluid_slab_backend filled from a Python object with the methods named below (the calls the driver
    makes on the engine, see include/fluid_slab.h).  Test infrastructure: the product computes with the
    HIP engine (SlabDriver without `compute`)."""

    def __init__(self, compute):
        self.compute = compute
        self.error = None
        self.k = 0
        self.table = BackendTable()
        self.table.struct_bytes = C.sizeof(BackendTable)
        self._keep = []
        for name, _ in _BACKEND_FIELDS:
            fn = _BACKEND_FN[name](self._guard(getattr(self, "_" + name)))
            self._keep.append(fn)
            setattr(self.table, name, fn)

    def _guard(self, f):
        def call(_user, *args):
            try:
                f(*args)
                return 0
            except Exception as e:
                self.error = e
                return -3
        return call

    @staticmethod
    def _region(t, ptr, nbytes):
        """t: a flat contiguous torch tensor / numpy array."""
        if hasattr(t, "data_ptr"):
            ptr[0], nbytes[0] = t.data_ptr(), t.numel() * t.element_size()
        else:
            ptr[0], nbytes[0] = t.ctypes.data, t.nbytes

    def _run_section(self, sid):
        self.compute.run_section(E.SECTION_NAMES[sid])

    def _run_section_group(self, first, n):
        self.compute.run_section_group(E.SECTION_NAMES[first], n)

    def _image_planes(self, image, first, count, ptr, nbytes):
        self._region(self.compute.image_planes(image, first, count), ptr, nbytes)

    def _ghost_planes_written(self, image):
        self.compute.halo_written(image)

    def _loop_limits(self, max_sweeps, max_halo):
        max_sweeps[0], max_halo[0] = self.compute.loop_max_sweeps(), self.compute.max_halo()

    def _loop_begin(self, halo, out, count):
        bufs = self.compute.loop_begin(halo)
        for i, (which, planes) in enumerate(bufs):
            out[i].which, out[i].planes = which, planes
        count[0] = len(bufs)

    def _loop_halo_exchanged(self, halo, first):
        self.compute.loop_halo_exchanged(halo, bool(first))

    def _loop_advance(self, k, sweeps, keep, part, lo, hi, written):
        name = {0: None, LOOP_PART_EDGES: "edges", LOOP_PART_INTERIOR: "interior"}[part]
        written[0] = self.compute.loop_advance(k, sweeps, bool(keep), name, (lo, hi))

    def _loop_end(self):
        self.compute.loop_end()

    def _loop_planes(self, which, first, count, ptr, nbytes):
        self._region(self.compute.planes(which, first, count), ptr, nbytes)

    def _slab_status(self, flag):
        flag[0] = 1 if self.compute.halo_violation() else 0

    def _set_sampler_halo(self, planes):
        self.compute.set_sampler_halo(planes)

    def _sampler_reach(self, planes):
        planes[0] = self.compute.sampler_reach()

    def _sampler_wide_begin(self, below, above):
        self.compute.sampler_wide_begin(below, above)

    def _sampler_wide_planes(self, first, count, ptr, nbytes):
        self._region(self.compute.sampler_wide_planes(first, count), ptr, nbytes)

    def _run_advect_wide(self, forces):
        self.compute.run_advect_wide(bool(forces))

    def _migrate_list(self, which, lst, cap):
        t, n = self.compute.migrate_list(which)
        p, b = (C.c_void_p * 1)(), (C.c_uint64 * 1)()
        self._region(t, p, b)
        lst[0], cap[0] = p[0], n

    def _collect(self, reset, counts, left):
        (counts[0], counts[1]), left[0] = self.compute.collect(bool(reset))

    def _adopt_received(self, nb, na, fwd):
        fwd[0], fwd[1] = self.compute.adopt_received(nb, na)

    def _sync(self):
        self.compute.sync()

    # a compute object without the engine's skipping: no activity layers to exchange, box unknown
    def _step_begin(self, section_list):
        getattr(self.compute, "step_begin", lambda: None)()

    def _step_end(self):
        getattr(self.compute, "step_end", lambda: None)()

    def _build_activity(self):
        pass

    def _activity_layer(self, which, ptr, nbytes):
        ptr[0], nbytes[0] = None, 0

    def _step_status(self, words):
        for i in range(8):
            words[i] = 0
        words[0] = 1 if self.compute.halo_violation() else 0

    def _set_box(self, valid, own, y0, y1, x0, x1):
        pass


class SlabError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"fluid slab driver error {code}: {message}")
        self.code = code


class SlabDriver:
    """One rank of a multi-GPU run (include/fluid_slab.h): construct, attach a transport, then
    ``run_init()`` once and ``run_step()`` per frame like main.cpp:111 / :172."""

    def __init__(self, params: FluidParams, rank: int, world: int, particle_capacity: int = 0,
                 pressure_iterations: int = 200, device: int = -1, halo_depth: int = 8,
                 overlap: Optional[int] = None, grouped: bool = True,
                 diffuse_mode: int = E.DIFFUSE_REFERENCE_EXACT, sampler_halo: int = 0, compute=None):
        self._lib = _lib()
        self._h = C.c_void_p()
        self.params = params.copy()
        self.rank, self.world = rank, world
        self.capacity = particle_capacity
        self.iterations = pressure_iterations
        self._blob = (C.c_uint8 * PARAMS_BYTES).from_buffer_copy(params.to_bytes())
        info = SlabCreateInfo()
        info.struct_bytes = C.sizeof(SlabCreateInfo)
        info.rank, info.world, info.device = rank, world, device
        info.params_blob = C.cast(self._blob, C.c_void_p)
        info.particle_capacity = particle_capacity
        info.pressure_iterations = pressure_iterations
        info.halo_depth = halo_depth
        info.overlap = -1 if overlap is None else overlap
        info.section_list = 0 if grouped else 1
        info.diffuse_mode = diffuse_mode
        info.sampler_halo = sampler_halo
        self._backend = CallbackBackend(compute) if compute is not None else None
        self.compute = compute
        if self._backend is not None:
            rc = self._lib.fluid_slab_create_custom(C.byref(self._h), C.byref(info),
                                                    C.byref(self._backend.table))
        else:
            rc = self._lib.fluid_slab_create(C.byref(self._h), C.byref(info))
        if rc != 0:
            msg = self._lib.fluid_slab_last_error(None)
            self._h = C.c_void_p()
            raise SlabError(rc, msg.decode() if msg else "fluid_slab_create failed")
        if self._backend is not None:
            self._backend.error = None   # creation ignores a backend without a sampler halo; do not keep its raise
        z0, n = C.c_uint32(), C.c_uint32()
        self._lib.fluid_slab_get_slab(self._h, C.byref(z0), C.byref(n))
        self.slab = (int(z0.value), int(n.value))
        self._transport = None
        self.engine = None
        if self._backend is None:
            self.engine = E.FluidEngine.from_handle(self._lib.fluid_slab_engine(self._h), params)

    # -- plumbing -----------------------------------------------------------------------------------
    def _check(self, rc: int):
        if rc == 0:
            return
        for holder in (self._backend, self._transport):  # a Python callback raised: that is the cause
            err = getattr(holder, "error", None)
            if err is not None:
                holder.error = None
                raise err
        msg = self._lib.fluid_slab_last_error(self._h)
        raise SlabError(rc, msg.decode() if msg else "")

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.fluid_slab_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- transport ------------------------------------------------------------------------------------
    def attach_rccl(self):
        """RCCL point-to-point between the GPUs: rank 0 makes the unique id, the torch.distributed group
        (gloo; host tensors) carries it to the others, every rank initialises its communicator."""
        buf = (C.c_uint8 * RCCL_ID_BYTES)()
        if self.rank == 0:
            rc = self._lib.fluid_slab_rccl_unique_id(buf)
            if rc != 0:
                msg = self._lib.fluid_slab_last_error(None)
                raise SlabError(rc, msg.decode() if msg else "")
        if self.world > 1:
            import torch
            import torch.distributed as dist

            t = torch.tensor(list(buf), dtype=torch.uint8)
            dist.broadcast(t, src=0)
            buf = (C.c_uint8 * RCCL_ID_BYTES)(*t.tolist())
        self._check(self._lib.fluid_slab_attach_rccl(self._h, buf))

    def attach_torch_transport(self, device_memory: bool = False):
        """The C++ schedule over torch.distributed point-to-point (gloo) instead of RCCL."""
        self._transport = TorchDistTransport(device_memory)
        self._check(self._lib.fluid_slab_attach_transport(self._h, C.byref(self._transport.table)))

    def attach_loopback(self, has_lower: bool = True, has_upper: bool = True):
        self._check(self._lib.fluid_slab_attach_loopback(self._h, int(has_lower), int(has_upper)))

    def attach_rccl_self(self, has_lower: bool = True, has_upper: bool = True):
        """The loopback rehearsal with ncclSend / ncclRecv to this rank itself (communicator of one)."""
        self._check(self._lib.fluid_slab_attach_rccl_self(self._h, int(has_lower), int(has_upper)))

    # -- the frame loop -------------------------------------------------------------------------------
    def run_init(self):
        self._check(self._lib.fluid_slab_run_init(self._h))

    def run_step(self):
        self._check(self._lib.fluid_slab_run_step(self._h))

    def pressure_step(self):
        """12a, 12b and the 12_solve_pressure loop section."""
        self._check(self._lib.fluid_slab_pressure_step(self._h))

    def solve(self, iterations: int = 0):
        self._check(self._lib.fluid_slab_solve(self._h, iterations))

    def exchange_image(self, image_id: int, planes: int = 1):
        self._check(self._lib.fluid_slab_exchange_image(self._h, image_id, planes))

    def set_option(self, option: int, value: int):
        self._check(self._lib.fluid_slab_set_option(self._h, option, value))

    def tune_exchange(self) -> dict:
        """fluid_slab_tune_exchange: the loop's exchange schedules measured where they run, the fastest adopted
        (collective).  Returns the choice and the times in ms keyed "h<depth>_overlap<mode>"."""
        r = TuneResult()
        self._check(self._lib.fluid_slab_tune_exchange(self._h, C.byref(r)))
        times = {f"h{h}_overlap{m}": r.times_us[3 * i + m] / 1e3
                 for i, h in enumerate((8, 6, 3)) for m in range(3) if r.times_us[3 * i + m]}
        return {"halo_depth": int(r.halo_depth), "overlap": int(r.overlap), "times_ms": times}

    def stat(self, which: int) -> int:
        v = C.c_uint64(0)
        self._check(self._lib.fluid_slab_get_stat(self._h, which, C.byref(v)))
        return int(v.value)

    @property
    def image_ghost(self) -> int:
        return min(E.FluidEngine.IMAGE_GHOST_PLANES, self.params.size[2] // self.world)

    # -- this rank's part of global arrays (tests, checkpoints, scenes) -----------------------------------
    def _store(self):
        return self.engine if self.engine is not None else self.compute

    def upload_image_global(self, image_id: int, array: np.ndarray):
        """Every rank passes the same global array; the slab keeps its planes and fetches its ghosts."""
        z0, n = self.slab
        self.upload_image(image_id, np.ascontiguousarray(array[z0:z0 + n]))

    def upload_image(self, image_id: int, local: np.ndarray):
        st = self._store()
        (st.upload_image if self.engine is not None else st.upload)(image_id, local)
        self.exchange_image(image_id, self.image_ghost)

    def download_image(self, image_id: int) -> np.ndarray:
        st = self._store()
        return (st.download_image if self.engine is not None else st.download)(image_id)

    def upload_particles_global(self, particles: np.ndarray):
        self._store().upload_particles(particles)  # the backend keeps the slots this slab owns

    def gather_image(self, image_id: int):
        """Rank 0 gets the global array, the others None."""
        import torch.distributed as dist

        local = self.download_image(image_id)
        if self.world == 1:
            return local
        parts = [None] * self.world if self.rank == 0 else None
        dist.gather_object(local, parts, dst=0)
        return np.concatenate(parts, axis=0) if self.rank == 0 else None

    def gather_particles(self):
        import torch.distributed as dist

        local = self._store().download_particles()
        if self.world == 1:
            return local
        parts = [None] * self.world if self.rank == 0 else None
        dist.gather_object(local, parts, dst=0)
        if self.rank != 0:
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

    # -- measurement (bench.py --gpus N) ------------------------------------------------------------------
    @classmethod
    def create_full_fluid(cls, size, iterations: int, ctx: DistContext, seed: Optional[int] = None,
                          **kw) -> "SlabDriver":
        """The Jacobi benchmark scene (scenes.py) on this rank's slab of the global grid, RCCL attached."""
        from . import scenes

        w, h, d = size
        drv = cls(default_params(w, h, d, 0), ctx.rank, ctx.world, pressure_iterations=iterations,
                  device=ctx.device, **kw)
        if ctx.world > 1:
            drv.attach_rccl()
        z0, n = drv.slab
        shape = (n, h, w)
        drv.upload_image(E.CELL_TYPES, scenes.full_fluid_types(shape, z0, d))
        drv.engine.upload_image(E.DIVERGENCES, scenes.full_fluid_divergence(
            shape, scenes.SEED_JACOBI if seed is None else seed, z0))
        return drv

    def benchmark(self, steps: int, warmup: int, step=None) -> dict:
        """`steps` timed calls of `step` (default: pressure_step) between barriers; MAX over the ranks."""
        import torch
        import torch.distributed as dist

        step = step or self.pressure_step
        multi = self.world > 1 and dist.is_initialized()
        probe = self._probe_overlap(step) if step == self.pressure_step else {"probed": False}

        def fence():
            self._store().sync()
            if multi:
                dist.barrier()
                self._store().sync()

        for _ in range(warmup):
            step()
        fence()
        if self.engine is not None:
            self.engine.enable_timing(True)
            self.engine.reset_timing()
        x0 = self.stat(STAT_EXCHANGES)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        fence()
        wall = time.perf_counter() - t0
        exchanges = (self.stat(STAT_EXCHANGES) - x0) / max(steps, 1)
        ms, calls = 0.0, 0
        if self.engine is not None:
            ms, calls = self.engine.section_time_ms("12_solve_pressure")
            self.engine.enable_timing(False)
        vals = torch.tensor([wall, ms / max(calls, 1)], dtype=torch.float64)
        if multi:
            dist.all_reduce(vals, op=dist.ReduceOp.MAX)
        wall, kernel = (float(v) for v in vals)
        w, h, _ = self.params.size
        return {"wall_s": wall, "local_cells": w * h * self.slab[1], "halo_overlap": probe,
                "exchanges_per_step": exchanges,
                "kernel_ms_per_sweep": kernel,
                "exchange_ms_per_sweep": max(0.0, 1e3 * wall / (steps * self.iterations) - kernel)}

    def _probe_overlap(self, step) -> dict:
        """Untimed, before the warm-up: the exchange schedule is measured where it runs.  For each halo depth
        (planes per exchange = sweeps between two exchanges: 8, 6 and 3 where the loop applies three sweeps per
        launch — 3 + 3 + 2, 3 + 3 or 3 sweeps per exchange) one step with the exchanges in line, one with only
        the pass before each exchange split and one with both passes split, each after a step of its own; every
        rank adopts the fastest combination (the times are MAX over the ranks, so all agree).  Whether hiding an
        exchange is worth two extra launches, and whether fewer, larger messages beat less recomputation of
        ghost planes, depends on the link.  FLUID_SLAB_OVERLAP=0 / before / 1 forces a schedule,
        FLUID_SLAB_HALO=h a depth."""
        import torch
        import torch.distributed as dist

        env = os.environ.get("FLUID_SLAB_OVERLAP")
        env_h = os.environ.get("FLUID_SLAB_HALO")
        if env_h is not None:
            self.set_option(OPT_HALO_DEPTH, int(env_h))
        if env is not None:
            mode = {"0": OVERLAP_NONE, "before": OVERLAP_BEFORE}.get(env, OVERLAP_BOTH)
            self.set_option(OPT_OVERLAP, mode)
            return {"used": mode, "probed": False, "halo_depth": self.stat(STAT_EFFECTIVE_HALO)}
        if self.world == 1 or not dist.is_initialized():
            return {"used": OVERLAP_NONE, "probed": False, "halo_depth": self.stat(STAT_EFFECTIVE_HALO)}
        if env_h is None:
            # the driver's own tuner (fluid_slab_tune_exchange): depths 8 / 6 / 3 x the three schedules
            r = self.tune_exchange()
            t = r["times_ms"]
            key = lambda m: t.get(f"h{r['halo_depth']}_overlap{m}")
            return {"used": r["overlap"], "probed": True, "halo_depth": r["halo_depth"],
                    "step_ms_inline": key(OVERLAP_NONE), "step_ms_overlap_before": key(OVERLAP_BEFORE),
                    "step_ms_overlapped": key(OVERLAP_BOTH),
                    "step_ms_by_halo_depth_and_schedule": {k: round(v, 4) for k, v in t.items()}}
        depths = [int(env_h)]  # a forced depth: its three schedules, timed here
        times = {}
        for h in depths:
            self.set_option(OPT_HALO_DEPTH, h)
            for mode in (OVERLAP_NONE, OVERLAP_BEFORE, OVERLAP_BOTH):
                self.set_option(OPT_OVERLAP, mode)
                step()
                self._store().sync()
                dist.barrier()
                t0 = time.perf_counter()
                step()
                self._store().sync()
                t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                times[(h, mode)] = float(t[0])
        best_h, best = min(times, key=times.get)
        self.set_option(OPT_HALO_DEPTH, best_h)
        self.set_option(OPT_OVERLAP, best)
        return {"used": best, "probed": True, "halo_depth": best_h,
                "step_ms_inline": 1e3 * times[(best_h, OVERLAP_NONE)],
                "step_ms_overlap_before": 1e3 * times[(best_h, OVERLAP_BEFORE)],
                "step_ms_overlapped": 1e3 * times[(best_h, OVERLAP_BOTH)],
                "step_ms_by_halo_depth_and_schedule": {f"h{h}_overlap{m}": round(1e3 * v, 4)
                                                       for (h, m), v in times.items()}}
