"""ctypes binding of the engine's C ABI (include/fluid_engine.h) — plumbing only.

The product is ``libfluid_engine.so`` (HIP kernels + C++ host, csrc/).  This module loads it,
declares every exported symbol, and wraps a context in a small class whose method names follow the
reference's section lists (/root/reference/fluid_flow_sections.h:136-338).  There is no fallback:
if the library is missing or no gfx950 device is present, construction raises.
"""
import ctypes as C
import os
from typing import Optional

import numpy as np

from .params import PARAMS_BYTES, FluidParams

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfluid_engine.so")

# ---- enums of include/fluid_engine.h -------------------------------------------------------------
# ImageAttachments / BufferAttachments, fluid_flow_sections.h:10-16
VELOCITIES_1, VELOCITIES_2, CELL_TYPES, NEW_CELL_TYPES = 0, 1, 2, 3
PRESSURES_1, PRESSURES_2, DIVERGENCES, PARTICLE_DENSITIES_IMG = 4, 5, 6, 7
DETAILED_DENSITIES_IMG, DETAILED_DENSITIES_INERTIA_IMG = 8, 9
PARTICLE_DENSITIES_FLOAT_1, PARTICLE_DENSITIES_FLOAT_2 = 10, 11
IMAGE_COUNT = 12
PARTICLES_BUF, MARCHING_CUBES_COUNTS_BUF, MARCHING_CUBES_EDGES_BUF, SIMULATION_PARAMS_BUF = 0, 1, 2, 3

IMAGE_DTYPES = {
    VELOCITIES_1: (np.float32, 4), VELOCITIES_2: (np.float32, 4),
    CELL_TYPES: (np.uint8, 1), NEW_CELL_TYPES: (np.uint8, 1),
    PRESSURES_1: (np.float32, 1), PRESSURES_2: (np.float32, 1), DIVERGENCES: (np.float32, 1),
    PARTICLE_DENSITIES_IMG: (np.uint32, 1),
}
# images on the detailed grid (contexts created with surface_prep=True)
SURFACE_DTYPES = {
    DETAILED_DENSITIES_IMG: np.uint32, DETAILED_DENSITIES_INERTIA_IMG: np.uint32,
    PARTICLE_DENSITIES_FLOAT_1: np.float32, PARTICLE_DENSITIES_FLOAT_2: np.float32,
}

SECTION_NAMES = [
    "init_clear_velocities_1", "init_clear_cell_types", "00_init_particles",
    "01a_clear_particle_densities", "01_update_densities", "02_update_water", "03_update_air",
    "04_compute_extrapolated_velocities", "05_set_extrapolated_velocities", "06_update_cell_types",
    "07_advect", "08_forces", "09_diffuse", "10_solids", "11_compute_divergence",
    "12a_clear_pressures_1", "12b_clear_pressures_2", "12_solve_pressure", "13_fix_divergence",
    "14_particles",
    # surface-prep passes (surface_prep contexts)
    "14a_clear_detailed_densities", "15_update_detailed_densities",
    "16_compute_detailed_densities_inertia", "17_compute_float_densities",
    "18_diffuse_float_densities", "init_clear_detailed_densities_inertia",
]
SECTION_IDS = {name: i for i, name in enumerate(SECTION_NAMES)}
SEC_12_SOLVE_PRESSURE = SECTION_IDS["12_solve_pressure"]
SECTION_COUNT = len(SECTION_NAMES)

DIFFUSE_REFERENCE_EXACT, DIFFUSE_INTENDED = 0, 1
OPT_PRESSURE_KERNEL = 0
OPT_JACOBI_FUSE = 1
OPT_STEP_FUSION = 2
OPT_QUIET_BRICKS = 3
OPT_ADVECT_KERNEL = 4
OPT_SURFACE_KERNEL = 5
OPT_LAUNCH_BOX = 6
OPT_EDGE_STREAM = 7
OPT_PARTICLE_SORT = 8
STAT_BRICKS, STAT_QUIET_BRICKS, STAT_PARTICLE_SORTS, STAT_PARTICLE_STRAYS, STAT_PARTICLE_BINNED = 0, 1, 2, 3, 4
STAT_PARTICLE_ENTRIES, STAT_OWNED_SQUEEZES = 5, 6

OK, ERR_INVALID_ARG, ERR_SIZE_MISMATCH, ERR_HIP, ERR_NO_DEVICE, ERR_UNSUPPORTED, ERR_OOM = (
    0, -1, -2, -3, -4, -5, -6)

# every symbol include/fluid_engine.h declares
EXPORTED_SYMBOLS = [
    "fluid_abi_version", "fluid_params_default", "fluid_required_arena_bytes", "fluid_create",
    "fluid_destroy", "fluid_last_error", "fluid_upload_image", "fluid_download_image",
    "fluid_upload_buffer", "fluid_download_buffer", "fluid_image_bytes", "fluid_buffer_bytes",
    "fluid_set_params", "fluid_set_pressure_iterations", "fluid_set_diffuse_mode",
    "fluid_run_section", "fluid_run_section_loop", "fluid_run_section_group", "fluid_clear_image",
    "fluid_run_surface_diffuse_dispatch", "fluid_set_pressure_solver",
    "fluid_run_pressure_dispatch", "fluid_run_init",
    "fluid_run_step", "fluid_sync", "fluid_enable_timing", "fluid_section_time_ms", "fluid_section_name",
    "fluid_reset_timing", "fluid_image_plane_ptr", "fluid_notify_image_written",
    "fluid_notify_ghost_planes_written", "fluid_get_stat", "fluid_pressure_residual",
    "fluid_pressure_loop_begin", "fluid_pressure_loop_max_sweeps", "fluid_pressure_loop_advance",
    "fluid_pressure_loop_advance_part", "fluid_pressure_loop_advance_part_n", "fluid_pressure_loop_edge_stream",
    "fluid_pressure_loop_halo_exchanged", "fluid_pressure_loop_end", "fluid_pressure_loop_plane_ptr",
    "fluid_slab_status", "fluid_particles_migrate_list", "fluid_particles_collect",
    "fluid_particles_adopt_received", "fluid_pressure_loop_available",
    "fluid_get_geometry", "fluid_set_option", "fluid_count_nonfinite",
    "fluid_set_sampler_halo", "fluid_sampler_reach", "fluid_sampler_wide_begin",
    "fluid_sampler_wide_plane_ptr", "fluid_run_advect_wide",
    "fluid_step_begin", "fluid_step_end", "fluid_step_build_activity", "fluid_activity_layer_ptr",
    "fluid_step_status", "fluid_step_set_box", "fluid_extract_surface",
]


class CreateInfo(C.Structure):
    _fields_ = [
        ("struct_bytes", C.c_uint32),
        ("device", C.c_int32),
        ("params_blob", C.c_void_p),
        ("particle_capacity", C.c_uint64),
        ("pressure_iterations", C.c_uint32),
        ("slab_z_begin", C.c_uint32),
        ("slab_z_count", C.c_uint32),
        ("hip_stream", C.c_void_p),
        ("arena", C.c_void_p),
        ("arena_bytes", C.c_uint64),
        ("surface_prep", C.c_uint32),
        ("surface_diffuse_steps", C.c_uint32),
    ]


class FluidEngineError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"fluid engine error {code}: {message}")
        self.code = code


_lib = None


def load_library(path: Optional[str] = None) -> C.CDLL:
    """Load libfluid_engine.so and declare the ABI.  Raises if the library has not been built
    (run ``python -c 'import __graft_entry__ as g; g.build()'`` or ``make -C …/csrc``)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise FileNotFoundError(
            f"{path} not found: the HIP engine has not been built (make -C "
            f"{os.path.join(_HERE, 'csrc')}); there is no CPU fallback")
    lib = C.CDLL(path)
    vp, i32, u32, u64 = C.c_void_p, C.c_int32, C.c_uint32, C.c_uint64
    sig = {
        "fluid_abi_version": (C.c_int, []),
        "fluid_params_default": (C.c_int, [vp, u32, u32, u32, u32]),
        "fluid_required_arena_bytes": (u64, [C.POINTER(CreateInfo)]),
        "fluid_create": (C.c_int, [C.POINTER(vp), C.POINTER(CreateInfo)]),
        "fluid_destroy": (None, [vp]),
        "fluid_last_error": (C.c_char_p, [vp]),
        "fluid_upload_image": (C.c_int, [vp, C.c_int, vp, u64]),
        "fluid_download_image": (C.c_int, [vp, C.c_int, vp, u64]),
        "fluid_upload_buffer": (C.c_int, [vp, C.c_int, vp, u64]),
        "fluid_download_buffer": (C.c_int, [vp, C.c_int, vp, u64]),
        "fluid_image_bytes": (C.c_int, [vp, C.c_int, C.POINTER(u64)]),
        "fluid_buffer_bytes": (C.c_int, [vp, C.c_int, C.POINTER(u64)]),
        "fluid_set_params": (C.c_int, [vp, vp]),
        "fluid_set_pressure_iterations": (C.c_int, [vp, u32]),
        "fluid_set_diffuse_mode": (C.c_int, [vp, C.c_int]),
        "fluid_run_section": (C.c_int, [vp, C.c_int]),
        "fluid_run_section_loop": (C.c_int, [vp, C.c_int, u32]),
        "fluid_run_section_group": (C.c_int, [vp, C.c_int, u32]),
        "fluid_run_surface_diffuse_dispatch": (C.c_int, [vp, u32]),
        "fluid_set_pressure_solver": (C.c_int, [vp, C.c_int, C.c_float]),
        "fluid_clear_image": (C.c_int, [vp, C.c_int, C.POINTER(u32 * 4)]),
        "fluid_run_pressure_dispatch": (C.c_int, [vp, u32]),
        "fluid_run_init": (C.c_int, [vp]),
        "fluid_run_step": (C.c_int, [vp]),
        "fluid_sync": (C.c_int, [vp]),
        "fluid_enable_timing": (C.c_int, [vp, C.c_int]),
        "fluid_section_time_ms": (C.c_int, [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(u64)]),
        "fluid_section_name": (C.c_char_p, [C.c_int]),
        "fluid_reset_timing": (C.c_int, [vp]),
        "fluid_image_plane_ptr": (C.c_int, [vp, C.c_int, i32, C.POINTER(vp), C.POINTER(u64)]),
        "fluid_notify_image_written": (C.c_int, [vp, C.c_int]),
        "fluid_notify_ghost_planes_written": (C.c_int, [vp, C.c_int]),
        "fluid_get_stat": (C.c_int, [vp, C.c_int, C.POINTER(C.c_uint64)]),
        "fluid_pressure_residual": (C.c_int, [vp, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_double),
                                              C.POINTER(C.c_uint64)]),
        "fluid_pressure_loop_begin": (C.c_int, [vp]),
        "fluid_pressure_loop_max_sweeps": (C.c_int, [vp]),
        "fluid_pressure_loop_advance": (C.c_int, [vp, u32, C.c_int, C.POINTER(C.c_int)]),
        "fluid_pressure_loop_advance_part": (C.c_int, [vp, C.c_int, C.c_int, C.c_int32, C.c_int32,
                                                       C.POINTER(C.c_int)]),
        "fluid_pressure_loop_advance_part_n": (C.c_int, [vp, u32, C.c_int, C.c_int, C.c_int32, C.c_int32,
                                                         C.POINTER(C.c_int)]),
        "fluid_pressure_loop_edge_stream": (C.c_int, [vp, C.POINTER(vp)]),
        "fluid_pressure_loop_halo_exchanged": (C.c_int, [vp, u32, u32]),
        "fluid_pressure_loop_end": (C.c_int, [vp]),
        "fluid_pressure_loop_plane_ptr": (C.c_int, [vp, C.c_int, i32, C.POINTER(vp),
                                                    C.POINTER(u64)]),
        "fluid_slab_status": (C.c_int, [vp, C.POINTER(u32)]),
        "fluid_particles_migrate_list": (C.c_int, [vp, C.c_int, C.POINTER(vp), C.POINTER(u32)]),
        "fluid_particles_collect": (C.c_int, [vp, C.c_int, C.POINTER(u32 * 2), C.POINTER(u32)]),
        "fluid_particles_adopt_received": (C.c_int, [vp, u32, u32, C.POINTER(u32 * 2)]),
        "fluid_pressure_loop_available": (C.c_int, [vp]),
        "fluid_get_geometry": (C.c_int, [vp, C.POINTER(u32 * 3), C.POINTER(u32), C.POINTER(u32),
                                         C.POINTER(u64)]),
        "fluid_set_option": (C.c_int, [vp, C.c_int, C.c_int64]),
        "fluid_count_nonfinite": (C.c_int, [vp, C.c_int, C.POINTER(C.c_uint64)]),
        "fluid_set_sampler_halo": (C.c_int, [vp, u32]),
        "fluid_sampler_reach": (C.c_int, [vp, C.POINTER(u32)]),
        "fluid_sampler_wide_begin": (C.c_int, [vp, u32, u32]),
        "fluid_sampler_wide_plane_ptr": (C.c_int, [vp, i32, C.POINTER(vp), C.POINTER(u64)]),
        "fluid_run_advect_wide": (C.c_int, [vp, C.c_int]),
        "fluid_step_begin": (C.c_int, [vp, C.c_int]),
        "fluid_step_end": (C.c_int, [vp]),
        "fluid_step_build_activity": (C.c_int, [vp]),
        "fluid_activity_layer_ptr": (C.c_int, [vp, C.c_int, C.POINTER(vp), C.POINTER(u64)]),
        "fluid_step_status": (C.c_int, [vp, C.POINTER(u32 * 8)]),
        "fluid_step_set_box": (C.c_int, [vp, C.c_int, u32, u32, u32, u32, u32]),
        "fluid_extract_surface": (C.c_int, [vp, C.c_int, vp, u64, C.POINTER(u64)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if path == LIB_PATH:
        _lib = lib
    return lib


class FluidEngine:
    """One engine context = one GPU (optionally one Z-slab of the global grid).

    Mirrors how main.cpp drives the reference: construct (SimulationDescriptors + params upload,
    fluid_flow_sections.h:26-96), ``run_init()`` once (main.cpp:111), ``run_step()`` per frame
    (main.cpp:172); individual sections by name via ``run_section``."""

    def __init__(self, params: FluidParams, particle_capacity: int = 0,
                 pressure_iterations: int = 200, device: int = -1,
                 slab: Optional[tuple] = None, stream: int = 0, arena: int = 0,
                 arena_bytes: int = 0, lib_path: Optional[str] = None,
                 surface_prep: bool = False, surface_diffuse_steps: int = 0):
        self._lib = load_library(lib_path)
        self.surface_prep = bool(surface_prep)
        self._h = C.c_void_p()
        self._blob = (C.c_uint8 * PARAMS_BYTES).from_buffer_copy(params.to_bytes())
        info = CreateInfo()
        info.struct_bytes = C.sizeof(CreateInfo)
        info.device = device
        info.params_blob = C.cast(self._blob, C.c_void_p)
        info.particle_capacity = particle_capacity
        info.pressure_iterations = pressure_iterations
        if slab is not None:
            info.slab_z_begin, info.slab_z_count = int(slab[0]), int(slab[1])
        info.hip_stream = stream or None
        info.arena = arena or None
        info.arena_bytes = arena_bytes
        info.surface_prep = 1 if surface_prep else 0
        info.surface_diffuse_steps = surface_diffuse_steps
        rc = self._lib.fluid_create(C.byref(self._h), C.byref(info))
        if rc != OK:
            msg = self._lib.fluid_last_error(None)
            self._h = C.c_void_p()
            raise FluidEngineError(rc, msg.decode() if msg else "fluid_create failed")
        self.params = params.copy()
        # run state the wrapper set (the C ABI has setters only): what a checkpoint records
        self.pressure_iterations = int(pressure_iterations) if pressure_iterations else 200
        self.diffuse_mode = DIFFUSE_REFERENCE_EXACT
        self.solver, self.sor_omega = 0, 1.0
        self.surface_diffuse_steps = int(surface_diffuse_steps) if surface_diffuse_steps else 4
        self._read_geometry()

    @classmethod
    def from_handle(cls, handle: int, params: FluidParams) -> "FluidEngine":
        """A view of a context somebody else owns (the slab driver's, fluid_slab_engine): every method
        works, close() leaves the context alone."""
        self = cls.__new__(cls)
        self._lib = load_library()
        self._h = C.c_void_p(handle)
        self._borrowed = True
        self.surface_prep = False
        self.params = params.copy()
        self.pressure_iterations, self.diffuse_mode = 200, DIFFUSE_REFERENCE_EXACT
        self.solver, self.sor_omega, self.surface_diffuse_steps = 0, 1.0, 4
        self._read_geometry()
        return self

    def _read_geometry(self):
        size = (C.c_uint32 * 3)()
        z0, zc, cap = C.c_uint32(), C.c_uint32(), C.c_uint64()
        self._check(self._lib.fluid_get_geometry(self._h, C.byref(size), C.byref(z0), C.byref(zc),
                                                 C.byref(cap)))
        self.global_size = (int(size[0]), int(size[1]), int(size[2]))
        self.slab_z_begin, self.slab_z_count = int(z0.value), int(zc.value)
        self.particle_capacity = int(cap.value)

    # -- helpers --------------------------------------------------------------------------------
    @staticmethod
    def required_arena_bytes(params: FluidParams, particle_capacity: int = 0,
                             slab: Optional[tuple] = None) -> int:
        lib = load_library()
        blob = (C.c_uint8 * PARAMS_BYTES).from_buffer_copy(params.to_bytes())
        info = CreateInfo()
        info.struct_bytes = C.sizeof(CreateInfo)
        info.params_blob = C.cast(blob, C.c_void_p)
        info.particle_capacity = particle_capacity
        if slab is not None:
            info.slab_z_begin, info.slab_z_count = int(slab[0]), int(slab[1])
        return int(lib.fluid_required_arena_bytes(C.byref(info)))

    def _check(self, rc: int):
        if rc != OK:
            msg = self._lib.fluid_last_error(self._h)
            raise FluidEngineError(rc, msg.decode() if msg else "")

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            if not getattr(self, "_borrowed", False):
                self._lib.fluid_destroy(self._h)
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

    @property
    def local_shape(self):
        w, h, _ = self.global_size
        return (self.slab_z_count, h, w)

    # -- data movement --------------------------------------------------------------------------
    @property
    def detailed_shape(self):
        """(D, H, W) of the detailed grid of the surface-prep passes."""
        r = int(self.params.detailed_resolution)
        w, h, d = self.global_size
        return (d * r, h * r, w * r)

    def image_shape(self, image_id: int):
        if image_id in SURFACE_DTYPES:
            return SURFACE_DTYPES[image_id], self.detailed_shape
        dtype, ch = IMAGE_DTYPES[image_id]
        shape = self.local_shape + ((ch,) if ch > 1 else ())
        return dtype, shape

    def upload_image(self, image_id: int, array: np.ndarray):
        if image_id in IMAGE_DTYPES or (image_id in SURFACE_DTYPES and self.surface_prep):
            dtype, shape = self.image_shape(image_id)
            array = np.ascontiguousarray(array, dtype=dtype)
            if array.size != int(np.prod(shape)):
                raise FluidEngineError(ERR_SIZE_MISMATCH,
                                       f"image {image_id} expects shape {shape}, got {array.shape}")
        else:
            array = np.ascontiguousarray(array)
        self._check(self._lib.fluid_upload_image(self._h, image_id, array.ctypes.data,
                                                 array.nbytes))

    def download_image(self, image_id: int) -> np.ndarray:
        if image_id not in IMAGE_DTYPES and not (image_id in SURFACE_DTYPES and self.surface_prep):
            self._check(self._lib.fluid_download_image(self._h, image_id, None, 0))
        dtype, shape = self.image_shape(image_id)
        out = np.empty(shape, dtype=dtype)
        self._check(self._lib.fluid_download_image(self._h, image_id, out.ctypes.data, out.nbytes))
        return out

    # -- state I/O: everything a step reads, as one .npz -------------------------
    CHECKPOINT_VERSION = 2

    @staticmethod
    def _checkpoint_path(path: str) -> str:
        """numpy appends ".npz" to a name without it; use one spelling on both sides."""
        path = os.fspath(path)
        return path if path.endswith(".npz") else path + ".npz"

    def save_checkpoint(self, path: str):
        """Images 0..7 (and the detailed-grid images of a surface_prep context), the particle buffer, the
        parameter block and the run state a resumed simulation needs for identical results (iteration
        counts, 09_diffuse mode, solver).  Whole-grid contexts; a Z-slab run gathers per image
        (slab.SlabSimulation).  `path` gets the suffix ".npz" if it lacks it."""
        data = {"params": np.frombuffer(self.params.to_bytes(), dtype=np.uint8),
                "particles": self.download_particles(),
                "meta": np.array([self.particle_capacity, 1 if self.surface_prep else 0], np.int64),
                "version": np.array([self.CHECKPOINT_VERSION], np.int64),
                "run_state": np.array([self.pressure_iterations, self.diffuse_mode, self.solver,
                                       self.surface_diffuse_steps], np.int64),
                "sor_omega": np.array([self.sor_omega], np.float32)}
        for img in IMAGE_DTYPES:
            data[f"image_{img}"] = self.download_image(img)
        if self.surface_prep:
            for img in SURFACE_DTYPES:
                data[f"image_{img}"] = self.download_image(img)
        with open(self._checkpoint_path(path), "wb") as f:
            np.savez(f, **data)

    def restore_checkpoint(self, path: str):
        """Load a checkpoint written by save_checkpoint into this context (same grid, capacity and
        parameter block sizes).  The parameter values and the run state of the file replace the context's
        (a version-1 file carries no run state: the context's stays).  The number of blur dispatches of a
        surface_prep context is fixed at creation: a file written with another count is refused."""
        with np.load(self._checkpoint_path(path), allow_pickle=False) as z:
            version = int(z["version"][0]) if "version" in z.files else 1
            if version > self.CHECKPOINT_VERSION:
                raise FluidEngineError(ERR_INVALID_ARG, f"checkpoint format {version} is newer than "
                                                        f"this build's ({self.CHECKPOINT_VERSION})")
            blob = z["params"].tobytes()
            if len(blob) != PARAMS_BYTES or int(z["meta"][0]) != self.particle_capacity:
                raise FluidEngineError(ERR_SIZE_MISMATCH, "checkpoint does not fit this context")
            params = FluidParams.from_buffer_copy(blob)
            if tuple(params.size) != tuple(self.global_size):
                raise FluidEngineError(ERR_SIZE_MISMATCH, "checkpoint grid differs from this context's")
            if version >= 2:
                iters, mode, solver, blur = (int(v) for v in z["run_state"])
                if self.surface_prep and int(z["meta"][1]) and blur != self.surface_diffuse_steps:
                    raise FluidEngineError(
                        ERR_SIZE_MISMATCH, f"checkpoint was written with {blur} blur dispatches, this "
                                           f"context runs {self.surface_diffuse_steps}")
                self.set_pressure_iterations(iters)
                self.set_diffuse_mode(mode)
                self.set_pressure_solver(solver, float(z["sor_omega"][0]))
            self.set_params(params)
            for img in IMAGE_DTYPES:
                self.upload_image(img, z[f"image_{img}"])
            if self.surface_prep and int(z["meta"][1]):
                for img in SURFACE_DTYPES:
                    self.upload_image(img, z[f"image_{img}"])
            if self.particle_capacity:
                self.upload_particles(z["particles"])

    def upload_particles(self, particles: np.ndarray):
        particles = np.ascontiguousarray(particles, dtype=np.float32)
        self._check(self._lib.fluid_upload_buffer(self._h, PARTICLES_BUF, particles.ctypes.data,
                                                  particles.nbytes))

    def download_particles(self) -> np.ndarray:
        out = np.empty((self.particle_capacity, 4), dtype=np.float32)
        self._check(self._lib.fluid_download_buffer(self._h, PARTICLES_BUF, out.ctypes.data,
                                                    out.nbytes))
        return out

    def set_params(self, params: FluidParams):
        blob = (C.c_uint8 * PARAMS_BYTES).from_buffer_copy(params.to_bytes())
        self._check(self._lib.fluid_set_params(self._h, C.cast(blob, C.c_void_p)))
        self.params = params.copy()

    def download_params(self) -> FluidParams:
        blob = (C.c_uint8 * PARAMS_BYTES)()
        self._check(self._lib.fluid_download_buffer(self._h, SIMULATION_PARAMS_BUF,
                                                    C.cast(blob, C.c_void_p), PARAMS_BYTES))
        return FluidParams.from_bytes(bytes(blob))

    # -- sections ---------------------------------------------------------------------------------
    def run_section(self, section):
        sid = SECTION_IDS[section] if isinstance(section, str) else int(section)
        self._check(self._lib.fluid_run_section(self._h, sid))

    def run_section_loop(self, section, iterations: int):
        sid = SECTION_IDS[section] if isinstance(section, str) else int(section)
        self._check(self._lib.fluid_run_section_loop(self._h, sid, iterations))

    def run_section_group(self, first_section, count: int):
        """`count` consecutive step sections as one unit (include/fluid_engine.h)."""
        sid = SECTION_IDS[first_section] if isinstance(first_section, str) else int(first_section)
        self._check(self._lib.fluid_run_section_group(self._h, sid, count))

    SOLVER_JACOBI, SOLVER_RED_BLACK_SOR = 0, 1

    def set_pressure_solver(self, solver: int, omega: float = 1.0):
        """Opt-in red-black SOR for the pressure system (include/fluid_engine.h; not the reference's)."""
        self._check(self._lib.fluid_set_pressure_solver(self._h, solver, omega))
        self.solver, self.sor_omega = int(solver), float(omega)

    def run_surface_diffuse_dispatch(self, is_even_iteration: int):
        self._check(self._lib.fluid_run_surface_diffuse_dispatch(self._h, is_even_iteration))

    def solve_pressure(self, iterations: int):
        """The 12_solve_pressure loop section (fluid_flow_sections.h:300-313)."""
        self.run_section_loop(SEC_12_SOLVE_PRESSURE, iterations)

    def clear_image(self, image_id: int, value):
        """FlowClearColorSection: `value` is a scalar or 4-tuple in the image's own type."""
        dtype, ch = IMAGE_DTYPES.get(image_id, (np.uint32, 1))
        vals = np.zeros(4, dtype=np.float32 if dtype == np.float32 else np.uint32)
        v = np.atleast_1d(np.asarray(value))
        vals[:len(v)] = v.astype(vals.dtype)
        bits = (C.c_uint32 * 4)(*[int(x) for x in vals.view(np.uint32)])
        self._check(self._lib.fluid_clear_image(self._h, image_id, C.byref(bits)))

    def run_pressure_dispatch(self, is_even_iteration: int):
        self._check(self._lib.fluid_run_pressure_dispatch(self._h, is_even_iteration))

    def run_init(self):
        self._check(self._lib.fluid_run_init(self._h))

    def run_step(self):
        self._check(self._lib.fluid_run_step(self._h))

    def sync(self):
        self._check(self._lib.fluid_sync(self._h))

    def set_pressure_iterations(self, iterations: int):
        self._check(self._lib.fluid_set_pressure_iterations(self._h, iterations))
        self.pressure_iterations = int(iterations)

    def set_diffuse_mode(self, mode: int):
        self._check(self._lib.fluid_set_diffuse_mode(self._h, mode))
        self.diffuse_mode = int(mode)

    def set_option(self, option: int, value: int):
        self._check(self._lib.fluid_set_option(self._h, option, value))

    # -- timing -----------------------------------------------------------------------------------
    def enable_timing(self, enabled: bool = True):
        self._check(self._lib.fluid_enable_timing(self._h, 1 if enabled else 0))

    def reset_timing(self):
        self._check(self._lib.fluid_reset_timing(self._h))

    def section_time_ms(self, section):
        sid = SECTION_IDS[section] if isinstance(section, str) else int(section)
        ms, calls = C.c_double(), C.c_uint64()
        self._check(self._lib.fluid_section_time_ms(self._h, sid, C.byref(ms), C.byref(calls)))
        return float(ms.value), int(calls.value)

    def section_times(self):
        return {name: self.section_time_ms(i) for i, name in enumerate(SECTION_NAMES)}

    # -- multi-GPU plumbing ---------------------------------------------------------------------------
    # the loop section in explicit form (fluid_pressure_loop_*)
    LOOP_MASK, LOOP_RHS = 3, 4  # buffer ids of fluid_pressure_loop_plane_ptr besides 0..2
    LOOP_MAX_HALO = 8           # FLUID_LOOP_MAX_HALO

    def pressure_loop_begin(self):
        self._check(self._lib.fluid_pressure_loop_begin(self._h))

    def pressure_loop_max_sweeps(self) -> int:
        n = self._lib.fluid_pressure_loop_max_sweeps(self._h)
        if n < 0:
            self._check(n)
        return n

    def pressure_loop_advance(self, sweeps: int, keep_intermediate: bool = False) -> int:
        written = C.c_int(-1)
        self._check(self._lib.fluid_pressure_loop_advance(self._h, sweeps,
                                                          1 if keep_intermediate else 0,
                                                          C.byref(written)))
        return int(written.value)

    LOOP_PART_EDGES, LOOP_PART_INTERIOR = 1, 2

    def pressure_loop_advance_part(self, keep_intermediate: bool, part: int, interior_begin: int,
                                   interior_end: int, sweeps: int = 2) -> int:
        """One of the two launches of a split pass of two or three sweeps (include/fluid_engine.h)."""
        written = C.c_int(-1)
        lo = max(int(interior_begin), -2 ** 31)
        hi = min(int(interior_end), 2 ** 31 - 1)
        self._check(self._lib.fluid_pressure_loop_advance_part_n(
            self._h, sweeps, 1 if keep_intermediate else 0, part, lo, hi, C.byref(written)))
        return int(written.value)

    def pressure_loop_edge_stream(self) -> int:
        """hipStream_t (as an integer) that EDGES launches use under OPT_EDGE_STREAM."""
        s = C.c_void_p()
        self._check(self._lib.fluid_pressure_loop_edge_stream(self._h, C.byref(s)))
        return int(s.value or 0)

    def pressure_loop_halo_exchanged(self, depth: int, aux_depth: int = 0):
        self._check(self._lib.fluid_pressure_loop_halo_exchanged(self._h, depth, aux_depth))

    def pressure_loop_end(self):
        self._check(self._lib.fluid_pressure_loop_end(self._h))

    def pressure_loop_plane_ptr(self, which: int, plane: int):
        ptr, nbytes = C.c_void_p(), C.c_uint64()
        self._check(self._lib.fluid_pressure_loop_plane_ptr(self._h, which, plane, C.byref(ptr),
                                                            C.byref(nbytes)))
        return int(ptr.value), int(nbytes.value)

    # Z-slab full step (include/fluid_engine.h)
    IMAGE_GHOST_PLANES = 4      # FLUID_IMAGE_GHOST_PLANES
    LEAVER_BYTES = 32           # {float4 data; uint32 index; 3 x pad}
    TOMBSTONE_BITS = 0x7FC0DEAD

    def slab_halo_violation(self) -> bool:
        v = C.c_uint32()
        self._check(self._lib.fluid_slab_status(self._h, C.byref(v)))
        return bool(v.value)

    def notify_image_written(self, image_id: int):
        self._check(self._lib.fluid_notify_image_written(self._h, image_id))

    def pressure_residual(self, image_id: int = PRESSURES_2):
        """(max |r|, sum r^2, water cells) of the pressure system for PRESSURES_1 / _2."""
        m, s, n = C.c_float(0), C.c_double(0), C.c_uint64(0)
        self._check(self._lib.fluid_pressure_residual(self._h, image_id, C.byref(m), C.byref(s),
                                                      C.byref(n)))
        return float(m.value), float(s.value), int(n.value)

    def upload_marching_cubes_tables(self, counts, edge_indices):
        """MarchingCubesBuffers::loadData (marching_cubes.h:30-33): uint[256] and uint[256 * 15]."""
        for buf, arr, n in ((MARCHING_CUBES_COUNTS_BUF, counts, 256), (MARCHING_CUBES_EDGES_BUF, edge_indices, 3840)):
            a = np.ascontiguousarray(arr, dtype=np.uint32).reshape(-1)
            if a.size != n:
                raise FluidEngineError(ERR_SIZE_MISMATCH, f"buffer {buf} holds {n} entries, got {a.size}")
            self._check(self._lib.fluid_upload_buffer(self._h, buf, a.ctypes.data, a.nbytes))

    def extract_surface(self, image_id: int = PARTICLE_DENSITIES_FLOAT_2) -> np.ndarray:
        """The triangles of the reference's marching-cubes surface as an (n, 4, 3) array: three vertices and
        the flat normal per triangle (include/fluid_engine.h: fluid_extract_surface)."""
        n = C.c_uint64(0)
        self._check(self._lib.fluid_extract_surface(self._h, image_id, None, 0, C.byref(n)))
        out = np.empty((int(n.value), 4, 3), np.float32)
        if out.size:
            self._check(self._lib.fluid_extract_surface(self._h, image_id, out.ctypes.data, out.shape[0],
                                                        C.byref(n)))
        return out

    def count_nonfinite(self, image_id: int) -> int:
        """inf / NaN words in the owned planes of a float image (include/fluid_engine.h)."""
        v = C.c_uint64(0)
        self._check(self._lib.fluid_count_nonfinite(self._h, image_id, C.byref(v)))
        return int(v.value)

    def get_stat(self, stat: int) -> int:
        v = C.c_uint64(0)
        self._check(self._lib.fluid_get_stat(self._h, stat, C.byref(v)))
        return int(v.value)

    def notify_ghost_planes_written(self, image_id: int):
        self._check(self._lib.fluid_notify_ghost_planes_written(self._h, image_id))

    def image_plane_ptr(self, image_id: int, plane: int):
        ptr, nbytes = C.c_void_p(), C.c_uint64()
        self._check(self._lib.fluid_image_plane_ptr(self._h, image_id, plane, C.byref(ptr),
                                                    C.byref(nbytes)))
        return int(ptr.value), int(nbytes.value)
