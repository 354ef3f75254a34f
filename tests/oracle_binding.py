"""ctypes binding of oracle/liboracle.so — the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
engine package never does.  See oracle/fluid_oracle.h for what the oracle is and is not
("parity unpinned").
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
# FLUID_ORACLE_LIB: another build of the same source, e.g. oracle/liboracle_asan.so (tests/test_oracle_asan.py)
ORACLE_LIB = os.environ.get("FLUID_ORACLE_LIB") or os.path.join(ORACLE_DIR, "liboracle.so")

import sys  # noqa: E402

if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import fluid_amd  # noqa: E402,F401
from fluid_amd.params import FluidParams  # noqa: E402

_lib = None


def build_oracle():
    subprocess.run(["make", "-C", ORACLE_DIR], check=True, stdout=subprocess.PIPE,
                   stderr=subprocess.STDOUT)
    return ORACLE_LIB


def lib():
    global _lib
    if _lib is not None:
        return _lib
    src = os.path.join(ORACLE_DIR, "fluid_oracle.c")
    if (not os.path.exists(ORACLE_LIB)
            or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(ORACLE_LIB))):
        try:
            build_oracle()
        except Exception:
            if not os.path.exists(ORACLE_LIB):
                raise
    L = C.CDLL(ORACLE_LIB)
    vp, u64, u32 = C.c_void_p, C.c_uint64, C.c_uint32
    pp = C.POINTER(FluidParams)
    sig = {
        "oracle_00_init_particles": [pp, vp, u64],
        "oracle_01_update_densities": [pp, vp, u64, vp],
        "oracle_02_update_water": [pp, vp, vp],
        "oracle_03_update_air": [pp, vp],
        "oracle_04_compute_extrapolated_velocities": [pp, vp, vp, vp],
        "oracle_05_set_extrapolated_velocities": [pp, vp, vp, vp, vp],
        "oracle_06_update_cell_types": [pp, vp, vp],
        "oracle_07_advect": [pp, vp, vp, vp],
        "oracle_08_forces": [pp, vp, vp],
        "oracle_09_diffuse": [pp, vp, vp, vp, C.c_int],
        "oracle_10_solids": [pp, vp, vp],
        "oracle_11_compute_divergence": [pp, vp, vp],
        "oracle_12_solve_pressure": [pp, vp, vp, vp, vp, u32],
        "oracle_12_solve_pressure_loop": [pp, vp, vp, vp, vp, u32],
        "oracle_13_fix_divergence": [pp, vp, vp, vp],
        "oracle_14_particles": [pp, vp, vp, u64],
        "oracle_12_sor_iteration": [pp, vp, vp, vp, C.c_float],
        "oracle_12_sor_loop": [pp, vp, vp, vp, vp, C.c_float, u32],
        "oracle_15_update_detailed_densities": [pp, vp, u64, vp],
        "oracle_16_compute_detailed_densities_inertia": [pp, vp, vp],
        "oracle_17_compute_float_densities": [pp, vp, vp],
        "oracle_18_diffuse_float_densities": [pp, vp, vp, vp, u32],
        "oracle_18_diffuse_float_densities_loop": [pp, vp, vp, vp, u32],
        "oracle_31_extract_surface": [pp, vp, vp, vp, vp, u64, C.POINTER(u64)],
    }
    for name, args in sig.items():
        fn = getattr(L, name)
        fn.restype = None
        fn.argtypes = args
    L.oracle_sample_velocity_component.restype = C.c_float
    L.oracle_sample_velocity_component.argtypes = [pp, vp, C.c_float, C.c_float, C.c_float, C.c_int]
    _lib = L
    return L


def _ptr(a: np.ndarray):
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data


class OracleState:
    """All attachments of the hot path as numpy arrays (host layout of include/fluid_engine.h:
    [z][y][x] C order == x fastest) plus one method per section."""

    def __init__(self, params: FluidParams, particle_capacity: int, pressure_iterations: int = 200,
                 diffuse_mode: int = 0, surface_prep: bool = False, surface_diffuse_steps: int = 4):
        self.surface_prep = bool(surface_prep)
        self.surface_diffuse_steps = int(surface_diffuse_steps)
        self.params = params.copy()
        self.capacity = int(particle_capacity)
        self.pressure_iterations = int(pressure_iterations)
        self.diffuse_mode = diffuse_mode
        w, h, d = params.size
        self.shape = (d, h, w)
        self.velocities_1 = np.zeros(self.shape + (4,), np.float32)
        self.velocities_2 = np.zeros(self.shape + (4,), np.float32)
        self.cell_types = np.zeros(self.shape, np.uint8)
        self.new_cell_types = np.zeros(self.shape, np.uint8)
        self.pressures_1 = np.zeros(self.shape, np.float32)
        self.pressures_2 = np.zeros(self.shape, np.float32)
        self.divergences = np.zeros(self.shape, np.float32)
        self.particle_densities = np.zeros(self.shape, np.uint32)
        self.particles = np.zeros((self.capacity, 4), np.float32)
        self._p = C.byref(self.params)
        if self.surface_prep:  # the detailed grid of sections 15-18 (images 8..11)
            r = int(params.detailed_resolution)
            self.detailed_shape = (d * r, h * r, w * r)
            self.detailed_densities = np.zeros(self.detailed_shape, np.uint32)
            self.detailed_densities_inertia = np.zeros(self.detailed_shape, np.uint32)
            self.float_densities_1 = np.zeros(self.detailed_shape, np.float32)
            self.float_densities_2 = np.zeros(self.detailed_shape, np.float32)

    FIELDS = ["velocities_1", "velocities_2", "cell_types", "new_cell_types", "pressures_1",
              "pressures_2", "divergences", "particle_densities", "particles"]

    SURFACE_FIELDS = ["detailed_densities", "detailed_densities_inertia", "float_densities_1",
                      "float_densities_2"]

    def copy(self) -> "OracleState":
        o = OracleState(self.params, self.capacity, self.pressure_iterations, self.diffuse_mode,
                        self.surface_prep, self.surface_diffuse_steps)
        for f in self.FIELDS + (self.SURFACE_FIELDS if self.surface_prep else []):
            getattr(o, f)[...] = getattr(self, f)
        o.sor_omega = self.sor_omega
        return o

    # ---- sections (names = engine section names) ------------------------------------------------
    def run_section(self, name: str):
        L, p = lib(), self._p
        if name == "init_clear_velocities_1":
            self.velocities_1[...] = 0.0
        elif name == "init_clear_cell_types":
            self.cell_types[...] = self.params.cell_type_inactive
        elif name == "00_init_particles":
            L.oracle_00_init_particles(p, _ptr(self.particles), self.capacity)
        elif name == "01a_clear_particle_densities":
            self.particle_densities[...] = 0
        elif name == "01_update_densities":
            L.oracle_01_update_densities(p, _ptr(self.particles), self.capacity,
                                         _ptr(self.particle_densities))
        elif name == "02_update_water":
            L.oracle_02_update_water(p, _ptr(self.particle_densities), _ptr(self.new_cell_types))
        elif name == "03_update_air":
            L.oracle_03_update_air(p, _ptr(self.new_cell_types))
        elif name == "04_compute_extrapolated_velocities":
            L.oracle_04_compute_extrapolated_velocities(p, _ptr(self.cell_types),
                                                        _ptr(self.velocities_1),
                                                        _ptr(self.velocities_2))
        elif name == "05_set_extrapolated_velocities":
            L.oracle_05_set_extrapolated_velocities(p, _ptr(self.new_cell_types),
                                                    _ptr(self.cell_types), _ptr(self.velocities_2),
                                                    _ptr(self.velocities_1))
        elif name == "06_update_cell_types":
            L.oracle_06_update_cell_types(p, _ptr(self.new_cell_types), _ptr(self.cell_types))
        elif name == "07_advect":
            L.oracle_07_advect(p, _ptr(self.cell_types), _ptr(self.velocities_1),
                               _ptr(self.velocities_2))
        elif name == "08_forces":
            L.oracle_08_forces(p, _ptr(self.cell_types), _ptr(self.velocities_2))
        elif name == "09_diffuse":
            L.oracle_09_diffuse(p, _ptr(self.cell_types), _ptr(self.velocities_2),
                                _ptr(self.velocities_1), self.diffuse_mode)
        elif name == "10_solids":
            L.oracle_10_solids(p, _ptr(self.cell_types), _ptr(self.velocities_1))
        elif name == "11_compute_divergence":
            L.oracle_11_compute_divergence(p, _ptr(self.velocities_1), _ptr(self.divergences))
        elif name == "12a_clear_pressures_1":
            self.pressures_1[...] = self.params.pressure_air
        elif name == "12b_clear_pressures_2":
            self.pressures_2[...] = self.params.pressure_air
        elif name == "13_fix_divergence":
            L.oracle_13_fix_divergence(p, _ptr(self.cell_types), _ptr(self.pressures_2),
                                       _ptr(self.velocities_1))
        elif name == "14_particles":
            L.oracle_14_particles(p, _ptr(self.velocities_1), _ptr(self.particles), self.capacity)
        elif name == "init_clear_detailed_densities_inertia":
            self.detailed_densities_inertia[...] = 0
        elif name == "14a_clear_detailed_densities":
            self.detailed_densities[...] = 0
        elif name == "15_update_detailed_densities":
            L.oracle_15_update_detailed_densities(p, _ptr(self.particles), self.capacity,
                                                  _ptr(self.detailed_densities))
        elif name == "16_compute_detailed_densities_inertia":
            L.oracle_16_compute_detailed_densities_inertia(p, _ptr(self.detailed_densities),
                                                           _ptr(self.detailed_densities_inertia))
        elif name == "17_compute_float_densities":
            self._surface_dispatch = 0
            L.oracle_17_compute_float_densities(p, _ptr(self.detailed_densities_inertia),
                                                _ptr(self.float_densities_1))
        elif name == "18_diffuse_float_densities":  # one dispatch; the counter restarts at 17
            k = getattr(self, "_surface_dispatch", 0)
            self._surface_dispatch = k + 1
            L.oracle_18_diffuse_float_densities(p, _ptr(self.cell_types), _ptr(self.float_densities_1),
                                                _ptr(self.float_densities_2), 1 if k % 2 == 0 else 0)
        else:
            raise KeyError(name)

    def diffuse_float_densities(self, iterations: int):
        self._surface_dispatch = iterations
        lib().oracle_18_diffuse_float_densities_loop(self._p, _ptr(self.cell_types),
                                                     _ptr(self.float_densities_1),
                                                     _ptr(self.float_densities_2), iterations)

    SURFACE_ORDER = ["14a_clear_detailed_densities", "15_update_detailed_densities",
                     "16_compute_detailed_densities_inertia", "17_compute_float_densities"]

    def run_surface_prep(self):
        for s in self.SURFACE_ORDER:
            self.run_section(s)
        self.diffuse_float_densities(self.surface_diffuse_steps)

    def pressure_dispatch(self, is_even_iteration: int):
        lib().oracle_12_solve_pressure(self._p, _ptr(self.cell_types), _ptr(self.divergences),
                                       _ptr(self.pressures_1), _ptr(self.pressures_2),
                                       is_even_iteration)

    sor_omega = None   # set to a float: the opt-in red-black SOR solver replaces the Jacobi loop

    def solve_pressure(self, iterations: int):
        if self.sor_omega is not None:
            lib().oracle_12_sor_loop(self._p, _ptr(self.cell_types), _ptr(self.divergences),
                                     _ptr(self.pressures_1), _ptr(self.pressures_2),
                                     float(self.sor_omega), iterations)
            return
        lib().oracle_12_solve_pressure_loop(self._p, _ptr(self.cell_types),
                                            _ptr(self.divergences), _ptr(self.pressures_1),
                                            _ptr(self.pressures_2), iterations)

    INIT_ORDER = ["init_clear_velocities_1", "init_clear_cell_types", "00_init_particles"]
    STEP_BEFORE_12 = ["01a_clear_particle_densities", "01_update_densities", "02_update_water",
                      "03_update_air", "04_compute_extrapolated_velocities",
                      "05_set_extrapolated_velocities", "06_update_cell_types", "07_advect",
                      "08_forces", "09_diffuse", "10_solids", "11_compute_divergence",
                      "12a_clear_pressures_1", "12b_clear_pressures_2"]
    STEP_AFTER_12 = ["13_fix_divergence", "14_particles"]

    def run_init(self):
        for s in self.INIT_ORDER:
            self.run_section(s)
            if s == "init_clear_cell_types" and self.surface_prep:
                self.run_section("init_clear_detailed_densities_inertia")

    def run_step(self):
        for s in self.STEP_BEFORE_12:
            self.run_section(s)
        self.solve_pressure(self.pressure_iterations)
        for s in self.STEP_AFTER_12:
            self.run_section(s)
        if self.surface_prep:
            self.run_surface_prep()

    def extract_surface(self, density: np.ndarray, counts: np.ndarray, edge_indices: np.ndarray) -> np.ndarray:
        """oracle_31_extract_surface: (n, 4, 3) — three vertices and the flat normal per triangle, cells in
        vertex-index order."""
        counts = np.ascontiguousarray(counts, np.uint32)
        edge_indices = np.ascontiguousarray(edge_indices, np.uint32)
        density = np.ascontiguousarray(density, np.float32)
        n = C.c_uint64(0)
        lib().oracle_31_extract_surface(self._p, _ptr(density), _ptr(counts), _ptr(edge_indices), None, 0,
                                        C.byref(n))
        out = np.empty((int(n.value), 4, 3), np.float32)
        lib().oracle_31_extract_surface(self._p, _ptr(density), _ptr(counts), _ptr(edge_indices), _ptr(out),
                                        out.shape[0], C.byref(n))
        return out

    def sample(self, field: np.ndarray, px: float, py: float, pz: float, comp: int) -> float:
        return float(lib().oracle_sample_velocity_component(self._p, _ptr(field), px, py, pz, comp))
