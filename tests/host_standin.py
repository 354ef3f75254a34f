"""CPU stand-in for the per-slab compute backend of slab.SlabSimulation / SlabPressureSolver, used by
the multi-process gloo tests.  Every rank holds GLOBAL-size arrays of which it trusts only its own
planes plus the ghost planes it has received; a section is the CPU oracle run on the global arrays,
after which everything outside the owned planes of the images it wrote is POISONED (NaN / invalid
type), so a halo exchange the driver forgets — or a sample that reaches past the ghost planes —
shows up as a mismatch instead of passing by luck.  Test infrastructure, not a product path."""
import ctypes as C

import numpy as np
import torch

from fluid_amd import engine as E
from oracle_binding import OracleState, lib as oracle_lib

TOMB = np.uint32(E.FluidEngine.TOMBSTONE_BITS)
POISON_TYPE = 77

WRITES = {
    "init_clear_velocities_1": ["velocities_1"], "init_clear_cell_types": ["cell_types"],
    "00_init_particles": [], "01a_clear_particle_densities": ["particle_densities"],
    "01_update_densities": ["particle_densities"], "02_update_water": ["new_cell_types"],
    "03_update_air": ["new_cell_types"], "04_compute_extrapolated_velocities": ["velocities_2"],
    "05_set_extrapolated_velocities": ["velocities_1"], "06_update_cell_types": [],
    "07_advect": ["velocities_2"], "08_forces": ["velocities_2"], "09_diffuse": ["velocities_1"],
    "10_solids": ["velocities_1"], "11_compute_divergence": ["divergences"],
    "12a_clear_pressures_1": ["pressures_1"], "12b_clear_pressures_2": ["pressures_2"],
    "13_fix_divergence": ["velocities_1"], "14_particles": [],
}
IMAGE_FIELD = {E.VELOCITIES_1: "velocities_1", E.VELOCITIES_2: "velocities_2",
               E.CELL_TYPES: "cell_types", E.NEW_CELL_TYPES: "new_cell_types",
               E.PRESSURES_1: "pressures_1", E.PRESSURES_2: "pressures_2",
               E.DIVERGENCES: "divergences", E.PARTICLE_DENSITIES_IMG: "particle_densities"}


def owner_plane(z, depth):
    """particle_owner_plane of csrc/device_common.h."""
    with np.errstate(invalid="ignore"):
        pl = np.where(z > 0, np.minimum(np.trunc(np.where(np.isfinite(z), z, 0)), depth - 1), 0)
        pl = np.where(z >= depth, depth - 1, pl)
    return pl.astype(np.int64)


class HostGlobalCompute:
    IMAGE_GHOST = E.FluidEngine.IMAGE_GHOST_PLANES
    TYPES, DIV = 3, 4

    def __init__(self, params, slab, particle_capacity=0, iterations=4, max_sweeps=2):
        self.params = params
        self.z0, self.dl = slab
        self.depth = params.size[2]
        self.st = OracleState(params, particle_capacity, iterations)
        self.capacity = particle_capacity
        self.max_sweeps = max_sweeps
        self.work = [np.zeros(self.st.shape, np.float32) for _ in range(3)]
        self.cur, self.prev, self.k = 0, -1, 0
        for f in IMAGE_FIELD.values():
            self._poison(f)

    # ---- bookkeeping ---------------------------------------------------------------------------
    def _poison(self, field):
        a = getattr(self.st, field)
        bad = np.ones(self.depth, bool)
        bad[self.z0:self.z0 + self.dl] = False
        a[bad] = POISON_TYPE if a.dtype == np.uint8 else (np.nan if a.dtype == np.float32 else 0xDEAD)

    def _owned(self, a):
        return a[self.z0:self.z0 + self.dl]

    def _particle_words(self):
        return self.st.particles.view(np.uint32)

    def _bury_foreign(self):
        p = self.st.particles
        pl = owner_plane(p[:, 2], self.depth)
        foreign = (pl < self.z0) | (pl >= self.z0 + self.dl)
        p[foreign] = 0.0
        self._particle_words()[foreign, 3] = TOMB

    # ---- sections ---------------------------------------------------------------------------------
    def run_section(self, name):
        with np.errstate(all="ignore"):
            self.st.run_section(name)
        if name == "00_init_particles":
            self._bury_foreign()
        for f in WRITES[name]:
            self._poison(f)

    STEP_LIST = list(OracleState.STEP_BEFORE_12)

    def run_section_group(self, first, count):
        """The sections of the slice back to back, poisoning afterwards: inside a grouped pass the
        engine computes from ghost planes what the list would have exchanged in between."""
        i = self.STEP_LIST.index(first)
        names = self.STEP_LIST[i:i + count]
        with np.errstate(all="ignore"):
            for name in names:
                self.st.run_section(name)
        for name in names:
            for f in WRITES[name]:
                self._poison(f)

    def set_diffuse_mode(self, mode):
        self.st.diffuse_mode = mode

    def upload(self, image_id, array):
        self._owned(getattr(self.st, IMAGE_FIELD[image_id]))[...] = array

    def download(self, image_id):
        return self._owned(getattr(self.st, IMAGE_FIELD[image_id])).copy()

    def upload_particles(self, particles):
        self.st.particles[...] = particles
        self._bury_foreign()

    def download_particles(self):
        return self.st.particles.copy()

    def image_planes(self, image_id, first, count):
        a = getattr(self.st, IMAGE_FIELD[image_id])
        return torch.from_numpy(a[self.z0 + first:self.z0 + first + count]).view(-1)

    def plane(self, image_id, local_z):
        return self.image_planes(image_id, local_z, 1)

    def halo_written(self, image_id):
        pass

    def halo_violation(self):
        return False  # a sample past the ghost planes reads poison and fails the comparison instead

    def sync(self):
        pass

    def close(self):
        pass

    # ---- particle migration ---------------------------------------------------------------------------
    def collect_leavers(self):
        p = self.st.particles
        words = self._particle_words()
        real = words[:, 3] != TOMB
        pl = owner_plane(p[:, 2], self.depth)
        gone = real & ((pl < self.z0) | (pl >= self.z0 + self.dl))
        idx = np.nonzero(gone)[0]
        entries = np.zeros((len(idx), 8), np.uint32)
        entries[:, :4] = words[idx]
        entries[:, 4] = idx
        p[idx] = 0.0
        words[idx, 3] = TOMB
        return torch.from_numpy(entries.view(np.uint8).reshape(-1)), len(idx)

    def adopt(self, entries, count):
        e = entries.cpu().numpy()[:count * 32].view(np.uint32).reshape(count, 8)
        z = e[:, 2].copy().view(np.float32)
        pl = owner_plane(z, self.depth)
        mine = (pl >= self.z0) & (pl < self.z0 + self.dl)
        self._particle_words()[e[mine, 4]] = e[mine, :4]

    # ---- the pressure loop (same interface as slab.HostSlabCompute, on global arrays) --------------
    def clear_pressures(self):
        self.run_section("12a_clear_pressures_1")
        self.run_section("12b_clear_pressures_2")

    def max_halo(self):
        return 8

    def loop_begin(self, halo):
        self.work[0][...] = np.nan
        self._owned(self.work[0])[...] = self._owned(self.st.pressures_1)
        self.cur, self.prev, self.k = 0, -1, 0
        return [(self.TYPES, halo), (self.DIV, max(halo - 1, 1)), (0, halo)]

    def loop_halo_exchanged(self, halo, first):
        pass

    def loop_max_sweeps(self):
        return self.max_sweeps

    def _other(self, a, b):
        return next(i for i in range(3) if i not in (a, b))

    def _sweep(self, src, dst):
        with np.errstate(all="ignore"):
            oracle_lib().oracle_12_solve_pressure(
                C.byref(self.st.params), self.st.cell_types.ctypes.data,
                self.st.divergences.ctypes.data, self.work[src].ctypes.data,
                self.work[dst].ctypes.data, 1)

    def loop_advance(self, k, sweeps, keep_mid):
        assert k == self.k
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
        if self.k:
            water = self._owned(self.st.cell_types) == int(self.params.cell_type_water)
            even, odd = (self.cur, self.prev) if self.k % 2 == 0 else (self.prev, self.cur)
            for field, buf in (("pressures_1", even), ("pressures_2", odd)):
                if buf >= 0:
                    self._owned(getattr(self.st, field))[water] = self._owned(self.work[buf])[water]
        self._poison("pressures_1")
        self._poison("pressures_2")

    def planes(self, buf, first, count):
        a = self.work[buf] if buf < 3 else (self.st.cell_types if buf == self.TYPES
                                            else self.st.divergences)
        return torch.from_numpy(a[self.z0 + first:self.z0 + first + count]).view(-1)
