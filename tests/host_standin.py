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

    # ---- velocity sampler halo (include/fluid_engine.h: fluid_sampler_*) -------------------------------
    def set_sampler_halo(self, planes):
        self.sampler_halo = planes

    def halo_violation(self):
        """A tap of 07 beyond the ghost planes received so far read poison (NaN): it shows in the owned
        planes of VELOCITIES_2.  (The scenes of the tests hold no NaN of their own.)"""
        return bool(np.isnan(self._owned(self.st.velocities_2)[..., :3]).any())

    def sampler_reach(self):
        vmax = float(np.abs(self._owned(self.st.velocities_1)[..., 2]).max())
        reach = vmax * abs(float(self.params.time_delta)) * (1.0 + 1e-5)
        if not np.isfinite(vmax) or reach >= self.depth:
            return self.depth
        return min(int(reach) + 2, self.depth)

    def sampler_wide_begin(self, below, above):
        self.wide = (min(below, self.z0), min(above, self.depth - self.z0 - self.dl))

    def sampler_wide_planes(self, first, count):
        # the arrays are global-size: the wide source IS VELOCITIES_1, whose far planes get filled
        assert -self.wide[0] <= first and first + count <= self.dl + self.wide[1]
        return self.image_planes(E.VELOCITIES_1, first, count)

    def run_advect_wide(self, forces):
        self.wide_runs += 1
        self.run_section_group("07_advect", 2) if forces else self.run_section("07_advect")
        assert not self.halo_violation(), "the wide source was still too narrow"

    sampler_halo, wide, wide_runs = 0, (0, 0), 0

    def sync(self):
        pass

    def close(self):
        pass

    # ---- particle hand-over (include/fluid_engine.h: fluid_particles_*) -------------------------------
    LIST_CAPACITY = 4096   # entries per list; tests shrink it to force several rounds

    def _lists(self):
        if not hasattr(self, "_mig"):
            self._mig = [np.zeros((self.LIST_CAPACITY, 8), np.uint32) for _ in range(4)]
            self._count = [0, 0]
        return self._mig

    def migrate_list(self, which):
        return torch.from_numpy(self._lists()[which].reshape(-1).view(np.uint8)), self.LIST_CAPACITY

    def _append(self, direction, words4, index):
        lists = self._lists()
        n = self._count[direction]
        if n >= self.LIST_CAPACITY:
            return False
        lists[direction][n, :4] = words4
        lists[direction][n, 4] = index
        self._count[direction] = n + 1
        return True

    def collect(self, reset):
        self._lists()
        if reset:
            self._count = [0, 0]
        p, words = self.st.particles, self._particle_words()
        real = words[:, 3] != TOMB
        pl = owner_plane(p[:, 2], self.depth)
        left = 0
        for i in np.nonzero(real & ((pl < self.z0) | (pl >= self.z0 + self.dl)))[0]:
            if self._append(0 if pl[i] < self.z0 else 1, words[i].copy(), i):
                p[i] = 0.0
                words[i, 3] = TOMB
            else:
                left += 1
        return tuple(self._count), left

    def adopt_received(self, from_below, from_above):
        lists = self._lists()
        self._count = [0, 0]
        words = self._particle_words()
        for src, n, direction in ((2, from_below, 1), (3, from_above, 0)):
            e = lists[src][:n]
            z = e[:, 2].copy().view(np.float32)
            pl = owner_plane(z, self.depth)
            for j in range(n):
                if self.z0 <= pl[j] < self.z0 + self.dl:
                    words[e[j, 4]] = e[j, :4]
                else:
                    assert self._append(direction, e[j, :4].copy(), e[j, 4])
        return tuple(self._count)

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

    def loop_advance(self, k, sweeps, keep_mid, part=None, interior=None):
        assert k == self.k and part is None   # max_halo 8 on slabs of >= 17 planes would split: not here
        if sweeps >= 2:   # a pass of 2 or 3 sweeps: only the last iterate (and, kept, the one before) stays
            dst = self._other(self.cur, self.cur)
            mid = self._other(self.cur, dst)
            src = self.cur
            if sweeps == 3:
                if not hasattr(self, "_scratch"):
                    self._scratch = np.zeros_like(self.work[0])
                self.work.append(self._scratch)
                self._sweep(src, 3)
                self.work.pop()
                self.work.append(self._scratch)
                self._sweep(3, mid)
                self.work.pop()
            else:
                self._sweep(src, mid)
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


class HostSlabCompute:
    """The pressure loop only, on slab-sized arrays with GW ghost planes per side (the layout of the
    engine's working buffers), the sweep being the CPU oracle over the slab INCLUDING its ghost planes as if
    they were cells: with valid data g planes deep in the ghost region the result is exact g-1 planes deep.
    Split passes store only their own planes, so a part that depended on ghost planes still in flight
    would show.  Loop buffers: 0..2 working pressures, 3 cell types, 4 divergence."""

    GW = 8
    TYPES, DIV = 3, 4

    def __init__(self, params, slab, max_sweeps=2):
        w, h, _ = params.size
        self.params = params
        self.z0, self.dl = slab
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
        self._part_done = None

    def _owned(self, t):
        return t[self.GW:self.GW + self.dl]

    def sweep_fn(self, src, dst):
        sub = self.params.copy()
        sub.fluid_size[2] = self.dl + 2 * self.GW
        # the oracle reads P1 / writes P2 for push constant 1
        oracle_lib().oracle_12_solve_pressure(C.byref(sub), self.arr[E.CELL_TYPES].numpy().ctypes.data,
                                              self.arr[E.DIVERGENCES].numpy().ctypes.data,
                                              src.numpy().ctypes.data, dst.numpy().ctypes.data, 1)

    # ---- what the driver calls --------------------------------------------------------------------------
    def run_section(self, name):
        img = {"12a_clear_pressures_1": E.PRESSURES_1, "12b_clear_pressures_2": E.PRESSURES_2}[name]
        self._owned(self.arr[img])[...] = float(self.params.pressure_air)

    def image_planes(self, image_id, first, count):
        return self.arr[image_id][first + self.GW:first + self.GW + count].view(-1)

    def halo_written(self, image_id):
        pass

    def upload(self, image_id, array):
        self._owned(self.arr[image_id])[...] = torch.from_numpy(np.ascontiguousarray(array))

    def download(self, image_id):
        return self._owned(self.arr[image_id]).numpy().copy()

    def max_halo(self):
        return self.GW

    def loop_max_sweeps(self):
        return self.max_sweeps

    def loop_begin(self, halo):
        self._owned(self.work[0])[...] = self._owned(self.arr[E.PRESSURES_1])
        self.cur, self.prev, self.k = 0, -1, 0
        return [(self.TYPES, halo), (self.DIV, max(halo - 1, 1)), (0, halo)]

    def loop_halo_exchanged(self, halo, first):
        pass

    def _other(self, a, b):
        return next(i for i in range(3) if i not in (a, b))

    def _fused_into(self, sweeps, t_mid, t_dst):
        """`sweeps` (2 or 3) oracle sweeps from the newest iterate: the last one into t_dst, the one before into
        t_mid."""
        src = self.work[self.cur]
        if sweeps == 3:
            first = torch.zeros_like(src)
            self.sweep_fn(src, first)
            src = first
        self.sweep_fn(src, t_mid)
        self.sweep_fn(t_mid, t_dst)

    def _split_pass(self, sweeps, keep_mid, part, interior):
        dst = self._other(self.cur, self.cur)
        mid = self._other(self.cur, dst)
        t_mid, t_dst = self.work[mid].clone(), self.work[dst].clone()
        self._fused_into(sweeps, t_mid, t_dst)
        n = self.dl + 2 * self.GW
        a = min(max(interior[0] + self.GW, 0), n)
        b = min(max(interior[1] + self.GW, a), n)
        for lo, hi in ([(a, b)] if part == "interior" else [(0, a), (b, n)]):
            self.work[dst][lo:hi] = t_dst[lo:hi]
            self.work[mid][lo:hi] = t_mid[lo:hi]
        if self._part_done is None:
            self._part_done = part
            return dst
        assert self._part_done != part
        self._part_done = None
        self.prev = mid if keep_mid else -1
        self.cur = dst
        self.k += sweeps
        return dst

    def loop_advance(self, k, sweeps, keep_mid, part=None, interior=None):
        assert k == self.k
        if part is not None:
            assert sweeps >= 2
            return self._split_pass(sweeps, keep_mid, part, interior)
        assert self._part_done is None
        if sweeps >= 2:
            dst = self._other(self.cur, self.cur)
            mid = self._other(self.cur, dst)
            self._fused_into(sweeps, self.work[mid], self.work[dst])
            self.prev = mid if keep_mid else -1
            self.cur = dst
        else:
            dst = self._other(self.cur, self.prev if self.prev >= 0 else self.cur)
            self.sweep_fn(self.work[self.cur], self.work[dst])
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
                self._owned(self.arr[img])[water] = self._owned(self.work[buf])[water]

    def planes(self, buf, first, count):
        t = self.work[buf] if buf < 3 else self.arr[E.CELL_TYPES if buf == self.TYPES else E.DIVERGENCES]
        return t[first + self.GW:first + self.GW + count].view(-1)

    def sync(self):
        pass

    def __getattr__(self, name):   # the full-step calls: not this stand-in's job
        raise NotImplementedError(f"HostSlabCompute.{name}: pressure loop only")
