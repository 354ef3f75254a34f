"""Known-answer tests that pin the CPU oracle to the shader source (SURVEY.md §8c, K1…K10).

The reference ships no tests or golden vectors, so every expected value here is derived by hand
from the GLSL text cited next to it (paths under /root/reference/shaders_fluid).  CPU only.
"""
import numpy as np
import pytest

from oracle_binding import OracleState
from fluid_amd.params import (CELL_AIR, CELL_INACTIVE, CELL_SOLID, CELL_WATER, default_params)

F = np.float32


def make_state(n=8, capacity=0, iters=4, **kw):
    p = default_params(n, n, n, capacity)
    for k, v in kw.items():
        setattr(p, k, v)
    return OracleState(p, capacity, iters)


# ---- K1: 00_init_particles/init_particles.comp:27-49 with simulation_constants.h:48-50 ----------
def test_k1_init_particles_reference_defaults():
    cap = 1000100
    s = OracleState(default_params(20, 20, 20, cap), cap)
    s.run_section("00_init_particles")
    assert s.particles[0].tolist() == [5.0, 2.0, 1.5, 1.0]
    last = s.particles[999999]
    exp = [F(5) + (F(99) / F(100)) * F(10), F(2) + (F(99) / F(100)) * F(10),
           F(1.5) + (F(99) / F(100)) * F(2), F(1)]
    assert last.tolist() == [float(v) for v in exp]
    np.testing.assert_allclose(last[:3], [14.9, 11.9, 3.48], rtol=0, atol=2e-6)
    # x is the fastest index (getPos :27-34): particle 1 is one lattice step along x
    assert s.particles[1].tolist() == [float(F(5) + (F(1) / F(100)) * F(10)), 2.0, 1.5, 1.0]
    assert s.particles[100].tolist() == [5.0, float(F(2) + (F(1) / F(100)) * F(10)), 1.5, 1.0]
    # invocations past the cube volume write an inactive particle (:46-48)
    assert np.all(s.particles[1000000:] == 0.0)


# ---- 01_update_densities/update_densities.comp:29-36 -------------------------------------------
def test_01_truncation_inactive_and_out_of_bounds():
    s = make_state(8, capacity=8)
    s.particles[:] = [
        [1.5, 2.5, 3.5, 1.0],     # cell (1,2,3)
        [1.99, 2.01, 3.0, 1.0],   # same cell
        [-0.5, 0.2, 0.9, 1.0],    # ivec3() truncates toward zero: (-0.5 -> 0)  => cell (0,0,0)
        [-1.0, 0.0, 0.0, 1.0],    # x = -1: out of the image, atomic dropped
        [8.0, 1.0, 1.0, 1.0],     # x = W: dropped
        [7.999, 7.999, 7.999, 1.0],  # last cell
        [3.0, 3.0, 3.0, 0.0],     # w != active_particle_w: ignored (:33)
        [np.nan, 1.0, 1.0, 1.0],  # defined as dropped
    ]
    s.run_section("01_update_densities")
    exp = np.zeros((8, 8, 8), np.uint32)
    exp[3, 2, 1] = 2
    exp[0, 0, 0] = 1
    exp[7, 7, 7] = 1
    np.testing.assert_array_equal(s.particle_densities, exp)
    # the section accumulates (imageAtomicAdd): a second run doubles the counts
    s.run_section("01_update_densities")
    np.testing.assert_array_equal(s.particle_densities, 2 * exp)


# ---- 02 / 03 -----------------------------------------------------------------------------------
def test_02_03_water_air_solid_classification():
    s = make_state(8)
    s.particle_densities[4, 4, 4] = 3
    s.new_cell_types[...] = 77  # 02 overwrites every cell (update_water.comp:33)
    s.run_section("02_update_water")
    assert s.new_cell_types[4, 4, 4] == CELL_WATER
    assert np.count_nonzero(s.new_cell_types == CELL_WATER) == 1
    assert np.count_nonzero(s.new_cell_types == CELL_INACTIVE) == 8 ** 3 - 1
    s.run_section("03_update_air")
    t = s.new_cell_types
    # faces are SOLID (update_active.comp:50-51)
    for ax in range(3):
        assert np.all(np.take(t, 0, axis=ax) == CELL_SOLID)
        assert np.all(np.take(t, 7, axis=ax) == CELL_SOLID)
    # the six face neighbours of the water cell become AIR (:54-63), nothing else
    air = {(4, 4, 5), (4, 4, 3), (4, 5, 4), (4, 3, 4), (5, 4, 4), (3, 4, 4)}
    got = set(map(tuple, np.argwhere(t == CELL_AIR)))
    assert got == air
    assert t[4, 4, 4] == CELL_WATER
    interior = t[1:7, 1:7, 1:7]
    assert np.count_nonzero(interior == CELL_INACTIVE) == 6 ** 3 - 7


def test_03_border_water_is_solid_first():
    """a particle in a border cell — the border cell becomes SOLID and never counts
    as water for its interior neighbour (the order-independent resolution of the reference race)."""
    s = make_state(8)
    s.new_cell_types[...] = CELL_INACTIVE
    s.new_cell_types[0, 3, 3] = CELL_WATER  # z = 0 face
    s.new_cell_types[3, 3, 3] = CELL_WATER
    s.run_section("03_update_air")
    assert s.new_cell_types[0, 3, 3] == CELL_SOLID
    assert s.new_cell_types[1, 3, 3] == CELL_INACTIVE  # not AIR: its only "water" nbr is on the border
    assert s.new_cell_types[2, 3, 3] == CELL_AIR


# ---- K2: sampler (advect.comp:52-56,70-73; fluid_flow_sections.h:95) ------------------------------
def test_k2_sampler_fixed_points_and_clamp():
    rng = np.random.default_rng(1)
    s = make_state(8)
    s.velocities_1[...] = rng.standard_normal(s.velocities_1.shape).astype(F)
    v = s.velocities_1
    for (x, y, z) in [(0, 0, 0), (3, 4, 5), (7, 7, 7), (1, 6, 2)]:
        for c in range(3):
            pos = [x + 0.5, y + 0.5, z + 0.5]
            pos[c] -= 0.5  # the face the component lives on
            assert s.sample(v, *pos, c) == float(v[z, y, x, c])
    # clamp to edge: far outside the grid the sample is the nearest edge texel
    assert s.sample(v, -5.0, 3.5, 2.5, 0) == float(v[2, 3, 0, 0])
    assert s.sample(v, 3.0, 100.0, 2.5, 0) == float(v[2, 7, 3, 0])
    assert s.sample(v, 3.0, 3.5, -9.0, 0) == float(v[0, 3, 3, 0])
    # halfway between two texels along x: the mean (weights 0.5/0.5 are exact in fp32)
    exp = F(0.5) * v[5, 4, 2, 0] + F(0.5) * v[5, 4, 3, 0]
    assert s.sample(v, 2.5, 4.5, 5.5, 0) == float(exp)
    # component y at a cell centre = mean of the two y faces of the cell
    exp = F(0.5) * v[5, 4, 3, 1] + F(0.5) * v[5, 5, 3, 1]
    assert s.sample(v, 3.5, 4.5, 5.5, 1) == float(exp)


def test_sampler_constant_field_is_reproduced():
    s = make_state(8)
    s.velocities_1[..., 0] = 3.0
    s.velocities_1[..., 1] = -2.0
    s.velocities_1[..., 2] = 0.5
    rng = np.random.default_rng(2)
    for _ in range(50):
        pos = rng.uniform(-2, 10, 3)
        for c, val in enumerate([3.0, -2.0, 0.5]):
            assert s.sample(s.velocities_1, *pos, c) == val


# ---- 07_advect --------------------------------------------------------------------------------------
def test_07_advect_zero_velocity_and_dry_cells_keep_values():
    rng = np.random.default_rng(3)
    s = make_state(8)
    s.cell_types[...] = CELL_AIR
    s.velocities_1[...] = rng.standard_normal(s.velocities_1.shape).astype(F)
    s.run_section("07_advect")
    # no water anywhere: every component is kept (advect.comp:68,79); w = 0 (:96)
    np.testing.assert_array_equal(s.velocities_2[..., :3], s.velocities_1[..., :3])
    assert np.all(s.velocities_2[..., 3] == 0)


def test_07_advect_uniform_flow_is_a_fixed_point():
    s = make_state(8)
    s.cell_types[...] = CELL_WATER
    s.velocities_1[..., 0] = 1.25
    s.velocities_1[..., 1] = -0.5
    s.velocities_1[..., 2] = 2.0
    s.run_section("07_advect")
    np.testing.assert_array_equal(s.velocities_2[..., :3], s.velocities_1[..., :3])


def test_07_advect_tests_the_plus_neighbour():
    """component c of cell i is advected iff i[c] != 0 and (cell i is water or the
    cell at i + e_c is water) — advect.comp:65-68 subtracts move = -1."""
    s = make_state(8)
    s.cell_types[...] = CELL_AIR
    s.cell_types[4, 4, 4] = CELL_WATER
    # a field that advection changes: v.x = x (sampling upstream changes the value)
    xs = np.arange(8, dtype=F)
    s.velocities_1[..., 0] = xs[None, None, :]
    s.run_section("07_advect")
    changed = np.argwhere(s.velocities_2[..., 0] != s.velocities_1[..., 0])
    # x component changes only in the water cell itself and in the cell at x-1 (whose +x nbr is water)
    assert set(map(tuple, changed)) == {(4, 4, 4), (4, 4, 3)}
    # backtrace: q = (x, y+.5, z+.5), S_x(q) = x, new = S_x(q - x*dt) = x - x*dt on this linear field
    got = s.velocities_2[4, 4, 4, 0]
    assert abs(got - (4.0 - 4.0 * 0.01)) < 1e-5


# ---- K3: 08_forces/forces.comp:39-54 ----------------------------------------------------------------
def test_k3_forces_gravity_and_fountain():
    s = make_state(8)
    fx, fy, fz = s.params.fountain_position[:]
    assert (fx, fy, fz) == (4, 6, 4)  # simulation_constants.h:85
    s.cell_types[...] = CELL_AIR
    s.cell_types[3, 3, 3] = CELL_WATER
    s.cell_types[fz, fy, fx] = CELL_WATER
    s.cell_types[5, 0, 5] = CELL_WATER  # y == 0: gravity never applies (:39)
    rng = np.random.default_rng(4)
    s.velocities_2[...] = rng.standard_normal(s.velocities_2.shape).astype(F)
    before = s.velocities_2.copy()
    s.run_section("08_forces")
    g = F(0.01) * F(10.0)
    exp = before.copy()
    # water cell and the cell below it in +y (whose -y neighbour is water) get gravity
    for (z, y, x) in [(3, 3, 3), (3, 4, 3), (5, 1, 5)]:
        exp[z, y, x, 1] = before[z, y, x, 1] + g
    exp[fz, fy + 1, fx, 1] = before[fz, fy + 1, fx, 1] + g
    exp[fz, fy, fx, 1] = before[fz, fy, fx, 1] + F(0.01) * (F(10.0) + F(-3000.0))
    np.testing.assert_array_equal(s.velocities_2, exp)
    assert abs(float(F(0.01) * (F(10.0) + F(-3000.0))) + 29.9) < 1e-5


# ---- K4: 10_solids/solids.comp:30-76 ----------------------------------------------------------------
@pytest.mark.parametrize("self_solid", [False, True])
@pytest.mark.parametrize("nbr_solid", [False, True])
@pytest.mark.parametrize("v", [-1.0, -0.01, -0.005, 0.0, 0.005, 0.01, 1.0])
def test_k4_solids_truth_table(self_solid, nbr_solid, v):
    s = make_state(8)
    s.cell_types[...] = CELL_AIR
    if self_solid:
        s.cell_types[4, 4, 4] = CELL_SOLID
    if nbr_solid:
        s.cell_types[4, 4, 3] = CELL_SOLID  # the -x neighbour
    s.velocities_1[4, 4, 4, 0] = v
    s.run_section("10_solids")
    r = F(0.01)
    e = F(v)
    if self_solid and e > -r:   # :32-33
        e = -r
    if nbr_solid and e < r:     # :50-51
        e = r
    assert s.velocities_1[4, 4, 4, 0] == e
    assert s.velocities_1[4, 4, 4, 3] == 1.0  # :76 stores w = 1


# ---- K5: 11_compute_divergence/compute_divergence.comp:21 ---------------------------------------------
def test_k5_divergence_linear_field_and_oob():
    s = make_state(8)
    s.velocities_1[..., 0] = np.arange(8, dtype=F)[None, None, :]
    s.run_section("11_compute_divergence")
    assert np.all(s.divergences[:, :, :7] == 1.0)
    assert np.all(s.divergences[:, :, 7] == -7.0)  # OOB load at x = W returns 0


# ---- K6: 12_solve_pressure/pressure.comp:41-76 --------------------------------------------------------
def test_k6_jacobi_isolated_cell_and_solid_count():
    s = make_state(8)
    s.cell_types[...] = CELL_AIR
    s.cell_types[4, 4, 4] = CELL_WATER
    s.divergences[4, 4, 4] = 0.03
    s.pressures_1[...] = 1.0
    s.pressures_2[...] = 1.0
    s.solve_pressure(1)
    # s = div*rho*dx/dt - 6*p_air ; p = -s/6
    sv = ((F(0.03) * F(1)) * F(1)) / F(0.01)
    for _ in range(6):
        sv = sv - F(1)
    assert s.pressures_2[4, 4, 4] == -sv / F(6)
    assert abs(s.pressures_2[4, 4, 4] - (1 - 0.03 / (6 * 0.01))) < 1e-6
    assert np.count_nonzero(s.pressures_2 != 1.0) == 1   # non-water cells are never written
    assert np.all(s.pressures_1 == 1.0)                  # dispatch 0 reads P1, writes P2
    # the value is a fixed point for further sweeps
    s.solve_pressure(5)
    assert s.pressures_1[4, 4, 4] == s.pressures_2[4, 4, 4] == -sv / F(6)

    # k solid neighbours: divide by 6 - k and skip their contribution
    for k in range(1, 6):
        s = make_state(8)
        s.cell_types[...] = CELL_AIR
        s.cell_types[4, 4, 4] = CELL_WATER
        nbrs = [(4, 4, 5), (4, 5, 4), (5, 4, 4), (4, 4, 3), (4, 3, 4), (3, 4, 4)]
        for n in nbrs[:k]:
            s.cell_types[n] = CELL_SOLID
        s.divergences[4, 4, 4] = -0.02
        s.pressures_1[...] = 1.0
        s.pressures_2[...] = 1.0
        s.solve_pressure(1)
        sv = ((F(-0.02) * F(1)) * F(1)) / F(0.01)
        for _ in range(6 - k):
            sv = sv - F(1)
        assert s.pressures_2[4, 4, 4] == -sv / F(6 - k)


def test_k6_jacobi_ping_pong_parity_and_water_neighbours():
    """F2: dispatch k reads P1 / writes P2 iff k is even; after an even N, P2 holds the (N-1)-sweep
    iterate and P1 the N-sweep iterate."""
    rng = np.random.default_rng(6)
    s = make_state(8)
    s.cell_types[...] = CELL_AIR
    s.cell_types[2:6, 2:6, 2:6] = CELL_WATER
    s.divergences[...] = rng.uniform(-1, 1, s.divergences.shape).astype(F)
    s.pressures_1[...] = 1.0
    s.pressures_2[...] = 1.0

    # manual Jacobi with explicit buffers, following pressure.comp:52-62 literally in numpy fp32
    def sweep(pin):
        out = pin.copy()
        t = s.cell_types
        for z, y, x in np.argwhere(t == CELL_WATER):
            sv = ((s.divergences[z, y, x] * F(1)) * F(1)) / F(0.01)
            aii = 0
            for dx, dy, dz in [(1, 0, 0), (0, 1, 0), (0, 0, 1), (-1, 0, 0), (0, -1, 0), (0, 0, -1)]:
                tt = t[z + dz, y + dy, x + dx]
                if tt != CELL_SOLID:
                    sv = sv - (pin[z + dz, y + dy, x + dx] if tt == CELL_WATER else F(1))
                    aii += 1
            out[z, y, x] = -sv / F(aii)
        return out

    it = [np.ones_like(s.pressures_1)]
    for _ in range(4):
        it.append(sweep(it[-1]))
    s.solve_pressure(4)
    np.testing.assert_array_equal(s.pressures_2, it[3])
    np.testing.assert_array_equal(s.pressures_1, it[4])
    # odd count: the last dispatch (k=2, even) writes P2
    s.pressures_1[...] = 1.0
    s.pressures_2[...] = 1.0
    s.solve_pressure(3)
    np.testing.assert_array_equal(s.pressures_2, it[3])
    np.testing.assert_array_equal(s.pressures_1, it[2])
    # explicit push constant: anything but 1 reads P2 and writes P1 (pressure.comp:71-75)
    s.pressures_1[...] = 1.0
    s.pressures_2[...] = it[1]
    s.pressure_dispatch(0)
    np.testing.assert_array_equal(s.pressures_1, it[2])


def test_jacobi_walled_in_cell_divides_by_zero():
    """aii == 0 is unguarded in the shader (pressure.comp:62): -s/0."""
    s = make_state(8)
    s.cell_types[...] = CELL_SOLID
    s.cell_types[4, 4, 4] = CELL_WATER
    s.divergences[4, 4, 4] = 0.5
    s.pressures_1[...] = 1.0
    s.pressures_2[...] = 1.0
    with np.errstate(all="ignore"):
        s.solve_pressure(1)
    assert np.isinf(s.pressures_2[4, 4, 4]) and s.pressures_2[4, 4, 4] < 0


# ---- K7: 13_fix_divergence/fix_divergence.comp:41-71 ---------------------------------------------------
def test_k7_fix_divergence_faces():
    rng = np.random.default_rng(7)
    s = make_state(8)
    s.cell_types[...] = CELL_AIR
    s.cell_types[4, 4, 4] = CELL_WATER
    s.cell_types[4, 4, 5] = CELL_SOLID   # +x neighbour of the water cell is solid
    s.pressures_2[...] = rng.uniform(0, 2, s.pressures_2.shape).astype(F)
    s.velocities_1[...] = rng.standard_normal(s.velocities_1.shape).astype(F)
    before = s.velocities_1.copy()
    p = s.pressures_2
    s.run_section("13_fix_divergence")
    k = (F(0.01) / F(1)) / F(1)
    exp = before.copy()
    exp[..., 3] = 0.0
    # faces of the water cell itself (its -x,-y,-z faces; all neighbours there are AIR)
    exp[4, 4, 4, 0] = before[4, 4, 4, 0] - k * (p[4, 4, 4] - p[4, 4, 3])
    exp[4, 4, 4, 1] = before[4, 4, 4, 1] - k * (p[4, 4, 4] - p[4, 3, 4])
    exp[4, 4, 4, 2] = before[4, 4, 4, 2] - k * (p[4, 4, 4] - p[3, 4, 4])
    # faces owned by the +y / +z neighbours (their -y / -z side is the water cell)
    exp[4, 5, 4, 1] = before[4, 5, 4, 1] - k * (p[4, 5, 4] - p[4, 4, 4])
    exp[5, 4, 4, 2] = before[5, 4, 4, 2] - k * (p[5, 4, 4] - p[4, 4, 4])
    # the +x neighbour is SOLID: its face is not corrected (:48).  v - k*0 leaves every other face
    # bit-identical.
    np.testing.assert_array_equal(s.velocities_1, exp)


def test_k7_no_correction_at_domain_face():
    s = make_state(8)
    s.cell_types[...] = CELL_WATER
    s.pressures_2[...] = np.random.default_rng(8).uniform(0, 2, s.pressures_2.shape).astype(F)
    s.velocities_1[...] = 1.0
    s.run_section("13_fix_divergence")
    # pos[c] == 0: `pos[comp_i] != -1` fails after the decrement (:43-46)
    assert np.all(s.velocities_1[:, :, 0, 0] == 1.0)
    assert np.all(s.velocities_1[:, 0, :, 1] == 1.0)
    assert np.all(s.velocities_1[0, :, :, 2] == 1.0)
    assert np.any(s.velocities_1[:, :, 1:, 0] != 1.0)


# ---- K8: 05_set_extrapolated_velocities/extrapolate_velocities.comp:48-84 -------------------------------
@pytest.mark.parametrize("old_self,old_nbr,new_self,new_nbr", [
    (a, b, c, d) for a in (CELL_INACTIVE, CELL_AIR, CELL_WATER, CELL_SOLID)
    for b in (CELL_INACTIVE, CELL_WATER) for c in (CELL_INACTIVE, CELL_AIR, CELL_SOLID)
    for d in (CELL_INACTIVE, CELL_WATER)])
def test_k8_velocity_state_table(old_self, old_nbr, new_self, new_nbr):
    s = make_state(8)
    s.cell_types[...] = CELL_INACTIVE
    s.new_cell_types[...] = CELL_INACTIVE
    s.cell_types[4, 4, 4], s.cell_types[4, 4, 3] = old_self, old_nbr
    s.new_cell_types[4, 4, 4], s.new_cell_types[4, 4, 3] = new_self, new_nbr
    s.velocities_1[4, 4, 4] = [7.0, 8.0, 9.0, 5.0]
    s.velocities_2[4, 4, 4] = [-1.0, -2.0, -3.0, 5.0]
    s.run_section("05_set_extrapolated_velocities")
    act = lambda t: t in (CELL_AIR, CELL_WATER)  # noqa: E731
    was = act(old_self) or act(old_nbr)
    now = act(new_self) or act(new_nbr)
    exp_x = 7.0 if was == now else (0.0 if was else -1.0)
    assert s.velocities_1[4, 4, 4, 0] == exp_x
    # y and z faces only see the cell itself (their -y/-z neighbours are inactive in both maps)
    was_s, now_s = act(old_self), act(new_self)
    assert s.velocities_1[4, 4, 4, 1] == (8.0 if was_s == now_s else (0.0 if was_s else -2.0))
    assert s.velocities_1[4, 4, 4, 2] == (9.0 if was_s == now_s else (0.0 if was_s else -3.0))
    assert s.velocities_1[4, 4, 4, 3] == 0.0


# ---- 04_compute_extrapolated_velocities --------------------------------------------------------------------
def test_04_mean_of_water_neighbours():
    s = make_state(8)
    s.cell_types[...] = CELL_AIR
    s.cell_types[4, 4, 3] = CELL_WATER   # -x
    s.cell_types[4, 5, 4] = CELL_WATER   # +y
    s.cell_types[5, 4, 4] = CELL_WATER   # +z
    s.velocities_1[4, 4, 3] = [1.0, 2.0, 3.0, 9.0]
    s.velocities_1[4, 5, 4] = [0.5, -1.0, 0.25, 9.0]
    s.velocities_1[5, 4, 4] = [10.0, 0.0, -4.0, 9.0]
    s.run_section("04_compute_extrapolated_velocities")
    exp = [(F(1.0) + F(0.5) + F(10.0)) / F(3), (F(2.0) + F(-1.0) + F(0.0)) / F(3),
           (F(3.0) + F(0.25) + F(-4.0)) / F(3), F(0)]
    assert s.velocities_2[4, 4, 4].tolist() == [float(e) for e in exp]
    assert s.velocities_2[1, 1, 1].tolist() == [0, 0, 0, 0]  # no water neighbour (:55)


# ---- K9: 09_diffuse/diffuse.comp:31-46 ------------------------------------------------------------------------
def test_k9_diffuse_as_written_is_a_copy_and_intended_differs():
    rng = np.random.default_rng(9)
    s = make_state(8)
    s.cell_types[...] = CELL_WATER
    s.velocities_2[...] = rng.standard_normal(s.velocities_2.shape).astype(F)
    s.velocities_1[...] = -5.0
    s.run_section("09_diffuse")
    np.testing.assert_array_equal(s.velocities_1[..., :3], s.velocities_2[..., :3])
    assert np.all(s.velocities_1[..., 3] == 0.0)
    s.diffuse_mode = 1
    s.run_section("09_diffuse")
    v = s.velocities_2
    a = F(0.01) * F(0.01)
    z, y, x = 4, 4, 4
    nb = v[z, y, x + 1, :3] + v[z, y, x - 1, :3]
    nb = nb + v[z, y + 1, x, :3]
    nb = nb + v[z, y - 1, x, :3]
    nb = nb + v[z + 1, y, x, :3]
    nb = nb + v[z - 1, y, x, :3]
    exp = (F(1.0) - F(6.0) * a) * v[z, y, x, :3] + a * nb
    np.testing.assert_array_equal(s.velocities_1[z, y, x, :3], exp)
    assert np.max(np.abs(s.velocities_1[..., :3] - v[..., :3])) < 6e-4 * np.max(np.abs(v)) * 2


# ---- K10: an empty grid is a fixed point of the step (up to walls) -----------------------------------------------
def test_k10_empty_grid_fixed_point():
    s = make_state(8, capacity=16, iters=4)
    s.run_init()
    s.particles[...] = 0.0  # all particles inactive
    s.run_step()
    t = s.cell_types
    assert np.all(t[1:7, 1:7, 1:7] == CELL_INACTIVE)
    assert np.count_nonzero(t == CELL_SOLID) == 8 ** 3 - 6 ** 3
    v = s.velocities_1
    r = F(0.01)
    assert set(np.unique(v[..., :3]).tolist()) <= {float(-r), 0.0, float(r)}
    # interior cell next to the x = 0 wall: its -x face is pushed to +r (solids.comp:50-51)
    assert v[3, 3, 1, 0] == r and v[3, 3, 1, 1] == 0 and v[3, 3, 1, 2] == 0
    # a wall cell at x = 7 (not an edge): own components -r (:32-33); its -x neighbour is interior
    # so x stays -r, its -y / -z neighbours are wall cells too so those faces flip to +r (:50-51)
    assert v[3, 3, 7].tolist()[:3] == [float(-r), float(r), float(r)]
    # wall cell at x = 0: -x neighbour is out of bounds -> stays -r
    assert v[3, 3, 0, 0] == -r
    assert np.all(s.pressures_1 == 1.0) and np.all(s.pressures_2 == 1.0)
    v1 = v.copy()
    t1 = t.copy()
    s.run_step()
    np.testing.assert_array_equal(s.velocities_1, v1)
    np.testing.assert_array_equal(s.cell_types, t1)


# ---- 14_particles ---------------------------------------------------------------------------------------------------
def test_14_particles_move_with_sampled_velocity():
    s = make_state(8, capacity=3)
    s.velocities_1[..., 0] = 2.0
    s.velocities_1[..., 1] = -1.0
    s.velocities_1[..., 2] = 0.5
    s.particles[:] = [[3.25, 4.5, 2.75, 1.0], [1.0, 1.0, 1.0, 0.0], [100.0, -3.0, 4.0, 1.0]]
    s.run_section("14_particles")
    dt = F(0.01)
    exp0 = [F(3.25) + F(2.0) * dt, F(4.5) + F(-1.0) * dt, F(2.75) + F(0.5) * dt, F(1)]
    assert s.particles[0].tolist() == [float(e) for e in exp0]
    assert s.particles[1].tolist() == [1.0, 1.0, 1.0, 0.0]      # inactive: untouched
    exp2 = [F(100.0) + F(2.0) * dt, F(-3.0) + F(-1.0) * dt, F(4.0) + F(0.5) * dt, F(1)]
    assert s.particles[2].tolist() == [float(e) for e in exp2]  # never clamped (F6)
