"""Known-answer tests of the oracle's surface-prep passes 15-18, derived from the shader text
(/root/reference/shaders_fluid/15...18).  CPU only."""
import numpy as np

import fluid_amd
from fluid_amd.params import CELL_AIR, CELL_SOLID, CELL_WATER
from oracle_binding import OracleState


def _state(size=(4, 3, 2), cap=8, res=None):
    p = fluid_amd.default_params(*size, cap)
    if res is not None:
        p.detailed_resolution = res
    return OracleState(p, cap, 4, surface_prep=True)


def test_15_counts_particles_on_the_detailed_grid():
    st = _state()
    assert st.detailed_shape == (10, 15, 20)  # reference resolution 5 (simulation_constants.h:36)
    st.particles[0] = (0.0, 0.0, 0.0, 1.0)          # -> cell (0,0,0)
    st.particles[1] = (0.39, 0.2, 0.0, 1.0)         # 1.95, 1.0, 0 -> (1,1,0)
    st.particles[2] = (0.39, 0.2, 0.0, 1.0)         # twice
    st.particles[3] = (3.999, 2.999, 1.999, 1.0)    # last cell (19,14,9)
    st.particles[4] = (-0.1, 0.0, 0.0, 1.0)         # -0.5 truncates toward zero -> x = 0
    st.particles[5] = (4.0, 0.0, 0.0, 1.0)          # 20.0: outside, dropped
    st.particles[6] = (1.0, 1.0, 1.0, 0.0)          # inactive
    st.particles[7] = (-0.2, 0.0, 0.0, 1.0)         # -1.0: outside (v > -1 fails), dropped
    st.run_section("15_update_detailed_densities")
    d = st.detailed_densities
    assert d[0, 0, 0] == 2 and d[0, 1, 1] == 2 and d[9, 14, 19] == 1 and d.sum() == 5


def test_16_inertia_rules():
    st = _state()
    p = st.params   # max 100, +4 if filled, +1 per filled neighbour if >= 1 hit, else -1
    d, i = st.detailed_densities, st.detailed_densities_inertia
    d[5, 5, 5] = 3
    i[...] = 10
    i[5, 5, 5] = 99
    i[0, 0, 0] = 0
    st.run_section("16_compute_detailed_densities_inertia")
    assert i[5, 5, 5] == 100                      # 99 + 4 capped at max_inertia
    assert i[5, 5, 6] == 11 and i[4, 5, 5] == 11  # one filled neighbour: + 1 * 1
    assert i[5, 6, 6] == 9                        # nothing added: decays by 1
    assert i[0, 0, 0] == 0                        # 0 stays 0 (0 > 1 is false -> 0)
    p.required_neighbour_hits = 2
    i[...] = 10
    st.run_section("16_compute_detailed_densities_inertia")
    assert i[5, 5, 6] == 9                        # one hit is not enough any more
    assert i[5, 5, 5] == 14


def test_17_and_18():
    st = _state(size=(3, 3, 3), cap=0, res=2)
    i = st.detailed_densities_inertia
    i[...] = 0
    i[2, 2, 2] = 60
    st.run_section("17_compute_float_densities")
    f1, f2 = st.float_densities_1, st.float_densities_2
    assert f1[2, 2, 2] == np.float32(60.0) / np.float32(30.0) and f1[0, 0, 0] == -1.0
    st.cell_types[...] = CELL_AIR
    st.cell_types[0] = CELL_SOLID            # simulation plane z = 0 -> detailed planes 0, 1 are not written
    f2[...] = 7.0
    st.run_section("18_diffuse_float_densities")   # first dispatch after 17: FLOAT_1 -> FLOAT_2
    a = np.float32(0.1)
    k0 = np.float32(1.0) - np.float32(6.0) * a
    s = np.float32(-6.0)                     # six neighbours of the centre hold -1
    assert f2[2, 2, 2] == k0 * np.float32(2.0) + a * s
    # a neighbour of the centre: ((((2 + -1) + -1) + -1) + -1) + -1 in the written order (+x, -x, ...)
    s2 = np.float32(-1.0) + np.float32(2.0)  # +x = -1 first?  cell (2,2,3): -x neighbour is the centre
    s2 = np.float32(-1.0) + np.float32(2.0)
    for _ in range(4):
        s2 = s2 + np.float32(-1.0)
    assert f2[2, 2, 3] == k0 * np.float32(-1.0) + a * s2
    assert np.all(f2[0:2] == 7.0)            # solid simulation cells: untouched
    # corner of the non-solid part: out-of-bounds neighbours load 0
    c = f2[5, 5, 5]
    assert c == k0 * np.float32(-1.0) + a * np.float32(-3.0)
    st.run_section("18_diffuse_float_densities")   # second dispatch: FLOAT_2 -> FLOAT_1
    assert f1[2, 2, 2] != np.float32(2.0) and np.all(f1[0:2] == -1.0)


def test_loop_and_step_order():
    st = _state(size=(4, 4, 4), cap=64)
    st.run_init()
    st.particles[:, :3] = np.random.default_rng(1).uniform(0.5, 3.5, (64, 3)).astype(np.float32)
    st.particles[:, 3] = 1.0
    ref = st.copy()
    st.run_step()
    ref_sections = (OracleState.STEP_BEFORE_12, OracleState.STEP_AFTER_12)
    for s in ref_sections[0]:
        ref.run_section(s)
    ref.solve_pressure(ref.pressure_iterations)
    for s in ref_sections[1]:
        ref.run_section(s)
    for s in OracleState.SURFACE_ORDER:
        ref.run_section(s)
    for _ in range(4):                          # float_density_diffuse_steps (simulation_constants.h:127)
        ref.run_section("18_diffuse_float_densities")
    for f in OracleState.SURFACE_FIELDS:
        assert np.array_equal(getattr(st, f).view(np.uint32), getattr(ref, f).view(np.uint32)), f
    assert np.count_nonzero(st.detailed_densities) > 0 and np.count_nonzero(st.float_densities_1 > 0) > 0


def mc_tables():
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "marching_cubes_tables.npz"))
    return g["counts"], g["edge_indices"]


def test_marching_cubes_tables_are_consistent():
    counts, edges = mc_tables()
    assert counts.shape == (256,) and edges.shape == (3840,)
    assert counts[0] == 0 and counts[255] == 0 and counts.max() == 5
    assert int(counts.sum()) == 820     # the classic table (Lorensen & Cline's cases as tabulated by Bourke)
    assert all(counts[1 << i] == 1 and counts[255 - (1 << i)] == 1 for i in range(8))  # one corner: one triangle
    for c in range(256):
        row = edges[15 * c:15 * c + 15]
        assert np.all(row[:3 * counts[c]] < 12) and np.all(row[3 * counts[c]:] == 255)


def test_kat_marching_cubes_one_corner_inside():
    """render_surface.geom:60-69 by hand: a single detailed-grid corner with positive density (configuration
    1 of the cell whose corner 0 it is) gives one triangle on the edges 0, 8, 3 = towards +x, +z, +y, each
    vertex where the density crosses zero: a = d0 / (d0 - d1); the other seven cells around that corner see
    it as another of their corners and cut it off likewise: eight triangles, an octahedron around it."""
    from oracle_binding import OracleState
    import fluid_amd
    counts, edges = mc_tables()
    p = fluid_amd.default_params(2, 2, 2, 0)
    p.detailed_resolution = 2
    st = OracleState(p, 0, 1)
    d = np.full((4, 4, 4), -1.0, np.float32)
    d[1, 1, 1] = 3.0          # inside; all neighbours -1: the crossing lies 3/4 of the way to each neighbour
    tris = st.extract_surface(d, counts, edges)
    assert tris.shape == (8, 4, 3)
    # the cell whose corner 0 is (1, 1, 1): vertex-index order puts it last
    t = tris[-1]
    base = (0.5 + 1.0)
    exp = np.array([[base + 0.75, base, base], [base, base, base + 0.75], [base, base + 0.75, base]],
                   np.float32) / np.float32(2.0)
    assert np.array_equal(t[:3], exp)
    n = np.cross(t[1] - t[0], t[2] - t[0])
    assert np.allclose(t[3], n / np.linalg.norm(n), atol=1e-6) and abs(np.linalg.norm(t[3]) - 1) < 1e-6
    # all 24 vertices lie at distance 0.75 / res from the inside corner's render position
    centre = np.float32((0.5 + 1.0) / 2.0)
    dist = np.linalg.norm(tris[:, :3].reshape(-1, 3) - centre, axis=1)
    assert np.allclose(dist, 0.375, atol=1e-6)
    # capacity smaller than the count: counted, not stored
    import ctypes as C
    from oracle_binding import lib
    nn = C.c_uint64(0)
    few = np.zeros((3, 4, 3), np.float32)
    lib().oracle_31_extract_surface(C.byref(st.params), d.ctypes.data, counts.ctypes.data, edges.ctypes.data,
                                    few.ctypes.data, 3, C.byref(nn))
    assert nn.value == 8 and np.array_equal(few, tris[:3])

