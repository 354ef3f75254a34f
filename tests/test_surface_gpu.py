"""Parity of the surface-prep passes 15-18 (detailed grid, images 8..11) against the oracle, through the
C ABI.  Bit-exact, like everything else."""
import numpy as np
import pytest

import fluid_amd
from fluid_amd import engine as E
from fluid_amd.params import CELL_SOLID, dam_break_params
from helpers import assert_bit_equal, assert_state_equal, random_state, upload_state
from oracle_binding import OracleState

pytestmark = pytest.mark.gpu

SURFACE_IMAGES = {"detailed_densities": E.DETAILED_DENSITIES_IMG,
                  "detailed_densities_inertia": E.DETAILED_DENSITIES_INERTIA_IMG,
                  "float_densities_1": E.PARTICLE_DENSITIES_FLOAT_1,
                  "float_densities_2": E.PARTICLE_DENSITIES_FLOAT_2}


def surface_state(size, res, seed, cap=3000):
    base = random_state(size, capacity=cap, seed=seed)
    base.params.detailed_resolution = res
    st = OracleState(base.params, cap, 4, surface_prep=True)
    for f in OracleState.FIELDS:
        getattr(st, f)[...] = getattr(base, f)
    rng = np.random.default_rng(seed)
    st.detailed_densities[...] = rng.integers(0, 2, st.detailed_shape) * rng.integers(0, 3, st.detailed_shape)
    st.detailed_densities_inertia[...] = rng.integers(0, 120, st.detailed_shape)
    st.float_densities_1[...] = rng.uniform(-1, 3, st.detailed_shape).astype(np.float32)
    st.float_densities_2[...] = rng.uniform(-1, 3, st.detailed_shape).astype(np.float32)
    return st


def engine_for(st):
    eng = fluid_amd.FluidEngine(st.params, particle_capacity=st.capacity, pressure_iterations=4,
                                surface_prep=True, surface_diffuse_steps=st.surface_diffuse_steps)
    upload_state(eng, st)
    for f, img in SURFACE_IMAGES.items():
        eng.upload_image(img, getattr(st, f))
    return eng


def assert_surface_equal(eng, st, ctx):
    for f, img in SURFACE_IMAGES.items():
        assert_bit_equal(eng.download_image(img), getattr(st, f), f"{ctx}{f}")


@pytest.mark.parametrize("size,res", [((12, 9, 7), 5), ((16, 5, 4), 3), ((7, 6, 5), 1), ((33, 3, 2), 2)])
@pytest.mark.parametrize("section", ["14a_clear_detailed_densities", "15_update_detailed_densities",
                                     "16_compute_detailed_densities_inertia",
                                     "17_compute_float_densities", "18_diffuse_float_densities",
                                     "init_clear_detailed_densities_inertia"])
def test_surface_section_matches_oracle(section, size, res):
    st = surface_state(size, res, seed=len(section) + size[0])
    with engine_for(st) as eng:
        eng.run_section(section)
        st.run_section(section)
        if section == "18_diffuse_float_densities":   # the second dispatch goes the other way
            eng.run_section(section)
            st.run_section(section)
        assert_surface_equal(eng, st, f"{section} {size} x{res}: ")
        assert_state_equal(eng, st, ctx="the simulation images are untouched: ")


@pytest.mark.parametrize("kernel", [0, 1])
def test_surface_loop_and_inertia_parameters(kernel):
    st = surface_state((10, 8, 6), 4, seed=2)
    st.params.required_neighbour_hits = 2
    st.params.inertia_increase_neighbour = 3
    st.params.inertia_decrease = 7
    st.params.max_inertia = 90
    st.params.dens_diffuse_k = 0.13
    with engine_for(st) as eng:
        eng.set_option(E.OPT_SURFACE_KERNEL, kernel)
        for s in ("16_compute_detailed_densities_inertia", "17_compute_float_densities"):
            eng.run_section(s)
            st.run_section(s)
        eng.run_section_loop("18_diffuse_float_densities", 5)
        st.diffuse_float_densities(5)
        assert_surface_equal(eng, st, "loop x5: ")


def test_surface_diffuse_many_planes_and_partial_tiles():
    """18 with the z-marching kernel over several z chunks (zchunk 32) and partial x / y tiles."""
    st = surface_state((68, 5, 18), 4, seed=9, cap=100)   # detailed 272 x 20 x 72
    with engine_for(st) as eng:
        eng.run_section_loop("18_diffuse_float_densities", 3)
        st.diffuse_float_densities(3)
        assert_surface_equal(eng, st, "z march: ")


@pytest.mark.parametrize("steps", [1, 2, 3, 4, 5, 6, 7])
@pytest.mark.parametrize("size,res", [((68, 5, 18), 4), ((63, 7, 3), 8), ((10, 8, 6), 4)])
def test_surface_diffuse_loop_two_dispatches_per_pass(size, res, steps):
    """The 18 loop runs two dispatches per pass over HBM (k18_pair): several x tiles of 248 cells (detailed
    widths 272 and 504), y tiles of 6 rows, more than one z chunk, random SOLID cells (their stale values in
    both float images feed their neighbours, diffuse_densities.comp:56), every loop count parity: FLOAT_1
    must hold the newest even iterate, FLOAT_2 the newest odd one, exactly as N single dispatches leave them."""
    st = surface_state(size, res, seed=sum(size) + steps, cap=100)
    with engine_for(st) as eng:
        if steps == 6:
            eng.set_option(E.OPT_SURFACE_KERNEL, 108)   # 8 rows per workgroup instead of 12
        eng.run_section_loop("18_diffuse_float_densities", steps)
        st.diffuse_float_densities(steps)
        assert_surface_equal(eng, st, f"{size} x{steps}: ")
        # and again on what the first loop left (the third image holds an old iterate now)
        eng.run_section_loop("18_diffuse_float_densities", 4)
        st.diffuse_float_densities(4)
        assert_surface_equal(eng, st, f"{size} x{steps} + 4: ")


def test_full_step_with_surface_prep_matches_oracle():
    """fluid_run_init / fluid_run_step of a surface_prep context = the reference's complete section
    lists (fluid_flow_sections.h:139-154, 163-388) up to the renderer."""
    size, iters = (24, 20, 16), 8
    p, cap = dam_break_params(*size)
    st = OracleState(p, cap, iters, surface_prep=True)
    with fluid_amd.FluidEngine(p, particle_capacity=cap, pressure_iterations=iters,
                               surface_prep=True) as eng:
        # the inertia image starts dirty: the init list clears it (:142)
        eng.upload_image(E.DETAILED_DENSITIES_INERTIA_IMG,
                         np.full(eng.detailed_shape, 17, np.uint32))
        eng.run_init()
        st.run_init()
        for k in range(4):
            eng.run_step()
            st.run_step()
            assert_state_equal(eng, st, ctx=f"step {k}: ")
            assert_surface_equal(eng, st, f"step {k}: ")
        assert np.count_nonzero(st.detailed_densities) > 100 and np.any(st.float_densities_1 != -1.0)
        times = None
        eng.enable_timing(True)
        eng.run_step()
        times = eng.section_times()
        assert times["18_diffuse_float_densities"][1] == 4 and times["15_update_detailed_densities"][1] == 1


def test_surface_images_need_a_surface_context():
    p, cap = dam_break_params(16, 16, 16)
    with fluid_amd.FluidEngine(p, particle_capacity=cap) as eng:
        with pytest.raises(fluid_amd.FluidEngineError, match="surface_prep"):
            eng.run_section("15_update_detailed_densities")
        with pytest.raises(fluid_amd.FluidEngineError) as ei:
            eng.download_image(E.PARTICLE_DENSITIES_FLOAT_1)
        assert ei.value.code == E.ERR_UNSUPPORTED
    with pytest.raises(fluid_amd.FluidEngineError):   # not on a Z slab
        fluid_amd.FluidEngine(p, particle_capacity=cap, slab=(0, 8), surface_prep=True)
    with fluid_amd.FluidEngine(p, particle_capacity=cap, surface_prep=True) as eng:
        q = p.copy()
        q.detailed_resolution = 3   # the detailed images were sized for 5
        with pytest.raises(fluid_amd.FluidEngineError, match="detailed_resolution"):
            eng.set_params(q)


def test_cpp_section_lists_with_surface_prep(tmp_path):
    """host/fluid_sim with `surface`: the C++ mirror's complete lists (init incl. the inertia clear, step
    incl. 15-18 through FlowSectionList) against the oracle."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "vulkan-3d-fluid-simulation_amd", "host", "fluid_sim")
    subprocess.run(["make", "-C", os.path.dirname(exe)], check=True, capture_output=True)
    size, frames, iters = (16, 16, 16), 3, 6
    res = subprocess.run([exe, *map(str, size), str(frames), str(iters), str(tmp_path), "surface"],
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "22 sections" in res.stdout  # 01a...14 plus the clear, 15, 16, 17 and the 18 loop
    p, cap = dam_break_params(*size)
    st = OracleState(p, cap, iters, surface_prep=True)
    st.run_init()
    for _ in range(frames):
        st.run_step()
    got = np.fromfile(os.path.join(str(tmp_path), "float_densities_1.bin"), dtype=np.float32)
    assert_bit_equal(got.reshape(st.detailed_shape), st.float_densities_1, "C++ float densities")
    got = np.fromfile(os.path.join(str(tmp_path), "particles.bin"), dtype=np.float32)
    assert_bit_equal(got.reshape(st.particles.shape), st.particles, "C++ particles")


def test_surface_extraction_matches_oracle_triangles():
    """the triangles of the reference's marching-cubes surface (31_render_surface) from a
    float density image of the detailed grid — a bumpy ball and a noisy slab with exact zeros,
    uploaded as PARTICLE_DENSITIES_FLOAT_2 / _1 — the engine's list (unordered) against the oracle's (cell
    order), as sorted sets, bit for bit; tables through MARCHING_CUBES_COUNTS_BUF / _EDGES_BUF."""
    import os
    from fluid_amd import engine as E
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "marching_cubes_tables.npz"))
    counts, edges = g["counts"], g["edge_indices"]
    p, cap = fluid_amd.dam_break_params(16, 12, 8)
    st = OracleState(p, cap, 2, surface_prep=True)
    d, h, w = st.detailed_shape          # 40 x 60 x 80
    z, y, x = np.meshgrid(np.arange(d), np.arange(h), np.arange(w), indexing="ij")
    ball = (17.3 - np.sqrt((x - 40.2) ** 2 + (y - 28.7) ** 2 + (z - 19.1) ** 2)
            + 2.5 * np.sin(0.7 * x) * np.cos(0.9 * y + 0.3 * z)).astype(np.float32)
    rng = np.random.default_rng(3)
    slab = (np.float32(9.5) - y + rng.standard_normal((d, h, w)) * 0.8).astype(np.float32)
    slab[rng.uniform(0, 1, slab.shape) < 0.01] = 0.0      # exact zeros: d > 0 is false there
    with fluid_amd.FluidEngine(p, particle_capacity=cap, pressure_iterations=2, surface_prep=True) as eng:
        with pytest.raises(fluid_amd.FluidEngineError, match="upload MARCHING_CUBES"):
            eng.extract_surface()
        eng.upload_marching_cubes_tables(counts, edges)
        eng.upload_image(E.PARTICLE_DENSITIES_FLOAT_2, ball)
        eng.upload_image(E.PARTICLE_DENSITIES_FLOAT_1, slab)

        def canon(t):
            rows = np.ascontiguousarray(t.reshape(len(t), 12)).view(np.uint32)
            return rows[np.lexsort(rows.T[::-1])]
        for img, field in ((E.PARTICLE_DENSITIES_FLOAT_2, ball), (E.PARTICLE_DENSITIES_FLOAT_1, slab)):
            got = eng.extract_surface(img)
            exp = st.extract_surface(field, counts, edges)
            assert got.shape == exp.shape and got.shape[0] > 5000
            assert np.array_equal(canon(got), canon(exp))
        assert eng.extract_surface().shape == eng.extract_surface(E.PARTICLE_DENSITIES_FLOAT_2).shape
        bad = counts.copy()
        bad[7] = 9
        with pytest.raises(fluid_amd.FluidEngineError, match="out of range"):
            eng.upload_marching_cubes_tables(bad, edges)
    with fluid_amd.FluidEngine(p, particle_capacity=cap) as plain:
        with pytest.raises(fluid_amd.FluidEngineError):
            plain.upload_marching_cubes_tables(counts, edges)
