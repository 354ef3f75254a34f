"""Parity of the HIP engine against the CPU oracle, through the C ABI.  Needs an MI355X.

Tolerance: BIT-EXACT for every field, integer and fp32 alike.  Both sides evaluate the shader
arithmetic in fp32 with one rounding per operation in source order (-ffp-contract=off, IEEE
division), so the stated north-star tolerance ("a stated float tolerance per field") is 0 ulp here;
any mismatch is a bug, not noise.
"""
import os
import zlib

import numpy as np
import pytest

import fluid_amd
from fluid_amd import engine as E
from fluid_amd import scenes
from fluid_amd.params import CELL_AIR, CELL_SOLID, CELL_WATER, dam_break_params, default_params
from helpers import (IMAGE_FIELDS, assert_bit_equal, assert_bit_equal_any_nan, assert_state_equal,
                     download_state, make_engine, random_state, upload_state)
from oracle_binding import OracleState

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

GRID_SECTIONS = ["02_update_water", "03_update_air", "04_compute_extrapolated_velocities",
                 "05_set_extrapolated_velocities", "06_update_cell_types", "07_advect",
                 "08_forces", "09_diffuse", "10_solids", "11_compute_divergence",
                 "13_fix_divergence"]
CLEAR_SECTIONS = ["init_clear_velocities_1", "init_clear_cell_types",
                  "01a_clear_particle_densities", "12a_clear_pressures_1", "12b_clear_pressures_2"]
PARTICLE_SECTIONS = ["00_init_particles", "01_update_densities", "14_particles"]

SIZES = [(24, 20, 16), (17, 13, 9), (64, 8, 5), (5, 5, 5), (260, 6, 4)]


@pytest.mark.parametrize("size", SIZES)
@pytest.mark.parametrize("section", GRID_SECTIONS + CLEAR_SECTIONS + PARTICLE_SECTIONS)
def test_section_matches_oracle(section, size):
    cap = 3000
    st = random_state(size, capacity=cap, seed=zlib.crc32(repr((section, size)).encode()) % 1000)
    if section == "00_init_particles":
        st.params.particle_spawn_cube_resolution[:] = (13, 11, 17)
        st.params.particle_spawn_cube_volume = 13 * 11 * 17  # < capacity: tail becomes inactive
        st.params.particle_spawn_cube_offset[:] = (0.3 * size[0], 0.1 * size[1], 0.2 * size[2])
        st.params.particle_spawn_cube_size[:] = (0.5 * size[0], 0.6 * size[1], 0.3 * size[2])
    with make_engine(st) as eng:
        eng.run_section(section)
        st.run_section(section)
        assert_state_equal(eng, st, ctx=f"{section} {size}: ")


STEP_LIST = list(OracleState.STEP_BEFORE_12)
# (first section, count, images the group may leave unspecified, a section whose own kernel must not run)
GROUPS = [("04_compute_extrapolated_velocities", 2, ["velocities_2"], None),
          ("07_advect", 2, [], "08_forces"),
          ("09_diffuse", 3, [], "10_solids"),
          ("02_update_water", 10, ["velocities_2"], "08_forces")]


@pytest.mark.parametrize("size", [(24, 20, 16), (64, 8, 5), (260, 6, 4), (8, 5, 5), (17, 13, 9)])
@pytest.mark.parametrize("group", GROUPS, ids=[g[0] + "x" + str(g[1]) for g in GROUPS])
def test_section_group_matches_oracle(group, size):
    """fluid_run_section_group: grouped passes (kernels_step_fused.h) against the oracle running the
    same sections one by one, on scenes with every cell type anywhere (interior solids included)."""
    first, count, unspecified, skipped = group
    st = random_state(size, capacity=200, seed=zlib.crc32(repr((first, size)).encode()) % 1000)
    with make_engine(st) as eng:
        # 13 leaves w = 0 in all of VELOCITIES_1, the precondition of the grouped 04+05
        eng.run_section("13_fix_divergence")
        st.run_section("13_fix_divergence")
        eng.enable_timing(True)
        eng.run_section_group(first, count)
        i = STEP_LIST.index(first)
        for s in STEP_LIST[i:i + count]:
            st.run_section(s)
        fields = [f for f in list(IMAGE_FIELDS) + ["particles"] if f not in unspecified]
        if count == 10:  # ... through 11: VELOCITIES_2 is rewritten by 07 (+08), so it is exact again
            fields = list(IMAGE_FIELDS) + ["particles"]
        assert_state_equal(eng, st, fields=fields, ctx=f"group {first}x{count} {size}: ")
        if skipped and size[0] % 4 == 0:
            assert eng.section_times()[skipped][1] == 0  # the grouped pass ran, not the list


def test_section_group_errors():
    st = random_state((8, 8, 8), seed=3)
    with make_engine(st) as eng:
        for first, count in [(E.SECTION_IDS["00_init_particles"], 1), (E.SECTION_IDS["14_particles"], 2),
                             (99, 1)]:
            with pytest.raises(fluid_amd.FluidEngineError, match="not a slice"):
                eng.run_section_group(first, count)
        eng.run_section_group("14_particles", 0)  # empty slice: nothing to do


@pytest.mark.parametrize("size", [(24, 20, 16), (32, 12, 40)])
def test_full_step_grouped_equals_list_and_oracle(size):
    """fluid_run_step with and without FLUID_OPT_STEP_FUSION from an arbitrary uploaded state: the
    first step runs 04/05 from the list (w of VELOCITIES_1 unknown after an upload), later ones grouped."""
    st = random_state(size, capacity=4000, seed=5, iters=6)
    with make_engine(st) as a, make_engine(st) as b:
        b.set_option(E.OPT_STEP_FUSION, 1)
        for k in range(3):
            a.run_step()
            b.run_step()
            st.run_step()
            assert_state_equal(a, st, ctx=f"grouped, step {k}: ")
            assert_state_equal(b, st, ctx=f"list, step {k}: ")


@pytest.mark.parametrize("size", [(24, 20, 16), (17, 13, 9)])
def test_diffuse_intended_mode(size):
    st = random_state(size, seed=11)
    st.diffuse_mode = E.DIFFUSE_INTENDED
    with make_engine(st) as eng:
        eng.run_section("09_diffuse")
        st.run_section("09_diffuse")
        assert_state_equal(eng, st, ctx="09_diffuse intended: ")


@pytest.mark.parametrize("variant", [1, 2, 3, 4])
@pytest.mark.parametrize("size", [(24, 20, 16), (64, 64, 64), (260, 12, 9), (512, 6, 3)])
@pytest.mark.parametrize("iters", [1, 2, 7])
def test_pressure_loop_matches_oracle(variant, size, iters):
    st = random_state(size, seed=iters + 17 * variant)
    with make_engine(st) as eng:
        eng.set_option(E.OPT_PRESSURE_KERNEL, variant)
        eng.solve_pressure(iters)
        st.solve_pressure(iters)
        assert_state_equal(eng, st, fields=["pressures_1", "pressures_2"],
                           ctx=f"12 x{iters} variant {variant} {size}: ")


def test_pressure_odd_width_uses_plain_kernel():
    st = random_state((17, 13, 9), seed=5)
    with make_engine(st) as eng:
        eng.set_option(E.OPT_PRESSURE_KERNEL, 2)  # z-march needs W % 4 == 0: engine falls back
        eng.solve_pressure(3)
        st.solve_pressure(3)
        assert_state_equal(eng, st, fields=["pressures_1", "pressures_2"])


def test_pressure_dispatch_counter_and_explicit_push_constant():
    st = random_state((24, 20, 16), seed=3)
    with make_engine(st) as eng:
        # after 12a/12b the section's own counter starts at "even" 
        eng.run_section("12a_clear_pressures_1")
        eng.run_section("12b_clear_pressures_2")
        st.run_section("12a_clear_pressures_1")
        st.run_section("12b_clear_pressures_2")
        for k in range(3):
            eng.run_section("12_solve_pressure")
            st.pressure_dispatch(1 if k % 2 == 0 else 0)
        assert_state_equal(eng, st, fields=["pressures_1", "pressures_2"], ctx="counter: ")
        eng.run_pressure_dispatch(0)
        st.pressure_dispatch(0)
        eng.run_pressure_dispatch(7)  # anything but 1 reads P2 (pressure.comp:71)
        st.pressure_dispatch(7)
        assert_state_equal(eng, st, fields=["pressures_1", "pressures_2"], ctx="explicit: ")


def test_pressure_walled_in_cell_produces_the_same_inf_nan():
    st = random_state((24, 20, 16), seed=8)
    st.cell_types[4:7, 4:7, 4:7] = CELL_SOLID
    st.cell_types[5, 5, 5] = CELL_WATER
    st.divergences[5, 5, 5] = 0.0  # -0/0 -> NaN
    st.cell_types[9:12, 9:12, 9:12] = CELL_SOLID
    st.cell_types[10, 10, 10] = CELL_WATER
    st.divergences[10, 10, 10] = 0.25  # -s/0 -> -inf
    for variant in (1, 2):
        s2 = st.copy()
        with make_engine(s2) as eng, np.errstate(all="ignore"):
            eng.set_option(E.OPT_PRESSURE_KERNEL, variant)
            eng.solve_pressure(3)
            s2.solve_pressure(3)
            assert np.isnan(s2.pressures_2[5, 5, 5]) and np.isinf(s2.pressures_2[10, 10, 10])
            assert_state_equal(eng, s2, fields=["pressures_1", "pressures_2"])
            # the scan the reference lacks (fluid_count_nonfinite) finds exactly these cells
            for img, arr in ((E.PRESSURES_1, s2.pressures_1), (E.PRESSURES_2, s2.pressures_2),
                             (E.VELOCITIES_1, s2.velocities_1), (E.DIVERGENCES, s2.divergences)):
                assert eng.count_nonfinite(img) == int(np.count_nonzero(~np.isfinite(arr)))
            assert eng.count_nonfinite(E.PRESSURES_2) == 2
            with pytest.raises(fluid_amd.FluidEngineError, match="fp32"):
                eng.count_nonfinite(E.CELL_TYPES)


@pytest.mark.parametrize("size,iters,steps,box", [((32, 32, 32), 20, 3, 0), ((24, 40, 20), 7, 2, 0),
                                                  ((32, 32, 32), 20, 3, 1)])
def test_full_step_dam_break_matches_oracle(size, iters, steps, box):
    p, cap = dam_break_params(*size)
    st = OracleState(p, cap, iters)
    with fluid_amd.FluidEngine(p, particle_capacity=cap, pressure_iterations=iters) as eng:
        eng.set_option(E.OPT_LAUNCH_BOX, box)   # 1: full-grid Jacobi launches, no per-step synchronisation
        eng.run_init()
        st.run_init()
        assert_state_equal(eng, st, fields=["velocities_1", "cell_types", "particles"], ctx="init: ")
        for k in range(steps):
            eng.run_step()
            st.run_step()
            assert_state_equal(eng, st, ctx=f"step {k}: ")
        assert np.count_nonzero(st.cell_types == CELL_WATER) > 100  # the scene is not trivial


def test_c1_config_64cubed_40_iterations():
    """BASELINE.json configs[0]: 64^3, 8 particles/cell, 40 Jacobi iterations."""
    p, cap = dam_break_params(64, 64, 64)
    st = OracleState(p, cap, 40)
    with fluid_amd.FluidEngine(p, particle_capacity=cap, pressure_iterations=40) as eng:
        eng.run_init()
        st.run_init()
        for k in range(2):
            eng.run_step()
            st.run_step()
        assert_state_equal(eng, st, ctx="C1: ")
    water = np.count_nonzero(st.cell_types == CELL_WATER)
    assert abs(cap / water - 8.0) < 1.5  # ~8 particles per water cell


def test_c2_config_128cubed_80_iterations():
    """BASELINE.json configs[1]: 128^3, 80 Jacobi iterations, the full 01a...14 pipeline on one GPU — two
    whole steps of the dam break (8 particles per cell), every image and the particles bit-equal to the
    oracle."""
    p, cap = dam_break_params(128, 128, 128)
    st = OracleState(p, cap, 80)
    with fluid_amd.FluidEngine(p, particle_capacity=cap, pressure_iterations=80) as eng:
        eng.run_init()
        st.run_init()
        for k in range(2):
            eng.run_step()
            st.run_step()
            assert_state_equal(eng, st, ctx=f"C2 step {k}: ")
    water = np.count_nonzero(st.cell_types == CELL_WATER)
    assert water > 50_000 and abs(cap / water - 8.0) < 1.5


def test_step_by_sections_equals_run_step():
    p, cap = dam_break_params(32, 32, 32)
    with fluid_amd.FluidEngine(p, particle_capacity=cap, pressure_iterations=10) as a, \
            fluid_amd.FluidEngine(p, particle_capacity=cap, pressure_iterations=10) as b:
        a.run_init()
        b.run_init()
        for _ in range(2):
            a.run_step()
            for s in OracleState.STEP_BEFORE_12:
                b.run_section(s)
            b.run_section_loop("12_solve_pressure", 10)
            for s in OracleState.STEP_AFTER_12:
                b.run_section(s)
        ref = OracleState(p, cap, 10)
        ga, gb = download_state(a, ref), download_state(b, ref)
        for k in ga:
            assert_bit_equal(ga[k], gb[k], k)


def test_no_particles_and_empty_grid_fixed_point():
    p = default_params(16, 16, 16, 0)
    st = OracleState(p, 0, 4)
    with fluid_amd.FluidEngine(p, particle_capacity=0, pressure_iterations=4) as eng:
        eng.run_init()
        st.run_init()
        for k in range(6):   # later steps run with every brick quiet / skipped early
            eng.run_step()
            st.run_step()
            assert_state_equal(eng, st, fields=list(IMAGE_FIELDS), ctx=f"empty grid, step {k}: ")
        assert eng.get_stat(E.STAT_QUIET_BRICKS) == eng.get_stat(E.STAT_BRICKS)


def test_golden_fixture_matches_engine():
    """The committed golden vectors (tests/golden, produced by the oracle) against the engine."""
    import os
    path = os.path.join(os.path.dirname(__file__), "golden", "dam_break_16_steps3.npz")
    g = np.load(path)
    size = tuple(int(v) for v in g["size"])
    iters = int(g["iterations"])
    p, cap = dam_break_params(*size)
    with fluid_amd.FluidEngine(p, particle_capacity=cap, pressure_iterations=iters) as eng:
        eng.run_init()
        for _ in range(int(g["steps"])):
            eng.run_step()
        got = download_state(eng, OracleState(p, cap, iters))
        for name in list(IMAGE_FIELDS) + ["particles"]:
            assert_bit_equal(got[name], g[name], f"golden {name}")


# ---- full-size checks through size-independent properties ---------------------------------------
def test_pressure_256cubed_variants_agree_and_window_matches_oracle():
    """256^3 full-fluid grid (BASELINE config C3 shape), 8 sweeps: (a) the z-marching kernel and
    the plain kernel — two independent HIP implementations — agree bit for bit everywhere;
    (b) a 24-plane window re-computed by the oracle matches on the planes outside the window's
    dependence cone (8 sweeps reach 8 planes)."""
    n, iters = 256, 8
    p = default_params(n, n, n, 0)
    shape = (n, n, n)
    t = scenes.full_fluid_types(shape)
    div = scenes.full_fluid_divergence(shape)
    results = {}
    for variant in (1, 2, 3, 5, 6, 7, 0):
        with fluid_amd.FluidEngine(p, particle_capacity=0) as eng:
            eng.set_option(E.OPT_PRESSURE_KERNEL, variant)
            eng.set_option(E.OPT_JACOBI_FUSE, 0 if variant == 0 else 1)
            eng.upload_image(E.CELL_TYPES, t)
            eng.upload_image(E.DIVERGENCES, div)
            eng.run_section("12a_clear_pressures_1")
            eng.run_section("12b_clear_pressures_2")
            eng.solve_pressure(iters)
            results[variant] = (eng.download_image(E.PRESSURES_1), eng.download_image(E.PRESSURES_2))
    for variant in (2, 3, 5, 6, 7, 0):  # 0 = the default: two sweeps per pass
        assert_bit_equal(results[variant][0], results[1][0], f"P1 variant {variant} vs plain")
        assert_bit_equal(results[variant][1], results[1][1], f"P2 variant {variant} vs plain")
    z0, zc = 100, 24
    pw = default_params(n, n, zc, 0)
    sw = OracleState(pw, 0, iters)
    sw.cell_types[...] = t[z0:z0 + zc]
    sw.divergences[...] = div[z0:z0 + zc]
    sw.pressures_1[...] = 1.0
    sw.pressures_2[...] = 1.0
    sw.solve_pressure(iters)
    lo, hi = iters, zc - iters
    assert_bit_equal(results[2][0][z0 + lo:z0 + hi], sw.pressures_1[lo:hi], "window P1")
    assert_bit_equal(results[2][1][z0 + lo:z0 + hi], sw.pressures_2[lo:hi], "window P2")
    # and the iteration is doing something: pressures moved away from p_air
    assert np.count_nonzero(results[2][0] != 1.0) > 0.9 * (n - 2) ** 3


def test_pressure_all_air_is_untouched_and_all_water_converges_to_fixed_point():
    """Properties that hold at any size: (a) without water nothing is written; (b) with zero
    divergence and p = p_air everywhere, p_air is a fixed point (every neighbour sum is exact)."""
    n = 128
    p = default_params(n, n, n, 0)
    with fluid_amd.FluidEngine(p, particle_capacity=0) as eng:
        eng.upload_image(E.CELL_TYPES, np.full((n, n, n), CELL_AIR, np.uint8))
        rng = np.random.default_rng(0)
        p1 = rng.uniform(0, 2, (n, n, n)).astype(np.float32)
        eng.upload_image(E.PRESSURES_1, p1)
        eng.upload_image(E.PRESSURES_2, p1 + 1)
        eng.upload_image(E.DIVERGENCES, rng.uniform(-1, 1, (n, n, n)).astype(np.float32))
        eng.solve_pressure(4)
        assert_bit_equal(eng.download_image(E.PRESSURES_1), p1, "all air P1")
        assert_bit_equal(eng.download_image(E.PRESSURES_2), p1 + 1, "all air P2")
        eng.upload_image(E.CELL_TYPES, scenes.full_fluid_types((n, n, n)))
        eng.upload_image(E.DIVERGENCES, np.zeros((n, n, n), np.float32))
        eng.run_section("12a_clear_pressures_1")
        eng.run_section("12b_clear_pressures_2")
        eng.solve_pressure(10)
        assert np.all(eng.download_image(E.PRESSURES_1) == 1.0)
        assert np.all(eng.download_image(E.PRESSURES_2) == 1.0)


# ---- the ABI's error behaviour on a live context ----------------------------------------------------
def test_error_codes_and_messages():
    p = default_params(16, 16, 16, 10)
    with fluid_amd.FluidEngine(p, particle_capacity=10) as eng:
        with pytest.raises(fluid_amd.FluidEngineError) as ei:
            eng.upload_image(E.PRESSURES_1, np.zeros(7, np.float32))
        assert ei.value.code == E.ERR_SIZE_MISMATCH
        with pytest.raises(fluid_amd.FluidEngineError) as ei:
            eng.run_section(99)
        assert ei.value.code == E.ERR_INVALID_ARG
        with pytest.raises(fluid_amd.FluidEngineError) as ei:
            eng.download_image(E.DETAILED_DENSITIES_IMG)
        assert ei.value.code == E.ERR_UNSUPPORTED
        with pytest.raises(fluid_amd.FluidEngineError) as ei:
            eng.run_section_loop("11_compute_divergence", 3)
        assert ei.value.code == E.ERR_INVALID_ARG and "loop" in str(ei.value)
        blob = eng.download_params()
        assert blob.to_bytes() == p.to_bytes()


def test_timing_counts_sections():
    p, cap = dam_break_params(32, 32, 32)
    with fluid_amd.FluidEngine(p, particle_capacity=cap, pressure_iterations=6) as eng:
        eng.enable_timing(True)
        eng.run_init()
        eng.run_step()
        eng.run_step()
        times = eng.section_times()
        assert times["12_solve_pressure"][1] == 12
        assert times["07_advect"][1] == 2 and times["07_advect"][0] > 0
        assert times["00_init_particles"][1] == 1
        eng.reset_timing()
        assert eng.section_time_ms("07_advect") == (0.0, 0)


def test_cpp_section_list_driver_matches_oracle(tmp_path):
    """The C++ host mirror of the reference's section lists (include/fluid_flow_sections_amd.hpp)
    driven in main.cpp's call order by host/fluid_sim: init list once, step list per frame."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "vulkan-3d-fluid-simulation_amd", "host", "fluid_sim")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.dirname(exe)], check=True)
    size, frames, iters = (32, 32, 32), 2, 10
    res = subprocess.run([exe, *map(str, size), str(frames), str(iters), str(tmp_path)],
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "17 sections" in res.stdout  # SimulationStepSections 01a…14
    p, cap = dam_break_params(*size)
    st = OracleState(p, cap, iters)
    st.run_init()
    for _ in range(frames):
        st.run_step()
    for name, dtype in [("velocities_1", np.float32), ("cell_types", np.uint8),
                        ("pressures_1", np.float32), ("pressures_2", np.float32),
                        ("particles", np.float32)]:
        got = np.fromfile(os.path.join(str(tmp_path), name + ".bin"), dtype=dtype)
        assert_bit_equal(got.reshape(getattr(st, name).shape), getattr(st, name), f"C++ {name}")


def test_clear_image_arbitrary_value():
    st = random_state((24, 20, 16), seed=2)
    with make_engine(st) as eng:
        eng.clear_image(E.VELOCITIES_2, (1.5, -2.0, 0.25, 9.0))
        eng.clear_image(E.NEW_CELL_TYPES, 3)
        eng.clear_image(E.DIVERGENCES, -0.75)
        eng.clear_image(E.PARTICLE_DENSITIES_IMG, 41)
        st.velocities_2[...] = (1.5, -2.0, 0.25, 9.0)
        st.new_cell_types[...] = 3
        st.divergences[...] = -0.75
        st.particle_densities[...] = 41
        assert_state_equal(eng, st)


# ---- the loop-section fast path of 12_solve_pressure (kernels_pressure.h, kernels_pressure_fused.h) ----
@pytest.mark.parametrize("variant", [0, 5, 6, 7])
@pytest.mark.parametrize("size", [(24, 20, 16), (64, 64, 64), (260, 12, 9), (512, 7, 3), (256, 16, 8)])
@pytest.mark.parametrize("iters", [1, 2, 9])
@pytest.mark.parametrize("fuse", [0, 1])
def test_pressure_canonical_path_matches_oracle(variant, size, iters, fuse):
    """The loop section on working buffers after the two clears; random types (all four kinds, no
    solid shell) and random divergences; single-sweep and two-sweeps-per-pass schedules."""
    st = random_state(size, seed=3 * iters + variant, solid_walls=(variant % 2 == 0))
    with make_engine(st) as eng:
        eng.set_option(E.OPT_PRESSURE_KERNEL, variant)
        eng.set_option(E.OPT_JACOBI_FUSE, fuse)
        for name in ("12a_clear_pressures_1", "12b_clear_pressures_2"):
            eng.run_section(name)
            st.run_section(name)
        eng.solve_pressure(iters)
        st.solve_pressure(iters)
        assert_state_equal(eng, st, fields=["pressures_1", "pressures_2"],
                           ctx=f"canon x{iters} variant {variant} {size}: ")
        # the loop can be continued, and single dispatches interleave with it
        eng.run_pressure_dispatch(1)
        st.pressure_dispatch(1)
        eng.solve_pressure(3)
        st.solve_pressure(3)
        assert_state_equal(eng, st, fields=["pressures_1", "pressures_2"], ctx="continued: ")


def test_pressure_fast_path_invalidation():
    """Everything that can make the cached mask / b_i / canonical flags stale."""
    size = (64, 24, 12)
    st = random_state(size, seed=77, solid_walls=False)
    with make_engine(st) as eng:
        eng.set_option(E.OPT_PRESSURE_KERNEL, 5)

        def clears():
            for name in ("12a_clear_pressures_1", "12b_clear_pressures_2"):
                eng.run_section(name)
                st.run_section(name)

        def check(ctx):
            assert_state_equal(eng, st, fields=["pressures_1", "pressures_2"], ctx=ctx + ": ")

        clears()
        eng.solve_pressure(3)
        st.solve_pressure(3)
        check("baseline")
        # new divergence: b_i must be rebuilt
        st.divergences[...] = np.random.default_rng(1).uniform(-1, 1, st.shape).astype(np.float32)
        eng.upload_image(E.DIVERGENCES, st.divergences)
        clears()
        eng.solve_pressure(2)
        st.solve_pressure(2)
        check("new divergence")
        # new cell types while the pressures hold sweep results: no longer canonical -> general path
        rng = np.random.default_rng(2)
        st.cell_types[...] = rng.integers(0, 4, st.shape).astype(np.uint8)
        eng.upload_image(E.CELL_TYPES, st.cell_types)
        eng.solve_pressure(2)
        st.solve_pressure(2)
        check("types changed, stale pressures")
        clears()
        eng.solve_pressure(4)
        st.solve_pressure(4)
        check("types changed, after clears")
        # uploaded pressures are not canonical
        st.pressures_1[...] = rng.uniform(0, 2, st.shape).astype(np.float32)
        eng.upload_image(E.PRESSURES_1, st.pressures_1)
        eng.solve_pressure(3)
        st.solve_pressure(3)
        check("uploaded P1")
        # clear_image with p_air is as good as the section; with another value it is not
        eng.clear_image(E.PRESSURES_1, 1.0)
        eng.clear_image(E.PRESSURES_2, 1.0)
        st.pressures_1[...] = 1.0
        st.pressures_2[...] = 1.0
        eng.solve_pressure(3)
        st.solve_pressure(3)
        check("clear_image p_air")
        eng.clear_image(E.PRESSURES_1, 0.5)
        st.pressures_1[...] = 0.5
        eng.solve_pressure(3)
        st.solve_pressure(3)
        check("clear_image 0.5")
        # parameters: p_air, dt, rho, dx and a permutation of the type values
        p2 = st.params.copy()
        p2.pressure_air, p2.time_delta, p2.fluid_density, p2.cell_width = 0.75, 0.02, 1.5, 0.5
        p2.cell_type_inactive, p2.cell_type_air, p2.cell_type_water, p2.cell_type_solid = 3, 2, 1, 0
        eng.set_params(p2)
        st.params = p2
        st._p = __import__("ctypes").byref(st.params)
        clears()
        eng.solve_pressure(3)
        st.solve_pressure(3)
        check("new params (solid == 0: fast path must refuse)")
        p3 = p2.copy()
        p3.cell_type_solid, p3.cell_type_inactive = 9, 0
        st.cell_types[st.cell_types == 0] = 9
        eng.upload_image(E.CELL_TYPES, st.cell_types)
        eng.set_params(p3)
        st.params = p3
        st._p = __import__("ctypes").byref(st.params)
        clears()
        eng.solve_pressure(3)
        st.solve_pressure(3)
        check("new params (solid == 9)")
        # the divergence section itself invalidates b_i
        eng.upload_image(E.VELOCITIES_1, st.velocities_1)
        eng.run_section("11_compute_divergence")
        st.run_section("11_compute_divergence")
        clears()
        eng.solve_pressure(2)
        st.solve_pressure(2)
        check("after section 11")


@pytest.mark.parametrize("size", [(64, 40, 24), (256, 13, 9), (512, 19, 35), (260, 6, 5), (768, 7, 6),
                                  (1024, 5, 4), (8, 8, 8), (1280, 6, 5), (256, 50, 11), (512, 29, 7)])
@pytest.mark.parametrize("iters", [2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13])
@pytest.mark.parametrize("fuse", [0, 2])
def test_pressure_fused_pairs_match_oracle(size, iters, fuse):
    """Several sweeps per pass (kernels_pressure_fused.h: two; kernels_pressure_fused3.h: three, the default
    on grids up to 512 cells wide; fuse = 2 limits the loop to pairs): every case of the loop schedule — every
    residue of the sweep count modulo 3 and 2, so launches of three followed by one or two of two, a kept
    next-to-last iterate from either kind, odd tails — on grids with 1, 2 and 4 x-tiles, ragged row groups and
    z chunks, random cell types with no solid shell."""
    st = random_state(size, seed=iters, solid_walls=False, water_fraction=0.6)
    with make_engine(st) as eng:
        eng.set_option(E.OPT_JACOBI_FUSE, fuse)
        w, _, d = size
        assert eng.pressure_loop_max_sweeps() == (3 if fuse == 0 and w % 4 == 0 and w <= 512 and d >= 3 else
                                                  2 if w % 4 == 0 and w <= 1024 and d >= 2 else 1)
        for name in ("12a_clear_pressures_1", "12b_clear_pressures_2"):
            eng.run_section(name)
            st.run_section(name)
        eng.solve_pressure(iters)
        st.solve_pressure(iters)
        assert_state_equal(eng, st, fields=["pressures_1", "pressures_2"],
                           ctx=f"fused x{iters} {size}: ")
        # a second loop on top (third buffer already canonical), then the next section of the step
        eng.solve_pressure(iters + 1)
        st.solve_pressure(iters + 1)
        eng.run_section("13_fix_divergence")
        st.run_section("13_fix_divergence")
        assert_state_equal(eng, st, fields=["pressures_1", "pressures_2", "velocities_1"],
                           ctx=f"fused second loop {size}: ")


def test_pressure_fused_streaming_store_kernels_match_oracle():
    """The kernels with streaming stores (pressure_fused_stream.hip), which the launch code picks for working
    sets of 1.6 GB and more (512^3: covered by the property tests there): forced onto small grids of 2 and 4
    x-tiles in a child process (FLUID_FUSED_NT is read once per process), same test as above."""
    import subprocess
    import sys as _sys

    code = (
        "import sys; sys.path[:0] = [%r, %r]\n"
        "import test_engine_parity_gpu as T\n"
        "for size in [(512, 19, 35), (768, 7, 6), (1024, 5, 4), (512, 29, 7)]:\n"
        "    for iters in (2, 3, 5, 8, 9):\n"
        "        for fuse in (0, 2):\n"
        "            T.test_pressure_fused_pairs_match_oracle(size, iters, fuse)\n"
        "print('streaming ok')\n" % (ROOT, os.path.join(ROOT, "tests")))
    env = dict(os.environ, FLUID_FUSED_NT="1")
    out = subprocess.run([_sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "streaming ok" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


def test_pressure_fused_sparse_scene_and_bricks():
    """Mostly dry grid: the activity bricks let whole workgroups leave; results unchanged."""
    size = (256, 40, 48)
    st = random_state(size, seed=4, solid_walls=True)
    st.cell_types[...] = 1  # air
    st.cell_types[20:30, 10:22, 100:180] = CELL_WATER
    st.cell_types[3, 3, 3] = CELL_WATER
    st.cell_types[40:44, 30:39, 250:256] = CELL_WATER
    with make_engine(st) as eng:
        for name in ("12a_clear_pressures_1", "12b_clear_pressures_2"):
            eng.run_section(name)
            st.run_section(name)
        eng.solve_pressure(10)
        st.solve_pressure(10)
        assert_state_equal(eng, st, fields=["pressures_1", "pressures_2"], ctx="sparse fused: ")


@pytest.mark.parametrize("size", [(64, 24, 40), (512, 9, 21)])
@pytest.mark.parametrize("sweeps", [2, 3])
def test_pressure_split_passes_equal_whole_passes(size, sweeps):
    """fluid_pressure_loop_advance_part(_n): a pass of two or three sweeps as EDGES + INTERIOR launches, in
    either order, equals the whole pass (and the oracle); misuse is reported."""
    st = random_state(size, seed=9, iters=8)
    d = size[2]
    with make_engine(st) as eng:
        eng.pressure_loop_begin()
        assert eng.pressure_loop_max_sweeps() == 3
        edges, interior = eng.LOOP_PART_EDGES, eng.LOOP_PART_INTERIOR
        # pass 1: edges first; pass 2: interior first; pass 3: interior covers everything; pass 4:
        # empty interior, keeps the next-to-last iterate (last launch of the loop)
        plans = [(False, edges, 5, d - 5), (False, interior, 2, d - 2),
                 (False, interior, -2 ** 31, 2 ** 31 - 1), (True, edges, 7, 7)]
        for keep, first, lo, hi in plans:
            other = interior if first == edges else edges
            w1 = eng.pressure_loop_advance_part(keep, first, lo, hi, sweeps)
            with pytest.raises(fluid_amd.FluidEngineError, match="other part"):
                eng.pressure_loop_advance_part(keep, first, lo, hi, sweeps)       # same part twice
            with pytest.raises(fluid_amd.FluidEngineError, match="other part"):
                eng.pressure_loop_advance_part(not keep, other, lo, hi, sweeps)  # other arguments
            with pytest.raises(fluid_amd.FluidEngineError, match="other part"):
                eng.pressure_loop_advance_part(keep, other, lo, hi, 5 - sweeps)  # another number of sweeps
            with pytest.raises(fluid_amd.FluidEngineError, match="half done"):
                eng.pressure_loop_advance(2, keep)
            with pytest.raises(fluid_amd.FluidEngineError, match="half done"):
                eng.pressure_loop_end()
            assert eng.pressure_loop_advance_part(keep, other, lo, hi, sweeps) == w1
        with pytest.raises(fluid_amd.FluidEngineError, match="cannot advance by 4"):
            eng.pressure_loop_advance(4, False)
        eng.pressure_loop_end()
        st.solve_pressure(4 * sweeps)
        assert_state_equal(eng, st, fields=["pressures_1", "pressures_2"], ctx="split passes: ")


@pytest.mark.parametrize("iters", [2, 3, 6])
@pytest.mark.parametrize("tiny", [True, False])
def test_pressure_division_special_values(iters, tiny):
    """The fused kernel replaces the IEEE division n / aii by a three-instruction exact quotient plus
    v_div_fixup_f32 (kernels_pressure_fused.h).  Isolated water cells with every neighbour count
    aii = 0..6 and numerators that are zeros, infinities, NaN, denormals, the extremes of the normal
    range and values on both sides of the kernel's 2^-100 guard: bit-identical to the oracle.  A wavefront
    that holds a tiny non-zero numerator takes the IEEE sequence for all its lanes, so the second variant
    leaves those out: zeros, infinities, NaN and the extremes then go through the short quotient."""
    w, h, d = 64, 48, 12
    p = default_params(w, h, d, 0)
    p.time_delta = 1.0      # b_i = ((div * rho) * dx) / dt = div exactly
    p.cell_width = 1.0
    p.fluid_density = 1.0
    p.pressure_air = 0.0    # dry neighbours contribute 0: the numerator is -b_i
    st = OracleState(p, 0, iters)
    st.cell_types[...] = CELL_AIR
    specials = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 1e-40, -3e-39,
                         2.0 ** -126, -(2.0 ** -126), 1.7e-38, 2.0 ** -100, 2.0 ** -91, -(2.0 ** -90),
                         2.0 ** -89, 3.0, -7.0, 1.0 / 3.0, 3.4028234e38, -3.4028234e38, 1e30, 5e-324,
                         2.0 ** -101, -(2.0 ** -100) * 1.5, 2.0 ** -99, -(2.0 ** -100) * (1 - 2.0 ** -24)],
                        np.float32)
    if not tiny:
        specials = specials[(specials == 0) | ~(np.abs(specials) < 2.0 ** -100)]   # keeps NaN and inf
    solid_dirs = [(0, 0, 1), (0, 0, -1), (0, 1, 0), (0, -1, 0), (1, 0, 0), (-1, 0, 0)]
    k = 0
    cells = []
    for z in range(2, d - 2, 3):
        for y in range(2, h - 2, 3):
            for x in range(2, w - 2, 3):
                nsolid = k % 7
                st.cell_types[z, y, x] = CELL_WATER
                for dz, dy, dx in solid_dirs[:nsolid]:
                    st.cell_types[z + dz, y + dy, x + dx] = CELL_SOLID
                st.divergences[z, y, x] = specials[(k // 7) % len(specials)]
                cells.append((z, y, x))
                k += 1
    assert k >= 7 * len(specials)
    with make_engine(st) as eng, np.errstate(all="ignore"):
        for name in ("12a_clear_pressures_1", "12b_clear_pressures_2"):
            eng.run_section(name)
            st.run_section(name)
        eng.solve_pressure(iters)
        st.solve_pressure(iters)
        assert_state_equal(eng, st, fields=["pressures_1", "pressures_2"], ctx="special values: ")
    got = np.array([st.pressures_1[c] for c in cells[:7 * len(specials)]])
    assert np.isnan(got).any() and np.isinf(got).any() and (got == 0).any()


@pytest.mark.parametrize("quiet,size", [(0, (64, 64, 96)), (1, (64, 64, 96)), (2, (64, 64, 96)),
                                        (2, (64, 62, 88))])   # brick layers cut by the grid: 88 = 5.5 x 16, 62 = 15.5 x 4
def test_full_step_quiet_bricks_match_oracle(quiet, size):
    """Ten dam-break steps on a grid with room around the water: from the third step on fluid_run_step
    skips the bricks far from the water in 07+08, 09+10+11 and 13 (quiet_bricks.h).  Every image equals
    the oracle's after every step; the same with the skipping turned off."""
    p, cap = dam_break_params(*size)
    iters = 6
    st = OracleState(p, cap, iters)
    with fluid_amd.FluidEngine(p, particle_capacity=cap, pressure_iterations=iters) as eng:
        eng.set_option(E.OPT_QUIET_BRICKS, quiet)
        eng.run_init()
        st.run_init()
        skipped = []
        for k in range(10):
            eng.run_step()
            st.run_step()
            assert_state_equal(eng, st, ctx=f"quiet={quiet} step {k}: ")
            skipped.append(eng.get_stat(E.STAT_QUIET_BRICKS))
        total = eng.get_stat(E.STAT_BRICKS)
        if quiet != 1:   # 2: the same skipping with one workgroup per brick layer (large grids' default)
            assert skipped[0] == 0 and skipped[1] == 0           # streaks build up first
            assert 0.3 * total < skipped[-1] < total             # most of this grid is far from the water
        else:
            assert skipped[-1] == 0
        # a write from outside resets the streaks: the next step processes everything
        eng.upload_image(E.VELOCITIES_1, st.velocities_1)
        eng.run_step()
        st.run_step()
        assert_state_equal(eng, st, ctx=f"quiet={quiet} after upload: ")
        assert eng.get_stat(E.STAT_QUIET_BRICKS) == 0


@pytest.mark.parametrize("size", [(64, 33, 17), (260, 32, 16)])
def test_lone_particles_in_wall_cells_are_cleared_every_step(size):
    """A particle alone in a border cell: 02 makes the cell WATER, 03 makes it SOLID again, so its brick
    never holds a water cell (with H = 33 / D = 17 the border planes y = 32 and z = 16 are bricks of their
    own).  The reference clears the whole density image every step (fluid_flow_sections.h:163); the
    engine's brick-wise 01a must clear these cells too — also after the particle has moved on (a steady
    downward velocity carries one of them out of the wall into the open grid)."""
    w, h, d = size
    cap = 8
    p = default_params(w, h, d, cap)
    iters = 4
    st = OracleState(p, cap, iters)
    st.run_section("init_clear_velocities_1")
    st.run_section("init_clear_cell_types")
    st.velocities_1[..., 1] = -70.0          # 0.7 cells per step towards -y, everywhere
    st.particles[:] = 0.0
    st.particles[0] = (10.5, h - 0.1, 5.5, 1.0)     # ceiling cell, leaves it after two steps
    st.particles[1] = (20.5, 10.5, d - 0.5, 1.0)    # +z wall
    st.particles[2] = (0.5, 7.5, 3.5, 1.0)          # -x wall
    st.particles[3] = (w - 0.5, h - 0.5, d - 0.5, 1.0)  # corner
    with make_engine(st) as eng:
        for k in range(6):
            eng.run_step()
            st.run_step()
            assert_state_equal(eng, st, ctx=f"wall droplets step {k}: ")
        assert st.particle_densities.max() == 1      # never accumulates
        assert st.particles[0, 1] < h - 1            # particle 0 has left the ceiling cell
        assert eng.get_stat(E.STAT_QUIET_BRICKS) > 0  # the brick-wise path was the one running


def _numpy_residual(st, pimg):
    """r = s + aii * P per WATER cell in fp32, pressure.comp:54-61 order (include/fluid_engine.h:
    fluid_pressure_residual)."""
    p = st.params
    f = np.float32
    t = st.cell_types
    d, h, w = t.shape
    tp = np.zeros((d + 2, h + 2, w + 2), np.uint8)       # out of bounds reads as type 0
    tp[1:-1, 1:-1, 1:-1] = t
    pp = np.zeros((d + 2, h + 2, w + 2), f)
    pp[1:-1, 1:-1, 1:-1] = pimg
    s = ((st.divergences * f(p.fluid_density)) * f(p.cell_width)) / f(p.time_delta)
    aii = np.zeros(t.shape, np.int32)
    for dz, dy, dx in [(0, 0, 1), (0, 1, 0), (1, 0, 0), (0, 0, -1), (0, -1, 0), (-1, 0, 0)]:
        ty = tp[1 + dz:1 + dz + d, 1 + dy:1 + dy + h, 1 + dx:1 + dx + w]
        q = pp[1 + dz:1 + dz + d, 1 + dy:1 + dy + h, 1 + dx:1 + dx + w]
        live = ty != p.cell_type_solid
        contrib = np.where(ty == p.cell_type_water, q, f(p.pressure_air)).astype(f)
        s = np.where(live, (s - contrib).astype(f), s).astype(f)
        aii += live
    r = (s + (aii.astype(f) * pimg).astype(f)).astype(f)
    return r[t == p.cell_type_water]


@pytest.mark.parametrize("size", [(24, 20, 16), (64, 9, 7), (17, 13, 9)])
def test_pressure_residual_readout(size):
    st = random_state(size, seed=13, iters=30)
    with make_engine(st) as eng, np.errstate(all="ignore"):
        for img, field in ((E.PRESSURES_1, "pressures_1"), (E.PRESSURES_2, "pressures_2")):
            r = _numpy_residual(st, getattr(st, field))
            mx, ss, n = eng.pressure_residual(img)
            assert n == r.size and n > 0
            assert np.float32(mx).view(np.uint32) == np.abs(r).max().view(np.uint32)   # exact
            assert abs(ss - float(np.sum(r.astype(np.float64) ** 2))) <= 1e-9 * ss
        before = eng.pressure_residual(E.PRESSURES_2)[0]
        eng.solve_pressure(30)   # Jacobi on a diagonally dominant system: the residual shrinks
        st.solve_pressure(30)
        after = eng.pressure_residual(E.PRESSURES_2)[0]
        r = _numpy_residual(st, st.pressures_2)
        assert np.float32(after).view(np.uint32) == np.abs(r).max().view(np.uint32)
        assert after < before
        with pytest.raises(fluid_amd.FluidEngineError, match="not a pressure image"):
            eng.pressure_residual(E.DIVERGENCES)


def test_pressure_512cubed_default_equals_plain_and_window_matches_oracle():
    """512^3 full-fluid grid (BASELINE config C4, the size the metric is quoted on), 8 sweeps: the
    default path (launches of 3 + 3 + 2 sweeps per pass, division-free quotient, chunked z march) and the
    pairs-only schedule agree bit for bit with the one-thread-per-cell kernel on the images everywhere, and
    a 20-plane window recomputed by the oracle matches outside its dependence cone."""
    n, iters = 512, 8
    p = default_params(n, n, n, 0)
    shape = (n, n, n)
    t = scenes.full_fluid_types(shape)
    div = scenes.full_fluid_divergence(shape)
    sums = {}
    keep = None
    for variant, fuse in ((1, 0), (0, 0), (0, 2)):
        with fluid_amd.FluidEngine(p, particle_capacity=0) as eng:
            eng.set_option(E.OPT_PRESSURE_KERNEL, variant)
            eng.set_option(E.OPT_JACOBI_FUSE, fuse)
            eng.upload_image(E.CELL_TYPES, t)
            eng.upload_image(E.DIVERGENCES, div)
            eng.run_section("12a_clear_pressures_1")
            eng.run_section("12b_clear_pressures_2")
            eng.solve_pressure(iters)
            p1, p2 = eng.download_image(E.PRESSURES_1), eng.download_image(E.PRESSURES_2)
        if keep is None:
            keep = (p1, p2)
        else:
            assert_bit_equal(p1, keep[0], f"512^3 P1 default (fuse option {fuse}) vs plain")
            assert_bit_equal(p2, keep[1], f"512^3 P2 default (fuse option {fuse}) vs plain")
        sums[variant] = int(p1.view(np.uint32).astype(np.uint64).sum())
    z0, zc = 300, 20
    pw = default_params(n, n, zc, 0)
    sw = OracleState(pw, 0, iters)
    sw.cell_types[...] = t[z0:z0 + zc]
    sw.divergences[...] = div[z0:z0 + zc]
    sw.pressures_1[...] = 1.0
    sw.pressures_2[...] = 1.0
    sw.solve_pressure(iters)
    lo, hi = iters, zc - iters
    assert_bit_equal(keep[0][z0 + lo:z0 + hi], sw.pressures_1[lo:hi], "512^3 window P1")
    assert_bit_equal(keep[1][z0 + lo:z0 + hi], sw.pressures_2[lo:hi], "512^3 window P2")


def c5_scene(shape, z_begin=0, global_depth=None, seed=5):
    """Full-fluid scene at the C5 shape (walls SOLID, the rest WATER); the divergence comes from numpy's
    generator (seeded per plane, so a slab sees the planes of the whole grid) instead of SplitMix64 to
    keep half a billion cells cheap."""
    d, h, w = shape
    t = scenes.full_fluid_types(shape, z_begin, global_depth)
    div = np.empty(shape, np.float32)
    for z in range(d):
        rng = np.random.default_rng([seed, z_begin + z])
        div[z] = rng.random((h, w), dtype=np.float32) * np.float32(2.0) - np.float32(1.0)
    return t, div


def test_c5_shape_1024x1024x512_default_equals_plain_and_window_matches_oracle():
    """BASELINE.json configs[4] on one GPU (25 GiB of attachments): 1024 x 1024 x 512 full-fluid grid,
    6 sweeps.  W = 1024 is the four-x-tile instantiation of the two-sweeps-per-pass kernel.  The default
    path agrees bit for bit with the one-thread-per-cell kernel on the images everywhere, and 20-plane
    windows recomputed by the oracle (both domain faces and the middle) match outside their dependence
    cones."""
    w, h, d, iters = 1024, 1024, 512, 6
    p = default_params(w, h, d, 0)
    t, div = c5_scene((d, h, w))
    keep = None
    with fluid_amd.FluidEngine(p, particle_capacity=0) as eng:
        eng.upload_image(E.CELL_TYPES, t)
        eng.upload_image(E.DIVERGENCES, div)
        for variant in (0, 1):
            eng.set_option(E.OPT_PRESSURE_KERNEL, variant)
            eng.run_section("12a_clear_pressures_1")
            eng.run_section("12b_clear_pressures_2")
            eng.solve_pressure(iters)
            p1, p2 = eng.download_image(E.PRESSURES_1), eng.download_image(E.PRESSURES_2)
            if keep is None:
                keep = (p1, p2)
            else:
                assert_bit_equal(p1, keep[0], "C5 P1 default vs plain")
                assert_bit_equal(p2, keep[1], "C5 P2 default vs plain")
            del p1, p2
    for z0 in (0, 246, d - 20):
        zc = 20
        pw = default_params(w, h, zc, 0)
        sw = OracleState(pw, 0, iters)
        sw.cell_types[...] = t[z0:z0 + zc]
        sw.divergences[...] = div[z0:z0 + zc]
        sw.pressures_1[...] = 1.0
        sw.pressures_2[...] = 1.0
        sw.solve_pressure(iters)
        lo = 0 if z0 == 0 else iters
        hi = zc if z0 + zc == d else zc - iters
        assert_bit_equal(keep[0][z0 + lo:z0 + hi], sw.pressures_1[lo:hi], f"C5 window z0={z0} P1")
        assert_bit_equal(keep[1][z0 + lo:z0 + hi], sw.pressures_2[lo:hi], f"C5 window z0={z0} P2")


def test_full_step_256cubed_grouped_and_quiet_equal_the_section_list():
    """256^3 dam break (BASELINE config C3 shape), six whole steps: fluid_run_step with grouped passes,
    quiet bricks and box-shaped launches against the plain section list (one kernel per section, every
    cell processed) — two code paths through the engine, every image and the particles bit-identical."""
    n, iters, steps = 256, 12, 6
    p, cap = dam_break_params(n, n, n)
    with fluid_amd.FluidEngine(p, particle_capacity=cap, pressure_iterations=iters) as a, \
            fluid_amd.FluidEngine(p, particle_capacity=cap, pressure_iterations=iters) as b:
        b.set_option(E.OPT_STEP_FUSION, 1)
        b.set_option(E.OPT_JACOBI_FUSE, 1)
        a.run_init()
        b.run_init()
        for _ in range(steps):
            a.run_step()
            b.run_step()
        assert a.get_stat(E.STAT_QUIET_BRICKS) > 0 and b.get_stat(E.STAT_QUIET_BRICKS) == 0
        for name, img in IMAGE_FIELDS.items():
            assert_bit_equal(a.download_image(img), b.download_image(img), f"256^3 {name}")
        assert_bit_equal(a.download_particles(), b.download_particles(), "256^3 particles")


@pytest.mark.parametrize("size,xr,iters", [((768, 12, 20), (256, 512), 16), ((1024, 10, 12), (256, 768), 17),
                                            ((512, 9, 8), (0, 256), 16), ((512, 9, 8), (256, 512), 18),
                                            ((1024, 6, 5), (512, 560), 16)])
def test_pressure_fused_x_window_launches(size, xr, iters):
    """Sparse scene whose water spans one or two 256-cell columns of a wider grid: the loop launches
    the two-sweeps kernel over that x window only (FusedRange::xwin0); the cells next to the window —
    air, solid and inactive ones, at the grid edge too — reach the kernel through the pad loads."""
    w, h, d = size
    st = random_state(size, seed=8, iters=iters)
    rng = np.random.default_rng(3)
    # no water outside [xr[0], xr[1]); everything else stays random (solids and air next to the window)
    outside = np.ones(w, bool)
    outside[xr[0]:xr[1]] = False
    wet_outside = (st.cell_types == CELL_WATER) & outside[None, None, :]
    st.cell_types[wet_outside] = rng.choice(np.array([CELL_AIR, CELL_SOLID], np.uint8),
                                            size=int(wet_outside.sum()))
    # water right up to both window edges in some rows
    st.cell_types[2:d - 2, 2:h - 2, xr[0]] = CELL_WATER
    st.cell_types[2:d - 2, 2:h - 2, xr[1] - 1] = CELL_WATER
    if xr[0] == 0 or xr[1] == w:   # keep the domain faces solid, as a step would
        st.cell_types[:, :, 0] = CELL_SOLID
        st.cell_types[:, :, -1] = CELL_SOLID
    with make_engine(st) as eng:
        eng.solve_pressure(iters)
        st.solve_pressure(iters)
        assert_state_equal(eng, st, fields=["pressures_1", "pressures_2"], ctx=f"x window {xr}: ")


@pytest.mark.parametrize("kernel", [0, 1])
@pytest.mark.parametrize("size,dt,scale", [((64, 12, 9), 0.01, 3.0), ((24, 20, 16), 0.3, 3.0),
                                           ((260, 6, 5), 2.0, 3.0), ((17, 13, 9), 1.0, 8.0)])
def test_advect_tiled_sampler_and_fallback(kernel, size, dt, scale):
    """07_advect with the LDS-tiled sampler (default) and with direct loads, for back-traces that stay in
    the tile (small dt * v), that leave it (dt * v of several cells: global fallback) and that leave the
    grid (clamp to edge)."""
    st = random_state(size, seed=17, velocity_scale=scale)
    st.params.time_delta = dt
    with make_engine(st) as eng:
        eng.set_option(E.OPT_ADVECT_KERNEL, kernel)
        eng.run_section("07_advect")
        st.run_section("07_advect")
        assert_state_equal(eng, st, ctx=f"07 kernel {kernel} dt {dt}: ")
        eng.run_section_group("07_advect", 2)
        st.run_section("07_advect")
        st.run_section("08_forces")
        assert_state_equal(eng, st, ctx=f"07+08 kernel {kernel} dt {dt}: ")


@pytest.mark.parametrize("fraction", [0.0, 0.002, 0.05])
@pytest.mark.parametrize("size", [(64, 8, 32), (128, 16, 64)])
def test_advect_face_shortcut_and_special_values(size, fraction):
    """On a grid whose extents are all powers of two the tiled 07 kernel takes the three face-position
    samples of advect.comp:75 without the sampler's arithmetic — unless the planes it reads hold a value
    for which `(1 - 0) * A + 0 * B` is not A: -0, a denormal (0.5 * A may underflow to -0), inf, NaN.  Those
    are sprinkled over VELOCITIES_1 here (whole planes stay clean at the low fraction, so both paths run in
    one launch); every texel must equal the oracle's bit for bit (a generated NaN matches any NaN: helpers)."""
    st = random_state(size, seed=77, iters=2, velocity_scale=2.0)
    rng = np.random.default_rng(5)
    v = st.velocities_1
    specials = np.array([-0.0, 1e-40, -3e-42, np.inf, -np.inf, np.nan, 0.0], np.float32)
    if fraction:
        hit = rng.uniform(0, 1, v.shape) < fraction
        if fraction < 0.01:
            hit[: v.shape[0] // 2] = False     # the lower half of the planes stays regular
        v[hit] = rng.choice(specials, size=int(hit.sum()))
    with make_engine(st) as eng, np.errstate(all="ignore"):
        for kernel in (0, 1):   # LDS-tiled with the shortcut; taps straight from global memory
            eng.set_option(E.OPT_ADVECT_KERNEL, kernel)
            eng.run_section("07_advect")
            st.run_section("07_advect")
            assert_bit_equal_any_nan(eng.download_image(E.VELOCITIES_2), st.velocities_2,
                                     f"07 kernel {kernel}, special fraction {fraction}")
            eng.run_section_group("07_advect", 2)
            st.run_section("07_advect")
            st.run_section("08_forces")
            assert_bit_equal_any_nan(eng.download_image(E.VELOCITIES_2), st.velocities_2,
                                     f"07+08 kernel {kernel}, special fraction {fraction}")


def test_checkpoint_round_trip(tmp_path):
    """save_checkpoint / restore_checkpoint: a run resumed from the file continues bit-identically — the
    file carries the run state too (a non-default iteration count and 09_diffuse mode here), and a path
    without the .npz suffix names the same file on both sides."""
    p, cap = dam_break_params(32, 24, 16)
    p.diffuse_k = 0.8
    path = str(tmp_path / "state")          # no suffix
    with fluid_amd.FluidEngine(p, particle_capacity=cap, pressure_iterations=9, surface_prep=True) as a:
        a.set_diffuse_mode(E.DIFFUSE_INTENDED)
        a.run_init()
        for _ in range(3):
            a.run_step()
        a.save_checkpoint(path)
        for _ in range(3):
            a.run_step()
        ref = {img: a.download_image(img) for img in list(E.IMAGE_DTYPES) + list(E.SURFACE_DTYPES)}
        ref_particles = a.download_particles()
    import os
    assert os.path.exists(path + ".npz") and not os.path.exists(path)
    # the resuming context is created with other settings: the file's replace them
    with fluid_amd.FluidEngine(p, particle_capacity=cap, pressure_iterations=200, surface_prep=True) as b:
        b.restore_checkpoint(path)
        assert b.pressure_iterations == 9 and b.diffuse_mode == E.DIFFUSE_INTENDED
        for _ in range(3):
            b.run_step()
        for img, exp in ref.items():
            assert_bit_equal(b.download_image(img), exp, f"resumed image {img}")
        assert_bit_equal(b.download_particles(), ref_particles, "resumed particles")
    with fluid_amd.FluidEngine(p, particle_capacity=cap + 1) as c:
        with pytest.raises(fluid_amd.FluidEngineError, match="does not fit"):
            c.restore_checkpoint(path + ".npz")
    with fluid_amd.FluidEngine(p, particle_capacity=cap, surface_prep=True, surface_diffuse_steps=6) as d:
        with pytest.raises(fluid_amd.FluidEngineError, match="blur dispatches"):
            d.restore_checkpoint(path)


@pytest.mark.parametrize("size,omega,iters", [((24, 20, 16), 1.5, 5), ((17, 13, 9), 1.0, 3),
                                              ((64, 12, 9), 1.9, 8)])
def test_red_black_sor_solver_matches_oracle(size, omega, iters):
    """The opt-in red-black SOR solver (fluid_set_pressure_solver) against the oracle's
    sequential restatement: bit-identical, as a loop (P2 := P1 afterwards) and as single iterations."""
    st = random_state(size, seed=23, iters=iters)
    st.sor_omega = omega
    with make_engine(st) as eng:
        eng.set_pressure_solver(eng.SOLVER_RED_BLACK_SOR, omega)
        eng.solve_pressure(iters)
        st.solve_pressure(iters)
        assert_state_equal(eng, st, ctx=f"SOR loop omega {omega}: ")
        from oracle_binding import lib as oracle_lib, _ptr
        import ctypes as C
        eng.run_section("12_solve_pressure")   # one more iteration, in place on PRESSURES_1
        oracle_lib().oracle_12_sor_iteration(C.byref(st.params), _ptr(st.cell_types), _ptr(st.divergences),
                                             _ptr(st.pressures_1), omega)
        assert_state_equal(eng, st, fields=["pressures_1", "pressures_2"], ctx="SOR single iteration: ")
        with pytest.raises(fluid_amd.FluidEngineError, match="omega"):
            eng.set_pressure_solver(eng.SOLVER_RED_BLACK_SOR, 2.0)
        eng.set_pressure_solver(eng.SOLVER_JACOBI)
        eng.solve_pressure(2)                  # back to the reference's loop
        st.sor_omega = None
        st.solve_pressure(2)
        assert_state_equal(eng, st, fields=["pressures_1", "pressures_2"], ctx="back to Jacobi: ")


def test_red_black_sor_full_steps_and_convergence():
    """Whole dam-break steps with the SOR solver equal the oracle's; and the reason to have it: on a
    pool under a free surface 40 SOR iterations leave a far smaller residual than 200 Jacobi sweeps."""
    size, iters = (32, 32, 32), 12
    p, cap = dam_break_params(*size)
    st = OracleState(p, cap, iters)
    st.sor_omega = 1.7
    with fluid_amd.FluidEngine(p, particle_capacity=cap, pressure_iterations=iters) as eng:
        eng.set_pressure_solver(eng.SOLVER_RED_BLACK_SOR, 1.7)
        eng.run_init()
        st.run_init()
        for k in range(4):
            eng.run_step()
            st.run_step()
            assert_state_equal(eng, st, ctx=f"SOR step {k}: ")
    # a pool under a free surface (walls SOLID, lower half WATER, AIR above) with a uniform source term:
    # a smooth problem whose error Jacobi removes at O(depth^2) sweeps.  (The residual is the wrong
    # yardstick here: over-relaxation keeps it large while the error collapses.)
    n = 32
    pp = default_params(n, n, n, 0)
    t = np.full((n, n, n), CELL_AIR, np.uint8)
    t[:, n // 2:, :] = CELL_WATER
    t[0], t[-1], t[:, 0], t[:, -1], t[:, :, 0], t[:, :, -1] = (CELL_SOLID,) * 6
    div = np.full((n, n, n), 0.05, np.float32)
    sol = {}
    for name, solver, omega, its in (("converged", 1, 1.8, 3000), ("jacobi 200", 0, 1.0, 200),
                                     ("sor 100", 1, 1.8, 100)):
        with fluid_amd.FluidEngine(pp, particle_capacity=0) as eng:
            eng.upload_image(E.CELL_TYPES, t)
            eng.upload_image(E.DIVERGENCES, div)
            eng.run_section("12a_clear_pressures_1")
            eng.run_section("12b_clear_pressures_2")
            eng.set_pressure_solver(solver, omega)
            eng.solve_pressure(its)
            sol[name] = eng.download_image(E.PRESSURES_1).astype(np.float64)
    wet = t == CELL_WATER
    err = {k: np.abs(v - sol["converged"])[wet].max() for k, v in sol.items()}
    assert err["jacobi 200"] > 0.5 * np.abs(sol["converged"][wet]).max()   # 200 sweeps: not even close
    assert err["sor 100"] < 0.1 * err["jacobi 200"], err


@pytest.mark.parametrize("seed,size", [(1, (512, 48, 96)), (2, (768, 32, 64)), (3, (256, 48, 96)),
                                       (4, (1024, 24, 64))])
def test_moving_blob_all_step_optimisations_equal_the_section_list(seed, size):
    """A blob of water drifting through a wide, mostly empty grid for 14 steps: the default fluid_run_step
    (grouped passes, quiet bricks, early test, box-shaped and x-windowed Jacobi launches, LDS sampler)
    against the plain section list with every cell processed — images and particles bit-identical after
    every step while the blob crosses brick, window and tile boundaries."""
    w, h, d = size
    rng = np.random.default_rng(seed)
    cap = 6000
    p = default_params(w, h, d, cap)
    p.time_delta = 0.05
    p.particle_compute_size[:] = (cap, 1)
    centre = np.array([0.35 * w, 0.5 * h, 0.4 * d], np.float32)
    particles = np.zeros((cap, 4), np.float32)
    particles[:, :3] = centre + rng.uniform(-1, 1, (cap, 3)).astype(np.float32) * \
        np.array([min(40.0, 0.1 * w), 3.0, 4.0], np.float32)
    particles[:, 3] = 1.0
    drift = np.zeros((d, h, w, 4), np.float32)
    drift[..., 0] = 45.0 + 10.0 * seed     # cells per second along x: > 2 cells per step
    drift[..., 2] = 6.0
    with fluid_amd.FluidEngine(p, particle_capacity=cap, pressure_iterations=20) as a, \
            fluid_amd.FluidEngine(p, particle_capacity=cap, pressure_iterations=20) as b:
        if seed % 2 == 0:
            a.set_option(E.OPT_QUIET_BRICKS, 2)   # one workgroup per brick layer, as on grids of >= 4096 bricks
        b.set_option(E.OPT_STEP_FUSION, 1)
        b.set_option(E.OPT_JACOBI_FUSE, 1)
        b.set_option(E.OPT_ADVECT_KERNEL, 1)
        b.set_option(E.OPT_LAUNCH_BOX, 1)
        for eng in (a, b):
            eng.run_init()
            eng.upload_particles(particles)
            eng.run_step()                       # cells become active; 05 replaces their velocities
            eng.upload_image(E.VELOCITIES_1, drift)
        quiet_seen = 0
        for k in range(14):
            a.run_step()
            b.run_step()
            quiet_seen = max(quiet_seen, a.get_stat(E.STAT_QUIET_BRICKS))
            for name, img in IMAGE_FIELDS.items():
                assert_bit_equal(a.download_image(img), b.download_image(img), f"step {k} {name}")
            assert_bit_equal(a.download_particles(), b.download_particles(), f"step {k} particles")
        moved = a.download_particles()[:, 0].mean() - particles[:, 0].mean()
        assert moved > 10.0 and quiet_seen > 0   # the blob travelled, and bricks did go quiet
