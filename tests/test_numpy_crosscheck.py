"""The C oracle against an independently written numpy restatement of the shaders
(tests/numpy_restatement.py) on random scenes — two readings of the GLSL must agree bit for bit.  CPU."""
import numpy as np
import pytest

import numpy_restatement as R
from helpers import assert_bit_equal, random_state

SIZES = [(12, 10, 8), (9, 7, 11), (16, 16, 16)]


@pytest.mark.parametrize("size", SIZES)
@pytest.mark.parametrize("seed", [0, 1])
def test_sections_agree(size, seed):
    cap = 500
    st = random_state(size, capacity=cap, seed=seed, solid_walls=(seed == 0))
    p = st.params

    def check(section, expected, field):
        s2 = st.copy()
        s2.run_section(section)
        assert_bit_equal(getattr(s2, field), expected, f"{section} {size} seed {seed}: {field}")

    check("02_update_water", R.update_water(p, st.particle_densities), "new_cell_types")
    check("03_update_air", R.update_air(p, st.new_cell_types), "new_cell_types")
    check("04_compute_extrapolated_velocities",
          R.extrapolated_velocities(p, st.cell_types, st.velocities_1), "velocities_2")
    check("05_set_extrapolated_velocities",
          R.set_extrapolated_velocities(p, st.new_cell_types, st.cell_types, st.velocities_2,
                                        st.velocities_1), "velocities_1")
    check("07_advect", R.advect(p, st.cell_types, st.velocities_1), "velocities_2")
    check("08_forces", R.forces(p, st.cell_types, st.velocities_2), "velocities_2")
    check("10_solids", R.solids(p, st.cell_types, st.velocities_1), "velocities_1")
    check("11_compute_divergence", R.divergence(st.velocities_1), "divergences")
    check("13_fix_divergence", R.fix_divergence(p, st.cell_types, st.pressures_2, st.velocities_1),
          "velocities_1")
    check("14_particles", R.move_particles(p, st.velocities_1, st.particles), "particles")
    dens = st.particle_densities + R.update_densities(p, st.particles, st.shape)
    check("01_update_densities", dens, "particle_densities")


@pytest.mark.parametrize("size", SIZES)
def test_pressure_loop_agrees(size):
    st = random_state(size, seed=5, solid_walls=False)
    p = st.params
    p1, p2 = st.pressures_1.copy(), st.pressures_2.copy()
    with np.errstate(all="ignore"):
        for k in range(5):  # dispatch k reads P1 iff k is even
            if k % 2 == 0:
                p2 = R.pressure_sweep(p, st.cell_types, st.divergences, p1, p2)
            else:
                p1 = R.pressure_sweep(p, st.cell_types, st.divergences, p2, p1)
        st.solve_pressure(5)
    assert_bit_equal(st.pressures_1, p1, "P1 after 5 dispatches")
    assert_bit_equal(st.pressures_2, p2, "P2 after 5 dispatches")
