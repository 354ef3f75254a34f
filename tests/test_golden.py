"""The committed golden vectors (tests/golden/*.npz, written by tests/golden/make_golden.py from
the oracle) still equal what the oracle produces — guards the oracle against accidental change.
CPU only."""
import os

import numpy as np
import pytest

from helpers import assert_bit_equal
from oracle_binding import OracleState

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("name", sorted(f for f in os.listdir(GOLDEN)
                                        if f.endswith(".npz") and f != "marching_cubes_tables.npz"))  # data, not a vector
def test_oracle_reproduces_golden(name):
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLDEN, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    g = np.load(os.path.join(GOLDEN, name))
    out = mg.run_case(tuple(int(v) for v in g["size"]), int(g["iterations"]), int(g["steps"]))
    for f in OracleState.FIELDS:
        assert_bit_equal(out[f], g[f], f"{name}:{f}")
    assert int(out["capacity"]) == int(g["capacity"])
    # the scene is alive: water present, velocities and pressures moved
    assert np.count_nonzero(g["cell_types"] == 2) > 50
    assert np.any(g["pressures_1"] != 1.0) and np.any(g["velocities_1"][..., :3] != 0)
