"""Generate the golden vectors in this directory from the CPU oracle.

    python tests/golden/make_golden.py

The reference repository holds no golden vectors (and cannot run here), so these are outputs of
oracle/fluid_oracle.c on the seeded dam-break scene: every hot-path attachment
after N full steps.  They pin the oracle against accidental change (tests/test_golden.py, CPU) and
give the engine a fixture that does not need the oracle at run time (tests/test_engine_parity_gpu.py).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle_binding import OracleState  # noqa: E402
from fluid_amd.params import dam_break_params  # noqa: E402

CASES = [("dam_break_16_steps3.npz", (16, 16, 16), 20, 3),
         ("dam_break_32x24x16_steps1.npz", (32, 24, 16), 12, 1)]


def run_case(size, iters, steps):
    p, cap = dam_break_params(*size)
    st = OracleState(p, cap, iters)
    st.run_init()
    for _ in range(steps):
        st.run_step()
    out = {f: getattr(st, f) for f in OracleState.FIELDS}
    out.update(size=np.array(size), iterations=np.array(iters), steps=np.array(steps),
               capacity=np.array(cap))
    return out


if __name__ == "__main__":
    for name, size, iters, steps in CASES:
        np.savez_compressed(os.path.join(HERE, name), **run_case(size, iters, steps))
        print("wrote", name, os.path.getsize(os.path.join(HERE, name)), "bytes")
