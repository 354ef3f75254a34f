"""A rank of `bench.py --gpus N` without a GPU: bench.py's own rank code (slab_rank_main: process group,
the C++ slab driver's pressure step, timing, checksum, JSON line) with the CPU oracle behind the driver's
compute callbacks and gloo as the transport.  The launcher test starts this file in place of bench.py
through FLUID_BENCH_CHILD.  Test infrastructure, not a product path."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class StandinRanks:
    full_step = False

    def make_solver(self, size, iters, dist_ctx, args):
        import fluid_amd
        from fluid_amd import engine as E
        from fluid_amd import scenes
        from fluid_amd import slab as S
        from host_standin import HostSlabCompute

        w, h, d = size
        params = fluid_amd.default_params(w, h, d, 0)
        slab = S.partition_z(d, dist_ctx.world)[dist_ctx.rank]
        comp = HostSlabCompute(params, slab, max_sweeps=2)
        drv = S.SlabDriver(params, dist_ctx.rank, dist_ctx.world, pressure_iterations=iters, compute=comp)
        drv.attach_torch_transport()
        z0, n = slab
        comp.upload(E.CELL_TYPES, scenes.full_fluid_types((n, h, w), z0, d))
        comp.upload(E.DIVERGENCES, scenes.full_fluid_divergence((n, h, w), scenes.SEED_JACOBI, z0))
        drv.exchange_image(E.CELL_TYPES, 1)
        return drv

    def one_rank_pressures_1(self, size, iters, device, args):
        import fluid_amd
        from fluid_amd import scenes
        from oracle_binding import OracleState

        w, h, d = size
        p = fluid_amd.default_params(w, h, d, 0)
        st = OracleState(p, 0, iters)
        st.cell_types[...] = scenes.full_fluid_types(st.shape)
        st.divergences[...] = scenes.full_fluid_divergence(st.shape, scenes.SEED_JACOBI)
        st.pressures_1[...] = p.pressure_air
        st.pressures_2[...] = p.pressure_air
        st.solve_pressure(iters)
        return st.pressures_1


if __name__ == "__main__":
    if os.environ.get("FLUID_BENCH_CHILD_FAIL_RANK") == os.environ.get("RANK"):
        sys.exit(7)   # the launcher test's failing rank
    b = _bench()
    b.slab_rank_main(b.parse_args(sys.argv[1:]), StandinRanks())
