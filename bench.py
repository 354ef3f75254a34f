#!/usr/bin/env python3
"""Benchmark of the fluid-step hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--grid 512] [--iters 200]

Primary metric (BASELINE.json): pressure Jacobi iterations/sec.  One "step" = one execution of the
reference's loop section `12_solve_pressure x iters` (fluid_flow_sections.h:300-313) preceded by
the two pressure clears (:298-299), on the "full-fluid" synthetic grid of SURVEY.md §8d (faces
SOLID, interior WATER, divergence ~ U(-1,1) from SplitMix64 seed 0x5EED0012): K steps are timed
between barriers, value = K*iters / wall time.  Inputs are resident in HBM before the timed region.
With N > 1 the grid is split into Z slabs, one process per GPU, driven by the C++ slab driver
(include/fluid_slab.h): h ghost planes every h sweeps by ncclSend / ncclRecv between Z-neighbours
(torch.distributed only carries the communicator's unique id and the barriers around the timed
region); the total grid is fixed (strong scaling).

Launching.  `python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts its own N
ranks: the parent never touches the GPU, spawns one child per GPU (RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_ADDR = 127.0.0.1 / MASTER_PORT set), relays rank 0's JSON line and exits non-zero with a one-line
reason if a rank fails, the box has fewer GPUs than ranks, or the run exceeds --launch-timeout.  Under
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` (WORLD_SIZE set) the process
IS a rank.  The N > 1 line adds n_ranks_rccl (ncclCommCount), exchange_ms_per_sweep, overlap_mode and —
after the timed region — a checksum of PRESSURES_1 over all ranks against a one-rank run of the same grid.

The same JSON line also carries
  roofline     : the Jacobi kernel's algorithmic bytes (13 B/cell/sweep) / its average launch
                 duration from HIP events on the engine's stream, against 8 TB/s HBM peak;
  cpu_baseline : the single-threaded CPU oracle timed on a bounded slab of the same workload
                 (rank 0, N = 1 only);
  full_step    : full simulation steps/sec (sections 01a…14) on the dam-break scene, N = 1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
JACOBI_BYTES_PER_CELL = 13.0   # Pin 4 + div 4 + type 1 + Pout 4


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--grid", type=int, nargs="+", default=[512],
                    help="W [H D]; default 512^3 (BASELINE config the metric is quoted on)")
    ap.add_argument("--iters", type=int, default=200, help="Jacobi sweeps per step")
    ap.add_argument("--pressure-kernel", type=int, default=0, help="engine option (0 = auto)")
    ap.add_argument("--no-fuse", action="store_true", help="one dispatch per sweep")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-full-step", action="store_true")
    ap.add_argument("--full-step-steps", type=int, default=20)
    ap.add_argument("--no-surface", action="store_true",
                    help="skip the surface-prep figure (128^3 + 640^3 detailed grid, 4.3 GB)")
    ap.add_argument("--no-checksum", action="store_true",
                    help="N > 1: skip the cross-rank PRESSURES_1 checksum against a one-rank run")
    ap.add_argument("--spawn", action="store_true",
                    help="start the ranks as child processes even for --gpus 1 (the launcher's own test)")
    ap.add_argument("--launch-timeout", type=float, default=3000.0,
                    help="seconds the self-launched ranks get before they are killed")
    return ap.parse_args(argv)


FUSED_KERNEL_SOURCES = ["kernels_pressure_fused.h", "kernels_pressure_fused3.h", "pressure_fused_launch.h",
                        "pressure_fused.hip", "pressure_fused3.hip", "pressure_fused_stream.hip", "pressure_common.h"]


def kernel_sources_sha16():
    """What a recorded counter figure is valid for: the sources of the Jacobi loop kernel and of its
    launch shaping (tools/make_pmc_traffic.py stamps the same hash into profiles/*/pmc_traffic_*.json)."""
    import hashlib
    h = hashlib.sha256()
    for name in FUSED_KERNEL_SOURCES:
        with open(os.path.join(ROOT, "vulkan-3d-fluid-simulation_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def all_kernel_sources_sha16():
    """What a whole-step counter record is valid for: every kernel source of the engine."""
    import glob
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "vulkan-3d-fluid-simulation_amd", "csrc")
    for path in sorted(glob.glob(os.path.join(csrc, "*.h")) + glob.glob(os.path.join(csrc, "*.hip"))):
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def recorded_step_traffic(n):
    """HBM-side bytes per step of the full tank from tools/pmc_step_traffic.sh, if recorded for these sources."""
    import glob
    want = all_kernel_sources_sha16()
    why = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", f"pmc_traffic_full_step_dense_{n}.json")),
                       reverse=True):
        try:
            with open(path) as f:
                rec = json.load(f)
        except (OSError, ValueError):
            continue
        if rec.get("all_kernel_sources_sha16") == want:
            return rec.get("traffic_bytes_per_step"), os.path.relpath(path, ROOT)
        why = (f"{os.path.relpath(path, ROOT)} was recorded for kernel sources "
               f"{rec.get('all_kernel_sources_sha16')}, the build has {want}")
    return None, why


def recorded_traffic(kernel, size):
    """HBM-side bytes per launch of `kernel` from the PMC passes kept under profiles/ (rocprofv3 --pmc
    cannot run inside this process): (bytes, source) for this kernel and grid — but only from a record
    stamped with the hash of the kernel sources as they are now; a stale record gives (None, why)."""
    import glob
    want = kernel_sources_sha16()
    stale = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "pmc_traffic_*.json")), reverse=True):
        try:
            with open(path) as f:
                rec = json.load(f)
        except (OSError, ValueError):
            continue
        if rec.get("kernel") == kernel and list(rec.get("grid", [])) == list(size):
            if rec.get("kernel_sources_sha16") == want:
                return rec.get("traffic_bytes_per_launch"), os.path.relpath(path, ROOT)
            stale = (f"{os.path.relpath(path, ROOT)} was recorded for kernel sources "
                     f"{rec.get('kernel_sources_sha16')}, the build has {want}")
    return None, stale


def grid_dims(grid):
    if len(grid) == 1:
        return grid[0], grid[0], grid[0]
    if len(grid) == 3:
        return tuple(grid)
    raise SystemExit("--grid takes 1 or 3 integers")


def cpu_baseline_jacobi(size, iters_hint):
    """Oracle (single thread) on a bounded sample of the Jacobi workload: a slab of the same XY
    extent, `planes` deep, full-fluid inputs; sized for 10-20 s of CPU work."""
    import fluid_amd
    from fluid_amd import scenes
    from oracle_binding import OracleState

    w, h, d = size
    planes = max(4, min(d, (1 << 24) // (w * h)))   # ~16.7 M cells
    sweeps = 64   # ~12 s on one core of the GPU box's host
    p = fluid_amd.default_params(w, h, planes, 0)
    st = OracleState(p, 0, sweeps)
    st.cell_types[...] = scenes.full_fluid_types(st.shape)
    st.divergences[...] = scenes.full_fluid_divergence(st.shape)
    st.pressures_1[...] = p.pressure_air
    st.pressures_2[...] = p.pressure_air
    st.solve_pressure(2)  # warm the caches / page in
    t0 = time.perf_counter()
    st.solve_pressure(sweeps)
    dt = time.perf_counter() - t0
    cells_per_s = w * h * planes * sweeps / dt
    return {
        "value": cells_per_s / (w * h * d),   # sweeps/s extrapolated linearly to the full grid
        "unit": "iterations/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{sweeps} sweeps of a {w}x{h}x{planes} full-fluid slab "
                  f"({dt:.1f} s, {cells_per_s / 1e6:.1f} Mcells/s), scaled by cell count to "
                  f"{w}x{h}x{d}",
        "cells_per_s": cells_per_s,
        "ms_per_sweep": 1e3 * (w * h * d) / cells_per_s,
        "cpu_model": host_cpu_model(),
        "full_step_c1": cpu_vs_gpu_full_step_c1(),
    }


def host_cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_vs_gpu_full_step_c1():
    """BASELINE.json configs[0] (64^3, 8 particles/cell, 40 Jacobi iterations, dam break): whole steps
    of the single-threaded oracle and of the engine on the same scene."""
    import fluid_amd
    from oracle_binding import OracleState

    size, iters, steps = (64, 64, 64), 40, 3
    p, cap = fluid_amd.dam_break_params(*size)
    st = OracleState(p, cap, iters)
    st.run_init()
    st.run_step()
    t0 = time.perf_counter()
    for _ in range(steps):
        st.run_step()
    cpu_dt = (time.perf_counter() - t0) / steps
    with fluid_amd.FluidEngine(p, particle_capacity=cap, pressure_iterations=iters) as eng:
        eng.run_init()
        for _ in range(4):
            eng.run_step()
        eng.sync()
        t0 = time.perf_counter()
        for _ in range(50):
            eng.run_step()
        eng.sync()
        gpu_dt = (time.perf_counter() - t0) / 50
    return {"workload": f"dam-break 64x64x64, {cap} particles, {iters} Jacobi iters",
            "cpu_steps_per_sec": 1.0 / cpu_dt, "cpu_cores": 1, "gpu_steps_per_sec": 1.0 / gpu_dt}


def measured_copy_bandwidth(device_index):
    """Secondary roofline denominator: a 1-GiB device-to-device copy (hipMemcpyAsync),
    read + write bytes per second, best of a few.  Uses the HIP runtime the engine library already
    loaded (a second runtime in the process would not see the GPU)."""
    import ctypes as C

    path = None
    with open("/proc/self/maps") as f:
        for line in f:
            if "libamdhip64" in line:
                path = line.split()[-1]
                break
    hip = C.CDLL(path or "libamdhip64.so")
    vp = C.c_void_p

    def ok(rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed: hipError {rc}")

    ok(hip.hipSetDevice(C.c_int(device_index)), "hipSetDevice")
    nbytes = 1 << 30
    src, dst, e0, e1 = vp(), vp(), vp(), vp()
    ok(hip.hipMalloc(C.byref(src), C.c_size_t(nbytes)), "hipMalloc")
    ok(hip.hipMalloc(C.byref(dst), C.c_size_t(nbytes)), "hipMalloc")
    try:
        ok(hip.hipMemsetAsync(src, C.c_int(1), C.c_size_t(nbytes), vp(0)), "hipMemsetAsync")
        ok(hip.hipEventCreate(C.byref(e0)), "hipEventCreate")
        ok(hip.hipEventCreate(C.byref(e1)), "hipEventCreate")
        best = 0.0
        for _ in range(6):
            ok(hip.hipEventRecord(e0, vp(0)), "hipEventRecord")
            ok(hip.hipMemcpyAsync(dst, src, C.c_size_t(nbytes), C.c_int(3), vp(0)), "hipMemcpyAsync")
            ok(hip.hipEventRecord(e1, vp(0)), "hipEventRecord")
            ok(hip.hipEventSynchronize(e1), "hipEventSynchronize")
            ms = C.c_float(0)
            ok(hip.hipEventElapsedTime(C.byref(ms), e0, e1), "hipEventElapsedTime")
            best = max(best, 2.0 * nbytes / (ms.value * 1e-3) / 1e9)
        hip.hipEventDestroy(e0)
        hip.hipEventDestroy(e1)
    finally:
        hip.hipFree(src)
        hip.hipFree(dst)
    return best


def full_step_bench(size, iters, steps, device):
    """Full simulation steps/sec (01a…14) on the dam-break scene, single GPU."""
    import fluid_amd
    from fluid_amd import engine as E

    p, cap = fluid_amd.dam_break_params(*size)
    with fluid_amd.FluidEngine(p, particle_capacity=cap, pressure_iterations=iters,
                               device=device) as eng:
        eng.run_init()
        for _ in range(10):  # ten warm-up steps (the simulation's steady state: bricks far
            eng.run_step()   # from the water have gone quiet)
        eng.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.run_step()
        eng.sync()
        dt = time.perf_counter() - t0
        quiet = eng.get_stat(E.STAT_QUIET_BRICKS) / max(eng.get_stat(E.STAT_BRICKS), 1)
        # per-section HIP-event times from a few more steps (not part of the rate above)
        eng.enable_timing(True)
        eng.reset_timing()
        tsteps = min(steps, 3)
        for _ in range(tsteps):
            eng.run_step()
        eng.sync()
        sections = {k: round(v[0] / tsteps, 4) for k, v in eng.section_times().items() if v[1]}
        particles = {"stored_sorted_by_bin": bool(eng.get_stat(E.STAT_PARTICLE_BINNED)),
                     "sorts": eng.get_stat(E.STAT_PARTICLE_SORTS)}
    return {
        "particle_storage": particles,
        "workload": f"dam-break {size[0]}x{size[1]}x{size[2]}, {cap} particles, {iters} Jacobi iters "
                    f"(sparse: {round(100 * quiet)} % of the bricks are skipped, so this is a rate, not a "
                    "roofline statement: see full_step_dense)",
        "steps_per_sec": steps / dt,
        "ms_per_step": 1e3 * dt / steps,
        "steps": steps,
        "quiet_brick_fraction": round(quiet, 4),
        "section_ms_per_step": sections,
        "note": ("grouped passes: 04/05 = the two type scans of 04+05, 07 = 07+08, 09 = 09+10+11; "
                 "12_solve_pressure includes its prepare / import / export passes"),
    }


def full_step_dense_bench(grid, iters, steps, device):
    """Full simulation steps/sec (01a...14) on a tank filled to the brim: every section works on nearly
    every cell and nothing is skipped, so the algorithmic bytes of SURVEY.md 8d (reference layout:
    293 + 13 x iterations B/cell, 48 B/particle) over the step time IS a roofline statement."""
    import fluid_amd
    from fluid_amd import engine as E

    w, h, d = grid
    size = (w - 4.0, h - 4.0, d - 4.0)
    res = tuple(int(round(2.0 * v)) for v in size)
    cap = res[0] * res[1] * res[2]
    p = fluid_amd.default_params(w, h, d, cap)
    p.particle_spawn_cube_resolution[:] = res
    p.particle_spawn_cube_volume = cap
    p.particle_spawn_cube_offset[:] = (2.0, 2.0, 2.0)
    p.particle_spawn_cube_size[:] = size
    with fluid_amd.FluidEngine(p, particle_capacity=cap, pressure_iterations=iters, device=device) as eng:
        eng.run_init()
        for _ in range(2):
            eng.run_step()
        eng.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.run_step()
        eng.sync()
        dt = (time.perf_counter() - t0) / steps
        quiet = eng.get_stat(E.STAT_QUIET_BRICKS)
        eng.enable_timing(True)
        eng.reset_timing()
        for _ in range(2):
            eng.run_step()
        eng.sync()
        sections = {k: round(v[0] / 2, 4) for k, v in eng.section_times().items() if v[1]}
    cells = w * h * d
    step_bytes = (293.0 + 13.0 * iters) * cells + 48.0 * cap   # SURVEY.md 8d, reference layout
    shape = f"{w}^3" if w == h == d else f"{w}x{h}x{d}"
    traffic, traffic_source = (recorded_step_traffic(w) if w == h == d and iters == 200 else (None, None))
    return {"workload": f"full tank {shape}, {cap} particles (8 per cell), {iters} Jacobi iters",
            "steps_per_sec": 1.0 / dt, "ms_per_step": 1e3 * dt, "steps": steps,
            "algorithmic_bytes_per_step": step_bytes,
            "algorithmic_GBps": step_bytes / dt / 1e9,
            "frac_of_hbm_peak": step_bytes / dt / 1e9 / HBM_PEAK_GBS,
            # the physical figure: counter bytes per step (every kernel of the step) over the step time
            "traffic": traffic, "traffic_source": traffic_source,
            "frac_traffic": (traffic / dt / 1e9 / HBM_PEAK_GBS) if traffic else None,
            "quiet_bricks": quiet,
            "section_ms_per_step": sections,
            "particles_per_sec": cap / dt}


def surface_prep_bench(n, iters, device):
    """The surface-prep tail of the reference's step list (sections 14a, 15-18 on the detailed grid,
    SURVEY.md 8f N3) on the dam-break scene: n^3 simulation cells, (5n)^3 detailed cells."""
    import fluid_amd

    p, cap = fluid_amd.dam_break_params(n, n, n)
    with fluid_amd.FluidEngine(p, particle_capacity=cap, pressure_iterations=iters, device=device,
                               surface_prep=True) as eng:
        eng.run_init()
        for _ in range(4):
            eng.run_step()
        eng.sync()
        steps = 5
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.run_step()
        eng.sync()
        dt = (time.perf_counter() - t0) / steps
        eng.enable_timing(True)
        eng.reset_timing()
        for _ in range(3):
            eng.run_step()
        eng.sync()
        times = {k: v[0] / 3 for k, v in eng.section_times().items() if v[1]}
        d = eng.detailed_shape
    cells = d[0] * d[1] * d[2]
    tail = sum(v for k, v in times.items() if k[:3] in ("14a", "15_", "16_", "17_", "18_"))
    # clear 4 + 16: R 4+4 W 4 + 17: R 4 W 4 + 18: (R 4 + W 4) x 4 dispatches = 56 B per detailed cell
    return {"workload": f"dam-break {n}^3 + detailed grid {d[2]}x{d[1]}x{d[0]}, {iters} Jacobi iters",
            "steps_per_sec": 1.0 / dt, "ms_per_step": 1e3 * dt, "surface_tail_ms": tail,
            "surface_tail_algorithmic_GBps": 56.0 * cells / (tail * 1e-3) / 1e9,
            "section_ms": {k: round(v, 4) for k, v in times.items()
                           if k[:3] in ("14a", "15_", "16_", "17_", "18_")}}


def slab_full_step_bench(size, iters, steps, dist_ctx, overlap=None):
    """Full simulation steps/sec on Z slabs (the C++ driver of include/fluid_slab.h), dam-break scene."""
    import torch.distributed as dist

    import fluid_amd
    from fluid_amd import slab as S

    p, cap = fluid_amd.dam_break_params(*size)
    sim = S.SlabDriver(p, dist_ctx.rank, dist_ctx.world, particle_capacity=cap, pressure_iterations=iters,
                       device=dist_ctx.device, overlap=overlap)
    if dist_ctx.world > 1:
        sim.attach_rccl()
    sim.run_init()
    for _ in range(2):
        sim.run_step()
    steps = min(steps, 5)
    sim.engine.sync()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        sim.run_step()
    sim.engine.sync()
    dist.barrier()
    dt = time.perf_counter() - t0
    out = {"workload": f"dam-break {size[0]}x{size[1]}x{size[2]}, {cap} particles, {iters} Jacobi "
                       f"iters, Z slabs over {dist_ctx.world} GPUs",
           "steps_per_sec": steps / dt, "ms_per_step": 1e3 * dt / steps, "steps": steps,
           "particles_migrated_by_rank0": sim.stat(S.STAT_MIGRATED),
           "sampler_halo_planes": sim.stat(S.STAT_SAMPLER_HALO),
           "advect_passes_redone": sim.stat(S.STAT_SAMPLER_RERUNS)}
    sim.close()
    return out


# ---- self-launch: one child process per GPU ---------------------------------------------------------------
def visible_gpu_count():
    """hipGetDeviceCount, asked in a throwaway child process: the launcher itself must never initialise
    the GPU (its children do, and a process that has may not be replaced or forked from)."""
    import subprocess
    code = ("import ctypes\n"
            "n = ctypes.c_int(0)\n"
            "for name in ('libamdhip64.so', 'libamdhip64.so.7', '/opt/rocm/lib/libamdhip64.so'):\n"
            "    try:\n"
            "        h = ctypes.CDLL(name)\n"
            "    except OSError:\n"
            "        continue\n"
            "    print(n.value if h.hipGetDeviceCount(ctypes.byref(n)) == 0 else 0)\n"
            "    break\n"
            "else:\n"
            "    print(0)\n")
    try:
        res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
        return int(res.stdout.strip().splitlines()[-1])
    except Exception:
        return 0


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher around it: spawn the N ranks, relay rank 0's JSON line.
    Returns the exit code.  No HIP call, no torch import in this process."""
    import signal
    import subprocess
    import threading

    n = args.gpus
    child = os.environ.get("FLUID_BENCH_CHILD")  # tests: another rank program (tests/bench_child_standin.py)
    if child is None:
        have = visible_gpu_count()
        if have < n:
            print(f"bench.py --gpus {n}: {have} GPU(s) visible to this process (hipGetDeviceCount); one rank "
                  f"per GPU needs {n}", file=sys.stderr, flush=True)
            return 2
    port = os.environ.get("MASTER_PORT") or str(free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen(
            [sys.executable, child or os.path.abspath(__file__)] + list(argv), env=env,
            stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr, text=True,
            start_new_session=True))   # a process group of its own: killable without a pattern
    last_json = [None]

    def relay():
        for line in procs[0].stdout:
            if line.startswith("{"):
                last_json[0] = line.rstrip("\n")
            else:
                sys.stderr.write(line)

    reader = threading.Thread(target=relay, daemon=True)
    reader.start()

    def kill_all():
        for q in procs:
            if q.poll() is None:
                try:
                    os.killpg(q.pid, signal.SIGTERM)
                except OSError:
                    pass
        t_end = time.time() + 10
        for q in procs:
            try:
                q.wait(timeout=max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(q.pid, signal.SIGKILL)
                except OSError:
                    pass
                q.wait()

    deadline = time.time() + args.launch_timeout
    why = None
    while True:
        codes = [q.poll() for q in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            # the others are (or will be) stuck in a collective with a rank that is gone
            grace = time.time() + 15
            while time.time() < grace and any(q.poll() is None for q in procs):
                time.sleep(0.2)
            why = "rank(s) failed: " + ", ".join(f"rank {r} exit code {c}" for r, c in bad)
            break
        if all(c == 0 for c in codes):
            break
        if time.time() > deadline:
            why = f"the {n} ranks did not finish within --launch-timeout {args.launch_timeout:.0f} s"
            break
        time.sleep(0.2)
    if why is not None:
        kill_all()
        print(f"bench.py --gpus {n}: {why}", file=sys.stderr, flush=True)
        return 1
    reader.join(timeout=10)
    if last_json[0] is None:
        print(f"bench.py --gpus {n}: the ranks exited cleanly but rank 0 printed no JSON line",
              file=sys.stderr, flush=True)
        return 1
    print(last_json[0], flush=True)
    return 0


# ---- a rank of an N > 1 run -----------------------------------------------------------------------------
OVERLAP_NAMES = {0: "inline", 1: "split_pass_before_exchange", 2: "split_passes_before_and_after"}


def image_checksum(a):
    """Order-independent exact checksum of an image's bits: (sum of the 32-bit words mod 2^64, their XOR,
    the number of words).  Slabs add / XOR to the whole grid's."""
    words = np.ascontiguousarray(a).view(np.uint32).ravel()
    with np.errstate(over="ignore"):
        total = int(words.sum(dtype=np.uint64))
    return total, int(np.bitwise_xor.reduce(words)) if words.size else 0, int(words.size)


class EngineRanks:
    """What a rank computes with: the HIP engine behind the C++ slab driver, RCCL attached (the product).
    tests/bench_child_standin.py substitutes a host stand-in to run this file's rank code without a GPU."""

    def make_solver(self, size, iters, dist_ctx, args):
        from fluid_amd import engine as E
        from fluid_amd import slab as S
        solver = S.SlabDriver.create_full_fluid(size, iters, dist_ctx)
        solver.engine.set_option(E.OPT_PRESSURE_KERNEL, args.pressure_kernel)
        return solver

    def one_rank_pressures_1(self, size, iters, device, args):
        """PRESSURES_1 after the same step (12a, 12b, the loop) on the whole grid in one context."""
        import fluid_amd
        from fluid_amd import engine as E
        from fluid_amd import scenes
        w, h, d = size
        with fluid_amd.FluidEngine(fluid_amd.default_params(w, h, d, 0), particle_capacity=0,
                                   pressure_iterations=iters, device=device) as eng:
            eng.set_option(E.OPT_PRESSURE_KERNEL, args.pressure_kernel)
            eng.upload_image(E.CELL_TYPES, scenes.full_fluid_types((d, h, w)))
            eng.upload_image(E.DIVERGENCES, scenes.full_fluid_divergence((d, h, w), scenes.SEED_JACOBI))
            eng.run_section("12a_clear_pressures_1")
            eng.run_section("12b_clear_pressures_2")
            eng.solve_pressure(iters)
            return eng.download_image(E.PRESSURES_1)

    full_step = True


def cross_rank_checksum(solver, size, iters, dist_ctx, args, ranks):
    """After the timed region: every rank sums the bits of its planes of PRESSURES_1 as the last timed step
    left them; rank 0 compares the total with a one-rank run of the whole grid."""
    import torch.distributed as dist
    from fluid_amd import engine as E

    mine = image_checksum(solver.download_image(E.PRESSURES_1))
    parts = [None] * dist_ctx.world
    dist.all_gather_object(parts, mine)
    if dist_ctx.rank != 0:
        return None
    total = sum(p[0] for p in parts) % (1 << 64)
    xor = 0
    for p in parts:
        xor ^= p[1]
    out = {"image": "PRESSURES_1", "words": sum(p[2] for p in parts),
           "sum_mod_2_64": total, "xor": xor}
    try:
        ref = image_checksum(ranks.one_rank_pressures_1(size, iters, dist_ctx.device, args))
        out["one_rank"] = {"sum_mod_2_64": ref[0], "xor": ref[1], "words": ref[2]}
        out["matches_one_rank_run"] = (ref[0], ref[1], ref[2]) == (total, xor, out["words"])
    except Exception as exc:
        out["one_rank"] = {"error": f"{type(exc).__name__}: {exc}"}
        out["matches_one_rank_run"] = None
    return out


def slab_rank_main(args, ranks=None):
    """One rank of `--gpus N` (N > 1, or FLUID_BENCH_FORCE_SLAB=1): the C++ slab driver's pressure step."""
    import fluid_amd  # noqa: F401
    from fluid_amd import slab as S

    ranks = ranks or EngineRanks()
    size = grid_dims(args.grid)
    w, h, d = size
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist_ctx = S.init_distributed(local_rank)
    solver = ranks.make_solver(size, args.iters, dist_ctx, args)
    result = solver.benchmark(args.steps, args.warmup)
    halo = solver.stat(S.STAT_EFFECTIVE_HALO)
    rccl_ranks = solver.stat(S.STAT_RCCL_RANKS)
    checksum = None
    if not args.no_checksum:
        try:
            checksum = cross_rank_checksum(solver, size, args.iters, dist_ctx, args, ranks)
        except Exception as exc:
            checksum = {"error": f"{type(exc).__name__}: {exc}"}
    solver.close()
    full = None
    if not args.no_full_step and ranks.full_step:
        try:
            full = slab_full_step_bench(size, args.iters, args.full_step_steps, dist_ctx,
                                        overlap=(result.get("halo_overlap") or {}).get("used"))
        except Exception as exc:  # the headline metric above must survive a failure here
            full = {"error": f"{type(exc).__name__}: {exc}"}
    if rank == 0:
        cells = w * h * d
        sweeps = args.steps * args.iters
        wall = result["wall_s"]
        kernel_ms = result["kernel_ms_per_sweep"]
        local_cells = result["local_cells"]
        achieved = (JACOBI_BYTES_PER_CELL * local_cells / (kernel_ms * 1e-3) / 1e9) if kernel_ms > 0 else None
        used = (result.get("halo_overlap") or {}).get("used")
        out = {
            "metric": "pressure_jacobi_iterations_per_sec",
            "value": sweeps / wall,
            "unit": "iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * wall / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"12_solve_pressure loop, {w}x{h}x{d} full-fluid grid, "
                                   f"{args.iters} Jacobi iterations per step, Z slabs over "
                                   f"{world} GPUs, halo exchange over RCCL Send/Recv",
                       "grid": [w, h, d], "jacobi_iterations": args.iters,
                       "parallelism": f"zslab{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
                         "traffic": None,
                         "kernel": ("k12_canon_t (per GPU, on its slab; launches of 3 + 3 + 2 sweeps between two "
                                    "exchanges of 8 planes)" if w % 4 == 0 and w <= 512 else
                                    "k12_canon2 (per GPU, on its slab; two sweeps per launch)"
                                    if w % 4 == 0 and w <= 1024 else "k12_zmarch / k12_plain"),
                         "kernel_ms_per_sweep": kernel_ms,
                         "note": "achieved = 13 B x local cells / kernel time per sweep (HIP events "
                                 "on the engine's stream, MAX over ranks); ghost-plane recompute "
                                 "of the deep halos is inside that time"},
            "n_ranks_rccl": rccl_ranks,
            "exchange_ms_per_sweep": result.get("exchange_ms_per_sweep"),
            "halo_exchange_ms_per_sweep": result.get("exchange_ms_per_sweep"),
            "halo_depth": halo,
            "overlap_mode": OVERLAP_NAMES.get(used, used),
            "halo_overlap": result.get("halo_overlap"),
            "exchanges_per_step": result["exchanges_per_step"],
            "checksum": checksum,
            "cells_per_sec": cells * sweeps / wall,
        }
        if full is not None:
            out["full_step"] = full
        print(json.dumps(out), flush=True)
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse_args(argv)
    size = grid_dims(args.grid)
    w, h, d = size
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.spawn):
        if args.gpus < 1:
            raise SystemExit("--gpus must be >= 1")
        sys.exit(launch_ranks(args, argv))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} in the environment")

    import fluid_amd
    from fluid_amd import engine as E
    from fluid_amd import scenes

    if not os.path.exists(E.LIB_PATH):
        fluid_amd.build_engine()

    if world > 1 or os.environ.get("FLUID_BENCH_FORCE_SLAB") == "1":  # (the env var: tests only)
        slab_rank_main(args)
        return

    # ------------------------------------------------------------------ single GPU
    p = fluid_amd.default_params(w, h, d, 0)
    eng = fluid_amd.FluidEngine(p, particle_capacity=0, pressure_iterations=args.iters,
                                device=local_rank)
    eng.set_option(E.OPT_PRESSURE_KERNEL, args.pressure_kernel)
    if args.no_fuse:
        eng.set_option(E.OPT_JACOBI_FUSE, 1)
    sweeps_per_launch = eng.pressure_loop_max_sweeps()   # 3 up to 512 cells wide, 2 above, 1 with --no-fuse
    shape = (d, h, w)
    eng.upload_image(E.CELL_TYPES, scenes.full_fluid_types(shape))
    eng.upload_image(E.DIVERGENCES, scenes.full_fluid_divergence(shape, scenes.SEED_JACOBI + rank))

    def step():
        eng.run_section("12a_clear_pressures_1")
        eng.run_section("12b_clear_pressures_2")
        eng.solve_pressure(args.iters)

    for _ in range(args.warmup):
        step()
    eng.sync()
    eng.enable_timing(True)
    eng.reset_timing()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    eng.sync()
    wall = time.perf_counter() - t0
    loop_ms, dispatches = eng.section_time_ms("12_solve_pressure")
    clear_ms = eng.section_time_ms("12a_clear_pressures_1")[0] + \
        eng.section_time_ms("12b_clear_pressures_2")[0]
    eng.enable_timing(False)
    # parity spot check of what was just timed would need the full-size oracle; tests cover it.
    # Convergence read-out of the iterate the loop left (after the timed region)
    res_max, res_sumsq, res_cells = eng.pressure_residual(E.PRESSURES_1 if args.iters % 2 == 0
                                                          else E.PRESSURES_2)
    eng.close()

    cells = w * h * d
    sweeps = args.steps * args.iters
    assert dispatches == sweeps, (dispatches, sweeps)  # loop sections count their sweeps
    # The loop section runs three (grids up to 512 cells wide) or two sweeps per kernel launch (temporal
    # blocking) unless --no-fuse: one launch then carries 3 (2) x 13 B/cell of algorithmic traffic.  A loop of
    # 200 is 66 launches of three and one of two; HIP events bracket the whole loop section (import / export
    # passes included) and the time is divided evenly over the sweeps, so the per-launch figure of the
    # dominant kernel is conservative.  The loop's mask / b_i pass (k12_prepare) runs only when CELL_TYPES /
    # DIVERGENCES change: once per step in a simulation (0.45 ms at 512^3, 1 % of a 200-iteration loop; inside
    # full_step and full_step_dense), once in all here.
    fused = sweeps_per_launch >= 2
    launch_ms = loop_ms / sweeps * sweeps_per_launch
    kernel_ms = loop_ms / sweeps  # per sweep
    achieved = JACOBI_BYTES_PER_CELL * cells * sweeps_per_launch / (launch_ms * 1e-3) / 1e9
    kernel_name = ({3: "k12_canon_t", 2: "k12_canon2"}.get(sweeps_per_launch, "k12_canon")
                   if w % 4 == 0 else "k12_plain")
    traffic, traffic_source = recorded_traffic(kernel_name, size)
    out = {
        "metric": "pressure_jacobi_iterations_per_sec",
        "value": sweeps / wall,
        "unit": "iterations/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * wall / args.steps,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"12_solve_pressure loop, {w}x{h}x{d} full-fluid grid, "
                               f"{args.iters} Jacobi iterations per step (+ the two pressure clears)",
                   "grid": [w, h, d], "jacobi_iterations": args.iters, "parallelism": "single"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     # the physical figure: counter bytes per launch over the launch time
                     "achieved_traffic": (traffic / (launch_ms * 1e-3) / 1e9) if traffic else None,
                     "frac_traffic": (traffic / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                     "traffic_over_single_pass_min": (traffic / (JACOBI_BYTES_PER_CELL * cells))
                     if traffic else None,
                     "traffic_source": traffic_source,
                     "kernel_sources_sha16": kernel_sources_sha16(),
                     "kernel": kernel_name,
                     "sweeps_per_launch": sweeps_per_launch,
                     "launch_ms": launch_ms, "ms_per_sweep": kernel_ms,
                     "algorithmic_bytes_per_launch":
                         JACOBI_BYTES_PER_CELL * cells * sweeps_per_launch,
                     "note": ("frac = ALGORITHMIC bytes (13 B/cell/sweep x the sweeps of a launch) / launch "
                              "time / peak: each launch applies sweeps_per_launch Jacobi sweeps while streaming "
                              "the grid once (temporal blocking), so it can exceed 1 (SURVEY.md 8d allows this).  "
                              "frac_traffic = bytes the L2s requested from the fabric (TCC_EA0_RDREQ x 128 B "
                              "+ WRITE_SIZE, separate rocprofv3 --pmc passes, profiles/) / launch time / "
                              "peak; traffic_over_single_pass_min = those bytes / one 13 B/cell pass") if fused
                     else ""},
        "residual_after_loop": {"max_abs": res_max,
                                "rms": (res_sumsq / max(res_cells, 1)) ** 0.5,
                                "water_cells": res_cells,
                                "note": "r = b - sum(neighbours) + aii*p of the last iterate "
                                        "(fluid_pressure_residual; not in the reference)"},
        "clears_ms_per_step": clear_ms / args.steps,
        "cells_per_sec": cells * sweeps / wall,
    }
    try:
        out["roofline"]["measured_copy_GBps"] = measured_copy_bandwidth(local_rank)
        out["roofline"]["frac_of_measured_copy"] = achieved / out["roofline"]["measured_copy_GBps"]
    except Exception as exc:  # secondary figure only
        out["roofline"]["measured_copy_GBps"] = None
        out["roofline"]["measured_copy_error"] = f"{type(exc).__name__}: {exc}"
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_jacobi(size, args.iters)
        out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
    if not args.no_full_step:
        out["full_step"] = full_step_bench(size, args.iters, args.full_step_steps, local_rank)
        try:
            out["full_step_dense"] = full_step_dense_bench(size, args.iters, 3, local_rank)
        except Exception as exc:  # secondary figure (needs 16 B x 8 particles per cell of HBM)
            out["full_step_dense"] = {"error": f"{type(exc).__name__}: {exc}"}
    if not args.no_surface and not args.no_full_step:
        try:
            out["surface_prep"] = surface_prep_bench(128, 80, local_rank)
        except Exception as exc:  # secondary figure
            out["surface_prep"] = {"error": f"{type(exc).__name__}: {exc}"}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
