"""CPU-side checks of the drop-in boundary: the shared library loads, exports every symbol
include/fluid_engine.h declares, and the host-only entry points behave.  No compute calls."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import fluid_amd
from fluid_amd import engine as E
from fluid_amd.params import LAYOUT_OFFSETS, PARAMS_BYTES, FluidParams, default_params

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions(header="fluid_engine.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fluid_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = fluid_amd.load_library()
    declared = header_functions()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/fluid_engine.h but not exported"
    assert sorted(E.EXPORTED_SYMBOLS) == declared
    assert lib.fluid_abi_version() == 1


def test_library_exports_every_symbol_of_the_slab_driver_header():
    """include/fluid_slab.h (the multi-GPU driver) lives in the same library; its host-only entry points
    work without a GPU, and creating an engine-backed driver without one fails loudly."""
    from fluid_amd import slab as S

    lib = S._lib()
    declared = header_functions("fluid_slab.h")
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/fluid_slab.h but not exported"
    assert sorted(S.EXPORTED_SYMBOLS) == declared
    assert S.partition_z(512, 8) == [(64 * r, 64) for r in range(8)]
    assert S.partition_z(10, 4) == [(0, 3), (3, 3), (6, 2), (8, 2)]
    if not os.path.exists("/dev/kfd"):
        with pytest.raises(S.SlabError) as e:
            S.SlabDriver(default_params(16, 16, 16, 0), 0, 2)
        assert e.value.code in (E.ERR_NO_DEVICE, E.ERR_HIP)
    with pytest.raises(S.SlabError, match="cannot give rank"):
        S.SlabDriver(default_params(16, 16, 4, 0), 5, 8)


def test_params_struct_matches_reference_layout_file():
    """Offsets of shaders_fluid/fluids_uniform_buffer_layout.txt:4-56 (copied as numbers into
    params.LAYOUT_OFFSETS) against the ctypes structure; total 264 bytes."""
    assert C.sizeof(FluidParams) == PARAMS_BYTES == 264
    for name, off in LAYOUT_OFFSETS.items():
        assert getattr(FluidParams, name).offset == off, name
    ref_layout = "/root/reference/shaders_fluid/fluids_uniform_buffer_layout.txt"
    if os.path.exists(ref_layout):  # build container only; the GPU box has no reference
        found = dict((m.group(2), int(m.group(1))) for m in re.finditer(
            r"layout\(offset = (\d+)\) \w+ (\w+);", open(ref_layout).read()))
        assert found == LAYOUT_OFFSETS


def test_params_default_c_and_python_agree_and_match_reference_constants():
    lib = fluid_amd.load_library()
    for (w, h, d, cap) in [(20, 20, 20, 1000000), (64, 32, 16, 5), (512, 512, 512, 26738688)]:
        c = FluidParams()
        assert lib.fluid_params_default(C.byref(c), w, h, d, cap) == 0
        assert c.to_bytes() == default_params(w, h, d, cap).to_bytes()
    p = default_params()
    # simulation_constants.h:7,29,48-50,56-64,69,85-89
    assert p.size == (20, 20, 20) and p.fluid_volume == 8000
    assert tuple(p.particle_compute_size) == (1000000, 1)
    assert tuple(p.particle_spawn_cube_resolution) == (100, 100, 100)
    assert p.particle_spawn_cube_volume == 1000000
    assert tuple(p.particle_spawn_cube_offset) == (5.0, 2.0, 1.5)
    assert tuple(p.particle_spawn_cube_size) == (10.0, 10.0, 2.0)
    assert (p.time_delta, p.pressure_air, p.cell_width, p.fluid_density) == (
        np.float32(0.01), 1.0, 1.0, 1.0)
    assert p.gravity == 10.0 and p.diffuse_k == np.float32(0.01)
    assert tuple(p.fountain_position) == (10, 18, 10) and p.fountain_force == -3000.0
    assert p.solid_repel_velocity == np.float32(0.01) and p.active_particle_w == 1.0
    assert (p.cell_type_inactive, p.cell_type_air, p.cell_type_water, p.cell_type_solid) == (0, 1, 2, 3)
    assert lib.fluid_params_default(None, 1, 1, 1, 1) == E.ERR_INVALID_ARG


def test_required_arena_bytes_is_host_arithmetic():
    p = default_params(64, 64, 64, 1000)
    n = fluid_amd.FluidEngine.required_arena_bytes(p, 1000)
    plane = 64 * 64
    # 50 B/cell of attachments with 4 ghost planes per side; 17 B/cell of solver
    # data (neighbour mask, b_i, three working buffers of the pressure loop) with 8 ghost planes per
    # side; particles; each block 4-KiB aligned
    want = plane * 72 * 50 + plane * 80 * 17 + 1000 * 16
    assert want <= n < want + 20 * 4096
    half = fluid_amd.FluidEngine.required_arena_bytes(p, 1000, slab=(0, 32))
    assert half < n  # (a slab context also carries the particle migration list)
    bad = default_params(64, 64, 64, 0)
    bad.cell_type_air = 2  # same value as water
    assert fluid_amd.FluidEngine.required_arena_bytes(bad, 0) == 0


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="GPU present: creation succeeds there")
def test_create_without_gpu_fails_loudly_not_silently():
    """There is no CPU fallback: without a device fluid_create returns an error and says why."""
    with pytest.raises(fluid_amd.FluidEngineError) as ei:
        fluid_amd.FluidEngine(default_params(8, 8, 8, 0))
    assert ei.value.code in (E.ERR_NO_DEVICE, E.ERR_HIP)
    assert "device" in str(ei.value).lower() or "hip" in str(ei.value).lower()


def test_create_rejects_bad_arguments_before_touching_the_gpu():
    lib = fluid_amd.load_library()
    h = C.c_void_p()
    assert lib.fluid_create(C.byref(h), None) == E.ERR_INVALID_ARG
    info = E.CreateInfo()
    info.struct_bytes = 4  # too small
    assert lib.fluid_create(C.byref(h), C.byref(info)) == E.ERR_INVALID_ARG
    p = default_params(8, 8, 8, 0)
    p.cell_type_solid = 300  # does not fit R8_UINT
    blob = (C.c_uint8 * PARAMS_BYTES).from_buffer_copy(p.to_bytes())
    info.struct_bytes = C.sizeof(E.CreateInfo)
    info.params_blob = C.cast(blob, C.c_void_p)
    assert lib.fluid_create(C.byref(h), C.byref(info)) == E.ERR_INVALID_ARG
    assert b"R8_UINT" in lib.fluid_last_error(None)
    p = default_params(8, 8, 8, 0)
    blob = (C.c_uint8 * PARAMS_BYTES).from_buffer_copy(p.to_bytes())
    info.params_blob = C.cast(blob, C.c_void_p)
    info.slab_z_begin, info.slab_z_count = 6, 4  # beyond the grid
    assert lib.fluid_create(C.byref(h), C.byref(info)) == E.ERR_INVALID_ARG
    assert lib.fluid_run_step(None) == E.ERR_INVALID_ARG
    assert lib.fluid_sync(None) == E.ERR_INVALID_ARG


def test_section_table_follows_the_reference_lists():
    """Order of SimulationInitializationSections + SimulationStepSections
    (fluid_flow_sections.h:139-154,163-338) is the order of the section ids."""
    names = E.SECTION_NAMES
    assert names[:3] == ["init_clear_velocities_1", "init_clear_cell_types", "00_init_particles"]
    numbered = [n for n in names if re.match(r"\d\d_", n)]
    assert numbered == sorted(numbered)
    ref = "/root/reference/shaders_fluid"
    if os.path.isdir(ref):
        dirs = sorted(d for d in os.listdir(ref) if re.match(r"(0\d|1[0-8])_", d))
        assert dirs == numbered
    text = open(os.path.join(ROOT, "include", "fluid_engine.h")).read()
    for i, n in enumerate(names):
        m = re.search(r"FLUID_SEC_%s\w* = (\d+)" % n.split("_")[0].upper(), text)
        assert m, n
    assert len(names) == 26


def test_section_names_are_the_reference_list_spellings():
    """fluid_section_name: the strings the reference's section lists use (fluid_flow_sections.h:139-388), the
    ones roctx ranges carry under FLUID_ROCTX=1; needs no GPU."""
    import ctypes as C

    from fluid_amd import engine as E
    lib = E.load_library()
    lib.fluid_section_name.restype = C.c_char_p
    lib.fluid_section_name.argtypes = [C.c_int]
    for i, name in enumerate(E.SECTION_NAMES):
        assert lib.fluid_section_name(i).decode() == name
    assert lib.fluid_section_name(-1) is None and lib.fluid_section_name(len(E.SECTION_NAMES)) is None
