"""Simulation parameters: the reference's 264-byte std140 uniform block as a ctypes structure.

Mirrors ``SimulationParametersBufferData`` (/root/reference/simulation_constants.h:153-174) and the
byte offsets of /root/reference/shaders_fluid/fluids_uniform_buffer_layout.txt:4-56.  The C twin is
``fluid_params`` in include/fluid_engine.h; ``default_params`` is the Python twin of
``fluid_params_default`` (tests check the two produce identical bytes).
"""
import ctypes as C

PARAMS_BYTES = 264

# simulation_constants.h:144-146
CELL_INACTIVE, CELL_AIR, CELL_WATER, CELL_SOLID = 0, 1, 2, 3


class FluidParams(C.Structure):
    _fields_ = [
        ("fluid_size", C.c_uint32 * 3),                      # 0
        ("fluid_volume", C.c_uint32),                        # 12
        ("cell_type_inactive", C.c_uint32),                  # 16
        ("cell_type_air", C.c_uint32),                       # 20
        ("cell_type_water", C.c_uint32),                     # 24
        ("cell_type_solid", C.c_uint32),                     # 28
        ("time_delta", C.c_float),                           # 32
        ("pressure_air", C.c_float),                         # 36
        ("cell_width", C.c_float),                           # 40
        ("fluid_density", C.c_float),                        # 44
        ("particle_compute_size", C.c_uint32 * 2),           # 48
        ("_pad56", C.c_uint32 * 2),                          # 56
        ("particle_spawn_cube_resolution", C.c_uint32 * 3),  # 64
        ("particle_spawn_cube_volume", C.c_uint32),          # 76
        ("particle_spawn_cube_offset", C.c_float * 3),       # 80
        ("_pad92", C.c_uint32),                              # 92
        ("particle_spawn_cube_size", C.c_float * 3),         # 96
        ("gravity", C.c_float),                              # 108
        ("diffuse_k", C.c_float),                            # 112
        ("detailed_resolution", C.c_int32),                  # 116
        ("detailed_resolution_volume", C.c_int32),           # 120
        ("max_inertia", C.c_int32),                          # 124
        ("inertia_increase_filled", C.c_int32),              # 128
        ("required_neighbour_hits", C.c_int32),              # 132
        ("inertia_increase_neighbour", C.c_int32),           # 136
        ("inertia_decrease", C.c_int32),                     # 140
        ("dens_division_coefficient", C.c_float),            # 144
        ("dens_diffuse_k", C.c_float),                       # 148
        ("_pad152", C.c_uint32 * 2),                         # 152
        ("particle_color", C.c_float * 3),                   # 160
        ("particle_base_size", C.c_float),                   # 172
        ("light_dir", C.c_float * 3),                        # 176
        ("_pad188", C.c_uint32),                             # 188
        ("ambient_color", C.c_float * 3),                    # 192
        ("_pad204", C.c_uint32),                             # 204
        ("diffuse_color", C.c_float * 3),                    # 208
        ("_pad220", C.c_uint32),                             # 220
        ("fluid_surface_render_size", C.c_uint32 * 3),       # 224
        ("active_particle_w", C.c_float),                    # 236
        ("fountain_position", C.c_uint32 * 3),               # 240
        ("fountain_force", C.c_float),                       # 252
        ("solid_repel_velocity", C.c_float),                 # 256
        ("particle_max_size", C.c_float),                    # 260
    ]

    def to_bytes(self) -> bytes:
        return bytes(memoryview(self))

    @classmethod
    def from_bytes(cls, blob: bytes) -> "FluidParams":
        if len(blob) != PARAMS_BYTES:
            raise ValueError(f"params blob must be {PARAMS_BYTES} bytes, got {len(blob)}")
        return cls.from_buffer_copy(blob)

    def copy(self) -> "FluidParams":
        return FluidParams.from_bytes(self.to_bytes())

    @property
    def size(self):
        return tuple(int(v) for v in self.fluid_size)

    @property
    def cells(self) -> int:
        w, h, d = self.size
        return w * h * d


assert C.sizeof(FluidParams) == PARAMS_BYTES

# offsets of fluids_uniform_buffer_layout.txt:4-56
LAYOUT_OFFSETS = {
    "fluid_size": 0, "fluid_volume": 12, "cell_type_inactive": 16, "cell_type_air": 20,
    "cell_type_water": 24, "cell_type_solid": 28, "time_delta": 32, "pressure_air": 36,
    "cell_width": 40, "fluid_density": 44, "particle_compute_size": 48,
    "particle_spawn_cube_resolution": 64, "particle_spawn_cube_volume": 76,
    "particle_spawn_cube_offset": 80, "particle_spawn_cube_size": 96, "gravity": 108,
    "diffuse_k": 112, "detailed_resolution": 116, "detailed_resolution_volume": 120,
    "max_inertia": 124, "inertia_increase_filled": 128, "required_neighbour_hits": 132,
    "inertia_increase_neighbour": 136, "inertia_decrease": 140, "dens_division_coefficient": 144,
    "dens_diffuse_k": 148, "particle_color": 160, "particle_base_size": 172, "light_dir": 176,
    "ambient_color": 192, "diffuse_color": 208, "fluid_surface_render_size": 224,
    "active_particle_w": 236, "fountain_position": 240, "fountain_force": 252,
    "solid_repel_velocity": 256, "particle_max_size": 260,
}


def default_params(width: int = 20, height: int = 20, depth: int = 20,
                   particle_capacity: int = 1000000) -> FluidParams:
    """Reference defaults (simulation_constants.h:7-139) for a grid of the given size."""
    p = FluidParams()
    p.fluid_size[:] = (width, height, depth)
    p.fluid_volume = (width * height * depth) & 0xFFFFFFFF
    p.cell_type_inactive, p.cell_type_air = CELL_INACTIVE, CELL_AIR
    p.cell_type_water, p.cell_type_solid = CELL_WATER, CELL_SOLID
    p.time_delta = 0.01          # :56
    p.pressure_air = 1.0         # :59
    p.cell_width = 1.0           # :61
    p.fluid_density = 1.0        # :62
    p.particle_compute_size[:] = (particle_capacity & 0xFFFFFFFF, 1)  # :33, :162
    p.particle_spawn_cube_resolution[:] = (100, 100, 100)             # :48
    p.particle_spawn_cube_volume = 100 * 100 * 100
    p.particle_spawn_cube_offset[:] = (5.0, 2.0, 1.5)                 # :49
    p.particle_spawn_cube_size[:] = (10.0, 10.0, 2.0)                 # :50
    p.gravity = 10.0             # :64
    p.diffuse_k = 0.01           # :69
    p.detailed_resolution = 5    # :36
    p.detailed_resolution_volume = (125 * width * height * depth) & 0x7FFFFFFF
    p.max_inertia = 100
    p.inertia_increase_filled = 4
    p.required_neighbour_hits = 1
    p.inertia_increase_neighbour = 1
    p.inertia_decrease = 1
    p.dens_division_coefficient = 30.0
    p.dens_diffuse_k = 0.1
    p.particle_color[:] = (1.0, 0.0, 0.0)
    p.particle_base_size = 10.0
    p.light_dir[:] = (1.0, -3.0, 1.0)
    p.ambient_color[:] = (0.0, 0.0, 0.3)
    p.diffuse_color[:] = (0.0, 0.8, 0.7)
    p.fluid_surface_render_size[:] = (5 * width - 1, 5 * height - 1, 5 * depth - 1)
    p.active_particle_w = 1.0    # :53
    p.fountain_position[:] = (width // 2, height - 2, depth // 2)     # :85
    p.fountain_force = -3000.0   # :87
    p.solid_repel_velocity = 0.01  # :89
    p.particle_max_size = 20.0
    return p


def dam_break_params(width: int, height: int, depth: int):
    """The benchmark scene of SURVEY.md §8(d): the reference spawn cube (offset (5,2,1.5), size
    (10,10,2) in a 20^3 grid, simulation_constants.h:48-50) scaled to the grid, sampled at
    round(2*size) points per axis = 8 particles per cell.  Returns (params, particle_capacity)."""
    size = (0.5 * width, 0.5 * height, 0.1 * depth)
    offset = (0.25 * width, 0.10 * height, 0.075 * depth)
    res = tuple(max(1, int(round(2.0 * s))) for s in size)
    volume = res[0] * res[1] * res[2]
    p = default_params(width, height, depth, volume)
    p.particle_spawn_cube_resolution[:] = res
    p.particle_spawn_cube_volume = volume
    p.particle_spawn_cube_offset[:] = offset
    p.particle_spawn_cube_size[:] = size
    return p, volume
