// device_common.h — shared device-side definitions for the gfx950 fluid-step kernels.
//
// Memory model of one context (one GPU, one Z-slab of the global grid):
//   every grid image holds Dl + 2*IMG_GHOST XY planes, x fastest: IMG_GHOST ghost planes below the Dl
//   owned planes and as many above.  Kernel pointers address owned plane 0, so local z in
//   [-IMG_GHOST, Dl + IMG_GHOST) is always a valid load.  Ghost planes at a domain face stay zero for the lifetime of
//   the context, which IS the reference's out-of-bounds image semantics in z ("load returns 0",
//   SURVEY.md F4); ghost planes between two slabs are filled by the caller's halo exchange.
//   x and y out-of-bounds are tested in the kernels.
//
// Arithmetic contract: fp32, one rounding per GLSL operation, in source order; this translation
// unit is compiled with -ffp-contract=off and IEEE division (hipcc default) so results are
// bit-identical to the CPU oracle (oracle/fluid_oracle.c), which is compiled the same way.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fluid {

// Ghost planes per side of every grid image.  The stencil passes read one; the velocity sampler of
// 07_advect / 14_particles may reach further (a backtrace or a particle can cross a slab face), up
// to this many planes, beyond which the context raises its halo-violation flag.
constexpr int IMG_GHOST = 4;

// Particle slots a Z-slab context does not own carry this bit pattern in w (a quiet NaN, so it is
// never equal to active_particle_w): "tombstones".  Global particle i lives, with its real data, in
// slot i of exactly one rank — the one whose slab holds the cell the particle counts towards.
constexpr uint32_t PARTICLE_TOMBSTONE_BITS = 0x7FC0DEADu;

struct GridK {
    int W, H, Dl;   // local extents (Dl = owned planes)
    int Dg;         // global depth
    int z0;         // global z of local plane 0
    int64_t plane;  // W*H cells
    // Velocity sampler (07_advect, 14_particles) on a Z slab: planes below local plane 0 / above plane
    // Dl-1 of the image being sampled that hold current data of the neighbouring slabs.  A tap beyond
    // them raises the context's halo-violation flag.  IMG_GHOST at most for VELOCITIES_1 itself; the wide
    // source of the fallback pass (fluid_sampler_wide_begin) has as many as the back-traces need.
    int sg_lo, sg_hi;
    // Planes a workgroup of the brick-skipping passes walks (quiet_bricks.h): 1, or BRICK_Z while a step skips
    // bricks — a launch over mostly skipped bricks costs what dispatching its workgroups costs, so there the
    // workgroups are one per brick row instead of one per row of a plane.
    int zl;
};

// Hot-path fields of the 264-byte params block (include/fluid_engine.h: fluid_params), passed by
// value to every kernel instead of a uniform buffer.
struct ParamsK {
    uint32_t t_inactive, t_air, t_water, t_solid;  // blob offsets 16..28
    float dt, p_air, dx, rho;                      // 32..44
    uint32_t spawn_res[3];                         // 64
    uint32_t spawn_volume;                         // 76
    float spawn_offset[3];                         // 80
    float spawn_size[3];                           // 96
    float gravity;                                 // 108
    float diffuse_k;                               // 112
    float active_w;                                // 236
    uint32_t fountain[3];                          // 240
    float fountain_force;                          // 252
    float repel;                                   // 256
};

// Global z plane a particle belongs to: the plane 01_update_densities counts it in (ivec3()
// truncation, update_densities.comp:35), clamped into the grid so every particle has an owner.
__device__ __forceinline__ int particle_owner_plane(float z, int Dg) {
    if (!(z > 0.0f)) return 0;            // also NaN and everything that truncates to <= 0
    if (z >= (float)Dg) return Dg - 1;
    return (int)z;
}

__device__ __forceinline__ int64_t cidx(const GridK& g, int x, int y, int lz) {
    return (int64_t)x + (int64_t)g.W * ((int64_t)y + (int64_t)g.H * (int64_t)lz);
}
__device__ __forceinline__ bool xy_in(const GridK& g, int x, int y) {
    return (unsigned)x < (unsigned)g.W && (unsigned)y < (unsigned)g.H;
}
// imageLoad with robust OOB semantics; lz must be in [-1, Dl].
__device__ __forceinline__ uint32_t type_at(const uint8_t* __restrict__ t, const GridK& g, int x,
                                            int y, int lz) {
    return xy_in(g, x, y) ? (uint32_t)t[cidx(g, x, y, lz)] : 0u;
}
__device__ __forceinline__ float f32_at(const float* __restrict__ a, const GridK& g, int x, int y,
                                        int lz) {
    return xy_in(g, x, y) ? a[cidx(g, x, y, lz)] : 0.0f;
}
__device__ __forceinline__ float4 vel_at(const float4* __restrict__ v, const GridK& g, int x, int y,
                                         int lz) {
    return xy_in(g, x, y) ? v[cidx(g, x, y, lz)] : make_float4(0.f, 0.f, 0.f, 0.f);
}

}  // namespace fluid
