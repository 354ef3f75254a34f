// kernels_surface.h — the surface-prep passes 15…18: particle counts on the
// "detailed" grid (detailed_resolution^3 cells per simulation cell), their temporal inertia filter, the
// conversion to float and the 7-point blur loop that feeds the reference's marching-cubes renderer
// (/root/reference/fluid_flow_sections.h:339-388).  Rendering itself stays out of scope; these passes
// are the last compute sections of the reference's step list.
//
// Detailed images are linear, x fastest, without ghost planes (whole-grid contexts only): DETAILED_
// DENSITIES_IMG and DETAILED_DENSITIES_INERTIA_IMG R32UI, PARTICLE_DENSITIES_FLOAT_1/2 R32F.  Out-of-
// bounds loads return 0, stores / atomics outside the image are dropped.  All four passes are
// bandwidth-bound streaming or stencil passes: one thread per cell, x on the lanes (256-B coalesced rows).
#pragma once

#include "device_common.h"

namespace fluid {

struct SurfK {
    int W, H, D, res;  // detailed extents, detailed_resolution
    int sW, sH;        // simulation grid extents (cell types)
    int64_t plane;     // W * H
};
__device__ __forceinline__ int64_t sidx(const SurfK& s, int x, int y, int z) {
    return (int64_t)x + (int64_t)s.W * ((int64_t)y + (int64_t)s.H * (int64_t)z);
}
__device__ __forceinline__ bool s_in(const SurfK& s, int x, int y, int z) {
    return (unsigned)x < (unsigned)s.W && (unsigned)y < (unsigned)s.H && (unsigned)z < (unsigned)s.D;
}

#define FLUID_SURF_THREAD()                                     \
    const int x = blockIdx.x * blockDim.x + threadIdx.x;        \
    const int y = blockIdx.y * blockDim.y + threadIdx.y;        \
    const int z = blockIdx.z;                                   \
    if (x >= s.W || y >= s.H) return;                           \
    const int64_t id = sidx(s, x, y, z);

// truncation toward zero of a particle coordinate scaled to the detailed grid; false = outside the image
// (same rule as 01_update_densities: v in (-1, N); NaN / inf are dropped)
__device__ __forceinline__ bool trunc_detailed(float v, int n, int& out) {
    if (!(v > -1.0f && v < (float)n)) return false;
    out = (int)v;
    return true;
}

// 15_update_detailed_densities/update_detailed_densities.comp:24-31 — one thread per particle, one global
// atomic each: the 8 particles of a simulation cell spread over its res^3 detailed cells, so there is
// little to combine (unlike 01)
__global__ void k15_update_detailed_densities(const float4* __restrict__ particles, uint64_t capacity,
                                              uint32_t* __restrict__ detailed, SurfK s, float active_w) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= capacity) return;
    const float4 q = particles[i];
    if (!(q.w == active_w)) return;  // :28
    const float fres = (float)s.res;
    int x, y, z;
    if (trunc_detailed(q.x * fres, s.W, x) && trunc_detailed(q.y * fres, s.H, y) &&
        trunc_detailed(q.z * fres, s.D, z))
        atomicAdd(&detailed[sidx(s, x, y, z)], 1u);  // :30
}

struct InertiaK {
    uint32_t max_inertia, increase_filled, increase_neighbour, decrease;
    int required_hits, increase_neighbour_i;
};

// 16_compute_detailed_densities_inertia/densities_inertia.comp:30-61 (in place on the inertia image: every
// thread reads and writes its own texel only; the neighbours it looks at are densities)
__global__ void k16_detailed_densities_inertia(const uint32_t* __restrict__ detailed,
                                               uint32_t* __restrict__ inertia, SurfK s, InertiaK k) {
    FLUID_SURF_THREAD();
    uint32_t in = inertia[id];  // :38
    const uint32_t old = in;
    if (detailed[id] > 0u) in += k.increase_filled;  // :42-44
    int hits = 0;
    auto filled = [&](int nx, int ny, int nz) {
        return s_in(s, nx, ny, nz) && detailed[sidx(s, nx, ny, nz)] > 0u;
    };
    hits += filled(x + 1, y, z) ? 1 : 0;  // :49-52, the order does not matter for a count
    hits += filled(x, y + 1, z) ? 1 : 0;
    hits += filled(x, y, z + 1) ? 1 : 0;
    hits += filled(x - 1, y, z) ? 1 : 0;
    hits += filled(x, y - 1, z) ? 1 : 0;
    hits += filled(x, y, z - 1) ? 1 : 0;
    if (hits >= k.required_hits) in += (uint32_t)(hits * k.increase_neighbour_i);  // :54
    if (in == old) {                                                                // :57-63
        if (in > k.decrease)
            in -= k.decrease;
        else
            in = 0u;
    }
    inertia[id] = min(k.max_inertia, in);  // :65
}

// 17_compute_float_densities/float_densities.comp:22-27
__global__ void k17_float_densities(const uint32_t* __restrict__ inertia, float* __restrict__ f1, SurfK s,
                                    float coefficient) {
    FLUID_SURF_THREAD();
    const uint32_t d = inertia[id];
    f1[id] = d == 0u ? -1.0f : (float)d / coefficient;
}

// 18_diffuse_float_densities/diffuse_densities.comp:45-62 — one dispatch src -> dst; detailed cells whose
// simulation cell is SOLID are not written.  ( 1.0 - 6 a) * d(i) + a * (((((+x) + (-x)) + (+y)) + (-y)) + (+z)) + (-z))
__global__ void k18_diffuse_float_densities(const uint8_t* __restrict__ types,
                                            const float* __restrict__ src, float* __restrict__ dst,
                                            SurfK s, float a, uint32_t t_solid) {
    FLUID_SURF_THREAD();
    // the cell-type image of the simulation grid has IMG_GHOST ghost planes below plane 0; `types`
    // addresses plane 0
    const int cx = x / s.res, cy = y / s.res, cz = z / s.res;  // :56
    const uint32_t t = types[(int64_t)cx + (int64_t)s.sW * ((int64_t)cy + (int64_t)s.sH * (int64_t)cz)];
    if (t == t_solid) return;
    auto ld = [&](int nx, int ny, int nz) { return s_in(s, nx, ny, nz) ? src[sidx(s, nx, ny, nz)] : 0.0f; };
    float sum = ld(x + 1, y, z) + ld(x - 1, y, z);
    sum = sum + ld(x, y + 1, z);
    sum = sum + ld(x, y - 1, z);
    sum = sum + ld(x, y, z + 1);
    sum = sum + ld(x, y, z - 1);
    const float k0 = 1.0f - 6.0f * a;
    const float t1 = k0 * src[id];
    const float t2 = a * sum;
    dst[id] = t1 + t2;
}

}  // namespace fluid
