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

// ---- four cells per thread (detailed width % 4 == 0): 16-byte accesses along x, the y / z neighbours as
// whole float4 / uint4 rows from L1 / L2, the two x neighbours outside the quad as scalars ---------------
#define FLUID_SURF4_THREAD()                                    \
    const int x = 4 * (blockIdx.x * blockDim.x + threadIdx.x);  \
    const int y = blockIdx.y * blockDim.y + threadIdx.y;        \
    const int z = blockIdx.z;                                   \
    if (x >= s.W || y >= s.H) return;                           \
    const int64_t id = sidx(s, x, y, z);

// 16 (+ 17 when f1 != nullptr: the float image is a pointwise function of the inertia just computed)
__global__ void k16_detailed_densities_inertia_v4(const uint32_t* __restrict__ detailed,
                                                  uint32_t* __restrict__ inertia,
                                                  float* __restrict__ f1, SurfK s, InertiaK k,
                                                  float coefficient) {
    FLUID_SURF4_THREAD();
    // (a select between a loaded vector and a named zero constant makes hipcc keep the constant in scratch)
    auto row = [&](int ny, int nz) {
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if ((unsigned)ny < (unsigned)s.H && (unsigned)nz < (unsigned)s.D)
            v = *reinterpret_cast<const uint4*>(detailed + sidx(s, x, ny, nz));
        return v;
    };
    const uint4 c = *reinterpret_cast<const uint4*>(detailed + id);
    const uint4 yp = row(y + 1, z), ym = row(y - 1, z), zp = row(y, z + 1), zm = row(y, z - 1);
    const uint32_t xl = x > 0 ? detailed[id - 1] : 0u, xr = x + 4 < s.W ? detailed[id + 4] : 0u;
    const uint4 old = *reinterpret_cast<const uint4*>(inertia + id);
    const uint32_t cc[4] = {c.x, c.y, c.z, c.w}, oo[4] = {old.x, old.y, old.z, old.w};
    const uint32_t a_yp[4] = {yp.x, yp.y, yp.z, yp.w}, a_ym[4] = {ym.x, ym.y, ym.z, ym.w};
    const uint32_t a_zp[4] = {zp.x, zp.y, zp.z, zp.w}, a_zm[4] = {zm.x, zm.y, zm.z, zm.w};
    uint32_t out[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint32_t in = oo[i];
        if (cc[i] > 0u) in += k.increase_filled;
        const uint32_t nxp = i == 3 ? xr : cc[i == 3 ? 3 : i + 1];
        const uint32_t nxm = i == 0 ? xl : cc[i == 0 ? 0 : i - 1];
        int hits = 0;
        hits += nxp > 0u ? 1 : 0;
        hits += a_yp[i] > 0u ? 1 : 0;
        hits += a_zp[i] > 0u ? 1 : 0;
        hits += nxm > 0u ? 1 : 0;
        hits += a_ym[i] > 0u ? 1 : 0;
        hits += a_zm[i] > 0u ? 1 : 0;
        if (hits >= k.required_hits) in += (uint32_t)(hits * k.increase_neighbour_i);
        if (in == oo[i]) {
            if (in > k.decrease)
                in -= k.decrease;
            else
                in = 0u;
        }
        out[i] = min(k.max_inertia, in);
    }
    *reinterpret_cast<uint4*>(inertia + id) = make_uint4(out[0], out[1], out[2], out[3]);
    if (f1) {  // 17_compute_float_densities on the values just stored
        float4 f;
        f.x = out[0] == 0u ? -1.0f : (float)out[0] / coefficient;
        f.y = out[1] == 0u ? -1.0f : (float)out[1] / coefficient;
        f.z = out[2] == 0u ? -1.0f : (float)out[2] / coefficient;
        f.w = out[3] == 0u ? -1.0f : (float)out[3] / coefficient;
        *reinterpret_cast<float4*>(f1 + id) = f;
    }
}

__global__ void k18_diffuse_float_densities_v4(const uint8_t* __restrict__ types,
                                               const float* __restrict__ src, float* __restrict__ dst,
                                               SurfK s, float a, uint32_t t_solid) {
    FLUID_SURF4_THREAD();
    auto row = [&](int ny, int nz) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((unsigned)ny < (unsigned)s.H && (unsigned)nz < (unsigned)s.D)
            v = *reinterpret_cast<const float4*>(src + sidx(s, x, ny, nz));
        return v;
    };
    const float4 c = *reinterpret_cast<const float4*>(src + id);
    const float4 yp = row(y + 1, z), ym = row(y - 1, z), zp = row(y, z + 1), zm = row(y, z - 1);
    const float xl = x > 0 ? src[id - 1] : 0.0f, xr = x + 4 < s.W ? src[id + 4] : 0.0f;
    const float cc[4] = {c.x, c.y, c.z, c.w};
    const float a_yp[4] = {yp.x, yp.y, yp.z, yp.w}, a_ym[4] = {ym.x, ym.y, ym.z, ym.w};
    const float a_zp[4] = {zp.x, zp.y, zp.z, zp.w}, a_zm[4] = {zm.x, zm.y, zm.z, zm.w};
    const int cy = y / s.res, cz = z / s.res;
    const int64_t trow = (int64_t)s.sW * ((int64_t)cy + (int64_t)s.sH * (int64_t)cz);
    const float k0 = 1.0f - 6.0f * a;
    bool solid[4];
    bool any_solid = false;
    int cx = x / s.res, rem = x - cx * s.res;  // one division per thread: (x + i) / res by carrying
#pragma unroll
    for (int i = 0; i < 4; i++) {
        solid[i] = (uint32_t)types[trow + cx] == t_solid;
        any_solid = any_solid || solid[i];
        if (++rem == s.res) {
            rem = 0;
            cx++;
        }
    }
    // cells of SOLID simulation cells are not written: they keep what the destination holds
    float4 old = make_float4(0.f, 0.f, 0.f, 0.f);
    if (any_solid) old = *reinterpret_cast<const float4*>(dst + id);
    float out[4] = {old.x, old.y, old.z, old.w};
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (solid[i]) continue;
        const float nxp = i == 3 ? xr : cc[i == 3 ? 3 : i + 1];
        const float nxm = i == 0 ? xl : cc[i == 0 ? 0 : i - 1];
        float sum = nxp + nxm;
        sum = sum + a_yp[i];
        sum = sum + a_ym[i];
        sum = sum + a_zp[i];
        sum = sum + a_zm[i];
        const float t1 = k0 * cc[i];
        const float t2 = a * sum;
        out[i] = t1 + t2;
    }
    *reinterpret_cast<float4*>(dst + id) = make_float4(out[0], out[1], out[2], out[3]);
}

// 18 with a z march: a workgroup of 64 x 8 threads owns 256 x 8 cells of an XY tile and walks `zchunk`
// planes.  Every thread keeps the z-1 / z / z+1 values of its four cells in registers and publishes the
// current plane's in LDS (double-buffered, one barrier per plane), where its neighbours find their y and x
// neighbours; only the tile's two halo rows and two edge columns come from global memory.  Each texel of
// the source then leaves HBM about 1.3 times per dispatch (PMC: the version that loaded the y neighbours
// from global memory fetched 2.8 times — three planes of every resident workgroup do not stay in a 4-MiB
// L2 next to the output stream).  Same arithmetic as k18_diffuse_float_densities.
constexpr int K18_ROWS = 8;
constexpr int K18_PAD = 4;                      // floats left of x = 0 of the tile (keeps rows 16-B aligned)
constexpr int K18_ROW_FLOATS = 256 + 2 * K18_PAD;
__global__ void __launch_bounds__(64 * K18_ROWS)
k18_diffuse_float_densities_zmarch(const uint8_t* __restrict__ types, const float* __restrict__ src,
                                   float* __restrict__ dst, SurfK s, float a, uint32_t t_solid,
                                   int zchunk) {
    __shared__ __attribute__((aligned(16))) float tile[2][K18_ROWS + 2][K18_ROW_FLOATS];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int x = 4 * (blockIdx.x * 64 + tx);
    const int y = blockIdx.y * K18_ROWS + ty;
    const bool valid = x < s.W && y < s.H;  // invalid threads still take part in the barriers
    const int zb = blockIdx.z * zchunk, ze = min(zb + zchunk, s.D);
    const float k0 = 1.0f - 6.0f * a;
    const int64_t rowoff = (int64_t)x + (int64_t)s.W * (int64_t)y;
    auto plane_ld = [&](int z) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid && (unsigned)z < (unsigned)s.D) v = *reinterpret_cast<const float4*>(src + rowoff + s.plane * z);
        return v;
    };
    // halo rows of the tile: the first / last thread row also loads the row below / above the tile
    const bool halo_row = ty == 0 || ty == K18_ROWS - 1;
    const int yh = ty == 0 ? y - 1 : y + 1;
    const bool halo_in = halo_row && x < s.W && (unsigned)yh < (unsigned)s.H;
    const int64_t halo_off = (int64_t)x + (int64_t)s.W * (int64_t)(halo_in ? yh : 0);
    // edge columns: lane 0 supplies x - 1, lane 63 supplies x + 4 of its row
    const bool edge = tx == 0 || tx == 63;
    const int xe = tx == 0 ? x - 1 : x + 4;
    const bool edge_in = edge && valid && (unsigned)xe < (unsigned)s.W;
    // this thread's simulation-cell columns (x is a multiple of 4; (x + i) / res by carrying)
    const int cx0 = x / s.res, rem0 = x - cx0 * s.res, cy = y / s.res;
    float4 zm = plane_ld(zb - 1), c = plane_ld(zb), zp = plane_ld(zb + 1);
    for (int z = zb; z < ze; z++) {
        const int buf = z & 1;
        const float4 zp2 = plane_ld(z + 2);  // two planes ahead: more bytes in flight per wavefront
        const int64_t id = rowoff + s.plane * z;
        // publish plane z of this tile
        float* row = &tile[buf][ty + 1][K18_PAD + 4 * tx];
        *reinterpret_cast<float4*>(row) = c;
        if (halo_row) {
            float4 h = make_float4(0.f, 0.f, 0.f, 0.f);
            if (halo_in) h = *reinterpret_cast<const float4*>(src + halo_off + s.plane * z);
            *reinterpret_cast<float4*>(&tile[buf][ty == 0 ? 0 : K18_ROWS + 1][K18_PAD + 4 * tx]) = h;
        }
        if (edge) row[tx == 0 ? -1 : 4] = edge_in ? src[id + (tx == 0 ? -1 : 4)] : 0.0f;
        __syncthreads();
        if (valid) {
            // rows outside the image hold 0 (the halo loads above / plane_ld of invalid rows)
            const float4 yp = *reinterpret_cast<const float4*>(&tile[buf][ty + 2][K18_PAD + 4 * tx]);
            const float4 ym = *reinterpret_cast<const float4*>(&tile[buf][ty][K18_PAD + 4 * tx]);
            const float xl = row[-1], xr = row[4];
            const int64_t trow = (int64_t)s.sW * ((int64_t)cy + (int64_t)s.sH * (int64_t)(z / s.res));
            bool solid[4];
            bool any_solid = false;
            int cx = cx0, rem = rem0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                solid[i] = (uint32_t)types[trow + cx] == t_solid;
                any_solid = any_solid || solid[i];
                if (++rem == s.res) {
                    rem = 0;
                    cx++;
                }
            }
            float4 old = make_float4(0.f, 0.f, 0.f, 0.f);
            if (any_solid) old = *reinterpret_cast<const float4*>(dst + id);
            const float cc[4] = {c.x, c.y, c.z, c.w};
            const float a_yp[4] = {yp.x, yp.y, yp.z, yp.w}, a_ym[4] = {ym.x, ym.y, ym.z, ym.w};
            const float a_zp[4] = {zp.x, zp.y, zp.z, zp.w}, a_zm[4] = {zm.x, zm.y, zm.z, zm.w};
            float out[4] = {old.x, old.y, old.z, old.w};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (solid[i]) continue;
                const float nxp = i == 3 ? xr : cc[i == 3 ? 3 : i + 1];
                const float nxm = i == 0 ? xl : cc[i == 0 ? 0 : i - 1];
                float sum = nxp + nxm;
                sum = sum + a_yp[i];
                sum = sum + a_ym[i];
                sum = sum + a_zp[i];
                sum = sum + a_zm[i];
                const float t1 = k0 * cc[i];
                const float t2 = a * sum;
                out[i] = t1 + t2;
            }
            *reinterpret_cast<float4*>(dst + id) = make_float4(out[0], out[1], out[2], out[3]);
        }
        zm = c;
        c = zp;
        zp = zp2;
    }
}

// ---- two dispatches of 18 per pass over HBM ---------------------------------------------------------------
// The loop ping-pongs FLOAT_1 -> FLOAT_2 -> FLOAT_1 ... (fluid_flow_sections.h:376-388), and a dispatch does
// not write the cells of SOLID simulation cells: they keep what the image it writes held before, so as
// neighbours they contribute stale values — FLOAT_2's own for the odd iterates, FLOAT_1's (what 17 stored,
// never touched by the loop) for the even ones.  k18_pair applies dispatches 2j and 2j+1 in one z march:
//   `src`  a complete image of iterate 2j (solid cells: FLOAT_1's),
//   `mid`  FLOAT_2: read at solid cells (iterate 2j+1 there = its stale value); STORE_MID also writes iterate
//          2j+1 to it (the loop's last pair: FLOAT_2 ends with the last odd iterate),
//   `dst`  receives iterate 2j+2 at every cell (solid cells: src's value), i.e. is complete again.
// src and dst must differ (other workgroups still read src), so the pairs alternate between FLOAT_1 and a
// third image; iterate 2j+1 lives in registers and LDS only.
// A workgroup of 64 x R threads holds 256 x R cells: every thread forms iterate 2j+1 for its four cells one
// plane ahead (stage 1), the inner 62 x (R-2) threads form iterate 2j+2 (stage 2) — rows and float4 columns
// overlap by two between neighbouring workgroups, iterate 2j+1 is recomputed there, never exchanged, so the
// results are those of two dispatches.  z neighbours come from the thread's registers (three planes per
// iterate), y / x neighbours from the planes both iterates publish in LDS (double-buffered, one barrier per
// plane).  Cells outside the image are 0 in every iterate (imageLoad out of bounds).
constexpr int K18_PAIR_ROWS = 12;
template <int R, bool STORE_MID>
__global__ void __launch_bounds__(64 * R)
k18_pair(const uint8_t* __restrict__ types, const float* __restrict__ src, float* __restrict__ mid,
         float* __restrict__ dst, SurfK s, float a, uint32_t t_solid, int zchunk) {
    __shared__ __attribute__((aligned(16))) float tile_a[2][R + 2][K18_ROW_FLOATS];  // iterate 2j
    __shared__ __attribute__((aligned(16))) float tile_s[2][R][K18_ROW_FLOATS];      // iterate 2j+1
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int x = (int)blockIdx.x * 248 - 4 + 4 * tx;
    const int y = (int)blockIdx.y * (R - 2) - 1 + ty;
    const bool valid = (unsigned)x < (unsigned)s.W && (unsigned)y < (unsigned)s.H;
    const bool writer = valid && tx >= 1 && tx <= 62 && ty >= 1 && ty <= R - 2;
    const int zb = (int)blockIdx.z * zchunk, ze = min(zb + zchunk, s.D);
    const float k0 = 1.0f - 6.0f * a;
    const int64_t rowoff = (int64_t)x + (int64_t)s.W * (int64_t)y;
    auto plane_ld = [&](int z) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid && (unsigned)z < (unsigned)s.D) v = *reinterpret_cast<const float4*>(src + rowoff + s.plane * z);
        return v;
    };
    // halo rows of iterate 2j: the first / last thread row also loads the row below / above the workgroup's
    const bool halo_row = ty == 0 || ty == R - 1;
    const int yh = ty == 0 ? y - 1 : y + 1;
    const bool halo_in = halo_row && (unsigned)x < (unsigned)s.W && (unsigned)yh < (unsigned)s.H;
    const int64_t halo_off = (int64_t)x + (int64_t)s.W * (int64_t)(halo_in ? yh : 0);
    const int cx0 = valid ? x / s.res : 0, rem0 = valid ? x - cx0 * s.res : 0, cy = valid ? y / s.res : 0;
    // which of the thread's four cells lie in SOLID simulation cells of detailed plane z
    auto solid_bits = [&](int z) {
        uint32_t m = 0u;
        if (valid && (unsigned)z < (unsigned)s.D) {
            const int64_t trow = (int64_t)s.sW * ((int64_t)cy + (int64_t)s.sH * (int64_t)(z / s.res));
            int cx = cx0, rem = rem0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if ((uint32_t)types[trow + cx] == t_solid) m |= 1u << i;
                if (++rem == s.res) {
                    rem = 0;
                    cx++;
                }
            }
        }
        return m;
    };
    // one dispatch for the thread's four cells: centre c, z neighbours zm / zp, y / x neighbours from `row`
    // (the thread's row of the published plane; rows are K18_ROW_FLOATS apart)
    auto blur4 = [&](const float4& c, const float4& zm, const float4& zp, const float* row, uint32_t solid,
                     const float4& at_solid) {
        const float4 yp = *reinterpret_cast<const float4*>(row + K18_ROW_FLOATS);
        const float4 ym = *reinterpret_cast<const float4*>(row - K18_ROW_FLOATS);
        const float xl = row[-1], xr = row[4];
        const float cc[4] = {c.x, c.y, c.z, c.w};
        const float a_yp[4] = {yp.x, yp.y, yp.z, yp.w}, a_ym[4] = {ym.x, ym.y, ym.z, ym.w};
        const float a_zp[4] = {zp.x, zp.y, zp.z, zp.w}, a_zm[4] = {zm.x, zm.y, zm.z, zm.w};
        float out[4] = {at_solid.x, at_solid.y, at_solid.z, at_solid.w};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (solid & (1u << i)) continue;
            const float nxp = i == 3 ? xr : cc[i == 3 ? 3 : i + 1];
            const float nxm = i == 0 ? xl : cc[i == 0 ? 0 : i - 1];
            float sum = nxp + nxm;  // diffuse_densities.comp:58-60, left to right
            sum = sum + a_yp[i];
            sum = sum + a_ym[i];
            sum = sum + a_zp[i];
            sum = sum + a_zm[i];
            const float t1 = k0 * cc[i];
            const float t2 = a * sum;
            out[i] = t1 + t2;
        }
        return make_float4(out[0], out[1], out[2], out[3]);
    };
    // Step k forms iterate 2j+1 of plane k+1 and (from k = zb on) iterate 2j+2 of plane k.
    float4 a_m = plane_ld(zb - 2), a_c = plane_ld(zb - 1), a_p = plane_ld(zb);  // iterate 2j: planes k, k+1, k+2
    float4 s_m = make_float4(0.f, 0.f, 0.f, 0.f), s_c = s_m;                    // iterate 2j+1: planes k-1, k
    uint32_t solid_c = 0u;  // SOLID cells of plane k (what stage 1 found for it a step ago)
    for (int k = zb - 2; k < ze; k++) {
        const int buf = k & 1;
        const float4 a_next = plane_ld(k + 3);
        // publish iterate 2j of plane k+1 (with the halo rows) and iterate 2j+1 of plane k
        float* row_a = &tile_a[buf][ty + 1][K18_PAD + 4 * tx];
        *reinterpret_cast<float4*>(row_a) = a_c;
        if (halo_row) {
            float4 h = make_float4(0.f, 0.f, 0.f, 0.f);
            if (halo_in && (unsigned)(k + 1) < (unsigned)s.D)
                h = *reinterpret_cast<const float4*>(src + halo_off + s.plane * (k + 1));
            *reinterpret_cast<float4*>(&tile_a[buf][ty == 0 ? 0 : R + 1][K18_PAD + 4 * tx]) = h;
        }
        float* row_s = &tile_s[buf][ty][K18_PAD + 4 * tx];
        *reinterpret_cast<float4*>(row_s) = s_c;
        __syncthreads();
        // stage 1: plane k+1
        float4 s_p = make_float4(0.f, 0.f, 0.f, 0.f);
        // a simulation cell spans `res` detailed planes: the mask changes only where a new one begins
        uint32_t solid_p = solid_c;
        if (k == zb - 2 || (k + 1) % s.res == 0) solid_p = solid_bits(k + 1);
        if (valid && (unsigned)(k + 1) < (unsigned)s.D) {
            const uint32_t solid = solid_p;
            float4 stale = make_float4(0.f, 0.f, 0.f, 0.f);
            if (solid) stale = *reinterpret_cast<const float4*>(mid + rowoff + s.plane * (k + 1));
            s_p = blur4(a_c, a_m, a_p, row_a, solid, stale);
            if (STORE_MID && writer && k + 1 >= zb && k + 1 < ze)
                *reinterpret_cast<float4*>(mid + rowoff + s.plane * (k + 1)) = s_p;
        }
        // stage 2: plane k (rows 0 and R-1 and lanes 0 and 63 only feed their neighbours)
        if (writer && k >= zb) {
            const float4 o = blur4(s_c, s_m, s_p, row_s, solid_c, a_m);
            *reinterpret_cast<float4*>(dst + rowoff + s.plane * k) = o;
        }
        a_m = a_c;
        a_c = a_p;
        a_p = a_next;
        s_m = s_c;
        s_c = s_p;
        solid_c = solid_p;
    }
}

// ---- 31_render_surface as a triangle list (offline visualisation) -------------------
// What the reference's marching-cubes geometry shader emits (render_surface.vert:19-25, render_surface.geom:
// 45-103), stored instead of rasterised: one thread per render cell ((W-1) x (H-1) x (D-1) cells of the
// detailed grid), configuration bit i = density(corner i) > 0, counts[configuration] triangles whose vertices
// lie on the cell edges named by edge_indices[configuration * 15 + ...] at a = d0 / (d0 - d1), flat normal
// v / sqrt(dot(v, v)) of cross(p1 - p0, p2 - p0) (normalize() as defined in oracle_31_extract_surface).
// 12 floats per triangle {p0, p1, p2, N}, appended through one atomic per cell with triangles: the order of
// the list is not defined (the reference draws, it does not order either).  Triangles beyond `capacity` are
// counted, not stored.
__global__ void k31_extract_surface(const float* __restrict__ density, SurfK s,
                                    const uint32_t* __restrict__ counts,
                                    const uint32_t* __restrict__ edge_indices, float* __restrict__ out,
                                    unsigned long long capacity, unsigned long long* __restrict__ total) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    const int z = blockIdx.z;
    if (x >= s.W - 1 || y >= s.H - 1) return;
    const int mv[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};
    const int ed[12][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 0}, {4, 5}, {5, 6}, {6, 7}, {7, 4}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};
    float d[8];
    int cfg = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        d[i] = density[sidx(s, x + mv[i][0], y + mv[i][1], z + mv[i][2])];
        cfg |= (d[i] > 0.0f ? 1 : 0) << i;  // :93
    }
    const uint32_t tris = counts[cfg];
    if (tris == 0) return;
    const unsigned long long first = atomicAdd(total, (unsigned long long)tris);
    const float fres = (float)s.res;
    const float cell[3] = {(float)x, (float)y, (float)z};
    for (uint32_t t = 0; t < tris; t++) {
        if (first + t >= capacity) return;
        float pt[3][3];
        for (int i = 0; i < 3; i++) {
            const uint32_t e = edge_indices[cfg * 15 + 3 * t + i];  // :60
            const int e0 = ed[e][0], e1 = ed[e][1];
            const float a = d[e0] / (d[e0] - d[e1]);                 // :64
            for (int c = 0; c < 3; c++) {
                float v = 0.5f + cell[c];                            // :66, left to right
                v = v + (float)mv[e0][c];
                v = v + (float)(mv[e1][c] - mv[e0][c]) * a;
                pt[i][c] = v / fres;
            }
        }
        float u[3], w[3], c3[3];
        for (int c = 0; c < 3; c++) {
            u[c] = pt[1][c] - pt[0][c];
            w[c] = pt[2][c] - pt[0][c];
        }
        c3[0] = u[1] * w[2] - w[1] * u[2];  // cross(), :69
        c3[1] = u[2] * w[0] - w[2] * u[0];
        c3[2] = u[0] * w[1] - w[0] * u[1];
        const float len = sqrtf((c3[0] * c3[0] + c3[1] * c3[1]) + c3[2] * c3[2]);
        float* o = out + 12ull * (first + t);
        for (int i = 0; i < 3; i++)
            for (int c = 0; c < 3; c++) o[3 * i + c] = pt[i][c];
        for (int c = 0; c < 3; c++) o[9 + c] = c3[c] / len;
    }
}

}  // namespace fluid
