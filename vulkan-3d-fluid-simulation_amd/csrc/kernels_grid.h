// kernels_grid.h — per-cell sections 02…11 and 13 (everything except the Jacobi sweep and the
// particle passes).  One thread per cell, x along the 64 lanes of a wavefront so every row access is
// a coalesced 64-, 256- or 1024-byte segment (R8 / R32 / RGBA32F).  Citations: /root/reference.
#pragma once

#include "device_common.h"
#include "quiet_bricks.h"

namespace fluid {

#define FLUID_CELL_THREAD()                                        \
    const int x = blockIdx.x * blockDim.x + threadIdx.x;           \
    const int y = blockIdx.y * blockDim.y + threadIdx.y;           \
    const int lz = blockIdx.z;                                     \
    if (x >= g.W || y >= g.H) return;                              \
    const int64_t id = cidx(g, x, y, lz);                          \
    const int gz = g.z0 + lz;                                      \
    (void)gz;

// FlowClearColorSection (fluid_flow_sections.h:140,141,163,298,299): 16-byte stores, grid-stride.
__global__ void k_fill_u32x4(uint4* __restrict__ dst, int64_t n_vec, uint4 v) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n_vec; i += stride) dst[i] = v;
}
__global__ void k_fill_u8_tail(uint8_t* __restrict__ dst, int64_t begin, int64_t end, uint8_t v) {
    int64_t i = begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < end) dst[i] = v;
}

// Diagnostics (fluid_count_nonfinite): number of fp32 words that are inf or NaN.  A water cell walled in
// by SOLID on all six sides divides by aii = 0 (pressure.comp:62) and the result spreads through 13 into the
// velocities; the reference never looks.  Grid-stride, one ballot + popcount per wavefront and step, one
// atomic per wavefront.
__global__ void k_count_nonfinite(const uint32_t* __restrict__ words, int64_t n,
                                  unsigned long long* __restrict__ count) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    uint32_t mine = 0;
    for (; i < n; i += stride) mine += (words[i] & 0x7F800000u) == 0x7F800000u ? 1u : 0u;
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off, 64);
    if ((threadIdx.x & 63u) == 0u && mine != 0u) atomicAdd(count, (unsigned long long)mine);
}

// The pressure clears inside fluid_run_step: R32F fill that leaves quiet bricks alone (quiet_bricks.h),
// four cells per thread, launched with cell4_grid() / cell_block().
__global__ void k_fill_f32_unless_quiet(float* __restrict__ dst, float v, GridK g,
                                        const uint8_t* __restrict__ quiet, BrickK bk) {
    FLUID_LEAVE_IF_QUIET_V4(quiet, bk)
    const int x = 4 * (blockIdx.x * blockDim.x + threadIdx.x);
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= g.W || y >= g.H) return;
    FLUID_FOR_PLANES_OF_WORKGROUP()
    *reinterpret_cast<float4*>(dst + cidx(g, x, y, lz)) = make_float4(v, v, v, v);
}

// 02_update_water/update_water.comp:23-33
__global__ void k02_update_water(const uint32_t* __restrict__ dens, uint8_t* __restrict__ newT,
                                 GridK g, ParamsK p) {
    FLUID_CELL_THREAD();
    newT[id] = (uint8_t)(dens[id] > 0u ? p.t_water : p.t_inactive);
}

__device__ __forceinline__ bool on_border(const GridK& g, int x, int y, int gz) {
    return x == 0 || x == g.W - 1 || y == 0 || y == g.H - 1 || gz == 0 || gz == g.Dg - 1;
}

// 03_update_air/update_active.comp:45-66 — in place.  "Solid first" resolution of the reference's
// border race: a neighbour on the domain border never counts as water, so the result
// does not depend on the order in which threads run; AIR writes never change an "is water" answer.
__global__ void k03_update_air(uint8_t* __restrict__ t, GridK g, ParamsK p) {
    FLUID_CELL_THREAD();
    if (on_border(g, x, y, gz)) {  // :50-51
        t[id] = (uint8_t)p.t_solid;
        return;
    }
    if ((uint32_t)t[id] == p.t_water) return;  // :54
    // interior cell: all six neighbours are inside the global grid (lz±1 may be a ghost plane
    // holding the neighbouring slab's cells)
    bool w = false;
    w = w || (!on_border(g, x + 1, y, gz) && (uint32_t)t[cidx(g, x + 1, y, lz)] == p.t_water);
    w = w || (!on_border(g, x, y + 1, gz) && (uint32_t)t[cidx(g, x, y + 1, lz)] == p.t_water);
    w = w || (!on_border(g, x, y, gz + 1) && (uint32_t)t[cidx(g, x, y, lz + 1)] == p.t_water);
    w = w || (!on_border(g, x - 1, y, gz) && (uint32_t)t[cidx(g, x - 1, y, lz)] == p.t_water);
    w = w || (!on_border(g, x, y - 1, gz) && (uint32_t)t[cidx(g, x, y - 1, lz)] == p.t_water);
    w = w || (!on_border(g, x, y, gz - 1) && (uint32_t)t[cidx(g, x, y, lz - 1)] == p.t_water);
    if (w) t[id] = (uint8_t)p.t_air;  // :61-62
}

// ---- four cells per thread for the byte-sized images (fluid_size.x % 4 == 0) ---------------------
// 02 / 03 move one byte per cell; one cell per lane means 64-byte accesses per wavefront.  Here a lane
// handles four consecutive cells: 16-byte density loads, 4-byte type words.
#define FLUID_CELL4_THREAD()                                        \
    const int x = 4 * (blockIdx.x * blockDim.x + threadIdx.x);      \
    const int y = blockIdx.y * blockDim.y + threadIdx.y;            \
    if (x >= g.W || y >= g.H) return;                               \
    FLUID_FOR_PLANES_OF_WORKGROUP() {                               \
        const int64_t id = cidx(g, x, y, lz);                       \
        const int gz = g.z0 + lz;
#define FLUID_CELL4_END }

__global__ void k02_update_water_v4(const uint32_t* __restrict__ dens, uint8_t* __restrict__ newT,
                                    GridK g, ParamsK p, const uint8_t* __restrict__ quiet, BrickK bk) {
    FLUID_LEAVE_IF_QUIET_V4(quiet, bk)  // quiet_bricks.h (02 and 03 skip the same bricks)
    FLUID_CELL4_THREAD();
    (void)gz;
    const uint4 d = *reinterpret_cast<const uint4*>(dens + id);
    const uint32_t a = d.x > 0u ? p.t_water : p.t_inactive, b = d.y > 0u ? p.t_water : p.t_inactive;
    const uint32_t c = d.z > 0u ? p.t_water : p.t_inactive, e = d.w > 0u ? p.t_water : p.t_inactive;
    *reinterpret_cast<uint32_t*>(newT + id) = a | (b << 8) | (c << 16) | (e << 24);
    FLUID_CELL4_END
}

// 06_update_cell_types while bricks are being skipped: NEW_CELL_TYPES -> CELL_TYPES outside the bricks that 02 and
// 03 left alone (there both images hold the previous step's types already; fluid_flow_sections.h:236-241 copies
// the whole image)
__global__ void k06_copy_types_v4(const uint8_t* __restrict__ newT, uint8_t* __restrict__ t, GridK g,
                                  const uint8_t* __restrict__ quiet, BrickK bk) {
    FLUID_LEAVE_IF_QUIET_V4(quiet, bk)
    FLUID_CELL4_THREAD();
    (void)gz;
    *reinterpret_cast<uint32_t*>(t + id) = *reinterpret_cast<const uint32_t*>(newT + id);
    FLUID_CELL4_END
}

__global__ void k03_update_air_v4(uint8_t* __restrict__ t, GridK g, ParamsK p,
                                  const uint8_t* __restrict__ quiet, BrickK bk) {
    FLUID_LEAVE_IF_QUIET_V4(quiet, bk)
    FLUID_CELL4_THREAD();
    auto word = [&](int yy, int llz) -> uint32_t {  // interior rows only: always inside the image
        return *reinterpret_cast<const uint32_t*>(t + cidx(g, x, yy, llz));
    };
    const uint32_t c = *reinterpret_cast<const uint32_t*>(t + id);
    const bool row_border = y == 0 || y == g.H - 1 || gz == 0 || gz == g.Dg - 1;
    uint32_t out = 0;
    if (row_border) {
        out = p.t_solid * 0x01010101u;  // update_active.comp:50-51
    } else {
        // neighbours on the domain border never count as water ("solid first")
        const bool ym_b = y - 1 == 0, yp_b = y + 1 == g.H - 1, zm_b = gz - 1 == 0,
                   zp_b = gz + 1 == g.Dg - 1;
        const uint32_t ym = word(y - 1, lz), yp = word(y + 1, lz);
        const uint32_t zm = word(y, lz - 1), zp = word(y, lz + 1);
        const uint32_t left_ = t[id - (x > 0 ? 1 : 0)], right_ = t[id + (x + 4 < g.W ? 4 : 3)];
        const uint32_t left = x > 0 ? left_ : 0u, right = x + 4 < g.W ? right_ : 0u;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int xi = x + i;
            const uint32_t self = (c >> (8 * i)) & 0xFFu;
            uint32_t r = self;
            if (xi == 0 || xi == g.W - 1) {
                r = p.t_solid;
            } else if (self != p.t_water) {  // :54
                const uint32_t xm = i == 0 ? left : (c >> (8 * (i - 1))) & 0xFFu;
                const uint32_t xp = i == 3 ? right : (c >> (8 * ((i + 1) & 3))) & 0xFFu;
                bool w = false;
                w = w || (xi + 1 != g.W - 1 && xp == p.t_water);
                w = w || (!yp_b && ((yp >> (8 * i)) & 0xFFu) == p.t_water);
                w = w || (!zp_b && ((zp >> (8 * i)) & 0xFFu) == p.t_water);
                w = w || (xi - 1 != 0 && xm == p.t_water);
                w = w || (!ym_b && ((ym >> (8 * i)) & 0xFFu) == p.t_water);
                w = w || (!zm_b && ((zm >> (8 * i)) & 0xFFu) == p.t_water);
                if (w) r = p.t_air;  // :61-62
            }
            out |= r << (8 * i);
        }
    }
    *reinterpret_cast<uint32_t*>(t + id) = out;
    FLUID_CELL4_END
}

// 04_compute_extrapolated_velocities/extrapolated_velocities.comp:37-63
__global__ void k04_extrapolated(const uint8_t* __restrict__ t, const float4* __restrict__ v1,
                                 float4* __restrict__ v2, GridK g, ParamsK p) {
    FLUID_CELL_THREAD();
    int c = 0;
    float sx = 0.f, sy = 0.f, sz = 0.f;
#define FLUID_ACC(cond, nx, ny, nlz)                                         \
    if ((cond) && (uint32_t)t[cidx(g, nx, ny, nlz)] == p.t_water) {          \
        const float4 q = v1[cidx(g, nx, ny, nlz)];                           \
        sx = sx + q.x;                                                       \
        sy = sy + q.y;                                                       \
        sz = sz + q.z;                                                       \
        c++;                                                                 \
    }
    FLUID_ACC(x != 0, x - 1, y, lz)          // :46
    FLUID_ACC(y != 0, x, y - 1, lz)          // :47
    FLUID_ACC(gz != 0, x, y, lz - 1)         // :48
    FLUID_ACC(x != g.W - 1, x + 1, y, lz)    // :49
    FLUID_ACC(y != g.H - 1, x, y + 1, lz)    // :50
    FLUID_ACC(gz != g.Dg - 1, x, y, lz + 1)  // :51
#undef FLUID_ACC
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c != 0) {  // :53
        const float fc = (float)c;
        o.x = sx / fc;
        o.y = sy / fc;
        o.z = sz / fc;
    }
    v2[id] = o;  // :62
}

// 05_set_extrapolated_velocities/extrapolate_velocities.comp:48-109
__global__ void k05_set_extrapolated(const uint8_t* __restrict__ newT,
                                     const uint8_t* __restrict__ oldT,
                                     const float4* __restrict__ v2, float4* __restrict__ v1,
                                     GridK g, ParamsK p) {
    FLUID_CELL_THREAD();
    auto active = [&](uint32_t a) { return a == p.t_water || a == p.t_air; };  // :34-36
    const bool was = active(oldT[id]);                                         // :88
    const bool is = active(newT[id]);                                          // :90
    const float4 base = v1[id];                                                // :93
    const float4 ext = v2[id];                                                 // :95
    float cur[3] = {base.x, base.y, base.z};
    const float ex[3] = {ext.x, ext.y, ext.z};
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const int nx = x - (c == 0), ny = y - (c == 1), nlz = lz - (c == 2);  // :74
        const bool vwas = was || active(type_at(oldT, g, nx, ny, nlz));       // :50
        const bool vis = is || active(type_at(newT, g, nx, ny, nlz));         // :52
        if (vwas && !vis)
            cur[c] = 0.0f;  // VELOCITY_RESET
        else if (!vwas && vis)
            cur[c] = ex[c];  // VELOCITY_EXTRAPOLATE
    }
    v1[id] = make_float4(cur[0], cur[1], cur[2], 0.0f);  // :108
}

// 06_update_cell_types/update_cell_types.comp:15-19 is a plane copy: done with hipMemcpyAsync.

// 08_forces/forces.comp:33-54
__global__ void k08_forces(const uint8_t* __restrict__ t, float4* __restrict__ v2, GridK g,
                           ParamsK p) {
    FLUID_CELL_THREAD();
    const uint32_t t1 = t[id];
    const uint32_t t2 = type_at(t, g, x, y - 1, lz);
    const bool wet = (t1 == p.t_water) || (t2 == p.t_water);
    float fy = 0.0f;
    if (y != 0 && wet) fy += p.gravity;  // :39-45
    if ((uint32_t)x == p.fountain[0] && (uint32_t)y == p.fountain[1] &&
        (uint32_t)gz == p.fountain[2] && wet)
        fy += p.fountain_force;  // :47-49
    if (fy != 0.0f) {            // :52-53
        float4 q = v2[id];
        q.x = q.x + p.dt * 0.0f;
        q.y = q.y + p.dt * fy;
        q.z = q.z + p.dt * 0.0f;
        q.w = q.w + 0.0f;
        v2[id] = q;
    }
}

// 09_diffuse/diffuse.comp:31-46.  INTENDED=false is the shader as written (the diffused value is
// assigned to a shadowing local, :40, so the copy is what is stored, :46).
template <bool INTENDED>
__global__ void k09_diffuse(const uint8_t* __restrict__ t, const float4* __restrict__ v2,
                            float4* __restrict__ v1, GridK g, ParamsK p) {
    FLUID_CELL_THREAD();
    float4 v = v2[id];  // :34
    if (INTENDED && (uint32_t)t[id] == p.t_water) {
        const float a = p.diffuse_k * p.dt;  // :38
        const float k0 = 1.0f - 6.0f * a;
        const float4 xp = vel_at(v2, g, x + 1, y, lz), xm = vel_at(v2, g, x - 1, y, lz);
        const float4 yp = vel_at(v2, g, x, y + 1, lz), ym = vel_at(v2, g, x, y - 1, lz);
        const float4 zp = v2[cidx(g, x, y, lz + 1)], zm = v2[cidx(g, x, y, lz - 1)];
        float sx = xp.x + xm.x, sy = xp.y + xm.y, sz = xp.z + xm.z;  // :41-43 left to right
        sx = sx + yp.x; sy = sy + yp.y; sz = sz + yp.z;
        sx = sx + ym.x; sy = sy + ym.y; sz = sz + ym.z;
        sx = sx + zp.x; sy = sy + zp.y; sz = sz + zp.z;
        sx = sx + zm.x; sy = sy + zm.y; sz = sz + zm.z;
        v.x = k0 * v.x + a * sx;
        v.y = k0 * v.y + a * sy;
        v.z = k0 * v.z + a * sz;
    }
    v.w = 0.0f;
    v1[id] = v;  // :46
}

// 10_solids/solids.comp:30-76
__global__ void k10_solids(const uint8_t* __restrict__ t, float4* __restrict__ v1, GridK g,
                           ParamsK p) {
    FLUID_CELL_THREAD();
    const float r = p.repel;
    const float4 q = v1[id];
    float v[3] = {q.x, q.y, q.z};
    if ((uint32_t)t[id] == p.t_solid) {  // :70-72
#pragma unroll
        for (int c = 0; c < 3; c++)
            if (v[c] > -r) v[c] = -r;  // :32-36
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {  // :73
        const int nx = x - (c == 0), ny = y - (c == 1), nlz = lz - (c == 2);
        if (type_at(t, g, nx, ny, nlz) == p.t_solid && v[c] < r) v[c] = r;  // :50-51
    }
    v1[id] = make_float4(v[0], v[1], v[2], 1.0f);  // :76
}

// 11_compute_divergence/compute_divergence.comp:18-30
__global__ void k11_divergence(const float4* __restrict__ v1, float* __restrict__ div, GridK g) {
    FLUID_CELL_THREAD();
    const float4 v = v1[id];
    const float ax = vel_at(v1, g, x + 1, y, lz).x;
    const float ay = vel_at(v1, g, x, y + 1, lz).y;
    const float az = v1[cidx(g, x, y, lz + 1)].z;
    float d = ax - v.x;  // :21 left to right
    d = d + ay;
    d = d - v.y;
    d = d + az;
    d = d - v.z;
    div[id] = d;
}

// 13_fix_divergence/fix_divergence.comp:41-72
__global__ void k13_fix_divergence(const uint8_t* __restrict__ t, const float* __restrict__ pr,
                                   float4* __restrict__ v1, GridK g, ParamsK p,
                                   const uint8_t* __restrict__ quiet, BrickK bk, int xchunks) {
    FLUID_LEAVE_IF_QUIET(quiet, bk, xchunks)  // quiet_bricks.h
    FLUID_FOR_CELLS_OF_ROW(xchunks)
    const int64_t id = cidx(g, x, y, lz);
    const int gz = g.z0 + lz;
    // all loads first, at addresses that exist whatever the cell (k091011_solids_divergence has the reason)
    const int64_t nb[3] = {id - (x > 0 ? 1 : 0), id - (y > 0 ? g.W : 0), id - g.plane};  // :43
    const uint32_t lt = t[id];  // :62
    const float lp = pr[id];    // :63
    const uint32_t nt[3] = {t[nb[0]], t[nb[1]], t[nb[2]]};  // :44
    const float np[3] = {pr[nb[0]], pr[nb[1]], pr[nb[2]]};
    float4 q = v1[id];
    const float k = (p.dt / p.rho) / p.dx;  // :71
    const int pos[3] = {x, y, gz};
    float dv[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 3; c++) {
        // (pos[c] == 0: no neighbour, or — z — the plane below a slab's face at the domain's floor; :46 skips it)
        const uint32_t ct = nt[c];
        if (pos[c] != 0 && (lt == p.t_water || ct == p.t_water)) {            // :46
            if (lt != p.t_solid && ct != p.t_solid)                           // :48
                dv[c] = lp - np[c];                                           // :50
        }
    }
    q.x = q.x - k * dv[0];
    q.y = q.y - k * dv[1];
    q.z = q.z - k * dv[2];
    q.w = 0.0f;
    v1[id] = q;
    FLUID_END_FOR_CELLS
}

}  // namespace fluid
