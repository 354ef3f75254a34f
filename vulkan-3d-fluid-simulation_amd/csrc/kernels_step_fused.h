// kernels_step_fused.h — section groups of one whole step (fluid_run_step) executed as single passes.
//
// The section list of the reference (fluid_flow_sections.h:163-338) moves the RGBA32F velocity images
// through HBM nine times per step.  Run as a list, each section is its own kernel (kernels_grid.h,
// kernels_sampler.h) and that is what fluid_run_section() does.  fluid_run_step() knows the whole
// list, so it may group sections whose intermediate images nobody can observe, as long as every image
// holds, when the step ends, exactly the bits the list would have left:
//
//   04 + 05   04 writes the extrapolated velocity of EVERY cell to VELOCITIES_2, 05 consumes it only in
//             the cells whose activity changed, and 07 then overwrites all of VELOCITIES_2.  Here the
//             extrapolation is evaluated only where 05 will use it (k0405_extrapolate), and 05 touches
//             VELOCITIES_1 only where a component changes (k0405_apply): two scans of the 1-byte type
//             images instead of four passes over 16-byte velocities.  Requires VELOCITIES_1.w == 0
//             everywhere on entry (true after 13_fix_divergence / the init clear; tracked by the engine),
//             because 05 also stores w = 0.
//   07 + 08   forces are a pointwise update of what advection just produced (k07_advect<true>).
//   09 + 10 + 11   09 (as written: a copy) and 10 are pointwise, 11 differences the result with its
//             +x/+y/+z neighbours: one pass reads VELOCITIES_2, applies 10 to the cell and to the three
//             neighbour components it needs, writes VELOCITIES_1 and DIVERGENCES.
//
// Arithmetic is the per-section kernels' (same operations, same order): results are bit-identical,
// which tests/test_engine_parity_gpu.py checks against the oracle and against the section list.
// On a Z slab the caller exchanges ghost planes around a group instead of inside it: 04+05 and 07+08
// need what 04 / 07 need; 09+10+11 needs one ghost plane of VELOCITIES_2 (and of CELL_TYPES) above the
// slab instead of one of VELOCITIES_1 between 10 and 11.
#pragma once

#include "device_common.h"
#include "quiet_bricks.h"

namespace fluid {

// ---- quiet bricks (quiet_bricks.h) ----------------------------------------------------------------
// A Z-slab context does not hold the bricks beyond its faces.  `edge`: for each face (0 = below, 1 =
// above) how to treat them: EDGE_NONE = domain face, nothing there; EDGE_GHOST = the neighbouring slab's
// edge layer of the same map is in ghost[face] (one byte per brick column, exchanged by the caller);
// EDGE_UNKNOWN = a neighbour exists but its layer is not known: counts as holding water.
enum : int { EDGE_NONE = 0, EDGE_GHOST = 1, EDGE_UNKNOWN = 2 };
struct BrickEdges {
    const uint8_t* ghost[2];
    int kind[2];
};
__device__ __forceinline__ uint32_t brick_neighbourhood(const uint8_t* __restrict__ a,
                                                        const uint8_t* __restrict__ b, const BrickK& bk,
                                                        const BrickEdges& e, int bx, int by, int bz) {
    uint32_t any = 0;
    for (int dz = -1; dz <= 1; dz++)
        for (int dy = -1; dy <= 1; dy++)
            for (int dx = -1; dx <= 1; dx++) {
                const int x = bx + dx, y = by + dy, z = bz + dz;
                if ((unsigned)x >= (unsigned)bk.nbx || (unsigned)y >= (unsigned)bk.nby) continue;
                if ((unsigned)z < (unsigned)bk.nbz) {
                    const int j = brick_index(bk, x, y, z);
                    any |= (uint32_t)a[j] | (b ? (uint32_t)b[j] : 0u);
                } else {
                    const int face = z < 0 ? 0 : 1;
                    if (e.kind[face] == EDGE_UNKNOWN) any |= 1u;
                    if (e.kind[face] == EDGE_GHOST) any |= (uint32_t)e.ghost[face][x + bk.nbx * y];
                }
            }
    return any;
}

__global__ void k_update_quiet(const uint8_t* __restrict__ active, uint8_t* __restrict__ streak,
                               BrickK bk, uint32_t* __restrict__ quiet_count, BrickEdges edges) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = bk.nbx * bk.nby * bk.nbz;
    if (i >= n) return;
    const int bx = i % bk.nbx, by = (i / bk.nbx) % bk.nby, bz = i / (bk.nbx * bk.nby);
    const uint32_t any = brick_neighbourhood(active, nullptr, bk, edges, bx, by, bz);
    const uint32_t s = streak[i];
    const uint32_t now = any ? 0u : (s < 255u ? s + 1u : 255u);
    streak[i] = (uint8_t)now;
    if (now >= QUIET_MIN_STREAK) atomicAdd(quiet_count, 1u);  // FLUID_STAT_QUIET_BRICKS
}


// early[b] = 255 where brick b and its 26 neighbours neither held water after the previous step (`old_water`)
// nor receive a particle in this one (`particle_bricks`), else 0 (quiet_bricks.h).  The particle marks of a
// neighbouring slab are not exchanged: its edge layer counts as marked (edges.kind = EDGE_UNKNOWN there).
__global__ void k_update_early_quiet(const uint8_t* __restrict__ old_water,
                                     const uint8_t* __restrict__ particle_bricks,
                                     uint8_t* __restrict__ early, BrickK bk, BrickEdges edges) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = bk.nbx * bk.nby * bk.nbz;
    if (i >= n) return;
    const int bx = i % bk.nbx, by = (i / bk.nbx) % bk.nby, bz = i / (bk.nbx * bk.nby);
    early[i] = brick_neighbourhood(old_water, particle_bricks, bk, edges, bx, by, bz) ? 0 : 255;
}

// 01a inside fluid_run_step: the density image is non-zero only in the cells where the previous step's
// 01 counted a particle, i.e. inside the bricks that pass marked (`old_particles`).  That is not the same
// as "where the previous step had water": 02 turns a border cell with a particle into WATER, 03 then
// into SOLID, so a lone droplet in a wall cell leaves a count in a brick without a water cell.  Both maps
// are tested (the water map costs nothing and keeps the clear a superset).  cell4_grid() / cell_block().
__global__ void k_clear_density_where_particles_were(uint32_t* __restrict__ dens, GridK g,
                                                     const uint8_t* __restrict__ old_water,
                                                     const uint8_t* __restrict__ old_particles, BrickK bk) {
    const int b = brick_index(bk, (int)(blockIdx.x * 256u) / BRICK_X, (int)(blockIdx.y * 4u) / BRICK_Y,
                              ((int)blockIdx.z * g.zl) / BRICK_Z);
    if ((old_water[b] | old_particles[b]) == 0) return;
    const int x = 4 * (blockIdx.x * blockDim.x + threadIdx.x);
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= g.W || y >= g.H) return;
    FLUID_FOR_PLANES_OF_WORKGROUP()
    *reinterpret_cast<uint4*>(dens + cidx(g, x, y, lz)) = make_uint4(0u, 0u, 0u, 0u);
}

// ---- 04 + 05 ------------------------------------------------------------------------------------------
// Per-component state of 05 (extrapolate_velocities.comp:48-56) for the four cells x..x+3 of a row:
// bits [2c, 2c+1] of byte i = component c of cell i: 1 = VELOCITY_RESET, 2 = VELOCITY_EXTRAPOLATE.
__device__ __forceinline__ uint32_t activity_states4(const uint8_t* __restrict__ oldT,
                                                     const uint8_t* __restrict__ newT, const GridK& g,
                                                     const ParamsK& p, int x, int y, int lz,
                                                     int64_t id) {
    auto act = [&](uint32_t a) { return a == p.t_water || a == p.t_air; };  // :34-36
    auto word = [&](const uint8_t* t, int64_t at) { return *reinterpret_cast<const uint32_t*>(t + at); };
    const uint32_t o_c = word(oldT, id), n_c = word(newT, id);
    // -y neighbours: out of bounds reads as type 0 (device_common.h: type_at); -z: ghost planes
    // (every load unconditional, at the cell itself where the neighbour does not exist, its value dropped by a
    // select: a load behind a bounds test waits for the loads before it)
    const int64_t iy = id - (y > 0 ? g.W : 0), il = id - (x > 0 ? 1 : 0);
    const uint32_t o_y_ = word(oldT, iy), n_y_ = word(newT, iy);
    const uint32_t o_z = word(oldT, id - g.plane), n_z = word(newT, id - g.plane);
    const uint32_t o_l_ = oldT[il], n_l_ = newT[il];
    const uint32_t o_y = y > 0 ? o_y_ : 0u, n_y = y > 0 ? n_y_ : 0u;
    const uint32_t o_l = x > 0 ? o_l_ : 0u, n_l = x > 0 ? n_l_ : 0u;
    uint32_t out = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int sh = 8 * i;
        const bool was = act((o_c >> sh) & 0xFFu), is = act((n_c >> sh) & 0xFFu);  // :88-90
        const uint32_t ox = i == 0 ? o_l : (o_c >> (sh - 8 * (i > 0))) & 0xFFu;
        const uint32_t nx = i == 0 ? n_l : (n_c >> (sh - 8 * (i > 0))) & 0xFFu;
        const uint32_t onb[3] = {ox, (o_y >> sh) & 0xFFu, (o_z >> sh) & 0xFFu};
        const uint32_t nnb[3] = {nx, (n_y >> sh) & 0xFFu, (n_z >> sh) & 0xFFu};
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const bool vwas = was || act(onb[c]);  // :50
            const bool vis = is || act(nnb[c]);    // :52
            const uint32_t st = (vwas && !vis) ? 1u : ((!vwas && vis) ? 2u : 0u);
            out |= st << (sh + 2 * c);
        }
    }
    return out;
}
constexpr uint32_t STATE_ANY_EXTRAPOLATE = 0x2Au;  // bits of the three "2" states in a byte

#define FLUID_CELL4_ROW_THREAD()                                    \
    const int x = 4 * (blockIdx.x * blockDim.x + threadIdx.x);      \
    const int y = blockIdx.y * blockDim.y + threadIdx.y;            \
    if (x >= g.W || y >= g.H) return;                               \
    FLUID_FOR_PLANES_OF_WORKGROUP() {                               \
        const int64_t id = cidx(g, x, y, lz);                       \
        const int gz = g.z0 + lz;
#define FLUID_CELL4_ROW_END }

// 04 where 05 needs it: VELOCITIES_2[cell] = mean velocity of the cell's WATER neighbours
// (extrapolated_velocities.comp:37-63) for cells with a component in state VELOCITY_EXTRAPOLATE.
__global__ void k0405_extrapolate(const uint8_t* __restrict__ oldT, const uint8_t* __restrict__ newT,
                                  const float4* __restrict__ v1, float4* __restrict__ v2, GridK g,
                                  ParamsK p, const uint8_t* __restrict__ quiet, BrickK bk) {
    FLUID_LEAVE_IF_QUIET_V4(quiet, bk)
    FLUID_CELL4_ROW_THREAD();
    const uint32_t st = activity_states4(oldT, newT, g, p, x, y, lz, id);
    if ((st & (STATE_ANY_EXTRAPOLATE * 0x01010101u)) == 0u) continue;
    for (int i = 0; i < 4; i++) {
        if (((st >> (8 * i)) & STATE_ANY_EXTRAPOLATE) == 0u) continue;
        const int xi = x + i;
        int n = 0;
        float sx = 0.f, sy = 0.f, sz = 0.f;
#define FLUID_ACC(cond, nx, ny, nlz)                                         \
    if ((cond) && (uint32_t)oldT[cidx(g, nx, ny, nlz)] == p.t_water) {       \
        const float4 q = v1[cidx(g, nx, ny, nlz)];                           \
        sx = sx + q.x;                                                       \
        sy = sy + q.y;                                                       \
        sz = sz + q.z;                                                       \
        n++;                                                                 \
    }
        FLUID_ACC(xi != 0, xi - 1, y, lz)         // :46
        FLUID_ACC(y != 0, xi, y - 1, lz)          // :47
        FLUID_ACC(gz != 0, xi, y, lz - 1)         // :48
        FLUID_ACC(xi != g.W - 1, xi + 1, y, lz)   // :49
        FLUID_ACC(y != g.H - 1, xi, y + 1, lz)    // :50
        FLUID_ACC(gz != g.Dg - 1, xi, y, lz + 1)  // :51
#undef FLUID_ACC
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n != 0) {  // :53
            const float fc = (float)n;
            o.x = sx / fc;
            o.y = sy / fc;
            o.z = sz / fc;
        }
        v2[id + i] = o;  // :62
    }
    FLUID_CELL4_ROW_END
}

// 05 where it changes anything (extrapolate_velocities.comp:88-108); cells without a RESET /
// EXTRAPOLATE component keep their bits (w is 0 already, see the header comment).
__global__ void k0405_apply(const uint8_t* __restrict__ oldT, const uint8_t* __restrict__ newT,
                            const float4* __restrict__ v2, float4* __restrict__ v1, GridK g,
                            ParamsK p, const uint8_t* __restrict__ quiet, BrickK bk) {
    FLUID_LEAVE_IF_QUIET_V4(quiet, bk)
    FLUID_CELL4_ROW_THREAD();
    (void)gz;
    const uint32_t st = activity_states4(oldT, newT, g, p, x, y, lz, id);
    if (st == 0u) continue;
    for (int i = 0; i < 4; i++) {
        const uint32_t s = (st >> (8 * i)) & 0x3Fu;
        if (s == 0u) continue;
        const float4 base = v1[id + i];  // :93
        float4 ext = make_float4(0.f, 0.f, 0.f, 0.f);
        if (s & STATE_ANY_EXTRAPOLATE) ext = v2[id + i];  // :95
        float cur[3] = {base.x, base.y, base.z};
        const float ex[3] = {ext.x, ext.y, ext.z};
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const uint32_t sc = (s >> (2 * c)) & 3u;
            if (sc == 1u)
                cur[c] = 0.0f;  // VELOCITY_RESET
            else if (sc == 2u)
                cur[c] = ex[c];  // VELOCITY_EXTRAPOLATE
        }
        v1[id + i] = make_float4(cur[0], cur[1], cur[2], 0.0f);  // :108
    }
    FLUID_CELL4_ROW_END
}

// ---- 09 (as written) + 10 + 11 -------------------------------------------------------------------------
// one component through 10_solids (solids.comp:32-36 then :50-51)
__device__ __forceinline__ float solids_component(float v, bool cell_solid, bool lower_solid, float r) {
    if (cell_solid && v > -r) v = -r;
    if (lower_solid && v < r) v = r;
    return v;
}

__global__ void k091011_solids_divergence(const uint8_t* __restrict__ t,
                                          const float4* __restrict__ v2, float4* __restrict__ v1,
                                          float* __restrict__ div, GridK g, ParamsK p,
                                          const uint8_t* __restrict__ quiet, BrickK bk,
                                          int xchunks, float* __restrict__ rhs) {
    // rhs (optional): b_i of the pressure loop, ((div * rho) * dx) / dt (pressure.comp:54; k12_prepare_v4 makes
    // it from DIVERGENCES in a pass of its own otherwise: 0.22 ms of the full 512^3 tank's step)
    FLUID_LEAVE_IF_QUIET(quiet, bk, xchunks)
    FLUID_FOR_CELLS_OF_ROW(xchunks)
    const int64_t id = cidx(g, x, y, lz);
    const int gz = g.z0 + lz;
    const float r = p.repel;
    // Every load of the cell is issued before the first one is used, at an address that exists whatever the cell
    // (a neighbour outside the image: the cell itself, its value dropped by a select) — behind a bounds test each
    // load would wait for the one before it, five memory round trips in a row.
    const bool has_xm = x > 0, has_ym = y > 0, has_xp = x + 1 < g.W, has_yp = y + 1 < g.H;
    const bool has_zp = gz + 1 < g.Dg;  // on a Z slab the plane above the slab is a ghost plane of VELOCITIES_2 / types
    const int64_t ixp = id + (has_xp ? 1 : 0), iyp = id + (has_yp ? g.W : 0), izp = id + (has_zp ? g.plane : 0);
    const float4 q = v2[id];  // diffuse.comp:34,46: the copy
    const uint32_t tc = t[id], txm = t[id - (has_xm ? 1 : 0)], tym = t[id - (has_ym ? g.W : 0)], tzm = t[id - g.plane];
    const uint32_t txp = t[ixp], typ = t[iyp], tzp = t[izp];
    const float qxp = v2[ixp].x, qyp = v2[iyp].y, qzp = v2[izp].z;
    const bool solid = tc == p.t_solid;
    // (a neighbour outside the image has type 0: type_at)
    const float vx = solids_component(q.x, solid, (has_xm ? txm : 0u) == p.t_solid, r);
    const float vy = solids_component(q.y, solid, (has_ym ? tym : 0u) == p.t_solid, r);
    const float vz = solids_component(q.z, solid, tzm == p.t_solid, r);
    v1[id] = make_float4(vx, vy, vz, 1.0f);  // solids.comp:76
    // compute_divergence.comp:18-30 on the field 10 produces: the +x/+y/+z neighbours' components go
    // through 10 here as well (their lower neighbour along that axis is this cell); outside the grid the
    // image load returns 0
    const float ax = has_xp ? solids_component(qxp, txp == p.t_solid, solid, r) : 0.0f;
    const float ay = has_yp ? solids_component(qyp, typ == p.t_solid, solid, r) : 0.0f;
    const float az = has_zp ? solids_component(qzp, tzp == p.t_solid, solid, r) : 0.0f;
    float d = ax - vx;  // :21 left to right
    d = d + ay;
    d = d - vy;
    d = d + az;
    d = d - vz;
    div[id] = d;
    if (rhs) rhs[id] = ((d * p.rho) * p.dx) / p.dt;
    FLUID_END_FOR_CELLS
}

}  // namespace fluid
