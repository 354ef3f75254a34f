// kernels_sampler.h — the MAC-grid velocity sampler and the three passes built on it or on the
// particle buffer: 07_advect, 14_particles, 00_init_particles, 01_update_densities.
// Citations: /root/reference.
#pragma once

#include "device_common.h"
#include "quiet_bricks.h"

// bitwise OR over the workgroup with a barrier (ROCm device library; __syncthreads_or reduces !!predicate)
extern "C" __device__ int __ockl_wgred_or_i32(int);

namespace fluid {

// One axis of `texture(velocities, (pos + move) / fluid_size)` with VK_FILTER_LINEAR and
// CLAMP_TO_EDGE (fluid_flow_sections.h:95; advect.comp:52-56).  Same fp32 definition as the oracle
// (oracle/fluid_oracle.c: axis_taps): s = coord/n; u = s*n; ub = u - 0.5; i0 = floor(ub);
// a = ub - i0; i1 = i0 + 1; indices clamped to [0, n-1].
// extent of one axis with what the division by it needs, built once per thread (make_axes)
struct AxisN {
    int n;
    float fn, inv;  // inv = 1 / n, exact when pow2
    bool pow2;
};
struct Axes {
    AxisN x, y, z;
};
__device__ __forceinline__ AxisN make_axis(int n) {
    AxisN a;
    a.n = n;
    a.fn = (float)n;
    a.pow2 = (n & (n - 1)) == 0;
    a.inv = 1.0f / a.fn;
    return a;
}
__device__ __forceinline__ Axes make_axes(const GridK& g) {
    Axes a;
    a.x = make_axis(g.W);
    a.y = make_axis(g.H);
    a.z = make_axis(g.Dg);
    return a;
}
// clamp of a tap index: v_med3_i32 (hipcc does not form it from min(max()) with a bound it cannot order)
__device__ __forceinline__ int clamp_index(int i, int last) {
    int r;
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(i), "v"(last));
    return r;
}
// product of two small integers (tile indices): v_mul_i32_i24 runs at full rate, the 32-bit v_mul_lo_u32 hipcc
// takes for an int product at a quarter of it
__device__ __forceinline__ int imul24(int a, int b) { return __mul24(a, b); }
// The lower tap before it is clamped into the image, lo in [-1, n] (the upper tap is lo + 1), and the weight.
__device__ __forceinline__ void axis_lo(float coord, const AxisN& ax, int& lo, float& a) {
    const float fn = ax.fn;
    // u = coord / n * n.  For n a power of two the quotient is coord * 2^-k, exactly, and the product gives
    // coord back — unless the quotient is subnormal and loses bits, but then |coord| < 2^-100 and ub below is
    // -0.5 with either value of u; zeros, infinities and NaN go through unchanged as well.  So u = coord: no
    // division (~11 instructions of an IEEE sequence), no multiplication; 12 samples x 3 axes per advected
    // cell make that worth a wave-uniform branch.
    float u = coord;
    if (!ax.pow2) {
        const float s = coord / fn;
        u = s * fn;
    }
    const float ub = u - 0.5f;
    float fl = floorf(ub);
    a = ub - fl;
    // fl into [-1, n]: v_med3_f32 returns the minimum of its operands when one is NaN, i.e. -1, which is what
    // "if (!(fl >= -1)) fl = -1; if (fl > fn) fl = fn" makes of a NaN too
    fl = __builtin_amdgcn_fmed3f(fl, -1.0f, fn);
    lo = (int)fl;
}
__device__ __forceinline__ void axis_taps(float coord, const AxisN& ax, int& i0, int& i1, float& a) {
    int lo;
    axis_lo(coord, ax, lo, a);
    i0 = clamp_index(lo, ax.n - 1);
    i1 = min(lo + 1, ax.n - 1);  // lo + 1 >= 0 already
}
__device__ __forceinline__ float lerp1(float A, float B, float a) { return (1.0f - a) * A + a * B; }

// Velocity tile in LDS (k07_advect_tiled): the 64 x 4 cells of a workgroup plus TILE_HALO cells around them
// in x and y, over a window of TILE_D planes that slides along z with the workgroup's march (a ring: the
// plane that enters overwrites the one that left); one array per component so that neighbouring lanes read
// neighbouring banks.  Tile cell (tx, ty, slot) holds the texel at grid index (x_org + tx, y_org + ty, local
// plane z): slot(z) = (z - z_lo + z_rot) mod TILE_D for the planes z_lo .. z_lo + TILE_D - 1 of the window;
// tile cells outside the image repeat the image's edge texel (sample_comp reads x and y taps before they are clamped).
#define FLUID_LDS_F __attribute__((address_space(3)))
constexpr int TILE_HALO = 2;
constexpr int TILE_W = 64 + 2 * TILE_HALO, TILE_H = 4 + 2 * TILE_HALO, TILE_D = 1 + 2 * TILE_HALO;
constexpr int TILE_CELLS = TILE_W * TILE_H * TILE_D;
struct NoTile {
    static constexpr bool enabled = false;
};
struct VelTile {
    static constexpr bool enabled = true;
    static constexpr int W = TILE_W, H = TILE_H;  // row length and rows per plane of comp[]
    const FLUID_LDS_F float* comp[3];
    int x_org, y_org, z_lo, z_rot;
    // ring slot of local plane z, or -1 when the plane is not in the window
    __device__ __forceinline__ int slot(int z) const {
        const int d = z - z_lo;
        if ((unsigned)d >= (unsigned)TILE_D) return -1;
        const int sl = d + z_rot;
        return sl >= TILE_D ? sl - TILE_D : sl;
    }
};

// Component COMP of the trilinear sample at world position (px,py,pz).  `v` addresses owned plane 0
// of an RGBA32F image; z taps are global indices converted to local planes (whole-grid contexts
// have z0 = 0).  The eight taps are scalar loads of one channel of the texel.
template <int COMP, typename Tile = NoTile>
__device__ __forceinline__ float sample_comp(const float4* __restrict__ v, const GridK& g,
                                             const Axes& axes, float px, float py, float pz,
                                             uint32_t* __restrict__ violation,
                                             const Tile& tile = Tile()) {
    const float mx = COMP == 0 ? 0.5f : 0.0f, my = COMP == 1 ? 0.5f : 0.0f,
                mz = COMP == 2 ? 0.5f : 0.0f;
    int xl, yl, z0, z1;  // x, y: the lower tap, not clamped yet (the tile path does not need it clamped)
    float ax, ay, az;
    axis_lo(px + mx, axes.x, xl, ax);
    axis_lo(py + my, axes.y, yl, ay);
    axis_taps(pz + mz, axes.z, z0, z1, az);
    z0 -= g.z0;
    z1 -= g.z0;
    {   // Z-slab contexts hold sg_lo / sg_hi current planes of the neighbouring slabs: a tap beyond them
        // cannot be served (never happens on a whole-grid context, whose taps are clamped into the grid
        // above)
        const int lo = -g.sg_lo, hi = g.Dl + g.sg_hi - 1;
        if (z0 < lo || z1 > hi) {
            *violation = 1u;
            z0 = min(max(z0, lo), hi);
            z1 = min(max(z1, lo), hi);
        }
    }
    float c000 = 0.f, c100 = 0.f, c010 = 0.f, c110 = 0.f, c001 = 0.f, c101 = 0.f, c011 = 0.f, c111 = 0.f;
    bool from_tile = false;
    if constexpr (Tile::enabled) {
        // The tile repeats the image's edge texels once beyond the edge (k07_advect_tiled stages it so): texels
        // xl and xl + 1 of the tile are what the clamped taps of axis_taps are in the image, and the four taps of a
        // plane sit at fixed distances from the first — one address per plane, two ds_read2_b32.
        const int tx = xl - tile.x_org, ty = yl - tile.y_org, tz0 = tile.slot(z0), tz1 = tile.slot(z1);
        constexpr int TW = Tile::W, TH = Tile::H;
        from_tile = (unsigned)tx < (unsigned)(TW - 1) && (unsigned)ty < (unsigned)(TH - 1) && tz0 >= 0 && tz1 >= 0;
        if (from_tile) {
            const int row = tx + imul24(TW, ty);
            const FLUID_LDS_F float* t0 = tile.comp[COMP] + (row + imul24(TW * TH, tz0));
            const FLUID_LDS_F float* t1 = tile.comp[COMP] + (row + imul24(TW * TH, tz1));
            c000 = t0[0];  c100 = t0[1];
            c010 = t0[TW]; c110 = t0[TW + 1];
            c001 = t1[0];  c101 = t1[1];
            c011 = t1[TW]; c111 = t1[TW + 1];
        }
    }
    if (!from_tile) {
        // one 64-bit texel address (the corner x0, y0, z0), the other seven taps at small 32-bit offsets
        // from it: the steps along the axes are 0 or 1 texel (0 where the tap is clamped at an edge)
        const int x0 = clamp_index(xl, g.W - 1), x1 = min(xl + 1, g.W - 1);
        const int y0 = clamp_index(yl, g.H - 1), y1 = min(yl + 1, g.H - 1);
        const float* __restrict__ f = reinterpret_cast<const float*>(v + cidx(g, x0, y0, z0)) + COMP;
        const int dx = 4 * (x1 - x0), dy = 4 * (y1 - y0) * g.W, dz = 4 * (z1 - z0) * (int)g.plane;
        c000 = f[0];       c100 = f[dx];
        c010 = f[dy];      c110 = f[dx + dy];
        c001 = f[dz];      c101 = f[dx + dz];
        c011 = f[dy + dz]; c111 = f[dx + dy + dz];
    }
    const float c00 = lerp1(c000, c100, ax), c10 = lerp1(c010, c110, ax);
    const float c01 = lerp1(c001, c101, ax), c11 = lerp1(c011, c111, ax);
    const float c0 = lerp1(c00, c10, ay), c1 = lerp1(c01, c11, ay);
    return lerp1(c0, c1, az);
}

// The three samples advect.comp:75 takes at a face position, without the sampler's arithmetic.  At the
// centre of the -COMP face of cell (x, y, z) the texture coordinates of the three components fall on texel
// centres or half way between two of them: component COMP has filter weight a = 0 on every axis, the other
// two have a = 1/2 along their own axis (texels p, p+1) and along COMP (texels p-1, p) and a = 0 on the third.
// That is exact only where coord / n * n returns coord — true for every coordinate when n is a power of two
// (axis_taps' pow2 path; all BASELINE grids), which the caller checks — and a lerp with a = 0,
// (1 - 0) * A + 0 * B, may be replaced by A only if A is not -0 and B is finite: the workgroup scans its
// tile for -0, denormals (0.5 * A may underflow to -0), inf and NaN while staging it and takes the general
// path if it finds one.  The a = 1/2 lerps are evaluated as written, (1 - 0.5f) * A + 0.5f * B.
__device__ __forceinline__ bool sampler_special_value(float v) {
    const uint32_t b = __float_as_uint(v), e = b & 0x7F800000u;
    return e == 0x7F800000u || (e == 0u && b != 0u);
}
template <int COMP>
__device__ __forceinline__ float3 face_velocity(const VelTile& tile, const GridK& g, int x, int y, int lz,
                                                int gz) {
    // clamped neighbour indices (CLAMP_TO_EDGE), as tile coordinates
    const int tx = x - tile.x_org, ty = y - tile.y_org, tz = tile.slot(lz);
    const int txm = tx - (x > 0), txp = tx + (x < g.W - 1);
    const int tym = ty - (y > 0), typ = ty + (y < g.H - 1);
    const int tzm = tile.slot(lz - (gz > 0)), tzp = tile.slot(lz + (gz < g.Dg - 1));
    auto at = [&](int comp, int ax, int ay, int az) {
        return tile.comp[comp][ax + imul24(TILE_W, ay + TILE_H * az)];
    };
    auto half = [](float A, float B) { return (1.0f - 0.5f) * A + 0.5f * B; };  // lerp1(A, B, 0.5f)
    float3 v;
    if (COMP == 0) {
        v.x = at(0, tx, ty, tz);
        v.y = half(half(at(1, txm, ty, tz), at(1, tx, ty, tz)), half(at(1, txm, typ, tz), at(1, tx, typ, tz)));
        v.z = half(half(at(2, txm, ty, tz), at(2, tx, ty, tz)), half(at(2, txm, ty, tzp), at(2, tx, ty, tzp)));
    } else if (COMP == 1) {
        v.x = half(half(at(0, tx, tym, tz), at(0, txp, tym, tz)), half(at(0, tx, ty, tz), at(0, txp, ty, tz)));
        v.y = at(1, tx, ty, tz);
        v.z = half(half(at(2, tx, tym, tz), at(2, tx, ty, tz)), half(at(2, tx, tym, tzp), at(2, tx, ty, tzp)));
    } else {
        v.x = half(half(at(0, tx, ty, tzm), at(0, txp, ty, tzm)), half(at(0, tx, ty, tz), at(0, txp, ty, tz)));
        v.y = half(half(at(1, tx, ty, tzm), at(1, tx, typ, tzm)), half(at(1, tx, ty, tz), at(1, tx, typ, tz)));
        v.z = at(2, tx, ty, tz);
    }
    return v;
}

template <int COMP, typename Tile = NoTile>
__device__ __forceinline__ float advect_component(const uint8_t* __restrict__ t,
                                                  const float4* __restrict__ v1, const GridK& g,
                                                  const Axes& axes, const ParamsK& p, int x, int y,
                                                  int lz, int gz,
                                                  bool cur_water, float keep,
                                                  uint32_t* __restrict__ violation,
                                                  const Tile& tile = Tile(), bool fast_faces = false,
                                                  uint32_t nt_loaded = 0xFFFFFFFFu) {
    const int pos = COMP == 0 ? x : (COMP == 1 ? y : gz);
    // advect.comp:65-68: move[c] = -1; cellAt(pos - move) is the cell at pos + e_c
    // (nt_loaded: the caller has that cell's type already, type_at's 0 for a cell outside the image included)
    const uint32_t nt = nt_loaded != 0xFFFFFFFFu
                            ? nt_loaded
                            : type_at(t, g, x + (COMP == 0), y + (COMP == 1), lz + (COMP == 2));
    if (pos != 0 && (cur_water || nt == p.t_water)) {
        const float qx = (float)x + (COMP == 0 ? 0.0f : 0.5f);  // :70-73
        const float qy = (float)y + (COMP == 1 ? 0.0f : 0.5f);
        const float qz = (float)gz + (COMP == 2 ? 0.0f : 0.5f);
        float vx = 0.f, vy = 0.f, vz = 0.f;  // :75
        bool done = false;
        if constexpr (Tile::enabled) {
            if (fast_faces) {  // workgroup-uniform
                const float3 v = face_velocity<COMP>(tile, g, x, y, lz, gz);
                vx = v.x;
                vy = v.y;
                vz = v.z;
                done = true;
            }
        }
        if (!done) {
            vx = sample_comp<0, Tile>(v1, g, axes, qx, qy, qz, violation, tile);
            vy = sample_comp<1, Tile>(v1, g, axes, qx, qy, qz, violation, tile);
            vz = sample_comp<2, Tile>(v1, g, axes, qx, qy, qz, violation, tile);
        }
        return sample_comp<COMP, Tile>(v1, g, axes, qx - vx * p.dt, qy - vy * p.dt, qz - vz * p.dt,
                                       violation, tile);  // :77
    }
    return keep;  // :79
}

// 08_forces on the value 07 produced (forces.comp:33-54), shared by both advect kernels
__device__ __forceinline__ float4 forces_on(float4 o, const uint8_t* __restrict__ t, const GridK& g,
                                            const ParamsK& p, int x, int y, int lz, int gz,
                                            bool cur_water, uint32_t below_loaded = 0xFFFFFFFFu) {
    const uint32_t t2 = below_loaded != 0xFFFFFFFFu ? below_loaded : type_at(t, g, x, y - 1, lz);
    const bool wet = cur_water || (t2 == p.t_water);
    float fy = 0.0f;
    if (y != 0 && wet) fy += p.gravity;  // :39-45
    if ((uint32_t)x == p.fountain[0] && (uint32_t)y == p.fountain[1] &&
        (uint32_t)gz == p.fountain[2] && wet)
        fy += p.fountain_force;  // :47-49
    if (fy != 0.0f) {            // :52-53
        o.x = o.x + p.dt * 0.0f;
        o.y = o.y + p.dt * fy;
        o.z = o.z + p.dt * 0.0f;
        o.w = o.w + 0.0f;
    }
    return o;
}

// 07_advect/advect.comp:84-97.  FORCES: 08_forces/forces.comp:33-54 applied to the value before it is
// stored (fluid_run_step; 08 is a pointwise update of VELOCITIES_2, see kernels_step_fused.h).
template <bool FORCES>
__global__ void k07_advect(const uint8_t* __restrict__ t, const float4* __restrict__ v1,
                           float4* __restrict__ v2, GridK g, ParamsK p,
                           uint32_t* __restrict__ violation, const uint8_t* __restrict__ quiet,
                           BrickK bk, int xchunks) {
    FLUID_LEAVE_IF_QUIET(quiet, bk, xchunks)  // quiet_bricks.h
    const Axes axes = make_axes(g);
    FLUID_FOR_CELLS_OF_ROW(xchunks)
    const int gz = g.z0 + lz;
    const int64_t id = cidx(g, x, y, lz);
    const float4 cur = v1[id];                          // :87
    const bool cur_water = (uint32_t)t[id] == p.t_water;  // :93
    float4 o;
    o.x = advect_component<0>(t, v1, g, axes, p, x, y, lz, gz, cur_water, cur.x, violation);
    o.y = advect_component<1>(t, v1, g, axes, p, x, y, lz, gz, cur_water, cur.y, violation);
    o.z = advect_component<2>(t, v1, g, axes, p, x, y, lz, gz, cur_water, cur.z, violation);
    o.w = 0.0f;
    if (FORCES) o = forces_on(o, t, g, p, x, y, lz, gz, cur_water);
    v2[id] = o;  // :96
    FLUID_END_FOR_CELLS
}

// 07_advect (+ 08_forces) with the velocity sampler tiled into LDS.  A workgroup of 64 x 4 threads marches
// over ZM planes (one brick layer) of its 64 x 4-cell column with a window of five planes of VELOCITIES_1
// (its cells + TILE_HALO cells around them) in LDS: per plane step it stages the ONE plane that enters the
// window (coalesced float4 loads along x, stored per component; 2.1 texels per thread instead of the 10.6 of
// a tile staged afresh for every plane), and the twelve trilinear samples per advected cell (three at the
// face position for the back-trace, one at the back-traced position, per component) read their 8 taps from
// the window: the face-position samples always lie inside it — and cost no sampler arithmetic where
// face_velocity applies —, a back-trace that leaves it (more than TILE_HALO cells away) falls back to global
// loads.  Planes in which no cell of the column is advected (cell or +x/+y/+z neighbour WATER,
// advect.comp:65-68) only copy.  Same arithmetic as k07_advect.
constexpr int K07_ZM = 16;  // = BRICK_Z: a workgroup stays inside one brick (quiet_bricks.h)
// Bitwise OR over the 256 threads of the workgroup with one barrier: three words of LDS in rotation (call n
// ORs into word n mod 3 and clears word n+1 mod 3, which was last read before the barrier of call n-1).
// (The device library's __ockl_wgred_or_i32 takes two barriers and 256 B of LDS.)
struct WgOr {
    uint32_t* words;
    int n;
    __device__ __forceinline__ uint32_t operator()(uint32_t v) {
        uint32_t* w = words + n % 3;
        if (threadIdx.x == 0 && threadIdx.y == 0) words[(n + 1) % 3] = 0u;
        if (v) atomicOr(w, v);
        __syncthreads();
        n++;
        return *w;
    }
};
// Four wavefronts per SIMD: left alone hipcc takes 137 VGPRs (three); held to 128 it spills 24 bytes per lane and
// the full 512^3 tank's pass drops from 4.6 to 3.9 ms; held to 96 (five, which the LDS would allow) it spills
// 150 and is back at 4.7.
template <bool FORCES>
__global__ void __launch_bounds__(256, 4)
k07_advect_tiled(const uint8_t* __restrict__ t, const float4* __restrict__ v1,
                 float4* __restrict__ v2, GridK g, ParamsK p, uint32_t* __restrict__ violation,
                 const uint8_t* __restrict__ quiet, BrickK bk, int xchunks,
                 const uint8_t* __restrict__ active) {
    static_assert(K07_ZM == BRICK_Z && TILE_D == 2 * TILE_HALO + 1, "march = one brick layer");
    if (quiet) {  // blockIdx.z counts brick layers here
        const int qb = brick_index(bk, (int)(blockIdx.x * 64u * (unsigned)xchunks) / BRICK_X,
                                   (int)(blockIdx.y * 4u) / BRICK_Y, (int)blockIdx.z);
        if ((uint32_t)quiet[qb] >= QUIET_MIN_STREAK) return;
    }
    const int y = blockIdx.y * 4 + threadIdx.y;
    const int zb = (int)blockIdx.z * K07_ZM, ze = min(zb + K07_ZM, g.Dl);
    // `active` (optional): the activity bricks of the CELL_TYPES this pass reads.  A cell is advected if it or
    // its +x / +y / +z neighbour is water (advect.comp:65-68) and takes a force if it or the cell below it is
    // water (forces.comp:39-49): with no water in this workgroup's brick nor in the bricks beyond its +x, +y,
    // +z and -y faces the pass is a copy with w = 0.  A brick layer of a neighbouring slab counts as wet.
    if (active) {
        const int bx = (int)(blockIdx.x * 64u * (unsigned)xchunks) / BRICK_X, by = (int)(blockIdx.y * 4u) / BRICK_Y,
                  bz = (int)blockIdx.z;
        uint32_t any = active[brick_index(bk, bx, by, bz)];
        if (bx + 1 < bk.nbx) any |= active[brick_index(bk, bx + 1, by, bz)];
        if (by + 1 < bk.nby) any |= active[brick_index(bk, bx, by + 1, bz)];
        if (by > 0) any |= active[brick_index(bk, bx, by - 1, bz)];
        if (bz + 1 < bk.nbz)
            any |= active[brick_index(bk, bx, by, bz + 1)];
        else
            any |= g.z0 + g.Dl < g.Dg ? 1u : 0u;
        if (any == 0) {
            if (y < g.H)
                for (int xc = 0; xc < xchunks; xc++) {
                    const int x = ((int)blockIdx.x * xchunks + xc) * 64 + (int)threadIdx.x;
                    if (x >= g.W) break;
                    for (int lz = zb; lz < ze; lz++) {
                        const int64_t id = cidx(g, x, y, lz);
                        float4 o = v1[id];
                        o.w = 0.0f;
                        v2[id] = o;
                    }
                }
            return;
        }
    }
    __shared__ float tile_mem[3][TILE_CELLS];
    __shared__ uint32_t or_words[3];
    if (threadIdx.x == 0 && threadIdx.y == 0) or_words[0] = 0u;
    __syncthreads();
    WgOr wg_or{or_words, 0};
    const Axes axes = make_axes(g);
    const bool pow2_grid = axes.x.pow2 && axes.y.pow2 && axes.z.pow2;  // coord / n * n == coord everywhere
    const int tid = threadIdx.y * 64 + threadIdx.x;
    // planes of the image that exist for this context (ghost planes of a slab; the grid itself otherwise)
    const int zlo = max(-g.sg_lo, -g.z0), zhi = min(g.Dl + g.sg_hi - 1, g.Dg - 1 - g.z0);
    for (int xc = 0; xc < xchunks; xc++) {
        const int xb = ((int)blockIdx.x * xchunks + xc) * 64;  // first cell of this chunk
        if (xb >= g.W) break;                                  // uniform
        const int x = xb + (int)threadIdx.x;
        const bool valid = x < g.W && y < g.H;
        VelTile tile;
        tile.comp[0] = (const FLUID_LDS_F float*)tile_mem[0];
        tile.comp[1] = (const FLUID_LDS_F float*)tile_mem[1];
        tile.comp[2] = (const FLUID_LDS_F float*)tile_mem[2];
        tile.x_org = xb - TILE_HALO;
        tile.y_org = (int)blockIdx.y * 4 - TILE_HALO;
        // stage plane cz of the window into ring slot `sl`; returns whether it holds a value the face
        // shortcut cannot take (per thread; the caller votes)
        auto stage = [&](int cz, int sl) -> bool {
            bool special = false;
            if (cz >= zlo && cz <= zhi) {
                // the thread's (up to) three texels, their loads in flight together: a tile cell outside the image
                // takes the nearest cell inside (what the sampler's clamp would read there: sample_comp relies on it),
                // so no load sits behind a bounds test
                constexpr int PER = (TILE_W * TILE_H + 255) / 256;
                float4 q[PER];
#pragma unroll
                for (int j = 0; j < PER; j++) {
                    const int i = min(tid + 256 * j, TILE_W * TILE_H - 1);
                    const int cx = min(max(tile.x_org + i % TILE_W, 0), g.W - 1);
                    const int cy = min(max(tile.y_org + i / TILE_W, 0), g.H - 1);
                    q[j] = v1[cidx(g, cx, cy, cz)];
                }
#pragma unroll
                for (int j = 0; j < PER; j++) {
                    const int i = tid + 256 * j;
                    if (i < TILE_W * TILE_H) {
                        const int o = i + TILE_W * TILE_H * sl;
                        tile_mem[0][o] = q[j].x;
                        tile_mem[1][o] = q[j].y;
                        tile_mem[2][o] = q[j].z;
                        special = special || sampler_special_value(q[j].x) || sampler_special_value(q[j].y) ||
                                  sampler_special_value(q[j].z);
                    }
                }
            }
            return special;
        };
        // The window is staged lazily: planes [have_lo, have_hi) are in the ring, plane z in slot
        // (z - zb + TILE_HALO) mod TILE_D.  A plane whose cells are not advected costs its vote only; a run of
        // advected planes stages one new plane per step.  (Workgroup-uniform bookkeeping.)
        auto slot_of = [&](int z) { return (z - zb + TILE_HALO) % TILE_D; };
        int have_lo = 0, have_hi = 0;
        uint32_t bad = 0;  // bit k: the plane in ring slot k holds a special value
        for (int lz = zb; lz < ze; lz++) {
            const int gz = g.z0 + lz;
            int64_t id = 0;
            float4 cur = make_float4(0.f, 0.f, 0.f, 0.f);
            bool cur_water = false, adv = false;
            uint32_t txp = 0u, typ = 0u, tzp = 0u, tym = 0u;  // types of the +x, +y, +z and -y neighbours (type_at)
            if (valid) {
                // the cell's five type bytes and its velocity, all issued before the first one is used (behind
                // type_at's bounds tests and a short-circuit || each load waited for the one before)
                id = cidx(g, x, y, lz);
                const bool has_xp = x + 1 < g.W, has_yp = y + 1 < g.H, has_ym = y > 0;
                cur = v1[id];                                   // :87
                const uint32_t tc = t[id];
                const uint32_t a = t[id + (has_xp ? 1 : 0)], b = t[id + (has_yp ? g.W : 0)], c = t[id + g.plane],
                               d = t[id - (has_ym ? g.W : 0)];
                txp = has_xp ? a : 0u;
                typ = has_yp ? b : 0u;
                tzp = c;
                tym = has_ym ? d : 0u;
                cur_water = tc == p.t_water;                    // :93
                adv = cur_water || txp == p.t_water || typ == p.t_water || tzp == p.t_water;
            }
            // the vote is also the barrier behind the previous plane's reads of the window
            const bool any_adv = wg_or(adv ? 1u : 0u) != 0u;
            float4 o = cur;
            if (any_adv) {
                const int need_lo = lz - TILE_HALO, need_hi = lz + TILE_HALO + 1;
                int from = need_lo;
                if (have_lo <= need_lo && have_hi > need_lo) from = min(have_hi, need_hi);  // extend the run
                uint32_t sp_bits = 0, staged = 0;
                for (int z = from; z < need_hi; z++) {
                    const int sl = slot_of(z);
                    staged |= 1u << sl;
                    if (stage(z, sl)) sp_bits |= 1u << sl;
                }
                have_lo = need_lo;
                have_hi = need_hi;
                // bitwise OR over the workgroup (and the barrier in front of the window's readers)
                bad = (bad & ~staged) | wg_or(sp_bits);
                tile.z_lo = need_lo;
                tile.z_rot = slot_of(need_lo);
                // the face samples touch the planes lz-1, lz, lz+1
                const uint32_t faces = (1u << slot_of(lz - 1)) | (1u << slot_of(lz)) | (1u << slot_of(lz + 1));
                const bool fast_faces = pow2_grid && (bad & faces) == 0u;
                if (adv) {
                    o.x = advect_component<0, VelTile>(t, v1, g, axes, p, x, y, lz, gz, cur_water, cur.x,
                                                       violation, tile, fast_faces, txp);
                    o.y = advect_component<1, VelTile>(t, v1, g, axes, p, x, y, lz, gz, cur_water, cur.y,
                                                       violation, tile, fast_faces, typ);
                    o.z = advect_component<2, VelTile>(t, v1, g, axes, p, x, y, lz, gz, cur_water, cur.z,
                                                       violation, tile, fast_faces, tzp);
                }
            }
            if (valid) {
                o.w = 0.0f;
                if (FORCES) o = forces_on(o, t, g, p, x, y, lz, gz, cur_water, tym);
                v2[id] = o;  // :96
            }
        }
        __syncthreads();  // the last plane's reads are done before the next chunk stages
    }
}

// How far 07_advect's back-traces can reach along z (fluid_sampler_reach): max |VELOCITIES_1.z| over the
// owned cells as a bit pattern — for non-negative floats the unsigned order of the bits is the order of
// the values, and a NaN sorts above +inf, so one atomicMax per wavefront keeps "some value is not finite"
// too.  Reduced per wavefront with wave64 shuffles.
__global__ void k_max_abs_vz(const float4* __restrict__ v1, int64_t cells, uint32_t* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    uint32_t m = 0u;
    for (; i < cells; i += stride) m = max(m, __float_as_uint(v1[i].z) & 0x7FFFFFFFu);
    for (int off = 32; off > 0; off >>= 1) m = max(m, (uint32_t)__shfl_down((int)m, off, 64));
    if ((threadIdx.x & 63u) == 0u && m != 0u) atomicMax(out, m);
}

// 14_particles/particles.comp:45-51
// owned (optional): a list of `capacity` slots to look at, of which some are OWNED_HOLE, instead of every slot
// of the buffer (Z-slab contexts store their particles compactly now and pass none)
constexpr uint32_t OWNED_HOLE = 0xFFFFFFFFu;
__global__ void k14_particles(const float4* __restrict__ v1, float4* __restrict__ particles,
                              uint64_t capacity, GridK g, ParamsK p,
                              uint32_t* __restrict__ violation, const uint32_t* __restrict__ owned) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= capacity) return;
    if (owned) {
        const uint32_t slot = owned[i];
        if (slot == OWNED_HOLE) return;
        i = slot;
    }
    float4 q = particles[i];
    if (q.w == p.active_w) {  // :48
        const Axes axes = make_axes(g);
        const float vx = sample_comp<0>(v1, g, axes, q.x, q.y, q.z, violation);
        const float vy = sample_comp<1>(v1, g, axes, q.x, q.y, q.z, violation);
        const float vz = sample_comp<2>(v1, g, axes, q.x, q.y, q.z, violation);
        q.x = q.x + vx * p.dt;  // :50
        q.y = q.y + vy * p.dt;
        q.z = q.z + vz * p.dt;
        particles[i] = q;
    }
}

// ---- particle ownership on Z-slab contexts ------------------------------------------------------
__device__ __forceinline__ bool is_tombstone(const float4& q) {
    return __float_as_uint(q.w) == PARTICLE_TOMBSTONE_BITS;
}
__device__ __forceinline__ bool slab_owns(const GridK& g, float z) {
    const int pl = particle_owner_plane(z, g.Dg) - g.z0;
    return (unsigned)pl < (unsigned)g.Dl;
}
__device__ __forceinline__ float4 tombstone() {
    return make_float4(0.f, 0.f, 0.f, __uint_as_float(PARTICLE_TOMBSTONE_BITS));
}
// A "leaver": a particle this slab holds but no longer owns (14_particles moved it across a face).
// Hand-over is between Z-neighbours: a leaver travels down (towards z = 0) or up, its new owner adopts
// it, and a slab that receives one it does not own either (the particle crossed more than one slab in a
// step) passes it on in the same direction.  Lists of 32-byte entries; dir 0 = down, 1 = up.
struct Leaver {
    float4 data;
    uint32_t index, pad0, pad1, pad2;
};
struct MigrateLists {
    Leaver* send[2];      // [down, up]
    uint32_t* count[2];   // entries appended so far
    uint32_t capacity;    // entries per list
};
__device__ __forceinline__ bool migrate_append(const MigrateLists& m, int dir, float4 data, uint32_t index) {
    const uint32_t slot = atomicAdd(m.count[dir], 1u);
    if (slot >= m.capacity) return false;  // count > capacity tells the host; the entry is not stored
    Leaver l;
    l.data = data;
    l.index = index;
    l.pad0 = l.pad1 = l.pad2 = 0u;
    m.send[dir][slot] = l;
    return true;
}
// Compact storage (Z-slab contexts): a slab stores the particles it OWNS — buf[0 .. n) with the slot (the index
// of the global particle array, which is what the API, the hand-over lists and 00_init_particles speak) of each
// one beside it — instead of a slot for every particle of the run with tombstones in the others' (8-way full
// tank at 512^3: 1.05 G slots on every rank for 132 M owned).  01, 14 and the search for leavers walk the n
// entries; a particle that leaves turns its entry into a tombstone (a hole), one that is adopted is appended;
// the host squeezes the holes out when they pile up and grows the arrays when an adoption would not fit
// (engine.hip: local_*).  The order of the entries means nothing: 01's counts are integer atomics and 14 treats
// every particle on its own.
struct CompactParticles {
    float4* buf;         // entries; null: count only
    uint32_t* pid;       // slot of each entry
    uint32_t* counters;  // [0] entries appended so far, [1] holes made so far
    uint32_t cap;        // entries the arrays can hold
};
// Appending keeps candidate order inside a block's 4096 candidates (one atomic per block; the blocks' runs land
// in any order): 01 folds the particles of 4096 consecutive entries in an LDS table before it touches the image,
// which pays when they share cells — particles of neighbouring slots do (they were spawned side by side).
constexpr int OWNED_BLOCK = 256, OWNED_PER_THREAD = 16;
template <typename Wants, typename Fetch>
__device__ __forceinline__ void compact_append_block(const CompactParticles& o, uint64_t n, Wants wants, Fetch fetch) {
    // wants(i) -> does candidate i of the block's range go into the arrays; fetch(i, data, slot) -> with what
    __shared__ uint32_t part[OWNED_PER_THREAD][OWNED_BLOCK / 64];
    __shared__ uint32_t block_base;
    const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
    const uint64_t base = (uint64_t)blockIdx.x * (OWNED_BLOCK * OWNED_PER_THREAD);
    unsigned long long mask[OWNED_PER_THREAD];
#pragma unroll
    for (int k = 0; k < OWNED_PER_THREAD; k++) {
        const uint64_t i = base + (uint64_t)k * OWNED_BLOCK + threadIdx.x;
        const bool want = i < n && wants(i);
        mask[k] = __builtin_amdgcn_ballot_w64(want);
        if (lane == 0) part[k][wave] = (uint32_t)__builtin_popcountll(mask[k]);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0u;
        for (int k = 0; k < OWNED_PER_THREAD; k++)
            for (int w = 0; w < OWNED_BLOCK / 64; w++) {
                const uint32_t c = part[k][w];
                part[k][w] = total;
                total += c;
            }
        block_base = total ? atomicAdd(o.counters, total) : 0u;
    }
    __syncthreads();
    if (!o.buf) return;  // counting pass
#pragma unroll
    for (int k = 0; k < OWNED_PER_THREAD; k++) {
        if ((mask[k] >> lane) & 1ull) {
            const uint32_t pos = block_base + part[k][wave] +
                                 (uint32_t)__builtin_popcountll(mask[k] & ((1ull << lane) - 1ull));
            if (pos < o.cap) {  // counters[0] > cap tells the host (never: it sizes the arrays first)
                const uint64_t i = base + (uint64_t)k * OWNED_BLOCK + threadIdx.x;
                float4 data;
                uint32_t slot;
                fetch(i, data, slot);
                o.buf[pos] = data;
                o.pid[pos] = slot;
            }
        }
    }
}
// squeeze the holes out: (src, src_pid, n entries) -> o
__global__ void __launch_bounds__(OWNED_BLOCK)
k_compact_squeeze(const float4* __restrict__ src, const uint32_t* __restrict__ src_pid, uint32_t n,
                  CompactParticles o) {
    compact_append_block(
        o, n, [&](uint64_t i) { return !is_tombstone(src[i]); },
        [&](uint64_t i, float4& data, uint32_t& slot) {
            data = src[i];
            slot = src_pid[i];
        });
}

// list the particles this slab holds but no longer owns and bury their entries; a particle that finds its
// list full stays where it is (the host runs another round)
__global__ void k_particles_collect_leavers(float4* __restrict__ buf, const uint32_t* __restrict__ pid, uint32_t n,
                                            GridK g, MigrateLists m, uint32_t* __restrict__ counters) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 q = buf[i];
    if (is_tombstone(q)) return;
    const int pl = particle_owner_plane(q.z, g.Dg) - g.z0;
    if ((unsigned)pl < (unsigned)g.Dl) return;
    if (migrate_append(m, pl < 0 ? 0 : 1, q, pid[i])) {
        buf[i] = tombstone();
        atomicAdd(counters + 1, 1u);
    }
}
// entries received from the neighbour below travel up (dir 1), those from above travel down (dir 0):
// adopt (append) what this slab owns, pass the rest on
__global__ void k_particles_adopt(GridK g, const Leaver* __restrict__ list, uint32_t count, int dir,
                                  MigrateLists m, CompactParticles o) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    bool adopted = false;
    Leaver l;
    l.index = 0u;
    l.data = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < count) {
        l = list[i];
        if (slab_owns(g, l.data.z))
            adopted = true;
        else
            migrate_append(m, dir, l.data, l.index);
    }
    // one atomic per wavefront, the lanes' entries in lane order
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(adopted);
    if (mask == 0ull) return;
    const int lane = (int)(threadIdx.x & 63u), leader = __builtin_ctzll(mask);
    uint32_t base = 0u;
    if (lane == leader) base = atomicAdd(o.counters, (uint32_t)__builtin_popcountll(mask));
    base = (uint32_t)__shfl((int)base, leader, 64);
    if (adopted) {
        const uint32_t pos = base + (uint32_t)__builtin_popcountll(mask & ((1ull << lane) - 1ull));
        if (pos < o.cap) {
            o.buf[pos] = l.data;
            o.pid[pos] = l.index;
        }
    }
}

// particle i of 00_init_particles/init_particles.comp:27-50
__device__ __forceinline__ float4 spawned_particle(uint64_t i, const ParamsK& p) {
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);  // :48
    if (i < (uint64_t)p.spawn_volume) {          // :39
        uint32_t n = (uint32_t)i;                // getPos :27-34
        const uint32_t cx = n % p.spawn_res[0];
        n /= p.spawn_res[0];
        const uint32_t cy = n % p.spawn_res[1];
        n /= p.spawn_res[1];
        const uint32_t cz = n % p.spawn_res[2];
        // :43  offset + 1.0 * idx / resolution * size
        o.x = p.spawn_offset[0] + ((1.0f * (float)cx) / (float)p.spawn_res[0]) * p.spawn_size[0];
        o.y = p.spawn_offset[1] + ((1.0f * (float)cy) / (float)p.spawn_res[1]) * p.spawn_size[1];
        o.z = p.spawn_offset[2] + ((1.0f * (float)cz) / (float)p.spawn_res[2]) * p.spawn_size[2];
        o.w = p.active_w;  // :45
    }
    return o;
}
// 00 on a Z slab: the particles of the run this slab owns, appended (o.buf null: counted)
__global__ void __launch_bounds__(OWNED_BLOCK)
k00_init_particles_compact(uint64_t capacity, ParamsK p, GridK g, CompactParticles o) {
    compact_append_block(
        o, capacity, [&](uint64_t i) { return slab_owns(g, spawned_particle(i, p).z); },
        [&](uint64_t i, float4& data, uint32_t& slot) {
            data = spawned_particle(i, p);
            slot = (uint32_t)i;
        });
}

// 00_init_particles/init_particles.comp:27-50
__global__ void k00_init_particles(float4* __restrict__ particles, uint64_t capacity, ParamsK p) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= capacity) return;
    particles[i] = spawned_particle(i, p);
}

// ivec3(pos.xyz) truncates toward zero; the imageAtomicAdd is dropped outside the image
// (update_densities.comp:35).  trunc(v) in [0, n-1] <=> v > -1 && v < n; NaN/inf dropped.
__device__ __forceinline__ bool trunc_index(float v, int n, int& out) {
    if (!(v > -1.0f && v < (float)n)) return false;
    out = (int)v;
    return true;
}

// 01_update_densities/update_densities.comp:29-36 — particle -> cell count scatter with LDS-binned
// atomics.  A block bins PPB consecutive particles into an LDS open-addressing table keyed by cell
// index (ds atomics), then flushes one global atomic per touched cell.  Particles keep their spawn
// order (x-fastest lattice), so the particles of one block land in a few hundred cells and the
// global atomic count drops by the particles-per-cell factor.  Counts are integers: the result is
// independent of order and identical to the per-particle atomics of the reference.
constexpr int K01_THREADS = 256;
constexpr int K01_PER_THREAD = 16;
constexpr int K01_TABLE = 4096;  // entries (32 KiB of LDS: keys + counts)
constexpr uint32_t K01_EMPTY = 0xFFFFFFFFu;

__global__ void __launch_bounds__(K01_THREADS)
k01_update_densities(const float4* __restrict__ particles, uint64_t capacity,
                     uint32_t* __restrict__ dens, GridK g, ParamsK p,
                     uint8_t* __restrict__ particle_bricks, BrickK bk, const uint32_t* __restrict__ owned) {
    // owned (Z-slab contexts, optional): `capacity` entries of the owned list instead of every slot
    // particle_bricks (optional): one byte per activity brick, set where a particle is counted
    // (quiet_bricks.h: the sections before 06 skip bricks far from old and new water)
    auto mark = [&](uint32_t key) {
        if (particle_bricks) {
            const int x = (int)(key % (uint32_t)g.W), yz = (int)(key / (uint32_t)g.W);
            particle_bricks[brick_index(bk, x / BRICK_X, (yz % g.H) / BRICK_Y, (yz / g.H) / BRICK_Z)] = 1;
        }
    };
    __shared__ uint32_t keys[K01_TABLE];
    __shared__ uint32_t counts[K01_TABLE];
    for (int i = threadIdx.x; i < K01_TABLE; i += K01_THREADS) {
        keys[i] = K01_EMPTY;
        counts[i] = 0u;
    }
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * (K01_THREADS * K01_PER_THREAD);
#pragma unroll 4
    for (int k = 0; k < K01_PER_THREAD; k++) {
        const uint64_t i = base + (uint64_t)k * K01_THREADS + threadIdx.x;  // coalesced 16-B loads
        if (i >= capacity) break;
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
        bool there = true;
        if (owned) {
            const uint32_t slot = owned[i];
            there = slot != OWNED_HOLE;
            if (there) q = particles[slot];
        } else {
            q = particles[i];
        }
        // cell this particle counts towards, or K01_EMPTY (inactive, outside the grid, another slab's)
        uint32_t key = K01_EMPTY;
        int cx, cy, cz;
        if (there && q.w == p.active_w &&  // :33
            trunc_index(q.x, g.W, cx) && trunc_index(q.y, g.H, cy) && trunc_index(q.z, g.Dg, cz)) {
            cz -= g.z0;  // slab contexts count only their own planes
            if ((unsigned)cz < (unsigned)g.Dl) key = (uint32_t)cidx(g, cx, cy, cz);
        }
        // Neighbouring lanes hold neighbouring particles, which mostly share a cell: the first lane of
        // every run of equal keys adds the whole run (wave64 shuffle + ballot), so a cell with two
        // particles in a row costs one pair of LDS atomics instead of two.
        const uint32_t prev = __shfl_up(key, 1, 64);
        const int lane = (int)(threadIdx.x & 63u);
        const bool leader = lane == 0 || key != prev;
        const unsigned long long leaders = __builtin_amdgcn_ballot_w64(leader);
        const unsigned long long live = __builtin_amdgcn_ballot_w64(true);  // lanes still in the loop
        if (!leader || key == K01_EMPTY) continue;
        const unsigned long long after = lane == 63 ? 0ull : (leaders >> (lane + 1));
        const int live_end = 64 - __builtin_clzll(live);  // lanes [0, live_end) took part
        const int run_end = after ? lane + 1 + __builtin_ctzll(after) : live_end;
        const uint32_t run = (uint32_t)(run_end - lane);
        uint32_t h = (key * 2654435761u) >> 20;  // top 12 bits -> [0, 4096)
        bool placed = false;
        for (int probe = 0; probe < 16; probe++) {
            const uint32_t old = atomicCAS(&keys[h], K01_EMPTY, key);
            if (old == K01_EMPTY || old == key) {
                atomicAdd(&counts[h], run);
                placed = true;
                break;
            }
            h = (h + 1) & (K01_TABLE - 1);
        }
        if (!placed) {  // table crowded: fall through to a global atomic
            atomicAdd(&dens[key], run);
            mark(key);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < K01_TABLE; i += K01_THREADS) {
        const uint32_t key = keys[i];
        if (key != K01_EMPTY) {
            atomicAdd(&dens[key], counts[i]);
            mark(key);
        }
    }
}

}  // namespace fluid
