// kernels_pressure.h — 12_solve_pressure: one Jacobi sweep of the pressure Poisson system
// (/root/reference/shaders_fluid/12_solve_pressure/pressure.comp:41-76).
//
// Algorithmic traffic per cell per sweep: read Pin 4 B + divergence 4 B + cell type 1 B, write
// Pout 4 B = 13 B.  ~8 flop per 13 B: HBM-bound, no MFMA.
//
// Two kernels compute the same arithmetic through jacobi_cell():
//   k12_plain   one thread per cell, neighbours straight from global memory (L1/L2 served);
//               any grid shape; used for small or odd-sized grids and as the on-GPU cross-check.
//   k12_zmarch  the roofline kernel (W % 4 == 0): a wavefront owns a 256-cell-wide row segment
//               (64 lanes x float4 = one 1-KiB coalesced access per row) of RY rows and marches
//               along z keeping the z-1 / z / z+1 planes of Pin and of the cell types in
//               registers, so every Pin value leaves HBM once per sweep; x neighbours come from
//               the adjacent lanes (wave64 lane shifts), y neighbours from the thread's own rows
//               plus one halo row above and below.
#pragma once

#include "pressure_common.h"

namespace fluid {

// pressure.comp:52-62 for one WATER cell.  Neighbour order +x,+y,+z,-x,-y,-z (:56-61).
// t*: neighbour cell types (OOB = 0), q*: neighbour Pin values (only used where the type is water).
__device__ __forceinline__ float jacobi_cell(float div, const ParamsK& p, uint32_t txp, float qxp,
                                             uint32_t typ, float qyp, uint32_t tzp, float qzp,
                                             uint32_t txm, float qxm, uint32_t tym, float qym,
                                             uint32_t tzm, float qzm) {
    int aii = 0;
    float s = ((div * p.rho) * p.dx) / p.dt;  // :54
#define FLUID_NB(t, q)                                   \
    if ((t) != p.t_solid) {                              \
        s = s - ((t) == p.t_water ? (q) : p.p_air);      \
        aii++;                                           \
    }
    FLUID_NB(txp, qxp)
    FLUID_NB(typ, qyp)
    FLUID_NB(tzp, qzp)
    FLUID_NB(txm, qxm)
    FLUID_NB(tym, qym)
    FLUID_NB(tzm, qzm)
#undef FLUID_NB
    return -s / (float)aii;  // :62
}

__global__ void k12_plain(const uint8_t* __restrict__ t, const float* __restrict__ div,
                          const float* __restrict__ pin, float* __restrict__ pout, GridK g,
                          ParamsK p) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    const int lz = blockIdx.z;
    if (x >= g.W || y >= g.H) return;
    const int64_t id = cidx(g, x, y, lz);
    if ((uint32_t)t[id] != p.t_water) return;  // :69
    const uint32_t txp = type_at(t, g, x + 1, y, lz), typ = type_at(t, g, x, y + 1, lz);
    const uint32_t tzp = t[cidx(g, x, y, lz + 1)];
    const uint32_t txm = type_at(t, g, x - 1, y, lz), tym = type_at(t, g, x, y - 1, lz);
    const uint32_t tzm = t[cidx(g, x, y, lz - 1)];
    // Pin is only read where the neighbour is water (:44-45)
    const float qxp = txp == p.t_water ? f32_at(pin, g, x + 1, y, lz) : 0.f;
    const float qyp = typ == p.t_water ? f32_at(pin, g, x, y + 1, lz) : 0.f;
    const float qzp = tzp == p.t_water ? pin[cidx(g, x, y, lz + 1)] : 0.f;
    const float qxm = txm == p.t_water ? f32_at(pin, g, x - 1, y, lz) : 0.f;
    const float qym = tym == p.t_water ? f32_at(pin, g, x, y - 1, lz) : 0.f;
    const float qzm = tzm == p.t_water ? pin[cidx(g, x, y, lz - 1)] : 0.f;
    pout[id] = jacobi_cell(div[id], p, txp, qxp, typ, qyp, tzp, qzp, txm, qxm, tym, qym, tzm, qzm);
}

// ---------------------------------------------------------------------------------------------
// z-marching kernel.
//
// Work decomposition: blockDim = 256 = 4 wavefronts stacked in y; wavefront w of block (bx, by, bz)
// owns x in [256*bx, 256*bx + 256), rows [RY*(4*by + w), +RY), planes [ZC*bz, ZC*bz + ZC).
// Per plane step the wave loads RY+2 rows of Pin (own rows of plane z+2, halo rows of plane z+1),
// RY+2 rows of types, RY rows of divergence (1 KiB / 256 B coalesced rows), plus the cells just
// outside its x range (one masked load covers all rows); everything else is register reuse.
__device__ __forceinline__ uint32_t byte_of(uint32_t w, int i) { return (w >> (8 * i)) & 0xFFu; }

template <int RY>
__global__ void __launch_bounds__(256)
k12_zmarch(const uint8_t* __restrict__ t, const float* __restrict__ div,
           const float* __restrict__ pin, float* __restrict__ pout, GridK g, ParamsK p, int zchunk) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int x0 = blockIdx.x * 256 + lane * 4;
    const int y0 = (blockIdx.y * 4 + wave) * RY;
    const int zb = blockIdx.z * zchunk;
    const int ze = min(zb + zchunk, g.Dl);
    if (y0 >= g.H) return;  // whole wave out of range (no block-level sync in this kernel)
    const bool xin = x0 < g.W;  // W % 4 == 0: a lane's 4 cells are all in or all out

    // Row r in [-1, RY]: -1 and RY are the halo rows.  A row outside [0,H) reads as zeros (OOB).
    auto row_ok = [&](int r) { return xin && (unsigned)(y0 + r) < (unsigned)g.H; };
    auto load_p = [&](int r, int lz) -> float4 {
        if (!row_ok(r)) return make_float4(0.f, 0.f, 0.f, 0.f);
        return *reinterpret_cast<const float4*>(pin + cidx(g, x0, y0 + r, lz));
    };
    auto load_t = [&](int r, int lz) -> uint32_t {
        if (!row_ok(r)) return 0u;
        return *reinterpret_cast<const uint32_t*>(t + cidx(g, x0, y0 + r, lz));
    };
    // the cell left of lane 0 / right of lane 63 of this wave's x range, rows 0..RY-1, one value
    // per lane: lane 2r -> left edge of row r, lane 2r+1 -> right edge of row r
    const int er = lane >> 1;
    const int ex = (lane & 1) ? (blockIdx.x * 256 + 256) : (blockIdx.x * 256 - 1);
    const bool e_ok = lane < 2 * RY && (unsigned)ex < (unsigned)g.W &&
                      (unsigned)(y0 + er) < (unsigned)g.H;
    auto load_ep = [&](int lz) -> float { return e_ok ? pin[cidx(g, ex, y0 + er, lz)] : 0.f; };
    auto load_et = [&](int lz) -> uint32_t {
        return e_ok ? (uint32_t)t[cidx(g, ex, y0 + er, lz)] : 0u;
    };

    // register planes (own rows): m = z-1, c = z, n = z+1, and nn = z+2 in flight (issued one
    // step ahead of its first use so HBM latency hides under a whole plane of work); halo rows
    // and x edges are needed for plane c only and are fetched one step ahead as well.
    float4 pm[RY], pc[RY], pn[RY], pnn[RY];
    uint32_t tm[RY], tc[RY], tn[RY], tnn[RY];
    float4 dc[RY], dn[RY];              // divergence of plane c / plane n
    float4 phl, phh, phl_n, phh_n;      // halo rows (row -1 / row RY) of plane c and of plane n
    uint32_t thl, thh, thl_n, thh_n;
    float ep, ep_n;                     // x-edge cells of plane c / n (per-lane encoding above)
    uint32_t et, et_n;

    auto load_d = [&](int r, int lz) -> float4 {
        if (!row_ok(r)) return make_float4(0.f, 0.f, 0.f, 0.f);
        return *reinterpret_cast<const float4*>(div + cidx(g, x0, y0 + r, lz));
    };

#pragma unroll
    for (int r = 0; r < RY; r++) {
        pm[r] = load_p(r, zb - 1);
        tm[r] = load_t(r, zb - 1);
        pc[r] = load_p(r, zb);
        tc[r] = load_t(r, zb);
        pn[r] = load_p(r, zb + 1);  // zb + 1 <= Dl: at worst the upper ghost plane
        tn[r] = load_t(r, zb + 1);
        dc[r] = load_d(r, zb);
    }
    phl = load_p(-1, zb);
    phh = load_p(RY, zb);
    thl = load_t(-1, zb);
    thh = load_t(RY, zb);
    ep = load_ep(zb);
    et = load_et(zb);

    for (int lz = zb; lz < ze; lz++) {
        // ---- issue the loads the NEXT step needs (plane lz+2 own rows; plane lz+1 halo, edges, div)
        const bool more = lz + 1 < ze;  // wave-uniform
#pragma unroll
        for (int r = 0; r < RY; r++) {
            pnn[r] = more ? load_p(r, lz + 2) : make_float4(0.f, 0.f, 0.f, 0.f);
            tnn[r] = more ? load_t(r, lz + 2) : 0u;
            dn[r] = more ? load_d(r, lz + 1) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        phl_n = more ? load_p(-1, lz + 1) : make_float4(0.f, 0.f, 0.f, 0.f);
        phh_n = more ? load_p(RY, lz + 1) : make_float4(0.f, 0.f, 0.f, 0.f);
        thl_n = more ? load_t(-1, lz + 1) : 0u;
        thh_n = more ? load_t(RY, lz + 1) : 0u;
        ep_n = more ? load_ep(lz + 1) : 0.f;
        et_n = more ? load_et(lz + 1) : 0u;

        // ---- compute plane lz
#pragma unroll
        for (int r = 0; r < RY; r++) {
            const float4 c = pc[r];
            const uint32_t tcw = tc[r];
            // x neighbours across lanes; lane 0 / 63 take the edge cells
            float left = __shfl_up(c.w, 1);
            float right = __shfl_down(c.x, 1);
            uint32_t tleft = __shfl_up(tcw, 1) >> 24;
            uint32_t tright = __shfl_down(tcw, 1) & 0xFFu;
            const float epl = __shfl(ep, 2 * r), epr = __shfl(ep, 2 * r + 1);
            const uint32_t etl = __shfl(et, 2 * r), etr = __shfl(et, 2 * r + 1);
            if (lane == 0) { left = epl; tleft = etl; }
            if (lane == 63) { right = epr; tright = etr; }
            // y neighbours
            const float4 ym = r == 0 ? phl : pc[r == 0 ? 0 : r - 1];
            const float4 yp = r == RY - 1 ? phh : pc[r == RY - 1 ? r : r + 1];
            const uint32_t tym = r == 0 ? thl : tc[r == 0 ? 0 : r - 1];
            const uint32_t typ = r == RY - 1 ? thh : tc[r == RY - 1 ? r : r + 1];
            const float4 zm = pm[r], zp = pn[r];
            const uint32_t tzm = tm[r], tzp = tn[r];

            const float cx[4] = {c.x, c.y, c.z, c.w};
            const float dvv[4] = {dc[r].x, dc[r].y, dc[r].z, dc[r].w};
            const float ymv[4] = {ym.x, ym.y, ym.z, ym.w}, ypv[4] = {yp.x, yp.y, yp.z, yp.w};
            const float zmv[4] = {zm.x, zm.y, zm.z, zm.w}, zpv[4] = {zp.x, zp.y, zp.z, zp.w};
            float o[4] = {0.f, 0.f, 0.f, 0.f};
            bool wtr[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                wtr[i] = byte_of(tcw, i) == p.t_water;
                const uint32_t txm_ = i == 0 ? tleft : byte_of(tcw, i == 0 ? 0 : i - 1);
                const uint32_t txp_ = i == 3 ? tright : byte_of(tcw, i == 3 ? 3 : i + 1);
                const float qxm_ = i == 0 ? left : cx[i == 0 ? 0 : i - 1];
                const float qxp_ = i == 3 ? right : cx[i == 3 ? 3 : i + 1];
                if (wtr[i])
                    o[i] = jacobi_cell(dvv[i], p, txp_, qxp_, byte_of(typ, i), ypv[i],
                                       byte_of(tzp, i), zpv[i], txm_, qxm_, byte_of(tym, i), ymv[i],
                                       byte_of(tzm, i), zmv[i]);
            }
            if (row_ok(r)) {
                float* dst = pout + cidx(g, x0, y0 + r, lz);
                if (wtr[0] && wtr[1] && wtr[2] && wtr[3]) {
                    *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
                } else {  // non-water cells are never written (pressure.comp:69)
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        if (wtr[i]) dst[i] = o[i];
                }
            }
        }

        // ---- rotate planes
#pragma unroll
        for (int r = 0; r < RY; r++) {
            pm[r] = pc[r];
            pc[r] = pn[r];
            pn[r] = pnn[r];
            tm[r] = tc[r];
            tc[r] = tn[r];
            tn[r] = tnn[r];
            dc[r] = dn[r];
        }
        phl = phl_n;
        phh = phh_n;
        thl = thl_n;
        thh = thh_n;
        ep = ep_n;
        et = et_n;
    }
}


// ---------------------------------------------------------------------------------------------
// The loop-section fast path: sweeps on internal working buffers.
//
// pressure.comp:41-50,69: a sweep never writes a non-water cell; a SOLID neighbour contributes
// nothing; a non-solid non-water neighbour contributes p_air instead of its stored pressure.  So a
// whole loop can run on working copies of the pressure in which every cell holds exactly what it
// contributes as a neighbour:
//     water cell: its current iterate      solid cell: +0.0f      any other cell: p_air
// (k12_import builds such a copy from PRESSURES_1; s - 0.0f == s bit for bit, so subtracting a solid
// neighbour's 0 is the shader's "skip").  A sweep then needs per cell only the byte
// {is water, number of non-solid neighbours}, the working pressure and the sweep-invariant
// b_i = ((div*rho)*dx)/dt, all of it streamed once: mask 1 + b 4 + Pin 4 + Pout 4 = the algorithmic
// 13 B/cell, no type-neighbourhood traffic, no data-dependent branches, every store a full 16-byte
// vector (non-water cells re-store their own constant).  Per water cell, in the shader's order:
//     s = b_i;  s -= W[+x]; s -= W[+y]; s -= W[+z]; s -= W[-x]; s -= W[-y]; s -= W[-z];
//     Wout = -s / aii
// Out-of-bounds neighbours contribute `p_oob` (type 0: p_air unless 0 is the solid or the water type
// value, in which case the shader's out-of-bounds arithmetic yields 0).  k12_export writes the last
// two iterates back into the water cells of PRESSURES_1 / PRESSURES_2; non-water cells of the images
// are never touched, exactly as in the reference.  Valid for any image contents and parameters.

// mask byte: number of non-SOLID neighbours if the cell is WATER, MASK_DRY otherwise
__global__ void k12_prepare(const uint8_t* __restrict__ t, const float* __restrict__ div,
                            uint8_t* __restrict__ mask, float* __restrict__ rhs,
                            uint8_t* __restrict__ active, BrickK bk, GridK g, ParamsK p,
                            int do_mask, int do_rhs) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    const int lz = blockIdx.z;
    if (x >= g.W || y >= g.H) return;
    const int64_t id = cidx(g, x, y, lz);
    if (do_rhs) rhs[id] = ((div[id] * p.rho) * p.dx) / p.dt;  // pressure.comp:54
    if (do_mask) {
        uint32_t m = 0;
        m += (type_at(t, g, x + 1, y, lz) != p.t_solid) ? 1u : 0u;
        m += (type_at(t, g, x, y + 1, lz) != p.t_solid) ? 1u : 0u;
        m += ((uint32_t)t[cidx(g, x, y, lz + 1)] != p.t_solid) ? 1u : 0u;
        m += (type_at(t, g, x - 1, y, lz) != p.t_solid) ? 1u : 0u;
        m += (type_at(t, g, x, y - 1, lz) != p.t_solid) ? 1u : 0u;
        m += ((uint32_t)t[cidx(g, x, y, lz - 1)] != p.t_solid) ? 1u : 0u;
        const bool water = (uint32_t)t[id] == p.t_water;
        mask[id] = (uint8_t)(water ? m : MASK_DRY);
        // same value from every writer: a benign race (the array was zeroed before this launch)
        if (water) active[brick_index(bk, x / BRICK_X, y / BRICK_Y, lz / BRICK_Z)] = 1;
    }
}

// Summary of the activity bricks for the host (one small workgroup): out[0] = bricks that hold water,
// out[1..2] = [lo, hi) brick range in y that holds them, out[3..4] = the same in z, out[5..6] = the x
// range of the water in cells (k12_prepare_v4's x_extent) (lo = hi = 0 when no brick holds water).  The host shapes the launches of sparse scenes with it.
__global__ void k12_count_bricks(const uint8_t* __restrict__ active, BrickK bk,
                                 uint32_t* __restrict__ out, const uint32_t* __restrict__ x_extent,
                                 int width) {
    __shared__ uint32_t sh[7];
    if (threadIdx.x < 7) sh[threadIdx.x] = (threadIdx.x & 1u) ? 0xFFFFFFFFu : 0u;  // odd slots: minima
    __syncthreads();
    const int n = bk.nbx * bk.nby * bk.nbz;
    uint32_t c = 0, ylo = 0xFFFFFFFFu, yhi = 0, zlo = 0xFFFFFFFFu, zhi = 0, xlo = 0xFFFFFFFFu, xhi = 0;
    // eight loads in flight per thread (a loop of dependent single-byte loads took 26 us for 8192 bricks)
    for (int base = 0; base < n; base += 256 * 8) {
        uint8_t a[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int i = base + k * 256 + (int)threadIdx.x;
            const uint8_t v = active[min(i, n - 1)];  // (unconditional: a guarded load waits for the one before)
            a[k] = i < n ? v : (uint8_t)0;
        }
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (a[k]) {
                const int i = base + k * 256 + (int)threadIdx.x;
                const uint32_t by = (uint32_t)((i / bk.nbx) % bk.nby), bz = (uint32_t)(i / (bk.nbx * bk.nby));
                const uint32_t bx = (uint32_t)(i % bk.nbx);
                c++;
                xlo = min(xlo, bx);
                xhi = max(xhi, bx + 1u);
                ylo = min(ylo, by);
                yhi = max(yhi, by + 1u);
                zlo = min(zlo, bz);
                zhi = max(zhi, bz + 1u);
            }
    }
    if (c) {
        atomicAdd(&sh[0], c);
        atomicMin(&sh[1], ylo);
        atomicMax(&sh[2], yhi);
        atomicMin(&sh[3], zlo);
        atomicMax(&sh[4], zhi);
        atomicMin(&sh[5], xlo);
        atomicMax(&sh[6], xhi);
    }
    __syncthreads();
    if (threadIdx.x < 7) {
        uint32_t v = sh[threadIdx.x];
        if (threadIdx.x == 5) v = (uint32_t)width - x_extent[0];  // cells, from the mask pass
        if (threadIdx.x == 6) v = x_extent[1];
        if (sh[0] == 0u) v = 0u;
        out[threadIdx.x] = v;
    }
}

// PRESSURES_1 -> working buffer (all planes incl. ghosts that hold neighbour slabs' cells)
__global__ void k12_import(const uint8_t* __restrict__ t, const float* __restrict__ pimg,
                           float* __restrict__ work, GridK g, ParamsK p, int lz0) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    const int lz = lz0 + (int)blockIdx.z;
    if (x >= g.W || y >= g.H) return;
    const int64_t id = cidx(g, x, y, lz);
    const uint32_t ty = t[id];
    work[id] = ty == p.t_water ? pimg[id] : background_value(ty, p);
}
// constant part of a working buffer: non-water cells (water cells are written by the sweeps)
__global__ void k12_background(const uint8_t* __restrict__ t, float* __restrict__ work, GridK g,
                               ParamsK p, int lz0) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    const int lz = lz0 + (int)blockIdx.z;
    if (x >= g.W || y >= g.H) return;
    const int64_t id = cidx(g, x, y, lz);
    work[id] = background_value(t[id], p);
}
// working buffers -> water cells of PRESSURES_1 (even iterate) and PRESSURES_2 (odd iterate)
__global__ void k12_export(const uint8_t* __restrict__ t, const float* __restrict__ w_even,
                           const float* __restrict__ w_odd, float* __restrict__ p1,
                           float* __restrict__ p2, GridK g, ParamsK p) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    const int lz = blockIdx.z;
    if (x >= g.W || y >= g.H) return;
    const int64_t id = cidx(g, x, y, lz);
    if ((uint32_t)t[id] != p.t_water) return;  // pressure.comp:69
    if (w_even) p1[id] = w_even[id];
    if (w_odd) p2[id] = w_odd[id];
}

template <int RY>
struct CanonAux {       // per-plane data that is only needed for the plane being computed
    float4 b[RY];       // b_i of the wave's own rows
    uint32_t m[RY];     // masks, 4 cells per word
    float4 hl, hh;      // Pin halo rows y0-1 / y0+RY
    float e;            // Pin of the cells just outside the wave's x range (per-lane encoding)
};

template <int RY>
struct CanonGeom {
    unsigned boff[RY];  // in-plane BYTE offsets of the own rows in an R32F image (32-bit: a plane
                        // is far below 4 GiB; keeps every access base(SGPR)+offset(VGPR))
    unsigned boff_lo, boff_hi, boff_e;
    bool rok[RY], lo_ok, hi_ok, e_ok, face_lo, face_hi;
    int lane;
    float p_air;  // value of an out-of-bounds neighbour (p_oob)
};


template <int RY>
__device__ __forceinline__ void canon_load_own(const CanonGeom<RY>& q, const float* pin,
                                               const GridK& g, int lz, float4* dst) {
    const float* pp = pin + (int64_t)lz * g.plane;  // wave-uniform
    const bool oob_z = (lz < 0 && q.face_lo) || (lz >= g.Dl && q.face_hi);
    const float4 pa4 = make_float4(q.p_air, q.p_air, q.p_air, q.p_air);
#pragma unroll
    for (int r = 0; r < RY; r++) {
        const float4 v = ld_f4(pp, q.boff[r]);
        dst[r] = (q.rok[r] && !oob_z) ? v : pa4;
    }
}

template <int RY>
__device__ __forceinline__ void canon_load_aux(const CanonGeom<RY>& q, const uint8_t* mask,
                                               const float* rhs, const float* pin, const GridK& g,
                                               int lz, CanonAux<RY>& a) {
    const float* pp = pin + (int64_t)lz * g.plane;
    const float* rr = rhs + (int64_t)lz * g.plane;
    const uint8_t* mm = mask + (int64_t)lz * g.plane;
    const float4 pa4 = make_float4(q.p_air, q.p_air, q.p_air, q.p_air);
#pragma unroll
    for (int r = 0; r < RY; r++) {
        a.b[r] = ld_f4(rr, q.boff[r]);
        const uint32_t m = *reinterpret_cast<const uint32_t*>(mm + (q.boff[r] >> 2));
        a.m[r] = q.rok[r] ? m : MASK_DRY4;  // outside the grid: not water, nothing computed or stored
    }
    const float4 lo = ld_f4(pp, q.boff_lo), hi = ld_f4(pp, q.boff_hi);
    a.hl = q.lo_ok ? lo : pa4;
    a.hh = q.hi_ok ? hi : pa4;
    const float ev = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(pp) + q.boff_e);
    a.e = q.e_ok ? ev : q.p_air;
}

// One plane: issue the loads of the next step (own rows of plane lz+2 into `pnn`, everything else
// of plane lz+1 into `an`), then compute plane lz from (pm, pc, pn, ac) and store it.
template <int RY>
__device__ __forceinline__ void canon_step(const CanonGeom<RY>& q, const uint8_t* mask,
                                           const float* rhs, const float* pin, float* pout,
                                           const GridK& g, int lz, int ze, const float4* pm,
                                           const float4* pc, const float4* pn, float4* pnn,
                                           const CanonAux<RY>& ac, CanonAux<RY>& an) {
    if (lz + 1 < ze) {  // wave-uniform
        canon_load_own<RY>(q, pin, g, lz + 2, pnn);
        canon_load_aux<RY>(q, mask, rhs, pin, g, lz + 1, an);
    }
    float* po = pout + (int64_t)lz * g.plane;
#pragma unroll
    for (int r = 0; r < RY; r++) {
        const float4 c = pc[r];
        const float el = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ac.e), 2 * r));
        const float er = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ac.e), 2 * r + 1));
        const float left = from_lane_below(c.w, el, q.lane);
        const float right = from_lane_above(c.x, er, q.lane);
        const float4 ym = r == 0 ? ac.hl : pc[r == 0 ? 0 : r - 1];
        const float4 yp = r == RY - 1 ? ac.hh : pc[r == RY - 1 ? r : r + 1];
        const float4 zm = pm[r], zp = pn[r];
        const float4 b = ac.b[r];
        const uint32_t m = ac.m[r];
        const bool wet = mask_any_water(m);  // any of the lane's four cells is water
        if (__builtin_amdgcn_ballot_w64(wet) == 0ull) continue;  // wave-uniform: dry row segment
        float4 o;
        o.x = canon_cell(b.x, m, 0, c.y, yp.x, zp.x, left, ym.x, zm.x);
        o.y = canon_cell(b.y, m, 1, c.z, yp.y, zp.y, c.x, ym.y, zm.y);
        o.z = canon_cell(b.z, m, 2, c.w, yp.z, zp.z, c.y, ym.z, zm.z);
        o.w = canon_cell(b.w, m, 3, right, yp.w, zp.w, c.z, ym.w, zm.w);
        // non-water cells re-store their own constant (the output buffer holds the same one)
        o.x = mask_is_water(m, 0) ? o.x : c.x;
        o.y = mask_is_water(m, 1) ? o.y : c.y;
        o.z = mask_is_water(m, 2) ? o.z : c.z;
        o.w = mask_is_water(m, 3) ? o.w : c.w;
        if (wet)  // rok[r] is implied: masks outside the grid are 0
            *reinterpret_cast<float4*>(reinterpret_cast<char*>(po) + q.boff[r]) = o;
    }
}

template <int RY>
__global__ void __launch_bounds__(256)
k12_canon(const uint8_t* __restrict__ mask, const float* __restrict__ rhs,
          const float* __restrict__ pin, float* __restrict__ pout,
          const uint8_t* __restrict__ active, BrickK bk, GridK g, float p_air, int zchunk,
          int zlo, int zhi) {
    CanonGeom<RY> q;
    q.lane = threadIdx.x & 63;
    q.p_air = p_air;
    const int wave = threadIdx.x >> 6;
    const int tx0 = blockIdx.x * 256;
    const int x0 = tx0 + q.lane * 4;
    const int y0 = (blockIdx.y * 4 + wave) * RY;
    const int zb = zlo + blockIdx.z * zchunk;  // this launch sweeps the local planes [zlo, zhi)
    const int ze = min(zb + zchunk, zhi);
    if (y0 >= g.H) return;  // wave-uniform; there is no block-level synchronisation below
    {   // nothing to do if none of the bricks this wavefront covers holds water (RY divides
        // BRICK_Y and y0 is a multiple of RY, so its rows sit in one brick row)
        uint32_t any = 0;
        for (int bz = zb / BRICK_Z; bz <= (ze - 1) / BRICK_Z; bz++)
            any |= active[brick_index(bk, blockIdx.x, y0 / BRICK_Y, bz)];
        if (any == 0) return;
    }
    const bool xin = x0 < g.W;  // W % 4 == 0: a lane's four cells are all inside or all outside

    // Out-of-range lanes/rows load from a safe in-range address and the value is replaced by p_air
    // afterwards (what an out-of-bounds neighbour contributes): no divergent control flow around the
    // loads.
    const unsigned xs = xin ? (unsigned)x0 : 0u;
#pragma unroll
    for (int r = 0; r < RY; r++) {
        q.rok[r] = xin && (y0 + r) < g.H;
        q.boff[r] = 4u * (xs + (unsigned)g.W * (unsigned)((y0 + r) < g.H ? y0 + r : y0));
    }
    q.lo_ok = xin && y0 > 0;
    q.hi_ok = xin && (y0 + RY) < g.H;
    q.boff_lo = 4u * (xs + (unsigned)g.W * (unsigned)(y0 > 0 ? y0 - 1 : y0));
    q.boff_hi = 4u * (xs + (unsigned)g.W * (unsigned)((y0 + RY) < g.H ? y0 + RY : y0));
    // cells just outside this wave's x range: lane 2r = left of row r, lane 2r+1 = right of row r;
    // lanes without an edge cell all read one wave-uniform address (a single cache-line access)
    const int er = q.lane >> 1;
    const int ex = (q.lane & 1) ? tx0 + 256 : tx0 - 1;
    q.e_ok = q.lane < 2 * RY && (unsigned)ex < (unsigned)g.W && (y0 + er) < g.H;
    q.boff_e = q.e_ok ? 4u * ((unsigned)ex + (unsigned)g.W * (unsigned)(y0 + er))
                      : (unsigned)__builtin_amdgcn_readfirstlane((int)q.boff[0]);
    // the z ghost plane at a domain face holds 0 (the general kernels' OOB value): p_air instead
    q.face_lo = g.z0 == 0;
    q.face_hi = g.z0 + g.Dl == g.Dg;

    // Four register planes of Pin rotate through the roles z-1 / z / z+1 / z+2(in flight); the
    // per-plane aux data is double-buffered.  The z loop is unrolled by 4 so the rotation is a
    // renaming, not register moves.
    float4 p0[RY], p1[RY], p2[RY], p3[RY];
    CanonAux<RY> a0, a1;
    canon_load_own<RY>(q, pin, g, zb - 1, p0);
    canon_load_own<RY>(q, pin, g, zb, p1);
    canon_load_own<RY>(q, pin, g, zb + 1, p2);  // zb + 1 <= Dl: at worst the upper ghost plane
    canon_load_aux<RY>(q, mask, rhs, pin, g, zb, a0);
    for (int lz = zb; lz < ze; lz += 4) {
        canon_step<RY>(q, mask, rhs, pin, pout, g, lz, ze, p0, p1, p2, p3, a0, a1);
        if (lz + 1 >= ze) break;
        canon_step<RY>(q, mask, rhs, pin, pout, g, lz + 1, ze, p1, p2, p3, p0, a1, a0);
        if (lz + 2 >= ze) break;
        canon_step<RY>(q, mask, rhs, pin, pout, g, lz + 2, ze, p2, p3, p0, p1, a0, a1);
        if (lz + 3 >= ze) break;
        canon_step<RY>(q, mask, rhs, pin, pout, g, lz + 3, ze, p3, p0, p1, p2, a1, a0);
    }
}

}  // namespace fluid
