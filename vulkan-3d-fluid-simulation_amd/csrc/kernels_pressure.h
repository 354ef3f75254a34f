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

#include "device_common.h"

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

}  // namespace fluid
