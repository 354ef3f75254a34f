// kernels_pressure_fused.h — two Jacobi sweeps of 12_solve_pressure per pass over HBM
// (temporal blocking of the loop section, /root/reference/fluid_flow_sections.h:300-313).
//
// The loop applies the same sweep N times; iterate j+2 of a cell depends on iterate j of the cells
// within distance 2.  k12_canon2 reads iterate j once (plus mask and b_i), forms iterate j+1 on
// chip for a region one cell larger in every direction, and writes iterate j+2: 13 B/cell of HBM
// traffic buy two sweeps instead of one.  Every value is computed by canon_cell() exactly as in the
// single-sweep kernels (same operations, same order), cells on overlapping region borders are simply
// computed twice, so the iterates are bit-identical to N separate dispatches.
//
// Work decomposition (wave64, LDS-tiled):
//   * a workgroup = 16 wavefronts = R rows x NT x-tiles (R = 16 / NT, NT = ceil(W / 256) <= 4): it spans
//     the whole x extent, so x neighbours never leave the group; it produces TY = R - 2 output rows
//     and marches along z over `zchunk` output planes;
//   * a wavefront owns one row segment of 256 cells (64 lanes x float4, 1-KiB coalesced rows) and
//     keeps the z-1 / z / z+1 planes of iterate j and of iterate j+1 of that row in registers;
//   * per plane step each wavefront publishes its row of iterate j (plane zc) and of iterate j+1
//     (plane zc-1) in LDS; y neighbours and the cells across an x-tile boundary are read from there
//     (ds_read_b128 rows / one ds_read_b32 for the two edge cells), in-row x neighbours come from
//     the adjacent lanes (DPP wave shifts); one s_barrier per plane, LDS double-buffered;
//   * rows 0 and R-1 of the group only compute iterate j+1 (halo rows); their outer y neighbour
//     row of iterate j is loaded from global memory.
// Works on the loop's internal working buffers (kernels_pressure.h) of a whole-grid context: planes
// and rows outside the grid are the constant p_oob for both iterates.
#pragma once

#include "pressure_common.h"

namespace fluid {

#define FLUID_LDS __attribute__((address_space(3)))
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 lds_ld4(const FLUID_LDS float* p) {
    const f32x4 v = *(const FLUID_LDS f32x4*)p;
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void lds_st4(FLUID_LDS float* p, float4 v) {
    const f32x4 r = {v.x, v.y, v.z, v.w};
    *(FLUID_LDS f32x4*)p = r;
}
__device__ __forceinline__ uint32_t ld_u32(const uint8_t* base, unsigned byte_off) {
    return *reinterpret_cast<const uint32_t*>(base + byte_off);
}
__device__ __forceinline__ void st_f4(float* base, unsigned byte_off, float4 v) {
    *reinterpret_cast<float4*>(reinterpret_cast<char*>(base) + byte_off) = v;
}

constexpr int FUSED_WAVES = 16;
constexpr int FUSED_THREADS = FUSED_WAVES * 64;
constexpr int FUSED_PAD = 4;  // floats of padding on each side of an LDS row (keeps rows 16-B aligned)

__host__ __device__ inline int fused_row_floats(int nt) { return nt * 256 + 2 * FUSED_PAD; }
__host__ __device__ inline size_t fused_lds_bytes(int nt) {
    const int r = FUSED_WAVES / nt;
    return (size_t)2 /*buffers*/ * 2 /*J,S*/ * r * fused_row_floats(nt) * sizeof(float);
}

template <int NT>
__global__ void __launch_bounds__(FUSED_THREADS)
k12_canon2(const uint8_t* __restrict__ mask, const float* __restrict__ rhs,
           const float* __restrict__ pin, float* __restrict__ pout, float* __restrict__ pmid,
           const uint8_t* __restrict__ active, BrickK bk, GridK g, float p_air, int zchunk) {
    constexpr int R = FUSED_WAVES / NT;   // rows of iterate j+1 per workgroup
    constexpr int TY = R - 2;             // output rows per workgroup
    constexpr int RW = NT * 256 + 2 * FUSED_PAD;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    typedef FLUID_LDS float lds_f;
    // lds layout: [buf][array J=0 / S=1][row][RW]
    auto row_ptr = [&](int buf, int arr, int row) -> lds_f* {
        return (lds_f*)lds + ((buf * 2 + arr) * R + row) * RW + FUSED_PAD;
    };

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int tx = wave % NT, rr = wave / NT;
    const int x0 = tx * 256 + lane * 4;
    const int y0 = blockIdx.y * TY;        // first output row
    const int y = y0 - 1 + rr;             // this wavefront's row
    const int zb = blockIdx.z * zchunk;
    const int ze = min(zb + zchunk, g.Dl);

    {   // the whole group leaves if no brick it touches holds water (uniform: before any barrier)
        uint32_t any = 0;
        const int by0 = max(y0 - 1, 0) / BRICK_Y, by1 = min(y0 + TY, g.H - 1) / BRICK_Y;
        const int bz0 = max(zb - 1, 0) / BRICK_Z, bz1 = min(ze, g.Dl - 1) / BRICK_Z;
        for (int bz = bz0; bz <= bz1; bz++)
            for (int by = by0; by <= by1; by++)
                for (int bx = 0; bx < bk.nbx; bx++) any |= active[brick_index(bk, bx, by, bz)];
        if (any == 0) return;
    }

    const float4 pa4 = make_float4(p_air, p_air, p_air, p_air);
    const bool xin = x0 < g.W;
    const bool row_in = xin && (unsigned)y < (unsigned)g.H;   // this lane's cells exist
    const bool is_out_row = rr >= 1 && rr <= R - 2 && row_in;
    const bool halo_lo = rr == 0, halo_hi = rr == R - 1;
    const int yh = halo_lo ? y - 1 : y + 1;                    // outer neighbour row of a halo wave
    const bool halo_in = (halo_lo || halo_hi) && xin && (unsigned)yh < (unsigned)g.H;
    // in-plane byte offsets (safe addresses for lanes / rows outside the grid)
    const unsigned xs = xin ? (unsigned)x0 : 0u;
    const unsigned boff = 4u * (xs + (unsigned)g.W * (unsigned)(((unsigned)y < (unsigned)g.H) ? y : 0));
    const unsigned boff_h =
        4u * (xs + (unsigned)g.W * (unsigned)(((unsigned)yh < (unsigned)g.H) ? yh : 0));

    // Loads are unconditional, from a valid address (plane pointer redirected to plane 0 when the
    // plane lies outside the grid), and the value is fixed up afterwards, late: a branch or a select
    // directly on a load makes hipcc wait for it (vmcnt(0)) on the spot, which would expose the HBM
    // latency in every step.
    auto plane_ok = [&](int lz) { return lz >= 0 && lz < g.Dl; };  // whole-grid context
    auto fix_j = [&](float4 v, bool ok, int lz) { return (ok && plane_ok(lz)) ? v : pa4; };
    const uint32_t lane_mask = row_in ? 0xFFFFFFFFu : 0u;  // masks outside the grid read as 0
    auto fix_m = [&](uint32_t m, int lz) { return plane_ok(lz) ? (m & lane_mask) : 0u; };

    // pad cells of every LDS row: x = -1 and x = NT*256 read as p_air (outside the grid)
    for (int i = threadIdx.x; i < 2 * 2 * R * 2 * FUSED_PAD; i += FUSED_THREADS) {
        const int side = i % (2 * FUSED_PAD), row = i / (2 * FUSED_PAD);
        lds_f* base = (lds_f*)lds + row * RW;
        base[side < FUSED_PAD ? side : RW - 2 * FUSED_PAD + side] = p_air;
    }

    // registers: iterate j at planes zc-1, zc, zc+1 (+ zc+2 in flight); iterate j+1 at zc-2, zc-1
    int zc = zb - 1;  // plane of iterate j+1 computed in the coming step
    auto pl = [&](const float* base, int lz) {  // wave-uniform plane pointer, always inside the image
        return base + (int64_t)(plane_ok(lz) ? lz : 0) * g.plane;
    };
    float4 jm = fix_j(ld_f4(pl(pin, zc - 1), boff), row_in, zc - 1);
    float4 jc = fix_j(ld_f4(pl(pin, zc), boff), row_in, zc);
    float4 jn = fix_j(ld_f4(pl(pin, zc + 1), boff), row_in, zc + 1), jnn;
    float4 hc = fix_j(ld_f4(pl(pin, zc), boff_h), halo_in, zc), hn;  // halo waves: outer y row
    float4 s_mm = pa4, s_m = pa4, s_c;
    float4 b_c = ld_f4(pl(rhs, zc), boff), b_m = make_float4(0.f, 0.f, 0.f, 0.f), b_n;
    uint32_t m_c = fix_m(ld_u32(mask + (pl(rhs, zc) - rhs), boff >> 2), zc), m_m = 0u, m_n;
    // the address of the cell across the x-tile boundary: lane 0 -> x0-1, lane 63 -> x0+4
    const int xe = lane == 0 ? x0 - 1 : x0 + 4;

    const int steps = ze - zb + 2;
    for (int k = 0; k < steps; k++, zc++) {
        const int buf = k & 1;
        // ---- loads the next step needs (raw; fixed up at the rotation below)
        jnn = ld_f4(pl(pin, zc + 2), boff);
        hn = ld_f4(pl(pin, zc + 1), boff_h);
        b_n = ld_f4(pl(rhs, zc + 1), boff);
        m_n = ld_u32(mask + (pl(rhs, zc + 1) - rhs), boff >> 2);

        // ---- publish this row: iterate j at plane zc, iterate j+1 at plane zc-1
        lds_f* jrow = row_ptr(buf, 0, rr);
        lds_f* srow = row_ptr(buf, 1, rr);
        lds_st4(jrow + x0, jc);
        lds_st4(srow + x0, s_m);
        __syncthreads();

        // ---- stage 1: iterate j+1 at plane zc for this row
        {
            const bool wet = (m_c & 0x40404040u) != 0u;
            s_c = jc;  // non-water (and out-of-grid) cells keep p_air
            if (__builtin_amdgcn_ballot_w64(wet) != 0ull) {
                const float4 ym = halo_lo ? hc : lds_ld4(row_ptr(buf, 0, halo_lo ? rr : rr - 1) + x0);
                const float4 yp = halo_hi ? hc : lds_ld4(row_ptr(buf, 0, halo_hi ? rr : rr + 1) + x0);
                const float e = jrow[xe];
                const float left = from_lane_below(jc.w, e, lane);
                const float right = from_lane_above(jc.x, e, lane);
                float4 o;
                o.x = canon_cell(b_c.x, m_c, 0, jc.y, yp.x, jn.x, left, ym.x, jm.x);
                o.y = canon_cell(b_c.y, m_c, 8, jc.z, yp.y, jn.y, jc.x, ym.y, jm.y);
                o.z = canon_cell(b_c.z, m_c, 16, jc.w, yp.z, jn.z, jc.y, ym.z, jm.z);
                o.w = canon_cell(b_c.w, m_c, 24, right, yp.w, jn.w, jc.z, ym.w, jm.w);
                s_c.x = (m_c & 0x40u) ? o.x : jc.x;
                s_c.y = (m_c & 0x4000u) ? o.y : jc.y;
                s_c.z = (m_c & 0x400000u) ? o.z : jc.z;
                s_c.w = (m_c & 0x40000000u) ? o.w : jc.w;
            }
        }

        // ---- stage 2: iterate j+2 at plane zc-1 from iterate j+1 at planes zc-2, zc-1, zc
        const int zo = zc - 1;
        if (zo >= zb && zo < ze && rr >= 1 && rr <= R - 2) {  // wave-uniform
            const bool wet = is_out_row && (m_m & 0x40404040u) != 0u;
            if (__builtin_amdgcn_ballot_w64(wet) != 0ull) {
                const float4 ym = lds_ld4(row_ptr(buf, 1, rr - 1) + x0);
                const float4 yp = lds_ld4(row_ptr(buf, 1, rr + 1) + x0);
                const float e = srow[xe];
                const float left = from_lane_below(s_m.w, e, lane);
                const float right = from_lane_above(s_m.x, e, lane);
                float4 o;
                o.x = canon_cell(b_m.x, m_m, 0, s_m.y, yp.x, s_c.x, left, ym.x, s_mm.x);
                o.y = canon_cell(b_m.y, m_m, 8, s_m.z, yp.y, s_c.y, s_m.x, ym.y, s_mm.y);
                o.z = canon_cell(b_m.z, m_m, 16, s_m.w, yp.z, s_c.z, s_m.y, ym.z, s_mm.z);
                o.w = canon_cell(b_m.w, m_m, 24, right, yp.w, s_c.w, s_m.z, ym.w, s_mm.w);
                o.x = (m_m & 0x40u) ? o.x : s_m.x;
                o.y = (m_m & 0x4000u) ? o.y : s_m.y;
                o.z = (m_m & 0x400000u) ? o.z : s_m.z;
                o.w = (m_m & 0x40000000u) ? o.w : s_m.w;
                if (wet) {
                    st_f4(pout + (int64_t)zo * g.plane, boff, o);
                    if (pmid)  // the odd iterate, kept only by the last pair of a loop
                        st_f4(pmid + (int64_t)zo * g.plane, boff, s_m);
                }
            }
        }

        // ---- rotate (and fix up what was loaded for the next step)
        jm = jc;
        jc = jn;
        jn = fix_j(jnn, row_in, zc + 2);
        hc = fix_j(hn, halo_in, zc + 1);
        s_mm = s_m;
        s_m = s_c;
        b_m = b_c;
        b_c = b_n;
        m_m = m_c;
        m_c = fix_m(m_n, zc + 1);
    }
}

}  // namespace fluid
