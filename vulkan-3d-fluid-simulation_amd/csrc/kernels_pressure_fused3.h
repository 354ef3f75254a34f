// kernels_pressure_fused3.h — T Jacobi sweeps of 12_solve_pressure per pass over HBM, T = 3 (and 2, for
// cross-checks): the temporal blocking of kernels_pressure_fused.h one stage deeper
// (/root/reference/fluid_flow_sections.h:300-313 applies the same dispatch N times;
// shaders_fluid/12_solve_pressure/pressure.comp:41-76 is the sweep).
//
// Same decomposition as k12_canon2 — a workgroup spans the whole x extent (NT tiles of 64 lanes x float4) x R
// rows and marches along z, a wavefront owns RG adjacent rows, y neighbours across wavefronts and the cells
// across the x-tile seams travel through LDS (double-buffered, one s_barrier per plane step) — with one more
// iterate in flight:
//   ring r (r = 0 .. T-1) of a row holds iterate j+r at the planes zc-r-1, zc-r, zc-r+1 in registers;
//   stage k (k = 0 .. T-1) of plane step zc forms iterate j+k+1 at plane zc-k from ring k (z neighbours:
//   registers; y neighbours: the adjacent rows' registers or LDS; x neighbours: DPP / LDS seam cells) and
//   either feeds ring k+1 or, for k = T-1, is stored.  Row rr of the workgroup forms stage k iff
//   k <= rr <= R-1-k: rows 0 / R-1 only iterate j+1 (their outer y neighbour row of iterate j comes from global
//   memory), R - 2(T-1) rows are written.  Iterates j+1 .. j+T-1 are recomputed on the overlaps between
//   workgroups and never stored (the last launch of a loop keeps iterate j+T-1: KEEP), so the result is
//   bit-identical to T separate dispatches: every value is formed by canon_lane(), the single-sweep
//   arithmetic in the shader's order.
// 13 B/cell of HBM traffic buy T sweeps.  A launch consumes T ghost planes of iterate j (T-1 of mask / b_i)
// per side (FusedRange).
#pragma once

#include "kernels_pressure_fused.h"

namespace fluid {

template <int NT, int RG, int T>
struct FusedGeomT {
    static constexpr int WAVES = fused_waves(RG);
    static constexpr int THREADS = WAVES * 64;
    static constexpr int GROUPS = WAVES / NT;
    static constexpr int R = GROUPS * RG;        // rows of iterate j+1 per workgroup
    static constexpr int TY = R - 2 * (T - 1);   // output rows per workgroup
    static constexpr int RW = NT * 256 + 2 * FUSED_PAD;
    static_assert(GROUPS >= 2, "a wavefront is the lower or the upper edge of its workgroup, not both");
    static_assert(TY >= 1, "no output rows");
    static constexpr int ROWS = 2 /*buffers*/ * T /*iterates*/ * R + 1 /*slack row in front*/;
    static constexpr size_t lds_bytes = (size_t)ROWS * RW * sizeof(float) + 128 /* DivEntry table */;
    // LDS addressing.  A ds instruction adds a 16-bit immediate to its address register, the rows span up to
    // 150 KB, and left alone hipcc keeps one address register per far row (dozens, spilled to scratch).  So:
    // three address registers per lane, SEGF floats apart, and every row access is one of them plus an immediate.
    static constexpr int SEGF = 15360;  // 61 440 bytes
    static_assert((ROWS * RW + SEGF - 1) / SEGF <= 3, "three segments cover the rows");
    // float offset of row (rr0 + dr), dr = -1 .. RG, of array `arr` in buffer `buf`, relative to row rr0 of the
    // slack row's position
    static constexpr int row_off(int buf, int arr, int dr) { return ((buf * T + arr) * R + dr + 1) * RW; }
};

// State of one row of a wavefront.  Every ring rotates with period 4 (the z loop is unrolled by 4: ring
// indices are compile-time constants); in plane step zc with phase I:
//   it[r][I], [I+1], [I+2] = iterate j+r at planes zc-r-1, zc-r, zc-r+1  ([I+2] of r >= 1 is formed in this
//                            step by stage r-1); it[0][I+3] receives plane zc+2
//   b / m [I+2-k]          = plane zc-k (stage k); [I+3] receives plane zc+1
//   padv[I+2-r]            (windowed launches) iterate j at the column just outside the window, plane zc-r+1;
//                          [I+3] receives plane zc+2
template <int T>
struct FusedRowT {
    float4 it[T][4], b[4];
    uint32_t m[4];
    float padv[4];
    int roff;                 // element offset of the row within a plane (wave-uniform: lives in an SGPR)
    bool row_in, pad_in;      // those cells exist
    bool is_out_row;          // ... and the row is one the workgroup writes
};

template <int NT, int RG, int T>
struct FusedCtxT {
    using G = FusedGeomT<NT, RG, T>;
    const uint8_t* mask;
    const float* rhs;
    const float* pin;
    float* pout;
    float* pmid;
    FLUID_LDS float* lds;
    const FLUID_LDS char* divtab;
    uint32_t lb[3], le[3], lp[3];  // LDS byte addresses (G::SEGF apart) of this lane's cells / its seam cell /
                                   // its pad cell (windowed launches) in row rr0 of the slack row's position
    int64_t plane;
    int zb, ze;
    int jlo, jhi, mlo, mhi;  // planes that hold cells of the grid (FusedRange)
    int lane, rr0, x0, xe;
    unsigned loff, loff_pad;  // byte offset of this lane's cells / of its pad column within a row: every global
                              // access is a wave-uniform base (plane, row) plus one of these
    int roff_h;               // the row just outside the workgroup (edge wavefronts)
    bool pad_writer;
    int pad_x;
    float p_oob;
    bool halo_in;
    bool halo_lo, halo_hi;  // wave-uniform: this group holds row 0 / row R-1 of the workgroup
    bool wave_clean;

    // row rr0 + dr of (buf, arr): this lane's cells / seam cell / pad cell.  All arguments are compile-time
    // constants where this is called, so the result is an address register plus an immediate.
    static __device__ __forceinline__ FLUID_LDS float* seg(const uint32_t (&base)[3], int off) {
        const int q = off / G::SEGF;
        return (FLUID_LDS float*)(uintptr_t)base[q] + (off - q * G::SEGF);
    }
    __device__ __forceinline__ FLUID_LDS float* cells(int buf, int arr, int dr) const {
        return seg(lb, G::row_off(buf, arr, dr));
    }
    __device__ __forceinline__ FLUID_LDS float* seam(int buf, int arr, int dr) const {
        return seg(le, G::row_off(buf, arr, dr));
    }
    __device__ __forceinline__ FLUID_LDS float* padc(int buf, int arr, int dr) const {
        return seg(lp, G::row_off(buf, arr, dr));
    }
    __device__ __forceinline__ bool j_ok(int lz) const { return lz >= jlo && lz < jhi; }
    __device__ __forceinline__ bool m_ok(int lz) const { return lz >= mlo && lz < mhi; }
    __device__ __forceinline__ int64_t j_off(int lz) const { return j_ok(lz) ? (int64_t)lz * plane : (int64_t)0; }
    __device__ __forceinline__ int64_t m_off(int lz) const { return m_ok(lz) ? (int64_t)lz * plane : (int64_t)0; }
    __device__ __forceinline__ float4 fix_j(float4 v, bool ok, int lz) const {
        if (wave_clean && j_ok(lz)) return v;
        const float4 pa4 = make_float4(p_oob, p_oob, p_oob, p_oob);
        return (ok && j_ok(lz)) ? v : pa4;
    }
    __device__ __forceinline__ float fix_pad(float v, bool pad_in, int lz) const {
        return (pad_in && j_ok(lz)) ? v : p_oob;
    }
    __device__ __forceinline__ uint32_t fix_m(uint32_t m, bool row_in, int lz) const {
        if (wave_clean && m_ok(lz)) return m;
        return (m_ok(lz) && row_in) ? m : MASK_DRY4;
    }
};

// One plane step (see kernels_pressure_fused.h: fused_step for why the order is what it is).
template <int NT, int RG, int T, int I, bool WIN, bool KEEP, bool NTS>
__device__ __forceinline__ void fused_step_t(const FusedCtxT<NT, RG, T>& c, FusedRowT<T> (&row)[RG], float4 (&h)[2],
                                             int zc) {
    using G = FusedGeomT<NT, RG, T>;
    constexpr int buf = I & 1;
    constexpr int S0 = I & 3, S1 = (I + 1) & 3, S2 = (I + 2) & 3, S3 = (I + 3) & 3;
    constexpr bool X_EDGE_FROM_LDS = NT > 1 || WIN;
    const bool is_halo = c.halo_lo || c.halo_hi;  // wave-uniform

    // ---- LDS reads: the rows next to the group and the cells across the x-tile seams, per stage, issued one
    // stage ahead of their use (all of them at once held 6 + 3 more registers per row group through the step)
    float4 ext_lo[T], ext_hi[T];
    float edge[T][RG];
    auto lds_reads = [&](int r) {
        if (!c.halo_lo) ext_lo[r] = lds_ld4(c.cells(buf, r, -1));
        if (!c.halo_hi) ext_hi[r] = lds_ld4(c.cells(buf, r, RG));
#pragma unroll
        for (int i = 0; i < RG; i++) edge[r][i] = X_EDGE_FROM_LDS ? *c.seam(buf, r, i) : c.p_oob;
    };
    lds_reads(0);

    // ---- what the previous step loaded, fixed up
    const int zo = zc - (T - 1);                 // the plane stored in this step
    const bool zo_in = zo >= c.zb && zo < c.ze;  // wave-uniform
    bool wet[RG];
#pragma unroll
    for (int i = 0; i < RG; i++) {
        row[i].it[0][S2] = c.fix_j(row[i].it[0][S2], row[i].row_in, zc + 1);
        row[i].m[S2] = c.fix_m(row[i].m[S2], row[i].row_in, zc);
        wet[i] = zo_in && row[i].is_out_row && mask_any_water(row[i].m[(I + 2 - (T - 1)) & 3]);
    }
    if (is_halo) {
        const float4 hc = c.fix_j(h[I & 1], c.halo_in, zc);
        if (c.halo_lo) ext_lo[0] = hc; else ext_hi[0] = hc;
    }

    // ---- the stages
    const int64_t oo = (int64_t)(zo_in ? zo : c.zb) * c.plane;
#ifndef FT3_LOAD_AFTER
#define FT3_LOAD_AFTER 0
#endif
    constexpr int LOAD_AFTER = (T >= 3 && RG >= 2) ? FT3_LOAD_AFTER : 0;  // the stage behind which the global loads are issued
#pragma unroll
    for (int k = 0; k < T; k++) {
        const int MS = (I + 2 - k) & 3;  // b / m slot of plane zc-k
        if (k + 1 < T) lds_reads(k + 1);
        bool any = false;
        float4 v[RG];
#pragma unroll
        for (int i = 0; i < RG; i++) {
            const bool forms = c.rr0 + i >= k && c.rr0 + i <= G::R - 1 - k;  // wave-uniform
            any = any || (forms && (k < T - 1 ? mask_any_water(row[i].m[MS]) : wet[i]));
            v[i] = row[i].it[k][S1];  // cells that are not water (and rows that do not form this stage) keep theirs
        }
        if (__builtin_amdgcn_ballot_w64(any) != 0ull) {
#pragma unroll
            for (int i = 0; i < RG; i++) {
                if (c.rr0 + i < k || c.rr0 + i > G::R - 1 - k) continue;  // wave-uniform
                const DivPairs d = div_pairs(row[i].m[MS], c.divtab);
                const float4 ce = row[i].it[k][S1];
                const float4 ym = i > 0 ? row[i > 0 ? i - 1 : 0].it[k][S1] : ext_lo[k];
                const float4 yp = i < RG - 1 ? row[i < RG - 1 ? i + 1 : 0].it[k][S1] : ext_hi[k];
                const float left = from_lane_below(ce.w, edge[k][i], c.lane);
                const float right = from_lane_above(ce.x, edge[k][i], c.lane);
                v[i] = canon_lane<false, (NT >= 2)>(row[i].b[MS], row[i].m[MS], ce, yp, row[i].it[k][S2], ym,
                                                    row[i].it[k][S0], left, right, d);
#ifdef FT3_SCHED_BARRIER
                __builtin_amdgcn_sched_barrier(0);
#endif
            }
        }
#pragma unroll
        for (int i = 0; i < RG; i++) {
            if (k < T - 1) {
                row[i].it[k + 1][S2] = v[i];
            } else {
                // stores: every condition is in the lane predicate; a lane stores only if one of its four cells
                // is water
                if (wet[i]) st_f4<NTS>(c.pout + oo + row[i].roff, c.loff, v[i]);
                if (KEEP) {  // iterate j+T-1, kept only by the last launch of a loop
                    if (wet[i]) st_f4<NTS>(c.pmid + oo + row[i].roff, c.loff, row[i].it[T - 1][S1]);
                }
            }
        }
        if (k == LOAD_AFTER) {
            // ---- iterate j two planes ahead (raw; fixed up in the next step), b_i and mask (and the row outside
            // the workgroup) one plane ahead, into registers of planes that are dead from here on
            const int64_t o2 = c.j_off(zc + 2);
            const int64_t o1 = c.j_off(zc + 1), a1 = c.m_off(zc + 1);
#pragma unroll
            for (int i = 0; i < RG; i++) {
                row[i].it[0][S3] = ld_f4(c.pin + o2 + row[i].roff, c.loff);
                if (WIN)
                    row[i].padv[S3] = *reinterpret_cast<const float*>(
                        reinterpret_cast<const char*>(c.pin + o2 + row[i].roff) + c.loff_pad);
                row[i].b[S3] = ld_f4(c.rhs + a1 + row[i].roff, c.loff);
                row[i].m[S3] = ld_u32(c.mask + a1 + row[i].roff, c.loff >> 2);
            }
            if (is_halo) h[(I + 1) & 1] = ld_f4(c.pin + o1 + c.roff_h, c.loff);
        }
    }

    // ---- publish the rows for the next step: iterate j+r at plane zc+1-r
    constexpr int nbuf = buf ^ 1;
#pragma unroll
    for (int i = 0; i < RG; i++) {
        if (i == 0 || i == RG - 1 || X_EDGE_FROM_LDS) {
#pragma unroll
            for (int r = 0; r < T; r++) {
                lds_st4(c.cells(nbuf, r, i), row[i].it[r][S2]);
                if (WIN) {
                    // the columns next to the window hold non-water constants: the same value in every iterate
                    if (c.pad_writer)
                        *c.padc(nbuf, r, i) = c.fix_pad(row[i].padv[(I + 2 - r) & 3], row[i].pad_in, zc + 1 - r);
                }
            }
        }
    }
    __syncthreads();
}

template <int NT, bool WIN, int RG, int T, bool KEEP, bool NTS = false>
__global__ void __launch_bounds__(fused_waves(RG) * 64)
k12_canon_t(const uint8_t* __restrict__ mask, const float* __restrict__ rhs, const float* __restrict__ pin,
            float* __restrict__ pout, float* __restrict__ pmid, const uint8_t* __restrict__ active, BrickK bk,
            GridK g, float p_air, int zchunk, FusedRange rg) {
    using G = FusedGeomT<NT, RG, T>;
    constexpr int R = G::R, TY = G::TY, RW = G::RW;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    FusedCtxT<NT, RG, T> c;
    c.mask = mask;
    c.rhs = rhs;
    c.pin = pin;
    c.pout = pout;
    c.pmid = pmid;
    c.lds = (FLUID_LDS float*)lds;
    c.plane = g.plane;
    c.jlo = rg.jlo;
    c.jhi = rg.jhi;
    c.mlo = rg.mlo;
    c.mhi = rg.mhi;
    c.p_oob = p_air;
    c.lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int tx = wave % NT;
    c.rr0 = (wave / NT) * RG;
    c.x0 = tx * 256 + c.lane * 4;
    int tile_y = (int)blockIdx.y, tile_z = (int)blockIdx.z;
    if (rg.xcd_rows > 0) {
        const int L = (int)blockIdx.x, xcd = L & 7, slot = L >> 3;
        const int u = rg.xcd_start[xcd] + slot;
        if (u >= rg.xcd_start[xcd + 1]) return;  // padding of the shorter lists (uniform)
        tile_z = u / rg.xcd_rows;
        tile_y = u - tile_z * rg.xcd_rows;
    }
    const int y0 = (tile_y + rg.ytile0) * TY;  // first output row
    if (tile_z < rg.nz_lo) {
        c.zb = rg.zout_lo + tile_z * zchunk;
        c.ze = min(c.zb + zchunk, rg.hole_lo);
    } else {
        c.zb = rg.hole_hi + (tile_z - rg.nz_lo) * zchunk;
        c.ze = min(c.zb + zchunk, rg.zout_hi);
    }

    if (c.zb >= 0 && c.ze <= g.Dl) {
        // the whole group leaves if no brick it touches holds water (uniform: before any barrier); an output
        // cell moves only if it is water, and then its brick is active
        uint32_t any = 0;
        const int by0 = max(y0, 0) / BRICK_Y, by1 = min(y0 + TY - 1, g.H - 1) / BRICK_Y;
        const int bz0 = max(c.zb, 0) / BRICK_Z, bz1 = min(c.ze - 1, g.Dl - 1) / BRICK_Z;
        for (int bz = bz0; bz <= bz1; bz++)
            for (int by = by0; by <= by1; by++)
                for (int bx = 0; bx < bk.nbx; bx++) any |= active[brick_index(bk, bx, by, bz)];
        if (any == 0) return;
    }

    const int gx0 = (WIN ? rg.xwin0 : 0) + c.x0;  // global x of this lane's first cell
    const bool xin = gx0 < g.W;
    const unsigned xs = xin ? (unsigned)gx0 : 0u;
    c.halo_lo = c.rr0 == 0;
    c.halo_hi = c.rr0 + RG == R;
    FusedRowT<T> row[RG];
    bool dirty = false;
    const int xl = rg.xwin0 - 1, xr = rg.xwin0 + NT * 256;  // the columns next to the window (WIN)
    const int gxp = c.lane < 32 ? xl : xr;
    const bool col_in = (unsigned)gxp < (unsigned)g.W;
    const int yrow0 = y0 - (T - 1);  // row 0 of the workgroup
#pragma unroll
    for (int i = 0; i < RG; i++) {
        const int y = yrow0 + c.rr0 + i;
        const bool yin = (unsigned)y < (unsigned)g.H;
        row[i].row_in = xin && yin;
        row[i].is_out_row = c.rr0 + i >= T - 1 && c.rr0 + i <= R - T && row[i].row_in;
        row[i].roff = __builtin_amdgcn_readfirstlane(g.W * (yin ? y : 0));
        row[i].pad_in = col_in && yin;
        dirty = dirty || !row[i].row_in;
    }
    const int yh = c.halo_lo ? yrow0 - 1 : yrow0 + R;  // the row just outside the workgroup
    const bool is_halo = c.halo_lo || c.halo_hi;
    c.halo_in = is_halo && xin && (unsigned)yh < (unsigned)g.H;
    c.roff_h = __builtin_amdgcn_readfirstlane(g.W * (((unsigned)yh < (unsigned)g.H) ? yh : 0));
    c.loff = 4u * xs;
    c.loff_pad = 4u * (col_in ? (unsigned)gxp : 0u);
    c.wave_clean = __builtin_amdgcn_ballot_w64(dirty || (is_halo && !c.halo_in)) == 0ull;
    c.xe = c.lane == 0 ? c.x0 - 1 : c.x0 + 4;
    c.pad_writer = (c.lane == 0 && tx == 0) || (c.lane == 63 && tx == NT - 1);
    c.pad_x = c.lane == 0 ? -1 : NT * 256;

    {
        // the three address registers per kind; opaque to the compiler from here on, so that it keeps them
        const uint32_t row0 = (uint32_t)(uintptr_t)c.lds + 4u * (uint32_t)(FUSED_PAD + c.rr0 * RW);
#pragma unroll
        for (int q = 0; q < 3; q++) {
            c.lb[q] = row0 + 4u * (uint32_t)(c.x0 + q * G::SEGF);
            c.le[q] = row0 + 4u * (uint32_t)(c.xe + q * G::SEGF);
            c.lp[q] = row0 + 4u * (uint32_t)(c.pad_x + q * G::SEGF);
            asm volatile("" : "+v"(c.lb[q]), "+v"(c.le[q]), "+v"(c.lp[q]));
        }
    }
    {   // DivEntry table
        FLUID_LDS float* tab = c.lds + G::ROWS * RW;
        c.divtab = (const FLUID_LDS char*)tab;
        if (threadIdx.x < DIV_TABLE_ENTRIES) {
            const float a = (float)threadIdx.x;
            tab[2 * threadIdx.x] = a;
            tab[2 * threadIdx.x + 1] = threadIdx.x == 0 ? 0.0f : 1.0f / a;  // RN(1/a): IEEE division
        }
    }
    // pad cells of every LDS row: x = -1 and x = NT*256 read as p_oob (outside the grid)
    for (int i = threadIdx.x; i < G::ROWS * 2 * FUSED_PAD; i += G::THREADS) {
        const int side = i % (2 * FUSED_PAD), r = i / (2 * FUSED_PAD);
        FLUID_LDS float* base = c.lds + r * RW;
        base[side < FUSED_PAD ? side : RW - 2 * FUSED_PAD + side] = p_air;
    }

    // prologue: the state a step with ring phase 0 and zc = zb - (T-1) expects
    const float4 pa4 = make_float4(p_air, p_air, p_air, p_air);
    float4 h[2];
    int zc = c.zb - (T - 1);  // plane of iterate j+1 formed in the coming step
#pragma unroll
    for (int i = 0; i < RG; i++) {
        FusedRowT<T>& r = row[i];
        r.it[0][0] = c.fix_j(ld_f4(pin + c.j_off(zc - 1) + r.roff, c.loff), r.row_in, zc - 1);
        r.it[0][1] = c.fix_j(ld_f4(pin + c.j_off(zc) + r.roff, c.loff), r.row_in, zc);
        r.it[0][2] = ld_f4(pin + c.j_off(zc + 1) + r.roff, c.loff);  // raw: fixed up by the first step
#pragma unroll
        for (int q = 1; q < T; q++) {
            r.it[q][0] = pa4;  // planes below the first one the march forms: whatever is computed from them is
            r.it[q][1] = pa4;  // never stored and feeds nothing that is
        }
        r.b[0] = r.b[1] = make_float4(0.f, 0.f, 0.f, 0.f);
        r.m[0] = r.m[1] = MASK_DRY4;
        r.b[2] = ld_f4(rhs + c.m_off(zc) + r.roff, c.loff);
        r.m[2] = ld_u32(mask + c.m_off(zc) + r.roff, c.loff >> 2);
        if (WIN) {
            auto pad_at = [&](int lz) {
                return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(pin + c.j_off(lz) + r.roff) +
                                                       c.loff_pad);
            };
            r.padv[0] = pad_at(zc - 1);
            r.padv[1] = pad_at(zc);
            r.padv[2] = pad_at(zc + 1);
        }
    }
    h[0] = ld_f4(pin + c.j_off(zc) + c.roff_h, c.loff);
    __syncthreads();  // pad cells are in place (the rows below overwrite two of them in windowed launches)
    // what the first step expects behind its barrier: iterate j+r at plane zc-r
#pragma unroll
    for (int i = 0; i < RG; i++) {
        FusedRowT<T>& r = row[i];
#pragma unroll
        for (int q = 0; q < T; q++) {
            lds_st4(c.cells(0, q, i), r.it[q][1]);
            if (WIN) {
                // (the pad column of the planes below zc-1 is not on hand: plane zc-1's value stands in; it
                // only feeds values of the warm-up planes, which are never stored)
                if (c.pad_writer) *c.padc(0, q, i) = c.fix_pad(r.padv[q == 0 ? 1 : 0], r.pad_in, q == 0 ? zc : zc - 1);
            }
        }
    }
    __syncthreads();

    const int steps = c.ze - c.zb + 2 * (T - 1);
    for (int k = 0; k < steps; k += 4, zc += 4) {
        fused_step_t<NT, RG, T, 0, WIN, KEEP, NTS>(c, row, h, zc);
        if (k + 1 >= steps) break;  // all wave-uniform: every wavefront takes the same barriers
        fused_step_t<NT, RG, T, 1, WIN, KEEP, NTS>(c, row, h, zc + 1);
        if (k + 2 >= steps) break;
        fused_step_t<NT, RG, T, 2, WIN, KEEP, NTS>(c, row, h, zc + 2);
        if (k + 3 >= steps) break;
        fused_step_t<NT, RG, T, 3, WIN, KEEP, NTS>(c, row, h, zc + 3);
    }
}

}  // namespace fluid
