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
//
// What it took to hold three iterates of three rows per wavefront in 256 registers without scratch, and to keep
// the wavefronts' critical path short (measured with the stamps of `make trace`, tools/fused_trace3.py):
//   * LDS rows are addressed as one of three address registers plus a 16-bit immediate (FusedGeomT::SEGF);
//     left alone, hipcc kept one address register per row beyond 64 KB and spilled them;
//   * global memory goes through buffer resources: a wave-uniform base (plane, row: SGPRs) plus ONE per-lane
//     offset register for every access of the wavefront; stores are issued unconditionally, lanes that must
//     not store carry an out-of-range offset (the hardware drops them) — with a branch around a store the
//     compiler's wait for the step's loads also waited for the stores (vmcnt counts in issue order and a
//     conditionally issued operation cannot be counted);
//   * the reciprocals of the quotient come from v_rcp_f32 instead of an LDS table (div_pairs_rcp): nine
//     dependent LDS round trips per plane step were the largest item of the step;
//   * fix-ups of loaded values are selects, not wave-uniform branches with a wait each.
#pragma once

#include "kernels_pressure_fused.h"

namespace fluid {

// dev build only (make trace): cycles per phase of a plane step, as in kernels_pressure_fused.h
#ifdef FLUID_FUSED_TRACE
constexpr int FUSED3_TRACE_PHASES = 7;
__device__ unsigned long long g_fused_trace3[64 * 16 * (FUSED3_TRACE_PHASES + 1)];
#define FT3_ARG , FusedTrace& ftr
#define FT3_PASS , ftr
#else
#define FT3_ARG
#define FT3_PASS
#endif

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
    static constexpr size_t lds_bytes = (size_t)(ROWS + 1 /*slack row behind*/) * RW * sizeof(float);
    // LDS addressing.  A ds instruction adds a 16-bit immediate to its address register, the rows span up to
    // 150 KB, and left alone hipcc keeps one address register per far row (dozens, spilled to scratch).  So:
    // three address registers per lane, SEGF floats apart, and every row access is one of them plus an immediate.
    static constexpr int SEGF = 15360;  // 61 440 bytes
    static_assert(((ROWS + 1) * RW + SEGF - 1) / SEGF <= 3, "three segments cover the rows");
    // float offset of row (rr0 + dr), dr = -1 .. RG, of array `arr` in buffer `buf`, relative to row rr0 of the
    // slack row's position
    static constexpr int row_off(int buf, int arr, int dr) { return ((buf * T + arr) * R + dr + 1) * RW; }
};

// The planes a z chunk of a launch touches, for the buffer resources: [zb - T, ze + T]
constexpr uint32_t FUSED3_OOB = 0x80000000u;  // a per-lane offset beyond every resource: the access is dropped

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
    uint32_t roff4;   // byte offset of the row within a plane (wave-uniform: an SGPR)
    bool yin;         // the row exists (wave-uniform)
    bool out_row;     // ... and is one the workgroup writes (wave-uniform)
};

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

template <int NT, int RG, int T>
struct FusedCtxT {
    using G = FusedGeomT<NT, RG, T>;
    __amdgpu_buffer_rsrc_t rj, rb, rm, ro, rk;  // iterate j, b_i, mask, output, kept iterate: base = plane zbase
    uint32_t lb[3], le[3], lp[3];  // LDS byte addresses (G::SEGF apart) of this lane's cells / its seam cell /
                                   // its pad cell (windowed launches) in row rr0 of the slack row's position
    uint32_t plane_b;              // bytes per plane of a float array
    int zbase;                     // plane at offset 0 of the buffer resources
    int zb, ze;
    int jlo, jhi, mlo, mhi;  // planes that hold cells of the grid (FusedRange)
    int lane, rr0;
    uint32_t loff, loff_pad;  // byte offset of this lane's cells / of its pad column within a row: every global
                              // access is a wave-uniform offset (plane, row) plus one of these
    uint32_t roff4_h;         // the row just outside the workgroup (edge wavefronts)
    bool xin, col_in;         // per lane: its cells / its pad column lie inside the grid
    bool pad_writer;
    float p_oob;
    bool yin_h;             // wave-uniform: that outer row exists
    bool halo_lo, halo_hi;  // wave-uniform: this group holds row 0 / row R-1 of the workgroup

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
    // wave-uniform byte offset of plane lz, redirected to the resources' first plane when lz is outside the
    // grid: loads are always issued, from a valid address, and the value is replaced later (fix_*)
    __device__ __forceinline__ uint32_t j_soff(int lz) const { return j_ok(lz) ? (uint32_t)(lz - zbase) * plane_b : 0u; }
    __device__ __forceinline__ uint32_t m_soff(int lz) const { return m_ok(lz) ? (uint32_t)(lz - zbase) * plane_b : 0u; }
    __device__ __forceinline__ float4 ld4(__amdgpu_buffer_rsrc_t r, uint32_t soff) const {
        const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, loff, soff, 0);
        return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    }
    // (selects, no branch: the values were loaded a step ago, and a wave-uniform shortcut here cost a branch
    // per row with its own wait for the loads)
    __device__ __forceinline__ float4 fix_j(float4 v, bool ok, int lz) const {
        const bool keep = ok && j_ok(lz);
        return make_float4(keep ? v.x : p_oob, keep ? v.y : p_oob, keep ? v.z : p_oob, keep ? v.w : p_oob);
    }
    __device__ __forceinline__ float fix_pad(float v, bool pad_in, int lz) const {
        return (pad_in && j_ok(lz)) ? v : p_oob;
    }
    __device__ __forceinline__ uint32_t fix_m(uint32_t m, bool row_in, int lz) const {
        return (m_ok(lz) && row_in) ? m : MASK_DRY4;
    }
};

// The reciprocals of the quotient n / aii (canon_lane_pk below) come from v_cvt_f32_ubyte + v_rcp_f32, not from the LDS
// table of kernels_pressure_fused.h: a table look-up per row and stage was a dependent LDS round trip on the
// wavefront's critical path, nine per plane step.  On gfx950 v_rcp_f32 returns RN(1 / a) for a = 1, 2, 4, 5 and the
// truncated value (one ulp below) for a = 3 and 6 (tools/micro/rcp_small_int.hip; tests/test_rcp_gpu.py checks it);
// tests/divide_small_int_check.c proves the quotient chain of div_small_int exact for all |n| >= 2^-125 with either.
// The four cells of a lane when every cell of the wavefront's row segment is a water cell with six non-solid
// neighbours (mask word 0x06060606 in all lanes: the inside of a body of water — most of a full tank): aii and its
// reciprocal are constants and no cell keeps its old value, which leaves the numerators and the three-instruction
// quotients — about half the instructions of the general form.
// (Packed arithmetic like canon_lane_pk below: two cells per instruction.)
typedef float f32x2_t __attribute__((ext_vector_type(2)));
template <bool ZEROS_QUICK>
__device__ __forceinline__ float4 canon_lane_all6(float4 b, float4 c, float4 yp, float4 zp, float4 ym, float4 zm,
                                                  float left, float right) {
    const f32x2_t w0 = {left, c.x}, w1 = {c.y, c.z}, w2 = {c.w, right};
    f32x2_t s01 = {b.x, b.y}, s23 = {b.z, b.w};
    s01 = s01 - w1;                      // + x   (pressure.comp:56-61 order: +x, +y, +z, -x, -y, -z)
    s23 = s23 - w2;
    s01 = s01 - (f32x2_t){yp.x, yp.y};   // + y
    s23 = s23 - (f32x2_t){yp.z, yp.w};
    s01 = s01 - (f32x2_t){zp.x, zp.y};   // + z
    s23 = s23 - (f32x2_t){zp.z, zp.w};
    s01 = s01 - w0;                      // - x
    s23 = s23 - w1;
    s01 = s01 - (f32x2_t){ym.x, ym.y};   // - y
    s23 = s23 - (f32x2_t){ym.z, ym.w};
    s01 = s01 - (f32x2_t){zm.x, zm.y};   // - z
    s23 = s23 - (f32x2_t){zm.z, zm.w};
    const f32x2_t n01 = -s01, n23 = -s23;
    const f32x2_t a = {6.0f, 6.0f}, r = {0x1.555556p-3f, 0x1.555556p-3f};  // aii and RN(1 / 6)
    const float least = fminf(fminf(fabsf(n01.x), fabsf(n01.y)), fminf(fabsf(n23.x), fabsf(n23.y)));
    bool slow = __builtin_amdgcn_ballot_w64(least < 0x1p-100f) != 0ull;  // the tiny-numerator test of canon_div4
    if (slow && ZEROS_QUICK) {
        const int ex = min(min(__builtin_amdgcn_frexp_expf(n01.x), __builtin_amdgcn_frexp_expf(n01.y)),
                           min(__builtin_amdgcn_frexp_expf(n23.x), __builtin_amdgcn_frexp_expf(n23.y)));
        slow = __builtin_amdgcn_ballot_w64(ex < -99) != 0ull;  // only zeros were small: quick
    }
    float4 o;
    if (!slow) {  // div_small_int, two cells at a time
        const f32x2_t p01 = n01 * r, p23 = n23 * r;
        const f32x2_t e01 = __builtin_elementwise_fma(-p01, a, n01), e23 = __builtin_elementwise_fma(-p23, a, n23);
        const f32x2_t q01 = __builtin_elementwise_fma(e01, r, p01), q23 = __builtin_elementwise_fma(e23, r, p23);
        o.x = __builtin_amdgcn_div_fixupf(q01.x, 6.0f, n01.x);
        o.y = __builtin_amdgcn_div_fixupf(q01.y, 6.0f, n01.y);
        o.z = __builtin_amdgcn_div_fixupf(q23.x, 6.0f, n23.x);
        o.w = __builtin_amdgcn_div_fixupf(q23.y, 6.0f, n23.y);
    } else {  // some 0 < |n| < 2^-100 in this wavefront: the IEEE sequence for all its lanes
        o.x = n01.x / 6.0f;
        o.y = n01.y / 6.0f;
        o.z = n23.x / 6.0f;
        o.w = n23.y / 6.0f;
    }
    return o;
}

// canon_lane of kernels_pressure_fused.h on pairs of cells: v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 work on two
// floats per lane in the four cycles one v_add_f32 takes a wavefront (this is where the chip's fp32 vector peak comes
// from: 16 lanes per SIMD and clock, two floats each), each element rounded as the scalar instruction rounds it — so
// the six subtractions and the three-instruction quotient of a lane's four cells take half the instructions, bit for
// bit the same.  The x neighbours of cells (0, 1) and (2, 3) are the pairs (left, c.x), (c.y, c.z), (c.w, right).
template <bool ZEROS_QUICK>
__device__ __forceinline__ float4 canon_lane_pk(float4 b, uint32_t m, float4 c, float4 yp, float4 zp, float4 ym,
                                                float4 zm, float left, float right) {
    const f32x2_t w0 = {left, c.x}, w1 = {c.y, c.z}, w2 = {c.w, right};
    f32x2_t s01 = {b.x, b.y}, s23 = {b.z, b.w};
    s01 = s01 - w1;                      // + x   (pressure.comp:56-61 order: +x, +y, +z, -x, -y, -z)
    s23 = s23 - w2;
    s01 = s01 - (f32x2_t){yp.x, yp.y};   // + y
    s23 = s23 - (f32x2_t){yp.z, yp.w};
    s01 = s01 - (f32x2_t){zp.x, zp.y};   // + z
    s23 = s23 - (f32x2_t){zp.z, zp.w};
    s01 = s01 - w0;                      // - x
    s23 = s23 - w1;
    s01 = s01 - (f32x2_t){ym.x, ym.y};   // - y
    s23 = s23 - (f32x2_t){ym.z, ym.w};
    s01 = s01 - (f32x2_t){zm.x, zm.y};   // - z
    s23 = s23 - (f32x2_t){zm.z, zm.w};
    const f32x2_t n01 = -s01, n23 = -s23;
    const float a0 = (float)(m & 0xFFu), a1 = (float)((m >> 8) & 0xFFu), a2 = (float)((m >> 16) & 0xFFu),
                a3 = (float)(m >> 24);
    float4 o;
    auto quick = [&]() {
        const f32x2_t a01 = {a0, a1}, a23 = {a2, a3};
        const f32x2_t r01 = {__builtin_amdgcn_rcpf(a0), __builtin_amdgcn_rcpf(a1)},
                      r23 = {__builtin_amdgcn_rcpf(a2), __builtin_amdgcn_rcpf(a3)};
        const f32x2_t p01 = n01 * r01, p23 = n23 * r23;  // div_small_int, two cells at a time
        const f32x2_t e01 = __builtin_elementwise_fma(-p01, a01, n01), e23 = __builtin_elementwise_fma(-p23, a23, n23);
        const f32x2_t q01 = __builtin_elementwise_fma(e01, r01, p01), q23 = __builtin_elementwise_fma(e23, r23, p23);
        o.x = __builtin_amdgcn_div_fixupf(q01.x, a0, n01.x);
        o.y = __builtin_amdgcn_div_fixupf(q01.y, a1, n01.y);
        o.z = __builtin_amdgcn_div_fixupf(q23.x, a2, n23.x);
        o.w = __builtin_amdgcn_div_fixupf(q23.y, a3, n23.y);
    };
    // the tiny-numerator test of canon_div4 (kernels_pressure_fused.h)
    const float least = fminf(fminf(fabsf(n01.x), fabsf(n01.y)), fminf(fabsf(n23.x), fabsf(n23.y)));
    bool slow = __builtin_amdgcn_ballot_w64(least < 0x1p-100f) != 0ull;
    if (slow && ZEROS_QUICK) {
        const int ex = min(min(__builtin_amdgcn_frexp_expf(n01.x), __builtin_amdgcn_frexp_expf(n01.y)),
                           min(__builtin_amdgcn_frexp_expf(n23.x), __builtin_amdgcn_frexp_expf(n23.y)));
        slow = __builtin_amdgcn_ballot_w64(ex < -99) != 0ull;  // only zeros were small: quick
    }
    if (!slow) {
        quick();
    } else {  // some 0 < |n| < 2^-100 in this wavefront: the IEEE sequence for all its lanes
        o.x = n01.x / a0;
        o.y = n01.y / a1;
        o.z = n23.x / a2;
        o.w = n23.y / a3;
    }
    o.x = mask_is_water(m, 0) ? o.x : c.x;  // non-water (and out-of-grid) cells keep their constant
    o.y = mask_is_water(m, 1) ? o.y : c.y;
    o.z = mask_is_water(m, 2) ? o.z : c.z;
    o.w = mask_is_water(m, 3) ? o.w : c.w;
    return o;
}

#ifndef FT3_LOAD_AFTER
#define FT3_LOAD_AFTER 1
#endif

// One plane step (see kernels_pressure_fused.h: fused_step for why the order is what it is).
template <int NT, int RG, int T, int I, bool WIN, bool KEEP, bool NTS>
__device__ __forceinline__ void fused_step_t(const FusedCtxT<NT, RG, T>& c, FusedRowT<T> (&row)[RG], float4 (&h)[2],
                                             int zc FT3_ARG) {
    FT_BEGIN();
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
        // (also by the groups at the workgroup's edges, whose outer row is another array's, or a slack row:
        // they do not use what they read, and a branch around the read costs more than the read)
        ext_lo[r] = lds_ld4(c.cells(buf, r, -1));
        ext_hi[r] = lds_ld4(c.cells(buf, r, RG));
#pragma unroll
        for (int i = 0; i < RG; i++) edge[r][i] = X_EDGE_FROM_LDS ? *c.seam(buf, r, i) : c.p_oob;
    };
    lds_reads(0);
    // Publishing for the next step — iterate j+r at plane zc+1-r into the OTHER buffer, which nobody reads in
    // this step — can be spread over the step: ring r goes out as soon as its row exists (ring 0 after the
    // fix-ups, ring k+1 behind stage k) instead of all rows in front of the barrier, where the wavefronts' stores
    // queue up and the barrier waits for them: + 1.5 % at 256^3.  The two-tile kernel with three rows per
    // wavefront sits at its 256 registers, every variant of it spills a few, and what they cost tracks the spill
    // (8 bytes per lane: 6 770 iterations/s at 512^3; 28 bytes: 5 760): there the rows go out together, and the
    // water test reads the mask bits (kept by the one-tile kernels too: same code).
    constexpr bool PUBLISH_EARLY = NT == 1;
    constexpr int nbuf = buf ^ 1;
    auto publish = [&](int r) {
#pragma unroll
        for (int i = 0; i < RG; i++) {
            if (i == 0 || i == RG - 1 || X_EDGE_FROM_LDS) {  // rows inside a group: only across the x-tile seam
                lds_st4(c.cells(nbuf, r, i), row[i].it[r][S2]);
                if (WIN) {
                    // the columns next to the window hold non-water constants: the same value in every iterate
                    if (c.pad_writer)
                        *c.padc(nbuf, r, i) =
                            c.fix_pad(row[i].padv[(I + 2 - r) & 3], c.col_in && row[i].yin, zc + 1 - r);
                }
            }
        }
    };

    // ---- what the previous step loaded, fixed up
    const int zo = zc - (T - 1);                 // the plane stored in this step
    const bool zo_in = zo >= c.zb && zo < c.ze;  // wave-uniform
    bool wet[RG];
#pragma unroll
    for (int i = 0; i < RG; i++) {
        const bool row_in = c.xin && row[i].yin;
        row[i].it[0][S2] = c.fix_j(row[i].it[0][S2], row_in, zc + 1);
        row[i].m[S2] = c.fix_m(row[i].m[S2], row_in, zc);
        wet[i] = zo_in && row[i].out_row && c.xin && mask_any_water(row[i].m[(I + 2 - (T - 1)) & 3]);
    }
    {
        const float4 hc = c.fix_j(h[I & 1], c.xin && c.yin_h, zc);
        if (c.halo_lo) ext_lo[0] = hc;
        if (c.halo_hi) ext_hi[0] = hc;
    }

    if (PUBLISH_EARLY) publish(0);
    FT(0);  // LDS reads issued, wait for the previous step's loads, fix-ups
    // ---- the stages
    const uint32_t oo = (uint32_t)((zo_in ? zo : c.zb) - c.zbase) * c.plane_b;
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
        // The launches of the fat kernels (several rows per wavefront) are the dense ones: there a stage without
        // a water cell in any of the wavefront's rows is rare, and the wave-uniform test for it costs more than
        // it saves (a branch and its merge per stage: 512^3 6 820 -> 7 270 iterations/s without it; dry cells
        // keep their values through the selects either way).  The thin kernels (sparse scenes) keep it.
        constexpr bool skip_test = RG >= 2;
        if (skip_test || __builtin_amdgcn_ballot_w64(any) != 0ull) {
#pragma unroll
            for (int i = 0; i < RG; i++) {
                if (c.rr0 + i < k || c.rr0 + i > G::R - 1 - k) continue;  // wave-uniform
                const float4 ce = row[i].it[k][S1];
                const float4 ym = i > 0 ? row[i > 0 ? i - 1 : 0].it[k][S1] : ext_lo[k];
                const float4 yp = i < RG - 1 ? row[i < RG - 1 ? i + 1 : 0].it[k][S1] : ext_hi[k];
                const float left = from_lane_below(ce.w, edge[k][i], c.lane);
                const float right = from_lane_above(ce.x, edge[k][i], c.lane);
                // (thin kernels only: in the fat ones the extra wave-uniform branch per row and stage costs more than
                // the shorter arithmetic saves — 512^3 7 250 -> 6 100 iterations/s — although it fits the registers)
                if (RG == 1 && __builtin_amdgcn_ballot_w64(row[i].m[MS] != 0x06060606u) == 0ull) {  // wave-uniform
                    v[i] = canon_lane_all6<(NT >= 2)>(row[i].b[MS], ce, yp, row[i].it[k][S2], ym, row[i].it[k][S0],
                                                      left, right);
                    continue;
                }
                v[i] = canon_lane_pk<(NT >= 2)>(row[i].b[MS], row[i].m[MS], ce, yp, row[i].it[k][S2], ym,
                                                row[i].it[k][S0], left, right);
            }
        }
#pragma unroll
        for (int i = 0; i < RG; i++) {
            if (k < T - 1) {
                row[i].it[k + 1][S2] = v[i];
                if (PUBLISH_EARLY && i == RG - 1) publish(k + 1);
            } else {
                // stores: issued by every wavefront in every step (an exact count for the waits on the loads
                // around them); a lane stores only if one of its four cells is water, the others — and every
                // lane of a step outside the chunk, of a halo row — carry an offset the hardware drops.
                // The plane / row offset rides in the per-lane offset, NOT in the instruction's SGPR offset:
                // a buffer_store_dwordx4 must not be followed at once by a VALU write to its data registers, hipcc
                // inserts the wait state only for the forms without an SGPR offset (the ISA manual exempts the
                // others), and on gfx950 the SGPR form needs it too — the v_cndmask of the NEXT row's offset
                // landed in the first data register of the store just issued, and about one launch in eight
                // came back with that offset in lanes 12-15 of each row of 16 of ONE float4 (plane 1, row 7 of
                // the workgroup: tools/dbg_single.py).
                const uint32_t vo = wet[i] ? c.loff + (oo + row[i].roff4) : FUSED3_OOB;
                const u32x4_t bits = {__float_as_uint(v[i].x), __float_as_uint(v[i].y), __float_as_uint(v[i].z),
                                      __float_as_uint(v[i].w)};
                __builtin_amdgcn_raw_buffer_store_b128(bits, c.ro, vo, 0, NTS ? 2 : 0);
                if (KEEP) {  // iterate j+T-1, kept only by the last launch of a loop
                    const float4 km = row[i].it[T - 1][S1];
                    const u32x4_t kb = {__float_as_uint(km.x), __float_as_uint(km.y), __float_as_uint(km.z),
                                        __float_as_uint(km.w)};
                    __builtin_amdgcn_raw_buffer_store_b128(kb, c.rk, vo, 0, NTS ? 2 : 0);
                }
            }
        }
#ifdef FLUID_FUSED_TRACE
        asm volatile("" ::"v"(v[0].x), "v"(v[RG - 1].w));  // the stage is done here
        if (k == 0) FT(1);
        if (k == 1) FT(2);
        if (k == 2) FT(3);
#endif
        if (k == LOAD_AFTER) {
            // ---- iterate j two planes ahead (raw; fixed up in the next step), b_i and mask (and the row outside
            // the workgroup) one plane ahead, into registers of planes that are dead from here on
            const uint32_t o2 = c.j_soff(zc + 2), o1 = c.j_soff(zc + 1), a1 = c.m_soff(zc + 1);
#pragma unroll
            for (int i = 0; i < RG; i++) {
                row[i].it[0][S3] = c.ld4(c.rj, o2 + row[i].roff4);
                if (WIN)
                    row[i].padv[S3] = __uint_as_float(
                        __builtin_amdgcn_raw_buffer_load_b32(c.rj, c.loff_pad, o2 + row[i].roff4, 0));
                row[i].b[S3] = c.ld4(c.rb, a1 + row[i].roff4);
                row[i].m[S3] = __builtin_amdgcn_raw_buffer_load_b32(c.rm, c.loff >> 2, (a1 + row[i].roff4) >> 2, 0);
            }
            // (issued by every wavefront, so that the count of loads in flight is the same on every path; only the
            // groups at the workgroup's edges fetch anything)
            {
                const u32x4_t hv = __builtin_amdgcn_raw_buffer_load_b128(c.rj, is_halo ? c.loff : FUSED3_OOB,
                                                                         o1 + c.roff4_h, 0);
                h[(I + 1) & 1] = make_float4(__uint_as_float(hv.x), __uint_as_float(hv.y), __uint_as_float(hv.z),
                                             __uint_as_float(hv.w));
            }
        }
    }

    if (!PUBLISH_EARLY) {
#pragma unroll
        for (int r = 0; r < T; r++) publish(r);
    }
    FT(4);  // publish
    __syncthreads();
    FT(5);  // barrier
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
    c.jlo = rg.jlo;
    c.jhi = rg.jhi;
    c.mlo = rg.mlo;
    c.mhi = rg.mhi;
    c.p_oob = p_air;
    c.lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int tx = wave % NT;
    c.rr0 = (wave / NT) * RG;
    const int x0 = tx * 256 + c.lane * 4;
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
        // the whole group leaves if no brick it writes into holds water (uniform: before any barrier): an output
        // cell moves only if it is water, and then its brick is active; groups that write ghost planes always
        // run (the activity map covers owned planes only)
        uint32_t any = 0;
        const int by0 = max(y0, 0) / BRICK_Y, by1 = min(y0 + TY - 1, g.H - 1) / BRICK_Y;
        const int bz0 = max(c.zb, 0) / BRICK_Z, bz1 = min(c.ze - 1, g.Dl - 1) / BRICK_Z;
        for (int bz = bz0; bz <= bz1; bz++)
            for (int by = by0; by <= by1; by++)
                for (int bx = 0; bx < bk.nbx; bx++) any |= active[brick_index(bk, bx, by, bz)];
        if (any == 0) return;
    }

    // buffer resources over the planes [zb - T, ze + T] of this chunk (the working buffers carry LOOP_GHOST >= T
    // ghost planes per side whatever the context, so the planes exist; the launch code checks that the range
    // stays below 2 GB)
    c.zbase = c.zb - T;
    c.plane_b = 4u * (uint32_t)g.plane;
    {
        const int64_t base = (int64_t)c.zbase * g.plane;
        const uint32_t span = (uint32_t)(c.ze - c.zb + 2 * T + 1) * c.plane_b;
        c.rj = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pin + base), 0, span, 0x00020000);
        c.rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(rhs + base), 0, span, 0x00020000);
        c.rm = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(mask + base), 0, span >> 2, 0x00020000);
        c.ro = __builtin_amdgcn_make_buffer_rsrc(pout + base, 0, span, 0x00020000);
        c.rk = __builtin_amdgcn_make_buffer_rsrc((KEEP ? pmid : pout) + base, 0, span, 0x00020000);
    }

    const int gx0 = (WIN ? rg.xwin0 : 0) + x0;  // global x of this lane's first cell
    c.xin = gx0 < g.W;
    const unsigned xs = c.xin ? (unsigned)gx0 : 0u;
    c.halo_lo = c.rr0 == 0;
    c.halo_hi = c.rr0 + RG == R;
    FusedRowT<T> row[RG];
    const int xl = rg.xwin0 - 1, xr = rg.xwin0 + NT * 256;  // the columns next to the window (WIN)
    const int gxp = c.lane < 32 ? xl : xr;
    c.col_in = (unsigned)gxp < (unsigned)g.W;
    const int yrow0 = y0 - (T - 1);  // row 0 of the workgroup
#pragma unroll
    for (int i = 0; i < RG; i++) {
        const int y = yrow0 + c.rr0 + i;
        const bool yin = (unsigned)y < (unsigned)g.H;
        row[i].yin = yin;
        row[i].out_row = c.rr0 + i >= T - 1 && c.rr0 + i <= R - T && yin;
        row[i].roff4 = (uint32_t)__builtin_amdgcn_readfirstlane(4 * g.W * (yin ? y : 0));
    }
    const int yh = c.halo_lo ? yrow0 - 1 : yrow0 + R;  // the row just outside the workgroup
    const bool is_halo = c.halo_lo || c.halo_hi;
    c.yin_h = is_halo && (unsigned)yh < (unsigned)g.H;
    c.roff4_h = (uint32_t)__builtin_amdgcn_readfirstlane(4 * g.W * (((unsigned)yh < (unsigned)g.H) ? yh : 0));
    c.loff = 4u * xs;
    c.loff_pad = 4u * (c.col_in ? (unsigned)gxp : 0u);
    // the cell across the x-tile boundary: lane 0 -> x0-1, lane 63 -> x0+4; the other lanes read along (cell
    // tx*256 + lane: consecutive banks — their own x0+4 was a four-way bank conflict) and do not use the value
    const int xe = c.lane == 0 ? x0 - 1 : (c.lane == 63 ? x0 + 4 : tx * 256 + c.lane);
    c.pad_writer = (c.lane == 0 && tx == 0) || (c.lane == 63 && tx == NT - 1);
    const int pad_x = c.lane == 0 ? -1 : NT * 256;

    FLUID_LDS float* const ldsp = (FLUID_LDS float*)lds;
    {
        // the three address registers per kind; opaque to the compiler from here on, so that it keeps them
        const uint32_t row0 = (uint32_t)(uintptr_t)ldsp + 4u * (uint32_t)(FUSED_PAD + c.rr0 * RW);
#pragma unroll
        for (int q = 0; q < 3; q++) {
            c.lb[q] = row0 + 4u * (uint32_t)(x0 + q * G::SEGF);
            c.le[q] = row0 + 4u * (uint32_t)(xe + q * G::SEGF);
            c.lp[q] = row0 + 4u * (uint32_t)(pad_x + q * G::SEGF);
            asm volatile("" : "+v"(c.lb[q]), "+v"(c.le[q]), "+v"(c.lp[q]));
        }
    }
    // pad cells of every LDS row: x = -1 and x = NT*256 read as p_oob (outside the grid)
    for (int i = threadIdx.x; i < (G::ROWS + 1) * 2 * FUSED_PAD; i += G::THREADS) {
        const int side = i % (2 * FUSED_PAD), r = i / (2 * FUSED_PAD);
        FLUID_LDS float* base = ldsp + r * RW;
        base[side < FUSED_PAD ? side : RW - 2 * FUSED_PAD + side] = p_air;
    }

    // prologue: the state a step with ring phase 0 and zc = zb - (T-1) expects
    const float4 pa4 = make_float4(p_air, p_air, p_air, p_air);
    float4 h[2];
    int zc = c.zb - (T - 1);  // plane of iterate j+1 formed in the coming step
#pragma unroll
    for (int i = 0; i < RG; i++) {
        FusedRowT<T>& r = row[i];
        const bool row_in = c.xin && r.yin;
        r.it[0][0] = c.fix_j(c.ld4(c.rj, c.j_soff(zc - 1) + r.roff4), row_in, zc - 1);
        r.it[0][1] = c.fix_j(c.ld4(c.rj, c.j_soff(zc) + r.roff4), row_in, zc);
        r.it[0][2] = c.ld4(c.rj, c.j_soff(zc + 1) + r.roff4);  // raw: fixed up by the first step
#pragma unroll
        for (int q = 1; q < T; q++) {
            r.it[q][0] = pa4;  // planes below the first one the march forms: whatever is computed from them is
            r.it[q][1] = pa4;  // never stored and feeds nothing that is
        }
        r.b[0] = r.b[1] = make_float4(0.f, 0.f, 0.f, 0.f);
        r.m[0] = r.m[1] = MASK_DRY4;
        r.b[2] = c.ld4(c.rb, c.m_soff(zc) + r.roff4);
        r.m[2] = __builtin_amdgcn_raw_buffer_load_b32(c.rm, c.loff >> 2, (c.m_soff(zc) + r.roff4) >> 2, 0);
        if (WIN) {
            auto pad_at = [&](int lz) {
                return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(c.rj, c.loff_pad, c.j_soff(lz) + r.roff4, 0));
            };
            r.padv[0] = pad_at(zc - 1);
            r.padv[1] = pad_at(zc);
            r.padv[2] = pad_at(zc + 1);
        }
    }
    h[0] = c.ld4(c.rj, c.j_soff(zc) + c.roff4_h);
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
                if (c.pad_writer)
                    *c.padc(0, q, i) = c.fix_pad(r.padv[q == 0 ? 1 : 0], c.col_in && r.yin, q == 0 ? zc : zc - 1);
            }
        }
    }
    __syncthreads();

    const int steps = c.ze - c.zb + 2 * (T - 1);
#ifdef FLUID_FUSED_TRACE
    FusedTrace ftr;
    for (int i = 0; i < FUSED_TRACE_PHASES; i++) ftr.sum[i] = 0;
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
#endif
    for (int k = 0; k < steps; k += 4, zc += 4) {
        fused_step_t<NT, RG, T, 0, WIN, KEEP, NTS>(c, row, h, zc FT3_PASS);
        if (k + 1 >= steps) break;  // all wave-uniform: every wavefront takes the same barriers
        fused_step_t<NT, RG, T, 1, WIN, KEEP, NTS>(c, row, h, zc + 1 FT3_PASS);
        if (k + 2 >= steps) break;
        fused_step_t<NT, RG, T, 2, WIN, KEEP, NTS>(c, row, h, zc + 2 FT3_PASS);
        if (k + 3 >= steps) break;
        fused_step_t<NT, RG, T, 3, WIN, KEEP, NTS>(c, row, h, zc + 3 FT3_PASS);
    }
#ifdef FLUID_FUSED_TRACE
    {
        const unsigned wg = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        if (wg < 64 && c.lane == 0) {
            unsigned long long* o = g_fused_trace3 + (wg * 16 + wave) * (FUSED3_TRACE_PHASES + 1);
            for (int i = 0; i < FUSED_TRACE_PHASES; i++) o[i] = ftr.sum[i];
            o[6] = __builtin_amdgcn_s_memtime() - t_begin;  // whole march
            o[FUSED3_TRACE_PHASES] = (unsigned long long)steps;
        }
    }
#endif
}

}  // namespace fluid
