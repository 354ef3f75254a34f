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
// Works on the loop's internal working buffers (kernels_pressure.h); rows and planes outside the grid
// are the constant p_oob for both iterates.  On a Z-slab context the working buffers carry ghost
// planes with the neighbouring slab's cells (FusedRange below).
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
// NTS: a streaming store.  The next launch reads these bytes again, so they should stay cached when a
// launch's working set fits the 256 MB memory-side cache (256^3: 48.6 k iterations/s, 37.0 k with streaming
// stores) and should not take cache space when it does not (512^3: 5 690 -> 5 920; a copy kernel over 1 GB:
// 4.9 -> 5.2 TB/s, tools/micro/mall_bw.hip).  A template parameter, chosen by the launch code: a wave-uniform
// branch around the two kinds of store cost the kernel 6 % (hipcc's waits for the loads then include them).
// The streaming kernels are instantiated in a translation unit of their own (pressure_fused_stream.hip):
// beside the others they moved the schedule of the one-tile kernel (256^3: - 6 %).
template <bool NTS>
__device__ __forceinline__ void st_f4(float* base, unsigned byte_off, float4 v) {
    if constexpr (NTS) {
        const f32x4 r = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(r, reinterpret_cast<f32x4*>(reinterpret_cast<char*>(base) + byte_off));
    } else {
        *reinterpret_cast<float4*>(reinterpret_cast<char*>(base) + byte_off) = v;
    }
}

// ---- the division of pressure.comp:62 without the IEEE division sequence -----------------------------
// A water cell's new pressure is n / aii with n = -s and aii in {0..6} (its mask byte).  hipcc expands an
// fp32 division into ~11 VALU instructions (v_div_scale x2, v_rcp, 5 FMAs, v_div_fmas, v_div_fixup),
// about 40 % of this kernel's arithmetic, and the kernel is issue-bound.  For a divisor known to be a
// small integer the correctly rounded quotient takes three:
//     q0 = n * r;  e = fma(-q0, a, n);  q = fma(e, r, q0)          with r = RN(1/a) from a table
// (Markstein's correction step: e is the exact remainder).  tests/divide_small_int_check.c compares
// this chain with n / a for ALL 2^32 fp32 values of n and a = 1..6 on the CPU: bit-identical except
// for a = 6 with |n| < 2^-125; a wave-uniform test sends any wavefront holding 0 < |n| < 2^-100 (kernels of one
// x tile: |n| < 2^-100, zeros included; canon_div4) down the IEEE path instead (2^-100 also keeps e out of the denormal range, so the result does not depend on the
// kernel's denormal mode).  v_div_fixup_f32 — the last instruction of the IEEE sequence, with the same
// operands — supplies the IEEE results for n = +-0, inf, NaN and a = 0 (inf / NaN), bit for bit what the
// division gives.  The test reads the exponents (v_frexp_exp_i32_f32: 0 for zero, inf and NaN) so that an
// exact zero does not count as tiny: numerators of exactly 0 are everyday — dry lanes, cells that touch only
// solids where the divergence is 0 — and with a test on |n| alone they sent a full tank's wavefronts down
// the slow path (13 % of the launch, tools/dense_jacobi_probe.py).
struct DivEntry {
    float a, r;
};
constexpr int DIV_TABLE_ENTRIES = 9;  // mask bytes 0..6 (aii) and MASK_DRY (result unused)
__device__ __forceinline__ float2 lds_ld2(const FLUID_LDS char* base, uint32_t byte_off) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 v = *(const FLUID_LDS f32x2*)(base + byte_off);
    return make_float2(v.x, v.y);
}
__device__ __forceinline__ float div_small_int(float n, float2 ar) {
    const float q0 = n * ar.y;
    const float e = __builtin_fmaf(-q0, ar.x, n);
    const float q = __builtin_fmaf(e, ar.y, q0);
    return __builtin_amdgcn_div_fixupf(q, ar.x, n);
}
// numerator of one water cell: -(b - sum of the six neighbours), pressure.comp:56-61 order
__device__ __forceinline__ float canon_num(float b, float qxp, float qyp, float qzp, float qxm,
                                           float qym, float qzm) {
    float s = b;
    s = s - qxp;
    s = s - qyp;
    s = s - qzp;
    s = s - qxm;
    s = s - qym;
    s = s - qzm;
    return -s;
}
// The (a, r) pairs of the four cells of a lane, from the LDS table.  Issued BEFORE the plane step's barrier
// (the mask word is known by then), so that the reads are in flight while the wave waits there instead of
// forming a second and third LDS round trip behind the neighbour rows.
struct DivPairs {
    float2 c[4];
};
__device__ __forceinline__ DivPairs div_pairs(uint32_t m, const FLUID_LDS char* table) {
    const uint32_t m8 = m << 3;  // byte i of m8 = 8 * (mask byte i) = offset of its table entry
    DivPairs d;
    d.c[0] = lds_ld2(table, m8 & 0xFFu);
    d.c[1] = lds_ld2(table, (m8 >> 8) & 0xFFu);
    d.c[2] = lds_ld2(table, (m8 >> 16) & 0xFFu);
    d.c[3] = lds_ld2(table, m8 >> 24);
    return d;
}
// the four quotients of a lane: n / aii, aii = bytes of m, (aii, RN(1 / aii)) = d.
// ZEROS_QUICK: exact zeros do not count as tiny (the exponents are read when the cheap look fires).  The
// kernels of wide grids (NT >= 2), whose launches are bound by bandwidth, are built that way — a full 512^3
// tank runs 13 % faster for it; those of one x tile (grids up to 256 cells wide, x-window launches on a small
// box of water) run launches bound by latency, where the longer code costs 4-7 % whether or not it is taken
// (tools/ab_libs.py at 128^3 and 256^3), and keep the plain test: wavefronts with a zero take the IEEE
// sequence there, as exact as the other.
template <bool ZEROS_QUICK>
__device__ __forceinline__ float4 canon_div4(float4 n, uint32_t m, const DivPairs& d) {
    auto quick = [&]() {
        float4 o;
        o.x = div_small_int(n.x, d.c[0]);
        o.y = div_small_int(n.y, d.c[1]);
        o.z = div_small_int(n.z, d.c[2]);
        o.w = div_small_int(n.w, d.c[3]);
        return o;
    };
    // wave-uniform: one cheap look (three v_min with |.|) and one branch in the common case
    const float least = fminf(fminf(fabsf(n.x), fabsf(n.y)), fminf(fabsf(n.z), fabsf(n.w)));
    if (__builtin_amdgcn_ballot_w64(least < 0x1p-100f) == 0ull) return quick();
    if (ZEROS_QUICK) {
        const int ex = min(min(__builtin_amdgcn_frexp_expf(n.x), __builtin_amdgcn_frexp_expf(n.y)),
                           min(__builtin_amdgcn_frexp_expf(n.z), __builtin_amdgcn_frexp_expf(n.w)));
        if (__builtin_amdgcn_ballot_w64(ex < -99) == 0ull) return quick();  // only zeros were small
    }
    float4 o;  // some |n| < 2^-100 in this wavefront: the IEEE sequence for all its lanes
    o.x = n.x / (float)(m & 0xFFu);
    o.y = n.y / (float)((m >> 8) & 0xFFu);
    o.z = n.z / (float)((m >> 16) & 0xFFu);
    o.w = n.w / (float)(m >> 24);
    return o;
}

// Dev build only (make trace: -DFLUID_FUSED_TRACE): where the cycles of a plane step go.  Every wavefront
// stamps s_memtime at six points of the step and sums the five phases over its march; wavefront w of the
// first 64 workgroups stores its sums in g_fused_trace (tools/fused_trace.py reads them).  Not compiled into
// the product library.
#ifdef FLUID_FUSED_TRACE
constexpr int FUSED_TRACE_PHASES = 6;
__device__ unsigned long long g_fused_trace[64 * 16 * (FUSED_TRACE_PHASES + 1)];
struct FusedTrace {
    unsigned long long sum[FUSED_TRACE_PHASES], last;
    unsigned steps;
};
#define FLUID_TRACE_ARG , FusedTrace& ftr
#define FLUID_TRACE_PASS , ftr
#define FT_BEGIN() ftr.last = __builtin_amdgcn_s_memtime()
#define FT(i)                                                        \
    do {                                                             \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
        ftr.sum[i] += t_ - ftr.last;                                 \
        ftr.last = t_;                                               \
    } while (0)
#else
#define FLUID_TRACE_ARG
#define FLUID_TRACE_PASS
#define FT_BEGIN()
#define FT(i)
#endif

constexpr int FUSED_PAD = 4;  // floats of padding on each side of an LDS row (keeps rows 16-B aligned)

// Shape of a workgroup.  NT = 256-cell x tiles per row (the workgroup spans the whole x extent, or the
// window), RG = rows per wavefront.  RG = 1: 16 wavefronts of one row each.  RG >= 2: 8 wavefronts (two
// per SIMD, 256 VGPRs each) that own RG adjacent rows each: the y neighbours inside a wavefront's rows
// are registers, only the group's edge rows go through LDS, the per-wavefront overhead of a plane step
// (addresses, waits, the barrier) is paid once per RG rows, and RG independent rows interleave in the
// one instruction stream — the SIMDs execute their wavefronts one after the other (oldest first) between
// two barriers, so work per wavefront, not wavefronts per SIMD, is what fills the VALU.
constexpr int fused_waves(int rg) { return rg == 1 ? 16 : rg == 2 ? 12 : 8; }
template <int NT, int RG>
struct FusedGeom {
    static constexpr int WAVES = fused_waves(RG);
    static constexpr int THREADS = WAVES * 64;
    static constexpr int GROUPS = WAVES / NT;  // row groups = wavefronts per x tile
    static constexpr int R = GROUPS * RG;      // rows of iterate j+1 per workgroup
    static constexpr int TY = R - 2;           // output rows per workgroup
    static constexpr int RW = NT * 256 + 2 * FUSED_PAD;
    static_assert(GROUPS >= 2, "a wavefront is the lower or the upper edge of its workgroup, not both");
    static constexpr size_t lds_bytes =
        (size_t)2 /*buffers*/ * 2 /*J,S*/ * R * RW * sizeof(float) + 128 /* DivEntry table */;
};

// Plane ranges of one launch (local plane indices; ghost planes are negative or >= Dl):
//   [zout_lo, zout_hi)  planes of iterate j+2 this launch writes.  On a Z slab it may reach into the
//                       ghost planes: with h valid ghost planes of iterate j per side a launch leaves
//                       h-2 valid ghost planes of iterate j+2, so the slabs exchange h planes every
//                       h sweeps instead of 2 planes every 2 (same bytes, h/2 times fewer messages)
//   [jlo, jhi)          planes of the input buffer that hold cells of the grid
//   [mlo, mhi)          same for mask / b_i
//   ytile0              first row tile of the launch (blockIdx.y = 0); launches of sparse scenes cover
//                       only the tiles and planes around the water (ActiveBox, pressure_api.h)
//   [hole_lo, hole_hi)  planes inside [zout_lo, zout_hi) this launch leaves out (a Z slab computes the
//                       planes near its faces and the planes in between in separate launches, so that
//                       the halo exchange overlaps the larger one); nz_lo = z-chunks below the hole.
//                       No hole: hole_lo = hole_hi = zout_hi, nz_lo = all chunks.
//   xwin0               (WIN kernels only) global x of the workgroups' first cell: the launch
//                       covers the window [xwin0, xwin0 + NT*256) of every row instead of the whole row.
//                       Valid when every water cell of the grid lies inside the window: the columns
//                       just outside it then hold non-water constants, the same in every iterate, which
//                       the edge lanes load into the pad cells of the LDS rows.
//   xcd_rows, xcd_start xcd_rows > 0: the launch is a 1-D grid; workgroup L runs on XCD L % 8 (the dispatcher
//                       deals workgroups to the 8 XCDs in turn) and takes unit xcd_start[L % 8] + L / 8 of the
//                       launch's (chunk, row tile) units, row tile first — a contiguous range per XCD, cut
//                       by the host so that the ranges cost the same — so that workgroups which share halo
//                       rows share an L2.  xcd_rows = row tiles in the launch.
struct FusedRange {
    int zout_lo, zout_hi, jlo, jhi, mlo, mhi, ytile0, hole_lo, hole_hi, nz_lo, xwin0, xcd_rows;
    int xcd_start[9];
};

// State of one row of a wavefront during the z march.  Everything rotates with period 4 (the z loop is
// unrolled by 4, so ring indices are compile-time constants and the rotation costs no register moves):
//   j[4]  iterate j   : slots (i, i+1, i+2) = planes zc-1, zc, zc+1; slot i+3 receives plane zc+2
//   s[4]  iterate j+1 : slots (i, i+1) = planes zc-2, zc-1; slot i+2 receives plane zc
//   b[4], m[4]        : slots (i+1, i+2) = planes zc-1 (stage 2), zc (stage 1); slot i+3 receives zc+1
//   padv[4] (windowed launches) : iterate j at the column just outside the window, slots (i, i+1) =
//                                  planes zc-1, zc; slot i+2 receives plane zc+1
struct FusedRow {
    float4 j[4], s[4], b[4];
    uint32_t m[4];
    float padv[4];
    int y;                    // the row (SOR kernels: its parity)
    unsigned boff, boff_pad;  // in-plane byte offsets of this lane's cells / of its pad column in this row
    bool row_in, pad_in;      // those cells exist
    bool is_out_row;          // ... and the row is one the workgroup writes
};

template <int NT, int RG>
struct FusedCtx {
    using G = FusedGeom<NT, RG>;
    const uint8_t* mask;
    const float* rhs;
    const float* pin;
    float* pout;
    float* pmid;
    FLUID_LDS float* lds;
    const FLUID_LDS char* divtab;  // DivEntry[DIV_TABLE_ENTRIES], behind the row buffers
    int64_t plane;
    int Dl, zb, ze;
    int jlo, jhi;  // local planes [jlo, jhi) of the working buffers hold cells of the grid: the owned
                   // planes plus two ghost planes per side where a neighbouring slab exists
    int mlo, mhi;  // same for mask / b_i (one ghost plane per side)
    int lane, rr0, x0, xe;  // rr0 = first row (within the workgroup) of this wavefront's group
    unsigned boff_h;        // the row just outside the workgroup (edge wavefronts)
    bool pad_writer;
    int pad_x;
    float p_oob;
    bool halo_in;           // per lane: that outer row's cells exist
    bool halo_lo, halo_hi;  // wave-uniform: this group holds row 0 / row R-1 of the workgroup, which only
                            // form iterate j+1 and whose outer y neighbour comes from global memory
    bool wave_clean;        // every cell of this wavefront (and of its outer row) lies inside the grid
    float omega;            // SOR kernels: the relaxation factor; gz0 = global z of local plane 0
    int gz0;

    __device__ __forceinline__ FLUID_LDS float* row_ptr(int buf, int arr, int row) const {
        return lds + ((buf * 2 + arr) * G::R + row) * G::RW + FUSED_PAD;
    }
    __device__ __forceinline__ bool j_ok(int lz) const { return lz >= jlo && lz < jhi; }
    __device__ __forceinline__ bool m_ok(int lz) const { return lz >= mlo && lz < mhi; }
    // element offset of plane lz, redirected to plane 0 when lz is outside the grid: loads are always
    // issued, from a valid address, and the value is replaced later (fix_*).  A select directly on a
    // load makes hipcc wait for it (vmcnt(0)) on the spot.
    __device__ __forceinline__ int64_t j_off(int lz) const {
        return j_ok(lz) ? (int64_t)lz * plane : (int64_t)0;
    }
    __device__ __forceinline__ int64_t m_off(int lz) const {
        return m_ok(lz) ? (int64_t)lz * plane : (int64_t)0;
    }
    __device__ __forceinline__ float4 fix_j(float4 v, bool ok, int lz) const {
        if (wave_clean && j_ok(lz)) return v;  // wave-uniform: the common case costs nothing
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

// the four cells of a lane: numerators from the six neighbours, quotients, water cells take them.
// SOR (the opt-in red-black solver, fluid_set_pressure_solver): the quotient is the Gauss-Seidel value gs and
// only the cells of the stage's colour move, c + omega * (gs - c) (oracle_12_sor_iteration's operations);
// a lane's cells 0, 2 have one colour and 1, 3 the other: `odd` = the stage's colour sits on cells 1, 3.
template <bool SOR, bool ZEROS_QUICK>
__device__ __forceinline__ float4 canon_lane(float4 b, uint32_t m, float4 c, float4 yp, float4 zp,
                                             float4 ym, float4 zm, float left, float right,
                                             const DivPairs& d, float omega = 0.f, bool odd = false) {
    float4 n;
    n.x = canon_num(b.x, c.y, yp.x, zp.x, left, ym.x, zm.x);
    n.y = canon_num(b.y, c.z, yp.y, zp.y, c.x, ym.y, zm.y);
    n.z = canon_num(b.z, c.w, yp.z, zp.z, c.y, ym.z, zm.z);
    n.w = canon_num(b.w, right, yp.w, zp.w, c.z, ym.w, zm.w);
    float4 o = canon_div4<ZEROS_QUICK>(n, m, d);
    if (SOR) {
        o.x = c.x + omega * (o.x - c.x);
        o.y = c.y + omega * (o.y - c.y);
        o.z = c.z + omega * (o.z - c.z);
        o.w = c.w + omega * (o.w - c.w);
        o.x = (mask_is_water(m, 0) && !odd) ? o.x : c.x;
        o.y = (mask_is_water(m, 1) && odd) ? o.y : c.y;
        o.z = (mask_is_water(m, 2) && !odd) ? o.z : c.z;
        o.w = (mask_is_water(m, 3) && odd) ? o.w : c.w;
        return o;
    }
    o.x = mask_is_water(m, 0) ? o.x : c.x;  // non-water (and out-of-grid) cells keep their constant
    o.y = mask_is_water(m, 1) ? o.y : c.y;
    o.z = mask_is_water(m, 2) ? o.z : c.z;
    o.w = mask_is_water(m, 3) ? o.w : c.w;
    return o;
}

// One plane step: I = ring phase (k mod 4), zc = plane of iterate j+1 formed in this step.  A step begins
// right behind the barrier that made the rows of plane zc (iterate j) and zc-1 (iterate j+1) visible, and
// ends with the barrier of the next one.  All wavefronts of a workgroup move through it in lock step, so a
// resource used in one burst (the 16 wavefronts' loads in the texture addresser, their LDS reads) idles
// the others; the order below spreads them out:
//   LDS reads -> loads of iterate j two planes ahead -> (wait for the loads of the PREVIOUS step) fix-ups,
//   table look-ups -> stage 1 -> loads of b_i / mask one plane ahead -> stage 2 -> stores -> publish -> barrier.
// Vector-memory operations retire in issue order and the compiler's s_waitcnt counts only what is issued
// on every path, so the stores carry all their conditions in the EXEC mask (no branch around them): the
// wait for a step's loads then never includes the stores issued after them.
template <int NT, int RG, int I, bool WIN, bool KEEP, bool SOR, bool NTS>
__device__ __forceinline__ void fused_step(const FusedCtx<NT, RG>& c, FusedRow (&row)[RG], float4 (&h)[2],
                                           int zc FLUID_TRACE_ARG) {
    constexpr int buf = I & 1;
    constexpr int JM = I & 3, JC = (I + 1) & 3, JN = (I + 2) & 3, JL = (I + 3) & 3;  // j ring
    constexpr int SMM = I & 3, SM = (I + 1) & 3, SC = (I + 2) & 3;                    // s ring
    constexpr int BM = (I + 1) & 3, BC = (I + 2) & 3, BL = (I + 3) & 3;               // b, m rings
    constexpr bool X_EDGE_FROM_LDS = NT > 1 || WIN;  // else the cells beyond the row ends are p_oob
    const bool is_halo = c.halo_lo || c.halo_hi;     // wave-uniform

    FT_BEGIN();
    // ---- everything both stages need from LDS, in one round trip: the rows next to the group, the cells
    // across the x-tile seams
    float4 jext_lo, jext_hi, sext_lo, sext_hi;
    float je[RG], se[RG];
    if (!c.halo_lo) {
        jext_lo = lds_ld4(c.row_ptr(buf, 0, c.rr0 - 1) + c.x0);
        sext_lo = lds_ld4(c.row_ptr(buf, 1, c.rr0 - 1) + c.x0);
    }
    if (!c.halo_hi) {
        jext_hi = lds_ld4(c.row_ptr(buf, 0, c.rr0 + RG) + c.x0);
        sext_hi = lds_ld4(c.row_ptr(buf, 1, c.rr0 + RG) + c.x0);
    }
#pragma unroll
    for (int i = 0; i < RG; i++) {
        je[i] = X_EDGE_FROM_LDS ? c.row_ptr(buf, 0, c.rr0 + i)[c.xe] : c.p_oob;
        se[i] = X_EDGE_FROM_LDS ? c.row_ptr(buf, 1, c.rr0 + i)[c.xe] : c.p_oob;
    }

    // ---- iterate j two planes ahead (raw; fixed up in the next step)
    const int64_t o2 = c.j_off(zc + 2);
#pragma unroll
    for (int i = 0; i < RG; i++) {
        row[i].j[JL] = ld_f4(c.pin + o2, row[i].boff);
        if (WIN)
            row[i].padv[(I + 3) & 3] = *reinterpret_cast<const float*>(
                reinterpret_cast<const char*>(c.pin + o2) + row[i].boff_pad);
    }

    // ---- what the previous step loaded, fixed up (only wavefronts / planes on the grid boundary do
    // anything here; those loads have had a whole step to land)
    uint32_t m_c[RG];
    bool water1 = false, wet2 = false;
    bool wet[RG];
    const int zo = zc - 1;
    const bool zo_in = zo >= c.zb && zo < c.ze;  // wave-uniform
#pragma unroll
    for (int i = 0; i < RG; i++) {
        row[i].j[JN] = c.fix_j(row[i].j[JN], row[i].row_in, zc + 1);
        m_c[i] = c.fix_m(row[i].m[BC], row[i].row_in, zc);
        row[i].m[BC] = m_c[i];
        water1 = water1 || mask_any_water(m_c[i]);
        wet[i] = zo_in && row[i].is_out_row && mask_any_water(row[i].m[BM]);
        wet2 = wet2 || wet[i];
    }
    if (is_halo) {
        const float4 hc = c.fix_j(h[I & 1], c.halo_in, zc);
        if (c.halo_lo) jext_lo = hc; else jext_hi = hc;
        if (c.halo_lo) sext_lo = hc; else sext_hi = hc;  // not used: the edge row forms no iterate j+2
    }
    const bool do1 = __builtin_amdgcn_ballot_w64(water1) != 0ull;  // wave-uniform
    const bool do2 = __builtin_amdgcn_ballot_w64(wet2) != 0ull;    // wave-uniform
    FT(0);  // LDS reads issued, j loads issued, wait for the previous step's loads

    // ---- stage 1: iterate j+1 at plane zc for every row
#pragma unroll
    for (int i = 0; i < RG; i++) row[i].s[SC] = row[i].j[JC];  // non-water cells keep their constant
    if (do1) {
        DivPairs d1[RG];
#pragma unroll
        for (int i = 0; i < RG; i++) d1[i] = div_pairs(m_c[i], c.divtab);
#pragma unroll
        for (int i = 0; i < RG; i++) {
            const float4 jc = row[i].j[JC];
            const float4 ym = i > 0 ? row[i > 0 ? i - 1 : 0].j[JC] : jext_lo;
            const float4 yp = i < RG - 1 ? row[i < RG - 1 ? i + 1 : 0].j[JC] : jext_hi;
            const float left = from_lane_below(jc.w, je[i], c.lane);
            const float right = from_lane_above(jc.x, je[i], c.lane);
            // SOR: stage 1 moves the cells with (x + y + z) even; x of a lane's cell 0 is a multiple of 4
            row[i].s[SC] = canon_lane<SOR, (NT >= 2)>(row[i].b[BC], m_c[i], jc, yp, row[i].j[JN], ym, row[i].j[JM], left,
                                           right, d1[i], c.omega, ((row[i].y + c.gz0 + zc) & 1) != 0);
        }
    }
#ifdef FLUID_FUSED_TRACE
    asm volatile("" ::"v"(row[0].s[SC].x), "v"(row[RG - 1].s[SC].w));  // stage 1 is done here
#endif
    FT(1);  // stage 1

    // ---- b_i, mask (and the row outside the workgroup) one plane ahead
    const int64_t o1 = c.j_off(zc + 1), a1 = c.m_off(zc + 1);
#pragma unroll
    for (int i = 0; i < RG; i++) {
        row[i].b[BL] = ld_f4(c.rhs + a1, row[i].boff);
        row[i].m[BL] = ld_u32(c.mask + a1, row[i].boff >> 2);
    }
    if (is_halo) h[(I + 1) & 1] = ld_f4(c.pin + o1, c.boff_h);

    // ---- stage 2: iterate j+2 at plane zc-1 from iterate j+1 at planes zc-2, zc-1, zc
    float4 o[RG];
#pragma unroll
    for (int i = 0; i < RG; i++) o[i] = row[i].s[SM];
    if (do2) {
        DivPairs d2[RG];
#pragma unroll
        for (int i = 0; i < RG; i++) d2[i] = div_pairs(row[i].m[BM], c.divtab);
#pragma unroll
        for (int i = 0; i < RG; i++) {
            // rows 0 and R-1 of the workgroup only form iterate j+1 (wave-uniform tests)
            if ((i == 0 && c.halo_lo) || (i == RG - 1 && c.halo_hi)) continue;
            const float4 sm = row[i].s[SM];
            const float4 ym = i > 0 ? row[i > 0 ? i - 1 : 0].s[SM] : sext_lo;
            const float4 yp = i < RG - 1 ? row[i < RG - 1 ? i + 1 : 0].s[SM] : sext_hi;
            const float left = from_lane_below(sm.w, se[i], c.lane);
            const float right = from_lane_above(sm.x, se[i], c.lane);
            o[i] = canon_lane<SOR, (NT >= 2)>(row[i].b[BM], row[i].m[BM], sm, yp, row[i].s[SC], ym, row[i].s[SMM], left,
                                   right, d2[i], c.omega, ((row[i].y + c.gz0 + zo) & 1) == 0);  // stage 2: odd cells
        }
    }
    // stores: every condition is in the lane predicate (see above); a lane stores only if one of its
    // four cells is water — the others hold the same constant in every working buffer
    {
        const int64_t oo = (int64_t)(zo_in ? zo : c.zb) * c.plane;
#pragma unroll
        for (int i = 0; i < RG; i++) {
            if (wet[i]) st_f4<NTS>(c.pout + oo, row[i].boff, o[i]);
            if (KEEP) {  // the odd iterate, kept only by the last pair of a loop
                if (wet[i]) st_f4<NTS>(c.pmid + oo, row[i].boff, row[i].s[SM]);
            }
        }
    }
    FT(2);  // b / mask loads, stage 2, stores

    // ---- publish the rows for the next step: iterate j at plane zc+1, iterate j+1 at plane zc
    constexpr int nbuf = buf ^ 1;
#pragma unroll
    for (int i = 0; i < RG; i++) {
        // rows inside a group are only needed across the x-tile seam
        if (i == 0 || i == RG - 1 || X_EDGE_FROM_LDS) {
            FLUID_LDS float* jrow = c.row_ptr(nbuf, 0, c.rr0 + i);
            FLUID_LDS float* srow = c.row_ptr(nbuf, 1, c.rr0 + i);
            lds_st4(jrow + c.x0, row[i].j[JN]);
            lds_st4(srow + c.x0, row[i].s[SC]);
            if (WIN) {
                // the columns next to the window hold non-water constants: the same value in both iterates
                if (c.pad_writer) {
                    jrow[c.pad_x] = c.fix_pad(row[i].padv[(I + 2) & 3], row[i].pad_in, zc + 1);
                    srow[c.pad_x] = c.fix_pad(row[i].padv[(I + 1) & 3], row[i].pad_in, zc);
                }
            }
        }
    }
    FT(3);  // publish
    __syncthreads();
    FT(4);  // barrier
}

template <int NT, bool WIN, int RG, bool KEEP, bool SOR, bool NTS = false>
__global__ void __launch_bounds__(fused_waves(RG) * 64)
k12_canon2(const uint8_t* __restrict__ mask, const float* __restrict__ rhs,
           const float* __restrict__ pin, float* __restrict__ pout, float* __restrict__ pmid,
           const uint8_t* __restrict__ active, BrickK bk, GridK g, float p_air, int zchunk,
           FusedRange rg, float omega) {
    using G = FusedGeom<NT, RG>;
    constexpr int R = G::R, TY = G::TY, RW = G::RW;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    FusedCtx<NT, RG> c;
    c.mask = mask;
    c.rhs = rhs;
    c.pin = pin;
    c.pout = pout;
    c.pmid = pmid;
    c.lds = (FLUID_LDS float*)lds;
    c.plane = g.plane;
    c.Dl = g.Dl;
    c.jlo = rg.jlo;
    c.jhi = rg.jhi;
    c.mlo = rg.mlo;
    c.mhi = rg.mhi;
    c.p_oob = p_air;
    c.omega = omega;
    c.gz0 = g.z0;
    c.lane = threadIdx.x & 63;
    // readfirstlane: tells hipcc the wave index (hence rows, tile and edge role) is wave-uniform, so it
    // lives in SGPRs and role tests become scalar branches instead of exec masking
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int tx = wave % NT;
    c.rr0 = (wave / NT) * RG;
    c.x0 = tx * 256 + c.lane * 4;
    int tile_y = (int)blockIdx.y, tile_z = (int)blockIdx.z;
    if (rg.xcd_rows > 0) {
        // XCD `xcd` takes the units [xcd_start[xcd], xcd_start[xcd + 1]): the workgroups it runs side by side
        // are y-neighbours of one chunk, whose shared halo rows its L2 then serves
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
        // the whole group leaves if no brick it touches holds water (uniform: before any barrier);
        // groups that write ghost planes always run (the activity map covers owned planes only)
        uint32_t any = 0;
        const int by0 = max(y0 - 1, 0) / BRICK_Y, by1 = min(y0 + TY, g.H - 1) / BRICK_Y;
        const int bz0 = max(c.zb - 1, 0) / BRICK_Z, bz1 = min(c.ze, g.Dl - 1) / BRICK_Z;
        for (int bz = bz0; bz <= bz1; bz++)
            for (int by = by0; by <= by1; by++)
                for (int bx = 0; bx < bk.nbx; bx++) any |= active[brick_index(bk, bx, by, bz)];
        if (any == 0) return;
    }

    const int gx0 = (WIN ? rg.xwin0 : 0) + c.x0;     // global x of this lane's first cell
    const bool xin = gx0 < g.W;
    const unsigned xs = xin ? (unsigned)gx0 : 0u;
    c.halo_lo = c.rr0 == 0;
    c.halo_hi = c.rr0 + RG == R;
    FusedRow row[RG];
    bool dirty = false;
    const int xl = rg.xwin0 - 1, xr = rg.xwin0 + NT * 256;  // the columns next to the window (WIN)
    const int gxp = c.lane < 32 ? xl : xr;
    const bool col_in = (unsigned)gxp < (unsigned)g.W;
#pragma unroll
    for (int i = 0; i < RG; i++) {
        const int y = y0 - 1 + c.rr0 + i;  // this row
        row[i].y = y;
        const bool yin = (unsigned)y < (unsigned)g.H;
        row[i].row_in = xin && yin;        // this lane's cells exist
        row[i].is_out_row = c.rr0 + i >= 1 && c.rr0 + i <= R - 2 && row[i].row_in;
        // in-plane byte offsets (safe addresses for lanes / rows outside the grid)
        row[i].boff = 4u * (xs + (unsigned)g.W * (unsigned)(yin ? y : 0));
        row[i].pad_in = col_in && yin;
        row[i].boff_pad = 4u * ((col_in ? (unsigned)gxp : 0u) + (unsigned)g.W * (unsigned)(yin ? y : 0));
        dirty = dirty || !row[i].row_in;
    }
    const int yh = c.halo_lo ? y0 - 2 : y0 - 1 + R;  // the row just outside the workgroup
    const bool is_halo = c.halo_lo || c.halo_hi;
    c.halo_in = is_halo && xin && (unsigned)yh < (unsigned)g.H;
    c.boff_h = 4u * (xs + (unsigned)g.W * (unsigned)(((unsigned)yh < (unsigned)g.H) ? yh : 0));
    c.wave_clean = __builtin_amdgcn_ballot_w64(dirty || (is_halo && !c.halo_in)) == 0ull;
    // the cell across the x-tile boundary: lane 0 -> x0-1, lane 63 -> x0+4 (other lanes: harmless)
    c.xe = c.lane == 0 ? c.x0 - 1 : c.x0 + 4;
    c.pad_writer = (c.lane == 0 && tx == 0) || (c.lane == 63 && tx == NT - 1);
    c.pad_x = c.lane == 0 ? -1 : NT * 256;

    {   // DivEntry table
        FLUID_LDS float* tab = c.lds + 2 * 2 * R * RW;  // DivEntry {a, r} pairs
        c.divtab = (const FLUID_LDS char*)tab;
        if (threadIdx.x < DIV_TABLE_ENTRIES) {
            const float a = (float)threadIdx.x;
            tab[2 * threadIdx.x] = a;
            tab[2 * threadIdx.x + 1] = threadIdx.x == 0 ? 0.0f : 1.0f / a;  // RN(1/a): IEEE division
        }
    }
    // pad cells of every LDS row: x = -1 and x = NT*256 read as p_oob (outside the grid)
    for (int i = threadIdx.x; i < 2 * 2 * R * 2 * FUSED_PAD; i += G::THREADS) {
        const int side = i % (2 * FUSED_PAD), r = i / (2 * FUSED_PAD);
        FLUID_LDS float* base = c.lds + r * RW;
        base[side < FUSED_PAD ? side : RW - 2 * FUSED_PAD + side] = p_air;
    }

    // prologue: the state a step with ring phase 0 and zc = zb - 1 expects
    const float4 pa4 = make_float4(p_air, p_air, p_air, p_air);
    float4 h[2];
    int zc = c.zb - 1;  // plane of iterate j+1 formed in the coming step
#pragma unroll
    for (int i = 0; i < RG; i++) {
        FusedRow& r = row[i];
        r.j[0] = c.fix_j(ld_f4(pin + c.j_off(zc - 1), r.boff), r.row_in, zc - 1);
        r.j[1] = c.fix_j(ld_f4(pin + c.j_off(zc), r.boff), r.row_in, zc);
        r.j[2] = ld_f4(pin + c.j_off(zc + 1), r.boff);  // raw: fixed up by the first step
        r.s[0] = pa4;
        r.s[1] = pa4;
        r.b[1] = make_float4(0.f, 0.f, 0.f, 0.f);
        r.m[1] = MASK_DRY4;
        r.b[2] = ld_f4(rhs + c.m_off(zc), r.boff);
        r.m[2] = ld_u32(mask + c.m_off(zc), r.boff >> 2);
        if (WIN) {
            auto pad_at = [&](int lz) {
                return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(pin + c.j_off(lz)) +
                                                       r.boff_pad);
            };
            r.padv[0] = pad_at(zc - 1);
            r.padv[1] = pad_at(zc);
        }
    }
    h[0] = ld_f4(pin + c.j_off(zc), c.boff_h);
    __syncthreads();  // pad cells are in place (the rows below overwrite two of them in windowed launches)
    // what the first step expects behind its barrier: iterate j at plane zc, iterate j+1 (constants) at zc-1
#pragma unroll
    for (int i = 0; i < RG; i++) {
        FusedRow& r = row[i];
        if (WIN) r.padv[2] = *reinterpret_cast<const float*>(
                     reinterpret_cast<const char*>(pin + c.j_off(zc + 1)) + r.boff_pad);
        FLUID_LDS float* jrow = c.row_ptr(0, 0, c.rr0 + i);
        FLUID_LDS float* srow = c.row_ptr(0, 1, c.rr0 + i);
        lds_st4(jrow + c.x0, r.j[1]);
        lds_st4(srow + c.x0, r.s[1]);
        if (WIN) {
            if (c.pad_writer) {
                jrow[c.pad_x] = c.fix_pad(r.padv[1], r.pad_in, zc);
                srow[c.pad_x] = c.fix_pad(r.padv[0], r.pad_in, zc - 1);
            }
        }
    }
    __syncthreads();

    const int steps = c.ze - c.zb + 2;  // iterate j+1 at planes zb-1 .. ze, iterate j+2 one behind
#ifdef FLUID_FUSED_TRACE
    FusedTrace ftr;
    for (int i = 0; i < FUSED_TRACE_PHASES; i++) ftr.sum[i] = 0;
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
#endif
    for (int k = 0; k < steps; k += 4, zc += 4) {
        fused_step<NT, RG, 0, WIN, KEEP, SOR, NTS>(c, row, h, zc FLUID_TRACE_PASS);
        if (k + 1 >= steps) break;  // all wave-uniform: every wavefront takes the same barriers
        fused_step<NT, RG, 1, WIN, KEEP, SOR, NTS>(c, row, h, zc + 1 FLUID_TRACE_PASS);
        if (k + 2 >= steps) break;
        fused_step<NT, RG, 2, WIN, KEEP, SOR, NTS>(c, row, h, zc + 2 FLUID_TRACE_PASS);
        if (k + 3 >= steps) break;
        fused_step<NT, RG, 3, WIN, KEEP, SOR, NTS>(c, row, h, zc + 3 FLUID_TRACE_PASS);
    }
#ifdef FLUID_FUSED_TRACE
    {
        const unsigned wg = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        if (wg < 64 && c.lane == 0) {
            unsigned long long* o = g_fused_trace + (wg * 16 + wave) * (FUSED_TRACE_PHASES + 1);
            ftr.sum[5] = __builtin_amdgcn_s_memtime() - t_begin;  // whole march
            for (int i = 0; i < FUSED_TRACE_PHASES; i++) o[i] = ftr.sum[i];
            o[FUSED_TRACE_PHASES] = (unsigned long long)steps;
        }
    }
#endif
}

}  // namespace fluid
