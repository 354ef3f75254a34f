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
__device__ __forceinline__ void st_f4(float* base, unsigned byte_off, float4 v) {
    *reinterpret_cast<float4*>(reinterpret_cast<char*>(base) + byte_off) = v;
}

// ---- the division of pressure.comp:62 without the IEEE division sequence -----------------------------
// A water cell's new pressure is n / aii with n = -s and aii in {0..6} (its mask byte).  hipcc expands an
// fp32 division into ~11 VALU instructions (v_div_scale x2, v_rcp, 5 FMAs, v_div_fmas, v_div_fixup),
// about 40 % of this kernel's arithmetic, and the kernel is issue-bound.  For a divisor known to be a
// small integer the correctly rounded quotient takes three:
//     q0 = n * r;  e = fma(-q0, a, n);  q = fma(e, r, q0)          with r = RN(1/a) from a table
// (Markstein's correction step: e is the exact remainder).  tests/divide_small_int_check.c compares
// this chain with n / a for ALL 2^32 fp32 values of n and a = 1..6 on the CPU: bit-identical except
// for a = 6 with |n| < 2^-125; a wave-uniform test sends any wavefront holding |n| < 2^-90 down the
// IEEE path instead (never taken in practice; 2^-90 also keeps e out of the denormal range, so the
// result does not depend on the kernel's denormal mode).  v_div_fixup_f32 — the last instruction of
// the IEEE sequence, with the same operands — supplies the IEEE results for n = +-0, inf, NaN and
// a = 0 (inf / NaN), bit for bit what the division gives.
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
// the four quotients of a lane: n / aii, aii = bytes of m, (aii, RN(1 / aii)) = d
__device__ __forceinline__ float4 canon_div4(float4 n, uint32_t m, const DivPairs& d) {
    const float tiny = fminf(fminf(fabsf(n.x), fabsf(n.y)), fminf(fabsf(n.z), fabsf(n.w)));
    float4 o;
    if (__builtin_amdgcn_ballot_w64(tiny < 0x1p-90f) == 0ull) {
        o.x = div_small_int(n.x, d.c[0]);
        o.y = div_small_int(n.y, d.c[1]);
        o.z = div_small_int(n.z, d.c[2]);
        o.w = div_small_int(n.w, d.c[3]);
    } else {
        o.x = n.x / (float)(m & 0xFFu);
        o.y = n.y / (float)((m >> 8) & 0xFFu);
        o.z = n.z / (float)((m >> 16) & 0xFFu);
        o.w = n.w / (float)(m >> 24);
    }
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

constexpr int FUSED_WAVES = 16;
constexpr int FUSED_THREADS = FUSED_WAVES * 64;
constexpr int FUSED_PAD = 4;  // floats of padding on each side of an LDS row (keeps rows 16-B aligned)

__host__ __device__ inline int fused_row_floats(int nt) { return nt * 256 + 2 * FUSED_PAD; }
__host__ __device__ inline size_t fused_lds_bytes(int nt) {
    const int r = FUSED_WAVES / nt;
    return (size_t)2 /*buffers*/ * 2 /*J,S*/ * r * fused_row_floats(nt) * sizeof(float) +
           128 /* DivEntry table */;
}

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
//   xwin0               (k12_canon2<NT, true> only) global x of the workgroups' first cell: the launch
//                       covers the window [xwin0, xwin0 + NT*256) of every row instead of the whole row.
//                       Valid when every water cell of the grid lies inside the window: the columns
//                       just outside it then hold non-water constants, the same in every iterate, which
//                       the edge lanes load into the pad cells of the LDS rows.
//   xcd_rows, xcd_nz    (experiment) > 0: the launch is a 1-D grid; workgroup L runs on XCD L % 8 (the
//                       dispatcher deals workgroups to the 8 XCDs in turn) and takes tile L / 8 of that
//                       XCD's own list — a band of row tiles x all z chunks — so that workgroups which
//                       share halo rows share an L2.  xcd_rows = row tiles in the launch, xcd_nz = chunks.
struct FusedRange {
    int zout_lo, zout_hi, jlo, jhi, mlo, mhi, ytile0, hole_lo, hole_hi, nz_lo, xwin0, xcd_rows, xcd_nz;
};

// Per-wavefront state of the z march.  Everything rotates with period 4 (the z loop is unrolled by
// 4, so ring indices are compile-time constants and the rotation costs no register moves):
//   j[4]  iterate j   : slots (i, i+1, i+2) = planes zc-1, zc, zc+1; slot i+3 receives plane zc+2
//   s[4]  iterate j+1 : slots (i, i+1) = planes zc-2, zc-1; slot i+2 receives plane zc
//   b[4], m[4]        : slots (i+1, i+2) = planes zc-1 (stage 2), zc (stage 1); slot i+3 receives zc+1
//   h[2]  halo row    : slot i&1 = plane zc; the other receives plane zc+1   (halo wavefronts only)
//   padv[4] (windowed launches) : iterate j at the column just outside the window, slots (i, i+1) =
//                                  planes zc-1, zc; slot i+2 receives plane zc+1
//   dv[2] (a, 1/a) pairs : slot i&1 = of the mask word of plane zc (stage 1 of this step), fetched before this
//                          step's barrier; the other slot = of plane zc-1, fetched a step ago (stage 2)
struct FusedState {
    float4 j[4], s[4], b[4], h[2];
    uint32_t m[4];
    float padv[4];
    DivPairs dv[2];
};

template <int NT>
struct FusedCtx {
    static constexpr int R = FUSED_WAVES / NT;
    static constexpr int RW = NT * 256 + 2 * FUSED_PAD;
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
    int lane, rr, x0, xe;
    unsigned boff, boff_h;
    // windowed launches: byte offset of this lane's pad column (left for lanes < 32, right otherwise) in
    // its row, whether that cell exists, whether this lane stores a pad cell and where (LDS x index)
    unsigned boff_pad;
    bool pad_in, pad_writer;
    int pad_x;
    float p_oob;
    bool row_in, halo_in, halo_lo, halo_hi, is_out_row;
    bool wave_clean;     // every lane of this wavefront lies inside the grid (wave-uniform)
    uint32_t lane_mask;  // ~0 for lanes inside the grid

    __device__ __forceinline__ FLUID_LDS float* row_ptr(int buf, int arr, int row) const {
        return lds + ((buf * 2 + arr) * R + row) * RW + FUSED_PAD;
    }
    __device__ __forceinline__ bool j_ok(int lz) const { return lz >= jlo && lz < jhi; }
    __device__ __forceinline__ bool m_ok(int lz) const { return lz >= mlo && lz < mhi; }
    // element offset of plane lz, redirected to plane 0 when lz is outside the grid: loads are always
    // issued, from a valid address, and the value is replaced later (fix_*).  A branch or a select
    // directly on a load makes hipcc wait for it (vmcnt(0)) on the spot.
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
    __device__ __forceinline__ float fix_pad(float v, int lz) const {
        return (pad_in && j_ok(lz)) ? v : p_oob;
    }
    __device__ __forceinline__ uint32_t fix_m(uint32_t m, int lz) const {
        if (wave_clean && m_ok(lz)) return m;
        return m_ok(lz) ? ((m & lane_mask) | (MASK_DRY4 & ~lane_mask)) : MASK_DRY4;
    }
};

// One plane step: I = ring phase (k mod 4), zc = plane of iterate j+1 formed in this step.
template <int NT, int I, bool WIN>
__device__ __forceinline__ void fused_step(const FusedCtx<NT>& c, FusedState& st, int zc FLUID_TRACE_ARG) {
    constexpr int R = FusedCtx<NT>::R;
    constexpr int buf = I & 1;
    float4& jm = st.j[I & 3];
    float4& jc = st.j[(I + 1) & 3];
    float4& jn = st.j[(I + 2) & 3];   // raw from the previous step's load until fixed up below
    float4& s_mm = st.s[I & 3];
    float4& s_m = st.s[(I + 1) & 3];
    float4& s_c = st.s[(I + 2) & 3];
    const float4 b_m = st.b[(I + 1) & 3], b_c = st.b[(I + 2) & 3];
    const uint32_t m_m = st.m[(I + 1) & 3];

    FT_BEGIN();
    // ---- loads the next step needs (raw; fixed up at the end of this step)
    const int64_t o1 = c.j_off(zc + 1), o2 = c.j_off(zc + 2), a1 = c.m_off(zc + 1);
    st.j[(I + 3) & 3] = ld_f4(c.pin + o2, c.boff);
    st.h[(I + 1) & 1] = ld_f4(c.pin + o1, c.boff_h);
    st.b[(I + 3) & 3] = ld_f4(c.rhs + a1, c.boff);
    st.m[(I + 3) & 3] = ld_u32(c.mask + a1, c.boff >> 2);
    if (WIN)
        st.padv[(I + 2) & 3] =
            *reinterpret_cast<const float*>(reinterpret_cast<const char*>(c.pin + o1) + c.boff_pad);

    // ---- what the previous step loaded, fixed up (only wavefronts / planes on the grid boundary do
    // anything here; those loads have had a whole step to land): known BEFORE the barrier, so the
    // decisions and the table look-ups that depend on the mask word are taken off the path behind it
    jn = c.fix_j(jn, c.row_in, zc + 1);
    const float4 hc = c.fix_j(st.h[I & 1], c.halo_in, zc);
    const uint32_t m_c = c.fix_m(st.m[(I + 2) & 3], zc);
    st.m[(I + 2) & 3] = m_c;
    const int zo = zc - 1;
    const bool do1 = __builtin_amdgcn_ballot_w64(mask_any_water(m_c)) != 0ull;            // wave-uniform
    const bool wet = c.is_out_row && mask_any_water(m_m);
    const bool do2 = zo >= c.zb && zo < c.ze && c.rr >= 1 && c.rr <= R - 2 &&            // wave-uniform
                     __builtin_amdgcn_ballot_w64(wet) != 0ull;
    FT(0);  // issue of the loads + wait for the previous step's
    if (do1) st.dv[I & 1] = div_pairs(m_c, c.divtab);  // stage 2 of the next step uses them again
    const DivPairs& d_c = st.dv[I & 1];
    const DivPairs& d_m = st.dv[(I + 1) & 1];

    // ---- publish this row: iterate j at plane zc, iterate j+1 at plane zc-1
    FLUID_LDS float* jrow = c.row_ptr(buf, 0, c.rr);
    FLUID_LDS float* srow = c.row_ptr(buf, 1, c.rr);
    lds_st4(jrow + c.x0, jc);
    lds_st4(srow + c.x0, s_m);
    if (WIN) {
        // the columns next to the window hold non-water constants: the same value in both iterates
        if (c.pad_writer) {
            jrow[c.pad_x] = c.fix_pad(st.padv[(I + 1) & 3], zc);
            srow[c.pad_x] = c.fix_pad(st.padv[I & 3], zc - 1);
        }
    }
    FT(1);  // publish
    __syncthreads();
    FT(2);  // barrier

    // ---- the neighbour rows of both stages in ONE round trip to LDS
    float4 jym = hc, jyp = hc, sym = s_m, syp = s_m;
    float je = 0.f, se = 0.f;
    if (do1) {
        if (!c.halo_lo) jym = lds_ld4(c.row_ptr(buf, 0, c.rr - 1) + c.x0);
        if (!c.halo_hi) jyp = lds_ld4(c.row_ptr(buf, 0, c.rr + 1) + c.x0);
        je = jrow[c.xe];
    }
    if (do2) {
        sym = lds_ld4(c.row_ptr(buf, 1, c.rr - 1) + c.x0);
        syp = lds_ld4(c.row_ptr(buf, 1, c.rr + 1) + c.x0);
        se = srow[c.xe];
    }

    // ---- stage 1: iterate j+1 at plane zc for this row
    s_c = jc;  // non-water (and out-of-grid) cells keep their constant
    if (do1) {
        const float left = from_lane_below(jc.w, je, c.lane);
        const float right = from_lane_above(jc.x, je, c.lane);
        float4 n;
        n.x = canon_num(b_c.x, jc.y, jyp.x, jn.x, left, jym.x, jm.x);
        n.y = canon_num(b_c.y, jc.z, jyp.y, jn.y, jc.x, jym.y, jm.y);
        n.z = canon_num(b_c.z, jc.w, jyp.z, jn.z, jc.y, jym.z, jm.z);
        n.w = canon_num(b_c.w, right, jyp.w, jn.w, jc.z, jym.w, jm.w);
        const float4 o = canon_div4(n, m_c, d_c);
        s_c.x = mask_is_water(m_c, 0) ? o.x : jc.x;
        s_c.y = mask_is_water(m_c, 1) ? o.y : jc.y;
        s_c.z = mask_is_water(m_c, 2) ? o.z : jc.z;
        s_c.w = mask_is_water(m_c, 3) ? o.w : jc.w;
    }
#ifdef FLUID_FUSED_TRACE
    asm volatile("" ::"v"(s_c.x), "v"(s_c.y), "v"(s_c.z), "v"(s_c.w));  // stage 1 is done here
#endif
    FT(3);  // LDS round trip + stage 1

    // ---- stage 2: iterate j+2 at plane zc-1 from iterate j+1 at planes zc-2, zc-1, zc
    if (do2) {
        const float left = from_lane_below(s_m.w, se, c.lane);
        const float right = from_lane_above(s_m.x, se, c.lane);
        float4 n;
        n.x = canon_num(b_m.x, s_m.y, syp.x, s_c.x, left, sym.x, s_mm.x);
        n.y = canon_num(b_m.y, s_m.z, syp.y, s_c.y, s_m.x, sym.y, s_mm.y);
        n.z = canon_num(b_m.z, s_m.w, syp.z, s_c.z, s_m.y, sym.z, s_mm.z);
        n.w = canon_num(b_m.w, right, syp.w, s_c.w, s_m.z, sym.w, s_mm.w);
        float4 o = canon_div4(n, m_m, d_m);
        o.x = mask_is_water(m_m, 0) ? o.x : s_m.x;
        o.y = mask_is_water(m_m, 1) ? o.y : s_m.y;
        o.z = mask_is_water(m_m, 2) ? o.z : s_m.z;
        o.w = mask_is_water(m_m, 3) ? o.w : s_m.w;
        if (wet) {
            const int64_t oo = (int64_t)zo * c.plane;
            st_f4(c.pout + oo, c.boff, o);
            if (c.pmid)  // the odd iterate, kept only by the last pair of a loop
                st_f4(c.pmid + oo, c.boff, s_m);
        }
    }
    FT(4);  // stage 2 + stores
}

template <int NT, bool WIN>
__global__ void __launch_bounds__(FUSED_THREADS)
k12_canon2(const uint8_t* __restrict__ mask, const float* __restrict__ rhs,
           const float* __restrict__ pin, float* __restrict__ pout, float* __restrict__ pmid,
           const uint8_t* __restrict__ active, BrickK bk, GridK g, float p_air, int zchunk,
           FusedRange rg) {
    constexpr int R = FUSED_WAVES / NT;   // rows of iterate j+1 per workgroup
    constexpr int TY = R - 2;             // output rows per workgroup
    constexpr int RW = NT * 256 + 2 * FUSED_PAD;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    FusedCtx<NT> c;
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
    c.lane = threadIdx.x & 63;
    // readfirstlane: tells hipcc the wave index (hence row, tile and halo role) is wave-uniform, so it
    // lives in SGPRs and role tests become scalar branches instead of exec masking
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int tx = wave % NT;
    c.rr = wave / NT;
    c.x0 = tx * 256 + c.lane * 4;
    int tile_y = (int)blockIdx.y, tile_z = (int)blockIdx.z;
    if (rg.xcd_rows > 0) {
        const int L = (int)blockIdx.x, xcd = L & 7, slot = L >> 3;
        // band of XCD `xcd`: rows [r0, r0 + nr) of the launch's row tiles (the first xcd_rows % 8 bands get
        // one more)
        const int base = rg.xcd_rows >> 3, extra = rg.xcd_rows & 7;
        const int nr = base + (xcd < extra ? 1 : 0);
        const int r0 = xcd * base + min(xcd, extra);
        if (slot >= nr * rg.xcd_nz) return;  // padding of the shorter lists (uniform)
        tile_z = slot / nr;
        tile_y = r0 + slot - tile_z * nr;
    }
    const int y0 = (tile_y + rg.ytile0) * TY;  // first output row
    const int y = y0 - 1 + c.rr;           // this wavefront's row
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
    c.row_in = xin && (unsigned)y < (unsigned)g.H;   // this lane's cells exist
    c.is_out_row = c.rr >= 1 && c.rr <= R - 2 && c.row_in;
    c.halo_lo = c.rr == 0;
    c.halo_hi = c.rr == R - 1;
    const int yh = c.halo_lo ? y - 1 : y + 1;        // outer neighbour row of a halo wavefront
    const bool is_halo = c.halo_lo || c.halo_hi;
    c.halo_in = is_halo && xin && (unsigned)yh < (unsigned)g.H;
    c.wave_clean = __builtin_amdgcn_ballot_w64(!c.row_in || (is_halo && !c.halo_in)) == 0ull;
    c.lane_mask = c.row_in ? 0xFFFFFFFFu : 0u;
    // in-plane byte offsets (safe addresses for lanes / rows outside the grid)
    const unsigned xs = xin ? (unsigned)gx0 : 0u;
    c.boff = 4u * (xs + (unsigned)g.W * (unsigned)(((unsigned)y < (unsigned)g.H) ? y : 0));
    c.boff_h = 4u * (xs + (unsigned)g.W * (unsigned)(((unsigned)yh < (unsigned)g.H) ? yh : 0));
    // the cell across the x-tile boundary: lane 0 -> x0-1, lane 63 -> x0+4 (other lanes: harmless)
    c.xe = c.lane == 0 ? c.x0 - 1 : c.x0 + 4;
    if (WIN) {
        const int xl = rg.xwin0 - 1, xr = rg.xwin0 + NT * 256;  // the columns next to the window
        const int gxp = c.lane < 32 ? xl : xr;
        const bool col_in = (unsigned)gxp < (unsigned)g.W;
        c.pad_in = col_in && (unsigned)y < (unsigned)g.H;
        c.boff_pad = 4u * ((col_in ? (unsigned)gxp : 0u) +
                           (unsigned)g.W * (unsigned)(((unsigned)y < (unsigned)g.H) ? y : 0));
        c.pad_writer = (c.lane == 0 && tx == 0) || (c.lane == 63 && tx == NT - 1);
        c.pad_x = c.lane == 0 ? -1 : NT * 256;
    }

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
    for (int i = threadIdx.x; i < 2 * 2 * R * 2 * FUSED_PAD; i += FUSED_THREADS) {
        const int side = i % (2 * FUSED_PAD), row = i / (2 * FUSED_PAD);
        FLUID_LDS float* base = c.lds + row * RW;
        base[side < FUSED_PAD ? side : RW - 2 * FUSED_PAD + side] = p_air;
    }

    __syncthreads();  // the table is read before the first barrier of the march (div_pairs)

    // prologue: the state a step with ring phase 0 and zc = zb - 1 expects
    const float4 pa4 = make_float4(p_air, p_air, p_air, p_air);
    FusedState st;
    int zc = c.zb - 1;  // plane of iterate j+1 formed in the coming step
    st.j[0] = c.fix_j(ld_f4(pin + c.j_off(zc - 1), c.boff), c.row_in, zc - 1);
    st.j[1] = c.fix_j(ld_f4(pin + c.j_off(zc), c.boff), c.row_in, zc);
    st.j[2] = ld_f4(pin + c.j_off(zc + 1), c.boff);  // raw: fixed up by the first step
    st.h[0] = ld_f4(pin + c.j_off(zc), c.boff_h);
    st.s[0] = pa4;
    st.s[1] = pa4;
    st.b[1] = make_float4(0.f, 0.f, 0.f, 0.f);
    st.m[1] = MASK_DRY4;
    st.b[2] = ld_f4(rhs + c.m_off(zc), c.boff);
    st.m[2] = ld_u32(mask + c.m_off(zc), c.boff >> 2);
    for (int i = 0; i < 4; i++) st.dv[0].c[i] = st.dv[1].c[i] = make_float2(0.f, 0.f);
    if (WIN) {
        auto pad_at = [&](int lz) {
            return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(pin + c.j_off(lz)) +
                                                   c.boff_pad);
        };
        st.padv[0] = pad_at(zc - 1);
        st.padv[1] = pad_at(zc);
    }

    const int steps = c.ze - c.zb + 2;  // iterate j+1 at planes zb-1 .. ze, iterate j+2 one behind
#ifdef FLUID_FUSED_TRACE
    FusedTrace ftr;
    for (int i = 0; i < FUSED_TRACE_PHASES; i++) ftr.sum[i] = 0;
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
#endif
    for (int k = 0; k < steps; k += 4, zc += 4) {
        fused_step<NT, 0, WIN>(c, st, zc FLUID_TRACE_PASS);
        if (k + 1 >= steps) break;  // all wave-uniform: every wavefront takes the same barriers
        fused_step<NT, 1, WIN>(c, st, zc + 1 FLUID_TRACE_PASS);
        if (k + 2 >= steps) break;
        fused_step<NT, 2, WIN>(c, st, zc + 2 FLUID_TRACE_PASS);
        if (k + 3 >= steps) break;
        fused_step<NT, 3, WIN>(c, st, zc + 3 FLUID_TRACE_PASS);
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
