// pressure_fused.hip — translation unit of the two-sweeps-per-pass kernel (kernels_pressure_fused.h).
#include "pressure_fused_launch.h"

namespace fluid {

bool k12_canon2_supports(const GridK& g) { return g.W % 4 == 0 && g.W <= 1024 && g.Dl >= 2; }

hipError_t k12_launch_canon2(hipStream_t s, const uint8_t* mask, const float* rhs, const float* pin,
                             float* pout, float* pmid, const uint8_t* bricks, const GridK& g,
                             float p_oob, int halo_lo, int halo_hi, int aux_lo, int aux_hi,
                             const ActiveBox& box, int part, int part_lo, int part_hi) {
    // halo_lo / halo_hi: valid ghost planes of the input below / above the owned planes (0 at a
    // domain face); aux_*: the same for mask and b_i.  A launch consumes two planes of halo.
    FusedRange rg;
    rg.jlo = -halo_lo;
    rg.jhi = g.Dl + halo_hi;
    rg.mlo = -aux_lo;
    rg.mhi = g.Dl + aux_hi;
    rg.zout_lo = -std::max(0, std::min(halo_lo - 2, aux_lo - 1));
    rg.zout_hi = g.Dl + std::max(0, std::min(halo_hi - 2, aux_hi - 1));
    rg.ytile0 = 0;
    rg.hole_lo = rg.hole_hi = rg.zout_hi;
    rg.nz_lo = 0;
    rg.xwin0 = 0;
    rg.xcd_rows = 0;
    const int nt = (g.W + 255) / 256;
    // Sparse scene: if the bricks that hold water (on a Z slab: here and in the neighbouring slabs) span one or two 256-cell columns of
    // a wider grid, launch over that x window only (fewer lanes, more rows per workgroup).
    if (box.valid && box.x_hi > box.x_lo) {
        const int x0 = box.x_lo & ~31;  // 128-byte aligned rows
        const int ntw = (box.x_hi - x0 + 255) / 256;
        if ((ntw == 1 || ntw == 2) && ntw < nt) {
            rg.xwin0 = x0;
            return k12_launch_canon2_win(s, ntw, mask, rhs, pin, pout, pmid, bricks, g, p_oob, rg, box,
                                         part, part_lo, part_hi);
        }
    }
    if (nt == 1) return launch_rg<1, false>(s, mask, rhs, pin, pout, pmid, bricks, g, p_oob, rg, box, part,
                                     part_lo, part_hi);
    if (nt == 2) return launch_rg<2, false>(s, mask, rhs, pin, pout, pmid, bricks, g, p_oob, rg, box, part,
                                     part_lo, part_hi);
    if (nt <= 4) return launch_rg<4, false>(s, mask, rhs, pin, pout, pmid, bricks, g, p_oob, rg, box, part,
                                     part_lo, part_hi);
    return hipErrorInvalidValue;
}

}  // namespace fluid

namespace fluid {

// One red-black SOR iteration (both colours) in one pass: work[src] -> work[dst].  Whole-grid contexts.
hipError_t k12_launch_canon2_sor(hipStream_t s, const uint8_t* mask, const float* rhs, const float* pin,
                                 float* pout, const uint8_t* bricks, const GridK& g, float p_oob,
                                 const ActiveBox& box, float omega) {
    FusedRange rg;
    rg.jlo = 0;
    rg.jhi = g.Dl;
    rg.mlo = 0;
    rg.mhi = g.Dl;
    rg.zout_lo = 0;
    rg.zout_hi = g.Dl;
    rg.ytile0 = 0;
    rg.hole_lo = rg.hole_hi = rg.zout_hi;
    rg.nz_lo = 0;
    rg.xwin0 = 0;
    rg.xcd_rows = 0;
    const int nt = (g.W + 255) / 256;
    // the instantiations the Jacobi loop defaults to for full rows (pressure_fused_launch.h)
    if (nt == 1)
        return launch_keep<1, false, 3, false, true>(s, mask, rhs, pin, pout, nullptr, bricks, g, p_oob, rg, box,
                                                     FUSED_WHOLE, 0, 0, omega);
    if (nt == 2)
        return launch_keep<2, false, 2, false, true>(s, mask, rhs, pin, pout, nullptr, bricks, g, p_oob, rg, box,
                                                     FUSED_WHOLE, 0, 0, omega);
    if (nt <= 4)
        return launch_keep<4, false, 3, false, true>(s, mask, rhs, pin, pout, nullptr, bricks, g, p_oob, rg, box,
                                                     FUSED_WHOLE, 0, 0, omega);
    return hipErrorInvalidValue;
}

}  // namespace fluid

#ifdef FLUID_FUSED_TRACE
// dev build: copy the phase sums of the last launch out (tools/fused_trace.py)
extern "C" int fluid_dev_fused_trace(unsigned long long* out, int words) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(fluid::g_fused_trace), sizeof(unsigned long long) * words);
}
#endif
