// pressure_fused3.hip — translation unit of the three-sweeps-per-pass kernel (kernels_pressure_fused3.h) and
// of its launcher.  Separate from pressure_fused.hip so that the kernels there keep their instruction schedule.
#include "pressure_fused_launch.h"

namespace fluid {

// Grids up to 512 cells wide: NT <= 2 x tiles leave 8 (NT = 2) or 20 / 12 (NT = 1) output rows per workgroup.
// At NT = 4 the workgroup's 6 rows would leave 2: those grids stay with two sweeps per pass.
// The kernel addresses a z chunk's planes through buffer resources with 32-bit offsets below 2 GB (140 planes
// of W x H floats at most): H <= 4096.
bool k12_canon3_supports(const GridK& g) { return g.W % 4 == 0 && g.W <= 512 && g.H <= 4096 && g.Dl >= 3; }

// Is a launch of three sweeps the better shape for this box of water?  One x tile (a narrow grid, or a window
// around the water): yes — the thin three-sweep kernel exists.  Two tiles: only the fat shape exists (three rows on
// 8 wavefronts), which a launch over a small box loses with (few workgroups, each slow): the 512^3 dam break once
// the water has spread over more than 256 columns ran its loop 7 % slower than with pairs on 16 thin wavefronts.
bool k12_canon3_suits(const GridK& g, const ActiveBox& box) {
    if (!box.valid) return true;
    const int nt = (g.W + 255) / 256;
    if (nt == 1) return true;
    if (box.x_hi > box.x_lo) {
        const int x0 = box.x_lo & ~31;
        if ((box.x_hi - x0 + 255) / 256 == 1) return true;
    }
    FusedRange rg{};
    rg.zout_lo = 0;
    rg.zout_hi = g.Dl;
    return !small_box_launch(g, rg, box, FusedGeomT<2, 3, 3>::TY);
}

template <int NT, int RG, bool KEEP>
hipError_t k12_launch_streaming3(const FusedLaunchArgs& a) {
    using G = FusedGeomT<NT, RG, 3>;
    static std::atomic<bool> attr_set[64] = {};  // the dynamic-LDS limit: once per instantiation and device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        const hipError_t e =
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k12_canon_t<NT, false, RG, 3, KEEP, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)a.lds);
        if (e != hipSuccess) return e;
        attr_set[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((k12_canon_t<NT, false, RG, 3, KEEP, true>), a.grid, dim3(G::THREADS), a.lds, a.stream, a.mask,
                       a.rhs, a.pin, a.pout, a.pmid, a.bricks, a.bk, a.g, a.p_oob, a.zchunk, a.r);
    return hipSuccess;
}
template hipError_t k12_launch_streaming3<2, 3, false>(const FusedLaunchArgs&);
template hipError_t k12_launch_streaming3<2, 3, true>(const FusedLaunchArgs&);

// rows per wavefront of a three-sweep launch: three on 8 wavefronts (R = 24 / NT rows per workgroup), or — one
// x tile only — one on 16 (R = 16): launches shaped to a small box of water, as for two sweeps
template <int NT, bool WIN>
static hipError_t launch3(hipStream_t s, const uint8_t* mask, const float* rhs, const float* pin, float* pout,
                          float* pmid, const uint8_t* bricks, const GridK& g, float p_oob, const FusedRange& rg,
                          const ActiveBox& box, int part, int part_lo, int part_hi) {
    if constexpr (NT == 1) {
        static const int forced = [] {
            const char* e = getenv("FLUID_FUSED_RG");
            return e ? atoi(e) : 0;
        }();
        // (windowed launches: always — the windowed three-row kernel does not fit its 256 registers, 92 to 140 bytes
        // of scratch per lane)
        const bool thin = WIN || (forced ? forced == 1 : small_box_launch(g, rg, box, FusedGeomT<1, 3, 3>::TY));
        if (thin)
            return launch_nt<1, WIN, 1, 3>(s, mask, rhs, pin, pout, pmid, bricks, g, p_oob, rg, box, part, part_lo,
                                           part_hi);
    }
    return launch_nt<NT, WIN, 3, 3>(s, mask, rhs, pin, pout, pmid, bricks, g, p_oob, rg, box, part, part_lo, part_hi);
}

hipError_t k12_launch_canon3(hipStream_t s, const uint8_t* mask, const float* rhs, const float* pin, float* pout,
                             float* pmid, const uint8_t* bricks, const GridK& g, float p_oob, int halo_lo,
                             int halo_hi, int aux_lo, int aux_hi, const ActiveBox& box, int part, int part_lo,
                             int part_hi) {
    // halo_lo / halo_hi: valid ghost planes of the input below / above the owned planes (0 at a domain face);
    // aux_*: the same for mask and b_i.  A launch consumes three planes of halo (two of mask / b_i).
    FusedRange rg;
    rg.jlo = -halo_lo;
    rg.jhi = g.Dl + halo_hi;
    rg.mlo = -aux_lo;
    rg.mhi = g.Dl + aux_hi;
    rg.zout_lo = -std::max(0, std::min(halo_lo - 3, aux_lo - 2));
    rg.zout_hi = g.Dl + std::max(0, std::min(halo_hi - 3, aux_hi - 2));
    rg.ytile0 = 0;
    rg.hole_lo = rg.hole_hi = rg.zout_hi;
    rg.nz_lo = 0;
    rg.xwin0 = 0;
    rg.xcd_rows = 0;
    const int nt = (g.W + 255) / 256;
    // sparse scene: an x window of one or two 256-cell columns around the water (pressure_fused.hip)
    if (box.valid && box.x_hi > box.x_lo) {
        const int x0 = box.x_lo & ~31;  // 128-byte aligned rows
        const int ntw = (box.x_hi - x0 + 255) / 256;
        if (ntw < nt && ntw == 1) {
            rg.xwin0 = x0;
            return launch3<1, true>(s, mask, rhs, pin, pout, pmid, bricks, g, p_oob, rg, box, part, part_lo, part_hi);
        }
    }
    if (nt == 1) return launch3<1, false>(s, mask, rhs, pin, pout, pmid, bricks, g, p_oob, rg, box, part, part_lo, part_hi);
    if (nt == 2) return launch3<2, false>(s, mask, rhs, pin, pout, pmid, bricks, g, p_oob, rg, box, part, part_lo, part_hi);
    return hipErrorInvalidValue;
}

}  // namespace fluid

#ifdef FLUID_FUSED_TRACE
// dev build: copy the phase sums of the last three-sweep launch out (tools/fused_trace3.py)
extern "C" int fluid_dev_fused_trace3(unsigned long long* out, int words) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(fluid::g_fused_trace3), sizeof(unsigned long long) * words);
}
#endif
