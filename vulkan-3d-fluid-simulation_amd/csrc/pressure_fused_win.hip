// pressure_fused_win.hip — translation unit of the windowed instantiations of the two-sweeps-per-pass
// kernel (kernels_pressure_fused.h, FusedRange::xwin0): sparse scenes whose water spans one or two
// 256-cell columns of a wider grid.  Separate from pressure_fused.hip so that the whole-row kernels keep
// their instruction schedule.
#include "pressure_fused_launch.h"

namespace fluid {

hipError_t k12_launch_canon2_win(hipStream_t s, int nt_window, const uint8_t* mask, const float* rhs,
                                 const float* pin, float* pout, float* pmid, const uint8_t* bricks,
                                 const GridK& g, float p_oob, const FusedRange& rg,
                                 const ActiveBox& box, int part, int part_lo, int part_hi) {
    if (nt_window == 1)
        return launch_rg<1, true>(s, mask, rhs, pin, pout, pmid, bricks, g, p_oob, rg, box, part,
                                  part_lo, part_hi);
    if (nt_window == 2)
        return launch_rg<2, true>(s, mask, rhs, pin, pout, pmid, bricks, g, p_oob, rg, box, part,
                                  part_lo, part_hi);
    return hipErrorInvalidValue;
}

}  // namespace fluid
