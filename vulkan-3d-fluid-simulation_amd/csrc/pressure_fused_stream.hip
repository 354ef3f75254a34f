// pressure_fused_stream.hip — translation unit of the two-sweeps-per-pass kernels with streaming stores
// (kernels_pressure_fused.h: st_f4<true>), which pressure_fused_launch.h picks for launches whose working set
// is several times the memory-side cache.  Separate from pressure_fused.hip so that the kernels there keep
// their instruction schedule.
#include "pressure_fused_launch.h"

namespace fluid {

template <int NT, int RG, bool KEEP>
hipError_t k12_launch_streaming(const FusedLaunchArgs& a) {
    using G = FusedGeom<NT, RG>;
    static std::atomic<bool> attr_set[64] = {};  // the dynamic-LDS limit: once per instantiation and device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        const hipError_t e =
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k12_canon2<NT, false, RG, KEEP, false, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)a.lds);
        if (e != hipSuccess) return e;
        attr_set[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((k12_canon2<NT, false, RG, KEEP, false, true>), a.grid, dim3(G::THREADS), a.lds, a.stream,
                       a.mask, a.rhs, a.pin, a.pout, a.pmid, a.bricks, a.bk, a.g, a.p_oob, a.zchunk, a.r, a.omega);
    return hipSuccess;
}

template hipError_t k12_launch_streaming<2, 2, false>(const FusedLaunchArgs&);
template hipError_t k12_launch_streaming<2, 2, true>(const FusedLaunchArgs&);
template hipError_t k12_launch_streaming<2, 3, false>(const FusedLaunchArgs&);
template hipError_t k12_launch_streaming<2, 3, true>(const FusedLaunchArgs&);
template hipError_t k12_launch_streaming<4, 2, false>(const FusedLaunchArgs&);
template hipError_t k12_launch_streaming<4, 2, true>(const FusedLaunchArgs&);
template hipError_t k12_launch_streaming<4, 3, false>(const FusedLaunchArgs&);
template hipError_t k12_launch_streaming<4, 3, true>(const FusedLaunchArgs&);

}  // namespace fluid
