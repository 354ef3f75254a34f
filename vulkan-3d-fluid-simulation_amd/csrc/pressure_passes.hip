// pressure_passes.hip — translation unit of the vectorised once-per-loop passes
// (kernels_pressure_passes.h).
#include "kernels_pressure_passes.h"
#include "pressure_api.h"

#include <algorithm>

namespace fluid {

static dim3 v4_block() { return dim3(64, 4, 1); }
static dim3 v4_grid(const GridK& g, int planes) {
    return dim3((g.W / 4 + 63) / 64, (g.H + 3) / 4, planes);
}
// for the passes that walk g.zl planes per workgroup (FLUID_FOR_PLANES_OF_WORKGROUP)
static dim3 v4_grid_zl(const GridK& g) { return v4_grid(g, (g.Dl + g.zl - 1) / g.zl); }

void k12_launch_prepare_v4(hipStream_t s, const uint8_t* t, const float* div, uint8_t* mask,
                           float* rhs, uint8_t* bricks, const GridK& g, const ParamsK& p,
                           bool do_mask, bool do_rhs, const uint8_t* quiet, uint32_t* x_extent) {
    BrickK bk;
    k12_brick_dims(g.W, g.H, g.Dl, bk.nbx, bk.nby, bk.nbz);
    hipLaunchKernelGGL(k12_prepare_v4, v4_grid_zl(g), v4_block(), 0, s, t, div, mask, rhs, bricks,
                       bk, g, p, do_mask ? 1 : 0, do_rhs ? 1 : 0, quiet, x_extent);
}

void k12_launch_import_v4(hipStream_t s, const uint8_t* t, const float* pimg, float* w0, float* w1,
                          float* w2, const GridK& g, const ParamsK& p, const uint8_t* quiet) {
    BrickK bk;
    k12_brick_dims(g.W, g.H, g.Dl, bk.nbx, bk.nby, bk.nbz);
    hipLaunchKernelGGL(k12_import_v4, v4_grid_zl(g), v4_block(), 0, s, t, pimg, w0, w1, w2, g, p,
                       quiet, bk);
}

void k12_launch_export_v4(hipStream_t s, const uint8_t* t, const float* w_even, const float* w_odd,
                          float* p1, float* p2, const GridK& g, const ParamsK& p, const uint8_t* active) {
    BrickK bk;
    k12_brick_dims(g.W, g.H, g.Dl, bk.nbx, bk.nby, bk.nbz);
    hipLaunchKernelGGL(k12_export_v4, v4_grid_zl(g), v4_block(), 0, s, t, w_even, w_odd, p1, p2, g,
                       p, active, bk);
}

void k12_launch_residual(hipStream_t s, const uint8_t* t, const float* div, const float* pimg,
                         const GridK& g, const ParamsK& p, void* out32) {
    const int64_t rows = (int64_t)g.H * g.Dl;
    const unsigned blocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>((rows + 3) / 4, 4096));
    hipLaunchKernelGGL(k12_residual, dim3(blocks), dim3(256), 0, s, t, div, pimg, g, p,
                       static_cast<ResidualOut*>(out32));
}

void k12_launch_sor_colour(hipStream_t s, const uint8_t* t, const float* div, float* pr, const GridK& g,
                           const ParamsK& p, float omega, int colour) {
    if (g.W % 4 == 0) {
        hipLaunchKernelGGL(k12_sor_colour_v4, v4_grid(g, g.Dl), v4_block(), 0, s, t, div, pr, g, p, omega,
                           colour);
        return;
    }
    const int half = (g.W + 1) / 2;
    hipLaunchKernelGGL(k12_sor_colour, dim3((half + 63) / 64, (g.H + 3) / 4, g.Dl), dim3(64, 4, 1), 0, s, t,
                       div, pr, g, p, omega, colour);
}

}  // namespace fluid
