// pressure_sweep.hip — translation unit of the single-sweep 12_solve_pressure kernels and the loop
// section's import / export / prepare passes (kernels_pressure.h) with their launchers.
#include "kernels_pressure.h"
#include "pressure_api.h"

#include <algorithm>

namespace fluid {

static dim3 cell_block() { return dim3(64, 4, 1); }
static dim3 cell_grid(const GridK& g, int planes) {
    return dim3((g.W + 63) / 64, (g.H + 3) / 4, planes);
}
static BrickK bricks_of(const GridK& g) {
    BrickK b;
    k12_brick_dims(g.W, g.H, g.Dl, b.nbx, b.nby, b.nbz);
    return b;
}

void k12_brick_dims(int W, int H, int Dl, int& nbx, int& nby, int& nbz) {
    nbx = (W + BRICK_X - 1) / BRICK_X;
    nby = (H + BRICK_Y - 1) / BRICK_Y;
    nbz = (Dl + BRICK_Z - 1) / BRICK_Z;
}

void k12_brick_cells(int& bx, int& by, int& bz) {
    bx = BRICK_X;
    by = BRICK_Y;
    bz = BRICK_Z;
}

void k12_launch_plain(hipStream_t s, const uint8_t* t, const float* div, const float* pin,
                      float* pout, const GridK& g, const ParamsK& p) {
    hipLaunchKernelGGL(k12_plain, cell_grid(g, g.Dl), cell_block(), 0, s, t, div, pin, pout, g, p);
}

// z chunk such that the launch has a few thousand workgroups (>> 256 CUs)
static int pick_zchunk(int nz, int64_t blocks_per_plane_row, int zmin, int64_t target) {
    int zchunk = nz;
    while (zchunk > zmin && blocks_per_plane_row * ((nz + zchunk - 1) / zchunk) < target)
        zchunk = (zchunk + 1) / 2;
    return zchunk;
}

void k12_launch_zmarch(hipStream_t s, int ry, const uint8_t* t, const float* div, const float* pin,
                       float* pout, const GridK& g, const ParamsK& p) {
    const int bx = (g.W + 255) / 256, by = (g.H + 4 * ry - 1) / (4 * ry);
    const int zchunk = pick_zchunk(g.Dl, (int64_t)bx * by, 16, 2048);
    const dim3 grid(bx, by, (g.Dl + zchunk - 1) / zchunk);
    if (ry == 4)
        hipLaunchKernelGGL(k12_zmarch<4>, grid, dim3(256), 0, s, t, div, pin, pout, g, p, zchunk);
    else if (ry == 1)
        hipLaunchKernelGGL(k12_zmarch<1>, grid, dim3(256), 0, s, t, div, pin, pout, g, p, zchunk);
    else
        hipLaunchKernelGGL(k12_zmarch<2>, grid, dim3(256), 0, s, t, div, pin, pout, g, p, zchunk);
}

void k12_launch_prepare(hipStream_t s, const uint8_t* t, const float* div, uint8_t* mask, float* rhs,
                        uint8_t* bricks, const GridK& g, const ParamsK& p, bool do_mask,
                        bool do_rhs) {
    hipLaunchKernelGGL(k12_prepare, cell_grid(g, g.Dl), cell_block(), 0, s, t, div, mask, rhs,
                       bricks, bricks_of(g), g, p, do_mask ? 1 : 0, do_rhs ? 1 : 0);
}

void k12_launch_count_bricks(hipStream_t s, const uint8_t* bricks, const GridK& g, uint32_t* out,
                             const uint32_t* x_extent) {
    BrickK bk;
    k12_brick_dims(g.W, g.H, g.Dl, bk.nbx, bk.nby, bk.nbz);
    hipLaunchKernelGGL(k12_count_bricks, dim3(1), dim3(256), 0, s, bricks, bk, out, x_extent, g.W);
}

void k12_launch_import(hipStream_t s, const uint8_t* t, const float* pimg, float* work,
                       const GridK& g, const ParamsK& p, int lz0, int nplanes) {
    hipLaunchKernelGGL(k12_import, cell_grid(g, nplanes), cell_block(), 0, s, t, pimg, work, g, p,
                       lz0);
}

void k12_launch_background(hipStream_t s, const uint8_t* t, float* work, const GridK& g,
                           const ParamsK& p, int lz0, int nplanes) {
    hipLaunchKernelGGL(k12_background, cell_grid(g, nplanes), cell_block(), 0, s, t, work, g, p,
                       lz0);
}

void k12_launch_export(hipStream_t s, const uint8_t* t, const float* w_even, const float* w_odd,
                       float* p1, float* p2, const GridK& g, const ParamsK& p) {
    hipLaunchKernelGGL(k12_export, cell_grid(g, g.Dl), cell_block(), 0, s, t, w_even, w_odd, p1, p2,
                       g, p);
}

void k12_launch_canon(hipStream_t s, int ry, const uint8_t* mask, const float* rhs,
                      const float* pin, float* pout, const uint8_t* bricks, const GridK& g,
                      float p_oob, int zlo, int zhi) {
    const int bx = (g.W + 255) / 256, by = (g.H + 4 * ry - 1) / (4 * ry);
    const int nz = zhi - zlo;
    const int zchunk = pick_zchunk(nz, (int64_t)bx * by, 16, 2048);
    const dim3 grid(bx, by, (nz + zchunk - 1) / zchunk);
    const BrickK bk = bricks_of(g);
    if (ry == 4)
        hipLaunchKernelGGL(k12_canon<4>, grid, dim3(256), 0, s, mask, rhs, pin, pout, bricks, bk, g,
                           p_oob, zchunk, zlo, zhi);
    else if (ry == 2)
        hipLaunchKernelGGL(k12_canon<2>, grid, dim3(256), 0, s, mask, rhs, pin, pout, bricks, bk, g,
                           p_oob, zchunk, zlo, zhi);
    else
        hipLaunchKernelGGL(k12_canon<1>, grid, dim3(256), 0, s, mask, rhs, pin, pout, bricks, bk, g,
                           p_oob, zchunk, zlo, zhi);
}

}  // namespace fluid
