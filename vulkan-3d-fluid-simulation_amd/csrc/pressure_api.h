// pressure_api.h — host-callable launchers of the 12_solve_pressure kernels.  The kernels live in two
// translation units of their own (pressure_sweep.hip, pressure_fused.hip; see pressure_common.h for
// why); engine.hip only sees these declarations.  Every launcher enqueues on `s` and returns.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_common.h"

namespace fluid {

struct BrickK;  // pressure_common.h

// ghost planes per side of the loop's working buffers, mask and b_i (Z-slab contexts exchange up to
// this many planes at a time and then run that many sweeps without communication)
constexpr int LOOP_GHOST = 8;

// activity-brick geometry for a grid (host arithmetic)
void k12_brick_dims(int W, int H, int Dl, int& nbx, int& nby, int& nbz);
void k12_brick_cells(int& bx, int& by, int& bz);  // cells per brick along x, y, z

// single dispatch on the images (any state)
void k12_launch_plain(hipStream_t s, const uint8_t* t, const float* div, const float* pin,
                      float* pout, const GridK& g, const ParamsK& p);
void k12_launch_zmarch(hipStream_t s, int rows_per_wave, const uint8_t* t, const float* div,
                       const float* pin, float* pout, const GridK& g, const ParamsK& p);

// loop section on working buffers
void k12_launch_prepare(hipStream_t s, const uint8_t* t, const float* div, uint8_t* mask, float* rhs,
                        uint8_t* bricks, const GridK& g, const ParamsK& p, bool do_mask, bool do_rhs);
// four-cells-per-thread forms of prepare / import (+ constants of the other buffers) / export, for
// fluid_size.x % 4 == 0 (pressure_passes.hip)
void k12_launch_prepare_v4(hipStream_t s, const uint8_t* t, const float* div, uint8_t* mask,
                           float* rhs, uint8_t* bricks, const GridK& g, const ParamsK& p,
                           bool do_mask, bool do_rhs, const uint8_t* quiet, uint32_t* x_extent);
// x_extent: two zeroed words; a mask pass leaves {W - (lowest x of a water cell, to 4 cells), highest
// x + 1 (to 4 cells)} there (both 0: no water)
void k12_launch_import_v4(hipStream_t s, const uint8_t* t, const float* pimg, float* w0, float* w1,
                          float* w2, const GridK& g, const ParamsK& p, const uint8_t* quiet = nullptr);
// `quiet`: per-brick streaks (quiet_bricks.h) — workgroups in a quiet brick leave at once; null = all
void k12_launch_export_v4(hipStream_t s, const uint8_t* t, const float* w_even, const float* w_odd,
                          float* p1, float* p2, const GridK& g, const ParamsK& p,
                          const uint8_t* active);
// out[0..6] = {bricks with water, y brick range lo, hi, z brick range lo, hi, x CELL range lo, hi (from
// the mask pass's x_extent)}
// one colour (0 / 1) of a red-black SOR iteration in place on a pressure image (opt-in solver)
void k12_launch_sor_colour(hipStream_t s, const uint8_t* t, const float* div, float* pr, const GridK& g,
                           const ParamsK& p, float omega, int colour);
// convergence read-out: out32 = 32 zeroed bytes {uint32 max|r| bits, pad, double sum r^2, uint64 water cells}
void k12_launch_residual(hipStream_t s, const uint8_t* t, const float* div, const float* pimg,
                         const GridK& g, const ParamsK& p, void* out32);
void k12_launch_count_bricks(hipStream_t s, const uint8_t* bricks, const GridK& g, uint32_t* out,
                             const uint32_t* x_extent);
// Where the water is, in cells (whole bricks), as known to the host: rows [y_lo, y_hi) and local planes
// [z_lo, z_hi) hold every brick with water; `fraction` of all bricks hold water.  valid = false: unknown
// (launch over the whole grid).
struct ActiveBox {
    bool valid = false;
    int x_lo = 0, x_hi = 0;  // cells (to 4), not bricks: an x window need not be brick-aligned
    int y_lo = 0, y_hi = 0, z_lo = 0, z_hi = 0;
    float fraction = -1.0f;
};
void k12_launch_import(hipStream_t s, const uint8_t* t, const float* pimg, float* work,
                       const GridK& g, const ParamsK& p, int lz0, int nplanes);
void k12_launch_background(hipStream_t s, const uint8_t* t, float* work, const GridK& g,
                           const ParamsK& p, int lz0, int nplanes);
void k12_launch_export(hipStream_t s, const uint8_t* t, const float* w_even, const float* w_odd,
                       float* p1, float* p2, const GridK& g, const ParamsK& p);
void k12_launch_canon(hipStream_t s, int rows_per_wave, const uint8_t* mask, const float* rhs,
                      const float* pin, float* pout, const uint8_t* bricks, const GridK& g,
                      float p_oob, int zlo, int zhi);
// two sweeps per pass; returns a hipError_t from the one-time LDS attribute call (hipSuccess else)
hipError_t k12_launch_canon2(hipStream_t s, const uint8_t* mask, const float* rhs, const float* pin,
                             float* pout, float* pmid, const uint8_t* bricks, const GridK& g,
                             float p_oob, int halo_lo, int halo_hi, int aux_lo, int aux_hi,
                             const ActiveBox& box, int part = 0, int part_lo = 0, int part_hi = 0);
// `part`: the whole pass, or one of the two launches a Z slab splits it into — FUSED_INTERIOR writes
// the output planes [part_lo, part_hi) (clipped to the pass's output range), FUSED_EDGES the rest
enum { FUSED_WHOLE = 0, FUSED_EDGES = 1, FUSED_INTERIOR = 2 };
bool k12_canon2_supports(const GridK& g);
// three sweeps per pass (kernels_pressure_fused3.h): same arguments; a launch consumes three ghost planes of the
// input and two of mask / b_i per side.  Grids up to 512 cells wide.
hipError_t k12_launch_canon3(hipStream_t s, const uint8_t* mask, const float* rhs, const float* pin,
                             float* pout, float* pmid, const uint8_t* bricks, const GridK& g,
                             float p_oob, int halo_lo, int halo_hi, int aux_lo, int aux_hi,
                             const ActiveBox& box, int part = 0, int part_lo = 0, int part_hi = 0);
bool k12_canon3_supports(const GridK& g);
// ... and is the better launch shape for this box of water (whole-grid contexts decide per loop; the ranks of a
// Z-slab run must agree and go by the grid alone)
bool k12_canon3_suits(const GridK& g, const ActiveBox& box);
// one red-black SOR iteration (colour 0 then colour 1) in one pass over HBM: work[src] -> work[dst]
hipError_t k12_launch_canon2_sor(hipStream_t s, const uint8_t* mask, const float* rhs, const float* pin,
                                 float* pout, const uint8_t* bricks, const GridK& g, float p_oob,
                                 const ActiveBox& box, float omega);
struct FusedRange;
// the same over an x window of nt_window (1 or 2) 256-cell columns starting at rg.xwin0
hipError_t k12_launch_canon2_win(hipStream_t s, int nt_window, const uint8_t* mask, const float* rhs,
                                 const float* pin, float* pout, float* pmid, const uint8_t* bricks,
                                 const GridK& g, float p_oob, const FusedRange& rg,
                                 const ActiveBox& box, int part, int part_lo, int part_hi);

}  // namespace fluid
