// quiet_bricks.h — skipping the parts of the grid where a step changes nothing.
#pragma once

#include "device_common.h"
#include "pressure_common.h"  // the activity bricks (BrickK, BRICK_*)

namespace fluid {

// Far from the water a step changes nothing: 05 / 07 / 08 / 13 act only on cells that are, or touch,
// water or air cells, and 10 is idempotent, so from the third consecutive step in which no water cell
// lies within two cells of a brick, VELOCITIES_1 == VELOCITIES_2 (xyz) there, w = 0, and DIVERGENCES is
// what 11 would store again.  fluid_run_step keeps, per activity brick (256 x 4 x 16 cells, the bricks
// of the pressure loop), the number of consecutive steps in which neither the brick nor any of its 26
// neighbours held water (k_update_quiet, from the map k12_prepare builds right after 06), and the
// workgroups of 07+08, 09+10+11 and 13 whose cells lie in a brick with a streak >= QUIET_MIN_STREAK leave
// at once; so do the pressure clears 12a / 12b (the loop writes water cells only, the rest still holds
// p_air), the b_i pass and the import pass of the pressure loop (DIVERGENCES and the cell types of such
// a brick have not changed since those passes last ran on it).  Images keep the bits the full passes would have written (inside the step, VELOCITIES_1.w of
// such a brick is 0 where the list has 1 between 10 and 13 — nobody reads it there).  Any write from
// outside fluid_run_step (uploads, clears, single sections, parameters) resets the streaks.
constexpr uint32_t QUIET_MIN_STREAK = 3;
// Sections 01a-05 run before the step's own map exists; they use a second, one-step test instead
// (k_update_early_quiet, after 01): a brick is skipped when neither it nor any of its 26 neighbours held
// water after the previous step (the activity bricks still in memory) or receives a particle in this one
// (bytes set by 01_update_densities).  There the density is already 0 (01a), 02 followed by 03 rewrites
// the types that are there (border SOLID -> INACTIVE -> SOLID, the rest INACTIVE; no AIR without water
// within a cell), and no cell's activity changes (04 + 05).  Encoded as streak 255 / 0 in an array of its
// own, so the same test macro applies.


// for kernels launched with cell_grid() / cell_block() (64 x 4 x 1 cells per workgroup): one brick per
// workgroup, a scalar test
// (x 4 cells with `xchunks` = 4: such launches give every workgroup four 64-cell chunks of a row to loop
// over, i.e. one workgroup per brick row — a launch over mostly quiet bricks is bound by the rate at
// which empty workgroups can be dispatched, about one per cycle)
#define FLUID_LEAVE_IF_QUIET(quiet, bk, xchunks)                                                 \
    if (quiet) {                                                                                 \
        const int qb_ = brick_index(bk, (int)(blockIdx.x * 64u * (unsigned)(xchunks)) / BRICK_X, \
                                    (int)(blockIdx.y * 4u) / BRICK_Y,                            \
                                    ((int)blockIdx.z * g.zl) / BRICK_Z);                         \
        if ((uint32_t)quiet[qb_] >= QUIET_MIN_STREAK) return;                                    \
    }

// the planes of a workgroup: blockIdx.z counts groups of g.zl planes (GridK::zl: 1, or BRICK_Z while bricks
// are being skipped — then a workgroup stays inside one brick and the test above holds for all its planes)
#define FLUID_FOR_PLANES_OF_WORKGROUP() \
    for (int lz = (int)blockIdx.z * g.zl, lz_end_ = min(lz + g.zl, g.Dl); lz < lz_end_; lz++)

// cell loop of those kernels: x runs over the workgroup's `xchunks` chunks of 64 cells
#define FLUID_FOR_CELLS_OF_ROW(xchunks)                                              \
    const int y = blockIdx.y * blockDim.y + threadIdx.y;                             \
    if (y >= g.H) return;                                                            \
    FLUID_FOR_PLANES_OF_WORKGROUP()                                                  \
    for (int xc_ = 0; xc_ < (xchunks); xc_++) {                                      \
        const int x = ((int)blockIdx.x * (xchunks) + xc_) * 64 + (int)threadIdx.x;   \
        if (x >= g.W) break;
#define FLUID_END_FOR_CELLS }

// the same for the four-cells-per-thread passes (64 x 4 threads = 256 x 4 x 1 cells per workgroup)
#define FLUID_LEAVE_IF_QUIET_V4(quiet, bk)                                                         \
    if (quiet) {                                                                                   \
        const int qb_ = brick_index(bk, (int)(blockIdx.x * 256u) / BRICK_X,                        \
                                    (int)(blockIdx.y * 4u) / BRICK_Y,                              \
                                    ((int)blockIdx.z * g.zl) / BRICK_Z);                           \
        if ((uint32_t)quiet[qb_] >= QUIET_MIN_STREAK) return;                                      \
    }

}  // namespace fluid
