// pressure_fused_launch.h — host side of k12_canon2 launches (z-chunk choice, box shaping, split passes),
// shared by the translation units that instantiate the kernel (pressure_fused.hip: whole rows;
// pressure_fused_win.hip: x windows).
#pragma once

#include "kernels_pressure_fused.h"
#include "kernels_pressure_fused3.h"
#include "pressure_api.h"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

namespace fluid {

// z-chunk choice.  One 16-wavefront workgroup occupies a CU, so a launch runs in "rounds" of one
// workgroup per CU and a tile count just above a multiple of the CU count wastes most of a round.
// Estimate the makespan of every candidate chunk length by dealing the tiles (cost = planes + 2
// pipeline steps + a fixed start-up) to the CUs in launch order, and take the best.  Host
// arithmetic, cached per geometry.
// depth2: a second segment of planes cut into chunks of the same length (the EDGES launch of a split pass covers
// the planes at both faces of a slab: twice the workgroups — modelled as one segment it took chunks of two planes
// at h = 8, 512 workgroups in two rounds of eight steps where 256 of ten do)
static inline int pick_zchunk(int row_groups, int depth, int cus, int sweeps = 2, int depth2 = 0) {
    // cached per geometry (a slab loop alternates between a few depths; the model costs ~1 ms)
    static std::map<std::tuple<int, int, int, int>, int> cache;
    static std::mutex cache_mutex;  // contexts of different host threads share this table
    std::lock_guard<std::mutex> lock(cache_mutex);
    const auto key = std::make_tuple(row_groups, depth + 4096 * depth2, cus, sweeps);
    const auto hit = cache.find(key);
    if (hit != cache.end()) return hit->second;
    const double startup = 3.0;
    double best_cost = 1e300;
    int best = std::min(depth, 32);
    for (int zc = std::min(depth, row_groups * (depth / 16) >= cus ? 16 : 2); zc <= std::min(depth, 128);
         zc++) {
        std::vector<double> busy(cus, 0.0);  // min-heap by finish time
        auto cmp = [](double a, double b) { return a > b; };
        std::make_heap(busy.begin(), busy.end(), cmp);
        double makespan = 0.0;
        for (int seg = 0; seg < 2; seg++) {
            const int len = seg == 0 ? depth : depth2;
            for (int z = 0; z * zc < len; z++) {
                const int planes = std::min(zc, len - z * zc);
                const double cost = planes + 2 * (sweeps - 1) + startup;  // pipeline steps of a T-sweep march
                for (int y = 0; y < row_groups; y++) {
                    std::pop_heap(busy.begin(), busy.end(), cmp);
                    busy.back() += cost;
                    makespan = std::max(makespan, busy.back());
                    std::push_heap(busy.begin(), busy.end(), cmp);
                }
            }
        }
        if (makespan < best_cost) {
            best_cost = makespan;
            best = zc;
        }
    }
    cache[key] = best;
    return best;
}

// Ranges of a launch's (chunk, row tile) units for the 8 XCDs (FusedRange::xcd_start), or none.  Two
// candidates — the same number of units each, the same cost each (the last chunk of a segment is shorter:
// pick_zchunk fills the last round of workgroups with it) — and the dispatcher's own deal (unit L to XCD
// L % 8) are run through the makespan model XCD by XCD; the better candidate is used unless the deal beats it
// by more than 6 %.
// Measured (profiles/round02/xcd_ranges_ab.txt): 256^3, 240 units of one round: ranges of 30 units 41 us,
// the deal 45 us, ranges of 29..32 units 59 us — the XCD handed 32 workgroups for its 32 CUs ran a second
// round, so a launch of one round is modelled on one CU fewer per XCD; 512^3, 520 units: equal counts give one
// XCD 65 full-length units, three rounds where the deal and equal costs run two; 1024 x 1024 x 64, 256 units:
// ranges 167 us, the deal 183 us.
struct XcdPlan {
    bool on;
    int start[9];
};
static inline XcdPlan xcd_plan(int by, int nz_lo, int nz, int zchunk, int seg1, int seg2, int cus, int sweeps = 2) {
    static std::map<std::tuple<int, int, int, int, int, int, int, int>, XcdPlan> cache;
    static std::mutex cache_mutex;
    std::lock_guard<std::mutex> lock(cache_mutex);
    const auto key = std::make_tuple(by, nz_lo, nz, zchunk, seg1, seg2, cus, sweeps);
    const auto hit = cache.find(key);
    if (hit != cache.end()) return hit->second;
    const int U = by * nz;
    std::vector<int> cost(U);
    long total = 0;
    for (int tz = 0, u = 0; tz < nz; tz++) {
        const int planes = tz < nz_lo ? std::min(zchunk, seg1 - tz * zchunk)
                                      : std::min(zchunk, seg2 - (tz - nz_lo) * zchunk);
        for (int y = 0; y < by; y++, u++) total += cost[u] = planes + 3 + 2 * (sweeps - 1);  // pick_zchunk's cost of a workgroup
    }
    auto makespan = [&](int u0, int u1, int step, int n_cus) {  // in-order list scheduling, as in pick_zchunk
        std::vector<long> busy(std::max(1, n_cus), 0);
        auto cmp = [](long a, long b) { return a > b; };
        long worst = 0;
        for (int u = u0; u < u1; u += step) {
            std::pop_heap(busy.begin(), busy.end(), cmp);
            busy.back() += cost[u];
            worst = std::max(worst, busy.back());
            std::push_heap(busy.begin(), busy.end(), cmp);
        }
        return worst;
    };
    XcdPlan by_count, by_cost;
    for (int x = 0; x <= 8; x++) by_count.start[x] = (int)((long)x * U / 8);
    {
        int x = 1;
        long acc = 0;  // cost of the units before u
        by_cost.start[0] = 0;
        for (int u = 0; u < U; acc += cost[u], u++)
            while (x < 8 && 8 * (2 * acc + cost[u]) > 2 * x * total) by_cost.start[x++] = u;
        while (x <= 8) by_cost.start[x++] = U;
    }
    const int per_xcd = std::max(1, cus / 8 - (U <= cus ? 1 : 0));
    long t_count = 0, t_cost = 0, t_deal = 0;
    for (int x = 0; x < 8; x++) {
        t_count = std::max(t_count, makespan(by_count.start[x], by_count.start[x + 1], 1, per_xcd));
        t_cost = std::max(t_cost, makespan(by_cost.start[x], by_cost.start[x + 1], 1, per_xcd));
        t_deal = std::max(t_deal, makespan(x, U, 8, per_xcd));
    }
    XcdPlan plan = t_cost < t_count ? by_cost : by_count;
    // (6 % of slack: the model does not see the fabric reads the ranges save; 512^3 runs 126 against 120
    // in the model and the same 0.347 ms on the GPU, with 1.35 instead of 1.69 GB read)
    plan.on = 100 * std::min(t_cost, t_count) <= 106 * t_deal;
    if (getenv("FLUID_FUSED_DEBUG"))  // one line per launch shape
        fprintf(stderr, "xcd_plan: %d row tiles x %d chunks of %d planes: by count %ld, by cost %ld, dealt %ld -> %s\n",
                by, nz, zchunk, t_count, t_cost, t_deal, plan.on ? (t_cost < t_count ? "by cost" : "by count") : "off");
    cache[key] = plan;
    return plan;
}

static inline int cu_count() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            n <= 0)
            n = 256;
    }
    return n;
}

// rows per wavefront (FusedGeom) by x tiles per row, measured on MI355X (Jacobi iterations/s):
//                              RG = 1, 16 waves / RG = 2, 8 waves / RG = 3, 8 waves / RG = 2, 12 waves
//   NT = 1 (256^3)             27.3 k / 24.6 k / 43.7 k (47.1 k with the XCD ranges) / 46.7 k
//   NT = 2 (512^3)             5.09 k / 5.58 k / 5.76 k / 5.73 k     (RG 3: 225 VGPRs, no spill since the
//                                                       plane step was reordered; before that 5.22 k)
//   NT = 4 (1024 x 1024 x 64)  9.3 k / 8.7 k / 11.8 k / 12.6 k
// Rows per workgroup: 16 / NT with one or two rows per wavefront on 16 or 8 wavefronts, 24 / NT with three rows
// on 8 wavefronts or two rows on 12 (160 VGPRs: three wavefronts per SIMD instead of two) — at NT = 4 a third
// instead of half of the rows are halo.  Three rows on 8 wavefronts, then, and two on 12 for the widest grids;
// also measured with three: the 512 x 512 x 64 slab of an 8-way run (+ 10 % over two on 8) and the loop inside
// a full-tank step (37.0 -> 35.4 ms).  FLUID_FUSED_RG = 1, 2 or 3 overrides it.
static inline int fused_rows_per_wave(int nt) {
    static const int forced = [] {
        const char* e = getenv("FLUID_FUSED_RG");
        const int v = e ? atoi(e) : 0;
        return (v >= 1 && v <= 3) ? v : 0;
    }();
    if (forced) return forced;
    return nt >= 3 ? 2 : 3;
}

// The launch of the kernel with streaming stores (k12_canon2<NT, false, RG, KEEP, false, true>), defined and
// instantiated in pressure_fused_stream.hip for NT = 2, 4 and RG = 2, 3.
struct FusedLaunchArgs {
    dim3 grid;
    size_t lds;
    hipStream_t stream;
    const uint8_t* mask;
    const float* rhs;
    const float* pin;
    float* pout;
    float* pmid;
    const uint8_t* bricks;
    BrickK bk;
    GridK g;
    float p_oob;
    int zchunk;
    FusedRange r;
    float omega;
};
template <int NT, int RG, bool KEEP>
hipError_t k12_launch_streaming(const FusedLaunchArgs& a);

// the geometry of the T-sweeps-per-pass kernel (T = 2: k12_canon2; T = 3: k12_canon_t)
template <int NT, int RG, int T>
struct FusedShape {
    using G = FusedGeom<NT, RG>;
};
template <int NT, int RG>
struct FusedShape<NT, RG, 3> {
    using G = FusedGeomT<NT, RG, 3>;
};
// the kernel with streaming stores (three sweeps: k12_canon_t<NT, false, RG, 3, KEEP, true>), pressure_fused3.hip
template <int NT, int RG, bool KEEP>
hipError_t k12_launch_streaming3(const FusedLaunchArgs& a);

template <int NT, bool WIN, int RG, bool KEEP, bool SOR = false, int T = 2>
static hipError_t launch_keep(hipStream_t s, const uint8_t* mask, const float* rhs, const float* pin,
                            float* pout, float* pmid, const uint8_t* bricks, const GridK& g,
                            float p_oob, const FusedRange& rg, const ActiveBox& box, int part,
                            int part_lo, int part_hi, float omega = 0.f) {
    // the dynamic-LDS limit is an attribute of the function on a device: once per instantiation and
    // device (a process may hold contexts on several)
    using G = typename FusedShape<NT, RG, T>::G;
    static_assert(T == 2 || (T == 3 && !SOR), "two or three Jacobi sweeps per pass");
    // the kernel with streaming stores exists for the full-row Jacobi launches of grids 512 cells wide or wider
    constexpr bool STREAMING_VARIANT = !WIN && !SOR && NT >= 2 && RG >= 2;
    static std::atomic<bool> attr_set[64] = {};  // contexts of different host threads share this table
    const size_t lds = G::lds_bytes;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        const void* fn;
        if constexpr (T == 3) fn = reinterpret_cast<const void*>(&k12_canon_t<NT, WIN, RG, 3, KEEP, false>);
        else fn = reinterpret_cast<const void*>(&k12_canon2<NT, WIN, RG, KEEP, SOR>);
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set[dev].store(true, std::memory_order_release);  // (setting it twice is harmless)
    }
    constexpr int TY = G::TY;
    FusedRange r = rg;
    int ty0 = 0, ty1 = (g.H + TY - 1) / TY;  // row tiles [ty0, ty1)
    // Sparse scene: launch only the tiles whose output rows / planes meet the bricks that hold water.  The
    // others would leave at once (no water cell to write), but a workgroup that starts and leaves still costs
    // a slot on a CU.  On a Z slab the box is the union with the neighbouring slabs' (engine.hip:
    // fluid_step_set_box) and the plane range stays what the halo depth asks for.
    if (box.valid) {
        if (box.y_hi <= box.y_lo || box.z_hi <= box.z_lo) return hipSuccess;  // no water at all
        ty0 = box.y_lo / TY;
        ty1 = (std::min(box.y_hi, g.H) + TY - 1) / TY;
        if (rg.zout_lo == 0 && rg.zout_hi == g.Dl) {
            r.zout_lo = std::max(0, box.z_lo);
            r.zout_hi = std::min(g.Dl, box.z_hi);
        } else {
            // a slab's launch also recomputes ghost planes; what lies two halo depths from this slab's own
            // water cannot reach it before the next exchange (values spread one plane per sweep, through
            // water cells only, and an exchange comes every LOOP_GHOST sweeps at the latest)
            r.zout_lo = std::max(rg.zout_lo, box.z_lo - 2 * LOOP_GHOST);
            r.zout_hi = std::min(rg.zout_hi, box.z_hi + 2 * LOOP_GHOST);
        }
    }
    r.ytile0 = ty0;
    const int by = ty1 - ty0;
    // planes to compute: [zout_lo, zout_hi) minus the hole [part_lo, part_hi) (part = edges), or only
    // the hole (part = interior); whole pass otherwise
    if (part != FUSED_WHOLE) {
        const int a = std::min(std::max(part_lo, r.zout_lo), r.zout_hi);
        const int b = std::min(std::max(part_hi, a), r.zout_hi);
        if (part == FUSED_INTERIOR) {
            r.zout_lo = a;
            r.zout_hi = b;
        } else {
            r.hole_lo = a;
            r.hole_hi = b;
        }
    }
    if (part != FUSED_EDGES) r.hole_lo = r.hole_hi = r.zout_hi;
    const int seg1 = r.hole_lo - r.zout_lo, seg2 = r.zout_hi - r.hole_hi;
    const int depth = std::max(seg1, seg2);
    if (depth <= 0) return hipSuccess;
    int zchunk = std::min(pick_zchunk(by, depth, cu_count(), T, std::min(seg1, seg2)), depth);
    // Sparse scene without a box (Z slab: the ghost planes are not covered by the activity map): most
    // workgroups leave at once and the few that work should be short, so that they run side by side.
    if (!box.valid && box.fraction >= 0.f &&
        box.fraction * by * ((depth + zchunk - 1) / zchunk) < 0.75f * cu_count())
        zchunk = std::min(zchunk, 24);
    if (const char* e = getenv("FLUID_FUSED_ZCHUNK")) zchunk = std::max(1, atoi(e));  // tuning aid
    r.nz_lo = (seg1 + zchunk - 1) / zchunk;
    const int nz = r.nz_lo + (seg2 + zchunk - 1) / zchunk;
    dim3 grid(1, by, nz);
    // Deal the launch's units (chunk, row tile) to the XCDs in contiguous ranges (workgroup L runs on XCD
    // L % 8 and takes unit L / 8 of that XCD's range): y-adjacent workgroups then share an L2, which serves
    // the halo rows they both read.  Fabric reads of a 512^3 launch: 1.91 -> 1.35 GB (TCC_EA0_RDREQ x 128 B,
    // profiles/round02).  The placement is an observed property of the dispatcher, used for speed only;
    // FLUID_FUSED_XCD=0 turns it off, 2 uses the ranges whatever the model says of them.
    // streaming stores when the launch's working set (13 B per cell of the planes it covers) is larger than the
    // 256 MB memory-side cache by a margin: little of what it writes is then still cached when the next launch
    // reads it, and the stores take no room from the rows the launch itself reads twice (st_f4).  Measured with
    // the stores forced either way: 0.27 GB (slab of an 8-way 512^3 run) - 2.5 %, 0.44 GB (4-way) + 1.3 %,
    // 0.87 GB (2-way; 1024 x 1024 x 64) + 2.4 % / + 1.4 %, 1.7 GB (512^3) + 4 %.  FLUID_FUSED_NT = 0 / 1 overrides.
    static const int nt_forced = [] {
        const char* e = getenv("FLUID_FUSED_NT");
        return e == nullptr ? -1 : atoi(e);
    }();
    const bool streaming =
        STREAMING_VARIANT && (nt_forced >= 0 ? nt_forced != 0
                                             : (int64_t)g.W * g.H * (r.zout_hi - r.zout_lo) * 13 >= (int64_t)3 << 27);
    r.xcd_rows = 0;
    for (int& v : r.xcd_start) v = 0;
    static const int xcd_ranges = [] {
        const char* e = getenv("FLUID_FUSED_XCD");
        return e == nullptr ? 1 : atoi(e);
    }();
    if (xcd_ranges && by * nz >= 16) {
        const XcdPlan plan = xcd_plan(by, r.nz_lo, nz, zchunk, seg1, seg2, cu_count(), T);
        if (plan.on || xcd_ranges == 2) {
            int longest = 0;
            for (int x = 0; x < 8; x++) longest = std::max(longest, plan.start[x + 1] - plan.start[x]);
            for (int x = 0; x <= 8; x++) r.xcd_start[x] = plan.start[x];
            r.xcd_rows = by;
            grid = dim3(8 * longest, 1, 1);
        }
    }
    BrickK bk;
    bk.nbx = (g.W + BRICK_X - 1) / BRICK_X;
    bk.nby = (g.H + BRICK_Y - 1) / BRICK_Y;
    bk.nbz = (g.Dl + BRICK_Z - 1) / BRICK_Z;
    if constexpr (STREAMING_VARIANT) {
        if (streaming) {
            FusedLaunchArgs a{grid, lds, s, mask, rhs, pin, pout, pmid, bricks, bk, g, p_oob, zchunk, r, omega};
            if constexpr (T == 3) return k12_launch_streaming3<NT, RG, KEEP>(a);
            else return k12_launch_streaming<NT, RG, KEEP>(a);
        }
    }
    (void)streaming;
    if constexpr (T == 3)
        hipLaunchKernelGGL((k12_canon_t<NT, WIN, RG, 3, KEEP, false>), grid, dim3(G::THREADS), lds, s, mask, rhs, pin,
                           pout, pmid, bricks, bk, g, p_oob, zchunk, r);
    else
        hipLaunchKernelGGL((k12_canon2<NT, WIN, RG, KEEP, SOR>), grid, dim3(G::THREADS), lds, s, mask, rhs, pin, pout,
                           pmid, bricks, bk, g, p_oob, zchunk, r, omega);
    return hipSuccess;
}

// with / without the store of the odd iterate (the last pair of a loop keeps it)
template <int NT, bool WIN, int RG, int T = 2>
static hipError_t launch_nt(hipStream_t s, const uint8_t* mask, const float* rhs, const float* pin,
                            float* pout, float* pmid, const uint8_t* bricks, const GridK& g, float p_oob,
                            const FusedRange& rg, const ActiveBox& box, int part, int part_lo, int part_hi) {
    if (pmid)
        return launch_keep<NT, WIN, RG, true, false, T>(s, mask, rhs, pin, pout, pmid, bricks, g, p_oob, rg, box,
                                                        part, part_lo, part_hi);
    return launch_keep<NT, WIN, RG, false, false, T>(s, mask, rhs, pin, pout, pmid, bricks, g, p_oob, rg, box, part,
                                                     part_lo, part_hi);
}

// is a launch shaped to `box` small enough that 16 thin wavefronts on many CUs beat 8 fat ones on few?
// (512^3 dam break, x-window launches: 33.7 us per two-sweep launch with RG = 1, 37.2 with RG = 3).  Estimates
// the workgroups of the fat shape (ty output rows each); below two per CU: thin.
static inline bool small_box_launch(const GridK& g, const FusedRange& rg, const ActiveBox& box, int ty) {
    if (!box.valid) return false;
    const int rows = std::max(0, std::min(box.y_hi, g.H) - box.y_lo);
    const int planes = (rg.zout_lo == 0 && rg.zout_hi == g.Dl) ? std::max(0, box.z_hi - box.z_lo)
                                                               : rg.zout_hi - rg.zout_lo;
    const int tiles = (rows + ty - 1) / std::max(ty, 1), chunks = (planes + 15) / 16;
    return tiles * chunks < 2 * cu_count();
}

// the instantiation for the configured rows per wavefront
template <int NT, bool WIN>
static hipError_t launch_rg(hipStream_t s, const uint8_t* mask, const float* rhs, const float* pin,
                            float* pout, float* pmid, const uint8_t* bricks, const GridK& g, float p_oob,
                            const FusedRange& rg, const ActiveBox& box, int part, int part_lo, int part_hi) {
    int rows_per_wave = fused_rows_per_wave(NT);
    if (rows_per_wave > 1 && box.valid && getenv("FLUID_FUSED_RG") == nullptr) {
        // A launch shaped to a small box of water has few workgroups: 16 thin wavefronts on many CUs finish
        // sooner than 8 fat ones on few (512^3 dam break, x-window launches: 33.7 us per two-sweep launch with
        // RG = 1, 37.2 with RG = 3).  Estimate the workgroups of the fat shape; below two per CU, go thin.
        const int rows = std::max(0, std::min(box.y_hi, g.H) - box.y_lo);
        const int planes = (rg.zout_lo == 0 && rg.zout_hi == g.Dl) ? std::max(0, box.z_hi - box.z_lo)
                                                                   : rg.zout_hi - rg.zout_lo;
        const int ty = (fused_waves(rows_per_wave) / NT) * rows_per_wave - 2;
        const int tiles = (rows + ty - 1) / std::max(ty, 1), chunks = (planes + 15) / 16;
        if (tiles * chunks < 2 * cu_count()) rows_per_wave = 1;
    }
    switch (rows_per_wave) {
        case 1:
            return launch_nt<NT, WIN, 1>(s, mask, rhs, pin, pout, pmid, bricks, g, p_oob, rg, box, part,
                                         part_lo, part_hi);
        case 3:
            return launch_nt<NT, WIN, 3>(s, mask, rhs, pin, pout, pmid, bricks, g, p_oob, rg, box, part,
                                         part_lo, part_hi);
        default:
            return launch_nt<NT, WIN, 2>(s, mask, rhs, pin, pout, pmid, bricks, g, p_oob, rg, box, part,
                                         part_lo, part_hi);
    }
}

}  // namespace fluid
