// engine.hip — host side of the C ABI in include/fluid_engine.h: context, device memory, the
// section table (the reference's SimulationInitializationSections / SimulationStepSections,
// /root/reference/fluid_flow_sections.h:136-338) and kernel launches on one in-order HIP stream.
//
// gfx950 only.  There is no CPU path in this library: every section either launches a HIP kernel
// or returns an error.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/fluid_engine.h"
#include "device_common.h"
#include "kernels_grid.h"
#include "pressure_api.h"
#include "kernels_sampler.h"
#include "kernels_particle_bins.h"
#include "kernels_step_fused.h"
#include "kernels_surface.h"

static_assert(sizeof(fluid_params) == FLUID_PARAMS_BYTES, "params block must be 264 bytes");
static_assert(offsetof(fluid_params, particle_compute_size) == 48, "std140 offset");
static_assert(offsetof(fluid_params, particle_spawn_cube_resolution) == 64, "std140 offset");
static_assert(offsetof(fluid_params, particle_spawn_cube_offset) == 80, "std140 offset");
static_assert(offsetof(fluid_params, particle_spawn_cube_size) == 96, "std140 offset");
static_assert(offsetof(fluid_params, gravity) == 108, "std140 offset");
static_assert(offsetof(fluid_params, dens_diffuse_k) == 148, "std140 offset");
static_assert(offsetof(fluid_params, particle_color) == 160, "std140 offset");
static_assert(offsetof(fluid_params, light_dir) == 176, "std140 offset");
static_assert(offsetof(fluid_params, ambient_color) == 192, "std140 offset");
static_assert(offsetof(fluid_params, diffuse_color) == 208, "std140 offset");
static_assert(offsetof(fluid_params, fluid_surface_render_size) == 224, "std140 offset");
static_assert(offsetof(fluid_params, active_particle_w) == 236, "std140 offset");
static_assert(offsetof(fluid_params, fountain_position) == 240, "std140 offset");
static_assert(offsetof(fluid_params, solid_repel_velocity) == 256, "std140 offset");

using namespace fluid;

namespace {

thread_local std::string g_create_error;

constexpr uint64_t kAlign = 4096;
inline uint64_t align_up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }

struct ImageDesc {
    uint32_t elem_bytes = 0;   // bytes per cell
    uint64_t offset = 0;       // arena offset of ghost plane -IMG_GHOST
    uint64_t bytes = 0;        // Dl + 2*IMG_GHOST planes
};

struct TimerSlot {
    hipEvent_t start = nullptr, stop = nullptr;
    int section = -1;
};

}  // namespace

struct fluid_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    uint8_t* arena = nullptr;
    bool own_arena = false;
    uint64_t arena_bytes = 0;

    fluid_params params{};
    ParamsK pk{};
    GridK g{};
    uint64_t particle_capacity = 0;
    uint64_t particles_offset = 0;
    // particles stored sorted by bin (kernels_particle_bins.h; whole-grid contexts, FLUID_OPT_PARTICLE_SORT).
    // Everything here lives outside the arena and is allocated by the first sort.
    struct ParticleSort {
        float4* alt = nullptr;                   // the second particle buffer (a sort scatters into the other one)
        uint32_t* slot_of[2] = {nullptr, nullptr};  // per buffer: the slot (index of the API) of each stored particle
        uint32_t* bin_count = nullptr;           // bins + 2 words each
        uint32_t* bin_start = nullptr;
        uint32_t* cursor = nullptr;
        uint32_t* strays = nullptr;              // cell indices, particle_capacity entries
        uint32_t* stray_count = nullptr;         // [0] entries of `strays`, [1] particles found outside their bin
        uint32_t* stray_host = nullptr;          // pinned copy of stray_count, one step behind
        hipEvent_t stray_ev = nullptr;
        bool stray_pending = false;
        PBinK bk{};
        int cur = 0;          // buffer that holds the particles: 0 = the arena's, 1 = alt
        bool valid = false;   // ... in an order of its own: slot_of[cur] says which slot each one belongs to
        bool binned = false;  // ... and bin_start describes it: 01 and 14 run their binned kernels
        uint32_t steps_since_sort = 0;
        uint32_t suspended = 0;  // 01 passes left before sorting is tried again (the flow outruns it)
        uint32_t backoff = 64;   // ... the next time: doubles with every suspension, back to 64 after a calm spell
        bool failed = false;  // an allocation failed: slot order from now on
        uint64_t sorts = 0;
        double stray_steps = 0.0;  // sum over the steps since the last sort of the fraction of strays
    } ps;
    bool dens_zero = false;  // PARTICLE_DENSITIES was cleared and nothing has counted into it since
    uint32_t pressure_iterations = 200;
    int diffuse_mode = FLUID_DIFFUSE_REFERENCE_EXACT;
    uint32_t pressure_dispatch_index = 0;  // loop counter of the 12_solve_pressure section
    int solver = FLUID_SOLVER_JACOBI;      // opt-in: FLUID_SOLVER_RED_BLACK_SOR (not the reference's results)
    float sor_omega = 1.0f;
    bool is_slab = false;

    ImageDesc img[8];
    int64_t opt[FLUID_OPT_COUNT] = {0};

    // surface-prep passes 15-18 on the detailed grid (kernels_surface.h); images 8..11
    bool surface = false;
    SurfK sk{};
    uint64_t surf_cells = 0;
    uint64_t surf_offset[4] = {0, 0, 0, 0};
    uint32_t surface_steps = 4;
    uint32_t surface_dispatch_index = 0;  // loop counter of the 18_diffuse_float_densities section
    bool surface_fuse17 = false;          // inside fluid_run_step: 16 also writes what 17 would
    float* blur_tmp = nullptr;            // third float image of the fused 18 loop (k18_pair), first use
    bool blur_tmp_failed = false;
    uint32_t* mc_tables = nullptr;        // MARCHING_CUBES_COUNTS_BUF (256 words) + _EDGES_BUF (256 * 15)
    bool mc_loaded[2] = {false, false};
    template <typename T>
    T* surf(int image_id) const {
        return reinterpret_cast<T*>(arena + surf_offset[image_id - 8]);
    }

    // loop-section fast path of 12_solve_pressure (kernels_pressure.h / kernels_pressure_fused.h)
    uint64_t mask_offset = 0, rhs_offset = 0;  // per-cell byte mask / b_i, Dl + 2*LOOP_GHOST planes
    uint64_t active_offset = 0, active_bytes = 0;  // one byte per 256x4x16 brick: holds water?
    uint64_t quiet_offset = 0;    // one byte per brick: steps since water was near (quiet_bricks.h)
    uint64_t early_offset = 0;    // one byte per brick: 255 = skipped by 02 / 03 / 04+05 of this step
    uint64_t pbricks_offset = 0;  // one byte per brick: a particle was counted there in this step
    bool early_in_use = false;    // inside fluid_run_step, between 01 and 06
    bool early_step = false;      // ... and until the step's mask pass: early() says where 02 ... 06 left the types alone
    bool early_wanted = false;    // the running 01 pass also derives the one-step test from its marks
    bool pbricks_valid = false;   // pbricks() marks every brick in which the last 01 counted a particle
    // a step driven section by section from outside (fluid_step_begin / _end: the Z-slab driver), with the
    // skipping fluid_run_step does
    bool driver_step = false, ds_quiet = false, ds_early = false;
    uint64_t ghost_bricks_offset = 0;  // the neighbouring slabs' edge layers of the activity bricks: 2 x nbx*nby
    bool ghost_bricks_valid = false;   // ... as exchanged in this step
    bool box_from_driver = false;      // c->box was set by fluid_step_set_box (union with the neighbours')
    bool quiet_valid = false;     // the streaks describe the images (nothing wrote them from outside)
    bool quiet_in_use = false;    // inside fluid_run_step, between 06 and 13: kernels may skip
    uint64_t work_offset[3] = {0, 0, 0};  // working pressure buffers, Dl + 2*LOOP_GHOST planes each
    uint64_t flags_offset = 0;            // [0] sampler halo violation, [1] leaver counter,
                                          // [2..8] brick summary, [9] quiet bricks of the last step, [10..11] x extent of the water
    uint32_t* brick_count_host = nullptr; // pinned, 7 words (k12_count_bricks); written by an async
                                          // copy after k12_prepare
    bool v1_w_zero = false;               // every texel of VELOCITIES_1 has w == +0.0f (kernels_step_fused.h)
    bool box_pending = false;             // that copy has been enqueued and not been waited for
    ActiveBox box;                        // where the water is, for launch shaping
    // wide sampler source of the 07 fallback pass (fluid_sampler_wide_begin): wide_lo + Dl + wide_hi planes
    // of RGBA32F, allocated on first use outside the arena
    float4* wide = nullptr;
    uint64_t wide_bytes = 0;
    int wide_lo = 0, wide_hi = 0;
    uint64_t leavers_offset = 0;          // Leaver list of the particle migration (slab contexts)
    uint32_t leavers_capacity = 0;
    // compact particle storage of a slab context (kernels_sampler.h: CompactParticles), outside the arena:
    // 20 bytes per particle the slab holds, plus headroom
    struct Local {
        float4* buf[2] = {nullptr, nullptr};   // two sets of arrays: the entries live in set `cur`, a squeeze or
        uint32_t* pid[2] = {nullptr, nullptr}; // a sort writes them into the other one
        uint64_t capn[2] = {0, 0};             // entries each set can hold
        int cur = 0;
        uint32_t* counters = nullptr;          // device: [0] entries appended, [1] holes made
        uint32_t n = 0;         // entries, holes included
        uint32_t holes = 0;
        uint32_t n_sorted = 0;  // while ps.binned: entries [0, n_sorted) are in bin order (ps.bin_start), the
                                // rest was adopted since the sort
        uint64_t strays_cap = 0;
        bool on = false;        // this context stores its particles this way
        uint64_t squeezes = 0, grows = 0;
    } loc;
    CompactParticles compact() const {
        CompactParticles o;
        o.buf = loc.buf[loc.cur];
        o.pid = loc.pid[loc.cur];
        o.counters = loc.counters;
        o.cap = (uint32_t)loc.capn[loc.cur];
        return o;
    }
    bool mask_valid = false;      // mask + bricks match CELL_TYPES and the cell type values
    bool rhs_valid = false;       // b_i matches DIVERGENCES and rho, dx, dt
    bool bg_valid[3] = {false, false, false};  // non-water cells of work[i] hold their constants
    // state of an open loop (fluid_pressure_loop_begin / _advance / _end, and run_fast_loop)
    bool loop_open = false;
    int loop_cur = 0;        // working buffer of the newest iterate
    int loop_prev = -1;      // working buffer of the iterate before it, -1 = not kept
    uint32_t loop_k = 0;     // index of the newest iterate
    int loop_halo = 0;       // Z slabs: valid ghost planes per side of work[loop_cur] (shrinks with
                             // every sweep, restored by the caller's halo exchange)
    int loop_aux_halo = 0;   // ... of mask / b_i (fixed for the loop)
    bool loop_ghost_bg = false;  // ghost planes of the other two buffers hold their constants
    hipStream_t edge_stream = nullptr;  // FLUID_OPT_EDGE_STREAM: EDGES launches of split passes
    hipEvent_t ev_pass_start = nullptr, ev_edges_done = nullptr;
    bool edges_pending = false;
    int loop_part_done = 0;      // FUSED_EDGES / FUSED_INTERIOR: that half of a split pass is launched
    bool loop_part_keep = false;
    uint32_t loop_part_sweeps = 2;
    int loop_part_lo = 0, loop_part_hi = 0;

    bool timing = false;
    std::vector<TimerSlot> pending;
    std::vector<TimerSlot> free_slots;
    double sec_ms[FLUID_SECTION_COUNT] = {0};
    uint64_t sec_calls[FLUID_SECTION_COUNT] = {0};

    std::string error;

    int fail(int code, const char* fmt, ...) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        error = buf;
        return code;
    }
    template <typename T>
    T* plane0(int image) const {  // owned plane 0 (IMG_GHOST ghost planes in front of it)
        return reinterpret_cast<T*>(arena + img[image].offset +
                                    (uint64_t)IMG_GHOST * g.plane * img[image].elem_bytes);
    }
    // The grid as the brick-skipping passes get it (GridK::zl, quiet_bricks.h): while a step skips bricks a
    // workgroup walks one brick layer instead of one plane — on grids with thousands of bricks, where
    // dispatching sixteen times as many mostly idle workgroups is what such a pass costs (512^3: 55 us each,
    // the step 5 % faster); with few bricks the finer workgroups win (256^3, 128^3: 8 % the other way).
    GridK g_bricks() const {
        GridK q = g;
        q.zl = ((quiet_in_use || early_in_use) && (active_bytes >= 4096 || opt[FLUID_OPT_QUIET_BRICKS] == 2))
                   ? BRICK_Z
                   : 1;
        return q;
    }
    float4* particles_home() const { return reinterpret_cast<float4*>(arena + particles_offset); }
    // where the particles are stored now (a slab context: the entries of its compact storage)
    float4* particles() const { return loc.on ? loc.buf[loc.cur] : (ps.cur ? ps.alt : particles_home()); }
    // owned plane 0 of the loop's arrays (LOOP_GHOST ghost planes in front of it)
    uint8_t* mask0() const { return arena + mask_offset + (uint64_t)LOOP_GHOST * g.plane; }
    float* rhs0() const { return reinterpret_cast<float*>(arena + rhs_offset) + LOOP_GHOST * g.plane; }
    float* work0(int i) const {
        return reinterpret_cast<float*>(arena + work_offset[i]) + LOOP_GHOST * g.plane;
    }
    uint8_t* bricks() const { return arena + active_offset; }
    uint8_t* quiet() const { return arena + quiet_offset; }
    // pointer for the kernels that may skip quiet bricks (null: process everything)
    const uint8_t* quiet_or_null() const { return quiet_in_use ? quiet() : nullptr; }
    uint8_t* early() const { return arena + early_offset; }
    uint8_t* pbricks() const { return arena + pbricks_offset; }
    uint8_t* ghost_bricks(int face) const {
        return arena + ghost_bricks_offset + (uint64_t)face * (active_bytes / (uint64_t)std::max(1, nbz()));
    }
    int nbz() const { return (g.Dl + 15) / 16; }
    // how the quiet / early tests see the brick layers beyond this context's faces
    BrickEdges brick_edges(bool ghosts_usable) const {
        BrickEdges e;
        const bool nb[2] = {g.z0 > 0, g.z0 + g.Dl < g.Dg};
        for (int f = 0; f < 2; f++) {
            e.ghost[f] = ghost_bricks(f);
            e.kind[f] = !nb[f] ? EDGE_NONE : (ghosts_usable && ghost_bricks_valid ? EDGE_GHOST : EDGE_UNKNOWN);
        }
        return e;
    }
    const uint8_t* early_or_null() const { return early_in_use ? early() : nullptr; }
    uint32_t* flags() const { return reinterpret_cast<uint32_t*>(arena + flags_offset); }
    // list 0 / 1: leavers to send down / up; 2 / 3: received from the neighbour below / above
    Leaver* leavers(int list) const {
        return reinterpret_cast<Leaver*>(arena + leavers_offset) + (uint64_t)list * leavers_capacity;
    }
    MigrateLists migrate_lists() const {
        MigrateLists m;
        m.send[0] = leavers(0);
        m.send[1] = leavers(1);
        m.count[0] = flags() + 28;
        m.count[1] = flags() + 29;
        m.capacity = leavers_capacity;
        return m;
    }
    // bookkeeping for the fast path: call whenever an image's device contents change
    void touched(int image) {
        if (image == FLUID_IMG_CELL_TYPES) {
            mask_valid = false;
            bg_valid[0] = bg_valid[1] = bg_valid[2] = false;
        } else if (image == FLUID_IMG_DIVERGENCES) {
            rhs_valid = false;
        } else if (image == FLUID_IMG_VELOCITIES_1) {
            v1_w_zero = false;
        } else if (image == FLUID_IMG_PARTICLE_DENSITIES_IMG) {
            dens_zero = false;
        }
    }
    uint64_t owned_cells() const { return (uint64_t)g.plane * (uint64_t)g.Dl; }
    void params_changed() {
        quiet_valid = false;
        mask_valid = rhs_valid = false;
        bg_valid[0] = bg_valid[1] = bg_valid[2] = false;
    }
};

#define HIP_TRY(ctx, call)                                                                    \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return (ctx)->fail(FLUID_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

namespace {

const uint32_t kElemBytes[8] = {16, 16, 1, 1, 4, 4, 4, 4};

int validate_params(const fluid_params& p, std::string& err) {
    char buf[256];
    if (p.fluid_size[0] == 0 || p.fluid_size[1] == 0 || p.fluid_size[2] == 0) {
        err = "fluid_size must be non-zero in every dimension";
        return FLUID_ERR_INVALID_ARG;
    }
    if ((uint64_t)p.fluid_size[0] * p.fluid_size[1] * p.fluid_size[2] > 0xFFFFFFFFull) {
        err = "fluid_size volume must fit in 32 bits (fluid_volume is a uint)";
        return FLUID_ERR_INVALID_ARG;
    }
    const uint32_t t[4] = {p.cell_type_inactive, p.cell_type_air, p.cell_type_water,
                           p.cell_type_solid};
    for (int i = 0; i < 4; i++) {
        if (t[i] > 255u) {
            err = "cell type values must fit the R8_UINT cell type image";
            return FLUID_ERR_INVALID_ARG;
        }
        for (int j = 0; j < i; j++)
            if (t[i] == t[j]) {
                snprintf(buf, sizeof buf, "cell type values must be distinct (%u used twice)", t[i]);
                err = buf;
                return FLUID_ERR_INVALID_ARG;
            }
    }
    if (p.particle_spawn_cube_resolution[0] == 0 || p.particle_spawn_cube_resolution[1] == 0 ||
        p.particle_spawn_cube_resolution[2] == 0) {
        err = "particle_spawn_cube_resolution must be non-zero";
        return FLUID_ERR_INVALID_ARG;
    }
    return FLUID_OK;
}

ParamsK make_params_k(const fluid_params& p) {
    ParamsK k{};
    k.t_inactive = p.cell_type_inactive;
    k.t_air = p.cell_type_air;
    k.t_water = p.cell_type_water;
    k.t_solid = p.cell_type_solid;
    k.dt = p.time_delta;
    k.p_air = p.pressure_air;
    k.dx = p.cell_width;
    k.rho = p.fluid_density;
    for (int i = 0; i < 3; i++) {
        k.spawn_res[i] = p.particle_spawn_cube_resolution[i];
        k.spawn_offset[i] = p.particle_spawn_cube_offset[i];
        k.spawn_size[i] = p.particle_spawn_cube_size[i];
        k.fountain[i] = p.fountain_position[i];
    }
    k.spawn_volume = p.particle_spawn_cube_volume;
    k.gravity = p.gravity;
    k.diffuse_k = p.diffuse_k;
    k.active_w = p.active_particle_w;
    k.fountain_force = p.fountain_force;
    k.repel = p.solid_repel_velocity;
    return k;
}

struct Layout {
    uint64_t img_offset[8], img_bytes[8];
    uint64_t particles_offset, particles_bytes;
    uint64_t mask_offset, rhs_offset, active_offset, active_bytes, quiet_offset, early_offset,
        pbricks_offset, work_offset[3];
    uint64_t flags_offset, leavers_offset, ghost_bricks_offset;
    uint32_t leavers_capacity;
    uint64_t surf_offset[4], surf_cells;
    uint64_t total;
};

bool info_has(const fluid_create_info* info, size_t field_end) { return info->struct_bytes >= field_end; }
bool wants_surface(const fluid_create_info* info) {
    return info_has(info, offsetof(fluid_create_info, surface_prep) + sizeof(uint32_t)) &&
           info->surface_prep != 0;
}

int compute_layout(const fluid_create_info* info, const fluid_params& p, Layout& L,
                   uint64_t& capacity, uint32_t& z0, uint32_t& dl) {
    z0 = 0;
    dl = p.fluid_size[2];
    if (info->slab_z_count != 0) {
        z0 = info->slab_z_begin;
        dl = info->slab_z_count;
        if ((uint64_t)z0 + dl > p.fluid_size[2]) return FLUID_ERR_INVALID_ARG;
    }
    capacity = info->particle_capacity;
    if (capacity == 0)
        capacity = (uint64_t)p.particle_compute_size[0] * (uint64_t)p.particle_compute_size[1];
    const uint64_t plane = (uint64_t)p.fluid_size[0] * p.fluid_size[1];
    uint64_t off = 0;
    for (int i = 0; i < 8; i++) {
        L.img_offset[i] = off;
        L.img_bytes[i] = plane * (uint64_t)(dl + 2 * IMG_GHOST) * kElemBytes[i];
        off = align_up(off + L.img_bytes[i], kAlign);
    }
    const bool slab_ctx = dl != p.fluid_size[2];
    L.particles_offset = off;
    L.particles_bytes = slab_ctx ? 0 : capacity * 16;  // a slab stores what it owns, outside the arena (Local)
    off = align_up(off + L.particles_bytes, kAlign);
    L.mask_offset = off;
    off = align_up(off + plane * (uint64_t)(dl + 2 * LOOP_GHOST), kAlign);
    L.rhs_offset = off;
    off = align_up(off + plane * (uint64_t)(dl + 2 * LOOP_GHOST) * 4, kAlign);
    L.active_offset = off;
    {
        int nbx, nby, nbz;
        k12_brick_dims((int)p.fluid_size[0], (int)p.fluid_size[1], (int)dl, nbx, nby, nbz);
        L.active_bytes = (uint64_t)nbx * nby * nbz;
    }
    off = align_up(off + L.active_bytes, kAlign);
    L.quiet_offset = off;
    off = align_up(off + L.active_bytes, kAlign);
    L.early_offset = off;
    off = align_up(off + L.active_bytes, kAlign);
    L.pbricks_offset = off;
    off = align_up(off + L.active_bytes, kAlign);
    for (int i = 0; i < 3; i++) {
        L.work_offset[i] = off;
        off = align_up(off + plane * (uint64_t)(dl + 2 * LOOP_GHOST) * 4, kAlign);
    }
    L.flags_offset = off;
    off = align_up(off + 256, kAlign);
    L.ghost_bricks_offset = off;
    off = align_up(off + 2 * L.active_bytes, kAlign);  // two layers would do; bytes are cheap here
    // Four lists (send down / up, received from below / above) of a sixteenth of this slab's share of the slots
    // each, at least 64 Ki entries.  More leavers than that in one step take more hand-over rounds
    // (fluid_particles_collect leaves them in place and says so), they are never dropped.
    const bool slab = slab_ctx;
    const uint64_t share = capacity / std::max<uint64_t>(1, p.fluid_size[2] / std::max<uint32_t>(dl, 1u));
    L.leavers_capacity =
        slab ? (uint32_t)std::min<uint64_t>(std::max<uint64_t>(share / 16, 65536), 0x0FFFFFFFu) : 0;
    L.leavers_offset = off;
    off = align_up(off + 4ull * L.leavers_capacity * sizeof(Leaver), kAlign);
    L.surf_cells = 0;
    for (int i = 0; i < 4; i++) L.surf_offset[i] = 0;
    if (wants_surface(info)) {
        if (slab || p.detailed_resolution < 1) return FLUID_ERR_INVALID_ARG;
        const uint64_t r = (uint64_t)p.detailed_resolution;
        L.surf_cells = (r * p.fluid_size[0]) * (r * p.fluid_size[1]) * (r * p.fluid_size[2]);
        for (int i = 0; i < 4; i++) {
            L.surf_offset[i] = off;
            off = align_up(off + L.surf_cells * 4, kAlign);
        }
    }
    L.total = std::max<uint64_t>(off, kAlign);
    return FLUID_OK;
}

dim3 cell_block() { return dim3(64, 4, 1); }
dim3 cell_grid(const GridK& g) { return dim3((g.W + 63) / 64, (g.H + 3) / 4, g.Dl); }
dim3 cell4_grid(const GridK& g) { return dim3((g.W / 4 + 63) / 64, (g.H + 3) / 4, g.Dl); }

// ---- timing ----------------------------------------------------------------------------------
int fold_timers(fluid_ctx* c) {
    if (c->pending.empty()) return FLUID_OK;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (auto& s : c->pending) {
        float ms = 0.f;
        HIP_TRY(c, hipEventElapsedTime(&ms, s.start, s.stop));
        c->sec_ms[s.section] += ms;
        c->sec_calls[s.section] += 1;
        c->free_slots.push_back(s);
    }
    c->pending.clear();
    return FLUID_OK;
}

// names of the sections as the reference's section lists spell them (fluid_flow_sections.h:139-388)
const char* const kSectionNames[FLUID_SECTION_COUNT] = {
    "init_clear_velocities_1", "init_clear_cell_types", "00_init_particles",
    "01a_clear_particle_densities", "01_update_densities", "02_update_water", "03_update_air",
    "04_compute_extrapolated_velocities", "05_set_extrapolated_velocities", "06_update_cell_types",
    "07_advect", "08_forces", "09_diffuse", "10_solids", "11_compute_divergence",
    "12a_clear_pressures_1", "12b_clear_pressures_2", "12_solve_pressure", "13_fix_divergence",
    "14_particles", "14a_clear_detailed_densities", "15_update_detailed_densities",
    "16_compute_detailed_densities_inertia", "17_compute_float_densities", "18_diffuse_float_densities",
    "init_clear_detailed_densities_inertia"};

// Tracing: with FLUID_ROCTX=1 in the environment every section is a roctx range named
// after it, so `rocprofv3 --marker-trace --kernel-trace` shows which kernels belong to which entry of the
// section list.  The roctx library is loaded on first use; without the variable (or the library) nothing happens.
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        const char* e = getenv("FLUID_ROCTX");
        if (!e || atoi(e) == 0) return;
        // rocprofv3 listens to the SDK's roctx; the roctracer one serves the older tools
        void* lib = dlopen("librocprofiler-sdk-roctx.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) return;
        push = reinterpret_cast<int (*)(const char*)>(dlsym(lib, "roctxRangePushA"));
        pop = reinterpret_cast<int (*)()>(dlsym(lib, "roctxRangePop"));
        if (!push || !pop) push = nullptr, pop = nullptr;
    }
};
const Roctx& roctx() {
    static const Roctx r;
    return r;
}

struct SectionTimer {
    fluid_ctx* c;
    TimerSlot slot;
    bool active = false;
    bool range = false;
    ~SectionTimer() {
        if (range) roctx().pop();
    }
    int begin(int section) {
        if (roctx().push && section >= 0 && section < FLUID_SECTION_COUNT) {
            roctx().push(kSectionNames[section]);
            range = true;
        }
        if (!c->timing) return FLUID_OK;
        if (c->pending.size() >= 8192) {
            int rc = fold_timers(c);
            if (rc) return rc;
        }
        if (!c->free_slots.empty()) {
            slot = c->free_slots.back();
            c->free_slots.pop_back();
        } else {
            HIP_TRY(c, hipEventCreate(&slot.start));
            HIP_TRY(c, hipEventCreate(&slot.stop));
        }
        slot.section = section;
        HIP_TRY(c, hipEventRecord(slot.start, c->stream));
        active = true;
        return FLUID_OK;
    }
    int end() {
        if (range) {
            roctx().pop();
            range = false;
        }
        if (!active) return FLUID_OK;
        HIP_TRY(c, hipEventRecord(slot.stop, c->stream));
        c->pending.push_back(slot);
        active = false;
        return FLUID_OK;
    }
};

// ---- fills -------------------------------------------------------------------------------------
// Fill the OWNED planes of an image with one texel value (FlowClearColorSection).  `v` holds the
// texel as 32-bit words: 4 for RGBA32F, 1 for R32*, low byte of v[0] for R8.
int fill_image4(fluid_ctx* c, int image, const uint32_t v[4]) {
    // images 8..11 live on the detailed grid (4-byte texels, no ghost planes)
    const uint32_t eb = image >= 8 ? 4u : c->img[image].elem_bytes;
    const uint64_t bytes = image >= 8 ? c->surf_cells * 4 : c->owned_cells() * eb;
    uint8_t* dst = image >= 8 ? c->surf<uint8_t>(image) : c->plane0<uint8_t>(image);
    uint4 pat;
    if (eb == 16)
        pat = make_uint4(v[0], v[1], v[2], v[3]);
    else if (eb == 4)
        pat = make_uint4(v[0], v[0], v[0], v[0]);
    else {
        const uint32_t b = (v[0] & 0xFFu) * 0x01010101u;
        pat = make_uint4(b, b, b, b);
    }
    // plane 0 is 16-byte aligned whenever W*H*elem is; tiny odd grids need byte/word head + tail
    const uint64_t addr = reinterpret_cast<uint64_t>(dst);
    const uint64_t head = std::min<uint64_t>(bytes, (16 - (addr & 15)) & 15);
    const uint64_t body = (bytes - head) / 16;
    const uint64_t tail_begin = head + body * 16;
    if (head % eb != 0) return c->fail(FLUID_ERR_UNSUPPORTED, "image base not element-aligned");
    if (body > 0) {
        const int blocks = (int)std::min<uint64_t>((body + 255) / 256, 256 * 8);
        hipLaunchKernelGGL(k_fill_u32x4, dim3(blocks), dim3(256), 0, c->stream,
                           reinterpret_cast<uint4*>(dst + head), (int64_t)body, pat);
    }
    auto small = [&](uint64_t b, uint64_t e) -> int {  // < 16 bytes, never RGBA32F
        if (e <= b) return FLUID_OK;
        if (eb == 1) {
            HIP_TRY(c, hipMemsetAsync(dst + b, (int)(pat.x & 0xFF), e - b, c->stream));
        } else {
            HIP_TRY(c, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(dst + b), (int)pat.x,
                                         (e - b) / 4, c->stream));
        }
        return FLUID_OK;
    };
    int rc = small(0, head);
    if (rc) return rc;
    rc = small(tail_begin, bytes);
    if (rc) return rc;
    HIP_TRY(c, hipGetLastError());
    return FLUID_OK;
}
int fill_image(fluid_ctx* c, int image, uint32_t pattern32) {
    const uint32_t v[4] = {pattern32, pattern32, pattern32, pattern32};
    return fill_image4(c, image, v);
}

uint32_t f32_bits(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}

// ---- 12_solve_pressure -----------------------------------------------------------------------------
// (a) one dispatch on the images themselves, any state: k12_plain / k12_zmarch
int launch_pressure(fluid_ctx* c, uint32_t is_even_iteration) {
    const GridK& g = c->g;
    // pressure.comp:71-75
    const float* pin = c->plane0<float>(is_even_iteration == 1 ? FLUID_IMG_PRESSURES_1
                                                               : FLUID_IMG_PRESSURES_2);
    float* pout = c->plane0<float>(is_even_iteration == 1 ? FLUID_IMG_PRESSURES_2
                                                          : FLUID_IMG_PRESSURES_1);
    const uint8_t* t = c->plane0<uint8_t>(FLUID_IMG_CELL_TYPES);
    const float* div = c->plane0<float>(FLUID_IMG_DIVERGENCES);
    int64_t variant = c->opt[FLUID_OPT_PRESSURE_KERNEL];
    const bool vec_ok = (g.W % 4 == 0);
    if (variant == 0 || variant >= 5) variant = (vec_ok && g.W >= 64) ? 2 : 1;
    if (variant >= 2 && !vec_ok) variant = 1;
    if (variant == 1)
        k12_launch_plain(c->stream, t, div, pin, pout, g, c->pk);
    else
        k12_launch_zmarch(c->stream, variant == 3 ? 4 : (variant == 4 ? 1 : 2), t, div, pin, pout, g,
                          c->pk);
    HIP_TRY(c, hipGetLastError());
    return FLUID_OK;
}

// (b) the loop section on working buffers -------------------------------------------------------------
bool fast_loop_possible(const fluid_ctx* c) {
    const int64_t v = c->opt[FLUID_OPT_PRESSURE_KERNEL];
    return c->g.W % 4 == 0 && (v == 0 || v >= 5);
}
// what an out-of-bounds neighbour contributes: type 0 is "not solid, not water" -> p_air, unless 0
// is the solid value (skipped -> 0) or the water value (its out-of-bounds pressure load is 0)
float oob_value(const fluid_ctx* c) {
    return (c->pk.t_solid == 0 || c->pk.t_water == 0) ? 0.0f : c->pk.p_air;
}
// mask, b_i and activity bricks, rebuilt only when CELL_TYPES / DIVERGENCES / parameters changed.
// `mask_only`: leave b_i for later (fluid_run_step builds the mask right after 06, before DIVERGENCES
// of this step exists, because the quiet-brick map needs the activity bricks).
int ensure_prepared(fluid_ctx* c, bool mask_only = false) {
    const bool want_rhs = !mask_only && !c->rhs_valid;
    // The mask pass of a step may leave the bricks alone that 02 ... 05 skipped (the one-step test,
    // quiet_bricks.h): their cell types are the previous step's (06 copied what was there already), they hold
    // no water — every mask byte is MASK_DRY as before, the brick stays inactive — and the previous mask
    // was complete (the test needs mask_valid).
    const uint8_t* unchanged = (mask_only && c->early_step) ? c->early() : nullptr;
    c->early_step = false;
    if (c->mask_valid && !want_rhs) return FLUID_OK;
    const GridK& g = c->g;
    const bool rebuilt_mask = !c->mask_valid;
    if (rebuilt_mask) {
        HIP_TRY(c, hipMemsetAsync(c->bricks(), 0, c->active_bytes, c->stream));
        HIP_TRY(c, hipMemsetAsync(c->flags() + 10, 0, 8, c->stream));  // x extent of the water
    }
    // fast_loop_possible() guarantees W % 4 == 0: four cells per thread
    GridK gb = c->g_bricks();
    // (the mask pass of a step runs between 06 and the streak update, when neither skipping flag is up: one
    // workgroup per brick layer all the same — 131 072 workgroups that test and leave cost 120 us at 512^3)
    if (rebuilt_mask && unchanged && c->active_bytes >= 4096) gb.zl = BRICK_Z;
    k12_launch_prepare_v4(c->stream, c->plane0<uint8_t>(FLUID_IMG_CELL_TYPES),
                          c->plane0<float>(FLUID_IMG_DIVERGENCES), c->mask0(), c->rhs0(),
                          c->bricks(), gb, c->pk, rebuilt_mask, want_rhs,
                          rebuilt_mask ? unchanged : c->quiet_or_null(), c->flags() + 10);
    HIP_TRY(c, hipGetLastError());
    c->mask_valid = true;
    if (want_rhs) c->rhs_valid = true;
    if (c->brick_count_host && rebuilt_mask) {
        k12_launch_count_bricks(c->stream, c->bricks(), g, c->flags() + 2, c->flags() + 10);
        HIP_TRY(c, hipMemcpyAsync(c->brick_count_host, c->flags() + 2, 28, hipMemcpyDeviceToHost,
                                  c->stream));
        c->box_pending = true;
        c->box.valid = false;  // fraction: the previous loop's, a hint until refresh_box()
    }
    return FLUID_OK;
}
// Wait for the brick summary of the current mask (one stream synchronisation per rebuilt mask, i.e.
// per step; a whole-grid loop of N sweeps follows) and turn it into the launch box.
int refresh_box(fluid_ctx* c) {
    if (!c->box_pending) return FLUID_OK;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->box_pending = false;
    const uint32_t* h = c->brick_count_host;
    int bx, by, bz;
    k12_brick_cells(bx, by, bz);
    c->box.valid = true;
    c->box.fraction = (float)h[0] / (float)c->active_bytes;
    c->box.y_lo = (int)h[1] * by;
    c->box.y_hi = std::min((int)h[2] * by, c->g.H);
    c->box.z_lo = (int)h[3] * bz;
    c->box.z_hi = std::min((int)h[4] * bz, c->g.Dl);
    c->box.x_lo = (int)h[5];  // cells
    c->box.x_hi = std::min((int)h[6], c->g.W);
    return FLUID_OK;
}
// Import / background cover the owned planes; ghost planes of the working buffers are filled by the
// caller's halo exchange (whole planes, so their non-water constants arrive with them).
int import_pressures(fluid_ctx* c, int image, int w) {
    // one pass also lays down the constants of the other two buffers when they are stale
    float* others[2] = {nullptr, nullptr};
    int n = 0;
    for (int i = 0; i < 3; i++)
        if (i != w) {
            if (!c->bg_valid[i]) others[n] = c->work0(i);
            n++;
        }
    k12_launch_import_v4(c->stream, c->plane0<uint8_t>(FLUID_IMG_CELL_TYPES), c->plane0<float>(image),
                         c->work0(w), others[0], others[1], c->g_bricks(), c->pk, c->quiet_or_null());
    HIP_TRY(c, hipGetLastError());
    c->bg_valid[0] = c->bg_valid[1] = c->bg_valid[2] = true;
    return FLUID_OK;
}
int ensure_background(fluid_ctx* c, int w) {
    if (c->bg_valid[w]) return FLUID_OK;
    k12_launch_background(c->stream, c->plane0<uint8_t>(FLUID_IMG_CELL_TYPES), c->work0(w), c->g,
                          c->pk, 0, c->g.Dl);
    HIP_TRY(c, hipGetLastError());
    c->bg_valid[w] = true;
    return FLUID_OK;
}
int export_pressures(fluid_ctx* c, int w_even, int w_odd) {
    k12_launch_export_v4(c->stream, c->plane0<uint8_t>(FLUID_IMG_CELL_TYPES),
                         w_even >= 0 ? c->work0(w_even) : nullptr,
                         w_odd >= 0 ? c->work0(w_odd) : nullptr,
                         c->plane0<float>(FLUID_IMG_PRESSURES_1),
                         c->plane0<float>(FLUID_IMG_PRESSURES_2), c->g_bricks(), c->pk,
                         c->mask_valid ? c->bricks() : nullptr);  // the bricks of these very cell types
    HIP_TRY(c, hipGetLastError());
    return FLUID_OK;
}

// one sweep work[src] -> work[dst] over the local planes [zlo, zhi)
int launch_work_sweep(fluid_ctx* c, int src, int dst, int zlo = 0, int zhi = -1) {
    if (zhi < 0) zhi = c->g.Dl;
    const int64_t variant = c->opt[FLUID_OPT_PRESSURE_KERNEL];
    const int ry = variant == 6 ? 4 : (variant == 5 ? 2 : 1);  // rows per wavefront
    k12_launch_canon(c->stream, ry, c->mask0(), c->rhs0(), c->work0(src), c->work0(dst), c->bricks(),
                     c->g, oob_value(c), zlo, zhi);
    HIP_TRY(c, hipGetLastError());
    return FLUID_OK;
}

// two or three sweeps in one pass (kernels_pressure_fused.h, kernels_pressure_fused3.h): work[src] = iterate j
// -> work[dst] = iterate j + sweeps, and iterate j + sweeps - 1 -> work[mid] when mid >= 0.
int launch_fused(fluid_ctx* c, int src, int dst, int mid, int part = FUSED_WHOLE, int part_lo = 0,
                 int part_hi = 0, hipStream_t stream = nullptr, int sweeps = 2) {
    const bool lo = c->g.z0 > 0, hi = c->g.z0 + c->g.Dl < c->g.Dg;  // neighbouring slabs
    auto* launch = sweeps == 3 ? &k12_launch_canon3 : &k12_launch_canon2;
    HIP_TRY(c, launch(stream ? stream : c->stream, c->mask0(), c->rhs0(), c->work0(src), c->work0(dst),
                      mid >= 0 ? c->work0(mid) : nullptr, c->bricks(), c->g, oob_value(c),
                      lo ? c->loop_halo : 0, hi ? c->loop_halo : 0, lo ? c->loop_aux_halo : 0,
                      hi ? c->loop_aux_halo : 0, c->box, part, part_lo, part_hi));
    HIP_TRY(c, hipGetLastError());
    return FLUID_OK;
}

// second stream for the EDGES launches of split passes (FLUID_OPT_EDGE_STREAM), created on first use
int ensure_edge_stream(fluid_ctx* c) {
    if (c->edge_stream) return FLUID_OK;
    int lo = 0, hi = 0;  // numerically lower = higher priority
    HIP_TRY(c, hipDeviceGetStreamPriorityRange(&lo, &hi));
    HIP_TRY(c, hipStreamCreateWithPriority(&c->edge_stream, hipStreamNonBlocking, hi));
    HIP_TRY(c, hipEventCreateWithFlags(&c->ev_pass_start, hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&c->ev_edges_done, hipEventDisableTiming));
    return FLUID_OK;
}

// ---- the loop as a small state machine: begin (import) / advance by 1 or 2 sweeps / end (export) ----
// Iterate k lives in a working buffer; dispatch k of the reference maps iterate k (read from
// PRESSURES_1 if k is even, else PRESSURES_2) to iterate k+1 in the other image.
int loop_begin(fluid_ctx* c) {
    int rc = ensure_prepared(c);
    if (rc == FLUID_OK) rc = import_pressures(c, FLUID_IMG_PRESSURES_1, 0);  // iterate 0 -> work[0]
    c->loop_open = rc == FLUID_OK;
    c->loop_cur = 0;
    c->loop_prev = -1;
    c->loop_k = 0;
    c->loop_halo = 0;
    c->loop_aux_halo = 0;
    c->loop_ghost_bg = false;
    c->loop_part_done = 0;
    return rc;
}
// a working buffer that is neither a nor b
int other_buffer(int a, int b) {
    for (int i = 0; i < 3; i++)
        if (i != a && i != b) return i;
    return -1;
}
// Advance by one sweep, or by two in one pass (kernels_pressure_fused.h).  With two, the
// intermediate iterate is kept only when `keep_mid` (the last pair of an even-length loop: iterate
// N-1 is what PRESSURES_2 must hold).  Returns the buffer written with the newest iterate.
int loop_advance(fluid_ctx* c, uint32_t sweeps, bool keep_mid, int* written, int part = FUSED_WHOLE,
                 int part_lo = 0, int part_hi = 0) {
    int rc = FLUID_OK;
    const int cur = c->loop_cur;
    if (c->is_slab && c->loop_halo < (int)sweeps)
        return c->fail(FLUID_ERR_INVALID_ARG,
                       "%u sweep(s) need %u valid ghost plane(s) of the newest iterate, %d left: "
                       "exchange halos and call fluid_pressure_loop_halo_exchanged",
                       sweeps, sweeps, c->loop_halo);
    // a pass in two launches (FUSED_EDGES / FUSED_INTERIOR, either order): the second one commits
    bool commit = true;
    if (part != FUSED_WHOLE) {
        if (sweeps < 2)
            return c->fail(FLUID_ERR_INVALID_ARG, "only a pass of two or three sweeps can be split into parts");
        if (c->loop_part_done == 0) {
            c->loop_part_done = part;
            c->loop_part_sweeps = sweeps;
            c->loop_part_keep = keep_mid;
            c->loop_part_lo = part_lo;
            c->loop_part_hi = part_hi;
            commit = false;
        } else if (c->loop_part_done == part || c->loop_part_keep != keep_mid ||
                   c->loop_part_lo != part_lo || c->loop_part_hi != part_hi ||
                   c->loop_part_sweeps != sweeps) {
            return c->fail(FLUID_ERR_INVALID_ARG,
                           "the second part of a split pass must be the other part with the same "
                           "arguments");
        } else {
            c->loop_part_done = 0;
        }
    } else if (c->loop_part_done != 0) {
        return c->fail(FLUID_ERR_INVALID_ARG, "a split pass is half done: launch its other part");
    }
    if (sweeps >= 2) {
        const int dst = other_buffer(cur, cur);
        const int mid = keep_mid ? other_buffer(cur, dst) : -1;
        rc = ensure_background(c, dst);
        if (rc == FLUID_OK && mid >= 0) rc = ensure_background(c, mid);
        // FLUID_OPT_EDGE_STREAM: the EDGES launch of a split pass goes to a second, high-priority stream so
        // that it runs beside the INTERIOR launch instead of before / after it.  Both read the source
        // buffer and write disjoint planes of the destination; both start after whatever preceded the
        // pass on the main stream (ev_pass_start); the main stream picks the edges up again at commit.
        hipStream_t where = nullptr;
        if (part != FUSED_WHOLE && c->opt[FLUID_OPT_EDGE_STREAM] != 0 && rc == FLUID_OK) {
            rc = ensure_edge_stream(c);
            if (rc == FLUID_OK && !commit) HIP_TRY(c, hipEventRecord(c->ev_pass_start, c->stream));
            if (rc == FLUID_OK && part == FUSED_EDGES) {
                HIP_TRY(c, hipStreamWaitEvent(c->edge_stream, c->ev_pass_start, 0));
                where = c->edge_stream;
            }
        }
        if (rc == FLUID_OK) rc = launch_fused(c, cur, dst, mid, part, part_lo, part_hi, where, (int)sweeps);
        if (where && rc == FLUID_OK) {
            HIP_TRY(c, hipEventRecord(c->ev_edges_done, c->edge_stream));
            c->edges_pending = true;
        }
        if (written) *written = dst;
        if (!commit) return rc;
        if (c->edges_pending) {
            HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_edges_done, 0));
            c->edges_pending = false;
        }
        c->loop_prev = mid;
        c->loop_cur = dst;
        c->loop_k += sweeps;
        // every sweep consumes a ghost plane of the iterate, every sweep but the first one of mask / b_i
        c->loop_halo = std::max(0, std::min(c->loop_halo - (int)sweeps, c->loop_aux_halo - ((int)sweeps - 1)));
    } else {
        const int dst = other_buffer(cur, c->loop_prev >= 0 ? c->loop_prev : cur);
        rc = ensure_background(c, dst);
        if (rc == FLUID_OK) rc = launch_work_sweep(c, cur, dst);
        c->loop_prev = cur;
        c->loop_cur = dst;
        c->loop_k += 1;
        c->loop_halo = 0;  // the one-sweep kernel writes owned planes only
    }
    if (written) *written = c->loop_cur;
    return rc;
}
int loop_end(fluid_ctx* c) {
    if (c->loop_part_done != 0)
        return c->fail(FLUID_ERR_INVALID_ARG, "a split pass is half done: launch its other part");
    c->loop_open = false;
    c->pressure_dispatch_index = c->loop_k;
    if (c->loop_k == 0) return FLUID_OK;
    // iterates N and N-1: the even one belongs in PRESSURES_1, the odd one in PRESSURES_2
    const bool n_even = (c->loop_k % 2u) == 0u;
    return export_pressures(c, n_even ? c->loop_cur : c->loop_prev,
                            n_even ? c->loop_prev : c->loop_cur);
}
bool fuse_enabled(const fluid_ctx* c) {
    return k12_canon2_supports(c->g) && c->opt[FLUID_OPT_JACOBI_FUSE] != 1;
}
// sweeps per pass over HBM the loop section may use on this context: a property of the grid and the options
// only (the ranks of a Z-slab run must come to the same number), FLUID_FUSED_T = 2 / 3 in the environment
// overrides the automatic choice (A/B runs)
int max_sweeps_per_pass(const fluid_ctx* c) {
    if (!fuse_enabled(c)) return 1;
    static const int forced = [] {
        const char* e = getenv("FLUID_FUSED_T");
        return e ? atoi(e) : 0;
    }();
    const int64_t opt = c->opt[FLUID_OPT_JACOBI_FUSE];
    const int want = opt == 2 ? 2 : (opt == 3 ? 3 : (forced == 2 || forced == 3 ? forced : 3));
    return want == 3 && k12_canon3_supports(c->g) ? 3 : 2;
}
// sweeps of the next launch of a loop with `left` sweeps to go, at most `most` (the ghost planes at hand) per
// launch: as many threes as possible, no single sweep at the end unless the loop has only one (4 = 2 + 2)
uint32_t next_launch_sweeps(uint32_t left, uint32_t most) {
    if (left <= 1 || most <= 1) return std::min<uint32_t>(left, 1);
    if (most == 2 || left == 2 || left == 4) return 2;
    return 3;
}

// FlowLoopPushConstantSection on working buffers (single context, no halo exchange)
int run_fast_loop(fluid_ctx* c, uint32_t iterations) {
    if (iterations == 0) return FLUID_OK;
    int rc = loop_begin(c);
    // a slab needs its halos between launches
    const uint32_t most = c->is_slab ? 1u : (uint32_t)max_sweeps_per_pass(c);
    // shaping the launches to the water costs one stream synchronisation per rebuilt mask; a caller that
    // wants fluid_run_step to stay fully asynchronous turns it off (FLUID_OPT_LAUNCH_BOX = 1)
    if (rc == FLUID_OK && most >= 2 && iterations >= 16 && c->opt[FLUID_OPT_LAUNCH_BOX] == 0)
        rc = refresh_box(c);
    const uint32_t most_here = (most == 3 && !k12_canon3_suits(c->g, c->box)) ? 2u : most;
    while (rc == FLUID_OK && c->loop_k < iterations) {
        const uint32_t left = iterations - c->loop_k;
        const uint32_t sweeps = next_launch_sweeps(left, most_here);
        rc = loop_advance(c, sweeps, sweeps >= 2 && left == sweeps, nullptr);
    }
    if (rc) return rc;
    return loop_end(c);
}

int slab_unsupported(fluid_ctx* c, const char* what) {
    return c->fail(FLUID_ERR_UNSUPPORTED,
                   "%s is not available on a Z-slab context yet (needs particle ownership / "
                   "wide velocity halos)",
                   what);
}


// ---- particles stored sorted by bin (kernels_particle_bins.h) -------------------------------------------
// FLUID_OPT_PARTICLE_SORT: 0 = on for whole-grid contexts with at least 4 M particle slots, 1 = off,
// 2 = on whatever the size, 3 = on and sorted again before every 01 (tests), 4 = on, sorted once and never
// again (tests: strays pile up).
// particle_owner_plane() of device_common.h, on the host
static inline int particle_owner_plane_host(float z, int Dg) {
    if (!(z > 0.0f)) return 0;
    if (z >= (float)Dg) return Dg - 1;
    return (int)z;
}
// ---- compact particle storage of a slab context -----------------------------------------------------------
void local_release(fluid_ctx* c) {
    for (int i = 0; i < 2; i++) {
        if (c->loc.buf[i]) (void)hipFree(c->loc.buf[i]), c->loc.buf[i] = nullptr;
        if (c->loc.pid[i]) (void)hipFree(c->loc.pid[i]), c->loc.pid[i] = nullptr;
        c->loc.capn[i] = 0;
    }
    if (c->loc.counters) (void)hipFree(c->loc.counters), c->loc.counters = nullptr;
    c->loc.n = c->loc.holes = c->loc.n_sorted = 0;
    c->loc.cur = 0;
}
// n and holes as the device has them (synchronises the stream)
int local_read(fluid_ctx* c) {
    if (!c->loc.on) return FLUID_OK;
    uint32_t v[2] = {0, 0};
    HIP_TRY(c, hipMemcpyAsync(v, c->loc.counters, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (v[0] > c->loc.capn[c->loc.cur])  // more entries than room: local_reserve() before every append rules it out
        return c->fail(FLUID_ERR_HIP, "particle storage overflow (%u entries, room for %llu)", v[0],
                       (unsigned long long)c->loc.capn[c->loc.cur]);
    c->loc.n = v[0];
    c->loc.holes = v[1];
    return FLUID_OK;
}
// set `which` of the arrays with room for `need` entries; keep = its first L.n entries survive a reallocation
int local_set_reserve(fluid_ctx* c, int which, uint64_t need, bool keep) {
    auto& L = c->loc;
    if (need <= L.capn[which]) return FLUID_OK;
    if (need >= 0xFFFFFFFFull) return c->fail(FLUID_ERR_UNSUPPORTED, "more than 2^32 particles in one slab");
    const uint64_t cap = std::max<uint64_t>(need + need / 4 + 65536, L.capn[which] + L.capn[which] / 2);
    void *nb = nullptr, *np = nullptr;
    if (hipMalloc(&nb, cap * 16) != hipSuccess || hipMalloc(&np, cap * 4) != hipSuccess) {
        (void)hipGetLastError();
        if (nb) (void)hipFree(nb);
        return c->fail(FLUID_ERR_OUT_OF_MEMORY,
                       "cannot grow the particle storage to %llu entries (%llu asked for; set %d holds %u of %llu)",
                       (unsigned long long)cap, (unsigned long long)need, which, L.n,
                       (unsigned long long)L.capn[which]);
    }
    if (keep && L.n) {
        HIP_TRY(c, hipMemcpyAsync(nb, L.buf[which], (uint64_t)L.n * 16, hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(np, L.pid[which], (uint64_t)L.n * 4, hipMemcpyDeviceToDevice, c->stream));
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));  // nothing in flight reads the old arrays any more
    if (L.buf[which]) (void)hipFree(L.buf[which]);
    if (L.pid[which]) (void)hipFree(L.pid[which]);
    L.buf[which] = static_cast<float4*>(nb);
    L.pid[which] = static_cast<uint32_t*>(np);
    if (L.capn[which]) L.grows++;
    L.capn[which] = cap;
    return FLUID_OK;
}
// room for `need` entries in the current set (the entries there are kept, in their order); grows by half at least
int local_reserve(fluid_ctx* c, uint64_t need) {
    auto& L = c->loc;
    if (!L.counters) {
        void* q = nullptr;
        HIP_TRY(c, hipMalloc(&q, 8));
        L.counters = static_cast<uint32_t*>(q);
        HIP_TRY(c, hipMemsetAsync(L.counters, 0, 8, c->stream));
    }
    return local_set_reserve(c, L.cur, need, true);
}
// squeeze the holes out when they are a quarter of the entries, or when `incoming` more entries would not fit.
// Sorted storage (ps.binned) keeps its holes — the next sort drops them — unless room is needed now.
int local_squeeze_if_needed(fluid_ctx* c, uint32_t incoming) {
    auto& L = c->loc;
    if (!L.on || L.holes == 0) return FLUID_OK;
    const bool tight = (uint64_t)L.n + incoming > L.capn[L.cur];
    const bool always = c->opt[FLUID_OPT_PARTICLE_SORT] == 3;  // test mode
    if (!tight && (c->ps.binned || (!always && (L.holes < 65536u || L.holes < L.n / 4u)))) return FLUID_OK;
    const int other = L.cur ^ 1;
    if (local_set_reserve(c, other, L.n, false) != FLUID_OK) return FLUID_OK;  // not an error: the holes stay
    const uint32_t n = L.n;
    HIP_TRY(c, hipMemsetAsync(L.counters, 0, 8, c->stream));
    const float4* src = L.buf[L.cur];
    const uint32_t* src_pid = L.pid[L.cur];
    L.cur = other;
    c->ps.binned = false;  // (the order survives a squeeze, the bins' segment starts do not)
    constexpr uint32_t per_block = OWNED_BLOCK * OWNED_PER_THREAD;
    hipLaunchKernelGGL(k_compact_squeeze, dim3((n + per_block - 1) / per_block), dim3(OWNED_BLOCK), 0, c->stream,
                       src, src_pid, n, c->compact());
    HIP_TRY(c, hipGetLastError());
    L.squeezes++;
    return local_read(c);
}
// 00_init_particles on a slab: count what it owns, make room, fill
int local_init(fluid_ctx* c) {
    auto& L = c->loc;
    if (c->particle_capacity >= 0xFFFFFFFFull)
        return c->fail(FLUID_ERR_UNSUPPORTED, "a Z-slab context numbers its particles with 32 bits");
    L.on = true;
    L.n = L.holes = L.n_sorted = 0;
    L.cur = 0;
    int rc = local_reserve(c, 0);
    if (rc) return rc;
    constexpr uint64_t per_block = (uint64_t)OWNED_BLOCK * OWNED_PER_THREAD;
    const dim3 grid((unsigned)((c->particle_capacity + per_block - 1) / per_block));
    HIP_TRY(c, hipMemsetAsync(L.counters, 0, 8, c->stream));
    CompactParticles count = c->compact();
    count.buf = nullptr;
    hipLaunchKernelGGL(k00_init_particles_compact, grid, dim3(OWNED_BLOCK), 0, c->stream, c->particle_capacity,
                       c->pk, c->g, count);
    HIP_TRY(c, hipGetLastError());
    uint32_t owned = 0;
    HIP_TRY(c, hipMemcpyAsync(&owned, L.counters, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    rc = local_reserve(c, owned);
    if (rc) return rc;
    HIP_TRY(c, hipMemsetAsync(L.counters, 0, 8, c->stream));
    hipLaunchKernelGGL(k00_init_particles_compact, grid, dim3(OWNED_BLOCK), 0, c->stream, c->particle_capacity,
                       c->pk, c->g, c->compact());
    HIP_TRY(c, hipGetLastError());
    return local_read(c);
}
// entries 01, 14 and the search for leavers look at
static inline uint64_t walk_entries(const fluid_ctx* c) {
    if (c->is_slab) return c->loc.on ? c->loc.n : 0;  // (before 00_init_particles / an upload: nothing stored)
    return c->particle_capacity;
}
static inline const uint32_t* walk_list(const fluid_ctx*) { return nullptr; }

// particles the sorted storage is about: every slot of a whole-grid context, the entries of a slab's storage
static inline uint64_t psort_n(const fluid_ctx* c) { return c->is_slab ? c->loc.n : c->particle_capacity; }
bool psort_wanted(const fluid_ctx* c) {
    const int64_t mode = c->opt[FLUID_OPT_PARTICLE_SORT];
    if (c->ps.failed || c->particle_capacity == 0 || c->particle_capacity >= (1ull << 32) || mode == 1) return false;
    if (c->is_slab && !c->loc.on) return false;
    return mode >= 2 || psort_n(c) >= (1ull << 22);
}
void psort_release(fluid_ctx* c) {
    auto& ps = c->ps;
    if (ps.alt) (void)hipFree(ps.alt);
    for (auto*& q : ps.slot_of)
        if (q) (void)hipFree(q), q = nullptr;
    if (ps.bin_count) (void)hipFree(ps.bin_count);
    if (ps.strays) (void)hipFree(ps.strays);
    if (ps.stray_count) (void)hipFree(ps.stray_count);
    if (ps.stray_host) (void)hipHostFree(ps.stray_host);
    if (ps.stray_ev) (void)hipEventDestroy(ps.stray_ev);
    ps.alt = nullptr;
    ps.bin_count = ps.bin_start = ps.cursor = ps.strays = ps.stray_count = ps.stray_host = nullptr;
    ps.stray_ev = nullptr;
}
// the particles are (again) in slot order in the arena's buffer: after 00_init_particles and uploads
void psort_reset(fluid_ctx* c) {
    c->ps.valid = false;
    c->ps.binned = false;
    c->ps.suspended = 0;
    c->ps.backoff = 64;
    c->ps.cur = 0;
    c->ps.stray_pending = false;
    c->ps.stray_steps = 0.0;
}
bool psort_alloc(fluid_ctx* c) {
    auto& ps = c->ps;
    if (c->is_slab) {
        // a slab sorts its compact storage between its two sets of arrays (Local); here: the bins (of its OWN
        // planes), the strays list (as many entries as the storage holds) and the small words
        auto& L = c->loc;
        void* q = nullptr;
        bool ok = true;
        if (!ps.bin_count) {
            ps.bk.nx = (c->g.W + PBIN_X - 1) / PBIN_X;
            ps.bk.ny = (c->g.H + PBIN_Y - 1) / PBIN_Y;
            ps.bk.nz = (c->g.Dl + PBIN_Z - 1) / PBIN_Z;
            ps.bk.bins = (uint32_t)ps.bk.nx * (uint32_t)ps.bk.ny * (uint32_t)ps.bk.nz;
            const uint64_t words = (uint64_t)ps.bk.bins + 2;
            if ((ok = hipMalloc(&q, 3 * words * 4) == hipSuccess)) {
                ps.bin_count = static_cast<uint32_t*>(q);
                ps.bin_start = ps.bin_count + words;
                ps.cursor = ps.bin_start + words;
            }
            if (ok && (ok = hipMalloc(&q, 8) == hipSuccess)) ps.stray_count = static_cast<uint32_t*>(q);
            if (ok && (ok = hipHostMalloc(&q, 8, hipHostMallocDefault) == hipSuccess))
                ps.stray_host = static_cast<uint32_t*>(q);
            if (ok) ok = hipEventCreateWithFlags(&ps.stray_ev, hipEventDisableTiming) == hipSuccess;
        }
        if (ok && L.strays_cap < L.capn[L.cur]) {
            if (ps.stray_pending) (void)hipStreamSynchronize(c->stream), ps.stray_pending = false;
            if (ps.strays) (void)hipStreamSynchronize(c->stream), (void)hipFree(ps.strays), ps.strays = nullptr;
            if ((ok = hipMalloc(&q, L.capn[L.cur] * 4) == hipSuccess)) {
                ps.strays = static_cast<uint32_t*>(q);
                L.strays_cap = L.capn[L.cur];
            }
        }
        if (!ok) {
            (void)hipGetLastError();
            psort_release(c);
            L.strays_cap = 0;
            ps.failed = true;
        }
        return ok;
    }
    if (ps.alt) return true;
    ps.bk.nx = (c->g.W + PBIN_X - 1) / PBIN_X;
    ps.bk.ny = (c->g.H + PBIN_Y - 1) / PBIN_Y;
    ps.bk.nz = (c->g.Dg + PBIN_Z - 1) / PBIN_Z;
    ps.bk.bins = (uint32_t)ps.bk.nx * (uint32_t)ps.bk.ny * (uint32_t)ps.bk.nz;
    const uint64_t n = c->particle_capacity, words = (uint64_t)ps.bk.bins + 2;
    void* q = nullptr;
    bool ok = hipMalloc(&q, n * 16) == hipSuccess;
    ps.alt = static_cast<float4*>(q);
    for (int i = 0; ok && i < 2; i++) {
        ok = hipMalloc(&q, n * 4) == hipSuccess;
        ps.slot_of[i] = ok ? static_cast<uint32_t*>(q) : nullptr;
    }
    if (ok && (ok = hipMalloc(&q, 3 * words * 4) == hipSuccess)) {
        ps.bin_count = static_cast<uint32_t*>(q);
        ps.bin_start = ps.bin_count + words;
        ps.cursor = ps.bin_start + words;
    }
    if (ok && (ok = hipMalloc(&q, n * 4) == hipSuccess)) ps.strays = static_cast<uint32_t*>(q);
    if (ok && (ok = hipMalloc(&q, 8) == hipSuccess)) ps.stray_count = static_cast<uint32_t*>(q);
    if (ok && (ok = hipHostMalloc(&q, 8, hipHostMallocDefault) == hipSuccess))
        ps.stray_host = static_cast<uint32_t*>(q);
    if (ok) ok = hipEventCreateWithFlags(&ps.stray_ev, hipEventDisableTiming) == hipSuccess;
    if (!ok) {  // not an error of the step: the particles simply stay in slot order
        (void)hipGetLastError();
        psort_release(c);
        ps.failed = true;
    }
    return ok;
}
int psort_sort(fluid_ctx* c) {
    auto& ps = c->ps;
    if (c->is_slab) {
        // the entries of the current set into the other one in bin order, their slots with them; the holes
        // (tombstones) are not taken along, so the count afterwards is what the scan added up
        auto& L = c->loc;
        const uint64_t n = L.n, per_block = (uint64_t)PSORT_THREADS * PSORT_PER_THREAD;
        const int other = L.cur ^ 1;
        int rc = local_set_reserve(c, other, n, false);  // (n, not the other set's room: the sets would leapfrog)
        if (rc) return rc;
        if (!psort_alloc(c)) return FLUID_OK;  // (strays list as large as the storage)
        HIP_TRY(c, hipMemsetAsync(ps.bin_count, 0, ((uint64_t)ps.bk.bins + 2) * 4, c->stream));
        if (n) {
            hipLaunchKernelGGL(k_pbin_histogram, dim3((unsigned)((n + per_block - 1) / per_block)), dim3(PSORT_THREADS),
                               0, c->stream, L.buf[L.cur], n, c->g, c->pk, ps.bk, ps.bin_count, true);
        }
        hipLaunchKernelGGL(k_pbin_scan, dim3(1), dim3(1024), 0, c->stream, ps.bin_count, ps.bk.bins + 1, ps.bin_start,
                           ps.cursor);
        if (n) {
            const uint64_t per_block2 = (uint64_t)PSORT_THREADS * PSCATTER_PER_THREAD;
            hipLaunchKernelGGL(k_pbin_scatter, dim3((unsigned)((n + per_block2 - 1) / per_block2)), dim3(PSORT_THREADS),
                               0, c->stream, L.buf[L.cur], L.pid[L.cur], n, c->g, c->pk, ps.bk, ps.cursor, L.buf[other],
                               L.pid[other], true);
        }
        HIP_TRY(c, hipGetLastError());
        // entries now: bin_start[bins + 1]; no holes
        HIP_TRY(c, hipMemcpyAsync(L.counters, ps.bin_start + ps.bk.bins + 1, 4, hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(c, hipMemsetAsync(L.counters + 1, 0, 4, c->stream));
        L.cur = other;
        rc = local_read(c);
        if (rc) return rc;
        L.n_sorted = L.n;
        ps.binned = true;
        ps.steps_since_sort = 0;
        ps.sorts++;
        ps.stray_pending = false;
        ps.stray_steps = 0.0;
        return FLUID_OK;
    }
    const uint64_t n = c->particle_capacity, per_block = (uint64_t)PSORT_THREADS * PSORT_PER_THREAD;
    const unsigned blocks = (unsigned)((n + per_block - 1) / per_block);
    const float4* in = c->particles();
    const uint32_t* slot_in = ps.valid ? ps.slot_of[ps.cur] : nullptr;
    const int dst = ps.cur ^ 1;
    float4* out = dst ? ps.alt : c->particles_home();
    HIP_TRY(c, hipMemsetAsync(ps.bin_count, 0, ((uint64_t)ps.bk.bins + 2) * 4, c->stream));
    hipLaunchKernelGGL(k_pbin_histogram, dim3(blocks), dim3(PSORT_THREADS), 0, c->stream, in, n, c->g, c->pk,
                       ps.bk, ps.bin_count, false);
    hipLaunchKernelGGL(k_pbin_scan, dim3(1), dim3(1024), 0, c->stream, ps.bin_count, ps.bk.bins + 1,
                       ps.bin_start, ps.cursor);
    const uint64_t per_block2 = (uint64_t)PSORT_THREADS * PSCATTER_PER_THREAD;
    hipLaunchKernelGGL(k_pbin_scatter, dim3((unsigned)((n + per_block2 - 1) / per_block2)), dim3(PSORT_THREADS), 0,
                       c->stream, in, slot_in, n, c->g,
                       c->pk, ps.bk, ps.cursor, out, ps.slot_of[dst], false);
    HIP_TRY(c, hipGetLastError());
    ps.cur = dst;
    ps.valid = true;
    ps.binned = true;
    ps.steps_since_sort = 0;
    ps.sorts++;
    ps.stray_pending = false;
    ps.stray_steps = 0.0;
    return FLUID_OK;
}
// Before 01_update_densities: sort if the storage is not sorted yet, or when the strays have cost as much
// as a sort would.  Measured per particle slot (512^3 dam break and full tank): 01 + 14 cost 30 ps in slot
// order and 10-15 ps freshly sorted; a stray costs 01 a global atomic and 14 its taps from global memory,
// about 75 ps; a sort 21-34 ps (two passes over the buffer).  So the storage is sorted again once the
// fractions of strays of the steps since the last sort add up to 0.3 — with strays growing linearly, r per
// step, that is the interval that minimises sort + stray time, sqrt(2 * 34 * 75 * r) = 72 sqrt(r) ps per step.
// Beyond r = 3 % per step that overhead eats what sorting saves (the collapse of the dam break is such a
// flow): the storage goes back to slot order and 01 and 14 run their slot-order kernels for the next 64 steps,
// after which sorting is tried again (128, 256 ... steps if it keeps failing).  The count of the last 01 is read one step late, without waiting for it.
// back to slot order in the arena's buffer (sorting switched off, or suspended)
int psort_to_slot_order(fluid_ctx* c) {
    auto& ps = c->ps;
    if (c->is_slab) {  // a slab's storage has no slot order to go back to: the bins are simply not used
        ps.binned = false;
        ps.stray_pending = false;
        ps.stray_steps = 0.0;
        return FLUID_OK;
    }
    if (ps.valid) {
        hipLaunchKernelGGL(k_pbin_to_slot_order, dim3((unsigned)((c->particle_capacity + 255) / 256)),
                           dim3(256), 0, c->stream, c->particles(), ps.slot_of[ps.cur],
                           c->particle_capacity, ps.cur ? c->particles_home() : ps.alt);
        HIP_TRY(c, hipGetLastError());
        if (!ps.cur) {  // the slot-ordered copy went to alt: bring it home
            HIP_TRY(c, hipMemcpyAsync(c->particles_home(), ps.alt, c->particle_capacity * 16,
                                      hipMemcpyDeviceToDevice, c->stream));
        }
    }
    const uint32_t suspended = ps.suspended, backoff = ps.backoff;
    psort_reset(c);
    ps.suspended = suspended;
    ps.backoff = backoff;
    return FLUID_OK;
}
int psort_before_count(fluid_ctx* c) {
    auto& ps = c->ps;
    if (!psort_wanted(c) || !psort_alloc(c)) {
        ps.suspended = 0;
        return psort_to_slot_order(c);
    }
    if (ps.stray_pending && hipEventQuery(ps.stray_ev) == hipSuccess) {
        ps.stray_steps += (double)ps.stray_host[1] / (double)std::max<uint64_t>(psort_n(c), 1);
        ps.stray_pending = false;
    }
    const int64_t mode = c->opt[FLUID_OPT_PARTICLE_SORT];
    ps.steps_since_sort++;
    if (ps.suspended) {
        if (--ps.suspended) return FLUID_OK;
        return psort_sort(c);  // try again
    }
    const bool again = mode == 3 || (mode != 4 && ps.stray_steps >= 0.3);
    if ((ps.valid || c->is_slab) && ps.binned && again && mode != 3) {
        // strays per step, from the triangle the fractions have summed to: sum = r T^2 / 2
        const double T = (double)ps.steps_since_sort;
        if (2.0 * ps.stray_steps / (T * T) > 0.03) {
            ps.suspended = ps.backoff;
            ps.backoff = std::min<uint32_t>(2 * ps.backoff, 2048);
            return psort_to_slot_order(c);  // the slot-order kernels are fastest on slot order
        }
        if (T >= 32.0) ps.backoff = 64;  // a calm spell
    }
    if ((!ps.valid && !c->is_slab) || !ps.binned || again) return psort_sort(c);
    return FLUID_OK;
}
// 01 on the sorted storage; `marks` = pbricks() or null
int psort_count(fluid_ctx* c, uint32_t* dens, uint8_t* marks, const BrickK& bk) {
    auto& ps = c->ps;
    const unsigned blocks = (unsigned)std::min<uint64_t>((uint64_t)ps.bk.bins + 1, 256 * 64);
    HIP_TRY(c, hipMemsetAsync(ps.stray_count, 0, 8, c->stream));
    if (c->dens_zero)
        hipLaunchKernelGGL(k01_binned<false>, dim3(blocks), dim3(256), 0, c->stream, c->particles(),
                           ps.bin_start, ps.bk, dens, c->g, c->pk, marks, bk, ps.strays, ps.stray_count);
    else
        hipLaunchKernelGGL(k01_binned<true>, dim3(blocks), dim3(256), 0, c->stream, c->particles(),
                           ps.bin_start, ps.bk, dens, c->g, c->pk, marks, bk, ps.strays, ps.stray_count);
    hipLaunchKernelGGL(k01_binned_strays, dim3(1024), dim3(256), 0, c->stream, ps.strays, ps.stray_count, dens,
                       c->g, marks, bk);
    if (c->is_slab && c->loc.n > c->loc.n_sorted) {
        // what the slab has adopted since the sort sits behind the sorted entries, in no order: the slot-order
        // kernel (it adds with atomics, after the plain stores of the bins in stream order)
        const uint64_t per_block = (uint64_t)K01_THREADS * K01_PER_THREAD, n = c->loc.n - c->loc.n_sorted;
        hipLaunchKernelGGL(k01_update_densities, dim3((unsigned)((n + per_block - 1) / per_block)), dim3(K01_THREADS), 0,
                           c->stream, c->particles() + c->loc.n_sorted, n, dens, c->g, c->pk, marks, bk,
                           (const uint32_t*)nullptr);
    }
    HIP_TRY(c, hipGetLastError());
    if (!ps.stray_pending) {
        HIP_TRY(c, hipMemcpyAsync(ps.stray_host, ps.stray_count, 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipEventRecord(ps.stray_ev, c->stream));
        ps.stray_pending = true;
    }
    return FLUID_OK;
}

// internal ids of the grouped passes, past the public section ids
enum : int {
    STEP_0405_EXTRAPOLATE = FLUID_SECTION_COUNT + 1,
    STEP_01A_CLEAR_WHERE_PARTICLES_WERE,
    STEP_01_UPDATE_DENSITIES_MARK_BRICKS,
    STEP_0405_APPLY,
    STEP_0708_ADVECT_FORCES,
    STEP_091011_SOLIDS_DIVERGENCE,
};

// ---- one section --------------------------------------------------------------------------------
int run_section_impl(fluid_ctx* c, int section) {
    const GridK& g = c->g;
    const ParamsK& pk = c->pk;
    const dim3 grid = cell_grid(g), block = cell_block();
    uint8_t* T = c->plane0<uint8_t>(FLUID_IMG_CELL_TYPES);
    uint8_t* newT = c->plane0<uint8_t>(FLUID_IMG_NEW_CELL_TYPES);
    float4* V1 = c->plane0<float4>(FLUID_IMG_VELOCITIES_1);
    float4* V2 = c->plane0<float4>(FLUID_IMG_VELOCITIES_2);
    uint32_t* dens = c->plane0<uint32_t>(FLUID_IMG_PARTICLE_DENSITIES_IMG);
    const unsigned pblocks = (unsigned)((c->particle_capacity + 255) / 256);
    BrickK bk;  // activity / quiet bricks (quiet_bricks.h)
    k12_brick_dims(g.W, g.H, g.Dl, bk.nbx, bk.nby, bk.nbz);
    // kernels that can skip quiet bricks: one workgroup per brick row while skipping is on
    const int qchunks = c->quiet_in_use ? 4 : 1;
    // ... and per brick layer (GridK::zl): gq is the grid those kernels get, zgroups their grid's z extent
    const GridK gq = c->g_bricks();
    const unsigned zgroups = (unsigned)((g.Dl + gq.zl - 1) / gq.zl);
    const dim3 q4grid((g.W / 4 + 63) / 64, grid.y, zgroups);  // four cells per thread
    // the passes with real work per cell (09+10+11, 13) keep four times as many workgroups of a quarter of a
    // brick layer each: with whole layers the few active bricks finish later than the dispatch saves
    GridK gqh = gq;
    gqh.zl = gq.zl > 1 ? BRICK_Z / 4 : 1;
    const dim3 qgrid((g.W + 64 * qchunks - 1) / (64 * qchunks), grid.y, (unsigned)((g.Dl + gqh.zl - 1) / gqh.zl));

    switch (section) {
        case FLUID_SEC_INIT_CLEAR_VELOCITIES_1:
            c->v1_w_zero = true;
            return fill_image(c, FLUID_IMG_VELOCITIES_1, 0u);
        case FLUID_SEC_INIT_CLEAR_CELL_TYPES:
            c->touched(FLUID_IMG_CELL_TYPES);
            return fill_image(c, FLUID_IMG_CELL_TYPES, pk.t_inactive);
        case FLUID_SEC_00_INIT_PARTICLES:
            if (c->particle_capacity == 0) return FLUID_OK;
            psort_reset(c);  // slot order, in the arena's buffer
            if (c->is_slab) return local_init(c);  // what this slab owns, compactly
            hipLaunchKernelGGL(k00_init_particles, dim3(pblocks), dim3(256), 0, c->stream,
                               c->particles(), c->particle_capacity, pk);
            break;
        case FLUID_SEC_01A_CLEAR_PARTICLE_DENSITIES: {
            const int rc = fill_image(c, FLUID_IMG_PARTICLE_DENSITIES_IMG, 0u);
            c->dens_zero = rc == FLUID_OK;
            return rc;
        }
        case STEP_01A_CLEAR_WHERE_PARTICLES_WERE:  // quiet_bricks.h; bricks() and pbricks() still hold the
                                                   // previous step's water / particle maps
        {
            GridK gb = g;  // brick-wise like the passes of g_bricks() (which this one precedes)
            gb.zl = (c->active_bytes >= 4096 || c->opt[FLUID_OPT_QUIET_BRICKS] == 2) ? BRICK_Z : 1;
            hipLaunchKernelGGL(k_clear_density_where_particles_were,
                               dim3(q4grid.x, q4grid.y, (unsigned)((g.Dl + gb.zl - 1) / gb.zl)), block, 0,
                               c->stream, dens, gb, c->bricks(), c->pbricks(), bk);
        }
            c->dens_zero = true;  // the other bricks have not been counted into since the last full clear
            break;
        case STEP_01_UPDATE_DENSITIES_MARK_BRICKS: {
            const int nb = (int)c->active_bytes;
            HIP_TRY(c, hipMemsetAsync(c->pbricks(), 0, c->active_bytes, c->stream));
            if (c->particle_capacity != 0) {
                int rc = psort_before_count(c);
                if (rc) return rc;
                if (c->ps.binned) {
                    rc = psort_count(c, dens, c->pbricks(), bk);
                    if (rc) return rc;
                } else {
                    const uint64_t per_block = (uint64_t)K01_THREADS * K01_PER_THREAD, n = walk_entries(c);
                    const unsigned blocks = (unsigned)((n + per_block - 1) / per_block);
                    if (blocks)
                        hipLaunchKernelGGL(k01_update_densities, dim3(blocks), dim3(K01_THREADS), 0, c->stream,
                                           c->particles(), n, dens, g, pk, c->pbricks(), bk,
                                           walk_list(c));
                }
            }
            c->dens_zero = false;
            c->pbricks_valid = true;
            if (c->early_wanted)
                hipLaunchKernelGGL(k_update_early_quiet, dim3((nb + 255) / 256), dim3(256), 0, c->stream,
                                   c->bricks(), c->pbricks(), c->early(), bk, c->brick_edges(false));
            break;
        }
        case FLUID_SEC_01_UPDATE_DENSITIES: {
            c->pbricks_valid = false;  // counted without marking the bricks
            if (c->particle_capacity == 0) return FLUID_OK;
            int rc = psort_before_count(c);
            if (rc) return rc;
            if (c->ps.binned) {
                rc = psort_count(c, dens, nullptr, bk);
                if (rc) return rc;
            } else {
                const uint64_t per_block = (uint64_t)K01_THREADS * K01_PER_THREAD, n = walk_entries(c);
                const unsigned blocks = (unsigned)((n + per_block - 1) / per_block);
                if (blocks)
                    hipLaunchKernelGGL(k01_update_densities, dim3(blocks), dim3(K01_THREADS), 0, c->stream,
                                       c->particles(), n, dens, g, pk, (uint8_t*)nullptr, bk,
                                       walk_list(c));
            }
            c->dens_zero = false;
            break;
        }
        case FLUID_SEC_02_UPDATE_WATER:
            if (g.W % 4 == 0)
                hipLaunchKernelGGL(k02_update_water_v4, q4grid, block, 0, c->stream, dens,
                                   newT, gq, pk, c->early_or_null(), bk);
            else
                hipLaunchKernelGGL(k02_update_water, grid, block, 0, c->stream, dens, newT, g, pk);
            break;
        case FLUID_SEC_03_UPDATE_AIR:
            if (g.W % 4 == 0)
                hipLaunchKernelGGL(k03_update_air_v4, q4grid, block, 0, c->stream, newT, gq,
                                   pk, c->early_or_null(), bk);
            else
                hipLaunchKernelGGL(k03_update_air, grid, block, 0, c->stream, newT, g, pk);
            break;
        case FLUID_SEC_04_COMPUTE_EXTRAPOLATED_VELOCITIES:
            hipLaunchKernelGGL(k04_extrapolated, grid, block, 0, c->stream, T, V1, V2, g, pk);
            break;
        case FLUID_SEC_05_SET_EXTRAPOLATED_VELOCITIES:
            hipLaunchKernelGGL(k05_set_extrapolated, grid, block, 0, c->stream, newT, T, V2, V1, g,
                               pk);
            c->v1_w_zero = true;
            break;
        // grouped passes of fluid_run_step (kernels_step_fused.h); not part of the public section ids
        case STEP_0405_EXTRAPOLATE:
            hipLaunchKernelGGL(k0405_extrapolate, q4grid, block, 0, c->stream, T, newT, V1, V2,
                               gq, pk, c->early_or_null(), bk);
            break;
        case STEP_0405_APPLY:
            hipLaunchKernelGGL(k0405_apply, q4grid, block, 0, c->stream, T, newT, V2, V1, gq, pk,
                               c->early_or_null(), bk);
            break;
        case STEP_0708_ADVECT_FORCES:
            if (c->opt[FLUID_OPT_ADVECT_KERNEL] == 1)
                hipLaunchKernelGGL(k07_advect<true>, qgrid, block, 0, c->stream, T, V1, V2, gqh, pk,
                                   c->flags(), c->quiet_or_null(), bk, qchunks);
            else
            {
                // A workgroup marches a brick layer of ONE 64-cell chunk of a row: with all four chunks of the
                // brick row in one workgroup (as the other skipping passes have it) a sparse scene's launch lasts
                // as long as 64 plane steps in a row take (0.41 ms at 512^3); four times the workgroups cost 10 us
                // of dispatch.  FLUID_ADVECT_XCHUNKS (dev) overrides.
                static const int forced = getenv("FLUID_ADVECT_XCHUNKS") ? atoi(getenv("FLUID_ADVECT_XCHUNKS")) : 0;
                const int xch = forced > 0 ? forced : 1;
                hipLaunchKernelGGL(k07_advect_tiled<true>,
                                   dim3((g.W + 64 * xch - 1) / (64 * xch), grid.y, (g.Dl + K07_ZM - 1) / K07_ZM),
                                   block, 0, c->stream, T, V1, V2, g, pk, c->flags(), c->quiet_or_null(), bk,
                                   xch, c->quiet_in_use ? c->bricks() : (const uint8_t*)nullptr);
            }
            break;
        case STEP_091011_SOLIDS_DIVERGENCE:
            c->touched(FLUID_IMG_DIVERGENCES);
            c->v1_w_zero = false;
        {
            // the loop's b_i rides along when the working-buffer loop will run on it (the same bricks are skipped
            // by this pass and by the b_i pass: where DIVERGENCES does not change, b_i does not either)
            float* rhs = fast_loop_possible(c) && c->solver != FLUID_SOLVER_RED_BLACK_SOR ? c->rhs0() : nullptr;
            hipLaunchKernelGGL(k091011_solids_divergence, qgrid, block, 0, c->stream, T, V2, V1,
                               c->plane0<float>(FLUID_IMG_DIVERGENCES), gqh, pk, c->quiet_or_null(), bk,
                               qchunks, rhs);
            if (rhs) c->rhs_valid = true;
            break;
        }
        case FLUID_SEC_06_UPDATE_CELL_TYPES:
            c->touched(FLUID_IMG_CELL_TYPES);
            // one ghost plane per side rides along: on a slab it holds the neighbour's new types
            // (exchanged after 03), at a domain face it is zero in both images
            if (c->early_step && g.W % 4 == 0 && c->active_bytes >= 4096) {
                // bricks that 02 and 03 skipped hold the same types in both images: copy the others
                GridK ge = g;
                ge.zl = BRICK_Z;
                hipLaunchKernelGGL(k06_copy_types_v4, dim3((g.W / 4 + 63) / 64, grid.y, (g.Dl + BRICK_Z - 1) / BRICK_Z),
                                   block, 0, c->stream, newT, T, ge, c->early(), bk);
                if (g.z0 > 0)
                    HIP_TRY(c, hipMemcpyAsync(T - g.plane, newT - g.plane, g.plane, hipMemcpyDeviceToDevice, c->stream));
                if (g.z0 + g.Dl < g.Dg)
                    HIP_TRY(c, hipMemcpyAsync(T + c->owned_cells(), newT + c->owned_cells(), g.plane,
                                              hipMemcpyDeviceToDevice, c->stream));
                break;
            }
            HIP_TRY(c, hipMemcpyAsync(T - g.plane, newT - g.plane, c->owned_cells() + 2 * g.plane,
                                      hipMemcpyDeviceToDevice, c->stream));
            return FLUID_OK;
        case FLUID_SEC_07_ADVECT:
            if (c->opt[FLUID_OPT_ADVECT_KERNEL] == 1)
                hipLaunchKernelGGL(k07_advect<false>, grid, block, 0, c->stream, T, V1, V2, g, pk,
                                   c->flags(), (const uint8_t*)nullptr, bk, 1);
            else
                hipLaunchKernelGGL(k07_advect_tiled<false>, dim3(grid.x, grid.y, (g.Dl + K07_ZM - 1) / K07_ZM),
                                   block, 0, c->stream, T, V1, V2, g, pk, c->flags(), (const uint8_t*)nullptr,
                                   bk, 1, (const uint8_t*)nullptr);
            break;
        case FLUID_SEC_08_FORCES:
            hipLaunchKernelGGL(k08_forces, grid, block, 0, c->stream, T, V2, g, pk);
            break;
        case FLUID_SEC_09_DIFFUSE:
            // intended mode is a 7-point stencil on VELOCITIES_2: on a Z slab the caller exchanges one
            // ghost plane of VELOCITIES_2 per side before this section
            if (c->diffuse_mode == FLUID_DIFFUSE_INTENDED)
                hipLaunchKernelGGL(k09_diffuse<true>, grid, block, 0, c->stream, T, V2, V1, g, pk);
            else
                hipLaunchKernelGGL(k09_diffuse<false>, grid, block, 0, c->stream, T, V2, V1, g, pk);
            c->v1_w_zero = true;
            break;
        case FLUID_SEC_10_SOLIDS:
            hipLaunchKernelGGL(k10_solids, grid, block, 0, c->stream, T, V1, g, pk);
            c->v1_w_zero = false;
            break;
        case FLUID_SEC_11_COMPUTE_DIVERGENCE:
            c->touched(FLUID_IMG_DIVERGENCES);
            hipLaunchKernelGGL(k11_divergence, grid, block, 0, c->stream, V1,
                               c->plane0<float>(FLUID_IMG_DIVERGENCES), g);
            break;
        case FLUID_SEC_12A_CLEAR_PRESSURES_1:
        case FLUID_SEC_12B_CLEAR_PRESSURES_2: {
            c->pressure_dispatch_index = 0;
            const int img = section == FLUID_SEC_12A_CLEAR_PRESSURES_1 ? FLUID_IMG_PRESSURES_1
                                                                       : FLUID_IMG_PRESSURES_2;
            if (!c->quiet_in_use) return fill_image(c, img, f32_bits(pk.p_air));
            hipLaunchKernelGGL(k_fill_f32_unless_quiet, q4grid, block, 0, c->stream,
                               c->plane0<float>(img), pk.p_air, gq, c->quiet(), bk);
            break;
        }
        case FLUID_SEC_12_SOLVE_PRESSURE: {
            if (c->solver == FLUID_SOLVER_RED_BLACK_SOR) {  // one iteration, in place on PRESSURES_1
                if (c->is_slab) return slab_unsupported(c, "the red-black SOR solver");
                float* p1 = c->plane0<float>(FLUID_IMG_PRESSURES_1);
                for (int colour = 0; colour < 2; colour++)
                    k12_launch_sor_colour(c->stream, T, c->plane0<float>(FLUID_IMG_DIVERGENCES), p1, g,
                                          pk, c->sor_omega, colour);
                c->pressure_dispatch_index++;
                break;
            }
            const uint32_t even = (c->pressure_dispatch_index % 2u) == 0u ? 1u : 0u;
            c->pressure_dispatch_index++;
            return launch_pressure(c, even);
        }
        case FLUID_SEC_13_FIX_DIVERGENCE:
            hipLaunchKernelGGL(k13_fix_divergence, qgrid, block, 0, c->stream, T,
                               c->plane0<float>(FLUID_IMG_PRESSURES_2), V1, gqh, pk, c->quiet_or_null(),
                               bk, qchunks);
            c->v1_w_zero = true;
            break;
        // ---- surface-prep passes (kernels_surface.h)
        case FLUID_SEC_INIT_CLEAR_DETAILED_DENSITIES_INERTIA:
        case FLUID_SEC_14A_CLEAR_DETAILED_DENSITIES:
        case FLUID_SEC_15_UPDATE_DETAILED_DENSITIES:
        case FLUID_SEC_16_COMPUTE_DETAILED_DENSITIES_INERTIA:
        case FLUID_SEC_17_COMPUTE_FLOAT_DENSITIES:
        case FLUID_SEC_18_DIFFUSE_FLOAT_DENSITIES: {
            if (!c->surface)
                return c->fail(FLUID_ERR_UNSUPPORTED,
                               "section %d is a surface-prep pass: create the context with "
                               "fluid_create_info.surface_prep", section);
            const SurfK& s = c->sk;
            const dim3 sgrid((s.W + 63) / 64, (s.H + 3) / 4, s.D);
            const dim3 sgrid4((s.W / 4 + 63) / 64, (s.H + 3) / 4, s.D);  // four cells per thread
            uint32_t* det = c->surf<uint32_t>(FLUID_IMG_DETAILED_DENSITIES_IMG);
            uint32_t* inertia = c->surf<uint32_t>(FLUID_IMG_DETAILED_DENSITIES_INERTIA_IMG);
            float* f1 = c->surf<float>(FLUID_IMG_PARTICLE_DENSITIES_FLOAT_1);
            float* f2 = c->surf<float>(FLUID_IMG_PARTICLE_DENSITIES_FLOAT_2);
            if (section == FLUID_SEC_INIT_CLEAR_DETAILED_DENSITIES_INERTIA)
                return fill_image(c, FLUID_IMG_DETAILED_DENSITIES_INERTIA_IMG, 0u);
            if (section == FLUID_SEC_14A_CLEAR_DETAILED_DENSITIES)
                return fill_image(c, FLUID_IMG_DETAILED_DENSITIES_IMG, 0u);
            if (section == FLUID_SEC_15_UPDATE_DETAILED_DENSITIES) {
                if (c->particle_capacity == 0) return FLUID_OK;
                hipLaunchKernelGGL(k15_update_detailed_densities, dim3(pblocks), dim3(256), 0, c->stream,
                                   c->particles(), c->particle_capacity, det, s, pk.active_w);
            } else if (section == FLUID_SEC_16_COMPUTE_DETAILED_DENSITIES_INERTIA) {
                InertiaK k;
                k.max_inertia = (uint32_t)c->params.max_inertia;
                k.increase_filled = (uint32_t)c->params.inertia_increase_filled;
                k.increase_neighbour = (uint32_t)c->params.inertia_increase_neighbour;
                k.decrease = (uint32_t)c->params.inertia_decrease;
                k.required_hits = c->params.required_neighbour_hits;
                k.increase_neighbour_i = c->params.inertia_increase_neighbour;
                if (s.W % 4 == 0)  // fuse17: fluid_run_step writes FLOAT_1 from the same pass
                    hipLaunchKernelGGL(k16_detailed_densities_inertia_v4, sgrid4, block, 0, c->stream,
                                       det, inertia, c->surface_fuse17 ? f1 : (float*)nullptr, s, k,
                                       c->params.dens_division_coefficient);
                else
                    hipLaunchKernelGGL(k16_detailed_densities_inertia, sgrid, block, 0, c->stream, det,
                                       inertia, s, k);
                if (c->surface_fuse17 && s.W % 4 == 0) c->surface_dispatch_index = 0;
            } else if (section == FLUID_SEC_17_COMPUTE_FLOAT_DENSITIES) {
                c->surface_dispatch_index = 0;
                hipLaunchKernelGGL(k17_float_densities, sgrid, block, 0, c->stream, inertia, f1, s,
                                   c->params.dens_division_coefficient);
            } else {  // one dispatch of the 18 loop; even dispatches read FLOAT_1 and write FLOAT_2
                const bool even = (c->surface_dispatch_index % 2u) == 0u;
                c->surface_dispatch_index++;
                if (s.W % 4 == 0 && c->opt[FLUID_OPT_SURFACE_KERNEL] == 0) {
                    const int zchunk = 32;  // planes per workgroup: 2 extra plane loads per 32
                    hipLaunchKernelGGL(k18_diffuse_float_densities_zmarch,
                                       dim3((s.W / 4 + 63) / 64, (s.H + K18_ROWS - 1) / K18_ROWS,
                                            (s.D + zchunk - 1) / zchunk),
                                       dim3(64, K18_ROWS, 1), 0, c->stream, T, even ? f1 : f2,
                                       even ? f2 : f1, s, c->params.dens_diffuse_k, pk.t_solid, zchunk);
                } else if (s.W % 4 == 0)
                    hipLaunchKernelGGL(k18_diffuse_float_densities_v4, sgrid4, block, 0, c->stream, T,
                                       even ? f1 : f2, even ? f2 : f1, s, c->params.dens_diffuse_k,
                                       pk.t_solid);
                else
                    hipLaunchKernelGGL(k18_diffuse_float_densities, sgrid, block, 0, c->stream, T,
                                       even ? f1 : f2, even ? f2 : f1, s, c->params.dens_diffuse_k,
                                       pk.t_solid);
            }
            break;
        }
        case FLUID_SEC_14_PARTICLES:
            if (c->particle_capacity == 0) return FLUID_OK;
            if (c->ps.binned) {
                // few full bins (a sparse scene): two workgroups share a bin, each staging its tile (512^3 dam
                // break: 0.36 -> 0.26 ms; the full tank loses 8 % that way)
                const uint32_t parts = psort_n(c) / (8u * PBIN_CELLS) < 256 * 32 ? 2u : 1u;
                const unsigned blocks =
                    (unsigned)std::min<uint64_t>(((uint64_t)c->ps.bk.bins + 1) * parts, 256 * 64);
                hipLaunchKernelGGL(k14_binned, dim3(blocks), dim3(256), 0, c->stream, V1, c->particles(),
                                   c->ps.bin_start, c->ps.bk, g, pk, c->flags(), parts);
                if (c->is_slab && c->loc.n > c->loc.n_sorted) {  // adopted since the sort: behind the bins, unordered
                    const uint64_t n = c->loc.n - c->loc.n_sorted;
                    hipLaunchKernelGGL(k14_particles, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, V1,
                                       c->particles() + c->loc.n_sorted, n, g, pk, c->flags(),
                                       (const uint32_t*)nullptr);
                }
            } else {
                const uint64_t n = walk_entries(c);
                if (n)
                    hipLaunchKernelGGL(k14_particles, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream,
                                       V1, c->particles(), n, g, pk, c->flags(), walk_list(c));
            }
            break;
        default:
            return c->fail(FLUID_ERR_INVALID_ARG, "unknown section id %d", section);
    }
    HIP_TRY(c, hipGetLastError());
    return FLUID_OK;
}

// `impl` (a grouped pass of fluid_run_step) runs in place of `section` and is timed under its id
int timed_section(fluid_ctx* c, int section, int impl = -1) {
    if (section < 0 || section >= FLUID_SECTION_COUNT)
        return c->fail(FLUID_ERR_INVALID_ARG, "unknown section id %d", section);
    SectionTimer tm{c};
    int rc = tm.begin(section);
    if (rc) return rc;
    rc = run_section_impl(c, impl >= 0 ? impl : section);
    int rc2 = tm.end();
    return rc ? rc : rc2;
}

int check_image(fluid_ctx* c, int image_id) {
    if (image_id < 0 || image_id >= FLUID_IMAGE_COUNT)
        return c->fail(FLUID_ERR_INVALID_ARG, "unknown image id %d", image_id);
    if (image_id >= 8 && !c->surface)
        return c->fail(FLUID_ERR_UNSUPPORTED,
                       "image %d belongs to the surface-prep passes (sections 15-18): create the context "
                       "with fluid_create_info.surface_prep",
                       image_id);
    return FLUID_OK;
}
// bytes of the part of an image that upload / download / clear address
uint64_t image_host_bytes(const fluid_ctx* c, int image_id) {
    return image_id >= 8 ? c->surf_cells * 4 : c->owned_cells() * c->img[image_id].elem_bytes;
}
uint8_t* image_host_base(const fluid_ctx* c, int image_id) {
    return image_id >= 8 ? c->surf<uint8_t>(image_id) : c->plane0<uint8_t>(image_id);
}

}  // namespace

// =================================================================================================
extern "C" {

int fluid_abi_version(void) { return FLUID_ENGINE_ABI_VERSION; }

const char* fluid_section_name(int section_id) {
    return (section_id >= 0 && section_id < FLUID_SECTION_COUNT) ? kSectionNames[section_id] : nullptr;
}

int fluid_params_default(fluid_params* p, uint32_t w, uint32_t h, uint32_t d, uint32_t capacity) {
    if (!p) return FLUID_ERR_INVALID_ARG;
    memset(p, 0, sizeof *p);
    // simulation_constants.h:7-139
    p->fluid_size[0] = w;
    p->fluid_size[1] = h;
    p->fluid_size[2] = d;
    p->fluid_volume = (uint32_t)((uint64_t)w * h * d);
    p->cell_type_inactive = FLUID_CELL_INACTIVE;
    p->cell_type_air = FLUID_CELL_AIR;
    p->cell_type_water = FLUID_CELL_WATER;
    p->cell_type_solid = FLUID_CELL_SOLID;
    p->time_delta = 0.01f;
    p->pressure_air = 1.0f;
    p->cell_width = 1.0f;
    p->fluid_density = 1.0f;
    p->particle_compute_size[0] = capacity;
    p->particle_compute_size[1] = 1;
    p->particle_spawn_cube_resolution[0] = 100;
    p->particle_spawn_cube_resolution[1] = 100;
    p->particle_spawn_cube_resolution[2] = 100;
    p->particle_spawn_cube_volume = 100u * 100u * 100u;
    p->particle_spawn_cube_offset[0] = 5.0f;
    p->particle_spawn_cube_offset[1] = 2.0f;
    p->particle_spawn_cube_offset[2] = 1.5f;
    p->particle_spawn_cube_size[0] = 10.0f;
    p->particle_spawn_cube_size[1] = 10.0f;
    p->particle_spawn_cube_size[2] = 2.0f;
    p->gravity = 10.0f;
    p->diffuse_k = 0.01f;
    p->detailed_resolution = 5;
    p->detailed_resolution_volume = (int32_t)((125ull * w * h * d) & 0x7FFFFFFFull);
    p->max_inertia = 100;
    p->inertia_increase_filled = 4;
    p->required_neighbour_hits = 1;
    p->inertia_increase_neighbour = 1;
    p->inertia_decrease = 1;
    p->dens_division_coefficient = 30.0f;
    p->dens_diffuse_k = 0.1f;
    p->particle_color[0] = 1.0f;
    p->particle_base_size = 10.0f;
    p->light_dir[0] = 1.0f;
    p->light_dir[1] = -3.0f;
    p->light_dir[2] = 1.0f;
    p->ambient_color[2] = 0.3f;
    p->diffuse_color[1] = 0.8f;
    p->diffuse_color[2] = 0.7f;
    p->fluid_surface_render_size[0] = 5 * w - 1;
    p->fluid_surface_render_size[1] = 5 * h - 1;
    p->fluid_surface_render_size[2] = 5 * d - 1;
    p->active_particle_w = 1.0f;
    p->fountain_position[0] = w / 2;
    p->fountain_position[1] = h - 2;
    p->fountain_position[2] = d / 2;
    p->fountain_force = -3000.0f;
    p->solid_repel_velocity = 0.01f;
    p->particle_max_size = 20.0f;
    return FLUID_OK;
}

uint64_t fluid_required_arena_bytes(const fluid_create_info* info) {
    if (!info || !info->params_blob) return 0;
    fluid_params p;
    memcpy(&p, info->params_blob, sizeof p);
    std::string err;
    if (validate_params(p, err)) return 0;
    Layout L;
    uint64_t cap;
    uint32_t z0, dl;
    if (compute_layout(info, p, L, cap, z0, dl)) return 0;
    return L.total;
}

int fluid_create(fluid_ctx** out, const fluid_create_info* info) {
    if (!out || !info || !info->params_blob || info->struct_bytes < FLUID_CREATE_INFO_V1_BYTES) {
        g_create_error = "fluid_create: null argument or struct_bytes too small";
        return FLUID_ERR_INVALID_ARG;
    }
    *out = nullptr;
    fluid_params p;
    memcpy(&p, info->params_blob, sizeof p);
    int rc = validate_params(p, g_create_error);
    if (rc) return rc;
    Layout L;
    uint64_t capacity;
    uint32_t z0, dl;
    rc = compute_layout(info, p, L, capacity, z0, dl);
    if (rc) {
        g_create_error = "slab_z_begin + slab_z_count exceeds fluid_size.z, or surface_prep on a Z-slab "
                         "context / with detailed_resolution < 1";
        return rc;
    }

    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) {
        g_create_error = std::string("no HIP device: ") + hipGetErrorString(e);
        return FLUID_ERR_NO_DEVICE;
    }
    int dev = info->device;
    if (dev < 0) {
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    }
    if (dev >= ndev) {
        g_create_error = "device ordinal out of range";
        return FLUID_ERR_INVALID_ARG;
    }
    if ((e = hipSetDevice(dev)) != hipSuccess) {
        g_create_error = std::string("hipSetDevice: ") + hipGetErrorString(e);
        return FLUID_ERR_HIP;
    }

    fluid_ctx* c = new fluid_ctx();
    c->device = dev;
    c->params = p;
    c->pk = make_params_k(p);
    c->g.W = (int)p.fluid_size[0];
    c->g.H = (int)p.fluid_size[1];
    c->g.Dg = (int)p.fluid_size[2];
    c->g.Dl = (int)dl;
    c->g.z0 = (int)z0;
    c->g.plane = (int64_t)p.fluid_size[0] * p.fluid_size[1];
    c->g.sg_lo = c->g.sg_hi = IMG_GHOST;
    c->g.zl = 1;
    c->is_slab = dl != p.fluid_size[2];
    c->particle_capacity = capacity;
    c->surface = L.surf_cells != 0;
    if (c->surface) {
        c->sk.res = p.detailed_resolution;
        c->sk.W = c->g.W * c->sk.res;
        c->sk.H = c->g.H * c->sk.res;
        c->sk.D = c->g.Dg * c->sk.res;
        c->sk.sW = c->g.W;
        c->sk.sH = c->g.H;
        c->sk.plane = (int64_t)c->sk.W * c->sk.H;
        c->surf_cells = L.surf_cells;
        for (int i = 0; i < 4; i++) c->surf_offset[i] = L.surf_offset[i];
        if (info_has(info, offsetof(fluid_create_info, surface_diffuse_steps) + sizeof(uint32_t)) &&
            info->surface_diffuse_steps != 0)
            c->surface_steps = info->surface_diffuse_steps;
    }
    c->pressure_iterations = info->pressure_iterations ? info->pressure_iterations : 200;
    for (int i = 0; i < 8; i++) {
        c->img[i].elem_bytes = kElemBytes[i];
        c->img[i].offset = L.img_offset[i];
        c->img[i].bytes = L.img_bytes[i];
    }
    c->particles_offset = L.particles_offset;
    c->mask_offset = L.mask_offset;
    c->rhs_offset = L.rhs_offset;
    c->active_offset = L.active_offset;
    c->active_bytes = L.active_bytes;
    c->quiet_offset = L.quiet_offset;
    c->early_offset = L.early_offset;
    c->pbricks_offset = L.pbricks_offset;
    for (int i = 0; i < 3; i++) c->work_offset[i] = L.work_offset[i];
    c->flags_offset = L.flags_offset;
    c->ghost_bricks_offset = L.ghost_bricks_offset;
    c->leavers_offset = L.leavers_offset;
    c->leavers_capacity = L.leavers_capacity;
    c->arena_bytes = L.total;

    auto bail = [&](int code, const std::string& msg) {
        g_create_error = msg;
        fluid_destroy(c);
        return code;
    };
    if (info->hip_stream) {
        c->stream = reinterpret_cast<hipStream_t>(info->hip_stream);
    } else {
        if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess)
            return bail(FLUID_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e));
        c->own_stream = true;
    }
    if (info->arena) {
        if (info->arena_bytes < L.total) return bail(FLUID_ERR_SIZE_MISMATCH, "arena too small");
        if (reinterpret_cast<uint64_t>(info->arena) % 256 != 0)
            return bail(FLUID_ERR_INVALID_ARG, "arena must be 256-byte aligned");
        c->arena = static_cast<uint8_t*>(info->arena);
    } else {
        void* ptr = nullptr;
        if ((e = hipMalloc(&ptr, L.total)) != hipSuccess)
            return bail(FLUID_ERR_OUT_OF_MEMORY, std::string("hipMalloc of ") +
                                                     std::to_string(L.total) +
                                                     " bytes: " + hipGetErrorString(e));
        c->arena = static_cast<uint8_t*>(ptr);
        c->own_arena = true;
    }
    {
        void* hp = nullptr;
        if (hipHostMalloc(&hp, 64, hipHostMallocDefault) == hipSuccess) {
            c->brick_count_host = static_cast<uint32_t*>(hp);
            memset(hp, 0, 64);
        }
    }
    // zero everything once: the ghost planes at the domain faces must read as 0 forever
    if ((e = hipMemsetAsync(c->arena, 0, L.total, c->stream)) != hipSuccess ||
        (e = hipStreamSynchronize(c->stream)) != hipSuccess)
        return bail(FLUID_ERR_HIP, std::string("arena clear: ") + hipGetErrorString(e));
    *out = c;
    return FLUID_OK;
}

void fluid_destroy(fluid_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto& s : c->pending) c->free_slots.push_back(s);
    for (auto& s : c->free_slots) {
        if (s.start) (void)hipEventDestroy(s.start);
        if (s.stop) (void)hipEventDestroy(s.stop);
    }
    if (c->brick_count_host) (void)hipHostFree(c->brick_count_host);
    if (c->ev_pass_start) (void)hipEventDestroy(c->ev_pass_start);
    if (c->ev_edges_done) (void)hipEventDestroy(c->ev_edges_done);
    if (c->edge_stream) (void)hipStreamDestroy(c->edge_stream);
    if (c->wide) (void)hipFree(c->wide);
    psort_release(c);
    local_release(c);
    if (c->blur_tmp) (void)hipFree(c->blur_tmp);
    if (c->mc_tables) (void)hipFree(c->mc_tables);
    if (c->own_arena && c->arena) (void)hipFree(c->arena);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* fluid_last_error(const fluid_ctx* c) {
    return c ? c->error.c_str() : g_create_error.c_str();
}

int fluid_image_bytes(const fluid_ctx* c, int image_id, uint64_t* bytes) {
    if (!c || !bytes) return FLUID_ERR_INVALID_ARG;
    if (image_id < 0 || image_id >= FLUID_IMAGE_COUNT) return FLUID_ERR_INVALID_ARG;
    if (image_id >= 8 && !c->surface) return FLUID_ERR_UNSUPPORTED;
    *bytes = image_host_bytes(c, image_id);
    return FLUID_OK;
}

int fluid_buffer_bytes(const fluid_ctx* c, int buffer_id, uint64_t* bytes) {
    if (!c || !bytes) return FLUID_ERR_INVALID_ARG;
    switch (buffer_id) {
        case FLUID_BUF_PARTICLES_BUF:
            *bytes = c->particle_capacity * 16;
            return FLUID_OK;
        case FLUID_BUF_SIMULATION_PARAMS_BUF:
            *bytes = FLUID_PARAMS_BYTES;
            return FLUID_OK;
        case FLUID_BUF_MARCHING_CUBES_COUNTS_BUF:  // marching_cubes.h:24-27: 4 bytes x 256 configurations
            if (!c->surface) return FLUID_ERR_UNSUPPORTED;
            *bytes = 4 * 256;
            return FLUID_OK;
        case FLUID_BUF_MARCHING_CUBES_EDGES_BUF:   // 4 bytes x 15 indices x 256 configurations
            if (!c->surface) return FLUID_ERR_UNSUPPORTED;
            *bytes = 4 * 15 * 256;
            return FLUID_OK;
        default:
            return FLUID_ERR_INVALID_ARG;
    }
}

int fluid_upload_image(fluid_ctx* c, int image_id, const void* host, uint64_t bytes) {
    if (c) c->quiet_valid = false;  // a write from outside fluid_run_step (quiet_bricks.h)
    if (!c) return FLUID_ERR_INVALID_ARG;
    int rc = check_image(c, image_id);
    if (rc) return rc;
    if (!host) return c->fail(FLUID_ERR_INVALID_ARG, "null host pointer");
    const uint64_t want = image_host_bytes(c, image_id);
    if (bytes != want)
        return c->fail(FLUID_ERR_SIZE_MISMATCH, "image %d holds %llu bytes, caller passed %llu",
                       image_id, (unsigned long long)want, (unsigned long long)bytes);
    HIP_TRY(c, hipSetDevice(c->device));
    c->touched(image_id);
    HIP_TRY(c, hipMemcpyAsync(image_host_base(c, image_id), host, bytes, hipMemcpyHostToDevice,
                              c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return FLUID_OK;
}

int fluid_download_image(fluid_ctx* c, int image_id, void* host, uint64_t bytes) {
    if (!c) return FLUID_ERR_INVALID_ARG;
    int rc = check_image(c, image_id);
    if (rc) return rc;
    if (!host) return c->fail(FLUID_ERR_INVALID_ARG, "null host pointer");
    const uint64_t want = image_host_bytes(c, image_id);
    if (bytes != want)
        return c->fail(FLUID_ERR_SIZE_MISMATCH, "image %d holds %llu bytes, caller passed %llu",
                       image_id, (unsigned long long)want, (unsigned long long)bytes);
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(host, image_host_base(c, image_id), bytes, hipMemcpyDeviceToHost,
                              c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return FLUID_OK;
}

int fluid_set_params(fluid_ctx* c, const void* blob) {
    if (!c) return FLUID_ERR_INVALID_ARG;
    if (!blob) return c->fail(FLUID_ERR_INVALID_ARG, "null params blob");
    fluid_params p;
    memcpy(&p, blob, sizeof p);
    std::string err;
    int rc = validate_params(p, err);
    if (rc) return c->fail(rc, "%s", err.c_str());
    for (int i = 0; i < 3; i++)
        if (p.fluid_size[i] != c->params.fluid_size[i])
            return c->fail(FLUID_ERR_SIZE_MISMATCH, "fluid_size cannot change on a live context");
    if (c->surface && p.detailed_resolution != c->sk.res)
        return c->fail(FLUID_ERR_SIZE_MISMATCH,
                       "detailed_resolution cannot change on a surface_prep context (the detailed "
                       "images were allocated for %d)", c->sk.res);
    c->params = p;
    c->pk = make_params_k(p);
    c->params_changed();  // cell type values, rho, dx, dt, p_air may have changed
    return FLUID_OK;
}

int fluid_upload_buffer(fluid_ctx* c, int buffer_id, const void* host, uint64_t bytes) {
    if (!c) return FLUID_ERR_INVALID_ARG;
    if (!host) return c->fail(FLUID_ERR_INVALID_ARG, "null host pointer");
    uint64_t want = 0;
    int rc = fluid_buffer_bytes(c, buffer_id, &want);
    if (rc) return c->fail(rc, "buffer %d is not part of this path", buffer_id);
    if (bytes != want)
        return c->fail(FLUID_ERR_SIZE_MISMATCH, "buffer %d holds %llu bytes, caller passed %llu",
                       buffer_id, (unsigned long long)want, (unsigned long long)bytes);
    if (buffer_id == FLUID_BUF_SIMULATION_PARAMS_BUF) return fluid_set_params(c, host);
    HIP_TRY(c, hipSetDevice(c->device));
    if (buffer_id == FLUID_BUF_MARCHING_CUBES_COUNTS_BUF || buffer_id == FLUID_BUF_MARCHING_CUBES_EDGES_BUF) {
        // MarchingCubesBuffers::loadData (marching_cubes.h:30-33): the two tables, validated on the way in
        const uint32_t* w = static_cast<const uint32_t*>(host);
        const bool is_counts = buffer_id == FLUID_BUF_MARCHING_CUBES_COUNTS_BUF;
        for (uint64_t i = 0; i < bytes / 4; i++)
            if (is_counts ? w[i] > 5u : (w[i] > 11u && w[i] != 255u))
                return c->fail(FLUID_ERR_INVALID_ARG, "marching-cubes table entry %llu = %u is out of range",
                               (unsigned long long)i, w[i]);
        if (!c->mc_tables) {
            void* ptr = nullptr;
            HIP_TRY(c, hipMalloc(&ptr, 4 * 256 * 16));
            c->mc_tables = static_cast<uint32_t*>(ptr);
        }
        HIP_TRY(c, hipMemcpyAsync(c->mc_tables + (is_counts ? 0 : 256), host, bytes, hipMemcpyHostToDevice,
                                  c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->mc_loaded[is_counts ? 0 : 1] = true;
        return FLUID_OK;
    }
    if (bytes) {
        psort_reset(c);  // the caller's array is in slot order
        if (c->is_slab) {
            // the caller passes the global array: this slab keeps what it owns, compactly (host-side selection —
            // an upload is no hot path, and the device never holds a slot per particle of the run)
            const float4* src = static_cast<const float4*>(host);
            std::vector<float4> keep;
            std::vector<uint32_t> ids;
            for (uint64_t i = 0; i < c->particle_capacity; i++) {
                const int pl = particle_owner_plane_host(src[i].z, c->g.Dg) - c->g.z0;
                if ((unsigned)pl < (unsigned)c->g.Dl) {
                    keep.push_back(src[i]);
                    ids.push_back((uint32_t)i);
                }
            }
            auto& L = c->loc;
            L.on = true;
            L.n = L.holes = L.n_sorted = 0;
            L.cur = 0;
            int rc2 = local_reserve(c, keep.size());
            if (rc2) return rc2;
            if (!keep.empty()) {
                HIP_TRY(c, hipMemcpyAsync(L.buf[0], keep.data(), keep.size() * 16, hipMemcpyHostToDevice, c->stream));
                HIP_TRY(c, hipMemcpyAsync(L.pid[0], ids.data(), ids.size() * 4, hipMemcpyHostToDevice, c->stream));
            }
            const uint32_t counters[2] = {(uint32_t)keep.size(), 0u};
            HIP_TRY(c, hipMemcpyAsync(L.counters, counters, 8, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));  // the vectors go out of scope
            L.n = counters[0];
            return FLUID_OK;
        }
        HIP_TRY(c, hipMemcpyAsync(c->particles(), host, bytes, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return FLUID_OK;
}

int fluid_download_buffer(fluid_ctx* c, int buffer_id, void* host, uint64_t bytes) {
    if (!c) return FLUID_ERR_INVALID_ARG;
    if (!host) return c->fail(FLUID_ERR_INVALID_ARG, "null host pointer");
    uint64_t want = 0;
    int rc = fluid_buffer_bytes(c, buffer_id, &want);
    if (rc) return c->fail(rc, "buffer %d is not part of this path", buffer_id);
    if (bytes != want)
        return c->fail(FLUID_ERR_SIZE_MISMATCH, "buffer %d holds %llu bytes, caller passed %llu",
                       buffer_id, (unsigned long long)want, (unsigned long long)bytes);
    if (buffer_id == FLUID_BUF_SIMULATION_PARAMS_BUF) {
        memcpy(host, &c->params, FLUID_PARAMS_BYTES);
        return FLUID_OK;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    if (buffer_id == FLUID_BUF_MARCHING_CUBES_COUNTS_BUF || buffer_id == FLUID_BUF_MARCHING_CUBES_EDGES_BUF) {
        const bool is_counts = buffer_id == FLUID_BUF_MARCHING_CUBES_COUNTS_BUF;
        if (!c->mc_loaded[is_counts ? 0 : 1]) return c->fail(FLUID_ERR_INVALID_ARG, "table not uploaded yet");
        HIP_TRY(c, hipMemcpyAsync(host, c->mc_tables + (is_counts ? 0 : 256), bytes, hipMemcpyDeviceToHost,
                                  c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        return FLUID_OK;
    }
    if (bytes && c->is_slab) {
        // the global array: the slots this slab holds with their data, a tombstone in every other one
        float4* out = static_cast<float4*>(host);
        float4 tomb;
        tomb.x = tomb.y = tomb.z = 0.f;
        const uint32_t tb = PARTICLE_TOMBSTONE_BITS;
        memcpy(&tomb.w, &tb, 4);
        for (uint64_t i = 0; i < c->particle_capacity; i++) out[i] = tomb;
        const uint32_t n = c->loc.on ? c->loc.n : 0u;
        if (n) {
            std::vector<float4> data(n);
            std::vector<uint32_t> ids(n);
            HIP_TRY(c, hipMemcpyAsync(data.data(), c->loc.buf[c->loc.cur], (uint64_t)n * 16, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipMemcpyAsync(ids.data(), c->loc.pid[c->loc.cur], (uint64_t)n * 4, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            for (uint32_t k = 0; k < n; k++) {
                uint32_t w;
                memcpy(&w, &data[k].w, 4);
                if (w != PARTICLE_TOMBSTONE_BITS && ids[k] < c->particle_capacity) out[ids[k]] = data[k];
            }
        }
        return FLUID_OK;
    }
    if (bytes) {
        const float4* src = c->particles();
        if (c->ps.valid) {  // stored sorted by bin: slot order into the buffer that is free between two sorts
            float4* tmp = c->ps.cur ? c->particles_home() : c->ps.alt;
            hipLaunchKernelGGL(k_pbin_to_slot_order, dim3((unsigned)((c->particle_capacity + 255) / 256)),
                               dim3(256), 0, c->stream, src, c->ps.slot_of[c->ps.cur], c->particle_capacity,
                               tmp);
            HIP_TRY(c, hipGetLastError());
            src = tmp;
        }
        HIP_TRY(c, hipMemcpyAsync(host, src, bytes, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return FLUID_OK;
}

int fluid_set_pressure_iterations(fluid_ctx* c, uint32_t iterations) {
    if (!c) return FLUID_ERR_INVALID_ARG;
    c->pressure_iterations = iterations;
    return FLUID_OK;
}

int fluid_set_diffuse_mode(fluid_ctx* c, int mode) {
    if (c) c->quiet_valid = false;  // a write from outside fluid_run_step (quiet_bricks.h)
    if (!c) return FLUID_ERR_INVALID_ARG;
    if (mode != FLUID_DIFFUSE_REFERENCE_EXACT && mode != FLUID_DIFFUSE_INTENDED)
        return c->fail(FLUID_ERR_INVALID_ARG, "unknown diffuse mode %d", mode);
    c->diffuse_mode = mode;
    return FLUID_OK;
}

int fluid_set_pressure_solver(fluid_ctx* c, int solver, float omega) {
    if (!c) return FLUID_ERR_INVALID_ARG;
    if (solver != FLUID_SOLVER_JACOBI && solver != FLUID_SOLVER_RED_BLACK_SOR)
        return c->fail(FLUID_ERR_INVALID_ARG, "unknown solver %d", solver);
    if (solver == FLUID_SOLVER_RED_BLACK_SOR && !(omega > 0.0f && omega < 2.0f))
        return c->fail(FLUID_ERR_INVALID_ARG, "SOR needs 0 < omega < 2, got %g", (double)omega);
    if (solver == FLUID_SOLVER_RED_BLACK_SOR && c->is_slab)
        return slab_unsupported(c, "the red-black SOR solver");
    c->solver = solver;
    c->sor_omega = omega;
    return FLUID_OK;
}

int fluid_set_option(fluid_ctx* c, int option, int64_t value) {
    if (c) c->quiet_valid = false;  // a write from outside fluid_run_step (quiet_bricks.h)
    if (!c) return FLUID_ERR_INVALID_ARG;
    if (option < 0 || option >= FLUID_OPT_COUNT)
        return c->fail(FLUID_ERR_INVALID_ARG, "unknown option %d", option);
    c->opt[option] = value;
    return FLUID_OK;
}

static int run_step_slice(fluid_ctx* c, int first, int count, bool grouped, bool whole_step);

int fluid_run_section(fluid_ctx* c, int section_id) {
    if (!c) return FLUID_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->driver_step && section_id >= FLUID_SEC_01A_CLEAR_PARTICLE_DENSITIES &&
        section_id <= FLUID_SEC_14_PARTICLES && section_id != FLUID_SEC_12_SOLVE_PRESSURE)
        return run_step_slice(c, section_id, 1, true, false);  // a step driven from outside: with its skipping
    c->quiet_valid = false;  // a write from outside a step (quiet_bricks.h)
    return timed_section(c, section_id);
}

int fluid_clear_image(fluid_ctx* c, int image_id, const uint32_t value_bits[4]) {
    if (c) c->quiet_valid = false;  // a write from outside fluid_run_step (quiet_bricks.h)
    if (!c) return FLUID_ERR_INVALID_ARG;
    int rc = check_image(c, image_id);
    if (rc) return rc;
    if (!value_bits) return c->fail(FLUID_ERR_INVALID_ARG, "null clear value");
    HIP_TRY(c, hipSetDevice(c->device));
    c->touched(image_id);
    if (image_id == FLUID_IMG_PRESSURES_1 || image_id == FLUID_IMG_PRESSURES_2)
        c->pressure_dispatch_index = 0;
    return fill_image4(c, image_id, value_bits);
}

int fluid_run_pressure_dispatch(fluid_ctx* c, uint32_t is_even_iteration) {
    if (!c) return FLUID_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    SectionTimer tm{c};
    int rc = tm.begin(FLUID_SEC_12_SOLVE_PRESSURE);
    if (rc) return rc;
    rc = launch_pressure(c, is_even_iteration);
    int rc2 = tm.end();
    return rc ? rc : rc2;
}

int fluid_run_surface_diffuse_dispatch(fluid_ctx* c, uint32_t is_even_iteration) {
    if (!c) return FLUID_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    // the section's own counter decides the direction: set it so that this dispatch has the asked parity
    c->surface_dispatch_index = is_even_iteration == 1u ? 0u : 1u;
    return timed_section(c, FLUID_SEC_18_DIFFUSE_FLOAT_DENSITIES);
}

// ---- the loop section in explicit form (multi-GPU: the caller exchanges halos between launches) ----
int fluid_pressure_loop_begin(fluid_ctx* c) {
    if (!c) return FLUID_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    if (!fast_loop_possible(c))
        return c->fail(FLUID_ERR_UNSUPPORTED,
                       "the working-buffer loop needs fluid_size.x %% 4 == 0 (and pressure kernel "
                       "option 0 or >= 5); use fluid_run_pressure_dispatch per sweep instead");
    return loop_begin(c);
}

int fluid_pressure_loop_max_sweeps(fluid_ctx* c) {
    if (!c) return FLUID_ERR_INVALID_ARG;
    return max_sweeps_per_pass(c);
}

int fluid_pressure_loop_available(fluid_ctx* c) {
    if (!c) return FLUID_ERR_INVALID_ARG;
    return fast_loop_possible(c) ? 1 : 0;
}

int fluid_pressure_loop_advance(fluid_ctx* c, uint32_t sweeps, int keep_intermediate,
                                int* written_buffer) {
    if (!c) return FLUID_ERR_INVALID_ARG;
    if (!c->loop_open) return c->fail(FLUID_ERR_INVALID_ARG, "fluid_pressure_loop_begin first");
    if (sweeps == 0 || (int)sweeps > max_sweeps_per_pass(c))
        return c->fail(FLUID_ERR_INVALID_ARG, "cannot advance by %u sweeps in one launch", sweeps);
    HIP_TRY(c, hipSetDevice(c->device));
    SectionTimer tm{c};
    int rc = tm.begin(FLUID_SEC_12_SOLVE_PRESSURE);
    if (rc) return rc;
    rc = loop_advance(c, sweeps, keep_intermediate != 0, written_buffer);
    int rc2 = tm.end();
    if (c->timing && rc == FLUID_OK && rc2 == FLUID_OK && sweeps >= 2)
        c->sec_calls[FLUID_SEC_12_SOLVE_PRESSURE] += sweeps - 1;  // count sweeps
    return rc ? rc : rc2;
}

int fluid_pressure_loop_advance_part_n(fluid_ctx* c, uint32_t sweeps, int keep_intermediate, int part,
                                       int32_t interior_begin, int32_t interior_end, int* written_buffer) {
    if (!c) return FLUID_ERR_INVALID_ARG;
    if (!c->loop_open) return c->fail(FLUID_ERR_INVALID_ARG, "fluid_pressure_loop_begin first");
    if (!fuse_enabled(c))
        return c->fail(FLUID_ERR_UNSUPPORTED, "split passes need the several-sweeps-per-pass kernels");
    if (sweeps < 2 || (int)sweeps > max_sweeps_per_pass(c))
        return c->fail(FLUID_ERR_INVALID_ARG, "a split pass of %u sweeps", sweeps);
    if (part != FLUID_LOOP_PART_EDGES && part != FLUID_LOOP_PART_INTERIOR)
        return c->fail(FLUID_ERR_INVALID_ARG, "unknown part %d", part);
    if (interior_end < interior_begin)
        return c->fail(FLUID_ERR_INVALID_ARG, "interior planes [%d, %d)", interior_begin, interior_end);
    HIP_TRY(c, hipSetDevice(c->device));
    SectionTimer tm{c};
    int rc = tm.begin(FLUID_SEC_12_SOLVE_PRESSURE);
    if (rc) return rc;
    const bool first = c->loop_part_done == 0;
    rc = loop_advance(c, sweeps, keep_intermediate != 0, written_buffer,
                      part == FLUID_LOOP_PART_EDGES ? FUSED_EDGES : FUSED_INTERIOR, interior_begin,
                      interior_end);
    int rc2 = tm.end();  // two launches = two timer calls: the first two sweeps of the pass in sec_calls
    if (c->timing && rc == FLUID_OK && rc2 == FLUID_OK && !first && sweeps > 2)
        c->sec_calls[FLUID_SEC_12_SOLVE_PRESSURE] += sweeps - 2;
    return rc ? rc : rc2;
}

int fluid_pressure_loop_advance_part(fluid_ctx* c, int keep_intermediate, int part,
                                     int32_t interior_begin, int32_t interior_end,
                                     int* written_buffer) {
    return fluid_pressure_loop_advance_part_n(c, 2, keep_intermediate, part, interior_begin, interior_end,
                                              written_buffer);
}

int fluid_pressure_loop_edge_stream(fluid_ctx* c, void** hip_stream) {
    if (!c) return FLUID_ERR_INVALID_ARG;
    if (!hip_stream) return c->fail(FLUID_ERR_INVALID_ARG, "null output pointer");
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = ensure_edge_stream(c);
    if (rc) return rc;
    *hip_stream = c->edge_stream;
    return FLUID_OK;
}

int fluid_pressure_loop_halo_exchanged(fluid_ctx* c, uint32_t depth, uint32_t aux_depth) {
    if (!c) return FLUID_ERR_INVALID_ARG;
    if (!c->loop_open) return c->fail(FLUID_ERR_INVALID_ARG, "fluid_pressure_loop_begin first");
    if (depth > (uint32_t)LOOP_GHOST || aux_depth > (uint32_t)LOOP_GHOST)
        return c->fail(FLUID_ERR_INVALID_ARG, "at most %d ghost planes", LOOP_GHOST);
    HIP_TRY(c, hipSetDevice(c->device));
    c->loop_halo = (int)depth;
    if (aux_depth) c->loop_aux_halo = (int)aux_depth;
    if (!c->loop_ghost_bg && c->is_slab && depth > 0) {
        // The sweeps store water cells only; the constants of the non-water cells in the ghost
        // planes of the other two buffers are those just received in this buffer.
        const uint64_t bytes = (uint64_t)depth * c->g.plane * 4;
        const bool lo = c->g.z0 > 0, hi = c->g.z0 + c->g.Dl < c->g.Dg;
        for (int i = 0; i < 3; i++) {
            if (i == c->loop_cur) continue;
            if (lo)
                HIP_TRY(c, hipMemcpyAsync(c->work0(i) - (int64_t)depth * c->g.plane,
                                          c->work0(c->loop_cur) - (int64_t)depth * c->g.plane, bytes,
                                          hipMemcpyDeviceToDevice, c->stream));
            if (hi)
                HIP_TRY(c, hipMemcpyAsync(c->work0(i) + (int64_t)c->g.Dl * c->g.plane,
                                          c->work0(c->loop_cur) + (int64_t)c->g.Dl * c->g.plane,
                                          bytes, hipMemcpyDeviceToDevice, c->stream));
        }
        c->loop_ghost_bg = true;
    }
    return FLUID_OK;
}

int fluid_pressure_loop_end(fluid_ctx* c) {
    if (!c) return FLUID_ERR_INVALID_ARG;
    if (!c->loop_open) return c->fail(FLUID_ERR_INVALID_ARG, "fluid_pressure_loop_begin first");
    HIP_TRY(c, hipSetDevice(c->device));
    return loop_end(c);
}

// which: 0..2 = working pressure buffers, 3 = mask (1 byte per cell), 4 = b_i; planes
// -LOOP_GHOST .. Dl + LOOP_GHOST - 1
int fluid_pressure_loop_plane_ptr(fluid_ctx* c, int which, int32_t plane, void** device_ptr,
                                  uint64_t* bytes) {
    if (!c) return FLUID_ERR_INVALID_ARG;
    if (!device_ptr || !bytes) return c->fail(FLUID_ERR_INVALID_ARG, "null output pointer");
    if (which < 0 || which > 4) return c->fail(FLUID_ERR_INVALID_ARG, "loop buffer %d", which);
    if (plane < -LOOP_GHOST || plane >= c->g.Dl + LOOP_GHOST)
        return c->fail(FLUID_ERR_INVALID_ARG, "plane %d outside [%d, %d)", plane, -LOOP_GHOST,
                       c->g.Dl + LOOP_GHOST);
    const uint64_t elem = which == 3 ? 1 : 4;
    const uint64_t pb = (uint64_t)c->g.plane * elem;
    const uint64_t base = which <= 2 ? c->work_offset[which]
                                     : (which == 3 ? c->mask_offset : c->rhs_offset);
    *device_ptr = c->arena + base + (uint64_t)(plane + LOOP_GHOST) * pb;
    *bytes = pb;
    return FLUID_OK;
}

int fluid_run_section_loop(fluid_ctx* c, int section_id, uint32_t iterations) {
    if (!c) return FLUID_ERR_INVALID_ARG;
    if (section_id == FLUID_SEC_18_DIFFUSE_FLOAT_DENSITIES) {
        // FlowLoopPushConstantSection(float_density_diffuse_steps, ...), fluid_flow_sections.h:376-388
        HIP_TRY(c, hipSetDevice(c->device));
        c->surface_dispatch_index = 0;
        if (!c->surface)
            return c->fail(FLUID_ERR_UNSUPPORTED,
                           "section %d is a surface-prep pass: create the context with "
                           "fluid_create_info.surface_prep", section_id);
        // Two dispatches per pass over HBM (k18_pair, kernels_surface.h).  The pairs alternate between
        // FLOAT_1 and a third image, which the first use allocates (on failure: one dispatch at a time).
        uint32_t k = 0;
        if (iterations >= 2 && c->sk.W % 4 == 0 &&
            (c->opt[FLUID_OPT_SURFACE_KERNEL] == 0 || c->opt[FLUID_OPT_SURFACE_KERNEL] >= 100) &&
            !c->blur_tmp_failed) {
            if (!c->blur_tmp) {
                void* q = nullptr;
                if (hipMalloc(&q, c->surf_cells * 4) != hipSuccess) {
                    (void)hipGetLastError();
                    c->blur_tmp_failed = true;
                }
                c->blur_tmp = static_cast<float*>(q);
            }
        }
        const int64_t sopt = c->opt[FLUID_OPT_SURFACE_KERNEL];
        if (iterations >= 2 && c->sk.W % 4 == 0 && (sopt == 0 || sopt >= 100) && c->blur_tmp) {
            const SurfK& s = c->sk;
            float* f1 = c->surf<float>(FLUID_IMG_PARTICLE_DENSITIES_FLOAT_1);
            float* f2 = c->surf<float>(FLUID_IMG_PARTICLE_DENSITIES_FLOAT_2);
            const uint8_t* T = c->plane0<uint8_t>(FLUID_IMG_CELL_TYPES);
            const int R = sopt >= 100 ? (int)(sopt - 100) : K18_PAIR_ROWS;
            const int zchunk = 32;
            const dim3 pgrid((s.W + 247) / 248, (s.H + R - 3) / (R - 2), (s.D + zchunk - 1) / zchunk);
            float* cur = f1;  // the complete image of the newest even iterate
            SectionTimer tm{c};
            int rc = tm.begin(section_id);
            if (rc) return rc;
            for (; k + 2 <= iterations; k += 2) {
                float* out = cur == f1 ? c->blur_tmp : f1;
                const bool last = k + 2 == iterations;  // FLOAT_2 ends with the last odd iterate
#define FLUID_PAIR(RR)                                                                                         \
    if (last)                                                                                                  \
        hipLaunchKernelGGL((k18_pair<RR, true>), pgrid, dim3(64, RR, 1), 0, c->stream, T, cur, f2, out, s,     \
                           c->params.dens_diffuse_k, c->pk.t_solid, zchunk);                                   \
    else                                                                                                       \
        hipLaunchKernelGGL((k18_pair<RR, false>), pgrid, dim3(64, RR, 1), 0, c->stream, T, cur, f2, out, s,    \
                           c->params.dens_diffuse_k, c->pk.t_solid, zchunk);
                if (R == 8) { FLUID_PAIR(8) }
                else if (R == 10) { FLUID_PAIR(10) }
                else if (R == 12) { FLUID_PAIR(12) }
                else if (R == 14) { FLUID_PAIR(14) }
                else { FLUID_PAIR(16) }
#undef FLUID_PAIR
                cur = out;
            }
            rc = hipGetLastError() == hipSuccess ? FLUID_OK : c->fail(FLUID_ERR_HIP, "k18_pair launch failed");
            if (rc == FLUID_OK && cur != f1)  // an odd number of pairs: the newest even iterate belongs in FLOAT_1
                rc = hipMemcpyAsync(f1, cur, c->surf_cells * 4, hipMemcpyDeviceToDevice, c->stream) == hipSuccess
                         ? FLUID_OK
                         : c->fail(FLUID_ERR_HIP, "copy of the blurred image failed");
            int rc2 = tm.end();
            if (rc || rc2) return rc ? rc : rc2;
            if (c->timing) c->sec_calls[section_id] += k - 1;  // count dispatches
            c->surface_dispatch_index = k;
        }
        for (; k < iterations; k++) {
            int rc = timed_section(c, section_id);
            if (rc) return rc;
        }
        return FLUID_OK;
    }
    if (section_id != FLUID_SEC_12_SOLVE_PRESSURE)
        return c->fail(FLUID_ERR_INVALID_ARG,
                       "section %d is not a loop section (12_solve_pressure and "
                       "18_diffuse_float_densities are)", section_id);
    HIP_TRY(c, hipSetDevice(c->device));
    // FlowLoopPushConstantSection (fluid_flow_sections.h:300-313): the loop owns its own counter —
    // dispatch k of this call has is_even_iteration = (k % 2 == 0).
    if (c->solver == FLUID_SOLVER_RED_BLACK_SOR) {
        // opt-in: `iterations` red-black SOR iterations on PRESSURES_1, then PRESSURES_2 := PRESSURES_1
        // (13_fix_divergence reads PRESSURES_2)
        if (c->is_slab) return slab_unsupported(c, "the red-black SOR solver");
        c->pressure_dispatch_index = 0;
        if (fast_loop_possible(c) && fuse_enabled(c) && iterations > 0) {
            // On the working buffers, both colours of an iteration in ONE pass over HBM: the z march forms
            // the even cells' new values a plane ahead of the odd cells' (kernels_pressure_fused.h, SOR) —
            // the traffic of two Jacobi sweeps per iteration instead of two whole read-modify-write passes.
            SectionTimer tm{c};
            int rc = tm.begin(FLUID_SEC_12_SOLVE_PRESSURE);
            if (rc) return rc;
            rc = loop_begin(c);
            if (rc == FLUID_OK && iterations >= 16 && c->opt[FLUID_OPT_LAUNCH_BOX] == 0) rc = refresh_box(c);
            for (uint32_t k = 0; k < iterations && rc == FLUID_OK; k++) {
                const int cur = c->loop_cur, dst = other_buffer(cur, cur);
                rc = ensure_background(c, dst);
                if (rc) break;
                hipError_t e = k12_launch_canon2_sor(c->stream, c->mask0(), c->rhs0(), c->work0(cur),
                                                     c->work0(dst), c->bricks(), c->g, oob_value(c), c->box,
                                                     c->sor_omega);
                if (e == hipSuccess) e = hipGetLastError();
                if (e != hipSuccess) rc = c->fail(FLUID_ERR_HIP, "SOR launch: %s", hipGetErrorString(e));
                c->loop_cur = dst;
            }
            if (rc == FLUID_OK) rc = export_pressures(c, c->loop_cur, -1);  // -> the water cells of PRESSURES_1
            c->loop_open = false;
            c->pressure_dispatch_index = iterations;
            int rc2 = tm.end();
            if (c->timing && rc == FLUID_OK && rc2 == FLUID_OK && iterations > 1)
                c->sec_calls[FLUID_SEC_12_SOLVE_PRESSURE] += iterations - 1;
            if (rc || rc2) return rc ? rc : rc2;
        } else {
            for (uint32_t k = 0; k < iterations; k++) {
                int rc = timed_section(c, FLUID_SEC_12_SOLVE_PRESSURE);
                if (rc) return rc;
            }
        }
        HIP_TRY(c, hipMemcpyAsync(c->plane0<float>(FLUID_IMG_PRESSURES_2),
                                  c->plane0<float>(FLUID_IMG_PRESSURES_1), c->owned_cells() * 4,
                                  hipMemcpyDeviceToDevice, c->stream));
        return FLUID_OK;
    }
    SectionTimer tm{c};
    int rc = tm.begin(FLUID_SEC_12_SOLVE_PRESSURE);
    if (rc) return rc;
    if (fast_loop_possible(c)) {
        rc = run_fast_loop(c, iterations);
    } else {
        for (uint32_t k = 0; k < iterations && rc == FLUID_OK; k++)
            rc = launch_pressure(c, (k % 2u) == 0u ? 1u : 0u);
    }
    c->pressure_dispatch_index = iterations;
    int rc2 = tm.end();
    if (c->timing && rc == FLUID_OK && rc2 == FLUID_OK && iterations > 1)
        c->sec_calls[FLUID_SEC_12_SOLVE_PRESSURE] += iterations - 1;  // count dispatches
    return rc ? rc : rc2;
}

int fluid_run_init(fluid_ctx* c) {
    if (c) c->quiet_valid = false;  // a write from outside fluid_run_step (quiet_bricks.h)
    if (!c) return FLUID_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    static const int order[] = {FLUID_SEC_INIT_CLEAR_VELOCITIES_1, FLUID_SEC_INIT_CLEAR_CELL_TYPES,
                                FLUID_SEC_00_INIT_PARTICLES};
    for (int s : order) {
        int rc = timed_section(c, s);
        if (rc) return rc;
        if (s == FLUID_SEC_INIT_CLEAR_CELL_TYPES && c->surface) {  // list order, :140-143
            rc = timed_section(c, FLUID_SEC_INIT_CLEAR_DETAILED_DENSITIES_INERTIA);
            if (rc) return rc;
        }
    }
    return FLUID_OK;
}

// fluid_flow_sections.h:339-388: the tail of SimulationStepSections on the detailed grid
static int run_surface_prep(fluid_ctx* c) {
    static const int order[] = {FLUID_SEC_14A_CLEAR_DETAILED_DENSITIES,
                                FLUID_SEC_15_UPDATE_DETAILED_DENSITIES,
                                FLUID_SEC_16_COMPUTE_DETAILED_DENSITIES_INERTIA,
                                FLUID_SEC_17_COMPUTE_FLOAT_DENSITIES};
    // 17 is a pointwise function of the inertia 16 stores: one pass writes both images (four-cells-per-
    // thread kernel, detailed width % 4 == 0) unless the section list is asked for
    const bool fuse17 = c->opt[FLUID_OPT_STEP_FUSION] == 0 && c->sk.W % 4 == 0;
    for (int s : order) {
        if (fuse17 && s == FLUID_SEC_17_COMPUTE_FLOAT_DENSITIES) continue;
        c->surface_fuse17 = fuse17 && s == FLUID_SEC_16_COMPUTE_DETAILED_DENSITIES_INERTIA;
        int rc = timed_section(c, s);
        c->surface_fuse17 = false;
        if (rc) return rc;
    }
    return fluid_run_section_loop(c, FLUID_SEC_18_DIFFUSE_FLOAT_DENSITIES, c->surface_steps);
}

// Entries [first, first + count) of SimulationStepSections (fluid_flow_sections.h:163-338; the ids are
// consecutive in list order).  `grouped`: sections 04+05, 07+08 and 09+10+11 run as the grouped passes
// of kernels_step_fused.h, each timed under the id of the section it stands in for.
static int run_step_slice(fluid_ctx* c, int first, int count, bool grouped, bool whole_step) {
    const bool group = grouped && c->g.W % 4 == 0;
    const int end = first + count;
    // quiet bricks (quiet_bricks.h): only inside a whole grouped step on a whole-grid context whose
    // pressure loop runs on the working buffers (that is where the activity bricks come from)
    const bool quiet = c->driver_step
                           ? c->ds_quiet
                           : (whole_step && group && !c->is_slab && fast_loop_possible(c) &&
                              c->diffuse_mode == FLUID_DIFFUSE_REFERENCE_EXACT &&
                              c->opt[FLUID_OPT_QUIET_BRICKS] != 1);
    // the one-step test for the sections before 06 needs the previous step's water map in bricks():
    // nothing may have written the images since that step (quiet_valid still set)
    const bool early = c->driver_step ? c->ds_early
                                      : (quiet && c->quiet_valid && c->mask_valid && c->pbricks_valid &&
                                         first == FLUID_SEC_01A_CLEAR_PARTICLE_DENSITIES);
    if (!c->driver_step) c->quiet_in_use = c->early_in_use = c->early_step = false;  // else they carry over between calls
    for (int s = first; s < end;) {
        int rc, used = 1;
        if (early && s == FLUID_SEC_01A_CLEAR_PARTICLE_DENSITIES) {
            rc = timed_section(c, s, STEP_01A_CLEAR_WHERE_PARTICLES_WERE);
            if (rc) return rc;
            s += 1;
            continue;
        }
        if (quiet && s == FLUID_SEC_01_UPDATE_DENSITIES) {
            // the pass marks the bricks it counts particles in: the next step's 01a clears only those, and —
            // when this step may skip already — the one-step test of 02 ... 05 is derived from the marks
            c->early_wanted = early;
            rc = timed_section(c, s, STEP_01_UPDATE_DENSITIES_MARK_BRICKS);
            c->early_wanted = false;
            if (rc) return rc;
            c->early_in_use = c->early_step = early;
            s += 1;
            continue;
        }
        if (s == FLUID_SEC_06_UPDATE_CELL_TYPES) c->early_in_use = false;
        if (quiet && s == FLUID_SEC_07_ADVECT) {
            // CELL_TYPES of this step is final (06): activity bricks now, then the streaks
            rc = ensure_prepared(c, true);
            if (rc) return rc;
            if (!c->quiet_valid)
                HIP_TRY(c, hipMemsetAsync(c->quiet(), 0, c->active_bytes, c->stream));
            BrickK bk;
            k12_brick_dims(c->g.W, c->g.H, c->g.Dl, bk.nbx, bk.nby, bk.nbz);
            const int nb = (int)c->active_bytes;
            HIP_TRY(c, hipMemsetAsync(c->flags() + 9, 0, 4, c->stream));
            hipLaunchKernelGGL(k_update_quiet, dim3((nb + 255) / 256), dim3(256), 0, c->stream,
                               c->bricks(), c->quiet(), bk, c->flags() + 9, c->brick_edges(true));
            HIP_TRY(c, hipGetLastError());
            c->quiet_valid = true;
            c->quiet_in_use = true;
        }
        if (group && s == FLUID_SEC_04_COMPUTE_EXTRAPOLATED_VELOCITIES && end - s >= 2 &&
            c->v1_w_zero) {  // the pair of type scans needs w == 0 in all of VELOCITIES_1
            rc = timed_section(c, s, STEP_0405_EXTRAPOLATE);
            if (rc == FLUID_OK)
                rc = timed_section(c, FLUID_SEC_05_SET_EXTRAPOLATED_VELOCITIES, STEP_0405_APPLY);
            used = 2;
        } else if (group && s == FLUID_SEC_07_ADVECT && end - s >= 2) {
            rc = timed_section(c, s, STEP_0708_ADVECT_FORCES);
            used = 2;
        } else if (group && s == FLUID_SEC_09_DIFFUSE && end - s >= 3 &&
                   c->diffuse_mode == FLUID_DIFFUSE_REFERENCE_EXACT) {
            rc = timed_section(c, s, STEP_091011_SOLIDS_DIVERGENCE);
            used = 3;
        } else if (s == FLUID_SEC_12_SOLVE_PRESSURE) {
            rc = fluid_run_section_loop(c, s, c->pressure_iterations);
        } else {
            rc = timed_section(c, s);
        }
        if (rc) {
            c->quiet_in_use = c->early_in_use = c->early_step = false;
            return rc;
        }
        if (s == FLUID_SEC_13_FIX_DIVERGENCE) c->quiet_in_use = false;
        s += used;
    }
    if (!c->driver_step) c->quiet_in_use = c->early_in_use = c->early_step = false;
    return FLUID_OK;
}

int fluid_run_step(fluid_ctx* c) {
    if (!c) return FLUID_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->is_slab) return slab_unsupported(c, "fluid_run_step");
    int rc = run_step_slice(c, FLUID_SEC_01A_CLEAR_PARTICLE_DENSITIES,
                            FLUID_SEC_14_PARTICLES - FLUID_SEC_01A_CLEAR_PARTICLE_DENSITIES + 1,
                            c->opt[FLUID_OPT_STEP_FUSION] == 0, true);
    if (rc == FLUID_OK && c->surface) rc = run_surface_prep(c);
    return rc;
}

int fluid_run_section_group(fluid_ctx* c, int first_section_id, uint32_t count) {
    if (c && !c->driver_step) c->quiet_valid = false;  // a write from outside a step (quiet_bricks.h)
    if (!c) return FLUID_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    if (first_section_id < FLUID_SEC_01A_CLEAR_PARTICLE_DENSITIES ||
        first_section_id > FLUID_SEC_14_PARTICLES ||
        count > (uint32_t)(FLUID_SEC_14_PARTICLES - first_section_id + 1))
        return c->fail(FLUID_ERR_INVALID_ARG,
                       "sections [%d, %d + %u) are not a slice of the step section list",
                       first_section_id, first_section_id, count);
    if (c->is_slab) {
        // between most sections of the list a slab needs ghost planes from its neighbours: only the
        // grouped passes themselves are slices a slab can run in one call
        const bool g0405 = first_section_id == FLUID_SEC_04_COMPUTE_EXTRAPOLATED_VELOCITIES && count == 2;
        const bool g0708 = first_section_id == FLUID_SEC_07_ADVECT && count == 2;
        const bool g0911 = first_section_id == FLUID_SEC_09_DIFFUSE && count == 3 &&
                           c->diffuse_mode == FLUID_DIFFUSE_REFERENCE_EXACT && c->g.W % 4 == 0;
        if (!g0405 && !g0708 && !g0911)
            return slab_unsupported(c, "this slice of the step list in one call (only 04+05, 07+08 "
                                       "and, for fluid_size.x % 4 == 0, 09+10+11)");
    }
    return run_step_slice(c, first_section_id, (int)count, true, false);
}

// ---- a step driven section by section from outside (the Z-slab driver), with fluid_run_step's skipping ----
int fluid_step_begin(fluid_ctx* c, int section_list) {
    if (!c) return FLUID_ERR_INVALID_ARG;
    if (c->driver_step) return c->fail(FLUID_ERR_INVALID_ARG, "fluid_step_begin twice");
    HIP_TRY(c, hipSetDevice(c->device));
    c->driver_step = true;
    // skipping goes with the grouped passes (as in fluid_run_step): not when the caller runs the section list
    c->ds_quiet = !section_list && c->g.W % 4 == 0 && c->opt[FLUID_OPT_STEP_FUSION] == 0 &&
                  fast_loop_possible(c) &&
                  c->diffuse_mode == FLUID_DIFFUSE_REFERENCE_EXACT && c->opt[FLUID_OPT_QUIET_BRICKS] != 1;
    c->ds_early = c->ds_quiet && c->quiet_valid && c->mask_valid && c->pbricks_valid;
    c->quiet_in_use = c->early_in_use = c->early_step = false;
    c->ghost_bricks_valid = false;
    c->box_from_driver = false;
    return FLUID_OK;
}

int fluid_step_end(fluid_ctx* c) {
    if (!c) return FLUID_ERR_INVALID_ARG;
    c->driver_step = false;
    c->quiet_in_use = c->early_in_use = c->early_step = false;
    return FLUID_OK;
}

int fluid_step_build_activity(fluid_ctx* c) {
    if (!c) return FLUID_ERR_INVALID_ARG;
    if (!c->driver_step) return c->fail(FLUID_ERR_INVALID_ARG, "fluid_step_begin first");
    HIP_TRY(c, hipSetDevice(c->device));
    if (!c->ds_quiet) return FLUID_OK;
    return ensure_prepared(c, true);  // mask + activity bricks of the CELL_TYPES 06 just made
}

int fluid_activity_layer_ptr(fluid_ctx* c, int which, void** device_ptr, uint64_t* bytes) {
    if (!c) return FLUID_ERR_INVALID_ARG;
    if (!device_ptr || !bytes) return c->fail(FLUID_ERR_INVALID_ARG, "null output pointer");
    if (which < 0 || which > 3) return c->fail(FLUID_ERR_INVALID_ARG, "activity layer %d", which);
    int nbx, nby, nbz;
    k12_brick_dims(c->g.W, c->g.H, c->g.Dl, nbx, nby, nbz);
    const uint64_t layer = (uint64_t)nbx * nby;
    *bytes = c->ds_quiet || !c->driver_step ? layer : 0;  // nothing to exchange when the step does not skip
    switch (which) {
        case 0: *device_ptr = c->bricks(); break;                                  // own bottom layer
        case 1: *device_ptr = c->bricks() + (uint64_t)(nbz - 1) * layer; break;    // own top layer
        case 2: *device_ptr = c->arena + c->ghost_bricks_offset; break;            // from below
        default: *device_ptr = c->arena + c->ghost_bricks_offset + layer; break;   // from above
    }
    if (which >= 2) c->ghost_bricks_valid = true;  // the caller is about to fill them
    return FLUID_OK;
}

int fluid_step_status(fluid_ctx* c, uint32_t words[8]) {
    if (!c) return FLUID_ERR_INVALID_ARG;
    if (!words) return c->fail(FLUID_ERR_INVALID_ARG, "null output pointer");
    HIP_TRY(c, hipSetDevice(c->device));
    uint32_t v = 0;
    HIP_TRY(c, hipMemcpyAsync(&v, c->flags(), 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));  // also lands the brick summary of ensure_prepared
    if (v) HIP_TRY(c, hipMemsetAsync(c->flags(), 0, 4, c->stream));
    memset(words, 0, 8 * sizeof(uint32_t));
    words[0] = v;
    if (c->box_pending && c->brick_count_host) {
        c->box_pending = false;
        const uint32_t* h = c->brick_count_host;
        int bx, by, bz;
        k12_brick_cells(bx, by, bz);
        words[1] = 1;                                             // the box below is known
        words[2] = h[0];                                          // bricks with water
        words[3] = h[1] * (uint32_t)by;                           // rows [3], [4)
        words[4] = std::min<uint32_t>(h[2] * (uint32_t)by, (uint32_t)c->g.H);
        words[5] = h[5];                                          // cells [5], [6) along x
        words[6] = std::min<uint32_t>(h[6], (uint32_t)c->g.W);
        c->box.z_lo = (int)h[3] * bz;                             // local planes: this context's own
        c->box.z_hi = std::min((int)h[4] * bz, c->g.Dl);
        // ... and for the driver: which owned planes hold water, [lo, hi) in the two halves of the word (the
        // slab driver leaves the loop's exchanges out at a face no water is near on either side)
        words[7] = ((uint32_t)c->box.z_lo & 0xFFFFu) | ((uint32_t)std::min(c->box.z_hi, 0xFFFF) << 16);
        c->box.fraction = (float)h[0] / (float)c->active_bytes;
    }
    return FLUID_OK;
}

int fluid_step_set_box(fluid_ctx* c, int valid, uint32_t own_bricks, uint32_t y_lo, uint32_t y_hi,
                       uint32_t x_lo, uint32_t x_hi) {
    if (!c) return FLUID_ERR_INVALID_ARG;
    c->box.valid = valid != 0;
    c->box_from_driver = valid != 0;
    if (!valid) return FLUID_OK;
    // own_bricks == 0: this context has no water cell, its launches write nothing (an empty y range says so)
    c->box.y_lo = own_bricks ? (int)y_lo : 0;
    c->box.y_hi = own_bricks ? (int)std::min<uint32_t>(y_hi, (uint32_t)c->g.H) : 0;
    c->box.x_lo = (int)x_lo;
    c->box.x_hi = (int)std::min<uint32_t>(x_hi, (uint32_t)c->g.W);
    // z: this context's own planes with water (fluid_step_status left them in the box)
    return FLUID_OK;
}

int fluid_sync(fluid_ctx* c) {
    if (!c) return FLUID_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return FLUID_OK;
}

int fluid_enable_timing(fluid_ctx* c, int enabled) {
    if (!c) return FLUID_ERR_INVALID_ARG;
    c->timing = enabled != 0;
    return FLUID_OK;
}

int fluid_section_time_ms(fluid_ctx* c, int section_id, double* total_ms, uint64_t* calls) {
    if (!c) return FLUID_ERR_INVALID_ARG;
    if (section_id < 0 || section_id >= FLUID_SECTION_COUNT)
        return c->fail(FLUID_ERR_INVALID_ARG, "unknown section id %d", section_id);
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = fold_timers(c);
    if (rc) return rc;
    if (total_ms) *total_ms = c->sec_ms[section_id];
    if (calls) *calls = c->sec_calls[section_id];
    return FLUID_OK;
}

int fluid_reset_timing(fluid_ctx* c) {
    if (!c) return FLUID_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = fold_timers(c);
    if (rc) return rc;
    for (int i = 0; i < FLUID_SECTION_COUNT; i++) {
        c->sec_ms[i] = 0;
        c->sec_calls[i] = 0;
    }
    return FLUID_OK;
}

int fluid_image_plane_ptr(fluid_ctx* c, int image_id, int32_t plane, void** device_ptr,
                          uint64_t* bytes) {
    if (!c) return FLUID_ERR_INVALID_ARG;
    if (!device_ptr || !bytes) return c->fail(FLUID_ERR_INVALID_ARG, "null output pointer");
    int rc = check_image(c, image_id);
    if (rc) return rc;
    if (image_id >= 8) {  // detailed grid: planes of W*H*res^2 texels, no ghost planes
        if (plane < 0 || plane >= c->sk.D)
            return c->fail(FLUID_ERR_INVALID_ARG, "plane %d outside [0, %d)", plane, c->sk.D);
        *bytes = (uint64_t)c->sk.plane * 4;
        *device_ptr = c->surf<uint8_t>(image_id) + (uint64_t)plane * *bytes;
        return FLUID_OK;
    }
    if (plane < -IMG_GHOST || plane >= c->g.Dl + IMG_GHOST)
        return c->fail(FLUID_ERR_INVALID_ARG, "plane %d outside [%d, %d)", plane, -IMG_GHOST,
                       c->g.Dl + IMG_GHOST);
    const uint64_t pb = (uint64_t)c->g.plane * c->img[image_id].elem_bytes;
    *device_ptr = c->arena + c->img[image_id].offset + (uint64_t)(plane + IMG_GHOST) * pb;
    *bytes = pb;
    return FLUID_OK;
}

int fluid_notify_image_written(fluid_ctx* c, int image_id) {
    if (c) c->quiet_valid = false;  // a write from outside fluid_run_step (quiet_bricks.h)
    if (!c) return FLUID_ERR_INVALID_ARG;
    int rc = check_image(c, image_id);
    if (rc) return rc;
    c->touched(image_id);  // derived data (neighbour mask, b_i, working-buffer constants) is rebuilt
    return FLUID_OK;
}

int fluid_notify_ghost_planes_written(fluid_ctx* c, int image_id) {
    if (c && !c->driver_step) c->quiet_valid = false;  // a write from outside a step (quiet_bricks.h)
    if (!c) return FLUID_ERR_INVALID_ARG;
    int rc = check_image(c, image_id);
    if (rc) return rc;
    const bool w0 = c->v1_w_zero;  // a property of the owned planes
    c->touched(image_id);
    c->v1_w_zero = w0;
    return FLUID_OK;
}

int fluid_pressure_residual(fluid_ctx* c, int image_id, float* max_abs, double* sum_squares,
                            uint64_t* water_cells) {
    if (!c) return FLUID_ERR_INVALID_ARG;
    if (image_id != FLUID_IMG_PRESSURES_1 && image_id != FLUID_IMG_PRESSURES_2)
        return c->fail(FLUID_ERR_INVALID_ARG, "image %d is not a pressure image", image_id);
    HIP_TRY(c, hipSetDevice(c->device));
    struct {
        uint32_t max_bits, pad;
        double sum_sq;
        unsigned long long cells;
    } host;
    static_assert(sizeof host == 24, "ResidualOut layout");
    void* dev = c->flags() + 16;  // 8-byte aligned scratch inside the flags block
    HIP_TRY(c, hipMemsetAsync(dev, 0, 32, c->stream));
    k12_launch_residual(c->stream, c->plane0<uint8_t>(FLUID_IMG_CELL_TYPES),
                        c->plane0<float>(FLUID_IMG_DIVERGENCES), c->plane0<float>(image_id), c->g,
                        c->pk, dev);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(&host, dev, sizeof host, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (max_abs) memcpy(max_abs, &host.max_bits, 4);
    if (sum_squares) *sum_squares = host.sum_sq;
    if (water_cells) *water_cells = host.cells;
    return FLUID_OK;
}

int fluid_count_nonfinite(fluid_ctx* c, int image_id, uint64_t* count) {
    if (!c) return FLUID_ERR_INVALID_ARG;
    if (!count) return c->fail(FLUID_ERR_INVALID_ARG, "null output pointer");
    int rc = check_image(c, image_id);
    if (rc) return rc;
    const bool is_float = image_id == FLUID_IMG_VELOCITIES_1 || image_id == FLUID_IMG_VELOCITIES_2 ||
                          image_id == FLUID_IMG_PRESSURES_1 || image_id == FLUID_IMG_PRESSURES_2 ||
                          image_id == FLUID_IMG_DIVERGENCES ||
                          image_id == FLUID_IMG_PARTICLE_DENSITIES_FLOAT_1 ||
                          image_id == FLUID_IMG_PARTICLE_DENSITIES_FLOAT_2;
    if (!is_float) return c->fail(FLUID_ERR_INVALID_ARG, "image %d does not hold fp32 texels", image_id);
    HIP_TRY(c, hipSetDevice(c->device));
    unsigned long long* dev = reinterpret_cast<unsigned long long*>(c->flags() + 24);  // 8-byte aligned
    HIP_TRY(c, hipMemsetAsync(dev, 0, 8, c->stream));
    const int64_t words = (int64_t)(image_host_bytes(c, image_id) / 4);
    if (words > 0) {
        const int blocks = (int)std::min<int64_t>((words + 255) / 256, 256 * 8);
        hipLaunchKernelGGL(k_count_nonfinite, dim3(blocks), dim3(256), 0, c->stream,
                           reinterpret_cast<const uint32_t*>(image_host_base(c, image_id)), words, dev);
        HIP_TRY(c, hipGetLastError());
    }
    unsigned long long host = 0;
    HIP_TRY(c, hipMemcpyAsync(&host, dev, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *count = host;
    return FLUID_OK;
}

int fluid_extract_surface(fluid_ctx* c, int image_id, float* host_triangles, uint64_t capacity,
                          uint64_t* count) {
    if (!c) return FLUID_ERR_INVALID_ARG;
    if (!count) return c->fail(FLUID_ERR_INVALID_ARG, "null output pointer");
    if (!c->surface)
        return c->fail(FLUID_ERR_UNSUPPORTED, "surface extraction needs a surface_prep context");
    if (image_id != FLUID_IMG_PARTICLE_DENSITIES_FLOAT_1 && image_id != FLUID_IMG_PARTICLE_DENSITIES_FLOAT_2)
        return c->fail(FLUID_ERR_INVALID_ARG, "image %d is not a float density image", image_id);
    if (!c->mc_loaded[0] || !c->mc_loaded[1])
        return c->fail(FLUID_ERR_INVALID_ARG,
                       "upload MARCHING_CUBES_COUNTS_BUF and MARCHING_CUBES_EDGES_BUF first (the reference's "
                       "surface_render_data/polygon_counts.txt / polygon_edge_indices.txt)");
    if (capacity && !host_triangles) return c->fail(FLUID_ERR_INVALID_ARG, "null triangle buffer");
    HIP_TRY(c, hipSetDevice(c->device));
    void* dev = nullptr;  // triangles + the counter behind them
    const uint64_t tri_bytes = capacity * 48;
    if (hipMalloc(&dev, tri_bytes + 8) != hipSuccess)
        return c->fail(FLUID_ERR_OUT_OF_MEMORY, "%llu bytes for the triangle list",
                       (unsigned long long)(tri_bytes + 8));
    auto* total = reinterpret_cast<unsigned long long*>(static_cast<char*>(dev) + tri_bytes);
    int rc = FLUID_OK;
    unsigned long long found = 0;
    const SurfK& s = c->sk;
    hipError_t e = hipMemsetAsync(total, 0, 8, c->stream);
    if (e == hipSuccess && s.W > 1 && s.H > 1 && s.D > 1) {
        hipLaunchKernelGGL(k31_extract_surface, dim3((s.W - 1 + 63) / 64, (s.H - 1 + 3) / 4, s.D - 1),
                           dim3(64, 4, 1), 0, c->stream, c->surf<float>(image_id), s, c->mc_tables,
                           c->mc_tables + 256, static_cast<float*>(dev), (unsigned long long)capacity, total);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&found, total, 8, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    const uint64_t stored = std::min<uint64_t>(found, capacity);
    if (e == hipSuccess && stored)
        e = hipMemcpy(host_triangles, dev, stored * 48, hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = c->fail(FLUID_ERR_HIP, "surface extraction: %s", hipGetErrorString(e));
    (void)hipFree(dev);
    *count = found;
    return rc;
}

int fluid_get_stat(fluid_ctx* c, int stat, uint64_t* value) {
    if (!c) return FLUID_ERR_INVALID_ARG;
    if (!value) return c->fail(FLUID_ERR_INVALID_ARG, "null output pointer");
    HIP_TRY(c, hipSetDevice(c->device));
    switch (stat) {
        case FLUID_STAT_BRICKS:
            *value = c->active_bytes;
            return FLUID_OK;
        case FLUID_STAT_QUIET_BRICKS: {
            uint32_t v = 0;
            if (c->quiet_valid) {
                HIP_TRY(c, hipMemcpyAsync(&v, c->flags() + 9, 4, hipMemcpyDeviceToHost, c->stream));
                HIP_TRY(c, hipStreamSynchronize(c->stream));
            }
            *value = v;
            return FLUID_OK;
        }
        case FLUID_STAT_PARTICLE_SORTS:
            *value = c->ps.sorts;
            return FLUID_OK;
        case FLUID_STAT_PARTICLE_BINNED:
            *value = c->ps.binned ? 1 : 0;
            return FLUID_OK;
        case FLUID_STAT_PARTICLE_ENTRIES:
            *value = walk_entries(c);
            return FLUID_OK;
        case FLUID_STAT_OWNED_SQUEEZES:
            *value = c->loc.squeezes;
            return FLUID_OK;
        case FLUID_STAT_PARTICLE_STRAYS: {
            uint32_t v[2] = {0, 0};
            if (c->ps.binned && c->ps.stray_count) {
                HIP_TRY(c, hipMemcpyAsync(v, c->ps.stray_count, 8, hipMemcpyDeviceToHost, c->stream));
                HIP_TRY(c, hipStreamSynchronize(c->stream));
            }
            *value = v[1];
            return FLUID_OK;
        }
        default:
            return c->fail(FLUID_ERR_INVALID_ARG, "unknown statistic %d", stat);
    }
}

int fluid_slab_status(fluid_ctx* c, uint32_t* halo_violation) {
    if (!c) return FLUID_ERR_INVALID_ARG;
    if (!halo_violation) return c->fail(FLUID_ERR_INVALID_ARG, "null output pointer");
    HIP_TRY(c, hipSetDevice(c->device));
    uint32_t v = 0;
    HIP_TRY(c, hipMemcpyAsync(&v, c->flags(), 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *halo_violation = v;
    if (v) HIP_TRY(c, hipMemsetAsync(c->flags(), 0, 4, c->stream));
    return FLUID_OK;
}

// ---- the velocity sampler on Z slabs: how many ghost planes are current, and the fallback when the
// ---- fluid moves further than that in one step ------------------------------------------
int fluid_set_sampler_halo(fluid_ctx* c, uint32_t planes) {
    if (!c) return FLUID_ERR_INVALID_ARG;
    if (planes < 1 || planes > (uint32_t)IMG_GHOST)
        return c->fail(FLUID_ERR_INVALID_ARG, "sampler halo of %u planes: 1..%d", planes, IMG_GHOST);
    c->g.sg_lo = c->g.sg_hi = (int)planes;
    return FLUID_OK;
}

int fluid_sampler_reach(fluid_ctx* c, uint32_t* planes) {
    if (!c) return FLUID_ERR_INVALID_ARG;
    if (!planes) return c->fail(FLUID_ERR_INVALID_ARG, "null output pointer");
    HIP_TRY(c, hipSetDevice(c->device));
    uint32_t* dev = c->flags() + 26;
    HIP_TRY(c, hipMemsetAsync(dev, 0, 4, c->stream));
    const int64_t cells = (int64_t)c->owned_cells();
    if (cells > 0) {
        const int blocks = (int)std::min<int64_t>((cells + 255) / 256, 256 * 8);
        hipLaunchKernelGGL(k_max_abs_vz, dim3(blocks), dim3(256), 0, c->stream,
                           c->plane0<float4>(FLUID_IMG_VELOCITIES_1), cells, dev);
        HIP_TRY(c, hipGetLastError());
    }
    uint32_t bits = 0;
    HIP_TRY(c, hipMemcpyAsync(&bits, dev, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    float vmax;
    memcpy(&vmax, &bits, 4);
    // advect.comp:75-77: the back-traced point lies x = |v.z * dt| from the face position (v.z a trilinear
    // blend of texels, each |.| <= vmax up to rounding: the factor) and its two z taps are the planes
    // around it, floor(x) + 1 planes from the cell's own at most; one more for the roundings of the
    // coordinate arithmetic.  (The driver checks the flag again after the redone pass in any case.)
    const double reach = (double)vmax * (double)fabsf(c->pk.dt) * (1.0 + 1e-5);
    uint32_t n = (uint32_t)c->g.Dg;
    if (vmax == vmax && reach < (double)c->g.Dg) n = std::min<uint32_t>((uint32_t)reach + 2u, n);
    *planes = n;
    return FLUID_OK;
}

int fluid_sampler_wide_begin(fluid_ctx* c, uint32_t below, uint32_t above) {
    if (!c) return FLUID_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    // planes that exist in the grid: the sampler clamps its taps to [0, Dg)
    const uint32_t lo = std::min<uint32_t>(below, (uint32_t)c->g.z0);
    const uint32_t hi = std::min<uint32_t>(above, (uint32_t)(c->g.Dg - c->g.z0 - c->g.Dl));
    const uint64_t bytes = (uint64_t)(lo + (uint32_t)c->g.Dl + hi) * (uint64_t)c->g.plane * 16u;
    if (bytes > c->wide_bytes) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (c->wide) (void)hipFree(c->wide);
    if (c->mc_tables) (void)hipFree(c->mc_tables);
        c->wide = nullptr;
        c->wide_bytes = 0;
        void* ptr = nullptr;
        if (hipMalloc(&ptr, bytes) != hipSuccess)
            return c->fail(FLUID_ERR_OUT_OF_MEMORY, "wide sampler source of %llu bytes",
                           (unsigned long long)bytes);
        c->wide = static_cast<float4*>(ptr);
        c->wide_bytes = bytes;
    }
    c->wide_lo = (int)lo;
    c->wide_hi = (int)hi;
    HIP_TRY(c, hipMemcpyAsync(c->wide + (uint64_t)lo * c->g.plane, c->plane0<float4>(FLUID_IMG_VELOCITIES_1),
                              c->owned_cells() * 16u, hipMemcpyDeviceToDevice, c->stream));
    return FLUID_OK;
}

int fluid_sampler_wide_plane_ptr(fluid_ctx* c, int32_t plane, void** device_ptr, uint64_t* bytes) {
    if (!c) return FLUID_ERR_INVALID_ARG;
    if (!device_ptr || !bytes) return c->fail(FLUID_ERR_INVALID_ARG, "null output pointer");
    if (!c->wide) return c->fail(FLUID_ERR_INVALID_ARG, "fluid_sampler_wide_begin first");
    if (plane < -c->wide_lo || plane >= c->g.Dl + c->wide_hi)
        return c->fail(FLUID_ERR_INVALID_ARG, "plane %d outside [%d, %d)", plane, -c->wide_lo,
                       c->g.Dl + c->wide_hi);
    *bytes = (uint64_t)c->g.plane * 16u;
    *device_ptr = c->wide + (uint64_t)(plane + c->wide_lo) * c->g.plane;
    return FLUID_OK;
}

int fluid_run_advect_wide(fluid_ctx* c, int with_forces) {
    if (c) c->quiet_valid = false;  // the pass below processes every cell: the streaks start over
    if (!c) return FLUID_ERR_INVALID_ARG;
    if (!c->wide) return c->fail(FLUID_ERR_INVALID_ARG, "fluid_sampler_wide_begin first");
    HIP_TRY(c, hipSetDevice(c->device));
    SectionTimer tm{c};
    int rc = tm.begin(FLUID_SEC_07_ADVECT);
    if (rc) return rc;
    GridK g = c->g;
    g.sg_lo = c->wide_lo;
    g.sg_hi = c->wide_hi;
    const float4* src = c->wide + (uint64_t)c->wide_lo * c->g.plane;  // owned plane 0 of the wide source
    BrickK bk;
    k12_brick_dims(g.W, g.H, g.Dl, bk.nbx, bk.nby, bk.nbz);
    HIP_TRY(c, hipMemsetAsync(c->flags(), 0, 4, c->stream));  // the pass that raised it is being redone
    const uint8_t* T = c->plane0<uint8_t>(FLUID_IMG_CELL_TYPES);
    float4* V2 = c->plane0<float4>(FLUID_IMG_VELOCITIES_2);
    if (with_forces)
        hipLaunchKernelGGL(k07_advect<true>, cell_grid(g), cell_block(), 0, c->stream, T, src, V2, g, c->pk,
                           c->flags(), (const uint8_t*)nullptr, bk, 1);
    else
        hipLaunchKernelGGL(k07_advect<false>, cell_grid(g), cell_block(), 0, c->stream, T, src, V2, g,
                           c->pk, c->flags(), (const uint8_t*)nullptr, bk, 1);
    HIP_TRY(c, hipGetLastError());
    return tm.end();
}

// ---- particle hand-over between Z-neighbours ------------------------------------------------------------
int fluid_particles_migrate_list(fluid_ctx* c, int which, void** device_list, uint32_t* capacity) {
    if (!c) return FLUID_ERR_INVALID_ARG;
    if (!device_list || !capacity) return c->fail(FLUID_ERR_INVALID_ARG, "null output pointer");
    if (which < 0 || which > 3) return c->fail(FLUID_ERR_INVALID_ARG, "migration list %d", which);
    *device_list = c->leavers(which);
    *capacity = c->leavers_capacity;
    return FLUID_OK;
}

static int read_migrate_counts(fluid_ctx* c, uint32_t counts[2], uint32_t* left_behind) {
    uint32_t n[2] = {0, 0};
    HIP_TRY(c, hipMemcpyAsync(n, c->flags() + 28, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    uint32_t over = 0;
    for (int d = 0; d < 2; d++) {
        if (n[d] > c->leavers_capacity) {
            over += n[d] - c->leavers_capacity;
            n[d] = c->leavers_capacity;
            // the device counter keeps counting past the capacity: put it back for whoever appends next
            HIP_TRY(c, hipMemcpyAsync(c->flags() + 28 + d, &c->leavers_capacity, 4, hipMemcpyHostToDevice,
                                      c->stream));
        }
        counts[d] = n[d];
    }
    if (left_behind) *left_behind = over;
    return FLUID_OK;
}

int fluid_particles_collect(fluid_ctx* c, int reset_lists, uint32_t counts[2], uint32_t* left_behind) {
    if (!c) return FLUID_ERR_INVALID_ARG;
    if (!counts || !left_behind) return c->fail(FLUID_ERR_INVALID_ARG, "null output pointer");
    HIP_TRY(c, hipSetDevice(c->device));
    counts[0] = counts[1] = 0;
    *left_behind = 0;
    if (!c->is_slab || c->particle_capacity == 0) return FLUID_OK;
    if (reset_lists) {
        HIP_TRY(c, hipMemsetAsync(c->flags() + 28, 0, 8, c->stream));
        const int rc_own = local_squeeze_if_needed(c, 0);  // once per step: the holes of the steps before
        if (rc_own) return rc_own;
    }
    const uint32_t n = c->loc.on ? c->loc.n : 0u;
    if (n) {
        hipLaunchKernelGGL(k_particles_collect_leavers, dim3((n + 255) / 256), dim3(256), 0, c->stream,
                           c->loc.buf[c->loc.cur], c->loc.pid[c->loc.cur], n, c->g, c->migrate_lists(), c->loc.counters);
        HIP_TRY(c, hipGetLastError());
    }
    int rc = read_migrate_counts(c, counts, left_behind);
    if (rc == FLUID_OK) rc = local_read(c);
    return rc;
}

int fluid_particles_adopt_received(fluid_ctx* c, uint32_t from_below, uint32_t from_above,
                                   uint32_t forwarded[2]) {
    if (!c) return FLUID_ERR_INVALID_ARG;
    if (!forwarded) return c->fail(FLUID_ERR_INVALID_ARG, "null output pointer");
    if (from_below > c->leavers_capacity || from_above > c->leavers_capacity)
        return c->fail(FLUID_ERR_INVALID_ARG, "%u / %u received entries, the lists hold %u", from_below,
                       from_above, c->leavers_capacity);
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemsetAsync(c->flags() + 28, 0, 8, c->stream));  // the send lists have been sent
    if (!c->loc.on) return c->fail(FLUID_ERR_INVALID_ARG, "no particle storage yet: 00_init_particles or an upload first");
    {   // room for what may be adopted: squeeze the holes out if that makes it, grow the arrays otherwise
        int rc_own = local_squeeze_if_needed(c, from_below + from_above);
        if (rc_own == FLUID_OK) rc_own = local_reserve(c, (uint64_t)c->loc.n + from_below + from_above);
        if (rc_own) return rc_own;
    }
    const uint32_t n[2] = {from_below, from_above};
    for (int src = 0; src < 2; src++) {
        if (n[src] == 0) continue;
        // from below: travelling up (dir 1); from above: travelling down (dir 0)
        hipLaunchKernelGGL(k_particles_adopt, dim3((n[src] + 255) / 256), dim3(256), 0, c->stream, c->g,
                           c->leavers(2 + src), n[src], src == 0 ? 1 : 0, c->migrate_lists(), c->compact());
        HIP_TRY(c, hipGetLastError());
    }
    uint32_t over = 0;
    int rc = read_migrate_counts(c, forwarded, &over);
    if (rc == FLUID_OK) rc = local_read(c);
    if (rc) return rc;
    if (over)  // cannot happen while every rank's lists have the same capacity (forwarded <= received)
        return c->fail(FLUID_ERR_OUT_OF_MEMORY, "%u forwarded particles did not fit the send lists", over);
    return FLUID_OK;
}

int fluid_get_geometry(const fluid_ctx* c, uint32_t global_size[3], uint32_t* z_begin,
                       uint32_t* z_count, uint64_t* capacity) {
    if (!c) return FLUID_ERR_INVALID_ARG;
    if (global_size) {
        global_size[0] = (uint32_t)c->g.W;
        global_size[1] = (uint32_t)c->g.H;
        global_size[2] = (uint32_t)c->g.Dg;
    }
    if (z_begin) *z_begin = (uint32_t)c->g.z0;
    if (z_count) *z_count = (uint32_t)c->g.Dl;
    if (capacity) *capacity = c->particle_capacity;
    return FLUID_OK;
}

}  // extern "C"
