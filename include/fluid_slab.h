/*
 * fluid_slab.h — C ABI of the multi-GPU driver: the solver path of Matezzzz/vulkan-3d-fluid-simulation
 * with the grid cut into Z slabs, one process (and one fluid_ctx, fluid_engine.h) per GPU.
 *
 * The reference is single-GPU: main.cpp:103-111 builds the section lists once, :156-177 runs
 * SimulationStepSections per frame.  This header is the same frame loop for a rank of a multi-GPU run:
 *
 *   SimulationInitializationSections::run   main.cpp:111   -> fluid_slab_run_init()
 *   SimulationStepSections::run             main.cpp:172   -> fluid_slab_run_step()
 *   the 12_solve_pressure loop section alone (fluid_flow_sections.h:298-313) -> fluid_slab_pressure_step()
 *
 * with everything a slab needs from its neighbours done inside: ghost planes of the images the stencils
 * and the velocity sampler read, deep halos for the Jacobi loop (h planes every h sweeps, the pass before
 * and after an exchange split so that the planes travel beside compute), hand-over of particles that
 * cross a slab face, and the wider sampler halo when the fluid is fast.  Results are those
 * of the single-GPU step bit for bit.
 *
 * Transport.  Planes travel point-to-point between Z-neighbours only (2 of the 7 xGMI links of a GPU);
 * the only collectives are 4-byte-per-rank MAX reductions (is any rank's sampler short of planes / are
 * particles still in flight).  Two transports:
 *   - RCCL (the product): fluid_slab_attach_rccl() creates this rank's communicator from a unique id
 *     the caller distributes; exchanges are ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd on the
 *     engine's stream (or on a communication stream ordered by events, for the overlapped ones);
 *   - callbacks (fluid_slab_transport): the same schedule over any byte mover — the multi-process CPU tests
 *     run it over gloo on host memory, the one-GPU rehearsals stage device planes through the host.
 * Likewise the per-slab compute is the HIP engine (fluid_slab_create) or, for the CPU tests only, a table
 * of callbacks (fluid_slab_create_custom) behind which the tests put the CPU oracle on poisoned arrays.
 *
 * Conventions as in fluid_engine.h: plain C, 0 or a negative fluid_status, fluid_slab_last_error() for
 * text, one host thread per driver.  A "plane" is a local z index: 0 .. z_count-1 owned, negative and
 * >= z_count ghost planes.
 */
#ifndef FLUID_SLAB_H
#define FLUID_SLAB_H

#include "fluid_engine.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fluid_slab fluid_slab;

/* Balanced contiguous split of `depth` planes over `world` ranks: the first depth % world ranks get one
 * plane more.  FLUID_ERR_INVALID_ARG unless 1 <= world <= depth and rank < world. */
int fluid_slab_partition(uint32_t depth, uint32_t world, uint32_t rank, uint32_t* z_begin,
                         uint32_t* z_count);

/* ---- transport by callbacks ------------------------------------------------------------------------- */
#define FLUID_XFER_SEND 1u        /* else receive                                                        */
#define FLUID_XFER_HOST_MEMORY 2u /* ptr is host memory even though the backend computes on the device   */
typedef struct fluid_slab_xfer {
    void* ptr;
    uint64_t bytes;
    int32_t peer;   /* rank */
    uint32_t flags; /* FLUID_XFER_* */
} fluid_slab_xfer;

typedef struct fluid_slab_transport {
    uint32_t struct_bytes;
    uint32_t reserved;
    void* user;
    /* Move all transfers of the list (every rank calls with the matching list) and return when the
     * received bytes are in place.  The driver has synchronised the compute stream before the call. */
    int (*exchange)(void* user, const fluid_slab_xfer* ops, uint32_t count);
    /* values[i] = MAX over ranks of values[i], in place, host memory. */
    int (*allreduce_max_u32)(void* user, uint32_t* values, uint32_t count);
} fluid_slab_transport;

/* ---- per-slab compute by callbacks (tests) ------------------------------------------------------------
 * The calls the driver makes on the engine, one to one (fluid_engine.h names in brackets).  Pointers the
 * callbacks return are what the transport is given. */
typedef struct fluid_slab_loop_buffer {
    int32_t which;   /* id for loop_planes */
    uint32_t planes; /* boundary planes per side to exchange */
} fluid_slab_loop_buffer;

typedef struct fluid_slab_backend {
    uint32_t struct_bytes;
    uint32_t reserved;
    void* user;
    int (*run_section)(void* user, int section_id);                          /* fluid_run_section       */
    int (*run_section_group)(void* user, int first_section_id, uint32_t n);  /* fluid_run_section_group */
    int (*image_planes)(void* user, int image_id, int32_t first_plane, uint32_t count, void** ptr,
                        uint64_t* bytes);                                    /* fluid_image_plane_ptr   */
    int (*ghost_planes_written)(void* user, int image_id);     /* fluid_notify_ghost_planes_written */
    /* the loop section in explicit form (fluid_pressure_loop_*).  begin: `halo` = planes per exchange;
     * fills the buffers whose boundary planes must be exchanged before the first sweep. */
    int (*loop_limits)(void* user, uint32_t* max_sweeps, uint32_t* max_halo);
    int (*loop_begin)(void* user, uint32_t halo, fluid_slab_loop_buffer out[4], uint32_t* count);
    int (*loop_halo_exchanged)(void* user, uint32_t halo, int first);
    int (*loop_advance)(void* user, uint32_t first_sweep, uint32_t sweeps, int keep_intermediate, int part,
                        int32_t interior_begin, int32_t interior_end, int* written_buffer);
    int (*loop_end)(void* user);
    int (*loop_planes)(void* user, int which, int32_t first_plane, uint32_t count, void** ptr,
                       uint64_t* bytes);                                     /* .._loop_plane_ptr       */
    /* velocity sampler halo (fluid_slab_status, fluid_set_sampler_halo, fluid_sampler_*) */
    int (*slab_status)(void* user, uint32_t* halo_violation);
    int (*set_sampler_halo)(void* user, uint32_t planes);
    int (*sampler_reach)(void* user, uint32_t* planes);
    int (*sampler_wide_begin)(void* user, uint32_t below, uint32_t above);
    int (*sampler_wide_planes)(void* user, int32_t first_plane, uint32_t count, void** ptr, uint64_t* bytes);
    int (*run_advect_wide)(void* user, int with_forces);
    /* particle hand-over (fluid_particles_*) */
    int (*migrate_list)(void* user, int which, void** list, uint32_t* capacity);
    int (*collect)(void* user, int reset_lists, uint32_t counts[2], uint32_t* left_behind);
    int (*adopt_received)(void* user, uint32_t from_below, uint32_t from_above, uint32_t forwarded[2]);
    int (*sync)(void* user);                                                 /* fluid_sync              */
    /* a step with the engine's skipping (fluid_step_*); a backend without it answers bytes = 0 for the
     * activity layers and words[1] = 0 (box unknown) */
    int (*step_begin)(void* user, int section_list);
    int (*step_end)(void* user);
    int (*build_activity)(void* user);
    int (*activity_layer)(void* user, int which, void** ptr, uint64_t* bytes);
    int (*step_status)(void* user, uint32_t words[8]);
    int (*set_box)(void* user, int valid, uint32_t own_bricks, uint32_t y_lo, uint32_t y_hi, uint32_t x_lo,
                   uint32_t x_hi);
} fluid_slab_backend;

/* ---- creation ---------------------------------------------------------------------------------------- */
typedef enum fluid_slab_overlap {
    FLUID_SLAB_OVERLAP_NONE = 0,   /* halo exchanges of the Jacobi loop in line                           */
    FLUID_SLAB_OVERLAP_BEFORE = 1, /* the pass before an exchange is split: the planes to send first, the
                                      exchange starts, the planes in between follow                       */
    FLUID_SLAB_OVERLAP_BOTH = 2    /* ... and the pass after it: the planes that need no ghost data while
                                      the exchange is in flight, the rest once it has landed (default)    */
} fluid_slab_overlap;

typedef struct fluid_slab_create_info {
    uint32_t struct_bytes;        /* = sizeof(fluid_slab_create_info)                                   */
    uint32_t rank, world;         /* this process and the number of slabs                               */
    int32_t device;               /* HIP device ordinal, -1 = current (engine-backed drivers)           */
    const void* params_blob;      /* 264 bytes, fluid_params; fluid_size = the GLOBAL grid              */
    uint64_t particle_capacity;   /* as fluid_create_info                                               */
    uint32_t pressure_iterations; /* 0 = 200                                                            */
    uint32_t halo_depth;          /* Jacobi loop: planes per exchange = sweeps between exchanges; 0 = 8;
                                     clipped to FLUID_LOOP_MAX_HALO and the thinnest slab, made even     */
    int32_t overlap;              /* fluid_slab_overlap; -1 = default                                   */
    uint32_t section_list;        /* 1 = one kernel per section; 0 = grouped passes 04+05, 07+08,
                                     09+10+11 (default)                                                 */
    int32_t diffuse_mode;         /* fluid_diffuse_mode                                                 */
    uint32_t sampler_halo;        /* ghost planes of VELOCITIES_1 exchanged for 07 as a matter of course,
                                     1..FLUID_IMAGE_GHOST_PLANES; 0 = 2 (flows below one cell per step);
                                     the driver widens it by itself when a step needs more               */
} fluid_slab_create_info;

/* The engine on this rank's slab (fluid_slab_partition) with its own stream and memory. */
int fluid_slab_create(fluid_slab** out, const fluid_slab_create_info* info);
/* The same schedule over a table of compute callbacks (copied).  Test infrastructure. */
int fluid_slab_create_custom(fluid_slab** out, const fluid_slab_create_info* info,
                             const fluid_slab_backend* backend);
void fluid_slab_destroy(fluid_slab* s);
const char* fluid_slab_last_error(const fluid_slab* s);

/* The engine context behind an engine-backed driver (uploads, downloads, options, timing); NULL for a
 * custom backend.  Owned by the driver. */
fluid_ctx* fluid_slab_engine(fluid_slab* s);
int fluid_slab_get_slab(const fluid_slab* s, uint32_t* z_begin, uint32_t* z_count);

/* ---- transport ---------------------------------------------------------------------------------------- */
#define FLUID_SLAB_RCCL_ID_BYTES 128
/* Rank 0 makes the id (ncclGetUniqueId), the caller gives every rank a copy (MPI, a file, a TCP store,
 * torch.distributed ...), every rank attaches (ncclCommInitRank: collective).  librccl is loaded when
 * first needed (dlopen): a build without it fails here, loudly, and nowhere else. */
int fluid_slab_rccl_unique_id(void* id_out /* FLUID_SLAB_RCCL_ID_BYTES */);
int fluid_slab_attach_rccl(fluid_slab* s, const void* id /* FLUID_SLAB_RCCL_ID_BYTES */);
int fluid_slab_attach_transport(fluid_slab* s, const fluid_slab_transport* transport);
/* world == 1 needs no transport.  A single-rank rehearsal of an interior rank's work (both neighbours
 * present, every received plane filled by a device copy of a plane being sent): */
int fluid_slab_attach_loopback(fluid_slab* s, int has_lower, int has_upper);
/* The same rehearsal through RCCL itself: a communicator of one, every plane range goes out with ncclSend and
 * comes back with ncclRecv (to / from this rank, inside one group: the i-th receive gets the i-th send) on
 * the streams a real run uses.  What one GPU can check of the wire: the library that gets loaded, the calls,
 * their stream order around split passes.  Bit-identical to fluid_slab_attach_loopback. */
int fluid_slab_attach_rccl_self(fluid_slab* s, int has_lower, int has_upper);

/* ---- the frame loop ----------------------------------------------------------------------------------- */
int fluid_slab_run_init(fluid_slab* s);
int fluid_slab_run_step(fluid_slab* s);
/* 12a, 12b and the loop section: what bench.py --gpus N times. */
int fluid_slab_pressure_step(fluid_slab* s);
/* The loop section alone, `iterations` dispatches from whatever PRESSURES_1 / _2 hold (0 = the driver's
 * pressure_iterations). */
int fluid_slab_solve(fluid_slab* s, uint32_t iterations);
/* Exchange `planes` ghost planes per side of an image with the Z-neighbours (after an upload). */
int fluid_slab_exchange_image(fluid_slab* s, int image_id, uint32_t planes);

typedef enum fluid_slab_option {
    FLUID_SLAB_OPT_OVERLAP = 0,      /* fluid_slab_overlap */
    FLUID_SLAB_OPT_HALO_DEPTH = 1,
    FLUID_SLAB_OPT_SAMPLER_HALO = 2, /* reset the adaptive sampler halo to this many planes */
    FLUID_SLAB_OPT_COUNT
} fluid_slab_option;
int fluid_slab_set_option(fluid_slab* s, int option, int64_t value);

/* Measure the loop's exchange schedules where they run and adopt the fastest: for every halo depth in {8, 6, 3}
 * the slabs allow (3 + 3 + 2, 3 + 3 or 3 sweeps between two exchanges where the loop applies three per launch) and
 * every fluid_slab_overlap, one fluid_slab_pressure_step after an untimed one of its own; the times are reduced
 * with MAX over the ranks, so every rank comes to the same choice and sets FLUID_SLAB_OPT_HALO_DEPTH /
 * _OVERLAP accordingly.  Collective; overwrites PRESSURES_1 / _2 (call it before the run, or between steps:
 * 12a / 12b clear them anyway).  Whether hiding an exchange is worth two more launches per pass, and whether
 * fewer, larger messages beat less recomputation of ghost planes, depends on the link: in a one-GPU rehearsal,
 * where an exchange is a device copy, three planes in line win; the defaults (8 planes, both passes split) are
 * what a wire with tens of microseconds per exchange wants.  times_us (optional): 9 entries, [3 * i + overlap]
 * for the i-th depth of {8, 6, 3}, 0 where not measured.  bench.py --gpus N calls this before its warm-up. */
typedef struct fluid_slab_tune_result {
    uint32_t halo_depth, overlap;  /* adopted */
    uint32_t times_us[9];
} fluid_slab_tune_result;
int fluid_slab_tune_exchange(fluid_slab* s, fluid_slab_tune_result* result /* may be NULL */);

typedef enum fluid_slab_stat {
    FLUID_SLAB_STAT_EXCHANGES = 0,        /* plane exchanges issued                                      */
    FLUID_SLAB_STAT_OVERLAPPED = 1,       /* ... of which started beside a split pass                     */
    FLUID_SLAB_STAT_MIGRATED = 2,         /* particles this rank handed to a neighbour                    */
    FLUID_SLAB_STAT_SAMPLER_RERUNS = 3,   /* steps whose 07 pass was redone with more ghost planes        */
    FLUID_SLAB_STAT_SAMPLER_WIDE = 4,     /* ... of which needed the wide source (beyond the image's)     */
    FLUID_SLAB_STAT_EFFECTIVE_HALO = 5,   /* planes per exchange the Jacobi loop actually uses            */
    FLUID_SLAB_STAT_SAMPLER_HALO = 6,     /* ghost planes of VELOCITIES_1 currently exchanged for 07      */
    FLUID_SLAB_STAT_MIGRATE_ROUNDS = 7,   /* hand-over rounds that moved particles                        */
    FLUID_SLAB_STAT_RCCL_RANKS = 8,       /* ncclCommCount of the attached communicator (0: no RCCL)      */
    FLUID_SLAB_STAT_DRY_FACE_SKIPS = 9,   /* loop exchanges left out at a face with no water near it      */
    FLUID_SLAB_STAT_COUNT
} fluid_slab_stat;
int fluid_slab_get_stat(fluid_slab* s, int stat, uint64_t* value);

#ifdef __cplusplus
}
#endif
#endif /* FLUID_SLAB_H */
