/*
 * fluid_engine.h — C ABI of the MI355X-native 3D Eulerian fluid-step engine.
 *
 * This is the drop-in boundary for ONE path of Matezzzz/vulkan-3d-fluid-simulation: the
 * solver sections 00…14 (reference `shaders_fluid/00_init_particles … 14_particles`) as they are
 * listed, ordered and wired in the reference's section lists:
 *
 *   SimulationInitializationSections   fluid_flow_sections.h:136-156   -> fluid_run_init()
 *   SimulationStepSections (01a … 14)  fluid_flow_sections.h:159-338   -> fluid_run_step()
 *     … and its tail 15 … 18 (:339-388, surface-prep passes) on contexts created with surface_prep
 *   one Flow*Section of those lists                                    -> fluid_run_section()
 *   FlowLoopPushConstantSection<…>(N, "12_solve_pressure")  :300-313   -> fluid_run_section_loop()
 *   ImageAttachments / BufferAttachments enums              :10-16     -> fluid_image_id / fluid_buffer_id
 *   SimulationParametersBufferData (264-byte std140 blob)   simulation_constants.h:153-174,
 *       byte offsets shaders_fluid/fluids_uniform_buffer_layout.txt:4-56  -> fluid_params
 *
 * The reference has no FFI of its own: its "operator API" is the C++ section list driven by
 * `complete()` once and `run(CommandBuffer&, FlowDescriptorContext&)` per frame (main.cpp:103-105,
 * 111, 172).  Everything below is plain C: opaque context, plain pointers and sizes, integer status
 * codes, no C++/torch types.  The C++ mirror of the reference classes (FlowSectionList & co.) sits on
 * top of this header in fluid_flow_sections_amd.hpp; the Python binding (ctypes) in
 * vulkan-3d-fluid-simulation_amd/engine.py.
 *
 * Conventions
 *   - Grid W×H×D = params.fluid_size, cell index i=(x,y,z), linear x-fastest: idx = x + W*(y + H*z).
 *   - Host-side layouts at upload/download are the reference image formats, densely packed:
 *       VELOCITIES_1/2   RGBA32F  16 B/cell  (A unused; fluid_flow_sections.h:36-38)
 *       CELL_TYPES / NEW_CELL_TYPES  R8_UINT  1 B/cell (:40-42)
 *       PRESSURES_1/2, DIVERGENCES   R32F   4 B/cell (:44-47)
 *       PARTICLE_DENSITIES_IMG       R32_UINT 4 B/cell (:49)
 *       PARTICLES_BUF    vec4[particle_capacity], xyz + w = active flag (:72)
 *   - Out-of-bounds image loads return 0, out-of-bounds stores/atomics are dropped (Vulkan
 *     robust-image semantics the shaders rely on).
 *   - run_* calls enqueue on the context's in-order HIP stream and return; stream order replaces the
 *     reference's image barriers.  fluid_sync() is the fence.  ONE exception by default: a pressure loop
 *     of >= 16 iterations on a whole-grid context (hence fluid_run_step) waits once, before its first
 *     sweep, for 28 bytes that say where the water is, so that the sweeps are launched over that box only
 *     (the work enqueued before it — at most the step's sections 01a..11 — has executed when the call
 *     returns).  FLUID_OPT_LAUNCH_BOX = 1 removes the wait: full-grid launches, every call asynchronous.
 *   - Every call returns 0 on success or a negative fluid_status; fluid_last_error() gives text.
 *   - One host thread per context.
 */
#ifndef FLUID_ENGINE_H
#define FLUID_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FLUID_ENGINE_ABI_VERSION 1

/* ---- status codes -------------------------------------------------------------------------- */
typedef enum fluid_status {
    FLUID_OK = 0,
    FLUID_ERR_INVALID_ARG = -1,   /* null pointer, unknown id, bad enum                          */
    FLUID_ERR_SIZE_MISMATCH = -2, /* host byte count does not match the attachment's size        */
    FLUID_ERR_HIP = -3,           /* a HIP runtime call failed; text in fluid_last_error()       */
    FLUID_ERR_NO_DEVICE = -4,     /* no usable gfx950 device                                     */
    FLUID_ERR_UNSUPPORTED = -5,   /* valid request the engine does not implement (e.g. section   */
                                  /* not meaningful on a Z-slab context)                         */
    FLUID_ERR_OUT_OF_MEMORY = -6
} fluid_status;

/* ---- attachments: values are the reference enum values (fluid_flow_sections.h:10-16) -------- */
typedef enum fluid_image_id {
    FLUID_IMG_VELOCITIES_1 = 0,
    FLUID_IMG_VELOCITIES_2 = 1,
    FLUID_IMG_CELL_TYPES = 2,
    FLUID_IMG_NEW_CELL_TYPES = 3,
    FLUID_IMG_PRESSURES_1 = 4,
    FLUID_IMG_PRESSURES_2 = 5,
    FLUID_IMG_DIVERGENCES = 6,
    FLUID_IMG_PARTICLE_DENSITIES_IMG = 7,
    /* 8..11: the surface-prep images on the detailed grid (fluid_size * detailed_resolution per axis,
       R32UI / R32UI / R32F / R32F).  Contexts created with
       fluid_create_info.surface_prep only, FLUID_ERR_UNSUPPORTED otherwise. */
    FLUID_IMG_DETAILED_DENSITIES_IMG = 8,
    FLUID_IMG_DETAILED_DENSITIES_INERTIA_IMG = 9,
    FLUID_IMG_PARTICLE_DENSITIES_FLOAT_1 = 10,
    FLUID_IMG_PARTICLE_DENSITIES_FLOAT_2 = 11,
    FLUID_IMAGE_COUNT = 12
} fluid_image_id;

typedef enum fluid_buffer_id {
    FLUID_BUF_PARTICLES_BUF = 0,
    FLUID_BUF_MARCHING_CUBES_COUNTS_BUF = 1, /* uint[256]: triangles per corner configuration; surface_prep  */
    FLUID_BUF_MARCHING_CUBES_EDGES_BUF = 2,  /* uint[256 * 15]: their edge indices (marching_cubes.h:24-33)    */
    FLUID_BUF_SIMULATION_PARAMS_BUF = 3,
    FLUID_BUFFER_COUNT = 4
} fluid_buffer_id;

/* ---- cell types (simulation_constants.h:144-146); the kernels read the values from params --- */
enum { FLUID_CELL_INACTIVE = 0, FLUID_CELL_AIR = 1, FLUID_CELL_WATER = 2, FLUID_CELL_SOLID = 3 };

/* ---- sections: one id per entry of the reference's section lists, named after the shader dir -- */
typedef enum fluid_section_id {
    /* SimulationInitializationSections, fluid_flow_sections.h:139-154 */
    FLUID_SEC_INIT_CLEAR_VELOCITIES_1 = 0, /* :140 FlowClearColorSection(VELOCITIES_1, 0)          */
    FLUID_SEC_INIT_CLEAR_CELL_TYPES = 1,   /* :141 FlowClearColorSection(CELL_TYPES, INACTIVE)     */
    FLUID_SEC_00_INIT_PARTICLES = 2,       /* :143-153, init_particles.comp                        */
    /* SimulationStepSections, fluid_flow_sections.h:163-338 */
    FLUID_SEC_01A_CLEAR_PARTICLE_DENSITIES = 3,        /* :163                                     */
    FLUID_SEC_01_UPDATE_DENSITIES = 4,                 /* :164-175 update_densities.comp           */
    FLUID_SEC_02_UPDATE_WATER = 5,                     /* :176-187 update_water.comp               */
    FLUID_SEC_03_UPDATE_AIR = 6,                       /* :188-198 update_active.comp              */
    FLUID_SEC_04_COMPUTE_EXTRAPOLATED_VELOCITIES = 7,  /* :199-211 extrapolated_velocities.comp    */
    FLUID_SEC_05_SET_EXTRAPOLATED_VELOCITIES = 8,      /* :212-225 extrapolate_velocities.comp     */
    FLUID_SEC_06_UPDATE_CELL_TYPES = 9,                /* :226-236 update_cell_types.comp          */
    FLUID_SEC_07_ADVECT = 10,                          /* :237-249 advect.comp                     */
    FLUID_SEC_08_FORCES = 11,                          /* :250-261 forces.comp                     */
    FLUID_SEC_09_DIFFUSE = 12,                         /* :262-274 diffuse.comp                    */
    FLUID_SEC_10_SOLIDS = 13,                          /* :275-286 solids.comp                     */
    FLUID_SEC_11_COMPUTE_DIVERGENCE = 14,              /* :287-297 compute_divergence.comp         */
    FLUID_SEC_12A_CLEAR_PRESSURES_1 = 15,              /* :298                                     */
    FLUID_SEC_12B_CLEAR_PRESSURES_2 = 16,              /* :299                                     */
    FLUID_SEC_12_SOLVE_PRESSURE = 17,                  /* :300-313 pressure.comp (loop section)    */
    FLUID_SEC_13_FIX_DIVERGENCE = 18,                  /* :314-326 fix_divergence.comp             */
    FLUID_SEC_14_PARTICLES = 19,                       /* :327-338 particles.comp                  */
    /* surface-prep passes on the detailed grid (fluid_flow_sections.h:339-388); contexts created with
     * fluid_create_info.surface_prep only */
    FLUID_SEC_14A_CLEAR_DETAILED_DENSITIES = 20,          /* :339                                     */
    FLUID_SEC_15_UPDATE_DETAILED_DENSITIES = 21,          /* :340-351 update_detailed_densities.comp  */
    FLUID_SEC_16_COMPUTE_DETAILED_DENSITIES_INERTIA = 22, /* :352-363 densities_inertia.comp          */
    FLUID_SEC_17_COMPUTE_FLOAT_DENSITIES = 23,            /* :364-375 float_densities.comp            */
    FLUID_SEC_18_DIFFUSE_FLOAT_DENSITIES = 24,            /* :376-388 diffuse_densities.comp (loop)   */
    FLUID_SEC_INIT_CLEAR_DETAILED_DENSITIES_INERTIA = 25, /* :142 (SimulationInitializationSections)  */
    FLUID_SECTION_COUNT = 26
} fluid_section_id;

/* ---- parameters: byte-for-byte the reference's 264-byte std140 uniform block ----------------
 * Offsets: shaders_fluid/fluids_uniform_buffer_layout.txt:4-56; write order
 * simulation_constants.h:156-172.  Fields at 116-232 and 260 are render/surface-only and are
 * carried but never read by this engine. */
typedef struct fluid_params {
    uint32_t fluid_size[3];                     /*   0 */
    uint32_t fluid_volume;                      /*  12 */
    uint32_t cell_type_inactive;                /*  16 */
    uint32_t cell_type_air;                     /*  20 */
    uint32_t cell_type_water;                   /*  24 */
    uint32_t cell_type_solid;                   /*  28 */
    float time_delta;                           /*  32 */
    float pressure_air;                         /*  36 */
    float cell_width;                           /*  40 */
    float fluid_density;                        /*  44 */
    uint32_t particle_compute_size[2];          /*  48 */
    uint32_t _pad56[2];                         /*  56 */
    uint32_t particle_spawn_cube_resolution[3]; /*  64 */
    uint32_t particle_spawn_cube_volume;        /*  76 */
    float particle_spawn_cube_offset[3];        /*  80 */
    uint32_t _pad92;                            /*  92 */
    float particle_spawn_cube_size[3];          /*  96 */
    float gravity;                              /* 108 */
    float diffuse_k;                            /* 112 */
    int32_t detailed_resolution;                /* 116 */
    int32_t detailed_resolution_volume;         /* 120 */
    int32_t max_inertia;                        /* 124 */
    int32_t inertia_increase_filled;            /* 128 */
    int32_t required_neighbour_hits;            /* 132 */
    int32_t inertia_increase_neighbour;         /* 136 */
    int32_t inertia_decrease;                   /* 140 */
    float dens_division_coefficient;            /* 144 */
    float dens_diffuse_k;                       /* 148 */
    uint32_t _pad152[2];                        /* 152 */
    float particle_color[3];                    /* 160 */
    float particle_base_size;                   /* 172 */
    float light_dir[3];                         /* 176 */
    uint32_t _pad188;                           /* 188 */
    float ambient_color[3];                     /* 192 */
    uint32_t _pad204;                           /* 204 */
    float diffuse_color[3];                     /* 208 */
    uint32_t _pad220;                           /* 220 */
    uint32_t fluid_surface_render_size[3];      /* 224 */
    float active_particle_w;                    /* 236 */
    uint32_t fountain_position[3];              /* 240 */
    float fountain_force;                       /* 252 */
    float solid_repel_velocity;                 /* 256 */
    float particle_max_size;                    /* 260 */
} fluid_params;                                 /* 264 */

#define FLUID_PARAMS_BYTES 264

/* Fill `p` with the reference defaults (simulation_constants.h:7-139) for a grid of the given size:
 * everything that the reference derives from fluid_size (fluid_volume, fountain_position :85,
 * surface sizes) is recomputed; the particle spawn cube keeps the reference's 100^3 / (5,2,1.5) /
 * (10,10,2) (:48-50) and particle_compute_size = (particle_capacity, 1) (:33,:162). Pure host code. */
int fluid_params_default(fluid_params* p, uint32_t width, uint32_t height, uint32_t depth,
                         uint32_t particle_capacity);

/* ---- 09_diffuse behaviour ---------------------------------------------------- */
typedef enum fluid_diffuse_mode {
    FLUID_DIFFUSE_REFERENCE_EXACT = 0, /* diffuse.comp as written: the diffused value is shadowed,   */
                                       /* VELOCITIES_1 = (VELOCITIES_2.xyz, 0)  (diffuse.comp:34-46) */
    FLUID_DIFFUSE_INTENDED = 1         /* the 7-point explicit diffusion the shader means (:38-43)    */
} fluid_diffuse_mode;

/* ---- context -------------------------------------------------------------------------------- */
typedef struct fluid_ctx fluid_ctx;

typedef struct fluid_create_info {
    uint32_t struct_bytes;       /* = sizeof(fluid_create_info), for ABI growth                      */
    int32_t device;              /* HIP device ordinal; -1 = current device                         */
    const void* params_blob;     /* 264 bytes, layout of fluid_params; fluid_size = GLOBAL grid      */
    uint64_t particle_capacity;  /* PARTICLE_BUFFER_SIZE of the shaders (simulation_constants.h:29); */
                                 /* 0 = particle_compute_size.x*.y from the blob                    */
    uint32_t pressure_iterations;/* divergence_solve_iterations (simulation_constants.h:74) used by  */
                                 /* fluid_run_step(); 0 = 200                                       */
    /* Z-slab decomposition (one context per GPU). z_count = 0 means the whole grid.              */
    uint32_t slab_z_begin;       /* first owned global z plane                                      */
    uint32_t slab_z_count;       /* owned planes; the context also holds one ghost plane per side    */
    void* hip_stream;            /* hipStream_t to enqueue on (borrowed); NULL = engine creates one  */
    void* arena;                 /* optional caller-owned device memory for all attachments          */
    uint64_t arena_bytes;        /*   (>= fluid_required_arena_bytes); NULL = engine hipMallocs      */
    /* ---- fields added after ABI v1's first layout: read only when struct_bytes covers them ------- */
    uint32_t surface_prep;       /* 1 = also allocate the four detailed-grid images (ids 8-11: 16 B x */
                                 /* detailed_resolution^3 per simulation cell) and run sections      */
                                 /* 14a-18 at the end of fluid_run_step; whole-grid contexts only     */
    uint32_t surface_diffuse_steps; /* float_density_diffuse_steps (simulation_constants.h:127) used   */
                                 /* by fluid_run_step(); 0 = 4                                       */
} fluid_create_info;
/* sizeof(fluid_create_info) before surface_prep was added: still accepted as struct_bytes */
#define FLUID_CREATE_INFO_V1_BYTES 64

/* Device bytes a context with this geometry needs (pure host arithmetic). */
uint64_t fluid_required_arena_bytes(const fluid_create_info* info);

/* Replaces: SimulationDescriptors ctor (fluid_flow_sections.h:26-96) + params upload (:86). */
int fluid_create(fluid_ctx** out, const fluid_create_info* info);
void fluid_destroy(fluid_ctx* ctx);

/* Text of the last error on this context (or of the last failed fluid_create when ctx == NULL). */
const char* fluid_last_error(const fluid_ctx* ctx);

int fluid_abi_version(void);

/* ---- data movement (the reference never reads back; these exist for tests, checkpoints and the
 *      caller that replaces the renderer).  Synchronous with respect to the context's stream.   */
int fluid_upload_image(fluid_ctx* ctx, int image_id, const void* host, uint64_t bytes);
int fluid_download_image(fluid_ctx* ctx, int image_id, void* host, uint64_t bytes);
int fluid_upload_buffer(fluid_ctx* ctx, int buffer_id, const void* host, uint64_t bytes);
int fluid_download_buffer(fluid_ctx* ctx, int buffer_id, void* host, uint64_t bytes);
/* Bytes of the owned (non-ghost) part of an attachment as seen by upload/download. */
int fluid_image_bytes(const fluid_ctx* ctx, int image_id, uint64_t* bytes);
int fluid_buffer_bytes(const fluid_ctx* ctx, int buffer_id, uint64_t* bytes);

/* Replace the params blob (re-reads every hot-path field; fluid_size must not change). */
int fluid_set_params(fluid_ctx* ctx, const void* params_blob);
int fluid_set_pressure_iterations(fluid_ctx* ctx, uint32_t iterations);
int fluid_set_diffuse_mode(fluid_ctx* ctx, int mode);

/* ---- section dispatch ----------------------------------------------------------------------- */
/* One entry of a section list.  For FLUID_SEC_12_SOLVE_PRESSURE this is ONE dispatch with the push
 * constant taken from the context's loop counter (first dispatch after 12A/12B: is_even_iteration
 * = 1, then alternating) — use fluid_run_section_loop or fluid_run_pressure_dispatch for explicit
 * control. */
int fluid_run_section(fluid_ctx* ctx, int section_id);

/* FlowLoopPushConstantSection<FlowComputePushConstantSection>(iterations, …) of
 * fluid_flow_sections.h:300-313: `iterations` dispatches, dispatch k has is_even_iteration =
 * (k % 2 == 0), i.e. reads PRESSURES_1 / writes PRESSURES_2 on even k (pressure.comp:28-31,71-75;
 * SURVEY.md F2).  Loop sections: FLUID_SEC_12_SOLVE_PRESSURE and, on surface_prep contexts,
 * FLUID_SEC_18_DIFFUSE_FLOAT_DENSITIES (:376-388, float_density_diffuse_steps dispatches). */
int fluid_run_section_loop(fluid_ctx* ctx, int section_id, uint32_t iterations);

/* `count` consecutive entries of SimulationStepSections, starting at `first_section_id`, as one unit
 * (a FlowSectionList slice, fluid_flow_sections.h:164-338).  Every image holds afterwards what the
 * sections run one by one would leave, EXCEPT an image that only carries data from one section of
 * the group to the next and that a later section of the step overwrites completely — the engine may
 * skip storing it (named per group below).  Groups with a grouped implementation, whole-grid
 * contexts with fluid_size.x % 4 == 0:
 *   (FLUID_SEC_04_COMPUTE_EXTRAPOLATED_VELOCITIES, 2)   VELOCITIES_2 is left unspecified (07 rewrites it)
 *   (FLUID_SEC_07_ADVECT, 2)                            exact
 *   (FLUID_SEC_09_DIFFUSE, 3)                           exact (FLUID_DIFFUSE_REFERENCE_EXACT)
 * Any other slice, or a grouped one whose precondition does not hold, runs section by section.
 * fluid_run_step uses these groups unless FLUID_OPT_STEP_FUSION = 1.
 * Z-slab contexts: only these three slices (other slices need ghost planes from the neighbours in
 * between).  Ghost planes they read: (04, 2) as 04 and 05; (07, 2) as 07; (09, 3) one plane of
 * VELOCITIES_2 and of CELL_TYPES above the slab — exchange VELOCITIES_2 before the group instead of
 * VELOCITIES_1 between 10 and 11. */
int fluid_run_section_group(fluid_ctx* ctx, int first_section_id, uint32_t count);

/* FlowClearColorSection(ctx, image, ClearValue) for an arbitrary image and value
 * (fluid_flow_sections.h:140-142,163,298-299 are the uses on this path).  `value_bits` holds the
 * texel as 32-bit patterns: 4 words for RGBA32F, 1 for R32F / R32_UINT, the low byte of word 0 for
 * R8_UINT.  Clearing PRESSURES_1/2 also resets the 12_solve_pressure loop counter. */
int fluid_clear_image(fluid_ctx* ctx, int image_id, const uint32_t value_bits[4]);

/* One 12_solve_pressure dispatch with an explicit push constant (pressure.comp:29-31). */
int fluid_run_pressure_dispatch(fluid_ctx* ctx, uint32_t is_even_iteration);

/* One 18_diffuse_float_densities dispatch with an explicit push constant (diffuse_densities.comp:38-40):
 * 1 reads PARTICLE_DENSITIES_FLOAT_1 and writes _2, 0 the reverse.  surface_prep contexts. */
int fluid_run_surface_diffuse_dispatch(fluid_ctx* ctx, uint32_t is_even_iteration);

/* SimulationInitializationSections::run (fluid_flow_sections.h:139-154; main.cpp:111). */
int fluid_run_init(fluid_ctx* ctx);
/* SimulationStepSections::run (main.cpp:172): 01a…14 (fluid_flow_sections.h:163-338) and, on a
 * surface_prep context, the rest of the list up to the renderer: the detailed-density clear and 15…18
 * (:339-388).  The init list of such a context also clears DETAILED_DENSITIES_INERTIA_IMG (:142). */
int fluid_run_step(fluid_ctx* ctx);

/* Fence: wait until everything enqueued so far has executed (replaces fence wait main.cpp:124). */
int fluid_sync(fluid_ctx* ctx);

/* ---- timing and tracing --------------------------------------------------------------------- */
/* The name the reference's section lists give a section ("07_advect", fluid_flow_sections.h:139-388), or NULL.
 * With FLUID_ROCTX=1 in the environment every section runs inside a roctx range of that name (libroctx64,
 * loaded on first use): rocprofv3 --marker-trace --kernel-trace then attributes kernels to list entries. */
const char* fluid_section_name(int section_id);

/* When enabled, every run_section / loop records HIP events on the context's stream.           */
int fluid_enable_timing(fluid_ctx* ctx, int enabled);
/* Accumulated device milliseconds and launch count of a section since the last reset.          */
int fluid_section_time_ms(fluid_ctx* ctx, int section_id, double* total_ms, uint64_t* calls);
int fluid_reset_timing(fluid_ctx* ctx);

/* ---- multi-GPU plumbing (Z-slab contexts) ------------------------------------------------------
 * Device address and byte count of one XY plane of an image, for halo exchange by the caller's
 * communicator (RCCL Send/Recv).  `plane` is a LOCAL z index: 0 … z_count-1 = owned planes,
 * -FLUID_IMAGE_GHOST_PLANES … -1 and z_count … z_count+FLUID_IMAGE_GHOST_PLANES-1 = ghost planes.
 * Consecutive planes are contiguous in memory.                                                   */
int fluid_image_plane_ptr(fluid_ctx* ctx, int image_id, int32_t plane, void** device_ptr,
                          uint64_t* bytes);
/* The loop section in explicit form, for callers that must act between launches (halo exchange):
 *   begin   : build the solver's working data from CELL_TYPES / DIVERGENCES / PRESSURES_1
 *             (iterate 0 -> working buffer 0)
 *   advance : 1 sweep, or 2 or 3 sweeps in one pass over HBM, up to fluid_pressure_loop_max_sweeps()
 *             (3 on grids up to 512 cells wide, 2 on wider ones, 1 with FLUID_OPT_JACOBI_FUSE = 1);
 *             `keep_intermediate` = also keep the iterate before the newest one (needed by the last
 *             launch of a loop: PRESSURES_1 / _2 end with the last even / odd iterate);
 *             *written_buffer = working buffer (0..2) now holding the newest iterate
 *   end     : write the last two iterates into the water cells of PRESSURES_1 (even) / PRESSURES_2 (odd)
 * begin; advance...; end equals fluid_run_section_loop(FLUID_SEC_12_SOLVE_PRESSURE, N) for the same
 * number of sweeps.
 * On a Z-slab context every sweep consumes one ghost plane per side of the newest iterate.  The
 * caller exchanges boundary planes with its Z-neighbours and reports it:
 *   after begin: h-1 planes of the mask (buffer 3) and of b_i (buffer 4) and h planes of working
 *   buffer 0, then fluid_pressure_loop_halo_exchanged(ctx, h, h-1); 2 <= h <= FLUID_LOOP_MAX_HALO,
 *   h <= slab_z_count.  The loop then advances h sweeps without communication (the engine computes
 *   the shrinking ghost region redundantly), after which the caller exchanges h planes of the buffer
 *   holding the newest iterate (*written_buffer) and calls halo_exchanged(ctx, h, 0) again.
 *   A one-sweep advance leaves no valid ghost planes.
 * Planes for the exchange: fluid_pressure_loop_plane_ptr (local plane numbers, -h..-1 and Dl..Dl+h-1
 * are ghost planes; consecutive planes are contiguous in memory).
 * Needs fluid_size.x % 4 == 0, else FLUID_ERR_UNSUPPORTED (then loop over
 * fluid_run_pressure_dispatch and exchange one plane of the written image per sweep instead). */
#define FLUID_LOOP_MAX_HALO 8
int fluid_pressure_loop_begin(fluid_ctx* ctx);
/* 1 if the explicit loop can run on this context (fluid_size.x % 4 == 0 and the pressure kernel option
 * allows the working-buffer path), 0 if the caller has to loop over fluid_run_pressure_dispatch. */
int fluid_pressure_loop_available(fluid_ctx* ctx);
int fluid_pressure_loop_max_sweeps(fluid_ctx* ctx);
int fluid_pressure_loop_advance(fluid_ctx* ctx, uint32_t sweeps, int keep_intermediate,
                                int* written_buffer);
int fluid_pressure_loop_halo_exchanged(fluid_ctx* ctx, uint32_t depth, uint32_t aux_depth);
int fluid_pressure_loop_end(fluid_ctx* ctx);
/* A two-sweep pass as two launches, so that a halo exchange can run beside the larger one:
 *   FLUID_LOOP_PART_INTERIOR  writes the output planes [interior_begin, interior_end) (local plane
 *                             numbers, clipped to the planes the pass writes)
 *   FLUID_LOOP_PART_EDGES     writes the rest, below and above, in one launch
 * in either order with the same arguments; the second call completes the pass (state as after
 * fluid_pressure_loop_advance(ctx, 2, keep_intermediate, …)); *written_buffer is the destination
 * buffer after either call.  The schedule this is for, with h ghost planes exchanged every h sweeps:
 *   last pass before an exchange:  EDGES with interior [h, z_count - h) first — the h planes per face
 *       to send are complete when it ends —, start the exchange on another stream, then INTERIOR;
 *   first pass after it:  call fluid_pressure_loop_halo_exchanged when the exchange is STARTED, launch
 *       INTERIOR with [2, z_count - 2) (its inputs are owned planes only), make this context's stream
 *       wait for the receives, then EDGES.
 * At a domain face there is no edge: pass the plane range's natural end there (e.g. INT32_MIN /
 * INT32_MAX; the engine clips). */
/* FLUID_OPT_EDGE_STREAM = 1: the EDGES launch goes to a second, high-priority stream of the context
 * (fluid_pressure_loop_edge_stream) and runs BESIDE the INTERIOR launch; the engine orders both after the
 * work that preceded the pass and lets the main stream wait for the edges when the pass completes.  The
 * caller then orders its exchange against THAT stream: start it after an event recorded on the edge stream
 * once EDGES is enqueued (pass before the exchange), and make the edge stream wait for the receives before
 * enqueueing EDGES (pass after it). */
#define FLUID_LOOP_PART_EDGES 1
#define FLUID_LOOP_PART_INTERIOR 2
/* ..._part_n: the same for a pass of `sweeps` (2 or 3) sweeps — the interior of the first pass after an
 * exchange is then [sweeps, z_count - sweeps); ..._part is ..._part_n with two. */
int fluid_pressure_loop_advance_part_n(fluid_ctx* ctx, uint32_t sweeps, int keep_intermediate, int part,
                                       int32_t interior_begin, int32_t interior_end, int* written_buffer);
int fluid_pressure_loop_advance_part(fluid_ctx* ctx, int keep_intermediate, int part,
                                     int32_t interior_begin, int32_t interior_end,
                                     int* written_buffer);
int fluid_pressure_loop_edge_stream(fluid_ctx* ctx, void** hip_stream);
int fluid_pressure_loop_plane_ptr(fluid_ctx* ctx, int which, int32_t plane, void** device_ptr,
                                  uint64_t* bytes);

/* Tell the engine that the caller wrote device memory of `image_id` through a pointer obtained
 * from fluid_image_plane_ptr (halo exchange), so data the engine derives from it is rebuilt.
 * Ghost planes of PRESSURES_1/2 must carry the neighbouring slab's cells of the same buffer. */
int fluid_notify_image_written(fluid_ctx* ctx, int image_id);
/* The same when only ghost planes were written (a halo exchange): what the engine knows about the owned
 * planes stays valid. */
int fluid_notify_ghost_planes_written(fluid_ctx* ctx, int image_id);
/* Z-slab contexts, full step.  The caller runs the sections one by one (fluid_run_section) and
 * exchanges ghost planes between them (fluid_image_plane_ptr, FLUID_IMAGE_GHOST_PLANES per side):
 *   after 02 and after 03: NEW_CELL_TYPES, 1 plane          (03 / 05 read z-1, z+1)
 *   after 05: VELOCITIES_1, FLUID_IMAGE_GHOST_PLANES planes (07 samples it; 04 of the next step reads z+-1)
 *   after 10: VELOCITIES_1, 1 plane                         (11 reads z+1)
 *   after the pressure loop: PRESSURES_2, 1 plane           (13 reads z-1)
 *   after 13: VELOCITIES_1, FLUID_IMAGE_GHOST_PLANES planes (14 samples it)
 *   after 14: particle migration (below)
 * In FLUID_DIFFUSE_INTENDED mode 09 is a 7-point stencil: VELOCITIES_2, 1 plane, after 08 (and the
 * sections 09, 10, 11 one by one: the grouped 09+10+11 exists for the reference-exact copy only).
 * With the grouped passes (fluid_run_section_group): 04+05 in place of 04, 05; 07+08 in place of 07,
 * 08; and, instead of the one-plane VELOCITIES_1 exchange after 10, one plane of VELOCITIES_2 after 08
 * followed by 09+10+11.
 * 06 copies one ghost plane per side along with the owned planes.  The velocity sampler of 07 can
 * reach as far as the fluid moves in one step; a tap beyond the current ghost planes raises the halo
 * violation flag (fluid_slab_status reads and clears it) and the pass is redone with more of them
 * (fluid_sampler_* below).  include/fluid_slab.h drives all of this.
 *
 * Particles: global particle i (slot i of the API's particle array) lives on the rank whose slab contains
 * the plane the particle counts towards (01_update_densities).  A rank STORES only what it owns — the
 * particles and their slots side by side, outside the arena: 20 bytes per owned particle plus headroom,
 * growing when an adoption would not fit —; fluid_upload_buffer(PARTICLES_BUF) takes the global array and
 * keeps the owned slots, fluid_download_buffer returns the global array with a tombstone (w = bit pattern
 * 0x7FC0DEAD) in every slot this rank does not hold.  After 14 the particles that left the slab are handed
 * to the Z-neighbours (fluid_particles_* below). */
#define FLUID_IMAGE_GHOST_PLANES 4
int fluid_slab_status(fluid_ctx* ctx, uint32_t* halo_violation);

/* Particle hand-over between Z-neighbours.  Four lists of 32-byte entries {float4 data; uint32 index;
 * 3 x pad} in device memory: FLUID_MIGRATE_SEND_DOWN / _UP hold what this slab passes to the neighbour
 * below (towards z = 0) / above, FLUID_MIGRATE_FROM_BELOW / _ABOVE are where the caller's Recv puts what
 * those neighbours pass.  One round:
 *   collect(reset, counts, left)   lists the particles this slab holds but does not own (appending to the
 *       send lists, after clearing them if `reset`) and buries their slots; counts[] = entries now in the
 *       down / up list; *left = particles that found their list full and stayed — another round is needed
 *   the caller sends counts[d] entries of list d to that neighbour and receives the neighbours' entries
 *   adopt_received(nb, na, fwd)    clears the send lists, adopts the received entries this slab owns and
 *       appends the others to the send list of the direction they were travelling in (a particle that
 *       crossed more than one slab in a step); fwd[] = entries now in the down / up list.
 * The hand-over is complete when, over all ranks, nothing was sent and nothing is left.  Every call
 * synchronises the stream (the counts are host values). */
enum { FLUID_MIGRATE_SEND_DOWN = 0, FLUID_MIGRATE_SEND_UP = 1, FLUID_MIGRATE_FROM_BELOW = 2,
       FLUID_MIGRATE_FROM_ABOVE = 3 };
int fluid_particles_migrate_list(fluid_ctx* ctx, int which, void** device_list, uint32_t* capacity_entries);
int fluid_particles_collect(fluid_ctx* ctx, int reset_lists, uint32_t counts[2], uint32_t* left_behind);
int fluid_particles_adopt_received(fluid_ctx* ctx, uint32_t from_below, uint32_t from_above,
                                   uint32_t forwarded[2]);

/* A step driven section by section from outside (what a Z-slab context needs: ghost planes travel between
 * the sections; include/fluid_slab.h does it), with the skipping fluid_run_step does on a whole-grid context:
 *   fluid_step_begin            the calls up to fluid_step_end run the step list 01a ... 14 once, in order
 *                               (fluid_run_section, fluid_run_section_group, the explicit pressure loop);
 *                               bricks far from the water are skipped as in fluid_run_step, and ghost-plane
 *                               exchanges do not count as writes from outside.  section_list = 1: the
 *                               caller runs one section per call (no grouped passes): nothing is skipped
 *   fluid_step_build_activity   after 06: the activity bricks of the new CELL_TYPES (one byte per 256 x 4 x 16
 *                               cells).  A brick is skipped when neither it nor any of its 26 neighbours has
 *                               held water for three steps — across a slab face the neighbours are the other
 *                               slab's edge layer: exchange fluid_activity_layer_ptr(0 / 1) = this context's
 *                               bottom / top layer with (2 / 3) = the layer received from below / above
 *                               (bytes = 0: this step does not skip, nothing to exchange).  Without the
 *                               exchange the bricks at a shared face are never skipped.
 *   fluid_step_status           after 07 (+ 08): words[0] = the halo-violation flag (read and cleared, as
 *                               fluid_slab_status), words[1] = 1 if the box of this context's water is known,
 *                               then [2] = bricks with water, rows [3], [4)), cells [5], [6)) along x,
 *                               [7] = the owned planes with water, first | (end << 16), in brick layers of 16.
 *                               Synchronises the stream once (the reduction the sampler protocol needs anyway).
 *   fluid_step_set_box          the launches of the pressure loop cover only these rows and — where the water
 *                               spans at most two 256-cell columns — this x window: the UNION of this context's
 *                               box with its Z-neighbours' (their water is what the ghost planes hold);
 *                               own_bricks = 0: this context has no water, its launches are skipped. */
int fluid_step_begin(fluid_ctx* ctx, int section_list);
int fluid_step_end(fluid_ctx* ctx);
int fluid_step_build_activity(fluid_ctx* ctx);
int fluid_activity_layer_ptr(fluid_ctx* ctx, int which, void** device_ptr, uint64_t* bytes);
int fluid_step_status(fluid_ctx* ctx, uint32_t words[8]);
int fluid_step_set_box(fluid_ctx* ctx, int valid, uint32_t own_bricks, uint32_t y_lo, uint32_t y_hi,
                       uint32_t x_lo, uint32_t x_hi);

/* The velocity sampler of 07_advect on a Z slab (particles and back-traces are never
 * clamped, advect.comp:63-78).  A back-trace reaches floor(|v.z| * dt) + 1 planes from its cell, so how
 * many ghost planes of VELOCITIES_1 07 needs depends on the flow.  The protocol the slab driver
 * (fluid_slab.h) follows, every step:
 *   1. exchange `n` ghost planes of VELOCITIES_1 after 05 (n = fluid_set_sampler_halo, 1..
 *      FLUID_IMAGE_GHOST_PLANES; the kernels treat planes beyond n as unknown) and run 07 (or 07+08);
 *   2. read fluid_slab_status and combine it over all ranks with MAX.  0: done.
 *   3. otherwise the pass is redone (07 reads VELOCITIES_1 and CELL_TYPES, writes VELOCITIES_2: nothing
 *      it read has changed): all-reduce(MAX) fluid_sampler_reach; if it fits the image's ghost planes,
 *      exchange that many and run the section again; if not, fluid_sampler_wide_begin(reach, reach)
 *      allocates a source with that many planes per side (clipped to the grid) holding this slab's
 *      planes, the caller fills the others from the ranks that own them (fluid_sampler_wide_plane_ptr),
 *      and fluid_run_advect_wide runs the pass on it.  Either way the flag is clear afterwards: the step
 *      never fails because the fluid is fast.
 * 14_particles samples at the position of a particle this slab owns: one ghost plane always suffices. */
int fluid_set_sampler_halo(fluid_ctx* ctx, uint32_t planes);
/* Upper bound of the planes a back-trace of 07 can reach beyond its cell, from max |VELOCITIES_1.z| over
 * the owned planes (wave64 shuffle reduction, one atomic per wavefront); fluid_size.z if a velocity is not
 * finite.  Synchronises the stream. */
int fluid_sampler_reach(fluid_ctx* ctx, uint32_t* planes);
int fluid_sampler_wide_begin(fluid_ctx* ctx, uint32_t planes_below, uint32_t planes_above);
/* plane: local index, -below' .. z_count + above' - 1 (the counts clipped to the grid) */
int fluid_sampler_wide_plane_ptr(fluid_ctx* ctx, int32_t plane, void** device_ptr, uint64_t* bytes);
int fluid_run_advect_wide(fluid_ctx* ctx, int with_forces);

/* Geometry of this context. */
int fluid_get_geometry(const fluid_ctx* ctx, uint32_t global_size[3], uint32_t* slab_z_begin,
                       uint32_t* slab_z_count, uint64_t* particle_capacity);

/* ---- opt-in pressure solver — NOT the reference's algorithm or results ----------
 * FLUID_SOLVER_JACOBI (default): 12_solve_pressure as the reference runs it.
 * FLUID_SOLVER_RED_BLACK_SOR: the same linear system (pressure.comp:41-62) relaxed by red-black
 * successive over-relaxation.  One iteration updates in place on PRESSURES_1 first the WATER cells with
 * (x + y + z) even, then the odd ones: gs = -s / aii from the current image, P += omega * (gs - P).
 * fluid_run_section(12) = one iteration; fluid_run_section_loop(12, n) (and fluid_run_step) = n
 * iterations followed by PRESSURES_2 := PRESSURES_1, so that 13_fix_divergence uses the final iterate.
 * For the same ERROR it needs a small fraction of the Jacobi iterations on pool-like scenes (the residual
 * is not the yardstick: over-relaxation keeps it large while the error collapses).
 * Whole-grid contexts only.  Results are bit-identical to the oracle's restatement of this solver. */
typedef enum fluid_solver { FLUID_SOLVER_JACOBI = 0, FLUID_SOLVER_RED_BLACK_SOR = 1 } fluid_solver;
int fluid_set_pressure_solver(fluid_ctx* ctx, int solver, float omega);

/* ---- engine options (performance variants of the same arithmetic; results are bit-identical) -- */
typedef enum fluid_option {
    FLUID_OPT_PRESSURE_KERNEL = 0, /* 0 = auto.  Single dispatches: 1 = one cell per thread, 2/3/4 =  */
                                   /* z-marching with 2/4/1 rows per wavefront.  Loop section: 1-4   */
                                   /* = that kernel once per sweep on the images; 0/5/6/7 = working- */
                                   /* buffer fast path with 1/2/4/1 rows per wavefront               */
    FLUID_OPT_JACOBI_FUSE = 1,     /* loop section: 0 = several sweeps per pass over HBM (default: three */
                                   /* on grids up to 512 cells wide, two on wider ones), 1 = one kernel   */
                                   /* launch per sweep, 2 = at most two sweeps per pass, 3 = as 0        */
    FLUID_OPT_STEP_FUSION = 2,     /* fluid_run_step: 0 = sections 04+05, 07+08 and 09+10+11 run as   */
                                   /* grouped passes (default; every image ends the step with the   */
                                   /* bits the section list leaves, intermediates are not stored),  */
                                   /* 1 = the section list, one kernel per section                  */
    FLUID_OPT_QUIET_BRICKS = 3,    /* fluid_run_step with grouped passes: 0 = 07+08, 09+10+11 and 13     */
                                   /* skip bricks of 256x4x16 cells that have had no water in or   */
                                   /* next to them for three steps — a step changes nothing there  */
                                   /* (default); 1 = process every cell; 2 = as 0, with one workgroup */
                                   /* per brick layer in the skipping passes whatever the grid size     */
                                   /* (the default does that from 4096 bricks; tests)                   */
    FLUID_OPT_ADVECT_KERNEL = 4,   /* 07_advect: 0 = velocity sampler tiled into LDS (default), 1 = taps    */
                                   /* straight from global memory                                    */
    FLUID_OPT_SURFACE_KERNEL = 5,  /* 18_diffuse_float_densities: 0 = z-marching kernels (default): the loop */
                                   /* section applies two dispatches per pass over HBM (a third float image */
                                   /* of the detailed grid is allocated on first use), single dispatches one; */
                                   /* 1 = four cells per thread, one plane per workgroup, one dispatch per     */
                                   /* pass; 100 + R = as 0 with R = 8, 10, 12, 14 or 16 rows per workgroup     */
                                   /* in the two-dispatch kernel (tuning; default 12)                          */
    FLUID_OPT_LAUNCH_BOX = 6,      /* pressure loop on a sparse scene: 0 = launches cover only the box / x  */
                                   /* window that holds the water, which costs ONE stream synchronisation  */
                                   /* per step (the host reads 28 bytes); 1 = full-grid launches, every    */
                                   /* call of fluid_run_step stays asynchronous                            */
    FLUID_OPT_EDGE_STREAM = 7,     /* split passes of the explicit loop API: 0 = both launches on the       */
                                   /* context's stream (default), 1 = EDGES on a second stream (see        */
                                   /* fluid_pressure_loop_advance_part)                                    */
    FLUID_OPT_PARTICLE_SORT = 8,   /* whole-grid contexts: the particles are STORED sorted by bins of 16 x 4 x  */
                                   /* 16 cells (01 becomes an LDS histogram with plain stores, 14 takes its    */
                                   /* taps from an LDS tile); uploads, downloads and 00 keep speaking slot      */
                                   /* order, so nothing is observable but the time.  0 = on from 4 M particle   */
                                   /* slots (default; needs 28 B per slot outside the arena, falls back to slot */
                                   /* order if that cannot be allocated), 1 = off, 2 = on at any size; 3 / 4 =   */
                                   /* test modes (sort before every 01 / sort once and never again).            */
                                   /* Z-slab contexts store the particles they OWN compactly (20 B per entry)   */
                                   /* and sort that storage by the bins of their own planes under the same      */
                                   /* values (0 = from 4 M entries; a second set of arrays while sorting is on); */
                                   /* a sort drops the holes leavers left, adopted particles sit behind the     */
                                   /* bins until the next sort                                                  */
    FLUID_OPT_COUNT
} fluid_option;
int fluid_set_option(fluid_ctx* ctx, int option, int64_t value);

/* Convergence read-out of the pressure solve (not in the reference, which never looks at its
 * residual; synchronises the stream).  For every WATER cell of this context, in the sweep's own fp32
 * arithmetic: s = b_i - sum over non-solid neighbours of (water ? P[nb] : pressure_air)
 * (pressure.comp:54-61), r = s + aii * P[cell] — the sweep stores -s / aii, so r = 0 at its fixed
 * point.  *max_abs = max |r| (exact; NaN residuals are ignored), *sum_squares = sum of r^2 in double
 * (accumulation order unspecified), *water_cells = number of cells.  `image_id` = FLUID_IMG_PRESSURES_1
 * or _2.  On a Z-slab context the figures cover the slab (its ghost planes must be current); combine
 * them across ranks with max / sum.  Reduced per wavefront with wave64 shuffles, one atomic each. */
int fluid_pressure_residual(fluid_ctx* ctx, int image_id, float* max_abs, double* sum_squares,
                            uint64_t* water_cells);

/* Number of fp32 words of an image's owned planes that are inf or NaN (synchronises the stream).  The
 * reference never checks: a WATER cell enclosed by SOLID on all six sides has aii = 0 (pressure.comp:53-62),
 * its pressure becomes inf / NaN and 13_fix_divergence carries that into the velocities.  Float images only
 * (VELOCITIES_1/2 count all four components, PRESSURES_1/2, DIVERGENCES, PARTICLE_DENSITIES_FLOAT_1/2). */
int fluid_count_nonfinite(fluid_ctx* ctx, int image_id, uint64_t* count);

/* Offline visualisation: the triangles the reference's marching-cubes renderer draws from a
 * float density image of the detailed grid (30-32 are render sections and stay with the caller; this is the
 * geometry of 31_render_surface — render_surface.vert:19-25, render_surface.geom:45-103 — as a list).
 * surface_prep contexts, after fluid_upload_buffer of FLUID_BUF_MARCHING_CUBES_COUNTS_BUF / _EDGES_BUF (what
 * MarchingCubesBuffers::loadData reads from surface_render_data/, marching_cubes.h:30-33).  image_id =
 * FLUID_IMG_PARTICLE_DENSITIES_FLOAT_2 is what the reference draws (fluid_flow_sections.h:431).  12 floats per
 * triangle: three vertices in simulation-cell units, then the flat normal (normalize() = v / sqrt(dot(v, v)),
 * IEEE); the order of the list is unspecified.  *count = triangles found; the first `capacity` of them are
 * written to host_triangles (call with capacity 0 to size the buffer).  Synchronous. */
int fluid_extract_surface(fluid_ctx* ctx, int image_id, float* host_triangles, uint64_t capacity,
                          uint64_t* count);

/* Diagnostics (synchronises the stream). */
typedef enum fluid_stat {
    FLUID_STAT_BRICKS = 0,       /* activity bricks of this context (256 x 4 x 16 cells each)          */
    FLUID_STAT_QUIET_BRICKS = 1, /* bricks the last fluid_run_step skipped in 07+08, 09+10+11 and 13   */
    FLUID_STAT_PARTICLE_SORTS = 2,   /* sorts of the particle storage so far (FLUID_OPT_PARTICLE_SORT)     */
    FLUID_STAT_PARTICLE_STRAYS = 3,  /* particles the last 01 found outside the bin they are stored in     */
    FLUID_STAT_PARTICLE_BINNED = 4,  /* 1 while 01 and 14 run on bins; 0 in slot order or while the flow    */
                                     /* moves the particles faster than sorting pays (tried again later)    */
    FLUID_STAT_PARTICLE_ENTRIES = 5, /* Z-slab contexts: entries (holes included) of the compact storage of the   */
                                     /* particles the slab owns, which 01, 14 and the search for leavers walk;  */
                                     /* whole-grid contexts: the number of slots                                */
    FLUID_STAT_OWNED_SQUEEZES = 6    /* times that storage had its holes squeezed out (sorts, which drop    */
                                     /* the holes too, are counted by FLUID_STAT_PARTICLE_SORTS)            */
} fluid_stat;
int fluid_get_stat(fluid_ctx* ctx, int stat, uint64_t* value);

#ifdef __cplusplus
}
#endif
#endif /* FLUID_ENGINE_H */
