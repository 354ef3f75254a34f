/*
 * fluid_oracle.h — CPU oracle for the fluid-step path (sections 00…14).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as the
 * checker / the timed CPU baseline.  The engine (vulkan-3d-fluid-simulation_amd/) never links,
 * imports or falls back to it.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures for this path, cannot be
 * built or run here (its Vulkan wrapper library `just-a-vulkan-library` is an empty, unpinned
 * submodule; no Vulkan SDK / GLSL compiler / GPU ICD in the image), and has no Python implementation
 * to import.  This oracle is a single-threaded fp32 restatement of the GLSL in
 * /root/reference/shaders_fluid/NN_name/name.comp, pinned only by analytic known-answer tests derived
 * from that source (tests/test_oracle_kat.py, K1…K10 of SURVEY.md §8c) and by an independent numpy
 * restatement (tests/numpy_restatement.py).  Two choices are definitions, not observations:
 *   - the trilinear sampler (hardware fixed-point lerp in Vulkan) is defined in full fp32,
 *   - 03_update_air's border race is resolved "solid first" (SURVEY.md F5).
 * The Jacobi loop parity (first dispatch reads PRESSURES_1) is inferred, not pinned.
 *
 * Build: `make -C oracle` → oracle/liboracle.so  (gcc -O2 -ffp-contract=off, no -ffast-math).
 *
 * Layouts are the host layouts of include/fluid_engine.h: dense, x-fastest, RGBA32F velocities.
 */
#ifndef FLUID_ORACLE_H
#define FLUID_ORACLE_H

#include <stdint.h>

#include "../include/fluid_engine.h"

#ifdef __cplusplus
extern "C" {
#endif

/* All arrays are caller-owned, sized from p->fluid_size (cells) or `capacity` (particles). */

void oracle_fill_f32(float* dst, uint64_t count, float value);
void oracle_fill_u32(uint32_t* dst, uint64_t count, uint32_t value);
void oracle_fill_u8(uint8_t* dst, uint64_t count, uint8_t value);

void oracle_00_init_particles(const fluid_params* p, float* particles, uint64_t capacity);
void oracle_01_update_densities(const fluid_params* p, const float* particles, uint64_t capacity,
                                uint32_t* densities);
void oracle_02_update_water(const fluid_params* p, const uint32_t* densities, uint8_t* new_types);
void oracle_03_update_air(const fluid_params* p, uint8_t* new_types);
void oracle_04_compute_extrapolated_velocities(const fluid_params* p, const uint8_t* types,
                                               const float* v1, float* v2);
void oracle_05_set_extrapolated_velocities(const fluid_params* p, const uint8_t* new_types,
                                           const uint8_t* types, const float* v2, float* v1);
void oracle_06_update_cell_types(const fluid_params* p, const uint8_t* new_types, uint8_t* types);
void oracle_07_advect(const fluid_params* p, const uint8_t* types, const float* v1, float* v2);
void oracle_08_forces(const fluid_params* p, const uint8_t* types, float* v2);
void oracle_09_diffuse(const fluid_params* p, const uint8_t* types, const float* v2, float* v1,
                       int mode);
void oracle_10_solids(const fluid_params* p, const uint8_t* types, float* v1);
void oracle_11_compute_divergence(const fluid_params* p, const float* v1, float* div);
/* one dispatch of 12_solve_pressure with the push constant is_even_iteration */
void oracle_12_solve_pressure(const fluid_params* p, const uint8_t* types, const float* div,
                              float* p1, float* p2, uint32_t is_even_iteration);
/* the loop section: `iterations` dispatches, dispatch k has is_even_iteration = (k%2==0) */
void oracle_12_solve_pressure_loop(const fluid_params* p, const uint8_t* types, const float* div,
                                   float* p1, float* p2, uint32_t iterations);
/* opt-in red-black SOR on the same system (not in the reference; fluid_oracle.c) */
void oracle_12_sor_iteration(const fluid_params* p, const uint8_t* types, const float* div, float* pr,
                             float omega);
void oracle_12_sor_loop(const fluid_params* p, const uint8_t* types, const float* div, float* p1,
                        float* p2, float omega, uint32_t iterations);
void oracle_13_fix_divergence(const fluid_params* p, const uint8_t* types, const float* p2,
                              float* v1);
void oracle_14_particles(const fluid_params* p, const float* v1, float* particles,
                         uint64_t capacity);

/* ---- surface-prep passes on the detailed grid: fluid_size * detailed_resolution
 * cells per axis, x fastest, no ghost planes; images R32UI / R32UI / R32F / R32F ------------------- */
/* 15_update_detailed_densities/update_detailed_densities.comp:24-31 */
void oracle_15_update_detailed_densities(const fluid_params* p, const float* particles,
                                         uint64_t capacity, uint32_t* detailed);
/* 16_compute_detailed_densities_inertia/densities_inertia.comp:30-61 (in place on `inertia`) */
void oracle_16_compute_detailed_densities_inertia(const fluid_params* p, const uint32_t* detailed,
                                                  uint32_t* inertia);
/* 17_compute_float_densities/float_densities.comp:22-27 */
void oracle_17_compute_float_densities(const fluid_params* p, const uint32_t* inertia, float* f1);
/* 18_diffuse_float_densities/diffuse_densities.comp:45-62: one dispatch; is_even_iteration == 1 reads f1
 * and writes f2, else the reverse; cells whose simulation cell is SOLID are not written */
void oracle_18_diffuse_float_densities(const fluid_params* p, const uint8_t* types, float* f1,
                                       float* f2, uint32_t is_even_iteration);
/* FlowLoopPushConstantSection(float_density_diffuse_steps, ... "18_diffuse_float_densities"),
 * fluid_flow_sections.h:376-388: dispatch k has is_even_iteration = (k % 2 == 0) */
void oracle_18_diffuse_float_densities_loop(const fluid_params* p, const uint8_t* types, float* f1,
                                            float* f2, uint32_t iterations);

/* 31_render_surface as a triangle list (render_surface.vert:19-25, render_surface.geom:45-103): 12 floats per
 * triangle {p0, p1, p2, N}; counts[256] / edge_indices[256 * 15] = the reference's
 * surface_render_data/polygon_counts.txt / polygon_edge_indices.txt (marching_cubes.h:30-33) */
void oracle_31_extract_surface(const fluid_params* p, const float* density, const uint32_t* counts,
                               const uint32_t* edge_indices, float* out, uint64_t capacity, uint64_t* count);

/* sampler exposed for the known-answer tests: component `comp` of the trilinear sample at world
 * position (px,py,pz) as advect.comp:52-56 / particles.comp:28-36 take it */
float oracle_sample_velocity_component(const fluid_params* p, const float* v, float px, float py,
                                       float pz, int comp);

/* Whole-state helpers mirroring the two section lists. */
typedef struct oracle_state {
    fluid_params params;
    uint64_t particle_capacity;
    uint32_t pressure_iterations;
    int diffuse_mode;
    float* velocities_1;   /* 4 floats per cell */
    float* velocities_2;
    uint8_t* cell_types;
    uint8_t* new_cell_types;
    float* pressures_1;
    float* pressures_2;
    float* divergences;
    uint32_t* particle_densities;
    float* particles;      /* 4 floats per particle */
} oracle_state;

/* SimulationInitializationSections (fluid_flow_sections.h:139-154) */
void oracle_run_init(oracle_state* s);
/* SimulationStepSections 01a…14 (fluid_flow_sections.h:163-338) */
void oracle_run_step(oracle_state* s);

#ifdef __cplusplus
}
#endif
#endif
