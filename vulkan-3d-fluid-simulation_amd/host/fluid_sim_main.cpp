// fluid_sim_main.cpp — headless driver that follows the call order of the reference's main.cpp
// (/root/reference/main.cpp:68-124 set-up + init submit, :156-177 per-frame step submit) on the
// MI355X engine, through the C++ section-list mirror (include/fluid_flow_sections_amd.hpp).
//
//   fluid_sim <W> <H> <D> <frames> <jacobi_iters> [out_dir] [surface]
//
// With the word `surface` the lists are the reference's complete ones: the init list also clears the
// inertia image and the step list ends with the surface-prep passes 15…18 on the detailed grid
// (fluid_flow_sections.h:142, 339-388); the float density image is dumped as well.
//
// Uses the dam-break scene of the benchmark plan (reference spawn cube scaled to the grid,
// 8 particles per cell).  Prints per-frame time and, when out_dir is given, dumps VELOCITIES_1,
// CELL_TYPES, PRESSURES_1/2 and the particles as raw little-endian files for the parity tests.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <vector>

#include "../../include/fluid_flow_sections_amd.hpp"

using namespace fluid_amd;

static void dump(const std::string& path, const std::vector<uint8_t>& bytes) {
    std::ofstream f(path, std::ios::binary);
    f.write(reinterpret_cast<const char*>(bytes.data()), (std::streamsize)bytes.size());
}

int main(int argc, char** argv) {
    if (argc < 6) {
        std::fprintf(stderr, "usage: %s W H D frames jacobi_iters [out_dir]\n", argv[0]);
        return 2;
    }
    const Size3 fluid_size{(uint32_t)std::atoi(argv[1]), (uint32_t)std::atoi(argv[2]),
                           (uint32_t)std::atoi(argv[3])};
    const int frames = std::atoi(argv[4]);
    const uint32_t divergence_solve_iterations = (uint32_t)std::atoi(argv[5]);
    const std::string out_dir = argc > 6 ? argv[6] : "";
    const bool surface = argc > 7 && std::string(argv[7]) == "surface";
    try {
        // dam-break spawn cube: ratios of simulation_constants.h:48-50 to the 20^3 grid
        const float size[3] = {0.5f * fluid_size.x, 0.5f * fluid_size.y, 0.1f * fluid_size.z};
        const float offset[3] = {0.25f * fluid_size.x, 0.10f * fluid_size.y, 0.075f * fluid_size.z};
        uint32_t res[3];
        for (int i = 0; i < 3; i++) {
            // Python's round(): half to even, as tests/params.dam_break_params computes it
            res[i] = (uint32_t)std::max(1.0, std::nearbyint(2.0 * (double)size[i]));
        }
        const uint32_t particle_space_size = res[0] * res[1] * res[2];

        // main.cpp:68  parameters block
        SimulationParametersBufferData params(fluid_size, particle_space_size);
        for (int i = 0; i < 3; i++) {
            params.params().particle_spawn_cube_resolution[i] = res[i];
            params.params().particle_spawn_cube_offset[i] = offset[i];
            params.params().particle_spawn_cube_size[i] = size[i];
        }
        params.params().particle_spawn_cube_volume = particle_space_size;
        // main.cpp:73  all images and buffers
        SimulationDescriptors descriptors(params, particle_space_size, divergence_solve_iterations, -1,
                                          surface);
        FlowDescriptorContext& flow_context = descriptors;
        // main.cpp:76,79  section lists; :103-104 complete()
        SimulationInitializationSections init_sections(flow_context, surface);
        SimulationStepSections draw_section_list(flow_context, divergence_solve_iterations,
                                                 params.params().pressure_air);
        if (surface) draw_section_list.addSurfacePrepSections(flow_context);  // :339-388
        init_sections.complete();
        draw_section_list.complete();
        // main.cpp:111-124  run the init list and wait
        init_sections.run(flow_context);
        descriptors.waitIdle();
        // main.cpp:156-177  one step per frame
        for (int f = 0; f < frames; f++) {
            auto t0 = std::chrono::steady_clock::now();
            draw_section_list.run(flow_context);
            descriptors.waitIdle();
            double ms = std::chrono::duration<double, std::milli>(
                            std::chrono::steady_clock::now() - t0).count();
            std::printf("frame %d: %.3f ms (%zu sections)\n", f, ms, draw_section_list.size());
        }
        if (!out_dir.empty()) {
            const ImageAttachments imgs[] = {VELOCITIES_1, CELL_TYPES, PRESSURES_1, PRESSURES_2};
            const char* names[] = {"velocities_1", "cell_types", "pressures_1", "pressures_2"};
            for (int i = 0; i < 4; i++) {
                std::vector<uint8_t> buf(descriptors.bytes(imgs[i]));
                descriptors.download(imgs[i], buf.data(), buf.size());
                dump(out_dir + "/" + names[i] + ".bin", buf);
            }
            if (surface) {
                std::vector<uint8_t> buf(descriptors.bytes(PARTICLE_DENSITIES_FLOAT_1));
                descriptors.download(PARTICLE_DENSITIES_FLOAT_1, buf.data(), buf.size());
                dump(out_dir + "/float_densities_1.bin", buf);
            }
            std::vector<uint8_t> part((size_t)particle_space_size * 16);
            descriptors.download(PARTICLES_BUF, part.data(), part.size());
            dump(out_dir + "/particles.bin", part);
        }
        std::printf("ok %u particles\n", particle_space_size);
    } catch (const FluidError& e) {
        std::fprintf(stderr, "fluid engine error %d: %s\n", e.code, e.what());
        return 1;
    }
    return 0;
}
