// fluid_sim_slab.cpp — the reference's frame loop (/root/reference/main.cpp:103-111 set-up and init
// list, :156-177 one step list per frame) as ONE RANK of a multi-GPU run: launch it once per GPU.
//
//   fluid_sim_slab <rank> <world> <id_file> <W> <H> <D> <frames> <jacobi_iters> [out_dir]
//
// Every rank owns a Z slab of the grid (fluid_slab_partition) on HIP device <rank> (device 0 when the
// process sees one device, e.g. under a launcher that sets HIP_VISIBLE_DEVICES per rank) and drives it
// through include/fluid_slab.h; ghost planes and particles travel between Z-neighbours by RCCL Send/Recv.
// Bootstrap without MPI: rank 0 writes the communicator's 128-byte unique id to <id_file> (a path all
// ranks see, e.g. /dev/shm/fluid.id; written under a temporary name and renamed), the others wait for it.
// With <out_dir> each rank dumps its planes of VELOCITIES_1, CELL_TYPES, PRESSURES_1/2 and its particle
// buffer as raw little-endian files <name>.<rank>.bin (concatenating the image files by rank gives the
// global images; a particle slot is owned by the one rank whose file does not hold the tombstone).
// Dam-break scene of the benchmark plan, as host/fluid_sim_main.cpp.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <vector>

#include "../../include/fluid_slab.h"

static int die(fluid_slab* s, const char* what, int rc) {
    std::fprintf(stderr, "%s failed (%d): %s\n", what, rc, fluid_slab_last_error(s));
    return 1;
}

int main(int argc, char** argv) {
    if (argc < 9) {
        std::fprintf(stderr, "usage: %s rank world id_file W H D frames jacobi_iters [out_dir]\n", argv[0]);
        return 2;
    }
    const uint32_t rank = (uint32_t)std::atoi(argv[1]), world = (uint32_t)std::atoi(argv[2]);
    const std::string id_file = argv[3];
    const uint32_t W = (uint32_t)std::atoi(argv[4]), H = (uint32_t)std::atoi(argv[5]),
                   D = (uint32_t)std::atoi(argv[6]);
    const int frames = std::atoi(argv[7]);
    const uint32_t iters = (uint32_t)std::atoi(argv[8]);
    const std::string out_dir = argc > 9 ? argv[9] : "";

    // dam-break spawn cube: ratios of simulation_constants.h:48-50 to the 20^3 grid, 8 particles per cell
    const float size[3] = {0.5f * W, 0.5f * H, 0.1f * D};
    const float offset[3] = {0.25f * W, 0.10f * H, 0.075f * D};
    uint32_t res[3], capacity = 1;
    for (int i = 0; i < 3; i++) {
        res[i] = (uint32_t)std::max(1.0, std::nearbyint(2.0 * (double)size[i]));
        capacity *= res[i];
    }
    fluid_params p;
    fluid_params_default(&p, W, H, D, capacity);  // main.cpp:68
    for (int i = 0; i < 3; i++) {
        p.particle_spawn_cube_resolution[i] = res[i];
        p.particle_spawn_cube_offset[i] = offset[i];
        p.particle_spawn_cube_size[i] = size[i];
    }
    p.particle_spawn_cube_volume = capacity;

    fluid_slab_create_info info{};
    info.struct_bytes = sizeof info;
    info.rank = rank;
    info.world = world;
    info.device = (int)rank;
    info.params_blob = &p;
    info.particle_capacity = capacity;
    info.pressure_iterations = iters;
    info.overlap = -1;
    fluid_slab* s = nullptr;
    int rc = fluid_slab_create(&s, &info);  // main.cpp:73 — this rank's images and buffers
    if (rc == FLUID_ERR_NO_DEVICE || rc == FLUID_ERR_INVALID_ARG) {
        info.device = 0;  // the launcher gave this process one visible device
        rc = fluid_slab_create(&s, &info);
    }
    if (rc) return die(nullptr, "fluid_slab_create", rc);

    if (world > 1) {
        unsigned char id[FLUID_SLAB_RCCL_ID_BYTES];
        if (rank == 0) {
            if ((rc = fluid_slab_rccl_unique_id(id))) return die(nullptr, "fluid_slab_rccl_unique_id", rc);
            const std::string tmp = id_file + ".tmp";
            std::ofstream(tmp, std::ios::binary).write(reinterpret_cast<const char*>(id), sizeof id);
            std::rename(tmp.c_str(), id_file.c_str());
        } else {
            for (int tries = 0;; tries++) {
                std::ifstream f(id_file, std::ios::binary);
                if (f && f.read(reinterpret_cast<char*>(id), sizeof id)) break;
                if (tries > 6000) {
                    std::fprintf(stderr, "rank %u: no unique id in %s after 60 s\n", rank, id_file.c_str());
                    return 1;
                }
                std::this_thread::sleep_for(std::chrono::milliseconds(10));
            }
        }
        if ((rc = fluid_slab_attach_rccl(s, id))) return die(s, "fluid_slab_attach_rccl", rc);
    }

    if ((rc = fluid_slab_run_init(s))) return die(s, "fluid_slab_run_init", rc);  // main.cpp:111
    fluid_ctx* ctx = fluid_slab_engine(s);
    fluid_sync(ctx);
    for (int f = 0; f < frames; f++) {  // main.cpp:156-177
        const auto t0 = std::chrono::steady_clock::now();
        if ((rc = fluid_slab_run_step(s))) return die(s, "fluid_slab_run_step", rc);
        fluid_sync(ctx);
        const double ms =
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (rank == 0) std::printf("frame %d: %.3f ms\n", f, ms);
        if (f == 0 && world > 1) {
            // the loop's exchange schedule, measured on this machine's links with the scene's water in place
            // (PRESSURES_1 / _2 are overwritten: the next frame clears them anyway; no result changes)
            fluid_slab_tune_result tr{};
            if ((rc = fluid_slab_tune_exchange(s, &tr))) return die(s, "fluid_slab_tune_exchange", rc);
            if (rank == 0)
                std::printf("exchange schedule: %u planes per exchange, overlap mode %u\n", tr.halo_depth, tr.overlap);
        }
    }
    if (!out_dir.empty()) {
        const int imgs[] = {FLUID_IMG_VELOCITIES_1, FLUID_IMG_CELL_TYPES, FLUID_IMG_PRESSURES_1,
                            FLUID_IMG_PRESSURES_2};
        const char* names[] = {"velocities_1", "cell_types", "pressures_1", "pressures_2"};
        auto dump = [&](const std::string& name, const std::vector<uint8_t>& bytes) {
            std::ofstream f(out_dir + "/" + name + "." + std::to_string(rank) + ".bin", std::ios::binary);
            f.write(reinterpret_cast<const char*>(bytes.data()), (std::streamsize)bytes.size());
        };
        for (int i = 0; i < 4; i++) {
            uint64_t n = 0;
            fluid_image_bytes(ctx, imgs[i], &n);
            std::vector<uint8_t> buf(n);
            if ((rc = fluid_download_image(ctx, imgs[i], buf.data(), n))) return die(s, "download", rc);
            dump(names[i], buf);
        }
        std::vector<uint8_t> part((size_t)capacity * 16);
        if ((rc = fluid_download_buffer(ctx, FLUID_BUF_PARTICLES_BUF, part.data(), part.size())))
            return die(s, "download particles", rc);
        dump("particles", part);
    }
    uint64_t migrated = 0, exchanges = 0;
    fluid_slab_get_stat(s, FLUID_SLAB_STAT_MIGRATED, &migrated);
    fluid_slab_get_stat(s, FLUID_SLAB_STAT_EXCHANGES, &exchanges);
    std::printf("rank %u ok: %u particle slots, %llu plane exchanges, %llu particles handed over\n", rank,
                capacity, (unsigned long long)exchanges, (unsigned long long)migrated);
    fluid_slab_destroy(s);
    return 0;
}
