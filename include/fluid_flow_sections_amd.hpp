// fluid_flow_sections_amd.hpp — C++17 host mirror of the reference's "operator API": the section
// lists of /root/reference/fluid_flow_sections.h, rebuilt on the C ABI of fluid_engine.h.
//
// A maintainer of the reference keeps the shape of their code: the same attachment enums
// (fluid_flow_sections.h:10-16), a SimulationParametersBufferData holding the 264-byte block
// (simulation_constants.h:153-174), a SimulationDescriptors that owns the device resources
// (:22-103), sections constructed from a shader directory name and collected in FlowSectionList
// subclasses (:136-156, :159-338) that are `complete()`d once and `run()` per frame
// (main.cpp:103-105,111,172).  What is gone: Vulkan (command buffers, descriptor usage lists,
// barriers — the engine's in-order HIP stream replaces them) and the render sections.
//
// Header-only; link against libfluid_engine.so.
#pragma once

#include <cstdint>
#include <cstring>
#include <fstream>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "fluid_engine.h"

namespace fluid_amd {

// fluid_flow_sections.h:10-16 — same names, same values
enum ImageAttachments {
    VELOCITIES_1, VELOCITIES_2, CELL_TYPES, NEW_CELL_TYPES, PRESSURES_1, PRESSURES_2, DIVERGENCES,
    PARTICLE_DENSITIES_IMG, DETAILED_DENSITIES_IMG, DETAILED_DENSITIES_INERTIA_IMG,
    PARTICLE_DENSITIES_FLOAT_1, PARTICLE_DENSITIES_FLOAT_2, IMAGE_COUNT
};
enum BufferAttachments {
    PARTICLES_BUF, MARCHING_CUBES_COUNTS_BUF, MARCHING_CUBES_EDGES_BUF, SIMULATION_PARAMS_BUF,
    BUFFER_COUNT
};
// simulation_constants.h:144-146
enum class CellType { CELL_INACTIVE, CELL_AIR, CELL_WATER, CELL_SOLID };

struct Size3 {
    uint32_t x, y, z;
    uint64_t volume() const { return (uint64_t)x * y * z; }
};

class FluidError : public std::runtime_error {
public:
    FluidError(int code, const std::string& what) : std::runtime_error(what), code(code) {}
    int code;
};

// simulation_constants.h:153-174 — the 264-byte std140 block, defaults of :7-139 for `fluid_size`
class SimulationParametersBufferData {
public:
    explicit SimulationParametersBufferData(Size3 fluid_size = {20, 20, 20},
                                            uint32_t particle_space_size = 1000000) {
        fluid_params_default(&m_params, fluid_size.x, fluid_size.y, fluid_size.z,
                             particle_space_size);
    }
    fluid_params& params() { return m_params; }
    const fluid_params& params() const { return m_params; }
    const void* data() const { return &m_params; }
    static constexpr size_t size() { return FLUID_PARAMS_BYTES; }

private:
    fluid_params m_params{};
};

// FlowDescriptorContext: all images and buffers of the simulation = one engine context
class FlowDescriptorContext {
public:
    FlowDescriptorContext() = default;
    explicit FlowDescriptorContext(fluid_ctx* ctx) : m_ctx(ctx) {}
    fluid_ctx* handle() const { return m_ctx; }
    void check(int rc) const {
        if (rc != FLUID_OK) throw FluidError(rc, fluid_last_error(m_ctx));
    }

private:
    fluid_ctx* m_ctx = nullptr;
};

// fluid_flow_sections.h:22-103 — allocates every attachment and uploads the params block
class SimulationDescriptors {
public:
    SimulationDescriptors(const SimulationParametersBufferData& fluid_params_uniform_buffer,
                          uint64_t particle_space_size, uint32_t divergence_solve_iterations = 200,
                          int device = -1, bool surface_prep = false,
                          uint32_t float_density_diffuse_steps = 4) {
        fluid_create_info info{};
        info.struct_bytes = sizeof info;
        // the detailed-grid images of sections 15-18 (:58-63): 16 B x detailed_resolution^3 per cell
        info.surface_prep = surface_prep ? 1u : 0u;
        info.surface_diffuse_steps = float_density_diffuse_steps;
        info.device = device;
        info.params_blob = fluid_params_uniform_buffer.data();
        info.particle_capacity = particle_space_size;
        info.pressure_iterations = divergence_solve_iterations;
        fluid_ctx* ctx = nullptr;
        int rc = fluid_create(&ctx, &info);
        if (rc != FLUID_OK) throw FluidError(rc, fluid_last_error(nullptr));
        m_owner.reset(ctx, fluid_destroy);
        m_context = FlowDescriptorContext(ctx);
    }
    operator FlowDescriptorContext&() { return m_context; }
    FlowDescriptorContext& context() { return m_context; }

    // not in the reference (it never reads back): host access for validation / checkpoints
    void upload(ImageAttachments img, const void* host, uint64_t bytes) {
        m_context.check(fluid_upload_image(m_context.handle(), img, host, bytes));
    }
    void download(ImageAttachments img, void* host, uint64_t bytes) {
        m_context.check(fluid_download_image(m_context.handle(), img, host, bytes));
    }
    void download(BufferAttachments buf, void* host, uint64_t bytes) {
        m_context.check(fluid_download_buffer(m_context.handle(), buf, host, bytes));
    }
    uint64_t bytes(ImageAttachments img) {
        uint64_t n = 0;
        m_context.check(fluid_image_bytes(m_context.handle(), img, &n));
        return n;
    }
    void waitIdle() { m_context.check(fluid_sync(m_context.handle())); }

private:
    std::shared_ptr<fluid_ctx> m_owner;
    FlowDescriptorContext m_context;
};

// MarchingCubesBuffers (marching_cubes.h:15-51): the two tables of the surface renderer, read from the same
// text files (whitespace-separated unsigned numbers) and copied into MARCHING_CUBES_COUNTS_BUF /
// MARCHING_CUBES_EDGES_BUF of a surface_prep context.  extractSurface() then gives the triangles
// 31_render_surface draws (fluid_extract_surface): {p0, p1, p2, normal} x 3 floats each.
class MarchingCubesBuffers {
public:
    void loadData(FlowDescriptorContext& ctx, const std::string& directory = "surface_render_data") {
        loadFromFile(ctx, MARCHING_CUBES_COUNTS_BUF, directory + "/polygon_counts.txt", 256);
        loadFromFile(ctx, MARCHING_CUBES_EDGES_BUF, directory + "/polygon_edge_indices.txt", 256 * 15);
    }
    static std::vector<float> extractSurface(FlowDescriptorContext& ctx,
                                             ImageAttachments density = PARTICLE_DENSITIES_FLOAT_2) {
        uint64_t n = 0;
        ctx.check(fluid_extract_surface(ctx.handle(), density, nullptr, 0, &n));
        std::vector<float> out(12 * n);
        if (n) ctx.check(fluid_extract_surface(ctx.handle(), density, out.data(), n, &n));
        return out;
    }

private:
    static void loadFromFile(FlowDescriptorContext& ctx, BufferAttachments buffer, const std::string& filename,
                             uint32_t size) {
        std::vector<uint32_t> data;
        data.reserve(size);
        std::ifstream data_file(filename);
        uint32_t a;
        for (uint32_t i = 0; i < size && (data_file >> a); i++) data.push_back(a);
        if (data.size() != size) throw FluidError(FLUID_ERR_SIZE_MISMATCH, filename + ": too few numbers");
        ctx.check(fluid_upload_buffer(ctx.handle(), buffer, data.data(), 4ull * size));
    }
};

// ClearValue of the reference's wrapper library: a float4 or a uint// ClearValue of the reference's wrapper library: a float4 or a uint
struct ClearValue {
    uint32_t bits[4];
    ClearValue(float r, float g, float b, float a) {
        const float v[4] = {r, g, b, a};
        std::memcpy(bits, v, sizeof bits);
    }
    explicit ClearValue(float r) : ClearValue(r, 0.f, 0.f, 0.f) {}
    explicit ClearValue(uint32_t v) : bits{v, 0, 0, 0} {}
    explicit ClearValue(int v) : bits{(uint32_t)v, 0, 0, 0} {}
};

class FlowSection {
public:
    virtual ~FlowSection() = default;
    virtual void run(FlowDescriptorContext& ctx) = 0;
    virtual std::string name() const = 0;
    // fluid_section_id of a plain compute section (one dispatch, no push constant), else -1: lets a
    // list hand runs of consecutive sections to fluid_run_section_group
    virtual int computeSectionId() const { return -1; }
};

// FlowClearColorSection(ctx, image, clear value) — fluid_flow_sections.h:140-142,163,298-299
class FlowClearColorSection : public FlowSection {
public:
    FlowClearColorSection(FlowDescriptorContext&, ImageAttachments image, ClearValue value)
        : m_image(image), m_value(value) {}
    void run(FlowDescriptorContext& ctx) override {
        ctx.check(fluid_clear_image(ctx.handle(), m_image, m_value.bits));
    }
    std::string name() const override { return "clear image " + std::to_string(m_image); }

private:
    ImageAttachments m_image;
    ClearValue m_value;
};

inline int sectionIdFromShaderDir(const std::string& dir) {
    static const char* const names[FLUID_SECTION_COUNT] = {
        nullptr, nullptr, "00_init_particles", nullptr, "01_update_densities", "02_update_water",
        "03_update_air", "04_compute_extrapolated_velocities", "05_set_extrapolated_velocities",
        "06_update_cell_types", "07_advect", "08_forces", "09_diffuse", "10_solids",
        "11_compute_divergence", nullptr, nullptr, "12_solve_pressure", "13_fix_divergence",
        "14_particles", nullptr, "15_update_detailed_densities",
        "16_compute_detailed_densities_inertia", "17_compute_float_densities",
        "18_diffuse_float_densities", nullptr};
    for (int i = 0; i < FLUID_SECTION_COUNT; i++)
        if (names[i] && dir == names[i]) return i;
    throw FluidError(FLUID_ERR_INVALID_ARG,
                     "no compute section named '" + dir + "' on the fluid-step path");
}

// FlowComputeSection(shader context, "NN_shader_dir", descriptors used, dispatch size): the
// descriptor-usage list and the dispatch size are derived by the engine (the kernels pick their own
// tiles), so only the directory name remains.
class FlowComputeSection : public FlowSection {
public:
    FlowComputeSection(FlowDescriptorContext&, const std::string& shader_dir_name)
        : m_name(shader_dir_name), m_section(sectionIdFromShaderDir(shader_dir_name)) {}
    void run(FlowDescriptorContext& ctx) override {
        ctx.check(fluid_run_section(ctx.handle(), m_section));
    }
    std::string name() const override { return m_name; }
    int computeSectionId() const override { return m_section; }

protected:
    std::string m_name;
    int m_section;
};

// a compute section with the `uint is_even_iteration` push constant (pressure.comp:29-31)
class FlowComputePushConstantSection : public FlowComputeSection {
public:
    using FlowComputeSection::FlowComputeSection;
    void run(FlowDescriptorContext& ctx, uint32_t is_even_iteration) {
        if (m_section == FLUID_SEC_18_DIFFUSE_FLOAT_DENSITIES)
            ctx.check(fluid_run_surface_diffuse_dispatch(ctx.handle(), is_even_iteration));
        else if (m_section == FLUID_SEC_12_SOLVE_PRESSURE)
            ctx.check(fluid_run_pressure_dispatch(ctx.handle(), is_even_iteration));
        else
            throw FluidError(FLUID_ERR_INVALID_ARG, m_name + " takes no push constant");
    }
    using FlowComputeSection::run;
    int section() const { return m_section; }
    int computeSectionId() const override { return -1; }
};

// FlowLoopPushConstantSection<FlowComputePushConstantSection>(iterations, …) — :300-313
template <typename Section>
class FlowLoopPushConstantSection : public FlowSection {
public:
    FlowLoopPushConstantSection(uint32_t iterations, FlowDescriptorContext& ctx,
                                const std::string& shader_dir_name)
        : m_iterations(iterations), m_inner(ctx, shader_dir_name) {}
    void run(FlowDescriptorContext& ctx) override {
        ctx.check(fluid_run_section_loop(ctx.handle(), m_inner.section(), m_iterations));
    }
    std::string name() const override {
        return m_inner.name() + " x" + std::to_string(m_iterations);
    }

private:
    uint32_t m_iterations;
    Section m_inner;
};

class FlowSectionList {
public:
    explicit FlowSectionList(FlowDescriptorContext& ctx) : m_ctx(&ctx) {}
    FlowSectionList(FlowDescriptorContext& ctx, std::initializer_list<FlowSection*> sections)
        : m_ctx(&ctx) {
        for (FlowSection* s : sections) m_sections.emplace_back(s);
    }
    void add(FlowSection* s) { m_sections.emplace_back(s); }
    // the reference updates descriptor sets here (main.cpp:103-105); nothing to bind on HIP
    void complete() { m_completed = true; }
    // reference: run(CommandBuffer&, FlowDescriptorContext&) records into a command buffer that
    // is submitted afterwards (main.cpp:111-122,170-176); here sections enqueue directly on the
    // context's stream, in list order.  A run of compute sections that are consecutive entries of
    // the reference's step list goes to the engine as one slice (fluid_run_section_group), which lets
    // it execute 04+05, 07+08 and 09+10+11 as single passes; results are the list's.
    void run(FlowDescriptorContext& ctx) {
        if (!m_completed) throw FluidError(FLUID_ERR_INVALID_ARG, "complete() the list first");
        const size_t n = m_sections.size();
        for (size_t i = 0; i < n;) {
            const int id = m_sections[i]->computeSectionId();
            size_t j = i + 1;
            if (m_group_slices && id >= FLUID_SEC_01_UPDATE_DENSITIES && id <= FLUID_SEC_14_PARTICLES &&
                id != FLUID_SEC_12_SOLVE_PRESSURE)
                while (j < n && m_sections[j]->computeSectionId() == id + (int)(j - i) &&
                       id + (int)(j - i) != FLUID_SEC_12_SOLVE_PRESSURE)
                    j++;
            if (j - i >= 2) {
                ctx.check(fluid_run_section_group(ctx.handle(), id, (uint32_t)(j - i)));
            } else {
                m_sections[i]->run(ctx);
                j = i + 1;
            }
            i = j;
        }
    }
    // false: one engine call per list entry, nothing grouped
    void setGroupSlices(bool on) { m_group_slices = on; }
    void run() { run(*m_ctx); }
    size_t size() const { return m_sections.size(); }
    const FlowSection& operator[](size_t i) const { return *m_sections[i]; }

private:
    FlowDescriptorContext* m_ctx;
    std::vector<std::unique_ptr<FlowSection>> m_sections;
    bool m_completed = false;
    bool m_group_slices = true;
};

// fluid_flow_sections.h:136-156 (minus the inertia clear :142, surface path)
class SimulationInitializationSections : public FlowSectionList {
public:
    // surface_prep: the context holds the detailed-grid images; the list then has the reference's
    // inertia clear (:142) as well
    explicit SimulationInitializationSections(FlowDescriptorContext& flow_context,
                                              bool surface_prep = false)
        : FlowSectionList{flow_context} {
        add(new FlowClearColorSection(flow_context, VELOCITIES_1, ClearValue(0.f, 0.f, 0.f, 0.f)));
        add(new FlowClearColorSection(flow_context, CELL_TYPES,
                                      ClearValue((uint32_t)CellType::CELL_INACTIVE)));
        if (surface_prep)
            add(new FlowClearColorSection(flow_context, DETAILED_DENSITIES_INERTIA_IMG,
                                          ClearValue((uint32_t)0)));
        add(new FlowComputeSection(flow_context, "00_init_particles"));
    }
};

// fluid_flow_sections.h:159-338 (01a … 14; 15-18 are the surface path)
class SimulationStepSections : public FlowSectionList {
public:
    SimulationStepSections(FlowDescriptorContext& flow_context,
                           uint32_t divergence_solve_iterations = 200,
                           float simulation_air_pressure = 1.0f)
        : FlowSectionList{
              flow_context,
              {new FlowClearColorSection(flow_context, PARTICLE_DENSITIES_IMG,
                                         ClearValue((uint32_t)0)),
               new FlowComputeSection(flow_context, "01_update_densities"),
               new FlowComputeSection(flow_context, "02_update_water"),
               new FlowComputeSection(flow_context, "03_update_air"),
               new FlowComputeSection(flow_context, "04_compute_extrapolated_velocities"),
               new FlowComputeSection(flow_context, "05_set_extrapolated_velocities"),
               new FlowComputeSection(flow_context, "06_update_cell_types"),
               new FlowComputeSection(flow_context, "07_advect"),
               new FlowComputeSection(flow_context, "08_forces"),
               new FlowComputeSection(flow_context, "09_diffuse"),
               new FlowComputeSection(flow_context, "10_solids"),
               new FlowComputeSection(flow_context, "11_compute_divergence"),
               new FlowClearColorSection(flow_context, PRESSURES_1,
                                         ClearValue(simulation_air_pressure)),
               new FlowClearColorSection(flow_context, PRESSURES_2,
                                         ClearValue(simulation_air_pressure)),
               new FlowLoopPushConstantSection<FlowComputePushConstantSection>(
                   divergence_solve_iterations, flow_context, "12_solve_pressure"),
               new FlowComputeSection(flow_context, "13_fix_divergence"),
               new FlowComputeSection(flow_context, "14_particles")}} {}

    // fluid_flow_sections.h:339-388: the surface-prep tail of the list (contexts created with
    // surface_prep); call before complete()
    void addSurfacePrepSections(FlowDescriptorContext& flow_context,
                                uint32_t float_density_diffuse_steps = 4) {
        add(new FlowClearColorSection(flow_context, DETAILED_DENSITIES_IMG, ClearValue((uint32_t)0)));
        add(new FlowComputeSection(flow_context, "15_update_detailed_densities"));
        add(new FlowComputeSection(flow_context, "16_compute_detailed_densities_inertia"));
        add(new FlowComputeSection(flow_context, "17_compute_float_densities"));
        add(new FlowLoopPushConstantSection<FlowComputePushConstantSection>(
            float_density_diffuse_steps, flow_context, "18_diffuse_float_densities"));
    }
};

}  // namespace fluid_amd
