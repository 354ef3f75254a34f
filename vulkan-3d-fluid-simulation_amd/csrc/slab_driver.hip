// slab_driver.hip — host side of include/fluid_slab.h: one rank of a multi-GPU run of the solver path
// (/root/reference main.cpp:156-177 is the single-GPU frame loop this replaces; the section order is
// fluid_flow_sections.h:139-154, 163-338).  Host C++ only: the schedule of section launches and plane
// exchanges for a Z slab, over the engine's C ABI (fluid_engine.h) and RCCL point-to-point (or a callback
// transport).  No kernels here.
//
// What a slab needs from its Z-neighbours, per step (ghost planes of an image = the neighbour's boundary
// planes; a domain face has none — those planes stay 0 = the reference's out-of-bounds load):
//   NEW_CELL_TYPES 1 plane after 02 and after 03      03 / 05 read z-1, z+1
//   VELOCITIES_1   `sampler_halo` planes after 05     07 back-traces (widened on demand, see advect())
//   VELOCITIES_2   1 plane before 09+10+11            11 differences with z+1 (what 10 makes of V2 there);
//                                                     list mode: VELOCITIES_1 after 10 instead
//   the Jacobi loop: h planes of the newest iterate every h sweeps (solve())
//   PRESSURES_2    1 plane after the loop             13 reads z-1
//   VELOCITIES_1   2 planes after 13                  14 samples around owned particles, 04 reads z+-1
//   particles that left the slab after 14             migrate()
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/fluid_slab.h"

namespace {

thread_local std::string g_slab_create_error;

std::string fmt(const char* f, ...) {
    char buf[640];
    va_list ap;
    va_start(ap, f);
    vsnprintf(buf, sizeof buf, f, ap);
    va_end(ap);
    return buf;
}

using Xfer = fluid_slab_xfer;
constexpr int32_t kBig = 1 << 30;

// ---- per-slab compute ------------------------------------------------------------------------------------
struct Backend {
    std::string err;
    virtual ~Backend() {}
    virtual int run_section(int id) = 0;
    virtual int run_group(int first, uint32_t n) = 0;
    virtual int image_planes(int image, int32_t first, uint32_t count, void** ptr, uint64_t* bytes) = 0;
    virtual int ghost_written(int image) = 0;
    virtual int loop_limits(uint32_t* max_sweeps, uint32_t* max_halo) = 0;
    virtual int loop_begin(uint32_t halo, fluid_slab_loop_buffer out[4], uint32_t* n) = 0;
    virtual int loop_halo_exchanged(uint32_t halo, bool first) = 0;
    virtual int loop_advance(uint32_t k, uint32_t sweeps, bool keep, int part, int32_t lo, int32_t hi,
                             int* written) = 0;
    virtual int loop_end() = 0;
    virtual int loop_planes(int which, int32_t first, uint32_t count, void** ptr, uint64_t* bytes) = 0;
    virtual int slab_status(uint32_t* violation) = 0;
    virtual int set_sampler_halo(uint32_t planes) = 0;
    virtual int sampler_reach(uint32_t* planes) = 0;
    virtual int sampler_wide_begin(uint32_t below, uint32_t above) = 0;
    virtual int sampler_wide_planes(int32_t first, uint32_t count, void** ptr, uint64_t* bytes) = 0;
    virtual int run_advect_wide(int with_forces) = 0;
    virtual int migrate_list(int which, void** list, uint32_t* capacity) = 0;
    virtual int collect(int reset, uint32_t counts[2], uint32_t* left) = 0;
    virtual int adopt_received(uint32_t nb, uint32_t na, uint32_t fwd[2]) = 0;
    virtual int sync() = 0;
    // a step with the engine's skipping (fluid_step_*): bricks far from the water, launches shaped to it
    virtual int step_begin(int section_list) = 0;
    virtual int step_end() = 0;
    virtual int build_activity() = 0;
    virtual int activity_layer(int which, void** ptr, uint64_t* bytes) = 0;
    virtual int step_status(uint32_t words[8]) = 0;
    virtual int set_box(int valid, uint32_t own, uint32_t y0, uint32_t y1, uint32_t x0, uint32_t x1) = 0;
    virtual hipStream_t stream() { return nullptr; }
    virtual bool on_device() const { return false; }
};

// the HIP engine on this slab
struct EngineBackend : Backend {
    fluid_ctx* c = nullptr;
    hipStream_t s = nullptr;
    bool fast = false;  // the working-buffer loop (fluid_pressure_loop_*) is in use for the running loop
    ~EngineBackend() override {
        if (c) fluid_destroy(c);
        if (s) (void)hipStreamDestroy(s);
    }
    int chk(int rc) {
        if (rc) err = fluid_last_error(c);
        return rc;
    }
    int planes_of(int rc, void** ptr, uint64_t* bytes, uint32_t count) {
        if (rc == FLUID_OK) *bytes *= count;  // consecutive planes are contiguous
        (void)ptr;
        return chk(rc);
    }
    int run_section(int id) override { return chk(fluid_run_section(c, id)); }
    int run_group(int first, uint32_t n) override { return chk(fluid_run_section_group(c, first, n)); }
    int image_planes(int image, int32_t first, uint32_t count, void** ptr, uint64_t* bytes) override {
        return planes_of(fluid_image_plane_ptr(c, image, first, ptr, bytes), ptr, bytes, count);
    }
    int ghost_written(int image) override { return chk(fluid_notify_ghost_planes_written(c, image)); }
    int loop_limits(uint32_t* max_sweeps, uint32_t* max_halo) override {
        *max_sweeps = 1;
        *max_halo = 1;
        fast = fluid_pressure_loop_available(c) == 1;  // may change with the engine's kernel option
        if (fast) {
            const int n = fluid_pressure_loop_max_sweeps(c);
            if (n < 0) return chk(n);
            *max_sweeps = (uint32_t)n;
            *max_halo = FLUID_LOOP_MAX_HALO;
        }
        return FLUID_OK;
    }
    int loop_begin(uint32_t halo, fluid_slab_loop_buffer out[4], uint32_t* n) override {
        if (!fast) {  // one dispatch per sweep on the images themselves: buffer i = PRESSURES_(i+1)
            out[0] = {0, 1};
            out[1] = {1, 1};
            *n = 2;
            return FLUID_OK;
        }
        int rc = fluid_pressure_loop_begin(c);
        if (rc) return chk(rc);
        const uint32_t aux = std::max<uint32_t>(halo - 1, 1);
        out[0] = {3, aux};  // neighbour mask
        out[1] = {4, aux};  // b_i
        out[2] = {0, halo};
        *n = 3;
        return FLUID_OK;
    }
    int loop_halo_exchanged(uint32_t halo, bool first) override {
        if (!fast) return FLUID_OK;
        return chk(fluid_pressure_loop_halo_exchanged(c, halo, first ? std::max<uint32_t>(halo - 1, 1) : 0));
    }
    int loop_advance(uint32_t k, uint32_t sweeps, bool keep, int part, int32_t lo, int32_t hi,
                     int* written) override {
        if (fast) {
            if (part != 0) return chk(fluid_pressure_loop_advance_part_n(c, sweeps, keep, part, lo, hi, written));
            return chk(fluid_pressure_loop_advance(c, sweeps, keep, written));
        }
        *written = (int)((k + 1) % 2);
        return chk(fluid_run_pressure_dispatch(c, k % 2 == 0 ? 1u : 0u));
    }
    int loop_end() override { return fast ? chk(fluid_pressure_loop_end(c)) : FLUID_OK; }
    int loop_planes(int which, int32_t first, uint32_t count, void** ptr, uint64_t* bytes) override {
        if (fast) return planes_of(fluid_pressure_loop_plane_ptr(c, which, first, ptr, bytes), ptr, bytes, count);
        return image_planes(which == 0 ? FLUID_IMG_PRESSURES_1 : FLUID_IMG_PRESSURES_2, first, count, ptr,
                            bytes);
    }
    int slab_status(uint32_t* v) override { return chk(fluid_slab_status(c, v)); }
    int set_sampler_halo(uint32_t n) override { return chk(fluid_set_sampler_halo(c, n)); }
    int sampler_reach(uint32_t* n) override { return chk(fluid_sampler_reach(c, n)); }
    int sampler_wide_begin(uint32_t lo, uint32_t hi) override { return chk(fluid_sampler_wide_begin(c, lo, hi)); }
    int sampler_wide_planes(int32_t first, uint32_t count, void** ptr, uint64_t* bytes) override {
        return planes_of(fluid_sampler_wide_plane_ptr(c, first, ptr, bytes), ptr, bytes, count);
    }
    int run_advect_wide(int f) override { return chk(fluid_run_advect_wide(c, f)); }
    int migrate_list(int which, void** list, uint32_t* cap) override {
        return chk(fluid_particles_migrate_list(c, which, list, cap));
    }
    int collect(int reset, uint32_t counts[2], uint32_t* left) override {
        return chk(fluid_particles_collect(c, reset, counts, left));
    }
    int adopt_received(uint32_t nb, uint32_t na, uint32_t fwd[2]) override {
        return chk(fluid_particles_adopt_received(c, nb, na, fwd));
    }
    int sync() override { return chk(fluid_sync(c)); }
    int step_begin(int l) override { return chk(fluid_step_begin(c, l)); }
    int step_end() override { return chk(fluid_step_end(c)); }
    int build_activity() override { return chk(fluid_step_build_activity(c)); }
    int activity_layer(int which, void** ptr, uint64_t* bytes) override {
        return chk(fluid_activity_layer_ptr(c, which, ptr, bytes));
    }
    int step_status(uint32_t words[8]) override { return chk(fluid_step_status(c, words)); }
    int set_box(int valid, uint32_t own, uint32_t y0, uint32_t y1, uint32_t x0, uint32_t x1) override {
        return chk(fluid_step_set_box(c, valid, own, y0, y1, x0, x1));
    }
    hipStream_t stream() override { return s; }
    bool on_device() const override { return true; }
};

// a table of callbacks (the CPU tests put the oracle behind it)
struct CallbackBackend : Backend {
    fluid_slab_backend cb{};
    int chk(int rc, const char* what) {
        if (rc) err = fmt("compute callback %s returned %d", what, rc);
        return rc;
    }
#define CB(name, ...) chk(cb.name(cb.user, ##__VA_ARGS__), #name)
    int run_section(int id) override { return CB(run_section, id); }
    int run_group(int first, uint32_t n) override { return CB(run_section_group, first, n); }
    int image_planes(int image, int32_t first, uint32_t count, void** ptr, uint64_t* bytes) override {
        return CB(image_planes, image, first, count, ptr, bytes);
    }
    int ghost_written(int image) override { return CB(ghost_planes_written, image); }
    int loop_limits(uint32_t* a, uint32_t* b) override { return CB(loop_limits, a, b); }
    int loop_begin(uint32_t halo, fluid_slab_loop_buffer out[4], uint32_t* n) override {
        return CB(loop_begin, halo, out, n);
    }
    int loop_halo_exchanged(uint32_t halo, bool first) override {
        return CB(loop_halo_exchanged, halo, first ? 1 : 0);
    }
    int loop_advance(uint32_t k, uint32_t sweeps, bool keep, int part, int32_t lo, int32_t hi,
                     int* written) override {
        return CB(loop_advance, k, sweeps, keep ? 1 : 0, part, lo, hi, written);
    }
    int loop_end() override { return CB(loop_end); }
    int loop_planes(int which, int32_t first, uint32_t count, void** ptr, uint64_t* bytes) override {
        return CB(loop_planes, which, first, count, ptr, bytes);
    }
    int slab_status(uint32_t* v) override { return CB(slab_status, v); }
    int set_sampler_halo(uint32_t n) override { return CB(set_sampler_halo, n); }
    int sampler_reach(uint32_t* n) override { return CB(sampler_reach, n); }
    int sampler_wide_begin(uint32_t lo, uint32_t hi) override { return CB(sampler_wide_begin, lo, hi); }
    int sampler_wide_planes(int32_t first, uint32_t count, void** ptr, uint64_t* bytes) override {
        return CB(sampler_wide_planes, first, count, ptr, bytes);
    }
    int run_advect_wide(int f) override { return CB(run_advect_wide, f); }
    int migrate_list(int which, void** list, uint32_t* cap) override { return CB(migrate_list, which, list, cap); }
    int collect(int reset, uint32_t counts[2], uint32_t* left) override { return CB(collect, reset, counts, left); }
    int adopt_received(uint32_t nb, uint32_t na, uint32_t fwd[2]) override {
        return CB(adopt_received, nb, na, fwd);
    }
    int sync() override { return CB(sync); }
    int step_begin(int l) override { return CB(step_begin, l); }
    int step_end() override { return CB(step_end); }
    int build_activity() override { return CB(build_activity); }
    int activity_layer(int which, void** ptr, uint64_t* bytes) override {
        return CB(activity_layer, which, ptr, bytes);
    }
    int step_status(uint32_t words[8]) override { return CB(step_status, words); }
    int set_box(int valid, uint32_t own, uint32_t y0, uint32_t y1, uint32_t x0, uint32_t x1) override {
        return CB(set_box, valid, own, y0, y1, x0, x1);
    }
#undef CB
};

// ---- transports --------------------------------------------------------------------------------------------
struct Transport {
    std::string err;
    virtual ~Transport() {}
    // true: exchange() enqueues on `stream` and returns (RCCL); false: it moves the bytes before it
    // returns and the caller must have synchronised the compute stream (callbacks)
    virtual bool stream_ordered() const = 0;
    virtual int exchange(const std::vector<Xfer>& ops, hipStream_t stream) = 0;
    virtual int allreduce_max(uint32_t* v, uint32_t n, hipStream_t stream) = 0;
};

struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                              hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
// librccl on demand.  It must sit on the SAME HIP runtime as this library: streams and device pointers
// are handles of one runtime instance, and a process may hold two (PyTorch wheels bundle their own
// libamdhip64 / libhsa-runtime64 / librccl next to the ROCm installation's).  So: find the file the HIP
// runtime this library is bound to was loaded from, and take the librccl of that directory — deep-bound, so
// that its own HIP calls resolve to that runtime too, whatever else is in the global scope.
const RcclApi* rccl_api(std::string& why) {
    static RcclApi api;
    static bool tried = false;
    static std::string failure;
    if (!tried) {
        tried = true;
        std::vector<std::string> names;
        Dl_info where{};
        if (dladdr(reinterpret_cast<void*>(&hipStreamCreateWithFlags), &where) && where.dli_fname) {
            std::string dir = where.dli_fname;
            const size_t slash = dir.rfind('/');
            if (slash != std::string::npos) {
                dir.resize(slash + 1);
                names.push_back(dir + "librccl.so.1");
                names.push_back(dir + "librccl.so");
            }
        }
        names.push_back("/opt/rocm/lib/librccl.so.1");
        names.push_back("librccl.so.1");
        std::string tried_names;
        for (const std::string& n : names) {
            if ((api.lib = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND))) break;
            tried_names += " " + n;
        }
        if (!api.lib) {
            failure = std::string("librccl could not be loaded (tried") + tried_names + "): " + dlerror();
        } else {
            auto sym = [&](const char* n) {
                void* p = dlsym(api.lib, n);
                if (!p && failure.empty()) failure = std::string("librccl lacks ") + n;
                return p;
            };
            api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
            api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
            api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
            api.CommCount = reinterpret_cast<decltype(api.CommCount)>(sym("ncclCommCount"));
            api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
            api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
            api.Send = reinterpret_cast<decltype(api.Send)>(sym("ncclSend"));
            api.Recv = reinterpret_cast<decltype(api.Recv)>(sym("ncclRecv"));
            api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
            api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
        }
    }
    why = failure;
    return failure.empty() ? &api : nullptr;
}

struct RcclTransport : Transport {
    const RcclApi* api = nullptr;
    ncclComm_t comm = nullptr;
    uint32_t* scratch = nullptr;  // device words for the small reductions
    static constexpr uint32_t kScratchWords = 1024;
    bool self_loop = false;       // fluid_slab_attach_rccl_self: a communicator of one, every peer is rank 0
    ~RcclTransport() override {
        if (scratch) (void)hipFree(scratch);
        if (comm && api) (void)api->CommDestroy(comm);
    }
    int fail(ncclResult_t r, const char* what) {
        err = fmt("%s failed: %s", what, api->GetErrorString(r));
        return FLUID_ERR_HIP;
    }
    bool stream_ordered() const override { return true; }
    int exchange(const std::vector<Xfer>& ops, hipStream_t stream) override {
        if (ops.empty()) return FLUID_OK;
        ncclResult_t r = api->GroupStart();
        if (r != ncclSuccess) return fail(r, "ncclGroupStart");
        for (const Xfer& x : ops) {
            const int peer = self_loop ? 0 : x.peer;
            if (x.flags & FLUID_XFER_SEND)
                r = api->Send(x.ptr, x.bytes, ncclUint8, peer, comm, stream);
            else
                r = api->Recv(x.ptr, x.bytes, ncclUint8, peer, comm, stream);
            if (r != ncclSuccess) {
                (void)api->GroupEnd();
                return fail(r, (x.flags & FLUID_XFER_SEND) ? "ncclSend" : "ncclRecv");
            }
        }
        r = api->GroupEnd();
        if (r != ncclSuccess) return fail(r, "ncclGroupEnd");
        return FLUID_OK;
    }
    int allreduce_max(uint32_t* v, uint32_t n, hipStream_t stream) override {
        if (n > kScratchWords) {
            err = "reduction too long";
            return FLUID_ERR_INVALID_ARG;
        }
        hipError_t e = hipMemcpyAsync(scratch, v, 4ull * n, hipMemcpyHostToDevice, stream);
        if (e == hipSuccess) {
            ncclResult_t r = api->AllReduce(scratch, scratch, n, ncclUint32, ncclMax, comm, stream);
            if (r != ncclSuccess) return fail(r, "ncclAllReduce");
            e = hipMemcpyAsync(v, scratch, 4ull * n, hipMemcpyDeviceToHost, stream);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) {
            err = std::string("reduction copies: ") + hipGetErrorString(e);
            return FLUID_ERR_HIP;
        }
        return FLUID_OK;
    }
};

struct CallbackTransport : Transport {
    fluid_slab_transport cb{};
    bool stream_ordered() const override { return false; }
    int exchange(const std::vector<Xfer>& ops, hipStream_t) override {
        if (ops.empty()) return FLUID_OK;
        const int rc = cb.exchange(cb.user, ops.data(), (uint32_t)ops.size());
        if (rc) err = fmt("transport callback exchange returned %d", rc);
        return rc;
    }
    int allreduce_max(uint32_t* v, uint32_t n, hipStream_t) override {
        const int rc = cb.allreduce_max_u32(cb.user, v, n);
        if (rc) err = fmt("transport callback allreduce_max_u32 returned %d", rc);
        return rc;
    }
};

// one process standing in for an interior rank: the i-th receive is filled by a device copy of the i-th
// plane range being sent (same sizes; the pairing a communicator of one gives ncclSend / ncclRecv to
// oneself), no communicator; reductions are the identity
struct LoopbackTransport : Transport {
    bool stream_ordered() const override { return true; }
    int exchange(const std::vector<Xfer>& ops, hipStream_t stream) override {
        std::vector<const Xfer*> sends, recvs;
        for (const Xfer& x : ops) ((x.flags & FLUID_XFER_SEND) ? sends : recvs).push_back(&x);
        for (size_t i = 0; i < recvs.size() && !sends.empty(); i++) {
            const Xfer* src = sends[i % sends.size()];
            hipError_t e = hipMemcpyAsync(recvs[i]->ptr, src->ptr, std::min(recvs[i]->bytes, src->bytes),
                                          hipMemcpyDeviceToDevice, stream);
            if (e != hipSuccess) {
                err = std::string("loopback copy: ") + hipGetErrorString(e);
                return FLUID_ERR_HIP;
            }
        }
        return FLUID_OK;
    }
    int allreduce_max(uint32_t*, uint32_t, hipStream_t) override { return FLUID_OK; }
};

}  // namespace

// ===========================================================================================================
struct fluid_slab {
    uint32_t rank = 0, world = 1;
    int device = -1;       // HIP device of the engine backend (-1: the compute is not on a device)
    int lo = -1, hi = -1;  // ranks of the neighbours below / above, -1 = domain face
    uint32_t W = 0, H = 0, D = 0, z0 = 0, dl = 0, thinnest = 0;
    uint64_t capacity = 0;
    uint32_t iterations = 200, halo_depth = 8;
    int overlap = FLUID_SLAB_OVERLAP_BOTH;
    bool grouped = true;
    int diffuse_mode = FLUID_DIFFUSE_REFERENCE_EXACT;
    uint32_t sampler_halo = 2, image_ghost = FLUID_IMAGE_GHOST_PLANES;
    std::unique_ptr<Backend> be;
    std::unique_ptr<Transport> tr;
    bool loopback = false;
    // faces (0 lower, 1 upper) with no water within FLUID_LOOP_MAX_HALO planes on either side, from the step's
    // table of boxes: the Jacobi loop's exchanges there would move planes that do not change
    bool dry_face[2] = {false, false};
    bool dry_valid = false;  // set by the step's table, used up by the loop that follows
    uint64_t stats[FLUID_SLAB_STAT_COUNT] = {0};
    std::string error;

    // overlapped exchanges on a stream-ordered transport
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_ready = nullptr, ev_done = nullptr;
    bool pending = false;

    std::map<std::tuple<int, int, uint32_t>, std::vector<Xfer>> plans;  // (kind, id, width)

    int fail(int code, const std::string& msg) {
        error = msg;
        return code;
    }
    int from_backend(int rc) {
        if (rc) error = be->err;
        return rc;
    }
    int from_transport(int rc) {
        if (rc) error = tr->err;
        return rc;
    }
    bool has_peers() const { return lo >= 0 || hi >= 0; }
};

#define BE(call)                                   \
    do {                                           \
        int rc_ = s->from_backend(s->be->call);    \
        if (rc_) return rc_;                       \
    } while (0)
#define TRY(expr)           \
    do {                    \
        int rc_ = (expr);   \
        if (rc_) return rc_;\
    } while (0)
#define HIPS(s, call)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return (s)->fail(FLUID_ERR_HIP, fmt("%s failed: %s", #call, hipGetErrorString(e_))); \
    } while (0)

namespace {

enum PlanKind { PLAN_IMAGE = 0, PLAN_LOOP = 1 };

// Boundary-plane exchange with the two Z-neighbours: send the first / last `width` owned planes down /
// up, receive their last / first owned planes into the ghost planes.  Pointers never move: built once.
// `skip`: bit 0 / 1 = leave the lower / upper face out (solve(): dry faces)
int get_plan(fluid_slab* s, int kind, int id, uint32_t width, const std::vector<Xfer>** out, uint32_t skip = 0) {
    const auto key = std::make_tuple(kind, id, width | (skip << 24));
    auto it = s->plans.find(key);
    if (it == s->plans.end()) {
        if (s->dl < width)
            return s->fail(FLUID_ERR_INVALID_ARG,
                           fmt("slab of %u planes is thinner than the halo (%u)", s->dl, width));
        std::vector<Xfer> plan;
        auto add = [&](int32_t first, int peer, bool send) -> int {
            Xfer x{};
            int rc = kind == PLAN_IMAGE ? s->be->image_planes(id, first, width, &x.ptr, &x.bytes)
                                        : s->be->loop_planes(id, first, width, &x.ptr, &x.bytes);
            if (rc) return s->from_backend(rc);
            x.peer = peer;
            x.flags = send ? FLUID_XFER_SEND : 0u;
            plan.push_back(x);
            return FLUID_OK;
        };
        const int32_t n = (int32_t)s->dl, w = (int32_t)width;
        if (s->lo >= 0 && !(skip & 1u)) {
            TRY(add(0, s->lo, true));
            TRY(add(-w, s->lo, false));
        }
        if (s->hi >= 0 && !(skip & 2u)) {
            TRY(add(n - w, s->hi, true));
            TRY(add(n, s->hi, false));
        }
        it = s->plans.emplace(key, std::move(plan)).first;
    }
    *out = &it->second;
    return FLUID_OK;
}

int ensure_comm_stream(fluid_slab* s) {
    if (s->comm_stream) return FLUID_OK;
    HIPS(s, hipStreamCreateWithFlags(&s->comm_stream, hipStreamNonBlocking));
    HIPS(s, hipEventCreateWithFlags(&s->ev_ready, hipEventDisableTiming));
    HIPS(s, hipEventCreateWithFlags(&s->ev_done, hipEventDisableTiming));
    return FLUID_OK;
}

// in line: ordered on the compute stream (RCCL) / after a synchronisation (callbacks)
int exchange_now(fluid_slab* s, const std::vector<Xfer>& ops) {
    if (ops.empty() || !s->tr) return FLUID_OK;
    s->stats[FLUID_SLAB_STAT_EXCHANGES]++;
    if (!s->tr->stream_ordered()) BE(sync());
    return s->from_transport(s->tr->exchange(ops, s->be->stream()));
}
// beside compute: on the communication stream, after what has been launched so far; launches made before
// exchange_finish() run beside it
int exchange_start(fluid_slab* s, const std::vector<Xfer>& ops) {
    if (ops.empty() || !s->tr) return FLUID_OK;
    s->stats[FLUID_SLAB_STAT_EXCHANGES]++;
    s->stats[FLUID_SLAB_STAT_OVERLAPPED]++;
    if (!s->tr->stream_ordered()) {  // moves the bytes now: the schedule is the same, the overlap is not
        BE(sync());
        return s->from_transport(s->tr->exchange(ops, nullptr));
    }
    TRY(ensure_comm_stream(s));
    HIPS(s, hipEventRecord(s->ev_ready, s->be->stream()));
    HIPS(s, hipStreamWaitEvent(s->comm_stream, s->ev_ready, 0));
    TRY(s->from_transport(s->tr->exchange(ops, s->comm_stream)));
    HIPS(s, hipEventRecord(s->ev_done, s->comm_stream));
    s->pending = true;
    return FLUID_OK;
}
int exchange_finish(fluid_slab* s) {
    if (!s->pending) return FLUID_OK;
    s->pending = false;
    HIPS(s, hipStreamWaitEvent(s->be->stream(), s->ev_done, 0));
    return FLUID_OK;
}

int exchange_image(fluid_slab* s, int image, uint32_t width) {
    if (!s->has_peers() || width == 0) return FLUID_OK;
    const std::vector<Xfer>* plan = nullptr;
    TRY(get_plan(s, PLAN_IMAGE, image, width, &plan));
    TRY(exchange_now(s, *plan));
    BE(ghost_written(image));
    return FLUID_OK;
}
int exchange_loop(fluid_slab* s, int buf, uint32_t width, uint32_t skip = 0) {
    if (!s->has_peers() || width == 0) return FLUID_OK;
    const std::vector<Xfer>* plan = nullptr;
    TRY(get_plan(s, PLAN_LOOP, buf, width, &plan, skip));
    return exchange_now(s, *plan);
}

int reduce_max(fluid_slab* s, uint32_t* v, uint32_t n) {
    if (s->world <= 1 || !s->tr) return FLUID_OK;
    if (!s->tr->stream_ordered()) BE(sync());
    return s->from_transport(s->tr->allreduce_max(v, n, s->be->stream()));
}

uint32_t effective_halo(fluid_slab* s, uint32_t max_sweeps, uint32_t max_halo) {
    // every rank must come to the same depth: limited by the thinnest slab of the partition
    uint32_t h = std::min(std::min(s->halo_depth, max_halo), s->thinnest);
    if (max_sweeps == 2 && h >= 2) h -= h % 2;  // pairs only: an odd plane would be exchanged for nothing
    return std::max<uint32_t>(h, 1);
}

// Sweeps of the next launch of a loop with `left` sweeps to go, at most `most` per launch: as many threes as
// possible, no single sweep at the end unless the loop has only one (4 = 2 + 2).  (engine.hip has the same.)
uint32_t next_launch_sweeps(uint32_t left, uint32_t most) {
    if (left <= 1 || most <= 1) return std::min<uint32_t>(left, 1);
    if (most == 2 || left == 2 || left == 4) return 2;
    return 3;
}
// The next launch of a loop with `left` sweeps to go and `valid` ghost planes of the newest iterate: the sweeps
// it applies and whether the slabs exchange h planes first.  Planes left over by the launches of three are used
// up by a launch of two (h = 8: 3 + 3 + 2 sweeps between two exchanges).  A function of the loop's position
// only: every rank comes to the same schedule.
struct NextLaunch {
    uint32_t sweeps;
    bool exchange;
};
NextLaunch plan_launch(uint32_t left, int32_t valid, uint32_t most, uint32_t h) {
    const uint32_t want = next_launch_sweeps(left, most);
    if ((int32_t)want <= valid) return {want, false};
    if (valid >= 2) {
        const uint32_t alt = next_launch_sweeps(left, std::min<uint32_t>(most, (uint32_t)valid));
        if (alt >= 2 && left - alt != 1) return {alt, false};
    }
    return {next_launch_sweeps(left, std::min(most, h)), true};
}

// FlowLoopPushConstantSection (fluid_flow_sections.h:300-313): dispatch k maps iterate k to
// iterate k+1; after N dispatches PRESSURES_1 holds the last even iterate, PRESSURES_2 the last odd one.
// Every sweep consumes one valid ghost plane per side of the newest iterate; when fewer are left than the
// next launch needs (2 or 3 for a launch of that many sweeps per pass) the slabs exchange h boundary planes —
// h sweeps then run without communication, the engine recomputing the shrinking ghost region: the same bytes
// on the wire as a plane per sweep in h times fewer messages.
// Overlap (h >= 4, slabs thicker than 2h): the pass before an exchange is split — the h planes per face
// that will be sent first, the exchange starts on the communication stream, the planes in between follow —
// and so is the pass after it: the planes that depend on owned planes only while the exchange is in
// flight, the rest once it has landed.  Same arithmetic, same iterates.
int solve(fluid_slab* s, uint32_t n) {
    uint32_t max_sweeps = 1, max_halo = 1;
    BE(loop_limits(&max_sweeps, &max_halo));
    const uint32_t h = effective_halo(s, max_sweeps, max_halo);
    s->stats[FLUID_SLAB_STAT_EFFECTIVE_HALO] = h;
    fluid_slab_loop_buffer bufs[4];
    uint32_t nbufs = 0;
    BE(loop_begin(h, bufs, &nbufs));
    for (uint32_t i = 0; i < nbufs; i++) TRY(exchange_loop(s, bufs[i].which, bufs[i].planes));
    // Dry faces (advect(): no water within the deepest halo on either side, this step): the planes the loop
    // would exchange there hold the constants of their non-water cells from the first sweep to the last, in all
    // three working buffers — they get them once, here, and the exchanges of the loop leave those faces out
    // (the launches still recompute the ghost region: the same constants).  Only with the engine's working
    // buffers (three of them, numbered 0 .. 2, constants laid down by the import pass).
    uint32_t dry = 0;
    if (max_sweeps >= 2 && nbufs == 3 && s->has_peers() && s->tr && s->dry_valid)
        dry = (s->dry_face[0] ? 1u : 0u) | (s->dry_face[1] ? 2u : 0u);
    s->dry_valid = false;  // (of one step's cell types: the next loop needs a new table)
    if (dry)
        for (int b = 1; b <= 2; b++) TRY(exchange_loop(s, b, h, 3u & ~dry));
    BE(loop_halo_exchanged(h, true));
    int32_t valid = (int32_t)h;  // valid ghost planes of the newest iterate
    int cur = 0;                 // buffer holding it
    const uint32_t most = (max_sweeps >= 2 && h >= 2) ? std::min(max_sweeps, h) : 1;
    // (a rank all of whose faces are dry has no exchange to hide: its passes stay whole)
    const bool all_dry = dry != 0 && dry == ((s->lo >= 0 ? 1u : 0u) | (s->hi >= 0 ? 2u : 0u));
    const bool split = s->overlap != FLUID_SLAB_OVERLAP_NONE && most >= 2 && h >= 4 && s->thinnest > 2 * h &&
                       s->has_peers() && s->tr && !all_dry;
    const int32_t dl = (int32_t)s->dl, hh = (int32_t)h;
    // interior of the pass before an exchange (local output planes)
    const int32_t before_lo = s->lo >= 0 ? hh : -kBig, before_hi = s->hi >= 0 ? dl - hh : kBig;
    bool in_flight = false;  // exchange started: finish before launching anything that reads ghost planes
    uint32_t k = 0;
    while (k < n) {
        if (all_dry) {
            // nothing to exchange with anybody for the whole loop: the ghost planes hold their constants in every
            // buffer, so they are as good as just exchanged before every launch, and the loop runs the schedule of a
            // whole-grid context (threes to the end) — this rank's own business, it shares no exchange with a neighbour
            if (k == 0 && n > h)  // (statistics: the exchanges a loop of n sweeps has at this depth)
                s->stats[FLUID_SLAB_STAT_DRY_FACE_SKIPS] += (uint64_t)((n - 1) / h) * ((dry & 1u) + ((dry >> 1) & 1u));
            // (exactly the planes the launch consumes: none is left over for it to recompute)
            const uint32_t want = next_launch_sweeps(n - k, most);
            BE(loop_halo_exchanged(want, false));
            valid = (int32_t)want;
        }
        // (with an exchange in flight `valid` already counts the planes it brings)
        const NextLaunch now = plan_launch(n - k, valid, most, h);
        const uint32_t sweeps = now.sweeps;
        const bool keep = sweeps >= 2 && n - k == sweeps;
        if (now.exchange) {
            if (dry) s->stats[FLUID_SLAB_STAT_DRY_FACE_SKIPS] += (dry & 1u) + ((dry >> 1) & 1u);
            TRY(exchange_loop(s, cur, h, dry));
            BE(loop_halo_exchanged(h, false));
            valid = hh;
        }
        const uint32_t left = n - k - sweeps;
        int32_t valid_after = sweeps >= 2 ? valid - (int32_t)sweeps : 0;
        const NextLaunch next = left ? plan_launch(left, valid_after, most, h) : NextLaunch{0, false};
        // interior of the pass after an exchange: the planes whose inputs are owned planes only
        const int32_t sw = (int32_t)sweeps;
        const int32_t after_lo = s->lo >= 0 ? sw : -kBig, after_hi = s->hi >= 0 ? dl - sw : kBig;
        if (in_flight && (s->overlap == FLUID_SLAB_OVERLAP_BEFORE || sweeps < 2)) {
            // half the overlap: only the pass before the exchange was split; wait, then a whole pass
            TRY(exchange_finish(s));
            in_flight = false;
            BE(loop_advance(k, sweeps, keep, 0, 0, 0, &cur));
        } else if (in_flight) {
            // first pass after the exchange started (valid == h: reported when it started)
            int w = 0;
            BE(loop_advance(k, sweeps, keep, FLUID_LOOP_PART_INTERIOR, after_lo, after_hi, &w));
            TRY(exchange_finish(s));
            in_flight = false;
            BE(loop_advance(k, sweeps, keep, FLUID_LOOP_PART_EDGES, after_lo, after_hi, &cur));
        } else if (split && sweeps >= 2 && next.exchange && next.sweeps >= 2) {
            int dst = 0;
            BE(loop_advance(k, sweeps, keep, FLUID_LOOP_PART_EDGES, before_lo, before_hi, &dst));
            const std::vector<Xfer>* plan = nullptr;
            TRY(get_plan(s, PLAN_LOOP, dst, h, &plan, dry));
            if (dry) s->stats[FLUID_SLAB_STAT_DRY_FACE_SKIPS] += (dry & 1u) + ((dry >> 1) & 1u);
            TRY(exchange_start(s, *plan));
            in_flight = true;
            BE(loop_advance(k, sweeps, keep, FLUID_LOOP_PART_INTERIOR, before_lo, before_hi, &cur));
            BE(loop_halo_exchanged(h, false));  // started; the next pass orders itself behind it
            valid_after = hh;
        } else {
            BE(loop_advance(k, sweeps, keep, 0, 0, 0, &cur));
        }
        valid = valid_after;
        k += sweeps;
    }
    if (in_flight) TRY(exchange_finish(s));  // cannot happen: a split needs a following fused pass
    BE(loop_end());
    return FLUID_OK;
}

int pressure_step(fluid_slab* s) {
    BE(run_section(FLUID_SEC_12A_CLEAR_PRESSURES_1));
    BE(run_section(FLUID_SEC_12B_CLEAR_PRESSURES_2));
    return solve(s, s->iterations);
}

// Slab of rank q (fluid_slab_partition)
void slab_of(uint32_t depth, uint32_t world, uint32_t q, uint32_t* z0, uint32_t* n) {
    const uint32_t base = depth / world, extra = depth % world;
    *n = base + (q < extra ? 1 : 0);
    *z0 = q * base + std::min(q, extra);
}

// 07 (or 07+08) when the back-traces reach beyond the image's ghost planes: every rank gets `reach`
// planes of VELOCITIES_1 per side (clipped to the grid) from whoever owns them — any rank, not only the
// neighbours — into the wide source, and the pass runs on that.
int advect_wide(fluid_slab* s, uint32_t reach) {
    s->stats[FLUID_SLAB_STAT_SAMPLER_WIDE]++;
    BE(sampler_wide_begin(reach, reach));
    std::vector<Xfer> ops;
    auto needs = [&](uint32_t qz0, uint32_t qn, int64_t* lo0, int64_t* lo1, int64_t* hi0, int64_t* hi1) {
        // global planes rank q needs beyond its slab: [lo0, lo1) below, [hi0, hi1) above
        *lo1 = qz0;
        *lo0 = std::max<int64_t>(0, (int64_t)qz0 - reach);
        *hi0 = (int64_t)qz0 + qn;
        *hi1 = std::min<int64_t>(s->D, (int64_t)qz0 + qn + reach);
    };
    const int64_t my0 = s->z0, my1 = (int64_t)s->z0 + s->dl;
    int64_t mlo0, mlo1, mhi0, mhi1;
    needs(s->z0, s->dl, &mlo0, &mlo1, &mhi0, &mhi1);
    for (uint32_t q = 0; q < s->world; q++) {
        if (q == s->rank) continue;
        uint32_t qz0, qn;
        slab_of(s->D, s->world, q, &qz0, &qn);
        const int64_t q0 = qz0, q1 = (int64_t)qz0 + qn;
        // what I receive from q: my needs within q's slab
        const int64_t ranges[2][2] = {{mlo0, mlo1}, {mhi0, mhi1}};
        for (auto& r : ranges) {
            const int64_t a = std::max(r[0], q0), b = std::min(r[1], q1);
            if (b <= a) continue;
            Xfer x{};
            BE(sampler_wide_planes((int32_t)(a - my0), (uint32_t)(b - a), &x.ptr, &x.bytes));
            x.peer = (int32_t)q;
            x.flags = 0;
            ops.push_back(x);
        }
        // what I send to q: q's needs within my slab, from the image itself
        int64_t qlo0, qlo1, qhi0, qhi1;
        needs(qz0, qn, &qlo0, &qlo1, &qhi0, &qhi1);
        const int64_t qranges[2][2] = {{qlo0, qlo1}, {qhi0, qhi1}};
        for (auto& r : qranges) {
            const int64_t a = std::max(r[0], my0), b = std::min(r[1], my1);
            if (b <= a) continue;
            Xfer x{};
            BE(image_planes(FLUID_IMG_VELOCITIES_1, (int32_t)(a - my0), (uint32_t)(b - a), &x.ptr, &x.bytes));
            x.peer = (int32_t)q;
            x.flags = FLUID_XFER_SEND;
            ops.push_back(x);
        }
    }
    // a pair of ranks may exchange two messages each way (below and above cannot both be q, but q's needs
    // below and above can both meet my slab when slabs are thin): per peer, sends and receives are listed
    // in the same global-plane order on both sides, which is the order a grouped Send / Recv matches them in
    TRY(exchange_now(s, ops));
    BE(run_advect_wide(s->grouped ? 1 : 0));
    if (!s->grouped) BE(run_section(FLUID_SEC_08_FORCES));
    return FLUID_OK;
}

// 07_advect (+ 08_forces) with as many ghost planes of VELOCITIES_1 as the flow needs (the
// back-trace of advect.comp:75-77 is not clamped).  Optimistic: the pass runs with the `sampler_halo`
// planes exchanged after 05; the kernels flag a tap beyond them; the flags are combined over the ranks
// (one 4-byte MAX reduction per step) and, if set anywhere, every rank redoes the pass with the halo the
// velocities call for — 07 reads VELOCITIES_1 and CELL_TYPES only, which it does not change.
int advect(fluid_slab* s) {
    auto run = [&]() -> int {
        if (s->grouped) {
            BE(run_group(FLUID_SEC_07_ADVECT, 2));
        } else {
            BE(run_section(FLUID_SEC_07_ADVECT));
            BE(run_section(FLUID_SEC_08_FORCES));
        }
        return FLUID_OK;
    };
    TRY(run());
    // One host synchronisation per step, here: the halo-violation flag of the pass and — it has had the pass
    // to arrive — the box of this slab's water (fluid_step_status).  One reduction tells every rank every
    // rank's flag and box; the launches of the pressure loop then cover the union of this slab's box with its
    // neighbours' (their water is what the ghost planes hold).
    uint32_t w[8];
    BE(step_status(w));
    uint32_t flag = w[0];
    constexpr uint32_t K = 8;  // words per rank: flag, box known, bricks, ~y_lo, y_hi, ~x_lo, x_hi, z range
    std::vector<uint32_t> table((size_t)K * s->world, 0u);
    uint32_t* mine = table.data() + (size_t)K * s->rank;
    mine[0] = flag;
    mine[1] = w[1];
    mine[2] = w[2];
    mine[3] = ~w[3];  // MAX of the complements = complement of the MIN
    mine[4] = w[4];
    mine[5] = ~w[5];
    mine[6] = w[6];
    mine[7] = w[7];
    TRY(reduce_max(s, table.data(), (uint32_t)table.size()));
    {
        // a face is dry when neither slab at it has water within the deepest halo of the loop: the planes an
        // exchange would move there hold constants (non-water cells are never written, pressure.comp:69)
        auto near = [&](int q, bool upper_face_of_q) {
            const uint32_t* t = table.data() + (size_t)K * q;
            if (t[1] == 0) return true;   // box not known: assume water
            if (t[2] == 0) return false;  // no water in that slab
            uint32_t qz0, qn;
            slab_of(s->D, s->world, (uint32_t)q, &qz0, &qn);
            const uint32_t zlo = t[7] & 0xFFFFu, zhi = t[7] >> 16;
            return upper_face_of_q ? zhi + FLUID_LOOP_MAX_HALO > qn : zlo < FLUID_LOOP_MAX_HALO;
        };
        const int me = (int)s->rank;
        if (s->loopback) {  // the neighbour is this slab's mirror image
            const bool wet = near(me, false) || near(me, true);
            s->dry_face[0] = s->lo >= 0 && !wet;
            s->dry_face[1] = s->hi >= 0 && !wet;
        } else {
            s->dry_face[0] = s->lo >= 0 && !near(me, false) && !near(s->lo, true);
            s->dry_face[1] = s->hi >= 0 && !near(me, true) && !near(s->hi, false);
        }
        s->dry_valid = true;
    }
    {
        bool known = true;
        uint32_t y0 = ~0u, y1 = 0, x0 = ~0u, x1 = 0;
        const int peers[3] = {(int)s->rank, s->loopback ? -1 : s->lo, s->loopback ? -1 : s->hi};
        for (int q : peers) {
            if (q < 0) continue;
            const uint32_t* t = table.data() + (size_t)K * q;
            known = known && t[1] != 0;
            if (t[2] == 0) continue;  // no water there
            y0 = std::min(y0, ~t[3]);
            y1 = std::max(y1, t[4]);
            x0 = std::min(x0, ~t[5]);
            x1 = std::max(x1, t[6]);
        }
        // (a loopback rehearsal takes its own box for its neighbours': same launch shapes as a real run of
        // a scene that looks alike on both sides of the face)
        if (y1 <= y0) y0 = y1 = x0 = x1 = 0;
        BE(set_box(known ? 1 : 0, mine[2], y0, y1, x0, x1));
    }
    flag = 0;
    for (uint32_t q = 0; q < s->world; q++) flag |= table[(size_t)K * q];
    if (!s->has_peers()) return FLUID_OK;  // a whole-grid context clamps its taps into the grid
    if (!flag) return FLUID_OK;
    s->stats[FLUID_SLAB_STAT_SAMPLER_RERUNS]++;
    uint32_t reach = 0;
    BE(sampler_reach(&reach));
    TRY(reduce_max(s, &reach, 1));
    reach = std::max<uint32_t>(reach, s->sampler_halo + 1);  // the flag says the current halo is too small
    if (reach <= s->image_ghost) {
        // fits the image's own ghost planes: keep the wider halo for the steps to come
        s->sampler_halo = reach;
        s->stats[FLUID_SLAB_STAT_SAMPLER_HALO] = reach;
        BE(set_sampler_halo(reach));
        TRY(exchange_image(s, FLUID_IMG_VELOCITIES_1, reach));
        TRY(run());
        // the bound is generous, but the kernels have the last word
        BE(slab_status(&flag));
        TRY(reduce_max(s, &flag, 1));
        if (!flag) return FLUID_OK;
        reach = s->D;
    }
    return advect_wide(s, reach);
}

// Particles that crossed a slab face change owner: down / up lists to the Z-neighbours, which adopt what
// they own and pass the rest on.  One table reduction per round tells every rank every count (and when
// to stop); in the common case of nobody leaving, that is all a step costs.
int migrate(fluid_slab* s) {
    if (s->world <= 1 || s->capacity == 0 || !s->tr || s->loopback) return FLUID_OK;
    uint32_t counts[2] = {0, 0}, left = 0;
    BE(collect(1, counts, &left));
    std::vector<uint32_t> table(3 * s->world);
    void* lists[4] = {nullptr, nullptr, nullptr, nullptr};
    uint32_t list_cap = 0;
    for (int round = 0; round < 4096; round++) {
        std::fill(table.begin(), table.end(), 0u);
        table[3 * s->rank] = counts[0];
        table[3 * s->rank + 1] = counts[1];
        table[3 * s->rank + 2] = left;
        TRY(reduce_max(s, table.data(), (uint32_t)table.size()));
        bool any = false, any_left = false;
        for (uint32_t q = 0; q < s->world; q++) {
            any |= (table[3 * q] | table[3 * q + 1] | table[3 * q + 2]) != 0;
            any_left |= table[3 * q + 2] != 0;
        }
        if (!any) return FLUID_OK;
        if (!lists[0])
            for (int i = 0; i < 4; i++) BE(migrate_list(i, &lists[i], &list_cap));
        constexpr uint64_t kEntry = 32;
        std::vector<Xfer> ops;
        uint32_t from_below = 0, from_above = 0;
        auto add = [&](void* ptr, uint32_t entries, int peer, bool send) {
            if (entries == 0) return;
            Xfer x{};
            x.ptr = ptr;
            x.bytes = kEntry * entries;
            x.peer = peer;
            x.flags = send ? FLUID_XFER_SEND : 0u;
            ops.push_back(x);
        };
        if (s->lo >= 0) {
            add(lists[FLUID_MIGRATE_SEND_DOWN], counts[0], s->lo, true);
            from_below = table[3 * s->lo + 1];  // what the neighbour below sends up
            add(lists[FLUID_MIGRATE_FROM_BELOW], from_below, s->lo, false);
        }
        if (s->hi >= 0) {
            add(lists[FLUID_MIGRATE_SEND_UP], counts[1], s->hi, true);
            from_above = table[3 * s->hi];      // what the neighbour above sends down
            add(lists[FLUID_MIGRATE_FROM_ABOVE], from_above, s->hi, false);
        }
        if (from_below > list_cap || from_above > list_cap)
            return s->fail(FLUID_ERR_OUT_OF_MEMORY, "a neighbour sends more particles than the lists hold");
        TRY(exchange_now(s, ops));
        s->stats[FLUID_SLAB_STAT_MIGRATED] += (uint64_t)counts[0] + counts[1];
        s->stats[FLUID_SLAB_STAT_MIGRATE_ROUNDS]++;
        BE(adopt_received(from_below, from_above, counts));  // counts: what is passed on
        left = 0;
        if (any_left) {  // somebody's list was full: everybody looks again (appending to what is passed on)
            BE(collect(0, counts, &left));
        }
    }
    return s->fail(FLUID_ERR_HIP, "particle hand-over did not terminate");
}

int run_init(fluid_slab* s) {
    BE(run_section(FLUID_SEC_INIT_CLEAR_VELOCITIES_1));
    BE(run_section(FLUID_SEC_INIT_CLEAR_CELL_TYPES));
    BE(run_section(FLUID_SEC_00_INIT_PARTICLES));
    // cleared images are uniform, but their value need not be the ghost planes' zero
    TRY(exchange_image(s, FLUID_IMG_CELL_TYPES, 1));
    TRY(exchange_image(s, FLUID_IMG_VELOCITIES_1, s->image_ghost));
    return FLUID_OK;
}

// SimulationStepSections 01a ... 14 (fluid_flow_sections.h:163-338) on this slab
int exchange_activity(fluid_slab* s) {
    BE(build_activity());
    if (!s->has_peers() || !s->tr) return FLUID_OK;
    void* ptr[4];
    uint64_t bytes[4];
    for (int i = 0; i < 4; i++) BE(activity_layer(i, &ptr[i], &bytes[i]));
    if (bytes[0] == 0) return FLUID_OK;  // this step does not skip
    std::vector<Xfer> ops;
    auto add = [&](int which, int peer, bool send) {
        Xfer x{};
        x.ptr = ptr[which];
        x.bytes = bytes[which];
        x.peer = peer;
        x.flags = send ? FLUID_XFER_SEND : 0u;
        ops.push_back(x);
    };
    if (s->lo >= 0) {
        add(0, s->lo, true);
        add(2, s->lo, false);
    }
    if (s->hi >= 0) {
        add(1, s->hi, true);
        add(3, s->hi, false);
    }
    return exchange_now(s, ops);
}

int run_step_sections(fluid_slab* s) {
    BE(run_section(FLUID_SEC_01A_CLEAR_PARTICLE_DENSITIES));
    BE(run_section(FLUID_SEC_01_UPDATE_DENSITIES));  // owned particles only, into owned planes
    BE(run_section(FLUID_SEC_02_UPDATE_WATER));
    TRY(exchange_image(s, FLUID_IMG_NEW_CELL_TYPES, 1));  // 03 looks at z-1 / z+1
    BE(run_section(FLUID_SEC_03_UPDATE_AIR));
    TRY(exchange_image(s, FLUID_IMG_NEW_CELL_TYPES, 1));  // 05 reads the final new types at z-1
    if (s->grouped) {                                     // old types / V1 at z+-1: still current
        BE(run_group(FLUID_SEC_04_COMPUTE_EXTRAPOLATED_VELOCITIES, 2));
    } else {
        BE(run_section(FLUID_SEC_04_COMPUTE_EXTRAPOLATED_VELOCITIES));
        BE(run_section(FLUID_SEC_05_SET_EXTRAPOLATED_VELOCITIES));
    }
    TRY(exchange_image(s, FLUID_IMG_VELOCITIES_1, s->sampler_halo));  // 07 samples V1 around each cell
    BE(run_section(FLUID_SEC_06_UPDATE_CELL_TYPES));  // carries one ghost plane per side along
    TRY(exchange_activity(s));                        // which bricks hold water, also across the faces
    TRY(advect(s));                                   // 07, 08
    const bool intended = s->diffuse_mode == FLUID_DIFFUSE_INTENDED;
    if (s->grouped && s->W % 4 == 0 && !intended) {
        TRY(exchange_image(s, FLUID_IMG_VELOCITIES_2, 1));  // 11, on what 10 makes of V2 at z+1
        BE(run_group(FLUID_SEC_09_DIFFUSE, 3));
    } else {
        if (intended) TRY(exchange_image(s, FLUID_IMG_VELOCITIES_2, 1));  // the diffusion stencil: z-1, z+1
        BE(run_section(FLUID_SEC_09_DIFFUSE));
        BE(run_section(FLUID_SEC_10_SOLIDS));
        TRY(exchange_image(s, FLUID_IMG_VELOCITIES_1, 1));  // 11 reads V1 at z+1
        BE(run_section(FLUID_SEC_11_COMPUTE_DIVERGENCE));
    }
    TRY(pressure_step(s));                                // 12a, 12b, the 12_solve_pressure loop
    TRY(exchange_image(s, FLUID_IMG_PRESSURES_2, 1));     // 13 reads P2 at z-1
    BE(run_section(FLUID_SEC_13_FIX_DIVERGENCE));
    // 14 samples V1 around particles this slab owns: taps in the planes z-1 .. z+1 of the particle's own,
    // plus — when a coordinate rounds up onto the slab face — a zero-weight tap one plane further; 04 of the
    // next step reads z-1 / z+1
    TRY(exchange_image(s, FLUID_IMG_VELOCITIES_1, std::min<uint32_t>(2, s->image_ghost)));
    BE(run_section(FLUID_SEC_14_PARTICLES));
    return migrate(s);
}

int run_step(fluid_slab* s) {
    BE(step_begin(s->grouped ? 0 : 1));
    const int rc = run_step_sections(s);
    const int rc2 = s->from_backend(s->be->step_end());
    return rc ? rc : rc2;
}

int fill_common(fluid_slab* s, const fluid_slab_create_info* info, const fluid_params& p) {
    s->rank = info->rank;
    s->world = info->world;
    s->W = p.fluid_size[0];
    s->H = p.fluid_size[1];
    s->D = p.fluid_size[2];
    int rc = fluid_slab_partition(s->D, s->world, s->rank, &s->z0, &s->dl);
    if (rc) return rc;
    s->thinnest = s->D / s->world;
    s->lo = s->rank > 0 ? (int)s->rank - 1 : -1;
    s->hi = s->rank + 1 < s->world ? (int)s->rank + 1 : -1;
    s->capacity = info->particle_capacity;
    if (s->capacity == 0)
        s->capacity = (uint64_t)p.particle_compute_size[0] * (uint64_t)p.particle_compute_size[1];
    s->iterations = info->pressure_iterations ? info->pressure_iterations : 200;
    s->halo_depth = info->halo_depth ? info->halo_depth : 8;
    s->overlap = info->overlap < 0 ? FLUID_SLAB_OVERLAP_BOTH : info->overlap;
    s->grouped = info->section_list == 0;
    s->diffuse_mode = info->diffuse_mode;
    s->image_ghost = std::min<uint32_t>(FLUID_IMAGE_GHOST_PLANES, s->thinnest);
    s->sampler_halo = std::min<uint32_t>(info->sampler_halo ? info->sampler_halo : 2, s->image_ghost);
    s->stats[FLUID_SLAB_STAT_SAMPLER_HALO] = s->sampler_halo;
    return FLUID_OK;
}

int check_info(const fluid_slab_create_info* info, fluid_params* p) {
    if (!info || info->struct_bytes < sizeof(fluid_slab_create_info) || !info->params_blob) {
        g_slab_create_error = "fluid_slab_create: null argument or struct_bytes too small";
        return FLUID_ERR_INVALID_ARG;
    }
    memcpy(p, info->params_blob, sizeof *p);
    if (info->world == 0 || info->rank >= info->world || p->fluid_size[2] < info->world) {
        g_slab_create_error = fmt("cannot give rank %u of %u a slab of %u planes", info->rank, info->world,
                                  p->fluid_size[2]);
        return FLUID_ERR_INVALID_ARG;
    }
    if (info->overlap > FLUID_SLAB_OVERLAP_BOTH ||
        (info->diffuse_mode != FLUID_DIFFUSE_REFERENCE_EXACT && info->diffuse_mode != FLUID_DIFFUSE_INTENDED) ||
        info->sampler_halo > FLUID_IMAGE_GHOST_PLANES) {
        g_slab_create_error = "fluid_slab_create: overlap, diffuse_mode or sampler_halo out of range";
        return FLUID_ERR_INVALID_ARG;
    }
    return FLUID_OK;
}

}  // namespace

// ===========================================================================================================
extern "C" {

int fluid_slab_partition(uint32_t depth, uint32_t world, uint32_t rank, uint32_t* z_begin, uint32_t* z_count) {
    if (world == 0 || depth < world || rank >= world || !z_begin || !z_count) return FLUID_ERR_INVALID_ARG;
    slab_of(depth, world, rank, z_begin, z_count);
    return FLUID_OK;
}

int fluid_slab_create(fluid_slab** out, const fluid_slab_create_info* info) {
    if (!out) return FLUID_ERR_INVALID_ARG;
    *out = nullptr;
    fluid_params p;
    int rc = check_info(info, &p);
    if (rc) return rc;
    std::unique_ptr<fluid_slab> s(new fluid_slab());
    rc = fill_common(s.get(), info, p);
    if (rc) return rc;
    std::unique_ptr<EngineBackend> be(new EngineBackend());
    fluid_create_info ci{};
    ci.struct_bytes = sizeof ci;
    ci.device = info->device;
    ci.params_blob = info->params_blob;
    ci.particle_capacity = info->particle_capacity;
    ci.pressure_iterations = info->pressure_iterations;
    if (s->world > 1) {
        ci.slab_z_begin = s->z0;
        ci.slab_z_count = s->dl;
    }
    // the driver owns the stream: RCCL operations and the engine's kernels are ordered on it
    if (info->device >= 0 && hipSetDevice(info->device) != hipSuccess) {
        g_slab_create_error = "hipSetDevice failed";
        return FLUID_ERR_NO_DEVICE;
    }
    hipError_t e = hipStreamCreateWithFlags(&be->s, hipStreamNonBlocking);
    if (e != hipSuccess) {
        g_slab_create_error = std::string("no usable HIP device: ") + hipGetErrorString(e);
        return FLUID_ERR_NO_DEVICE;
    }
    ci.hip_stream = be->s;
    rc = fluid_create(&be->c, &ci);
    if (rc) {
        g_slab_create_error = fluid_last_error(nullptr);
        return rc;
    }
    if (info->diffuse_mode != FLUID_DIFFUSE_REFERENCE_EXACT) (void)fluid_set_diffuse_mode(be->c, info->diffuse_mode);
    if (s->world > 1) (void)fluid_set_sampler_halo(be->c, s->sampler_halo);
    s->be = std::move(be);
    if (info->device >= 0)
        s->device = info->device;
    else
        (void)hipGetDevice(&s->device);
    *out = s.release();
    return FLUID_OK;
}

int fluid_slab_create_custom(fluid_slab** out, const fluid_slab_create_info* info,
                             const fluid_slab_backend* backend) {
    if (!out) return FLUID_ERR_INVALID_ARG;
    *out = nullptr;
    fluid_params p;
    int rc = check_info(info, &p);
    if (rc) return rc;
    if (!backend || backend->struct_bytes < sizeof(fluid_slab_backend)) {
        g_slab_create_error = "fluid_slab_create_custom: backend table missing or too small";
        return FLUID_ERR_INVALID_ARG;
    }
    const void* const* fns = reinterpret_cast<const void* const*>(&backend->run_section);
    const size_t nfns = (sizeof(fluid_slab_backend) - offsetof(fluid_slab_backend, run_section)) / sizeof(void*);
    for (size_t i = 0; i < nfns; i++)
        if (!fns[i]) {
            g_slab_create_error = fmt("fluid_slab_create_custom: callback %zu of the table is null", i);
            return FLUID_ERR_INVALID_ARG;
        }
    std::unique_ptr<fluid_slab> s(new fluid_slab());
    rc = fill_common(s.get(), info, p);
    if (rc) return rc;
    std::unique_ptr<CallbackBackend> be(new CallbackBackend());
    be->cb = *backend;
    s->be = std::move(be);
    if (s->world > 1) (void)s->be->set_sampler_halo(s->sampler_halo);
    *out = s.release();
    return FLUID_OK;
}

void fluid_slab_destroy(fluid_slab* s) {
    if (!s) return;
    if (s->be) (void)s->be->sync();
    if (s->comm_stream) (void)hipStreamSynchronize(s->comm_stream);
    s->tr.reset();  // the communicator before the memory it addressed
    if (s->ev_ready) (void)hipEventDestroy(s->ev_ready);
    if (s->ev_done) (void)hipEventDestroy(s->ev_done);
    if (s->comm_stream) (void)hipStreamDestroy(s->comm_stream);
    delete s;
}

const char* fluid_slab_last_error(const fluid_slab* s) {
    return s ? s->error.c_str() : g_slab_create_error.c_str();
}

fluid_ctx* fluid_slab_engine(fluid_slab* s) {
    if (!s) return nullptr;
    auto* e = dynamic_cast<EngineBackend*>(s->be.get());
    return e ? e->c : nullptr;
}

int fluid_slab_get_slab(const fluid_slab* s, uint32_t* z_begin, uint32_t* z_count) {
    if (!s) return FLUID_ERR_INVALID_ARG;
    if (z_begin) *z_begin = s->z0;
    if (z_count) *z_count = s->dl;
    return FLUID_OK;
}

int fluid_slab_rccl_unique_id(void* id_out) {
    static_assert(sizeof(ncclUniqueId) == FLUID_SLAB_RCCL_ID_BYTES, "ncclUniqueId size");
    if (!id_out) return FLUID_ERR_INVALID_ARG;
    std::string why;
    const RcclApi* api = rccl_api(why);
    if (!api) {
        g_slab_create_error = why;
        return FLUID_ERR_UNSUPPORTED;
    }
    ncclUniqueId id;
    const ncclResult_t r = api->GetUniqueId(&id);
    if (r != ncclSuccess) {
        g_slab_create_error = std::string("ncclGetUniqueId: ") + api->GetErrorString(r);
        return FLUID_ERR_HIP;
    }
    memcpy(id_out, &id, sizeof id);
    return FLUID_OK;
}

int fluid_slab_attach_rccl(fluid_slab* s, const void* id_bytes) {
    if (!s) return FLUID_ERR_INVALID_ARG;
    if (!id_bytes) return s->fail(FLUID_ERR_INVALID_ARG, "null unique id");
    if (s->device >= 0) HIPS(s, hipSetDevice(s->device));  // the communicator binds to the current device
    if (!s->be->on_device())
        return s->fail(FLUID_ERR_UNSUPPORTED, "RCCL moves device memory: this driver computes on the host");
    std::string why;
    const RcclApi* api = rccl_api(why);
    if (!api) return s->fail(FLUID_ERR_UNSUPPORTED, why);
    std::unique_ptr<RcclTransport> t(new RcclTransport());
    t->api = api;
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof id);
    const ncclResult_t r = api->CommInitRank(&t->comm, (int)s->world, id, (int)s->rank);
    if (r != ncclSuccess)
        return s->fail(FLUID_ERR_HIP, std::string("ncclCommInitRank: ") + api->GetErrorString(r));
    void* ptr = nullptr;
    HIPS(s, hipMalloc(&ptr, 4ull * RcclTransport::kScratchWords));
    t->scratch = static_cast<uint32_t*>(ptr);
    int ranks = 0;  // what the communicator itself says its size is (bench.py reports it)
    if (api->CommCount(t->comm, &ranks) == ncclSuccess) s->stats[FLUID_SLAB_STAT_RCCL_RANKS] = (uint64_t)ranks;
    s->tr = std::move(t);
    s->plans.clear();
    return FLUID_OK;
}

int fluid_slab_attach_transport(fluid_slab* s, const fluid_slab_transport* transport) {
    if (!s) return FLUID_ERR_INVALID_ARG;
    if (!transport || transport->struct_bytes < sizeof(fluid_slab_transport) || !transport->exchange ||
        !transport->allreduce_max_u32)
        return s->fail(FLUID_ERR_INVALID_ARG, "transport table missing, too small or incomplete");
    std::unique_ptr<CallbackTransport> t(new CallbackTransport());
    t->cb = *transport;
    s->tr = std::move(t);
    s->plans.clear();
    return FLUID_OK;
}

int fluid_slab_attach_loopback(fluid_slab* s, int has_lower, int has_upper) {
    if (!s) return FLUID_ERR_INVALID_ARG;
    if (!s->be->on_device()) return s->fail(FLUID_ERR_UNSUPPORTED, "loopback copies device memory");
    s->tr.reset(new LoopbackTransport());
    s->loopback = true;
    s->lo = has_lower ? (int)s->rank : -1;
    s->hi = has_upper ? (int)s->rank : -1;
    s->plans.clear();
    return FLUID_OK;
}

int fluid_slab_attach_rccl_self(fluid_slab* s, int has_lower, int has_upper) {
    if (!s) return FLUID_ERR_INVALID_ARG;
    if (s->device >= 0) HIPS(s, hipSetDevice(s->device));  // the communicator binds to the current device
    if (!s->be->on_device())
        return s->fail(FLUID_ERR_UNSUPPORTED, "RCCL moves device memory: this driver computes on the host");
    std::string why;
    const RcclApi* api = rccl_api(why);
    if (!api) return s->fail(FLUID_ERR_UNSUPPORTED, why);
    std::unique_ptr<RcclTransport> t(new RcclTransport());
    t->api = api;
    t->self_loop = true;
    ncclUniqueId id;
    ncclResult_t r = api->GetUniqueId(&id);
    if (r == ncclSuccess) r = api->CommInitRank(&t->comm, 1, id, 0);
    if (r != ncclSuccess)
        return s->fail(FLUID_ERR_HIP, std::string("communicator of one: ") + api->GetErrorString(r));
    void* ptr = nullptr;
    HIPS(s, hipMalloc(&ptr, 4ull * RcclTransport::kScratchWords));
    t->scratch = static_cast<uint32_t*>(ptr);
    int ranks = 0;
    if (api->CommCount(t->comm, &ranks) == ncclSuccess) s->stats[FLUID_SLAB_STAT_RCCL_RANKS] = (uint64_t)ranks;
    s->tr = std::move(t);
    s->loopback = true;
    s->lo = has_lower ? (int)s->rank : -1;
    s->hi = has_upper ? (int)s->rank : -1;
    s->plans.clear();
    return FLUID_OK;
}

static int need_transport(fluid_slab* s) {
    if (s->world > 1 && !s->tr)
        return s->fail(FLUID_ERR_INVALID_ARG,
                       "no transport attached: fluid_slab_attach_rccl / _transport first");
    return FLUID_OK;
}

int fluid_slab_run_init(fluid_slab* s) {
    if (!s) return FLUID_ERR_INVALID_ARG;
    TRY(need_transport(s));
    return run_init(s);
}

int fluid_slab_run_step(fluid_slab* s) {
    if (!s) return FLUID_ERR_INVALID_ARG;
    TRY(need_transport(s));
    return run_step(s);
}

int fluid_slab_pressure_step(fluid_slab* s) {
    if (!s) return FLUID_ERR_INVALID_ARG;
    TRY(need_transport(s));
    return pressure_step(s);
}

int fluid_slab_tune_exchange(fluid_slab* s, fluid_slab_tune_result* result) {
    if (!s) return FLUID_ERR_INVALID_ARG;
    TRY(need_transport(s));
    fluid_slab_tune_result r{};
    r.halo_depth = s->halo_depth;
    r.overlap = (uint32_t)s->overlap;
    if (s->world > 1 && s->tr) {
        uint32_t max_sweeps = 1, max_halo = 1;
        BE(loop_limits(&max_sweeps, &max_halo));
        const uint32_t depths[3] = {8, 6, 3};
        const uint32_t old_h = s->halo_depth;
        const int old_overlap = s->overlap;
        uint32_t best_t = 0xFFFFFFFFu;
        for (int i = 0; i < 3; i++) {
            const uint32_t h = depths[i];
            // (a depth the slabs or the loop cannot use would only repeat another one's measurement)
            if (h > std::min(max_halo, s->thinnest)) continue;
            if (i > 0 && max_sweeps < 3 && h % 2) continue;
            for (int mode = FLUID_SLAB_OVERLAP_NONE; mode <= FLUID_SLAB_OVERLAP_BOTH; mode++) {
                s->halo_depth = h;
                s->overlap = mode;
                int rc = pressure_step(s);  // untimed: the first loop of a schedule builds its plans
                if (rc == FLUID_OK) rc = s->from_backend(s->be->sync());
                uint32_t zero = 0;
                if (rc == FLUID_OK) rc = reduce_max(s, &zero, 1);  // everybody starts the timed loop together
                const auto t0 = std::chrono::steady_clock::now();
                if (rc == FLUID_OK) rc = pressure_step(s);
                if (rc == FLUID_OK) rc = s->from_backend(s->be->sync());
                if (rc) {
                    s->halo_depth = old_h;
                    s->overlap = old_overlap;
                    return rc;
                }
                const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                uint32_t t = (uint32_t)std::min(us, 4.0e9);
                TRY(reduce_max(s, &t, 1));
                r.times_us[3 * i + mode] = t;
                if (t < best_t) {
                    best_t = t;
                    r.halo_depth = h;
                    r.overlap = (uint32_t)mode;
                }
            }
        }
        s->halo_depth = r.halo_depth;
        s->overlap = (int)r.overlap;
        s->stats[FLUID_SLAB_STAT_EFFECTIVE_HALO] = 0;
    }
    if (result) *result = r;
    return FLUID_OK;
}

int fluid_slab_solve(fluid_slab* s, uint32_t iterations) {
    if (!s) return FLUID_ERR_INVALID_ARG;
    TRY(need_transport(s));
    return solve(s, iterations ? iterations : s->iterations);
}

int fluid_slab_exchange_image(fluid_slab* s, int image_id, uint32_t planes) {
    if (!s) return FLUID_ERR_INVALID_ARG;
    TRY(need_transport(s));
    if (image_id < 0 || image_id > FLUID_IMG_PARTICLE_DENSITIES_IMG)
        return s->fail(FLUID_ERR_INVALID_ARG, fmt("image %d has no ghost planes", image_id));
    if (planes > s->image_ghost)
        return s->fail(FLUID_ERR_INVALID_ARG, fmt("%u planes: an image has %u ghost planes per side here",
                                                  planes, s->image_ghost));
    return exchange_image(s, image_id, planes);
}

int fluid_slab_set_option(fluid_slab* s, int option, int64_t value) {
    if (!s) return FLUID_ERR_INVALID_ARG;
    switch (option) {
        case FLUID_SLAB_OPT_OVERLAP:
            if (value < FLUID_SLAB_OVERLAP_NONE || value > FLUID_SLAB_OVERLAP_BOTH)
                return s->fail(FLUID_ERR_INVALID_ARG, "overlap mode out of range");
            s->overlap = (int)value;
            return FLUID_OK;
        case FLUID_SLAB_OPT_HALO_DEPTH:
            if (value < 1 || value > FLUID_LOOP_MAX_HALO)
                return s->fail(FLUID_ERR_INVALID_ARG, "halo depth out of range");
            s->halo_depth = (uint32_t)value;
            return FLUID_OK;
        case FLUID_SLAB_OPT_SAMPLER_HALO:
            if (value < 1 || value > (int64_t)s->image_ghost)
                return s->fail(FLUID_ERR_INVALID_ARG, "sampler halo out of range");
            s->sampler_halo = (uint32_t)value;
            s->stats[FLUID_SLAB_STAT_SAMPLER_HALO] = s->sampler_halo;
            if (s->world > 1) BE(set_sampler_halo(s->sampler_halo));
            return FLUID_OK;
        default:
            return s->fail(FLUID_ERR_INVALID_ARG, fmt("unknown option %d", option));
    }
}

int fluid_slab_get_stat(fluid_slab* s, int stat, uint64_t* value) {
    if (!s) return FLUID_ERR_INVALID_ARG;
    if (!value || stat < 0 || stat >= FLUID_SLAB_STAT_COUNT)
        return s->fail(FLUID_ERR_INVALID_ARG, fmt("unknown statistic %d", stat));
    if (stat == FLUID_SLAB_STAT_EFFECTIVE_HALO && s->stats[stat] == 0) {
        uint32_t a = 1, b = 1;
        BE(loop_limits(&a, &b));
        s->stats[stat] = effective_halo(s, a, b);
    }
    *value = s->stats[stat];
    return FLUID_OK;
}

}  // extern "C"
