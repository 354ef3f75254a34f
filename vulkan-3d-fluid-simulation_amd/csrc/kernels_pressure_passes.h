// kernels_pressure_passes.h — the once-per-loop passes around the 12_solve_pressure sweeps (build the
// neighbour mask / b_i, import PRESSURES_1 into the working buffers, export the result), four cells per
// thread: cell types move as 4-byte words and pressures as 16-byte vectors, so these passes stream at
// the HBM rate instead of issuing one byte load per lane.  Same arithmetic as the one-cell-per-thread
// kernels in kernels_pressure.h (which remain the path for widths that are not a multiple of 4).
#pragma once

#include "pressure_common.h"
#include "quiet_bricks.h"

namespace fluid {

__device__ __forceinline__ uint32_t ld_types4(const uint8_t* __restrict__ t, const GridK& g, int x,
                                              int y, int lz) {
    // four cell types starting at x (x % 4 == 0, W % 4 == 0); a row outside the grid reads as 0.  The load itself is
    // unconditional (at the image's first word of the plane when the row does not exist): a load behind a bounds
    // test waits for the loads before it
    const bool in = (unsigned)x < (unsigned)g.W && (unsigned)y < (unsigned)g.H;
    const uint32_t w = *reinterpret_cast<const uint32_t*>(t + cidx(g, in ? x : 0, in ? y : 0, lz));
    return in ? w : 0u;
}
__device__ __forceinline__ uint32_t byte_at(uint32_t w, int i) { return (w >> (8 * i)) & 0xFFu; }

#define FLUID_V4_THREAD()                                          \
    const int x = 4 * (blockIdx.x * blockDim.x + threadIdx.x);     \
    const int y = blockIdx.y * blockDim.y + threadIdx.y;           \
    if (x >= g.W || y >= g.H) return;

// mask byte + b_i + activity bricks (k12_prepare of kernels_pressure.h, four cells per thread)
__global__ void k12_prepare_v4(const uint8_t* __restrict__ t, const float* __restrict__ div,
                               uint8_t* __restrict__ mask, float* __restrict__ rhs,
                               uint8_t* __restrict__ active, BrickK bk, GridK g, ParamsK p,
                               int do_mask, int do_rhs, const uint8_t* __restrict__ quiet,
                               uint32_t* __restrict__ x_extent) {
    FLUID_LEAVE_IF_QUIET_V4(quiet, bk)  // b_i-only passes of fluid_run_step (quiet_bricks.h)
    FLUID_V4_THREAD();
    FLUID_FOR_PLANES_OF_WORKGROUP() {
    const int64_t id = cidx(g, x, y, lz);
    if (do_rhs) {
        const float4 d = *reinterpret_cast<const float4*>(div + id);
        float4 b;
        b.x = ((d.x * p.rho) * p.dx) / p.dt;  // pressure.comp:54
        b.y = ((d.y * p.rho) * p.dx) / p.dt;
        b.z = ((d.z * p.rho) * p.dx) / p.dt;
        b.w = ((d.w * p.rho) * p.dx) / p.dt;
        *reinterpret_cast<float4*>(rhs + id) = b;
    }
    if (do_mask) {
        const uint32_t c = *reinterpret_cast<const uint32_t*>(t + id);
        const uint32_t yp = ld_types4(t, g, x, y + 1, lz), ym = ld_types4(t, g, x, y - 1, lz);
        const uint32_t zp = *reinterpret_cast<const uint32_t*>(t + cidx(g, x, y, lz + 1));
        const uint32_t zm = *reinterpret_cast<const uint32_t*>(t + cidx(g, x, y, lz - 1));
        const uint32_t left_ = t[id - (x > 0 ? 1 : 0)], left = x > 0 ? left_ : 0u;
        const uint32_t right_ = t[id + (x + 4 < g.W ? 4 : 3)], right = x + 4 < g.W ? right_ : 0u;
        uint32_t out = 0;
        bool any_water = false;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t xm = i == 0 ? left : byte_at(c, i == 0 ? 0 : i - 1);
            const uint32_t xp = i == 3 ? right : byte_at(c, i == 3 ? 3 : i + 1);
            uint32_t m = 0;
            m += xp != p.t_solid ? 1u : 0u;
            m += byte_at(yp, i) != p.t_solid ? 1u : 0u;
            m += byte_at(zp, i) != p.t_solid ? 1u : 0u;
            m += xm != p.t_solid ? 1u : 0u;
            m += byte_at(ym, i) != p.t_solid ? 1u : 0u;
            m += byte_at(zm, i) != p.t_solid ? 1u : 0u;
            const bool water = byte_at(c, i) == p.t_water;
            any_water = any_water || water;
            out |= (water ? m : MASK_DRY) << (8 * i);
        }
        *reinterpret_cast<uint32_t*>(mask + id) = out;
        // same value from every writer: a benign race (the array was zeroed before this launch)
        if (any_water) active[brick_index(bk, x / BRICK_X, y / BRICK_Y, lz / BRICK_Z)] = 1;
        // x extent of the water, to 4 cells: x_extent[0] = max (W - x_lo), x_extent[1] = max x_hi (both
        // zeroed before the launch); one pair of atomics per wavefront that holds water
        const unsigned long long wet = __builtin_amdgcn_ballot_w64(any_water);
        if (wet != 0ull) {
            const int lane = (int)(threadIdx.x & 63u);
            const int first = __builtin_ctzll(wet), last = 63 - __builtin_clzll(wet);
            if (lane == first) {
                // look before the atomic: after the first few wavefronts hardly any raises an extreme
                // (a stale look only costs a redundant atomic)
                const uint32_t lo = (uint32_t)(g.W - x), hi = (uint32_t)(x + 4 * (last - first) + 4);
                if (lo > __atomic_load_n(&x_extent[0], __ATOMIC_RELAXED)) atomicMax(&x_extent[0], lo);
                if (hi > __atomic_load_n(&x_extent[1], __ATOMIC_RELAXED)) atomicMax(&x_extent[1], hi);
            }
        }
    }
    }  // planes of the workgroup
}

// PRESSURES_1 -> working buffer w0 (water: its value, else the cell's constant) and, where given, the
// constants of the other two working buffers (their water cells are written by the sweeps)
__global__ void k12_import_v4(const uint8_t* __restrict__ t, const float* __restrict__ pimg,
                              float* __restrict__ w0, float* __restrict__ w1,
                              float* __restrict__ w2, GridK g, ParamsK p,
                              const uint8_t* __restrict__ quiet, BrickK bk) {
    FLUID_LEAVE_IF_QUIET_V4(quiet, bk)  // quiet_bricks.h: the constants are in place already
    FLUID_V4_THREAD();
    FLUID_FOR_PLANES_OF_WORKGROUP() {
    const int64_t id = cidx(g, x, y, lz);
    const uint32_t c = *reinterpret_cast<const uint32_t*>(t + id);
    const float4 v = *reinterpret_cast<const float4*>(pimg + id);
    const float bg[4] = {background_value(byte_at(c, 0), p), background_value(byte_at(c, 1), p),
                         background_value(byte_at(c, 2), p), background_value(byte_at(c, 3), p)};
    float4 o;
    o.x = byte_at(c, 0) == p.t_water ? v.x : bg[0];
    o.y = byte_at(c, 1) == p.t_water ? v.y : bg[1];
    o.z = byte_at(c, 2) == p.t_water ? v.z : bg[2];
    o.w = byte_at(c, 3) == p.t_water ? v.w : bg[3];
    *reinterpret_cast<float4*>(w0 + id) = o;
    const float4 b4 = make_float4(bg[0], bg[1], bg[2], bg[3]);
    if (w1) *reinterpret_cast<float4*>(w1 + id) = b4;
    if (w2) *reinterpret_cast<float4*>(w2 + id) = b4;
    }  // planes of the workgroup
}

// working buffers -> water cells of PRESSURES_1 (even iterate) and PRESSURES_2 (odd iterate)
__global__ void k12_export_v4(const uint8_t* __restrict__ t, const float* __restrict__ w_even,
                              const float* __restrict__ w_odd, float* __restrict__ p1,
                              float* __restrict__ p2, GridK g, ParamsK p,
                              const uint8_t* __restrict__ active, BrickK bk) {
    // `active` (optional): the activity bricks of the cell types being read — a brick without water has
    // nothing to export.  Needs the workgroup inside one brick (g.zl planes of one layer).
    if (active && active[brick_index(bk, (int)(blockIdx.x * 256u) / BRICK_X, (int)(blockIdx.y * 4u) / BRICK_Y,
                                     ((int)blockIdx.z * g.zl) / BRICK_Z)] == 0)
        return;
    FLUID_V4_THREAD();
    FLUID_FOR_PLANES_OF_WORKGROUP() {
    const int64_t id = cidx(g, x, y, lz);
    // (the iterates are loaded with the types, not behind the test of the types: one memory round trip, not two)
    const uint32_t c = *reinterpret_cast<const uint32_t*>(t + id);
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 ve = w_even ? *reinterpret_cast<const float4*>(w_even + id) : zero;
    const float4 vo = w_odd ? *reinterpret_cast<const float4*>(w_odd + id) : zero;
    const bool w[4] = {byte_at(c, 0) == p.t_water, byte_at(c, 1) == p.t_water,
                       byte_at(c, 2) == p.t_water, byte_at(c, 3) == p.t_water};
    if (!(w[0] || w[1] || w[2] || w[3])) continue;  // pressure.comp:69: non-water cells are never written
    const bool all = w[0] && w[1] && w[2] && w[3];
    auto put = [&](const float4& v, float* dst) {
        if (all) {
            *reinterpret_cast<float4*>(dst + id) = v;
        } else {
            if (w[0]) dst[id] = v.x;
            if (w[1]) dst[id + 1] = v.y;
            if (w[2]) dst[id + 2] = v.z;
            if (w[3]) dst[id + 3] = v.w;
        }
    };
    if (w_even) put(ve, p1);
    if (w_odd) put(vo, p2);
    }  // planes of the workgroup
}

// ---- convergence read-out (not in the reference, which never looks at its residual) -----------------
// Residual of the sweep's linear system at a WATER cell, in the sweep's own fp32 arithmetic:
//     s = b_i - sum over non-solid neighbours of (water ? P[nb] : p_air)      (pressure.comp:54-61)
//     r = s + aii * P[cell]            (the sweep stores P' = -s / aii, so r = 0 at its fixed point)
// Threads accumulate over many cells, wavefronts reduce with wave64 shuffles, workgroups through LDS, then
// one set of atomics per workgroup (a few thousand in all): max |r| as the bit pattern of a non-negative
// float (monotonic as an unsigned integer), sum of r^2 and the number of water cells in double / 64-bit.  NaN residuals are ignored by the maximum and poison the sum.
struct ResidualOut {
    unsigned int max_abs_bits;
    unsigned int pad;
    double sum_sq;
    unsigned long long water_cells;
};

// launch: 1-D grid of at most a few thousand workgroups of 256 threads; wavefront w of workgroup b walks
// the rows (y, z) numbered 4 b + w, 4 (b + gridDim.x) + w, ..., its lanes stride over x
__global__ void __launch_bounds__(256)
k12_residual(const uint8_t* __restrict__ t, const float* __restrict__ div,
             const float* __restrict__ pimg, GridK g, ParamsK p, ResidualOut* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float r_abs = 0.0f;
    double r_sq = 0.0;
    unsigned int wet = 0;
    const int64_t rows = (int64_t)g.H * g.Dl;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
        const int y = (int)(row % g.H), lz = (int)(row / g.H);
        for (int x = lane; x < g.W; x += 64) {
            const int64_t id = cidx(g, x, y, lz);
            if ((uint32_t)t[id] != p.t_water) continue;
            int aii = 0;
            float s = ((div[id] * p.rho) * p.dx) / p.dt;
            auto nb = [&](uint32_t ty, float q) {
                if (ty != p.t_solid) {
                    s = s - (ty == p.t_water ? q : p.p_air);
                    aii++;
                }
            };
            const uint32_t txp = type_at(t, g, x + 1, y, lz), typ = type_at(t, g, x, y + 1, lz);
            const uint32_t tzp = t[cidx(g, x, y, lz + 1)];
            const uint32_t txm = type_at(t, g, x - 1, y, lz), tym = type_at(t, g, x, y - 1, lz);
            const uint32_t tzm = t[cidx(g, x, y, lz - 1)];
            nb(txp, txp == p.t_water ? f32_at(pimg, g, x + 1, y, lz) : 0.f);
            nb(typ, typ == p.t_water ? f32_at(pimg, g, x, y + 1, lz) : 0.f);
            nb(tzp, tzp == p.t_water ? pimg[cidx(g, x, y, lz + 1)] : 0.f);
            nb(txm, txm == p.t_water ? f32_at(pimg, g, x - 1, y, lz) : 0.f);
            nb(tym, tym == p.t_water ? f32_at(pimg, g, x, y - 1, lz) : 0.f);
            nb(tzm, tzm == p.t_water ? pimg[cidx(g, x, y, lz - 1)] : 0.f);
            const float ap = (float)aii * pimg[id];
            const float r = s + ap;
            r_abs = fmaxf(r_abs, fabsf(r));
            r_sq += (double)r * (double)r;
            wet++;
        }
    }
    // wave64 butterfly: every lane ends with the wavefront's max / sums
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        r_abs = fmaxf(r_abs, __shfl_xor(r_abs, off, 64));
        r_sq += __shfl_xor(r_sq, off, 64);
        wet += __shfl_xor(wet, off, 64);
    }
    // the four wavefronts of the workgroup through LDS, then one set of atomics per workgroup
    __shared__ float s_abs[4];
    __shared__ double s_sq[4];
    __shared__ unsigned int s_wet[4];
    if (lane == 0) {
        s_abs[wave] = r_abs;
        s_sq[wave] = r_sq;
        s_wet[wave] = wet;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = fmaxf(fmaxf(s_abs[0], s_abs[1]), fmaxf(s_abs[2], s_abs[3]));
        const double q = ((s_sq[0] + s_sq[1]) + s_sq[2]) + s_sq[3];
        const unsigned int n = s_wet[0] + s_wet[1] + s_wet[2] + s_wet[3];
        if (n != 0) {
            atomicMax(&out->max_abs_bits, __float_as_uint(m));
            atomicAdd(&out->sum_sq, q);
            atomicAdd(&out->water_cells, (unsigned long long)n);
        }
    }
}

// ---- opt-in pressure solver (not in the reference): one colour of a red-black SOR
// iteration, in place on PRESSURES_1.  One thread per cell of the colour: x = 2 i + ((y + z + colour) & 1).
//     gs = -s / aii (s as in the Jacobi sweep, from the current image);  P = P + omega * (gs - P)
// Same-colour cells are not neighbours, so the update order inside a launch does not matter and the
// result is bit-identical to the oracle's sequential loop (oracle_12_sor_iteration).
__global__ void k12_sor_colour(const uint8_t* __restrict__ t, const float* __restrict__ div,
                               float* __restrict__ pr, GridK g, ParamsK p, float omega, int colour) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    const int lz = blockIdx.z;
    const int x = 2 * i + ((y + g.z0 + lz + colour) & 1);
    if (x >= g.W || y >= g.H) return;
    const int64_t id = cidx(g, x, y, lz);
    if ((uint32_t)t[id] != p.t_water) return;
    int aii = 0;
    float s = ((div[id] * p.rho) * p.dx) / p.dt;
    auto nb = [&](uint32_t ty, float q) {
        if (ty != p.t_solid) {
            s = s - (ty == p.t_water ? q : p.p_air);
            aii++;
        }
    };
    const uint32_t txp = type_at(t, g, x + 1, y, lz), typ = type_at(t, g, x, y + 1, lz);
    const uint32_t tzp = t[cidx(g, x, y, lz + 1)];
    const uint32_t txm = type_at(t, g, x - 1, y, lz), tym = type_at(t, g, x, y - 1, lz);
    const uint32_t tzm = t[cidx(g, x, y, lz - 1)];
    nb(txp, txp == p.t_water ? f32_at(pr, g, x + 1, y, lz) : 0.f);
    nb(typ, typ == p.t_water ? f32_at(pr, g, x, y + 1, lz) : 0.f);
    nb(tzp, tzp == p.t_water ? pr[cidx(g, x, y, lz + 1)] : 0.f);
    nb(txm, txm == p.t_water ? f32_at(pr, g, x - 1, y, lz) : 0.f);
    nb(tym, tym == p.t_water ? f32_at(pr, g, x, y - 1, lz) : 0.f);
    nb(tzm, tzm == p.t_water ? pr[cidx(g, x, y, lz - 1)] : 0.f);
    const float gs = -s / (float)aii;
    const float old = pr[id];
    const float d = gs - old;
    const float t2 = omega * d;
    pr[id] = old + t2;
}

// The same with four cells per thread (fluid_size.x % 4 == 0): the quad's two cells of the colour are
// updated, the other two are written back unchanged (they belong to this thread's quad, so nobody else
// writes them in this launch); pressures and divergences move as float4 rows, types as 4-byte words.
__global__ void k12_sor_colour_v4(const uint8_t* __restrict__ t, const float* __restrict__ div,
                                  float* __restrict__ pr, GridK g, ParamsK p, float omega, int colour) {
    FLUID_V4_THREAD();
    const int lz = blockIdx.z;
    const int64_t id = cidx(g, x, y, lz);
    const uint32_t c = *reinterpret_cast<const uint32_t*>(t + id);
    const uint32_t w4 = p.t_water * 0x01010101u;
    // any water cell of this colour in the quad?  (cheap exit for dry quads)
    bool any = false;
#pragma unroll
    for (int i = 0; i < 4; i++)
        any = any || (byte_at(c, i) == p.t_water && ((x + i + y + g.z0 + lz + colour) & 1) == 0);
    (void)w4;
    if (!any) return;
    const uint32_t typ = ld_types4(t, g, x, y + 1, lz), tym = ld_types4(t, g, x, y - 1, lz);
    const uint32_t tzp = *reinterpret_cast<const uint32_t*>(t + cidx(g, x, y, lz + 1));
    const uint32_t tzm = *reinterpret_cast<const uint32_t*>(t + cidx(g, x, y, lz - 1));
    const uint32_t tl_ = t[id - (x > 0 ? 1 : 0)], tr_ = t[id + (x + 4 < g.W ? 4 : 3)];
    const uint32_t tl = x > 0 ? tl_ : 0u, tr = x + 4 < g.W ? tr_ : 0u;
    auto prow = [&](int ny) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((unsigned)ny < (unsigned)g.H) v = *reinterpret_cast<const float4*>(pr + cidx(g, x, ny, lz));
        return v;
    };
    const float4 pc = *reinterpret_cast<const float4*>(pr + id);
    const float4 pyp = prow(y + 1), pym = prow(y - 1);
    const float4 pzp = *reinterpret_cast<const float4*>(pr + cidx(g, x, y, lz + 1));
    const float4 pzm = *reinterpret_cast<const float4*>(pr + cidx(g, x, y, lz - 1));
    const float pl = x > 0 ? pr[id - 1] : 0.0f, prr = x + 4 < g.W ? pr[id + 4] : 0.0f;
    const float4 d4 = *reinterpret_cast<const float4*>(div + id);
    const float cc[4] = {pc.x, pc.y, pc.z, pc.w}, dd[4] = {d4.x, d4.y, d4.z, d4.w};
    const float a_yp[4] = {pyp.x, pyp.y, pyp.z, pyp.w}, a_ym[4] = {pym.x, pym.y, pym.z, pym.w};
    const float a_zp[4] = {pzp.x, pzp.y, pzp.z, pzp.w}, a_zm[4] = {pzm.x, pzm.y, pzm.z, pzm.w};
    float out[4] = {pc.x, pc.y, pc.z, pc.w};
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (byte_at(c, i) != p.t_water || ((x + i + y + g.z0 + lz + colour) & 1) != 0) continue;
        int aii = 0;
        float s = ((dd[i] * p.rho) * p.dx) / p.dt;
        auto nb = [&](uint32_t ty, float q) {
            if (ty != p.t_solid) {
                s = s - (ty == p.t_water ? q : p.p_air);
                aii++;
            }
        };
        nb(i == 3 ? tr : byte_at(c, i == 3 ? 3 : i + 1), i == 3 ? prr : cc[i == 3 ? 3 : i + 1]);  // +x
        nb(byte_at(typ, i), a_yp[i]);                                                             // +y
        nb(byte_at(tzp, i), a_zp[i]);                                                             // +z
        nb(i == 0 ? tl : byte_at(c, i == 0 ? 0 : i - 1), i == 0 ? pl : cc[i == 0 ? 0 : i - 1]);    // -x
        nb(byte_at(tym, i), a_ym[i]);                                                             // -y
        nb(byte_at(tzm, i), a_zm[i]);                                                             // -z
        const float gs = -s / (float)aii;
        const float d = gs - cc[i];
        const float t2 = omega * d;
        out[i] = cc[i] + t2;
    }
    *reinterpret_cast<float4*>(pr + id) = make_float4(out[0], out[1], out[2], out[3]);
}

}  // namespace fluid
