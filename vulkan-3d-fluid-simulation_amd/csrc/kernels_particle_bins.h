// kernels_particle_bins.h — particles kept sorted by "bin" (16 x 4 x 16 cells) on whole-grid contexts, and
// the two particle passes of the step on that order.
//
// Why: 01_update_densities (update_densities.comp:29-36) is a scatter of one atomic per particle and
// 14_particles (particles.comp:45-51) a gather of 24 taps per particle.  In slot order both go through
// global memory: on a full 512^3 tank 01 is bound by cross-XCD atomics (two per cell) and 14 by the texture
// addresser.  Particle order enters no result — the counts are integer sums, every particle is advected on
// its own — so the engine is free to store the particles in any order as long as it remembers which slot
// each one belongs to (`slot_of`): uploads, downloads and 00_init_particles speak slot order, everything
// in between runs on the sorted storage.  With all particles of a bin contiguous,
//   * 01 is a histogram of the bin in LDS (ds_add) and one plain 4-byte store per cell — no global atomics;
//   * 14 stages the bin's velocities + 1 halo cell in LDS once and serves the 24 taps of each of its
//     particles (about 8 k in water) from there.
// Particles move, so between two sorts a bin's segment also holds "strays" that have left the bin: 01
// lists them and a second small kernel adds them with global atomics (after the plain stores: stream
// order), 14 takes the taps of a sample that leaves the tile from global memory, exactly as
// k07_advect_tiled does.  The engine sorts again when the strays exceed a few per cent.
//
// The sort is a counting sort over bins in two passes of the particle buffer (histogram; scatter), each
// block aggregating its 4096 particles per bin in an LDS table first so that sorted or spawn-ordered input
// costs a handful of global atomics per block.  The order inside a bin is whatever the atomics gave: it
// does not matter to the kernels above, and the slot of every particle travels with it.
#pragma once

#include "kernels_sampler.h"

namespace fluid {

// 16 x 4 x 16 cells.  (Twice as high — strays are particles that cross a bin face, and gravity makes y the
// direction they cross fastest — was tried: 14's larger tile costs the full tank 10 %, and the dam break's
// collapse then sorts every few steps at a small loss instead of giving up.)
constexpr int PBIN_X = 16, PBIN_Y = 4, PBIN_Z = 16;
constexpr int PBIN_CELLS = PBIN_X * PBIN_Y * PBIN_Z;
constexpr int PBIN_BRICKS_Y = PBIN_Y / BRICK_Y;
static_assert(BRICK_X % PBIN_X == 0 && PBIN_Y % BRICK_Y == 0 && BRICK_Z == PBIN_Z,
              "a bin is a whole number of bricks high and lies in one brick otherwise");

struct PBinK {
    int nx, ny, nz;
    uint32_t bins;  // nx * ny * nz; key `bins` = "counts nowhere" (inactive, outside the grid, NaN)
};

// the bin a particle counts in: the cell of update_densities.comp:35 (ivec3 truncation; dropped outside)
// the cell a particle counts towards, as this context addresses it (cz = LOCAL plane), or false: inactive,
// outside the grid, a tombstone, or — Z-slab contexts — in a plane of another slab
__device__ __forceinline__ bool particle_cell(const float4& q, const GridK& g, const ParamsK& p, int& cx,
                                              int& cy, int& cz) {
    if (!(q.w == p.active_w && trunc_index(q.x, g.W, cx) && trunc_index(q.y, g.H, cy) &&
          trunc_index(q.z, g.Dg, cz)))
        return false;
    cz -= g.z0;
    return (unsigned)cz < (unsigned)g.Dl;
}
__device__ __forceinline__ uint32_t particle_bin(const float4& q, const GridK& g, const ParamsK& p,
                                                 const PBinK& b) {
    int cx, cy, cz;
    if (!particle_cell(q, g, p, cx, cy, cz)) return b.bins;
    return (uint32_t)(cx / PBIN_X + b.nx * (cy / PBIN_Y + b.ny * (cz / PBIN_Z)));
}

// ---- counting sort ------------------------------------------------------------------------------------
constexpr int PSORT_THREADS = 256, PSORT_PER_THREAD = 16, PSORT_TABLE = 4096;
constexpr uint32_t PSORT_EMPTY = 0xFFFFFFFFu;

// insert `key` into the block's LDS table; returns the entry, or -1 when the table is crowded
__device__ __forceinline__ int psort_insert(uint32_t* keys, uint32_t key) {
    uint32_t h = (key * 2654435761u) >> 20;
    for (int probe = 0; probe < 32; probe++) {
        const uint32_t old = atomicCAS(&keys[h], PSORT_EMPTY, key);
        if (old == PSORT_EMPTY || old == key) return (int)h;
        h = (h + 1) & (PSORT_TABLE - 1);
    }
    return -1;
}
// The lanes of a wavefront mostly hold particles of the same bin (the storage is sorted already, or in spawn
// order): one lane per distinct key does the table work for all of them.  Peels the wavefront key by key
// (`active`: lanes that take part); returns this lane's table entry (or -1), its rank among the lanes of
// its key and their number.  leader = the lane that should add `peers` to the entry's count.
struct WaveKey {
    int entry;
    uint32_t rank, peers;
    bool leader;
};
__device__ __forceinline__ WaveKey psort_wave_insert(uint32_t* keys, uint32_t key, bool active) {
    WaveKey r;
    r.entry = -1;
    r.rank = r.peers = 0u;
    r.leader = false;
    const int lane = (int)(threadIdx.x & 63u);
    unsigned long long todo = __builtin_amdgcn_ballot_w64(active);
    while (todo) {  // wave-uniform
        const int first = __builtin_ctzll(todo);
        const uint32_t k0 = (uint32_t)__builtin_amdgcn_readlane((int)key, first);
        const unsigned long long same = __builtin_amdgcn_ballot_w64(active && key == k0) & todo;
        int e = 0;
        if (lane == first) e = psort_insert(keys, k0);
        e = __builtin_amdgcn_readlane(e, first);
        if ((same >> lane) & 1ull) {
            r.entry = e;
            r.rank = (uint32_t)__builtin_popcountll(same & ((1ull << lane) - 1ull));
            r.peers = (uint32_t)__builtin_popcountll(same);
            r.leader = lane == first;
        }
        todo &= ~same;
    }
    return r;
}

// pass 1: particles per bin
__global__ void __launch_bounds__(PSORT_THREADS)
k_pbin_histogram(const float4* __restrict__ particles, uint64_t capacity, GridK g, ParamsK p, PBinK b,
                 uint32_t* __restrict__ bin_count, bool drop_tombstones) {
    __shared__ uint32_t keys[PSORT_TABLE];
    __shared__ uint32_t counts[PSORT_TABLE];
    for (int i = threadIdx.x; i < PSORT_TABLE; i += PSORT_THREADS) {
        keys[i] = PSORT_EMPTY;
        counts[i] = 0u;
    }
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * (PSORT_THREADS * PSORT_PER_THREAD);
#pragma unroll 4
    for (int k = 0; k < PSORT_PER_THREAD; k++) {
        const uint64_t i = base + (uint64_t)k * PSORT_THREADS + threadIdx.x;
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < capacity) q = particles[i];
        const bool in = i < capacity && !(drop_tombstones && is_tombstone(q));  // (a slab's holes are dropped by the sort)
        const uint32_t key = in ? particle_bin(q, g, p, b) : 0u;
        const WaveKey w = psort_wave_insert(keys, key, in);
        if (w.leader) {
            if (w.entry >= 0)
                atomicAdd(&counts[w.entry], w.peers);
            else
                atomicAdd(&bin_count[key], w.peers);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < PSORT_TABLE; i += PSORT_THREADS)
        if (keys[i] != PSORT_EMPTY) atomicAdd(&bin_count[keys[i]], counts[i]);
}

// exclusive prefix sum of bin_count[0 .. n) -> bin_start[0 .. n] and the scatter cursors (one workgroup:
// n is a few hundred thousand at most)
__global__ void __launch_bounds__(1024)
k_pbin_scan(const uint32_t* __restrict__ bin_count, uint32_t n, uint32_t* __restrict__ bin_start,
            uint32_t* __restrict__ cursor) {
    __shared__ uint32_t part[1024];
    const uint32_t per = (n + 1023u) / 1024u;
    const uint32_t lo = min(n, threadIdx.x * per), hi = min(n, lo + per);
    uint32_t s = 0;
    for (uint32_t i = lo; i < hi; i++) s += bin_count[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {  // Hillis-Steele, inclusive
        const uint32_t v = threadIdx.x >= (unsigned)off ? part[threadIdx.x - off] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - s;
    for (uint32_t i = lo; i < hi; i++) {
        bin_start[i] = run;
        cursor[i] = run;
        run += bin_count[i];
    }
    if (threadIdx.x == 1023) bin_start[n] = part[1023];
}

// (eight particles per thread here: the sixteen of pass 1 held in registers leave three wavefronts per SIMD)
constexpr int PSCATTER_PER_THREAD = 8;
// pass 2: every particle to its bin's segment of `out`, its slot with it.  slot_in == nullptr: the input
// is in slot order.
__global__ void __launch_bounds__(PSORT_THREADS)
k_pbin_scatter(const float4* __restrict__ particles, const uint32_t* __restrict__ slot_in, uint64_t capacity,
               GridK g, ParamsK p, PBinK b, uint32_t* __restrict__ cursor, float4* __restrict__ out,
               uint32_t* __restrict__ slot_out, bool drop_tombstones) {
    __shared__ uint32_t keys[PSORT_TABLE];
    __shared__ uint32_t counts[PSORT_TABLE];  // particles of the block per entry, then the entry's base
    for (int i = threadIdx.x; i < PSORT_TABLE; i += PSORT_THREADS) {
        keys[i] = PSORT_EMPTY;
        counts[i] = 0u;
    }
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * (PSORT_THREADS * PSCATTER_PER_THREAD);
    float4 q[PSCATTER_PER_THREAD];
    int entry[PSCATTER_PER_THREAD];      // table entry, -1: `rank` is the final position already, -2: no particle
    uint32_t rank[PSCATTER_PER_THREAD];
#pragma unroll
    for (int k = 0; k < PSCATTER_PER_THREAD; k++) {
        const uint64_t i = base + (uint64_t)k * PSORT_THREADS + threadIdx.x;
        q[k] = i < capacity ? particles[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        const bool in = i < capacity && !(drop_tombstones && is_tombstone(q[k]));
        const uint32_t key = in ? particle_bin(q[k], g, p, b) : 0u;
        const WaveKey w = psort_wave_insert(keys, key, in);
        // one reservation per wavefront and key: the leader's old count is where its lanes start
        uint32_t first = 0u;
        if (w.leader) first = w.entry >= 0 ? atomicAdd(&counts[w.entry], w.peers) : atomicAdd(&cursor[key], w.peers);
        const unsigned long long leaders = __builtin_amdgcn_ballot_w64(w.leader);
        // every lane reads its leader's `first`: the leader is the lowest lane of its key
        uint32_t mine = 0u;
        for (unsigned long long l = leaders; l; l &= l - 1) {  // wave-uniform: one round per distinct key
            const int ll = __builtin_ctzll(l);
            const uint32_t f = (uint32_t)__builtin_amdgcn_readlane((int)first, ll);
            const uint32_t kk = (uint32_t)__builtin_amdgcn_readlane((int)key, ll);
            if (in && key == kk) mine = f;
        }
        entry[k] = in ? w.entry : -2;
        rank[k] = mine + w.rank;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < PSORT_TABLE; i += PSORT_THREADS)
        if (keys[i] != PSORT_EMPTY) counts[i] = atomicAdd(&cursor[keys[i]], counts[i]);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PSCATTER_PER_THREAD; k++) {
        if (entry[k] == -2) continue;
        const uint64_t i = base + (uint64_t)k * PSORT_THREADS + threadIdx.x;
        const uint32_t dst = entry[k] >= 0 ? counts[entry[k]] + rank[k] : rank[k];
        out[dst] = q[k];
        slot_out[dst] = slot_in ? slot_in[i] : (uint32_t)i;
    }
}

// sorted storage -> slot order (downloads)
__global__ void k_pbin_to_slot_order(const float4* __restrict__ sorted, const uint32_t* __restrict__ slot_of,
                                     uint64_t capacity, float4* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < capacity) out[slot_of[i]] = sorted[i];
}

// ---- 01_update_densities on sorted particles ----------------------------------------------------------
// One workgroup per bin segment (grid-stride).  In-bin particles: LDS histogram, then one 4-byte store per
// cell that counted something (ADD: an atomic add instead — the image was not cleared just before, so what
// it holds has to stay, update_densities.comp:35 adds).  Strays (and the whole last segment, whose
// particles counted nowhere when they were sorted): cell index appended to `strays`, added by
// k01_binned_strays behind this kernel.  stray_count[0] = entries, [1] = particles found outside their bin.
constexpr int K01_STRAY_BUF = 1024;  // stray keys a workgroup collects in LDS per bin before it reserves list space
template <bool ADD>
__global__ void __launch_bounds__(256)
k01_binned(const float4* __restrict__ particles, const uint32_t* __restrict__ bin_start, PBinK b,
           uint32_t* __restrict__ dens, GridK g, ParamsK p, uint8_t* __restrict__ particle_bricks, BrickK bk,
           uint32_t* __restrict__ strays, uint32_t* __restrict__ stray_count) {
    __shared__ uint32_t hist[PBIN_CELLS];
    __shared__ uint32_t sbuf[K01_STRAY_BUF];
    __shared__ uint32_t sn, sbase;  // keys in sbuf; where they go in the list
    __shared__ int any[PBIN_BRICKS_Y];
    if (threadIdx.x == 0) sn = 0u;
    __syncthreads();
    // one reservation in the global list per flush, not per wavefront: the list's one counter is what a
    // scene full of strays would otherwise queue up on.  Called by all threads.
    auto flush = [&](bool real) {
        __syncthreads();
        const uint32_t n = min(sn, (uint32_t)K01_STRAY_BUF);  // (what did not fit went to the list directly)
        if (n) {
            if (threadIdx.x == 0) {
                sbase = atomicAdd(&stray_count[0], n);
                if (real) atomicAdd(&stray_count[1], n);
            }
            __syncthreads();
            for (uint32_t i = threadIdx.x; i < n; i += 256) strays[sbase + i] = sbuf[i];
            __syncthreads();
            if (threadIdx.x == 0) sn = 0u;
            __syncthreads();
        }
    };
    for (uint32_t bin = blockIdx.x; bin <= b.bins; bin += gridDim.x) {
        const uint32_t s = bin_start[bin], e = bin_start[bin + 1];
        if (s == e) continue;  // workgroup-uniform
        const bool real = bin < b.bins;
        const int bx = (int)(bin % (uint32_t)b.nx), byz = (int)(bin / (uint32_t)b.nx);
        const int x0 = bx * PBIN_X, y0 = (byz % b.ny) * PBIN_Y, z0 = (byz / b.ny) * PBIN_Z;
        for (int i = threadIdx.x; i < PBIN_CELLS; i += 256) hist[i] = 0u;
        if (threadIdx.x < PBIN_BRICKS_Y) any[threadIdx.x] = 0;
        __syncthreads();
        // four particles per thread and round, their loads in flight together (unconditional: a guarded load waits
        // for the one before it); a stray that finds the workgroup's buffer full goes to the list by itself, so
        // the loop has no barrier
        for (uint32_t i0 = s; i0 < e; i0 += 4u * 256u) {
            float4 q[4];
#pragma unroll
            for (int k = 0; k < 4; k++) q[k] = particles[min(i0 + (uint32_t)k * 256u + threadIdx.x, e - 1u)];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t i = i0 + (uint32_t)k * 256u + threadIdx.x;
                int cx, cy, cz;
                if (i < e && particle_cell(q[k], g, p, cx, cy, cz)) {
                    const int lx = cx - x0, ly = cy - y0, lz = cz - z0;
                    if (real && (unsigned)lx < (unsigned)PBIN_X && (unsigned)ly < (unsigned)PBIN_Y &&
                        (unsigned)lz < (unsigned)PBIN_Z) {
                        atomicAdd(&hist[lx + PBIN_X * (ly + PBIN_Y * lz)], 1u);
                    } else {
                        const uint32_t key = (uint32_t)cidx(g, cx, cy, cz), slot = atomicAdd(&sn, 1u);
                        if (slot < (uint32_t)K01_STRAY_BUF) {
                            sbuf[slot] = key;
                        } else {
                            strays[atomicAdd(&stray_count[0], 1u)] = key;
                            if (real) atomicAdd(&stray_count[1], 1u);
                        }
                    }
                }
            }
        }
        flush(real);  // also the barrier in front of the histogram's readers
        if (real) {
            for (int c = threadIdx.x; c < PBIN_CELLS; c += 256) {
                const uint32_t n = hist[c];
                if (n == 0u) continue;
                const int ly = (c / PBIN_X) % PBIN_Y;
                const int x = x0 + (c % PBIN_X), y = y0 + ly, z = z0 + c / (PBIN_X * PBIN_Y);
                if (ADD)
                    atomicAdd(&dens[cidx(g, x, y, z)], n);
                else
                    dens[cidx(g, x, y, z)] = n;
                any[ly / BRICK_Y] = 1;
            }
            __syncthreads();
            if (threadIdx.x < PBIN_BRICKS_Y && any[threadIdx.x] && particle_bricks)
                particle_bricks[brick_index(bk, x0 / BRICK_X, y0 / BRICK_Y + (int)threadIdx.x, z0 / BRICK_Z)] = 1;
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256)
k01_binned_strays(const uint32_t* __restrict__ strays, const uint32_t* __restrict__ stray_count,
                  uint32_t* __restrict__ dens, GridK g, uint8_t* __restrict__ particle_bricks, BrickK bk) {
    const uint32_t n = stray_count[0];
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const uint32_t key = strays[i];
        atomicAdd(&dens[key], 1u);
        if (particle_bricks) {
            const int x = (int)(key % (uint32_t)g.W), yz = (int)(key / (uint32_t)g.W);
            particle_bricks[brick_index(bk, x / BRICK_X, (yz % g.H) / BRICK_Y, (yz / g.H) / BRICK_Z)] = 1;
        }
    }
}

// ---- 14_particles on sorted particles -----------------------------------------------------------------
// Velocity tile of a bin: its cells and one cell around them (a particle in cell c takes its taps from
// c-1 .. c+1, axis_taps), one array per component.  Cells outside the image are never addressed: taps are
// clamped into the image before the lookup.
constexpr int PTILE_W = PBIN_X + 2, PTILE_H = PBIN_Y + 2, PTILE_D = PBIN_Z + 2;
constexpr int PTILE_CELLS = PTILE_W * PTILE_H * PTILE_D;
struct BinTile {
    static constexpr bool enabled = true;
    static constexpr int W = PTILE_W, H = PTILE_H;
    const FLUID_LDS_F float* comp[3];
    int x_org, y_org, z_org;
    __device__ __forceinline__ int slot(int z) const {
        const int d = z - z_org;
        return (unsigned)d < (unsigned)PTILE_D ? d : -1;
    }
};

// The three samples of particles.comp:46-47 at one position, taps from the bin's tile.  Same arithmetic
// as three sample_comp calls (axis_taps per axis and stagger, the lerps in the same order), organised
// around what they share: along each axis the component staggered on it uses one pair of taps and the other
// two use another, so there are six axis_taps instead of nine and ONE in-tile test for all 24 taps (the
// unstaggered pair starts lowest, the staggered pair ends highest: axis_taps is monotonic).  The kernel is
// bound by VALU issue (counters: 93 % busy), so independent fp32 operations of the same kind go in pairs
// through the packed instructions (v_pk_mul_f32 / v_pk_add_f32: two IEEE operations per issue, each rounded
// like its scalar form; no contraction): both staggers of an axis, both x-lerps of a row pair, both y-lerps.
// Returns false — nothing sampled — when a tap lies outside the tile.
typedef float pk2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pk2 lerp2(pk2 A, pk2 B, pk2 a) { return ((pk2)(1.0f) - a) * A + a * B; }
// The lower tap of axis_taps for coord + 0.5f ([0], .x) and coord + 0.0f ([1], .y) at once, NOT clamped into the
// image: lo in [-1, n].  The upper tap is lo + 1.  The bin's tile repeats the image's edge cells once beyond the
// edge (k14_binned stages it so), so reading tile cells lo and lo + 1 gives what the clamped taps of axis_taps
// give — and the eight taps of a sample sit at fixed distances from the first one.
__device__ __forceinline__ void axis_lo2(float coord, const AxisN& ax, int (&lo)[2], pk2& a) {
    const pk2 c = (pk2)(coord) + (pk2){0.5f, 0.0f};
    pk2 u = c;  // a power of two: c / n * n gives c back where it matters (axis_taps)
    if (!ax.pow2) {
        const pk2 s = (pk2){c.x / ax.fn, c.y / ax.fn};
        u = s * (pk2)(ax.fn);
    }
    const pk2 ub = u - (pk2)(0.5f);
    const pk2 fl = (pk2){floorf(ub.x), floorf(ub.y)};
    a = ub - fl;
    lo[0] = (int)__builtin_amdgcn_fmed3f(fl.x, -1.0f, ax.fn);  // NaN -> -1 (axis_taps)
    lo[1] = (int)__builtin_amdgcn_fmed3f(fl.y, -1.0f, ax.fn);
}
__device__ __forceinline__ bool tile_velocity(const BinTile& t, const GridK& g, const Axes& axes, float px,
                                              float py, float pz, float& vx, float& vy, float& vz) {
    int xl[2], yl[2], zl[2];  // [0] staggered (+0.5), [1] not
    pk2 ax, ay, az;
    axis_lo2(px, axes.x, xl, ax);
    axis_lo2(py, axes.y, yl, ay);
    axis_lo2(pz, axes.z, zl, az);
    const int xo = t.x_org, yo = t.y_org, zo = t.z_org + g.z0;  // z taps are global planes
    // all 24 taps inside the tile: the unstaggered pair starts lowest, the staggered pair ends highest
    if (!((unsigned)(xl[1] - xo) < (unsigned)PTILE_W && (unsigned)(xl[0] + 1 - xo) < (unsigned)PTILE_W &&
          (unsigned)(yl[1] - yo) < (unsigned)PTILE_H && (unsigned)(yl[0] + 1 - yo) < (unsigned)PTILE_H &&
          (unsigned)(zl[1] - zo) < (unsigned)PTILE_D && (unsigned)(zl[0] + 1 - zo) < (unsigned)PTILE_D))
        return false;
    // a Z slab: a tap whose plane (clamped into the grid) lies beyond the ghost planes that hold the neighbouring
    // slab's current data is the global sampler's business (it raises the halo-violation flag)
    if (max(zl[1], 0) - g.z0 < -g.sg_lo || min(zl[0] + 1, g.Dg - 1) - g.z0 >= g.Dl + g.sg_hi) return false;
    // byte offset of the first tap (lowest x, y, z) inside a component's array, axis by axis and stagger
    int X[2], Y[2], Z[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        X[k] = 4 * (xl[k] - xo);
        Y[k] = imul24(4 * PTILE_W, yl[k] - yo);
        Z[k] = imul24(4 * PTILE_W * PTILE_H, zl[k] - zo);
    }
    // component `c`, staggered on the axes whose s* flag is 0 ([0] = staggered taps)
    auto tri = [&](const FLUID_LDS_F float* c, int sx, int sy, int sz) {
        const FLUID_LDS_F float* q = reinterpret_cast<const FLUID_LDS_F float*>(
            reinterpret_cast<const FLUID_LDS_F char*>(c) + (X[sx] + Y[sy] + Z[sz]));
        constexpr int DY = PTILE_W, DZ = PTILE_W * PTILE_H;
        const float wx = sx ? ax.y : ax.x, wy = sy ? ay.y : ay.x;
        const pk2 cz0 = lerp2((pk2){q[0], q[DY]}, (pk2){q[1], q[DY + 1]}, (pk2)(wx));                      // c00, c10
        const pk2 cz1 = lerp2((pk2){q[DZ], q[DZ + DY]}, (pk2){q[DZ + 1], q[DZ + DY + 1]}, (pk2)(wx));      // c01, c11
        return lerp2((pk2){cz0.x, cz1.x}, (pk2){cz0.y, cz1.y}, (pk2)(wy));  // c0, c1
    };
    const pk2 cx = tri(t.comp[0], 0, 1, 1), cy = tri(t.comp[1], 1, 0, 1), cz = tri(t.comp[2], 1, 1, 0);
    const pk2 vxy = lerp2((pk2){cx.x, cy.x}, (pk2){cx.y, cy.y}, (pk2)(az.y));  // x and y: z not staggered
    vx = vxy.x;
    vy = vxy.y;
    vz = lerp1(cz.x, cz.y, az.x);
    return true;
}

// One workgroup per bin segment (grid-stride), or per `parts`-th of one when there are few full bins.
__global__ void __launch_bounds__(256)
k14_binned(const float4* __restrict__ v1, float4* __restrict__ particles,
           const uint32_t* __restrict__ bin_start, PBinK b, GridK g, ParamsK p,
           uint32_t* __restrict__ violation, uint32_t parts) {
    __shared__ float tile[3][PTILE_CELLS];
    const Axes axes = make_axes(g);
    const uint32_t work = (b.bins + 1u) * parts;
    for (uint32_t w = blockIdx.x; w < work; w += gridDim.x) {
        const uint32_t bin = w / parts, part = w - bin * parts;
        uint32_t s = bin_start[bin], e = bin_start[bin + 1];
        if (s == e) continue;  // workgroup-uniform
        if (parts > 1u) {
            const uint32_t len = (e - s + parts - 1u) / parts;
            s = min(e, s + part * len);
            e = min(e, s + len);
            if (s == e) continue;
        }
        if (bin == b.bins) {   // counted nowhere when sorted: wherever they are now, the taps come from memory
            for (uint32_t i = s + threadIdx.x; i < e; i += 256) {
                float4 q = particles[i];
                if (q.w == p.active_w) {
                    const float vx = sample_comp<0>(v1, g, axes, q.x, q.y, q.z, violation);
                    const float vy = sample_comp<1>(v1, g, axes, q.x, q.y, q.z, violation);
                    const float vz = sample_comp<2>(v1, g, axes, q.x, q.y, q.z, violation);
                    q.x = q.x + vx * p.dt;
                    q.y = q.y + vy * p.dt;
                    q.z = q.z + vz * p.dt;
                    particles[i] = q;
                }
            }
            continue;
        }
        const int bx = (int)(bin % (uint32_t)b.nx), byz = (int)(bin / (uint32_t)b.nx);
        BinTile t;
        t.x_org = bx * PBIN_X - 1;
        t.y_org = (byz % b.ny) * PBIN_Y - 1;
        t.z_org = (byz / b.ny) * PBIN_Z - 1;
        __syncthreads();  // the previous bin's taps are done with the tile
        for (int c = threadIdx.x; c < PTILE_CELLS; c += 256) {
            // a tile cell beyond an edge of the grid repeats the edge cell (tile_velocity reads unclamped taps)
            const int x = min(max(t.x_org + c % PTILE_W, 0), g.W - 1);
            const int y = min(max(t.y_org + (c / PTILE_W) % PTILE_H, 0), g.H - 1);
            const int z = min(max(g.z0 + t.z_org + c / (PTILE_W * PTILE_H), 0), g.Dg - 1) - g.z0;
            // (local planes; a slab's ghost planes are loaded too — whether a tap may use them is tile_velocity's test)
            if (z >= -IMG_GHOST && z < g.Dl + IMG_GHOST) {
                const float4 v = v1[cidx(g, x, y, z)];
                tile[0][c] = v.x;
                tile[1][c] = v.y;
                tile[2][c] = v.z;
            }
        }
        __syncthreads();
        t.comp[0] = (const FLUID_LDS_F float*)tile[0];
        t.comp[1] = (const FLUID_LDS_F float*)tile[1];
        t.comp[2] = (const FLUID_LDS_F float*)tile[2];
        for (uint32_t i = s + threadIdx.x; i < e; i += 256) {
            float4 q = particles[i];
            if (q.w == p.active_w) {  // particles.comp:48
                float vx, vy, vz;
                if (!tile_velocity(t, g, axes, q.x, q.y, q.z, vx, vy, vz)) {  // a stray
                    vx = sample_comp<0>(v1, g, axes, q.x, q.y, q.z, violation);
                    vy = sample_comp<1>(v1, g, axes, q.x, q.y, q.z, violation);
                    vz = sample_comp<2>(v1, g, axes, q.x, q.y, q.z, violation);
                }
                q.x = q.x + vx * p.dt;  // :50
                q.y = q.y + vy * p.dt;
                q.z = q.z + vz * p.dt;
                particles[i] = q;
            }
        }
    }
}

}  // namespace fluid
