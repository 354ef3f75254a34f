/*
 * fluid_oracle.c — single-threaded fp32 restatement of shaders_fluid/00…14 (see fluid_oracle.h:
 * TEST INFRASTRUCTURE ONLY, PARITY UNPINNED).  Citations are /root/reference paths.
 *
 * Arithmetic rules: every float expression is evaluated in fp32 in the order the GLSL source
 * writes it, one rounding per operation; compile with -ffp-contract=off and without -ffast-math.
 * Out-of-bounds image loads return 0 / stores are dropped.
 */
#include "fluid_oracle.h"

#include <math.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------- */
typedef struct grid {
    int W, H, D;
} grid;

static grid grid_of(const fluid_params* p) {
    grid g = {(int)p->fluid_size[0], (int)p->fluid_size[1], (int)p->fluid_size[2]};
    return g;
}
static int in_bounds(grid g, int x, int y, int z) {
    return x >= 0 && x < g.W && y >= 0 && y < g.H && z >= 0 && z < g.D;
}
static uint64_t cell(grid g, int x, int y, int z) {
    return (uint64_t)x + (uint64_t)g.W * ((uint64_t)y + (uint64_t)g.H * (uint64_t)z);
}
/* imageLoad of an R8_UINT image: OOB -> 0 */
static uint32_t type_at(grid g, const uint8_t* t, int x, int y, int z) {
    return in_bounds(g, x, y, z) ? t[cell(g, x, y, z)] : 0u;
}
/* imageLoad of an R32F image: OOB -> 0 */
static float f32_at(grid g, const float* a, int x, int y, int z) {
    return in_bounds(g, x, y, z) ? a[cell(g, x, y, z)] : 0.0f;
}
/* imageLoad of one channel of an RGBA32F image: OOB -> 0 */
static float vel_at(grid g, const float* v, int x, int y, int z, int comp) {
    return in_bounds(g, x, y, z) ? v[4 * cell(g, x, y, z) + comp] : 0.0f;
}

void oracle_fill_f32(float* dst, uint64_t count, float value) {
    for (uint64_t i = 0; i < count; i++) dst[i] = value;
}
void oracle_fill_u32(uint32_t* dst, uint64_t count, uint32_t value) {
    for (uint64_t i = 0; i < count; i++) dst[i] = value;
}
void oracle_fill_u8(uint8_t* dst, uint64_t count, uint8_t value) { memset(dst, value, count); }

/* ---- 00_init_particles/init_particles.comp:27-50 ------------------------------------------- */
void oracle_00_init_particles(const fluid_params* p, float* particles, uint64_t capacity) {
    const uint32_t rx = p->particle_spawn_cube_resolution[0];
    const uint32_t ry = p->particle_spawn_cube_resolution[1];
    const uint32_t rz = p->particle_spawn_cube_resolution[2];
    for (uint64_t i = 0; i < capacity; i++) {
        float* o = particles + 4 * i;
        if (i < (uint64_t)p->particle_spawn_cube_volume) { /* :39 */
            /* getPos :27-34 */
            uint32_t n = (uint32_t)i;
            uint32_t x = n % rx;
            n /= rx;
            uint32_t y = n % ry;
            n /= ry;
            uint32_t z = n % rz;
            /* :43  offset + 1.0 * idx / resolution * size, left to right */
            o[0] = p->particle_spawn_cube_offset[0] +
                   ((1.0f * (float)x) / (float)rx) * p->particle_spawn_cube_size[0];
            o[1] = p->particle_spawn_cube_offset[1] +
                   ((1.0f * (float)y) / (float)ry) * p->particle_spawn_cube_size[1];
            o[2] = p->particle_spawn_cube_offset[2] +
                   ((1.0f * (float)z) / (float)rz) * p->particle_spawn_cube_size[2];
            o[3] = p->active_particle_w; /* :45 */
        } else {
            o[0] = o[1] = o[2] = o[3] = 0.0f; /* :48 */
        }
    }
}

/* ---- 01_update_densities/update_densities.comp:29-36 ---------------------------------------- */
/* ivec3(pos.xyz) truncates toward zero; the atomic add is dropped when the texel is outside the
 * image.  trunc(v) in [0, N-1]  <=>  v > -1 && v < N; NaN / inf / out-of-int-range positions are
 * defined as dropped. */
static int trunc_index(float v, int n, int* out) {
    if (!(v > -1.0f && v < (float)n)) return 0;
    *out = (int)v;
    return 1;
}
void oracle_01_update_densities(const fluid_params* p, const float* particles, uint64_t capacity,
                                uint32_t* densities) {
    grid g = grid_of(p);
    for (uint64_t i = 0; i < capacity; i++) {
        const float* q = particles + 4 * i;
        if (q[3] == p->active_particle_w) { /* :33 */
            int x, y, z;
            if (trunc_index(q[0], g.W, &x) && trunc_index(q[1], g.H, &y) &&
                trunc_index(q[2], g.D, &z))
                densities[cell(g, x, y, z)] += 1u; /* :35 */
        }
    }
}

/* ---- 02_update_water/update_water.comp:23-33 ------------------------------------------------ */
void oracle_02_update_water(const fluid_params* p, const uint32_t* densities, uint8_t* new_types) {
    grid g = grid_of(p);
    uint64_t n = (uint64_t)g.W * g.H * g.D;
    for (uint64_t i = 0; i < n; i++)
        new_types[i] =
            (uint8_t)(densities[i] > 0 ? p->cell_type_water : p->cell_type_inactive); /* :27-33 */
}

/* ---- 03_update_air/update_active.comp:45-66 -------------------------------------------------
 * In-place on one image.  Border cells become SOLID; an interior non-water cell with a water
 * neighbour becomes AIR.  The reference races at the border (a border thread may overwrite WATER
 * with SOLID before or after its neighbour looks at it); the oracle defines "solid first": a cell
 * on the domain border never counts as water. */
static int on_border(grid g, int x, int y, int z) {
    return x == 0 || x == g.W - 1 || y == 0 || y == g.H - 1 || z == 0 || z == g.D - 1; /* :48-50 */
}
void oracle_03_update_air(const fluid_params* p, uint8_t* t) {
    grid g = grid_of(p);
    static const int mv[6][3] = {{1, 0, 0},  {0, 1, 0},  {0, 0, 1},
                                 {-1, 0, 0}, {0, -1, 0}, {0, 0, -1}}; /* :26 */
    /* pass 1: borders (:50-51) */
    for (int z = 0; z < g.D; z++)
        for (int y = 0; y < g.H; y++)
            for (int x = 0; x < g.W; x++)
                if (on_border(g, x, y, z)) t[cell(g, x, y, z)] = (uint8_t)p->cell_type_solid;
    /* pass 2: interior (:52-64).  AIR writes never change "is water", so in-place is exact. */
    for (int z = 0; z < g.D; z++)
        for (int y = 0; y < g.H; y++)
            for (int x = 0; x < g.W; x++) {
                if (on_border(g, x, y, z)) continue;
                if (t[cell(g, x, y, z)] == p->cell_type_water) continue; /* :54 */
                int water_around = 0;
                for (int j = 0; j < 6; j++) {
                    int nx = x + mv[j][0], ny = y + mv[j][1], nz = z + mv[j][2];
                    if (on_border(g, nx, ny, nz)) continue; /* solid first */
                    if (t[cell(g, nx, ny, nz)] == p->cell_type_water) water_around = 1; /* :58 */
                }
                if (water_around) t[cell(g, x, y, z)] = (uint8_t)p->cell_type_air; /* :61-62 */
            }
}

/* ---- 04_compute_extrapolated_velocities/extrapolated_velocities.comp:37-63 ------------------- */
void oracle_04_compute_extrapolated_velocities(const fluid_params* p, const uint8_t* types,
                                               const float* v1, float* v2) {
    grid g = grid_of(p);
    const uint32_t water = p->cell_type_water;
    /* neighbour order :46-51 : -x, -y, -z, +x, +y, +z; the guards `i.c != 0` / `i.c != b.c` keep
     * every accepted neighbour in bounds */
    static const int mv[6][3] = {{-1, 0, 0}, {0, -1, 0}, {0, 0, -1},
                                 {1, 0, 0},  {0, 1, 0},  {0, 0, 1}};
    for (int z = 0; z < g.D; z++)
        for (int y = 0; y < g.H; y++)
            for (int x = 0; x < g.W; x++) {
                int c = 0;
                float s[4] = {0.0f, 0.0f, 0.0f, 0.0f}; /* :41 */
                for (int j = 0; j < 6; j++) {
                    int nx = x + mv[j][0], ny = y + mv[j][1], nz = z + mv[j][2];
                    if (!in_bounds(g, nx, ny, nz)) continue;
                    if (types[cell(g, nx, ny, nz)] == water) {
                        const float* q = v1 + 4 * cell(g, nx, ny, nz);
                        s[0] = s[0] + q[0];
                        s[1] = s[1] + q[1];
                        s[2] = s[2] + q[2];
                        c++;
                    }
                }
                float* o = v2 + 4 * cell(g, x, y, z);
                if (c != 0) { /* :53  v.xyz / c */
                    o[0] = s[0] / (float)c;
                    o[1] = s[1] / (float)c;
                    o[2] = s[2] / (float)c;
                } else {
                    o[0] = o[1] = o[2] = 0.0f; /* :55 */
                }
                o[3] = 0.0f; /* :62 */
            }
}

/* ---- 05_set_extrapolated_velocities/extrapolate_velocities.comp:48-109 ----------------------- */
static int is_active(const fluid_params* p, uint32_t t) {
    return t == p->cell_type_water || t == p->cell_type_air; /* :34-36 */
}
void oracle_05_set_extrapolated_velocities(const fluid_params* p, const uint8_t* new_types,
                                           const uint8_t* types, const float* v2, float* v1) {
    grid g = grid_of(p);
    for (int z = 0; z < g.D; z++)
        for (int y = 0; y < g.H; y++)
            for (int x = 0; x < g.W; x++) {
                uint64_t id = cell(g, x, y, z);
                int was = is_active(p, types[id]);     /* :88 */
                int is = is_active(p, new_types[id]);  /* :90 */
                float out[3];
                for (int c = 0; c < 3; c++) {
                    int nx = x - (c == 0), ny = y - (c == 1), nz = z - (c == 2); /* :74 */
                    int vel_was = was || is_active(p, type_at(g, types, nx, ny, nz));    /* :50 */
                    int vel_is = is || is_active(p, type_at(g, new_types, nx, ny, nz));  /* :52 */
                    if (vel_was && !vel_is)
                        out[c] = 0.0f; /* VELOCITY_RESET :57-58,:81 */
                    else if (!vel_was && vel_is)
                        out[c] = v2[4 * id + c]; /* VELOCITY_EXTRAPOLATE :61-62,:83 */
                    else
                        out[c] = v1[4 * id + c]; /* VELOCITY_DO_NOTHING :79 */
                }
                v1[4 * id + 0] = out[0];
                v1[4 * id + 1] = out[1];
                v1[4 * id + 2] = out[2];
                v1[4 * id + 3] = 0.0f; /* :108 */
            }
}

/* ---- 06_update_cell_types/update_cell_types.comp:15-19 --------------------------------------- */
void oracle_06_update_cell_types(const fluid_params* p, const uint8_t* new_types, uint8_t* types) {
    grid g = grid_of(p);
    memcpy(types, new_types, (uint64_t)g.W * g.H * g.D);
}

/* ---- the velocities sampler (fluid_flow_sections.h:95: LINEAR / CLAMP_TO_EDGE, normalized coords)
 * texture(velocities, (pos + move) / fluid_size)[comp]   advect.comp:52-56, particles.comp:28-36.
 * Definition (fp32): s = (pos+move)/size; u = s*size; ub = u - 0.5; i0 = floor(ub);
 * a = ub - i0; i1 = i0 + 1; both indices clamped to [0, size-1]; lerp along x, then y, then z with
 * lerp(A,B,a) = (1-a)*A + a*B. */
static void axis_taps(float coord, int n, int* i0, int* i1, float* a) {
    float s = coord / (float)n;
    float u = s * (float)n;
    float ub = u - 0.5f;
    float fl = floorf(ub);
    *a = ub - fl;
    /* clamp in float first so the int conversion is defined for any input (NaN -> -1 -> 0) */
    if (!(fl >= -1.0f)) fl = -1.0f;
    if (fl > (float)n) fl = (float)n;
    int lo = (int)fl;
    int hi = lo + 1;
    if (lo < 0) lo = 0;
    if (lo > n - 1) lo = n - 1;
    if (hi < 0) hi = 0;
    if (hi > n - 1) hi = n - 1;
    *i0 = lo;
    *i1 = hi;
}
static float lerp1(float A, float B, float a) { return (1.0f - a) * A + a * B; }

static float sample_comp(grid g, const float* v, float px, float py, float pz, int comp) {
    /* move :53-54 */
    float mx = comp == 0 ? 0.5f : 0.0f, my = comp == 1 ? 0.5f : 0.0f, mz = comp == 2 ? 0.5f : 0.0f;
    int x0, x1, y0, y1, z0, z1;
    float ax, ay, az;
    axis_taps(px + mx, g.W, &x0, &x1, &ax);
    axis_taps(py + my, g.H, &y0, &y1, &ay);
    axis_taps(pz + mz, g.D, &z0, &z1, &az);
    float c000 = v[4 * cell(g, x0, y0, z0) + comp], c100 = v[4 * cell(g, x1, y0, z0) + comp];
    float c010 = v[4 * cell(g, x0, y1, z0) + comp], c110 = v[4 * cell(g, x1, y1, z0) + comp];
    float c001 = v[4 * cell(g, x0, y0, z1) + comp], c101 = v[4 * cell(g, x1, y0, z1) + comp];
    float c011 = v[4 * cell(g, x0, y1, z1) + comp], c111 = v[4 * cell(g, x1, y1, z1) + comp];
    float c00 = lerp1(c000, c100, ax), c10 = lerp1(c010, c110, ax);
    float c01 = lerp1(c001, c101, ax), c11 = lerp1(c011, c111, ax);
    float c0 = lerp1(c00, c10, ay), c1 = lerp1(c01, c11, ay);
    return lerp1(c0, c1, az);
}
float oracle_sample_velocity_component(const fluid_params* p, const float* v, float px, float py,
                                       float pz, int comp) {
    return sample_comp(grid_of(p), v, px, py, pz, comp);
}

/* ---- 07_advect/advect.comp:63-97 -------------------------------------------------------------- */
void oracle_07_advect(const fluid_params* p, const uint8_t* types, const float* v1, float* v2) {
    grid g = grid_of(p);
    const float dt = p->time_delta;
    for (int z = 0; z < g.D; z++)
        for (int y = 0; y < g.H; y++)
            for (int x = 0; x < g.W; x++) {
                uint64_t id = cell(g, x, y, z);
                float out[3] = {v1[4 * id], v1[4 * id + 1], v1[4 * id + 2]}; /* :87 */
                int cur_water = types[id] == p->cell_type_water;            /* :93 */
                int pos[3] = {x, y, z};
                for (int c = 0; c < 3; c++) {
                    /* :65-68  move[c] = -1; cellAt(pos - move) = the cell at pos + e_c (F3) */
                    int nx = x + (c == 0), ny = y + (c == 1), nz = z + (c == 2);
                    if (pos[c] != 0 &&
                        (cur_water || type_at(g, types, nx, ny, nz) == p->cell_type_water)) {
                        /* :70-73 */
                        float qx = (float)x + (c == 0 ? 0.0f : 0.5f);
                        float qy = (float)y + (c == 1 ? 0.0f : 0.5f);
                        float qz = (float)z + (c == 2 ? 0.0f : 0.5f);
                        /* :75 */
                        float vx = sample_comp(g, v1, qx, qy, qz, 0);
                        float vy = sample_comp(g, v1, qx, qy, qz, 1);
                        float vz = sample_comp(g, v1, qx, qy, qz, 2);
                        /* :77  pos_in_tex - cur_v*time_delta */
                        out[c] = sample_comp(g, v1, qx - vx * dt, qy - vy * dt, qz - vz * dt, c);
                    }
                }
                v2[4 * id + 0] = out[0];
                v2[4 * id + 1] = out[1];
                v2[4 * id + 2] = out[2];
                v2[4 * id + 3] = 0.0f; /* :96 */
            }
}

/* ---- 08_forces/forces.comp:33-54 --------------------------------------------------------------- */
void oracle_08_forces(const fluid_params* p, const uint8_t* types, float* v2) {
    grid g = grid_of(p);
    const uint32_t water = p->cell_type_water;
    for (int z = 0; z < g.D; z++)
        for (int y = 0; y < g.H; y++)
            for (int x = 0; x < g.W; x++) {
                float fx = 0.0f, fy = 0.0f, fz = 0.0f; /* :36 */
                uint32_t t1 = types[cell(g, x, y, z)];
                uint32_t t2 = type_at(g, types, x, y - 1, z);
                if (y != 0) { /* :39-45 */
                    if (t1 == water || t2 == water) fy += p->gravity;
                }
                if ((uint32_t)x == p->fountain_position[0] &&
                    (uint32_t)y == p->fountain_position[1] &&
                    (uint32_t)z == p->fountain_position[2] && (t1 == water || t2 == water))
                    fy += p->fountain_force; /* :47-49 */
                if (fx != 0.0f || fy != 0.0f || fz != 0.0f) { /* :52-53 */
                    float* q = v2 + 4 * cell(g, x, y, z);
                    q[0] = q[0] + p->time_delta * fx;
                    q[1] = q[1] + p->time_delta * fy;
                    q[2] = q[2] + p->time_delta * fz;
                    q[3] = q[3] + 0.0f;
                }
            }
}

/* ---- 09_diffuse/diffuse.comp:31-46 -------------------------------------------------------------- */
void oracle_09_diffuse(const fluid_params* p, const uint8_t* types, const float* v2, float* v1,
                       int mode) {
    grid g = grid_of(p);
    for (int z = 0; z < g.D; z++)
        for (int y = 0; y < g.H; y++)
            for (int x = 0; x < g.W; x++) {
                uint64_t id = cell(g, x, y, z);
                float out[3] = {v2[4 * id], v2[4 * id + 1], v2[4 * id + 2]}; /* :34 */
                if (mode == FLUID_DIFFUSE_INTENDED && types[id] == p->cell_type_water) {
                    /* :38-43 — what the shader computes into the shadowed inner `velocity` */
                    float a = p->diffuse_k * p->time_delta; /* :38 */
                    float k0 = 1.0f - 6.0f * a;
                    for (int c = 0; c < 3; c++) {
                        float sum = vel_at(g, v2, x + 1, y, z, c) + vel_at(g, v2, x - 1, y, z, c);
                        sum = sum + vel_at(g, v2, x, y + 1, z, c);
                        sum = sum + vel_at(g, v2, x, y - 1, z, c);
                        sum = sum + vel_at(g, v2, x, y, z + 1, c);
                        sum = sum + vel_at(g, v2, x, y, z - 1, c);
                        out[c] = k0 * v2[4 * id + c] + a * sum;
                    }
                }
                /* as written (:40 shadows, :46 stores the outer value): a copy */
                v1[4 * id + 0] = out[0];
                v1[4 * id + 1] = out[1];
                v1[4 * id + 2] = out[2];
                v1[4 * id + 3] = 0.0f;
            }
}

/* ---- 10_solids/solids.comp:30-76 ---------------------------------------------------------------- */
void oracle_10_solids(const fluid_params* p, const uint8_t* types, float* v1) {
    grid g = grid_of(p);
    const uint32_t solid = p->cell_type_solid;
    const float r = p->solid_repel_velocity;
    for (int z = 0; z < g.D; z++)
        for (int y = 0; y < g.H; y++)
            for (int x = 0; x < g.W; x++) {
                float* q = v1 + 4 * cell(g, x, y, z);
                float v[3] = {q[0], q[1], q[2]}; /* :67 */
                if (types[cell(g, x, y, z)] == solid) /* :70-72, :30-44 */
                    for (int c = 0; c < 3; c++)
                        if (v[c] > -r) v[c] = -r;
                for (int c = 0; c < 3; c++) { /* :73, :45-62 */
                    int nx = x - (c == 0), ny = y - (c == 1), nz = z - (c == 2);
                    if (type_at(g, types, nx, ny, nz) == solid && v[c] < r) v[c] = r;
                }
                q[0] = v[0];
                q[1] = v[1];
                q[2] = v[2];
                q[3] = 1.0f; /* :76 */
            }
}

/* ---- 11_compute_divergence/compute_divergence.comp:18-30 ---------------------------------------- */
void oracle_11_compute_divergence(const fluid_params* p, const float* v1, float* div) {
    grid g = grid_of(p);
    for (int z = 0; z < g.D; z++)
        for (int y = 0; y < g.H; y++)
            for (int x = 0; x < g.W; x++) {
                uint64_t id = cell(g, x, y, z);
                float vx = v1[4 * id], vy = v1[4 * id + 1], vz = v1[4 * id + 2];
                /* :21 left to right */
                float d = vel_at(g, v1, x + 1, y, z, 0) - vx;
                d = d + vel_at(g, v1, x, y + 1, z, 1);
                d = d - vy;
                d = d + vel_at(g, v1, x, y, z + 1, 2);
                d = d - vz;
                div[id] = d;
            }
}

/* ---- 12_solve_pressure/pressure.comp:41-76 ------------------------------------------------------- */
void oracle_12_solve_pressure(const fluid_params* p, const uint8_t* types, const float* div,
                              float* p1, float* p2, uint32_t is_even_iteration) {
    grid g = grid_of(p);
    const float* pin = is_even_iteration == 1 ? p1 : p2; /* :71-75 */
    float* pout = is_even_iteration == 1 ? p2 : p1;
    const uint32_t water = p->cell_type_water, solid = p->cell_type_solid;
    static const int mv[6][3] = {{1, 0, 0},  {0, 1, 0},  {0, 0, 1},
                                 {-1, 0, 0}, {0, -1, 0}, {0, 0, -1}}; /* :56-61 */
    for (int z = 0; z < g.D; z++)
        for (int y = 0; y < g.H; y++)
            for (int x = 0; x < g.W; x++) {
                uint64_t id = cell(g, x, y, z);
                if (types[id] != water) continue; /* :69 */
                int aii = 0;
                /* :54  div * fluid_density * cell_width / time_delta */
                float s = ((div[id] * p->fluid_density) * p->cell_width) / p->time_delta;
                for (int j = 0; j < 6; j++) {
                    int nx = x + mv[j][0], ny = y + mv[j][1], nz = z + mv[j][2];
                    uint32_t t = type_at(g, types, nx, ny, nz); /* :42 */
                    if (t != solid) {                           /* :43 */
                        if (t == water)
                            s = s - f32_at(g, pin, nx, ny, nz); /* :45 */
                        else
                            s = s - p->pressure_air; /* :47 */
                        aii++;
                    }
                }
                pout[id] = -s / (float)aii; /* :62 */
            }
}
void oracle_12_solve_pressure_loop(const fluid_params* p, const uint8_t* types, const float* div,
                                   float* p1, float* p2, uint32_t iterations) {
    /* FlowLoopPushConstantSection, fluid_flow_sections.h:300-313; first dispatch has
     * is_even_iteration = 1 (inferred) */
    for (uint32_t k = 0; k < iterations; k++)
        oracle_12_solve_pressure(p, types, div, p1, p2, (k % 2u) == 0u ? 1u : 0u);
}

/* ---- opt-in pressure solver (NOT in the reference): red-black successive over-
 * relaxation on the linear system of pressure.comp:41-62.  One iteration updates, in place, first the
 * WATER cells with (x + y + z) even, then those with (x + y + z) odd:
 *     gs = -s / aii   with s as in the Jacobi sweep but from the CURRENT contents of `pr`
 *     pr = pr + omega * (gs - pr)
 * Cells of one colour have no neighbour of the same colour, so the order inside a colour is immaterial. */
void oracle_12_sor_iteration(const fluid_params* p, const uint8_t* types, const float* div, float* pr,
                             float omega) {
    grid g = grid_of(p);
    const uint32_t water = p->cell_type_water, solid = p->cell_type_solid;
    static const int mv[6][3] = {{1, 0, 0},  {0, 1, 0},  {0, 0, 1},
                                 {-1, 0, 0}, {0, -1, 0}, {0, 0, -1}};
    for (int colour = 0; colour < 2; colour++)
        for (int z = 0; z < g.D; z++)
            for (int y = 0; y < g.H; y++)
                for (int x = 0; x < g.W; x++) {
                    if (((x + y + z) & 1) != colour) continue;
                    uint64_t id = cell(g, x, y, z);
                    if (types[id] != water) continue;
                    int aii = 0;
                    float s = ((div[id] * p->fluid_density) * p->cell_width) / p->time_delta;
                    for (int j = 0; j < 6; j++) {
                        int nx = x + mv[j][0], ny = y + mv[j][1], nz = z + mv[j][2];
                        uint32_t t = type_at(g, types, nx, ny, nz);
                        if (t != solid) {
                            if (t == water)
                                s = s - f32_at(g, pr, nx, ny, nz);
                            else
                                s = s - p->pressure_air;
                            aii++;
                        }
                    }
                    const float gs = -s / (float)aii;
                    const float d = gs - pr[id];
                    const float t2 = omega * d;
                    pr[id] = pr[id] + t2;
                }
}
/* the loop section under this solver: `iterations` iterations on PRESSURES_1, then PRESSURES_2 :=
 * PRESSURES_1 (13_fix_divergence reads PRESSURES_2) */
void oracle_12_sor_loop(const fluid_params* p, const uint8_t* types, const float* div, float* p1,
                        float* p2, float omega, uint32_t iterations) {
    grid g = grid_of(p);
    for (uint32_t k = 0; k < iterations; k++) oracle_12_sor_iteration(p, types, div, p1, omega);
    memcpy(p2, p1, sizeof(float) * (size_t)g.W * g.H * g.D);
}

/* ---- 13_fix_divergence/fix_divergence.comp:41-72 -------------------------------------------------- */
void oracle_13_fix_divergence(const fluid_params* p, const uint8_t* types, const float* pr,
                              float* v1) {
    grid g = grid_of(p);
    const uint32_t water = p->cell_type_water, solid = p->cell_type_solid;
    /* :71  time_delta / fluid_density / cell_width, left to right */
    const float k = (p->time_delta / p->fluid_density) / p->cell_width;
    for (int z = 0; z < g.D; z++)
        for (int y = 0; y < g.H; y++)
            for (int x = 0; x < g.W; x++) {
                uint64_t id = cell(g, x, y, z);
                uint32_t lt = types[id]; /* :62 */
                float lp = pr[id];       /* :63 */
                int pos[3] = {x, y, z};
                float dv[3] = {0.0f, 0.0f, 0.0f};
                for (int c = 0; c < 3; c++) {
                    int nx = x - (c == 0), ny = y - (c == 1), nz = z - (c == 2); /* :43 */
                    uint32_t ct = type_at(g, types, nx, ny, nz);                 /* :44 */
                    if (pos[c] - 1 != -1 && (lt == water || ct == water)) {      /* :46 */
                        if (lt != solid && ct != solid)                          /* :48 */
                            dv[c] = lp - pr[cell(g, nx, ny, nz)];                /* :50 */
                    }
                }
                float* q = v1 + 4 * id;
                q[0] = q[0] - k * dv[0];
                q[1] = q[1] - k * dv[1];
                q[2] = q[2] - k * dv[2];
                q[3] = 0.0f;
            }
}

/* ---- 14_particles/particles.comp:45-51 ------------------------------------------------------------ */
void oracle_14_particles(const fluid_params* p, const float* v1, float* particles,
                         uint64_t capacity) {
    grid g = grid_of(p);
    const float dt = p->time_delta;
    for (uint64_t i = 0; i < capacity; i++) {
        float* q = particles + 4 * i;
        if (q[3] == p->active_particle_w) { /* :48 */
            float vx = sample_comp(g, v1, q[0], q[1], q[2], 0);
            float vy = sample_comp(g, v1, q[0], q[1], q[2], 1);
            float vz = sample_comp(g, v1, q[0], q[1], q[2], 2);
            q[0] = q[0] + vx * dt; /* :50 */
            q[1] = q[1] + vy * dt;
            q[2] = q[2] + vz * dt;
        }
    }
}

/* ================================================================================================
 * Surface-prep passes on the detailed grid (fluid_flow_sections.h:339-388).  The detailed grid has
 * detailed_resolution^3 cells per simulation cell; out-of-bounds image loads return 0, stores and
 * atomics outside the image are dropped (as everywhere else).
 * ============================================================================================== */
typedef struct {
    int W, H, D, res;
} dgrid;
static dgrid dgrid_of(const fluid_params* p) {
    dgrid g;
    g.res = p->detailed_resolution;
    g.W = (int)p->fluid_size[0] * g.res;
    g.H = (int)p->fluid_size[1] * g.res;
    g.D = (int)p->fluid_size[2] * g.res;
    return g;
}
static uint64_t dcell(dgrid g, int x, int y, int z) {
    return (uint64_t)x + (uint64_t)g.W * ((uint64_t)y + (uint64_t)g.H * (uint64_t)z);
}
static int dinside(dgrid g, int x, int y, int z) {
    return x >= 0 && x < g.W && y >= 0 && y < g.H && z >= 0 && z < g.D;
}

/* update_detailed_densities.comp:24-31: ivec3(pos.xyz * detailed_resolution) — the int is converted to
 * float, one fp32 product per axis, then truncation toward zero (same rule as 01_update_densities) */
void oracle_15_update_detailed_densities(const fluid_params* p, const float* particles,
                                         uint64_t capacity, uint32_t* detailed) {
    dgrid g = dgrid_of(p);
    const float fres = (float)g.res;
    for (uint64_t i = 0; i < capacity; i++) {
        const float* q = particles + 4 * i;
        if (q[3] == p->active_particle_w) { /* :28 */
            int x, y, z;
            const float sx = q[0] * fres, sy = q[1] * fres, sz = q[2] * fres;
            if (trunc_index(sx, g.W, &x) && trunc_index(sy, g.H, &y) && trunc_index(sz, g.D, &z))
                detailed[dcell(g, x, y, z)] += 1u; /* :30 */
        }
    }
}

/* densities_inertia.comp:30-61.  GLSL mixes uint and int: `inertia += int` converts the int to uint,
 * `inertia > inertia_decrease` and min(max_inertia, inertia) compare as uint. */
void oracle_16_compute_detailed_densities_inertia(const fluid_params* p, const uint32_t* detailed,
                                                  uint32_t* inertia) {
    dgrid g = dgrid_of(p);
    static const int mv[6][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {-1, 0, 0}, {0, -1, 0}, {0, 0, -1}};
    for (int z = 0; z < g.D; z++)
        for (int y = 0; y < g.H; y++)
            for (int x = 0; x < g.W; x++) {
                const uint64_t id = dcell(g, x, y, z);
                uint32_t in = inertia[id]; /* :38 */
                const uint32_t old = in;
                if (detailed[id] > 0u) in += (uint32_t)p->inertia_increase_filled; /* :42-44 */
                int hits = 0;
                for (int j = 0; j < 6; j++) { /* :49-52 */
                    const int nx = x + mv[j][0], ny = y + mv[j][1], nz = z + mv[j][2];
                    if (dinside(g, nx, ny, nz) && detailed[dcell(g, nx, ny, nz)] > 0u) hits += 1;
                }
                if (hits >= p->required_neighbour_hits) /* :54 */
                    in += (uint32_t)(hits * p->inertia_increase_neighbour);
                if (in == old) { /* :57-63 */
                    if (in > (uint32_t)p->inertia_decrease)
                        in -= (uint32_t)p->inertia_decrease;
                    else
                        in = 0u;
                }
                const uint32_t cap = (uint32_t)p->max_inertia; /* :65 */
                inertia[id] = cap < in ? cap : in;
            }
}

/* float_densities.comp:22-27: -1 where the inertia is 0, else float(inertia) / coefficient (IEEE fp32) */
void oracle_17_compute_float_densities(const fluid_params* p, const uint32_t* inertia, float* f1) {
    dgrid g = dgrid_of(p);
    const uint64_t n = (uint64_t)g.W * g.H * g.D;
    for (uint64_t i = 0; i < n; i++)
        f1[i] = inertia[i] == 0u ? -1.0f : (float)inertia[i] / p->dens_division_coefficient;
}

/* diffuse_densities.comp:45-62 */
static float dload(dgrid g, const float* a, int x, int y, int z) {
    return dinside(g, x, y, z) ? a[dcell(g, x, y, z)] : 0.0f;
}
void oracle_18_diffuse_float_densities(const fluid_params* p, const uint8_t* types, float* f1,
                                       float* f2, uint32_t is_even_iteration) {
    dgrid g = dgrid_of(p);
    grid sg = grid_of(p);
    const float a = p->dens_diffuse_k;
    const float* src = is_even_iteration == 1u ? f1 : f2; /* :57-61 */
    float* dst = is_even_iteration == 1u ? f2 : f1;
    const float k0 = 1.0f - 6.0f * a; /* ( 1.0 - 6 * dens_diffuse_a) */
    for (int z = 0; z < g.D; z++)
        for (int y = 0; y < g.H; y++)
            for (int x = 0; x < g.W; x++) {
                /* :56 the simulation cell of this detailed cell; integer division of non-negative ints */
                const uint32_t t = types[cell(sg, x / g.res, y / g.res, z / g.res)];
                if (t == p->cell_type_solid) continue;
                float s = dload(g, src, x + 1, y, z) + dload(g, src, x - 1, y, z); /* :47-50 */
                s = s + dload(g, src, x, y + 1, z);
                s = s + dload(g, src, x, y - 1, z);
                s = s + dload(g, src, x, y, z + 1);
                s = s + dload(g, src, x, y, z - 1);
                const float t1 = k0 * src[dcell(g, x, y, z)];
                const float t2 = a * s;
                dst[dcell(g, x, y, z)] = t1 + t2;
            }
}
void oracle_18_diffuse_float_densities_loop(const fluid_params* p, const uint8_t* types, float* f1,
                                            float* f2, uint32_t iterations) {
    for (uint32_t k = 0; k < iterations; k++)
        oracle_18_diffuse_float_densities(p, types, f1, f2, (k % 2u) == 0u ? 1u : 0u);
}

/* ---- section lists ----------------------------------------------------------------------------- */
void oracle_run_init(oracle_state* s) {
    const fluid_params* p = &s->params;
    uint64_t n = (uint64_t)p->fluid_size[0] * p->fluid_size[1] * p->fluid_size[2];
    oracle_fill_f32(s->velocities_1, 4 * n, 0.0f);                      /* fluid_flow_sections.h:140 */
    oracle_fill_u8(s->cell_types, n, (uint8_t)p->cell_type_inactive);   /* :141 */
    oracle_00_init_particles(p, s->particles, s->particle_capacity);    /* :143-153 */
}

void oracle_run_step(oracle_state* s) {
    const fluid_params* p = &s->params;
    uint64_t n = (uint64_t)p->fluid_size[0] * p->fluid_size[1] * p->fluid_size[2];
    oracle_fill_u32(s->particle_densities, n, 0u);                                           /* :163 */
    oracle_01_update_densities(p, s->particles, s->particle_capacity, s->particle_densities);/* :164 */
    oracle_02_update_water(p, s->particle_densities, s->new_cell_types);                     /* :176 */
    oracle_03_update_air(p, s->new_cell_types);                                              /* :188 */
    oracle_04_compute_extrapolated_velocities(p, s->cell_types, s->velocities_1,
                                              s->velocities_2);                              /* :199 */
    oracle_05_set_extrapolated_velocities(p, s->new_cell_types, s->cell_types, s->velocities_2,
                                          s->velocities_1);                                  /* :212 */
    oracle_06_update_cell_types(p, s->new_cell_types, s->cell_types);                        /* :226 */
    oracle_07_advect(p, s->cell_types, s->velocities_1, s->velocities_2);                    /* :237 */
    oracle_08_forces(p, s->cell_types, s->velocities_2);                                     /* :250 */
    oracle_09_diffuse(p, s->cell_types, s->velocities_2, s->velocities_1, s->diffuse_mode);  /* :262 */
    oracle_10_solids(p, s->cell_types, s->velocities_1);                                     /* :275 */
    oracle_11_compute_divergence(p, s->velocities_1, s->divergences);                        /* :287 */
    oracle_fill_f32(s->pressures_1, n, p->pressure_air);                                     /* :298 */
    oracle_fill_f32(s->pressures_2, n, p->pressure_air);                                     /* :299 */
    oracle_12_solve_pressure_loop(p, s->cell_types, s->divergences, s->pressures_1,
                                  s->pressures_2, s->pressure_iterations);                   /* :300 */
    oracle_13_fix_divergence(p, s->cell_types, s->pressures_2, s->velocities_1);             /* :314 */
    oracle_14_particles(p, s->velocities_1, s->particles, s->particle_capacity);             /* :327 */
}

/* ---- 31_render_surface: the geometry the marching-cubes renderer emits (render_surface.vert:19-25,
 * render_surface.geom:45-103), as a triangle list instead of a rasterised strip.  One render cell per vertex
 * index, fluid_surface_render_size = detailed extent - 1 per axis (simulation_constants.h); corner i of cell
 * pos is pos + moves[i] (:50); configuration bit i = density(corner i) > 0 (:93); counts[configuration]
 * triangles, the vertex on edge e = vertex_edge_indices[configuration * 15 + 3 t + i] (:60-66):
 *     a = d[e0] / (d[e0] - d[e1]);  point = (vec3(0.5) + pos + moves[e0] + (moves[e1] - moves[e0]) * a) / res
 * and the flat normal normalize(cross(p1 - p0, p2 - p0)) (:69).  GLSL leaves normalize()'s precision open; it is
 * DEFINED here as v / sqrt(dot(v, v)) with IEEE sqrt and division, dot = (x*x + y*y) + z*z (parity unpinned,
 * like the sampler).  Output: 12 floats per triangle {p0, p1, p2, N}, cells in vertex-index order (x fastest);
 * *count = triangles found, of which the first `capacity` are stored. */
void oracle_31_extract_surface(const fluid_params* p, const float* density, const uint32_t* counts,
                               const uint32_t* edge_indices, float* out, uint64_t capacity, uint64_t* count) {
    static const int mv[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0},
                                 {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};
    static const int ed[12][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 0}, {4, 5}, {5, 6},
                                  {6, 7}, {7, 4}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};
    const int res = p->detailed_resolution;
    const int W = (int)p->fluid_size[0] * res, H = (int)p->fluid_size[1] * res, D = (int)p->fluid_size[2] * res;
    const float fres = (float)res;
    uint64_t n = 0;
    for (int z = 0; z < D - 1; z++)
        for (int y = 0; y < H - 1; y++)
            for (int x = 0; x < W - 1; x++) {
                float d[8];
                int cfg = 0;
                for (int i = 0; i < 8; i++) {
                    d[i] = density[(uint64_t)(x + mv[i][0]) +
                                   (uint64_t)W * ((uint64_t)(y + mv[i][1]) + (uint64_t)H * (uint64_t)(z + mv[i][2]))];
                    cfg |= (d[i] > 0.0f ? 1 : 0) << i;
                }
                const uint32_t tris = counts[cfg];
                for (uint32_t t = 0; t < tris; t++) {
                    float pt[3][3];
                    for (int i = 0; i < 3; i++) {
                        const uint32_t e = edge_indices[cfg * 15 + 3 * t + i];
                        const int e0 = ed[e][0], e1 = ed[e][1];
                        const float a = d[e0] / (d[e0] - d[e1]);
                        const float cell[3] = {(float)x, (float)y, (float)z};
                        for (int c = 0; c < 3; c++) {
                            float v = 0.5f + cell[c];
                            v = v + (float)mv[e0][c];
                            v = v + (float)(mv[e1][c] - mv[e0][c]) * a;
                            pt[i][c] = v / fres;
                        }
                    }
                    float u[3], w[3], c3[3];
                    for (int c = 0; c < 3; c++) {
                        u[c] = pt[1][c] - pt[0][c];
                        w[c] = pt[2][c] - pt[0][c];
                    }
                    c3[0] = u[1] * w[2] - w[1] * u[2];
                    c3[1] = u[2] * w[0] - w[2] * u[0];
                    c3[2] = u[0] * w[1] - w[0] * u[1];
                    const float len = sqrtf((c3[0] * c3[0] + c3[1] * c3[1]) + c3[2] * c3[2]);
                    if (n < capacity) {
                        float* o = out + 12 * n;
                        for (int i = 0; i < 3; i++)
                            for (int c = 0; c < 3; c++) o[3 * i + c] = pt[i][c];
                        for (int c = 0; c < 3; c++) o[9 + c] = c3[c] / len;
                    }
                    n++;
                }
            }
    *count = n;
}

