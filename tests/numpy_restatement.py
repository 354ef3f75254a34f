"""A second, independent restatement of the shader math in vectorised numpy — written array-wise from
the GLSL text, not from oracle/fluid_oracle.c — used only to cross-check the C oracle
(tests/test_numpy_crosscheck.py).  fp32 throughout, one operation per numpy call in the shader's order.
Out-of-bounds image loads are modelled by zero padding.  Paths: /root/reference/shaders_fluid.
"""
import numpy as np

F = np.float32


def pad1(a, value=0):
    """One-cell border of `value` around the three grid axes (extra trailing axes untouched)."""
    width = [(1, 1)] * 3 + [(0, 0)] * (a.ndim - 3)
    return np.pad(a, width, constant_values=value)


def shifted(ap, dx, dy, dz):
    """View of the padded array `ap` that gives, at [z,y,x], the value of cell (x+dx, y+dy, z+dz)."""
    d, h, w = ap.shape[0] - 2, ap.shape[1] - 2, ap.shape[2] - 2
    return ap[1 + dz:1 + dz + d, 1 + dy:1 + dy + h, 1 + dx:1 + dx + w]


def coords(shape):
    d, h, w = shape
    z, y, x = np.meshgrid(np.arange(d), np.arange(h), np.arange(w), indexing="ij")
    return x, y, z


# 02_update_water/update_water.comp:23-33
def update_water(p, dens):
    return np.where(dens > 0, p.cell_type_water, p.cell_type_inactive).astype(np.uint8)


# 03_update_air/update_active.comp:45-66, "solid first" 
def update_air(p, t):
    d, h, w = t.shape
    x, y, z = coords(t.shape)
    border = (x == 0) | (x == w - 1) | (y == 0) | (y == h - 1) | (z == 0) | (z == d - 1)
    water = (t == p.cell_type_water) & ~border
    wp = pad1(water, False)
    around = np.zeros(t.shape, bool)
    for dx, dy, dz in [(1, 0, 0), (0, 1, 0), (0, 0, 1), (-1, 0, 0), (0, -1, 0), (0, 0, -1)]:
        around |= shifted(wp, dx, dy, dz)
    out = t.copy()
    out[~border & (t != p.cell_type_water) & around] = p.cell_type_air
    out[border] = p.cell_type_solid
    return out


# 04_compute_extrapolated_velocities/extrapolated_velocities.comp:37-63
def extrapolated_velocities(p, t, v1):
    water = pad1(t == p.cell_type_water, False)
    vp = pad1(v1)
    s = np.zeros(t.shape + (3,), F)
    c = np.zeros(t.shape, np.int32)
    for dx, dy, dz in [(-1, 0, 0), (0, -1, 0), (0, 0, -1), (1, 0, 0), (0, 1, 0), (0, 0, 1)]:  # :46-51
        m = shifted(water, dx, dy, dz)
        s = np.where(m[..., None], s + shifted(vp, dx, dy, dz)[..., :3], s).astype(F)
        c = c + m
    out = np.zeros(t.shape + (4,), F)
    with np.errstate(all="ignore"):
        q = (s / np.maximum(c, 1)[..., None].astype(F)).astype(F)
    out[..., :3] = np.where((c != 0)[..., None], q, F(0))
    return out


# 05_set_extrapolated_velocities/extrapolate_velocities.comp:48-109
def set_extrapolated_velocities(p, new_t, old_t, v2, v1):
    act = lambda a: (a == p.cell_type_water) | (a == p.cell_type_air)  # noqa: E731
    was, now = act(old_t), act(new_t)
    wasp, nowp = pad1(was, False), pad1(now, False)
    out = np.zeros_like(v1)
    for c, (dx, dy, dz) in enumerate([(-1, 0, 0), (0, -1, 0), (0, 0, -1)]):
        vwas = was | shifted(wasp, dx, dy, dz)
        vnow = now | shifted(nowp, dx, dy, dz)
        out[..., c] = np.where(vwas & ~vnow, F(0), np.where(~vwas & vnow, v2[..., c], v1[..., c]))
    return out


# the sampler: fluid_flow_sections.h:95, advect.comp:52-56
def sample(v, comp, px, py, pz):
    d, h, w = v.shape[:3]

    def taps(coord, n):
        s = (coord / F(n)).astype(F)
        u = (s * F(n)).astype(F)
        ub = (u - F(0.5)).astype(F)
        fl = np.floor(ub).astype(F)
        a = (ub - fl).astype(F)
        fl = np.where(fl >= -1, fl, F(-1))
        fl = np.where(fl > n, F(n), fl)
        i0 = fl.astype(np.int64)
        return np.clip(i0, 0, n - 1), np.clip(i0 + 1, 0, n - 1), a

    half = [F(0.5) if comp == i else F(0) for i in range(3)]
    x0, x1, ax = taps((px + half[0]).astype(F), w)
    y0, y1, ay = taps((py + half[1]).astype(F), h)
    z0, z1, az = taps((pz + half[2]).astype(F), d)
    f = v[..., comp]
    lerp = lambda A, B, a: ((F(1) - a).astype(F) * A + a * B).astype(F)  # noqa: E731
    c00 = lerp(f[z0, y0, x0], f[z0, y0, x1], ax)
    c10 = lerp(f[z0, y1, x0], f[z0, y1, x1], ax)
    c01 = lerp(f[z1, y0, x0], f[z1, y0, x1], ax)
    c11 = lerp(f[z1, y1, x0], f[z1, y1, x1], ax)
    return lerp(lerp(c00, c10, ay), lerp(c01, c11, ay), az)


# 07_advect/advect.comp:63-97
def advect(p, t, v1):
    x, y, z = coords(t.shape)
    water = t == p.cell_type_water
    wp = pad1(water, False)
    dt = F(p.time_delta)
    out = np.zeros_like(v1)
    pos = [x, y, z]
    for c, (dx, dy, dz) in enumerate([(1, 0, 0), (0, 1, 0), (0, 0, 1)]):  # the +neighbour (F3)
        do = (pos[c] != 0) & (water | shifted(wp, dx, dy, dz))
        q = [(pos[i].astype(F) + (F(0) if i == c else F(0.5))).astype(F) for i in range(3)]
        vel = [sample(v1, k, q[0], q[1], q[2]) for k in range(3)]
        b = [(q[i] - (vel[i] * dt).astype(F)).astype(F) for i in range(3)]
        out[..., c] = np.where(do, sample(v1, c, b[0], b[1], b[2]), v1[..., c])
    return out


# 08_forces/forces.comp:33-54
def forces(p, t, v2):
    x, y, z = coords(t.shape)
    water = t == p.cell_type_water
    wet = water | shifted(pad1(water, False), 0, -1, 0)
    fy = np.zeros(t.shape, F)
    fy = np.where((y != 0) & wet, (fy + F(p.gravity)).astype(F), fy)
    fx, fyy, fz = p.fountain_position[:]
    fountain = (x == fx) & (y == fyy) & (z == fz) & wet
    fy = np.where(fountain, (fy + F(p.fountain_force)).astype(F), fy)
    out = v2.copy()
    hit = fy != 0
    out[..., 1] = np.where(hit, (v2[..., 1] + (F(p.time_delta) * fy).astype(F)).astype(F), v2[..., 1])
    return out


# 10_solids/solids.comp:30-76
def solids(p, t, v1):
    r = F(p.solid_repel_velocity)
    solid = t == p.cell_type_solid
    sp = pad1(solid, False)
    out = v1.copy()
    for c, (dx, dy, dz) in enumerate([(-1, 0, 0), (0, -1, 0), (0, 0, -1)]):
        v = v1[..., c]
        v = np.where(solid & (v > -r), -r, v)
        v = np.where(shifted(sp, dx, dy, dz) & (v < r), r, v)
        out[..., c] = v
    out[..., 3] = 1.0
    return out


# 11_compute_divergence/compute_divergence.comp:21
def divergence(v1):
    vp = pad1(v1)
    d = (shifted(vp, 1, 0, 0)[..., 0] - v1[..., 0]).astype(F)
    d = (d + shifted(vp, 0, 1, 0)[..., 1]).astype(F)
    d = (d - v1[..., 1]).astype(F)
    d = (d + shifted(vp, 0, 0, 1)[..., 2]).astype(F)
    return (d - v1[..., 2]).astype(F)


# 12_solve_pressure/pressure.comp:41-76, one dispatch
def pressure_sweep(p, t, div, pin, pout):
    water = t == p.cell_type_water
    tp = pad1(t, 0)
    pp = pad1(pin)
    s = (((div * F(p.fluid_density)).astype(F) * F(p.cell_width)).astype(F) / F(p.time_delta)).astype(F)
    aii = np.zeros(t.shape, np.int32)
    for dx, dy, dz in [(1, 0, 0), (0, 1, 0), (0, 0, 1), (-1, 0, 0), (0, -1, 0), (0, 0, -1)]:
        tn = shifted(tp, dx, dy, dz)
        contrib = np.where(tn == p.cell_type_water, shifted(pp, dx, dy, dz), F(p.pressure_air)).astype(F)
        live = tn != p.cell_type_solid
        s = np.where(live, (s - contrib).astype(F), s)
        aii = aii + live
    with np.errstate(all="ignore"):
        new = (-s / aii.astype(F)).astype(F)
    return np.where(water, new, pout)


# 13_fix_divergence/fix_divergence.comp:41-72
def fix_divergence(p, t, pr, v1):
    x, y, z = coords(t.shape)
    water, solid = t == p.cell_type_water, t == p.cell_type_solid
    wp, sp, pp = pad1(water, False), pad1(solid, False), pad1(pr)
    k = ((F(p.time_delta) / F(p.fluid_density)).astype(F) / F(p.cell_width)).astype(F)
    out = v1.copy()
    pos = [x, y, z]
    for c, (dx, dy, dz) in enumerate([(-1, 0, 0), (0, -1, 0), (0, 0, -1)]):
        ok = (pos[c] != 0) & (water | shifted(wp, dx, dy, dz)) & ~solid & ~shifted(sp, dx, dy, dz)
        dv = np.where(ok, (pr - shifted(pp, dx, dy, dz)).astype(F), F(0))
        out[..., c] = (v1[..., c] - (k * dv).astype(F)).astype(F)
    out[..., 3] = 0.0
    return out


# 14_particles/particles.comp:45-51
def move_particles(p, v1, particles):
    out = particles.copy()
    act = particles[:, 3] == F(p.active_particle_w)
    q = [particles[:, i] for i in range(3)]
    for c in range(3):
        vel = sample(v1, c, q[0], q[1], q[2])
        out[:, c] = np.where(act, (particles[:, c] + (vel * F(p.time_delta)).astype(F)).astype(F),
                             particles[:, c])
    return out


# 01_update_densities/update_densities.comp:29-36
def update_densities(p, particles, shape):
    d, h, w = shape
    dens = np.zeros(shape, np.uint32)
    with np.errstate(invalid="ignore"):
        act = particles[:, 3] == F(p.active_particle_w)
        ok = act.copy()
        idx = []
        for c, n in enumerate([w, h, d]):
            v = particles[:, c]
            ok &= (v > -1) & (v < n)
            idx.append(np.trunc(np.where(np.isfinite(v), v, 0)).astype(np.int64))
    np.add.at(dens, (idx[2][ok], idx[1][ok], idx[0][ok]), 1)
    return dens
