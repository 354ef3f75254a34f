"""Synthetic inputs of the benchmark plan.

full_fluid   : Jacobi micro-benchmark — the 6 domain faces SOLID, every interior cell WATER,
               divergence ~ U(-1,1) from SplitMix64 (seed 0x5EED0012 + rank), P1 = P2 = p_air.
dam_break    : full-step scene — the reference spawn cube scaled to the grid, 8 particles/cell
               (params.dam_break_params); everything else comes from run_init().
"""
import numpy as np

from .params import CELL_SOLID, CELL_WATER

SEED_JACOBI = 0x5EED0012


def splitmix64(counter: np.ndarray, seed: int) -> np.ndarray:
    """Vectorised SplitMix64: the n-th output (n = counter) of the generator seeded with `seed`."""
    with np.errstate(over="ignore"):
        z = (counter.astype(np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
        z += np.uint64(seed)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform_pm1(n: int, seed: int, offset: int = 0) -> np.ndarray:
    """n floats in [-1, 1): top 24 bits of SplitMix64 output number offset+i -> exact fp32.
    (int64 arange / int64->float32 conversions: numpy's uint64 paths are ~50x slower.)"""
    out = np.empty(n, np.float32)
    chunk = 1 << 22
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        ctr = np.arange(offset + s, offset + e, dtype=np.int64).view(np.uint64)
        bits = splitmix64(ctr, seed)
        bits >>= np.uint64(40)
        o = out[s:e]
        o[:] = bits.view(np.int64)
        o *= np.float32(2.0 ** -23)
        o -= np.float32(1.0)
    return out


def full_fluid_types(shape, z_begin: int = 0, global_depth: int = None) -> np.ndarray:
    """Cell types of the Jacobi micro-benchmark for the planes [z_begin, z_begin+shape[0]) of a
    grid of depth `global_depth` (default: the slab is the whole grid)."""
    d, h, w = shape
    gd = global_depth if global_depth is not None else d
    t = np.full(shape, CELL_WATER, np.uint8)
    t[:, 0, :] = CELL_SOLID
    t[:, h - 1, :] = CELL_SOLID
    t[:, :, 0] = CELL_SOLID
    t[:, :, w - 1] = CELL_SOLID
    if z_begin == 0:
        t[0] = CELL_SOLID
    if z_begin + d == gd:
        t[d - 1] = CELL_SOLID
    return t


def full_fluid_divergence(shape, seed: int = SEED_JACOBI, z_begin: int = 0) -> np.ndarray:
    d, h, w = shape
    return uniform_pm1(d * h * w, seed, offset=z_begin * h * w).reshape(shape)
