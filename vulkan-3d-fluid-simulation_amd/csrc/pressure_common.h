// pressure_common.h — device helpers shared by the single-sweep kernels (kernels_pressure.h) and the
// two-sweeps-per-pass kernel (kernels_pressure_fused.h) of 12_solve_pressure.  The two kernel families
// are compiled as separate translation units on purpose: hipcc's instruction schedule of one kernel
// moves by several percent when an unrelated kernel in the same unit changes (measured with
// tools/ab_libs.py), and the sweep kernels sit at the streaming ceiling in their current form.
#pragma once

#include "device_common.h"

namespace fluid {

// Activity bricks: one byte per 256 x 4 x 16 cells (x, y, z), non-zero iff the brick holds a water
// cell.  A sweep touches nothing in a brick without water, so whole wavefronts skip such regions
// (the reference's threads return at `if (t == cell_type_water)`, pressure.comp:69).
constexpr int BRICK_X = 256, BRICK_Y = 4, BRICK_Z = 16;
struct BrickK {
    int nbx, nby, nbz;
};
__device__ __forceinline__ int brick_index(const BrickK& k, int bx, int by, int bz) {
    return bx + k.nbx * (by + k.nby * bz);
}

// Working-buffer value of a cell that is not water (what it contributes as a neighbour).
__device__ __forceinline__ float background_value(uint32_t type, const ParamsK& p) {
    return type == p.t_solid ? 0.0f : p.p_air;
}

// lane i <- lane i-1 / lane i+1 across the whole wavefront (DPP wave_shr:1 / wave_shl:1);
// lanes 0 / 63 receive `edge`.
__device__ __forceinline__ float from_lane_below(float v, float edge, int lane) {
    const int r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xF, 0xF, false);
    return lane == 0 ? edge : __int_as_float(r);
}
__device__ __forceinline__ float from_lane_above(float v, float edge, int lane) {
    const int r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xF, 0xF, false);
    return lane == 63 ? edge : __int_as_float(r);
}

// Mask byte of a cell (k12_prepare): the number of its non-solid neighbours, 0..6, if the cell is
// WATER (aii of pressure.comp:53-61), MASK_DRY otherwise.
constexpr uint32_t MASK_DRY = 8u;
constexpr uint32_t MASK_DRY4 = 0x08080808u;
// any of the four cells packed in a mask word is water
__device__ __forceinline__ bool mask_any_water(uint32_t m) { return (m & MASK_DRY4) != MASK_DRY4; }
// cell i (0..3) of a mask word is water
__device__ __forceinline__ bool mask_is_water(uint32_t m, int i) {
    return ((m >> (8 * i)) & MASK_DRY) == 0u;
}

// one water cell: b = b_i, byte i of m = its mask byte, q* = working pressures of the six
// neighbours (solid ones hold +0.0f, so subtracting them is the shader's "skip")
__device__ __forceinline__ float canon_cell(float b, uint32_t m, int i, float qxp, float qyp,
                                            float qzp, float qxm, float qym, float qzm) {
    float s = b;
    s = s - qxp;  // pressure.comp:56-61 order: +x, +y, +z, -x, -y, -z
    s = s - qyp;
    s = s - qzp;
    s = s - qxm;
    s = s - qym;
    s = s - qzm;
    const float aii = (float)((m >> (8 * i)) & 0xFFu);  // v_cvt_f32_ubyte<i>
    return -s / aii;  // :62
}

__device__ __forceinline__ float4 ld_f4(const float* base, unsigned byte_off) {
    return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + byte_off);
}

}  // namespace fluid
