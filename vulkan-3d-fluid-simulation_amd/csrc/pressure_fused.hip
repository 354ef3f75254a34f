// pressure_fused.hip — translation unit of the two-sweeps-per-pass kernel (kernels_pressure_fused.h).
#include "kernels_pressure_fused.h"
#include "pressure_api.h"

namespace fluid {

bool k12_canon2_supports(const GridK& g) { return g.W % 4 == 0 && g.W <= 1024 && g.z0 == 0 && g.Dl == g.Dg; }

template <int NT>
static hipError_t launch_nt(hipStream_t s, const uint8_t* mask, const float* rhs, const float* pin,
                            float* pout, float* pmid, const uint8_t* bricks, const GridK& g,
                            float p_oob) {
    static bool attr_set = false;  // per process and instantiation; the attribute is per function
    const size_t lds = fused_lds_bytes(NT);
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k12_canon2<NT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    constexpr int TY = FUSED_WAVES / NT - 2;
    const int by = (g.H + TY - 1) / TY;
    int zchunk = g.Dl;
    while (zchunk > 32 && (int64_t)by * ((g.Dl + zchunk - 1) / zchunk) < 1024)
        zchunk = (zchunk + 1) / 2;
    const dim3 grid(1, by, (g.Dl + zchunk - 1) / zchunk);
    BrickK bk;
    bk.nbx = (g.W + BRICK_X - 1) / BRICK_X;
    bk.nby = (g.H + BRICK_Y - 1) / BRICK_Y;
    bk.nbz = (g.Dl + BRICK_Z - 1) / BRICK_Z;
    hipLaunchKernelGGL(k12_canon2<NT>, grid, dim3(FUSED_THREADS), lds, s, mask, rhs, pin, pout, pmid,
                       bricks, bk, g, p_oob, zchunk);
    return hipSuccess;
}

hipError_t k12_launch_canon2(hipStream_t s, const uint8_t* mask, const float* rhs, const float* pin,
                             float* pout, float* pmid, const uint8_t* bricks, const GridK& g,
                             float p_oob) {
    const int nt = (g.W + 255) / 256;
    if (nt == 1) return launch_nt<1>(s, mask, rhs, pin, pout, pmid, bricks, g, p_oob);
    if (nt == 2) return launch_nt<2>(s, mask, rhs, pin, pout, pmid, bricks, g, p_oob);
    if (nt <= 4) return launch_nt<4>(s, mask, rhs, pin, pout, pmid, bricks, g, p_oob);
    return hipErrorInvalidValue;
}

}  // namespace fluid
