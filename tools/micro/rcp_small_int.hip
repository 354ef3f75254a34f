// dev: is v_rcp_f32 the correctly rounded reciprocal of the integers 1..8 on this GPU?  (The Jacobi kernels'
// quotient n / aii, aii = 0..6, needs RN(1 / aii): tests/divide_small_int_check.c proves the three-instruction
// chain exact for exactly that value.)  hipcc --offload-arch=gfx950 -o rcp_small_int rcp_small_int.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
__global__ void k(float* out) {
    const float a = (float)threadIdx.x;
    out[threadIdx.x] = __builtin_amdgcn_rcpf(a);
    out[16 + threadIdx.x] = 1.0f / a;
}
int main() {
    float* d;
    float h[32];
    if (hipMalloc(&d, sizeof h) != hipSuccess) return 2;
    hipLaunchKernelGGL(k, dim3(1), dim3(16), 0, 0, d);
    if (hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    // exit status 0: for a = 1..6 (the aii of pressure.comp:53-61) the instruction returns RN(1 / a) or one ulp
    // less — the two values tests/divide_small_int_check.c proves the quotient chain exact for
    int bad = 0;
    for (int i = 0; i < 16; i++) {
        unsigned a, b;
        memcpy(&a, &h[i], 4);
        memcpy(&b, &h[16 + i], 4);
        printf("a = %2d  v_rcp_f32 %08x  1.0f / a %08x  %s\n", i, a, b, a == b ? "same" : (a + 1 == b ? "one ulp less" : "DIFFERENT"));
        if (i >= 1 && i <= 6) bad += !(a == b || a + 1 == b);
    }
    return bad ? 1 : 0;
}
