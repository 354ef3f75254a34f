// Dev micro-benchmark: read and read+write rates over working sets from 32 MB to 2 GB, each walked 20 times in a
// row — does a working set that fits the 256 MB memory-side cache stream faster than HBM?  (Input to DESIGN.md
// "next": whether blocking the pressure loop in z so that a slab's planes stay cached between sweeps can pay.)
//   hipcc --offload-arch=gfx950 -O3 -o mall_bw mall_bw.hip && ./mall_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void __launch_bounds__(256) k_read(const float4* __restrict__ a, size_t n, float* sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (; i < n; i += stride) {
        const float4 v = a[i];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (s.x + s.y + s.z + s.w == 12345.678f) *sink = s.x;
}
__global__ void __launch_bounds__(256) k_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) b[i] = a[i];
}
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_copy_nt(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) __builtin_nontemporal_store(((const f4*)a)[i], &((f4*)b)[i]);
}
__global__ void __launch_bounds__(256) k_copy_nt2(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) __builtin_nontemporal_store(__builtin_nontemporal_load(&((const f4*)a)[i]), &((f4*)b)[i]);
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    const size_t max_bytes = (size_t)2 << 30;
    float4 *a, *b; float* sink;
    CK(hipMalloc(&a, max_bytes)); CK(hipMalloc(&b, max_bytes)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(a, 0, max_bytes)); CK(hipMemset(b, 0, max_bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 20;
    printf("working set | read GB/s | copy GB/s (read + write bytes; set = source + destination)\n");
    for (size_t mb : {32, 64, 96, 128, 192, 256, 384, 512, 1024, 2048}) {
        const size_t bytes = mb << 20, n = bytes / 16;
        float ms_r = 0.f, ms_c = 0.f;
        for (int w = 0; w < 2; w++) {  // first round warms up
            CK(hipEventRecord(e0));
            for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_read, dim3(256 * 16), dim3(256), 0, 0, a, n, sink);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_r, e0, e1));
            CK(hipEventRecord(e0));
            for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_copy, dim3(256 * 16), dim3(256), 0, 0, a, b, n / 2);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_c, e0, e1));
        }
        float ms_n = 0.f, ms_n2 = 0.f;
        for (int w = 0; w < 2; w++) {
            CK(hipEventRecord(e0));
            for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_copy_nt, dim3(256 * 16), dim3(256), 0, 0, a, b, n / 2);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_n, e0, e1));
            CK(hipEventRecord(e0));
            for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_copy_nt2, dim3(256 * 16), dim3(256), 0, 0, a, b, n / 2);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_n2, e0, e1));
        }
        printf("%6zu MB   | %8.0f  | %8.0f | nt store %8.0f | nt load + store %8.0f\n", mb,
               (double)bytes * reps / (ms_r * 1e-3) / 1e9, (double)bytes * reps / (ms_c * 1e-3) / 1e9,
               (double)bytes * reps / (ms_n * 1e-3) / 1e9, (double)bytes * reps / (ms_n2 * 1e-3) / 1e9);
    }
    return 0;
}
