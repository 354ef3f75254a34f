// Dev tool: what does a pure streaming kernel with the Jacobi sweep's traffic shape reach on this
// GPU?  (a) float4 copy; (b) 2 x float4 + 1 x uchar4 in, float4 out  (13 B/cell, no stencil).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_copy(const float4* __restrict__ a, float4* __restrict__ o, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, s = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += s) o[i] = a[i];
}
__global__ void k_shape(const float4* __restrict__ a, const float4* __restrict__ b,
                        const uint32_t* __restrict__ m, float4* __restrict__ o, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, s = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += s) {
        float4 x = a[i], y = b[i];
        uint32_t w = m[i];
        float f = (float)(w & 3u);
        o[i] = make_float4(x.x + y.x * f, x.y + y.y, x.z + y.z, x.w + y.w);
    }
}
// same, but each block streams a contiguous chunk (like a z-marching wave) instead of grid-stride
__global__ void k_shape_chunk(const float4* __restrict__ a, const float4* __restrict__ b,
                              const uint32_t* __restrict__ m, float4* __restrict__ o, size_t n,
                              size_t per_block) {
    size_t begin = (size_t)blockIdx.x * per_block, end = begin + per_block;
    if (end > n) end = n;
    for (size_t i = begin + threadIdx.x; i < end; i += blockDim.x) {
        float4 x = a[i], y = b[i];
        uint32_t w = m[i];
        float f = (float)(w & 3u);
        o[i] = make_float4(x.x + y.x * f, x.y + y.y, x.z + y.z, x.w + y.w);
    }
}

int main() {
    const size_t cells = 512ull * 512 * 512, n = cells / 4;
    float4 *a, *b, *o; uint32_t* m;
    CK(hipMalloc(&a, cells * 4)); CK(hipMalloc(&b, cells * 4)); CK(hipMalloc(&o, cells * 4));
    CK(hipMalloc(&m, cells));
    CK(hipMemset(a, 0, cells * 4)); CK(hipMemset(b, 0, cells * 4)); CK(hipMemset(m, 1, cells));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](auto launch, const char* name, double bytes) {
        for (int i = 0; i < 3; i++) launch();
        hipEventRecord(e0);
        const int reps = 20;
        for (int i = 0; i < reps; i++) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
        printf("%-40s %.4f ms  %.2f TB/s\n", name, ms, bytes / ms / 1e9);
    };
    for (int blocks : {2048, 4096, 8192, 16384}) {
        char nm[64];
        snprintf(nm, 64, "copy float4, %d blocks", blocks);
        time([&] { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, a, o, n); }, nm, cells * 8.0);
        snprintf(nm, 64, "13B shape grid-stride, %d blocks", blocks);
        time([&] { hipLaunchKernelGGL(k_shape, dim3(blocks), dim3(256), 0, 0, a, b, m, o, n); }, nm, cells * 13.0);
        snprintf(nm, 64, "13B shape chunked, %d blocks", blocks);
        size_t per = (n + blocks - 1) / blocks;
        time([&] { hipLaunchKernelGGL(k_shape_chunk, dim3(blocks), dim3(256), 0, 0, a, b, m, o, n, per); }, nm, cells * 13.0);
    }
    return 0;
}
