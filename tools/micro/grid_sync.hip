// Dev micro-benchmark: what does a device-wide barrier cost on this GPU, against back-to-back launches of an
// (almost) empty kernel?  hipcc --offload-arch=gfx950 -O3 -o grid_sync grid_sync.hip && ./grid_sync
#include <hip/hip_cooperative_groups.h>
#include <hip/hip_runtime.h>
#include <cstdio>
namespace cg = cooperative_groups;

__global__ void __launch_bounds__(1024) k_sync(int n, unsigned* sink) {
    cg::grid_group grid = cg::this_grid();
    unsigned v = threadIdx.x;
    for (int i = 0; i < n; i++) {
        v = v * 1664525u + 1013904223u;
        grid.sync();
    }
    if (v == 0xdeadbeefu) *sink = v;
}
// hand-made barrier: one counter per phase, first thread of each workgroup arrives and spins
__global__ void __launch_bounds__(1024) k_spin(int n, unsigned* counters, unsigned* sink) {
    unsigned v = threadIdx.x;
    for (int i = 0; i < n; i++) {
        v = v * 1664525u + 1013904223u;
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            atomicAdd(&counters[i], 1u);
            while (__atomic_load_n(&counters[i], __ATOMIC_ACQUIRE) < gridDim.x) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
    }
    if (v == 0xdeadbeefu) *sink = v;
}
__global__ void __launch_bounds__(1024) k_empty(unsigned* sink) {
    if (threadIdx.x == 0xffffffffu) *sink = 1;
}

int main() {
    unsigned *sink, *counters;
    hipMalloc(&sink, 4);
    const int n = 200;
    hipMalloc(&counters, 4 * n);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int blocks : {128, 247, 256}) {
        for (int rep = 0; rep < 2; rep++) {
            int nn = n;
            void* args[] = {&nn, &sink};
            hipEventRecord(a);
            hipError_t e = hipLaunchCooperativeKernel((void*)k_sync, dim3(blocks), dim3(1024), args, 0, 0);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms = 0;
            hipEventElapsedTime(&ms, a, b);
            if (rep) printf("blocks %d: cooperative grid.sync  %.2f us each (%s)\n", blocks, 1e3 * ms / n, hipGetErrorString(e));
            hipMemset(counters, 0, 4 * n);
            hipEventRecord(a);
            void* args2[] = {&nn, &counters, &sink};
            e = hipLaunchCooperativeKernel((void*)k_spin, dim3(blocks), dim3(1024), args2, 0, 0);
            hipEventRecord(b);
            hipEventSynchronize(b);
            hipEventElapsedTime(&ms, a, b);
            if (rep) printf("blocks %d: atomic counter barrier  %.2f us each (%s)\n", blocks, 1e3 * ms / n, hipGetErrorString(e));
            hipEventRecord(a);
            for (int i = 0; i < n; i++) hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(1024), 0, 0, sink);
            hipEventRecord(b);
            hipEventSynchronize(b);
            hipEventElapsedTime(&ms, a, b);
            if (rep) printf("blocks %d: empty launches           %.2f us each\n", blocks, 1e3 * ms / n);
        }
    }
    return 0;
}
