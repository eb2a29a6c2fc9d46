// probe: cost of a software grid barrier (agent-scope atomic counter + spin) on MI355X, per barrier
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(256) void k(unsigned *ctr, int n, float *sink, const float *src) {
    float acc = 0.f;
    for (int i = 0; i < n; ++i) {
        acc += src[(blockIdx.x * 256 + threadIdx.x + i) & 0xffff];   // a little memory work per phase
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(i + 1) * gridDim.x;
            while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
    }
    sink[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_relaxed(unsigned *ctr, int n, float *sink, const float *src) {
    float acc = 0.f;
    for (int i = 0; i < n; ++i) {
        acc += src[(blockIdx.x * 256 + threadIdx.x + i) & 0xffff];
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(i + 1) * gridDim.x;
            while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
    }
    sink[blockIdx.x * 256 + threadIdx.x] = acc;
}
int main() {
    unsigned *ctr; float *sink, *src;
    (void)hipMalloc(&ctr, 4); (void)hipMalloc(&sink, 1024 * 256 * 4); (void)hipMalloc(&src, 65536 * 4);
    (void)hipMemset(src, 0, 65536 * 4);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int n = 200;
    for (int relaxed = 0; relaxed < 2; ++relaxed)
        for (int g : {16, 64, 128, 256, 512, 1024}) {
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipMemset(ctr, 0, 4);
                (void)hipEventRecord(a);
                if (relaxed) hipLaunchKernelGGL(k_relaxed, dim3(g), dim3(256), 0, 0, ctr, n, sink, src);
                else hipLaunchKernelGGL(k, dim3(g), dim3(256), 0, 0, ctr, n, sink, src);
                (void)hipEventRecord(b); (void)hipEventSynchronize(b);
                float ms; (void)hipEventElapsedTime(&ms, a, b);
                if (ms < best) best = ms;
            }
            printf("%s  blocks %5d : %.2f us per barrier\n", relaxed ? "relaxed        " : "release/acquire", g, best * 1e3f / n);
        }
    return 0;
}
