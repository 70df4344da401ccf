// Micro-benchmark: LDS atomic throughput on gfx950 for random addresses (tuning aid, not product).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int MODE>
__global__ __launch_bounds__(1024) void k(const uint32_t *idx, int n_per_thread, float *out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *f = (float *)smem; uint32_t *u = (uint32_t *)smem; unsigned long long *q = (unsigned long long *)smem;
    for (int e = threadIdx.x; e < 32768; e += 1024) u[e] = 0;
    __syncthreads();
    const uint32_t *p = idx + (size_t)blockIdx.x * 1024 * n_per_thread + threadIdx.x;
    for (int i = 0; i < n_per_thread; ++i) {
        uint32_t a = p[(size_t)i * 1024] & 16383;
        if (MODE == 0) atomicAdd(&f[a], 1.25f);
        if (MODE == 1) atomicAdd(&u[a], 3u);
        if (MODE == 2) atomicAdd(&q[a], 3ull);
        if (MODE == 3) { f[a] += 1.25f; }                    // non-atomic RMW (racy; rate reference)
        if (MODE == 4) atomicAdd(&f[a & 31], 1.25f);        // heavy same-address conflicts
        if (MODE == 5) atomicMax(&u[a], a);
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = f[0] + f[77];
}
int main() {
    const int blocks = 1024, npt = 2048; size_t n = (size_t)blocks * 1024 * npt;
    uint32_t *h = (uint32_t *)malloc(n * 4); uint32_t s = 12345;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = s >> 8; }
    uint32_t *d; float *o; hipMalloc(&d, n * 4); hipMalloc(&o, blocks * 4); hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const char *names[] = {"ds_add_f32", "ds_add_u32", "ds_add_u64", "plain RMW f32", "ds_add_f32 32 addrs", "ds_max_u32"};
    for (int m = 0; m < 6; ++m) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a);
            size_t lds = 131072;
            if (m == 0) { hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); k<0><<<blocks, 1024, lds>>>(d, npt, o); }
            if (m == 1) { hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); k<1><<<blocks, 1024, lds>>>(d, npt, o); }
            if (m == 2) { hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); k<2><<<blocks, 1024, lds>>>(d, npt, o); }
            if (m == 3) { hipFuncSetAttribute((const void*)k<3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); k<3><<<blocks, 1024, lds>>>(d, npt, o); }
            if (m == 4) { hipFuncSetAttribute((const void*)k<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); k<4><<<blocks, 1024, lds>>>(d, npt, o); }
            if (m == 5) { hipFuncSetAttribute((const void*)k<5>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); k<5><<<blocks, 1024, lds>>>(d, npt, o); }
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (rep) printf("%-22s %8.3f ms  %7.1f G ops/s  (%.2f lanes/clk/CU @2.1GHz)\n", names[m], ms, n / ms / 1e6, n / ms / 1e6 / 256 / 2.1);
        }
    }
    return 0;
}
