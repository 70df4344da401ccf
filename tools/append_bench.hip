// Micro-benchmark behind the hashed-level append of csrc/ren_hashgrid_binned.hip: 256-thread workgroups write runs of
// RUN 16-byte records to pseudo-random destinations (one run per (pass, bin) as the scatter does), with the run start
// aligned to `align` bytes.  Prints GB/s for contiguous / 128-B-aligned / 16-B-aligned runs.
//   hipcc --offload-arch=gfx950 -O3 tools/append_bench.hip -o /tmp/append_bench && /tmp/append_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <int RUN>
__global__ __launch_bounds__(256) void append_kernel(float4 *out, uint64_t n_slots, int passes, int mode, int align16) {
    constexpr int RUNS = 2048 / RUN;                       // runs per pass of 2048 records (32 KB)
    __shared__ uint64_t base[RUNS];
    for (int p = 0; p < passes; ++p) {
        const uint32_t wgp = blockIdx.x * passes + p;
        if (threadIdx.x < RUNS) {
            uint64_t b;
            if (mode == 0) b = ((uint64_t)wgp * RUNS + threadIdx.x) * RUN;                       // contiguous slab
            else {
                const uint64_t r = ((uint64_t)mix(wgp * RUNS + threadIdx.x) << 8 | mix(threadIdx.x * 977 + wgp)) % (n_slots - 64);
                b = r & ~(uint64_t)(align16 - 1);                                                // start aligned to align16 records
            }
            base[threadIdx.x] = b;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = threadIdx.x + 256 * j;
            out[base[k / RUN] + (k % RUN)] = make_float4((float)k, (float)p, 1.f, 2.f);
        }
        __syncthreads();
    }
}

int main() {
    const uint64_t n_slots = (uint64_t)6 << 26;             // 6.4 GB of 16-byte slots
    float4 *buf;
    hipMalloc(&buf, n_slots * 16);
    hipMemset(buf, 0, n_slots * 16);
    const int wgs = 4096, passes = 48;                      // 4096 x 48 x 32 KB = 6.4 GB written
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](auto kern, const char *name, int mode, int align16) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 0, 0, buf, n_slots, passes, mode, align16);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %7.3f ms  %6.0f GB/s\n", name, ms, (double)wgs * passes * 32768 / ms / 1e6);
    };
    run(append_kernel<32>, "contiguous 32 KB per pass", 0, 1);
    run(append_kernel<32>, "runs of 32 records (512 B), 128-B aligned", 1, 8);
    run(append_kernel<32>, "runs of 32 records (512 B), 16-B aligned", 1, 1);
    run(append_kernel<64>, "runs of 64 records (1 KB), 128-B aligned", 1, 8);
    run(append_kernel<64>, "runs of 64 records (1 KB), 16-B aligned", 1, 1);
    run(append_kernel<16>, "runs of 16 records (256 B), 128-B aligned", 1, 8);
    run(append_kernel<16>, "runs of 16 records (256 B), 16-B aligned", 1, 1);
    return 0;
}
