// Does the 256 MiB Infinity Cache absorb a write -> read round trip?  (round 6: the binned scatter writes 11.8 GB of records and the
// accumulate pass reads them back -- from HBM.  If a working set below the Infinity Cache's size comes back at more than HBM speed,
// a (sample chunk, level) schedule whose records are consumed while still on die pays.)
//   hipcc --offload-arch=gfx950 -O3 tools/mall_probe.hip -o /tmp/mall_probe && /tmp/mall_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void wr(float4 *p, size_t n16, float v) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) p[i] = make_float4(v, v, v, v);
}
__global__ __launch_bounds__(256) void rd(const float4 *p, size_t n16, float *out) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) { const float4 v = p[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 1234.5f) out[0] = s;
}
// scattered appends like the scatter kernel's: every workgroup writes 512-byte runs at pseudo-random 512-byte-aligned places of the region
__global__ __launch_bounds__(256) void wr_runs(float4 *p, size_t n16, float v) {
    const size_t runs = n16 / 32;                       // 32 x 16 B = 512 B
    const size_t stride = (size_t)gridDim.x * (blockDim.x / 32);
    const int sub = threadIdx.x & 31;
    for (size_t r = (size_t)blockIdx.x * (blockDim.x / 32) + threadIdx.x / 32; r < runs; r += stride) {
        const size_t at = (r * 2654435761ull) % runs;
        p[at * 32 + sub] = make_float4(v, v, v, v);
    }
}

int main() {
    const size_t MAXB = (size_t)4 << 30;
    float4 *buf; float *out;
    hipMalloc(&buf, MAXB); hipMalloc(&out, 4);
    hipEvent_t e0, e1, e2; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
    const int grid = 256 * 16;
    printf("%10s  %10s %10s %10s | chunked: total bytes 4 GiB moved as write(chunk) -> read(chunk) pairs\n", "chunk MiB", "write TB/s", "read TB/s", "w+r TB/s");
    for (size_t mb : {32, 64, 96, 128, 192, 256, 384, 512, 1024, 4096}) {
        const size_t bytes = mb << 20, n16 = bytes / 16, reps = MAXB / bytes;
        for (int variant = 0; variant < 2; ++variant) {
            // warm
            hipLaunchKernelGGL(wr, dim3(grid), dim3(256), 0, 0, buf, n16, 1.f);
            hipLaunchKernelGGL(rd, dim3(grid), dim3(256), 0, 0, buf, n16, out);
            hipDeviceSynchronize();
            float tw = 0, tr = 0;
            for (size_t r = 0; r < reps; ++r) {
                float4 *p = buf + r * n16;              // a fresh region every time: no reuse beyond the pair itself
                hipEventRecord(e0);
                if (variant == 0) hipLaunchKernelGGL(wr, dim3(grid), dim3(256), 0, 0, p, n16, 2.f);
                else hipLaunchKernelGGL(wr_runs, dim3(grid), dim3(256), 0, 0, p, n16, 2.f);
                hipEventRecord(e1);
                hipLaunchKernelGGL(rd, dim3(grid), dim3(256), 0, 0, p, n16, out);
                hipEventRecord(e2);
                hipEventSynchronize(e2);
                float a, b; hipEventElapsedTime(&a, e0, e1); hipEventElapsedTime(&b, e1, e2);
                tw += a; tr += b;
            }
            const double tot = (double)bytes * reps;
            printf("%10zu  %10.2f %10.2f %10.2f  %s\n", mb, tot / tw / 1e9, tot / tr / 1e9, 2 * tot / (tw + tr) / 1e9, variant ? "512-B runs at random places" : "streaming");
        }
    }
    return 0;
}
