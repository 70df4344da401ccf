// Does plain VALU work hide under v_mfma_f32_32x32x2_f32 on gfx950?  (tuning aid)
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_bench.hip -o /tmp/mvb && /tmp/mvb
// One workgroup per CU, W waves per SIMD.  Each iteration: 4 MFMAs on 4 independent accumulators
// followed (in program order, interleaved 1:K) by K independent VALU fma / transcendental ops.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int K, int T, bool MF>
__global__ __launch_bounds__(1024) void k_mix(float *out, int iters) {
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t)
        for (int g = 0; g < 16; ++g) acc[t][g] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    float v[16];
    for (int j = 0; j < 16; ++j) v[j] = a + j;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (MF) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < K; ++j) v[j % 16] = fmaf(v[j % 16], b, a);
#pragma unroll
            for (int j = 0; j < T; ++j) v[j % 16] = __builtin_amdgcn_exp2f(v[j % 16]);
        }
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t)
        for (int g = 0; g < 16; ++g) s += acc[t][g];
    for (int j = 0; j < 16; ++j) s += v[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// same loop with the bf16 matrix-core instruction (32x32x16, 32 cycles): does VALU hide under THAT?
template <int K, int T, int NM>
__global__ __launch_bounds__(1024) void k_mix_bf16(float *out, int iters) {
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t)
        for (int g = 0; g < 16; ++g) acc[t][g] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    bf16x8 fa, fb;
    for (int j = 0; j < 8; ++j) { fa[j] = (__bf16)(a + j); fb[j] = (__bf16)(b + j); }
    float v[16];
    for (int j = 0; j < 16; ++j) v[j] = a + j;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int m = 0; m < NM; ++m) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[t], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < K; ++j) v[j % 16] = fmaf(v[j % 16], b, a);
#pragma unroll
            for (int j = 0; j < T; ++j) v[j % 16] = __builtin_amdgcn_exp2f(v[j % 16]);
        }
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t)
        for (int g = 0; g < 16; ++g) s += acc[t][g];
    for (int j = 0; j < 16; ++j) s += v[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int K, int T, int NM>
void run_bf16(int waves_per_simd, float *out) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid(256), blk(256 * waves_per_simd);
    hipLaunchKernelGGL((k_mix_bf16<K, T, NM>), grid, blk, 0, 0, out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_mix_bf16<K, T, NM>), grid, blk, 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double cyc = ms * 1e-3 * 2.4e9 / (iters * 4.0) / waves_per_simd;
    printf("bf16 mfma x%d K=%2d T=%d waves/SIMD=%d : %7.1f cycles per group per wave-slot (%.3f ms)\n", NM, K, T,
           waves_per_simd, cyc, ms);
}

template <int K, int T, bool MF>
void run(int waves_per_simd, float *out) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid(256), blk(256 * waves_per_simd);
    hipLaunchKernelGGL((k_mix<K, T, MF>), grid, blk, 0, 0, out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_mix<K, T, MF>), grid, blk, 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // cycles per (MFMA + K VALU + T trans) group per wave, at 2.4 GHz
    const double cyc = ms * 1e-3 * 2.4e9 / (iters * 4.0) / waves_per_simd;
    printf("mfma=%d K=%2d T=%d waves/SIMD=%d : %7.1f cycles per group per wave-slot (%.3f ms)\n", (int)MF, K, T,
           waves_per_simd, cyc, ms);
}

int main() {
    float *out;
    hipMalloc(&out, 256 * 1024 * 4);
    for (int w = 1; w <= 2; ++w) {
        run<0, 0, true>(w, out);
        run<4, 0, true>(w, out);
        run<8, 0, true>(w, out);
        run<12, 0, true>(w, out);
        run<16, 0, true>(w, out);
        run<24, 0, true>(w, out);
        run<8, 0, false>(w, out);
        run<16, 0, false>(w, out);
        run<0, 2, true>(w, out);
        run<0, 4, true>(w, out);
        run<4, 2, true>(w, out);
        run<8, 4, true>(w, out);
        run<0, 4, false>(w, out);
        run<8, 4, false>(w, out);
    }
    for (int w = 1; w <= 2; ++w) {
        run_bf16<0, 0, 1>(w, out);
        run_bf16<8, 0, 1>(w, out);
        run_bf16<0, 0, 6>(w, out);
        run_bf16<8, 0, 6>(w, out);
        run_bf16<16, 0, 6>(w, out);
        run_bf16<32, 0, 6>(w, out);
        run_bf16<48, 0, 6>(w, out);
        run_bf16<16, 4, 6>(w, out);
        run_bf16<24, 8, 6>(w, out);
    }
    return 0;
}
