// How do the matrix pipe and the VALU of one SIMD share time in an "MLP layer" loop on gfx950?  (tuning aid, stand-alone)
//   hipcc --offload-arch=gfx950 -O3 tools/phase_overlap_bench.hip -o /tmp/pob && /tmp/pob
// One iteration = one 64 -> 64 layer of the split-bf16 chain of csrc/ren_mlp_x.hip for a 32-sample block:
//   MFMA phase: NM v_mfma_f32_32x32x16_bf16 into two accumulators (operands: registers made by the previous VALU phase and
//               "weight" fragments from LDS (LDSW) or registers),
//   VALU phase: the 32 accumulator values per lane -> softplus-like activation (ACT) -> 3-piece bf16 split -> the next operands.
// Run as W waves per SIMD (one workgroup of 256 W threads per CU).  Printed: cycles per iteration and SIMD at the clock
// the chip reports, for the full loop, the MFMA phase alone and the VALU phase alone: sum vs. max tells how much of the
// two pipes' time a second / fourth wave per SIMD can overlap.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MFMAB(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ void split3(float v, __bf16 (&t)[3]) {
    t[0] = (__bf16)v;
    const float r = v - (float)t[0];
    t[1] = (__bf16)r;
    t[2] = (__bf16)(r - (float)t[1]);
}
__device__ __forceinline__ void split8(const float *v, bf16x8 (&out)[3]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        __bf16 t[3];
        split3(v[j], t);
#pragma unroll
        for (int k = 0; k < 3; ++k) out[k][j] = t[k];
    }
}
__device__ __forceinline__ float softplus100(float x) {
    const float t = __builtin_amdgcn_exp2f(fabsf(x) * -144.26950408889634f);
    return fmaf(__builtin_amdgcn_logf(1.f + t), 0.006931471805599453f, __builtin_amdgcn_fmed3f(x, 0.f, 3.0e38f));
}

// DO_M: issue the MFMAs; DO_V: do the VALU phase (otherwise the operands stay what they were); ACT: activation in the VALU
// phase; LDSW: weight fragments through ds_read_b128 (one per MFMA) instead of registers
template <bool DO_M, bool DO_V, bool ACT, bool LDSW>
__global__ __launch_bounds__(1024) void layer_loop(float *out, int iters, long long *clk) {
    const long long c0 = clock64(), w0c = wall_clock64();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16x8 *wl = reinterpret_cast<bf16x8 *>(smem);
    const int lane = threadIdx.x & 63;
    for (int e = threadIdx.x; e < 2 * 4 * 3 * 64; e += blockDim.x) {
        bf16x8 w;
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] = (__bf16)(0.01f * ((e * 7 + j * 3) % 17 - 8));
        wl[e] = w;
    }
    __syncthreads();
    bf16x8 wr[3];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) wr[k][j] = (__bf16)(0.01f * ((lane + j + k) % 13 - 6));
    bf16x8 b[4][3];
    {
        float v[8];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = 0.001f * (lane + 8 * c + j);
            split8(v, b[c]);
        }
    }
    constexpr int W6[6] = {2, 0, 1, 1, 0, 0}, A6[6] = {0, 2, 1, 0, 1, 0};
    float keep = 0.f;
    for (int it = 0; it < iters; ++it) {
        int zo = 0;
        asm volatile("" : "+v"(zo));
        const bf16x8 *W = wl + zo;
        f32x16 a0, a1;
#pragma unroll
        for (int g = 0; g < 16; ++g) { a0[g] = 0.01f; a1[g] = -0.01f; }
        if (DO_M) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const bf16x8 w0 = LDSW ? W[((0 * 4 + c) * 3 + W6[k]) * 64 + lane] : wr[W6[k]];
                    const bf16x8 w1 = LDSW ? W[((1 * 4 + c) * 3 + W6[k]) * 64 + lane] : wr[(W6[k] + 1) % 3];
                    a0 = MFMAB(w0, b[c][A6[k]], a0);
                    a1 = MFMAB(w1, b[c][A6[k]], a1);
                }
        } else {
#pragma unroll
            for (int g = 0; g < 16; ++g) { a0[g] += (float)b[g & 3][0][g >> 1]; a1[g] -= (float)b[g & 3][1][g >> 1]; }   // 32 distinct values
        }
        if (DO_V) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float y[16];
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const float x = t ? a1[g] : a0[g];
                    y[g] = ACT ? softplus100(x) : x * 0.5f;
                }
                split8(y, b[2 * t]);
                split8(y + 8, b[2 * t + 1]);
            }
        } else {
#pragma unroll
            for (int g = 0; g < 16; ++g) keep += a0[g] + a1[g];
        }
    }
    float s = keep;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int j = 0; j < 8; ++j) s += (float)b[c][k][j];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0c; }   // shader cycles, 100 MHz ticks
}

static double clock_ghz = 2.4, last_ghz = 0, last_ms = 0;

template <bool DO_M, bool DO_V, bool ACT, bool LDSW>
double run(int waves, float *out) {
    static long long *clk = nullptr;
    if (!clk) hipHostMalloc(&clk, 16);
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const dim3 grid(256), blk(256 * waves);
    const size_t lds = 96 * 1024;          // > half of the CU's LDS: exactly one workgroup per CU (the dispatcher otherwise packs two on one CU)
    (void)hipFuncSetAttribute((const void *)layer_loop<DO_M, DO_V, ACT, LDSW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((layer_loop<DO_M, DO_V, ACT, LDSW>), grid, blk, lds, 0, out, 50, clk);
    hipEventRecord(e0);
    hipLaunchKernelGGL((layer_loop<DO_M, DO_V, ACT, LDSW>), grid, blk, lds, 0, out, iters, clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // cycles by the shader clock counter (s_memtime) of workgroup 0 -- the chip clocks down under matrix-core load, so the
    // event time at the nominal clock overstates cycles; the effective clock is printed once per configuration
    last_ghz = (double)clk[0] / (double)clk[1] * 0.1;
    last_ms = ms;
    // NOT clk[0] / iters / waves: the counter is read by wave 0 of its workgroup, the OLDEST wave of its SIMD, which wins the
    // issue arbitration and finishes at nearly its stand-alone speed while the younger waves take what is left; the event
    // time covers all of them
    return ms * 1e-3 * last_ghz * 1e9 / iters / waves;
}

int main() {
    float *out;
    hipMalloc(&out, sizeof(float) * 256 * 1024);
    printf("one 64->64 split-bf16 layer per iteration: 48 MFMAs (1536 matrix-pipe cycles) + 32 values/lane activation + 3-piece split\n");
    printf("cycles per layer and SIMD = event time x effective clock (s_memtime / s_memrealtime of one wave) [that clock, GHz]; W = waves per SIMD\n");
    for (int w : {1, 2, 4}) {
        double c[7], g[7];
        c[0] = run<true, true, true, false>(w, out);   g[0] = last_ghz;     // full
        c[1] = run<true, false, true, false>(w, out);  g[1] = last_ghz;     // MFMA phase only
        c[2] = run<false, true, true, false>(w, out);  g[2] = last_ghz;     // VALU phase only
        c[3] = run<true, true, true, true>(w, out);    g[3] = last_ghz;     // full, weights from LDS
        c[4] = run<true, false, true, true>(w, out);   g[4] = last_ghz;     // MFMA only, weights from LDS
        c[5] = run<true, true, false, false>(w, out);  g[5] = last_ghz;     // full, split only (no activation)
        c[6] = run<false, true, false, false>(w, out); g[6] = last_ghz;     // VALU only, split only
        printf("W=%d  full %6.0f [%.2f]  mfma-only %6.0f [%.2f]  valu-only %6.0f [%.2f] | LDS weights: full %6.0f [%.2f] mfma-only %6.0f [%.2f] | "
               "no activation: full %6.0f [%.2f] valu-only %6.0f [%.2f]\n", w, c[0], g[0], c[1], g[1], c[2], g[2], c[3], g[3], c[4], g[4],
               c[5], g[5], c[6], g[6]);
    }
    return 0;
}
