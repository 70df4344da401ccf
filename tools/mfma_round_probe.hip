// How does v_mfma_f32_32x32x16_bf16 round its fp32 accumulation?  D = A.B + C with exact bf16 products that fall between
// two fp32 neighbours of C.  hipcc --offload-arch=gfx950 -O2 tools/mfma_round_probe.hip -o /tmp/mfma_round_probe && /tmp/mfma_round_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// every lane: A row = [a, 0, 0, ...] (k slot 0 only for lanes < 32), B col = [b, 0, ...]; C = c everywhere
__global__ void probe(const float *cases, int n, float *out) {
    const int lane = threadIdx.x;
    for (int t = 0; t < n; ++t) {
        const float a = cases[4 * t], b = cases[4 * t + 1], c = cases[4 * t + 2], a2 = cases[4 * t + 3];
        bf16x8 A, B;
        for (int j = 0; j < 8; ++j) { A[j] = (__bf16)0.f; B[j] = (__bf16)0.f; }
        if (lane < 32) { A[0] = (__bf16)a; B[0] = (__bf16)b; A[1] = (__bf16)a2; B[1] = (__bf16)b; }
        f32x16 C;
        for (int g = 0; g < 16; ++g) C[g] = c;
        C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, C, 0, 0, 0);
        if (lane == 0) out[t] = C[0];
    }
    // many tiny addends beside one large C: all 16 k-slots carry tiny * 2^-15
    for (int t = 0; t < 4; ++t) {
        const float tiny = (t & 1) ? -ldexpf(1.f, -15) : ldexpf(1.f, -15), c = (t & 2) ? -1.f : 1.f;
        bf16x8 A, B;
        for (int j = 0; j < 8; ++j) { A[j] = (__bf16)tiny; B[j] = (__bf16)ldexpf(1.f, -15); }
        f32x16 C;
        for (int g = 0; g < 16; ++g) C[g] = c;
        C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, C, 0, 0, 0);
        if (lane == 0) out[n + t] = C[0];
    }
    // a statistical probe: random bf16 operands, mean of (mfma - exact) over many outputs, in units of 2^-24 |result scale|
}

int main() {
    const float e25 = ldexpf(1.f, -25), e24 = ldexpf(1.f, -24);
    // a * b (+ a2 * b) + c
    float cases[][4] = {
        {e25, 1.f, 1.f, 0.f},        // 1 + 2^-25   : RNE 1            RTZ 1            floor 1
        {-e25, 1.f, 1.f, 0.f},       // 1 - 2^-25   : RNE 1            RTZ 1 - 2^-24    floor 1 - 2^-24
        {e25, 1.f, -1.f, 0.f},       // -1 + 2^-25  : RNE -1           RTZ -(1 - 2^-24) floor -1
        {-e25, 1.f, -1.f, 0.f},      // -1 - 2^-25  : RNE -1           RTZ -1           floor -(1 + 2^-23)
        {3 * e25, 1.f, 1.f, 0.f},    // 1 + 1.5 ulp/2... = 1 + 3*2^-25: RNE 1 + 2^-23   trunc 1
        {e24, 1.f, 1.f, e24},        // 1 + 2^-24 + 2^-24 = 1 + 2^-23 exactly if the products are summed first
        {e25, 1.f, 1.f, e25},        // 1 + 2^-24: tie -> RNE 1 (even)
    };
    const int n = sizeof(cases) / sizeof(cases[0]);
    float *dc, *dout, out[16];
    hipMalloc(&dc, sizeof(cases)); hipMalloc(&dout, sizeof(out));
    hipMemcpy(dc, cases, sizeof(cases), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dc, n, dout);
    hipMemcpy(out, dout, (n + 4) * sizeof(float), hipMemcpyDeviceToHost);
    for (int t = 0; t < n; ++t) {
        const double exact = (double)cases[t][0] * cases[t][1] + (double)cases[t][3] * cases[t][1] + cases[t][2];
        printf("case %d: exact %.12g  mfma %.9g  (diff from c: %+.3g ulp(1)=%.3g)  rne %.9g\n", t, exact, out[t],
               (double)out[t] - cases[t][2], (double)ldexpf(1.f, -23), (float)exact);
    }
    for (int t = 0; t < 4; ++t)
        printf("16 products of %s2^-30 + (%s1): exact %.12g  mfma %.9g (rne %.9g)\n", (t & 1) ? "-" : "+", (t & 2) ? "-" : "+",
               ((t & 2) ? -1.0 : 1.0) + ((t & 1) ? -1.0 : 1.0) * 16 * ldexp(1.0, -30), out[n + t],
               (float)(((t & 2) ? -1.0 : 1.0) + ((t & 1) ? -1.0 : 1.0) * 16 * ldexp(1.0, -30)));
    return 0;
}
