// Shared device helpers of the fused-MLP kernels (parameter block / LDS image layout, activations,
// SH encoder, per-sample geometry, slab reduction).  Internal; see ren_mlp.hip for the design notes.
#pragma once
#include "ren_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// ---- parameter block offsets (floats), torch nn.Linear layout ------------------------------
constexpr int P_BW0 = 0;          // base.w0 [64,32]
constexpr int P_BB0 = 2048;       // base.b0 [64]
constexpr int P_BWO = 2112;       // base.wo [16,64]
constexpr int P_BBO = 3136;       // base.bo [16]
constexpr int P_HW0 = 3152;       // head.w0 [64,31]
constexpr int P_HB0 = 5136;       // head.b0 [64]
constexpr int P_HW1 = 5200;       // head.w1 [64,64]
constexpr int P_HB1 = 9296;       // head.b1 [64]
constexpr int P_HWO = 9360;       // head.wo [C,64]   ; head.bo [C] follows
constexpr int P_BASE_N = 3152;    // number of base-MLP parameters
__host__ __device__ constexpr int p_total(int C) { return P_HWO + 65 * C; }

// ---- LDS weight image (floats) -----------------------------------------------------------
constexpr int L_W1 = 0;                    // [64][33]
constexpr int L_W2 = L_W1 + 64 * 33;       // [32][65]  rows >= 16 are zero
constexpr int L_WH1 = L_W2 + 32 * 65;      // [64][33]  input order v: 0 = sigma slot (w=0), 1..15 geo, 16..31 SH
constexpr int L_WH2 = L_WH1 + 64 * 33;     // [64][65]
constexpr int L_WH3 = L_WH2 + 64 * 65;     // [3][64]
constexpr int L_B1 = L_WH3 + 192;          // [64]
constexpr int L_B2 = L_B1 + 64;            // [32]  >= 16 zero
constexpr int L_BH1 = L_B2 + 32;           // [64]
constexpr int L_BH2 = L_BH1 + 64;          // [64]
constexpr int L_BH3 = L_BH2 + 64;          // [4]
constexpr int L_WEIGHTS_END = L_BH3 + 4;   // 10884 floats = 43 536 B

__device__ __forceinline__ constexpr int rowc(int g) { return (g & 3) + 8 * (g >> 2); }   // + 4*hi

__device__ __forceinline__ float log1p_fast(float e) {       // e >= 0
    return e < 1e-3f ? e * (1.f - e * (0.5f - e * 0.33333333f)) : __logf(1.f + e);
}
// torch softplus(beta=100, threshold=20) = max(x,0) + log1p(exp(-100|x|))/100 up to 2e-11 (the
// threshold branch drops exactly that term).  One v_exp_f32 + one v_log_f32, no branches; the
// absolute error of log2(1+t) near t -> 0 (6e-8 * ln2/100 = 4e-10) is far below the fp32 noise of
// the layer sums it feeds.  REN_ACT_VARIANT 0 keeps the branchy log1p form for A/B timing.
#ifndef REN_ACT_VARIANT
#define REN_ACT_VARIANT 1
#endif
__device__ __forceinline__ float softplus100(float x) {
#if REN_ACT_VARIANT == 0
    const float z = 100.f * x;
    return z > 20.f ? x : log1p_fast(__expf(z)) * 0.01f;
#elif REN_ACT_VARIANT == 2
    return x;                                                  // timing experiment only
#else
    const float t = __builtin_amdgcn_exp2f(fabsf(x) * -144.26950408889634f);
    // fmed3 instead of fmaxf: no IEEE canonicalisation (an extra v_max per value) of the MFMA output
    return fmaf(__builtin_amdgcn_logf(1.f + t), 0.006931471805599453f, __builtin_amdgcn_fmed3f(x, 0.f, 3.0e38f));
#endif
}
// output activation (beta = 1): small outputs matter relatively (log intensity), keep log1p exact
__device__ __forceinline__ float softplus1(float x) { return x > 20.f ? x : log1p_fast(__expf(x)); }
// derivative of softplus(beta) expressed through its OUTPUT y: sigmoid(beta x) = 1 - exp(-beta y)
__device__ __forceinline__ float dsoftplus_from_out(float y, float beta) {
#if REN_ACT_VARIANT != 0
    if (beta == 100.f) {
#if REN_ACT_VARIANT == 2
        return 1.f;
#endif
        return 1.f - __builtin_amdgcn_exp2f(y * -144.26950408889634f);
    }
#endif
    const float t = beta * y;
    return t < 1e-3f ? t * (1.f - t * (0.5f - t * 0.16666667f)) : 1.f - __expf(-t);
}

// ---- activation alternatives of the YAML (robust_e_nerf/models/nerf.py:8-29): hidden layers {softplus beta 100 | relu}
// separately for the base and the head MLP, density {shifted_trunc_exp | softplus beta 1 | shifted_softplus = softplus(x - 1)},
// radiance {softplus beta 1 | sigmoid}.  The exact-f32 kernels (ren_mlp.hip, ren_mlp_jvp.hip, ren_jvp2.hip) take them at
// run time as one code (REN_KNOB_ACTIVATIONS, include/ren_amd.h): bits 0-1 base hidden, 2-3 density, 4-5 head hidden, 6-7
// radiance; 0 = the shipped configs.  Derivatives are expressed through the OUTPUT (hidden, radiance) or the raw
// pre-activation (density), as the kernels keep those.
struct ActKinds { int bh, dn, hh, rd; };
__device__ __forceinline__ ActKinds act_kinds(int code) { return ActKinds{code & 3, (code >> 2) & 3, (code >> 4) & 3, (code >> 6) & 3}; }
__device__ __forceinline__ float act_hidden(float x, int k) { return k == 1 ? fmaxf(x, 0.f) : softplus100(x); }
__device__ __forceinline__ float dact_hidden(float y, int k) { return k == 1 ? (y > 0.f ? 1.f : 0.f) : dsoftplus_from_out(y, 100.f); }
__device__ __forceinline__ float d2act_hidden(float s, int k) { return k == 1 ? 0.f : 100.f * (1.f - s) * s; }     // from s = act'
__device__ __forceinline__ float act_density(float raw, int k) {                       // ngp.py:45-65,247-250; nerf.py:8-13,21-25
    if (k == 0) return __expf(raw - 1.f);
    return softplus1(k == 2 ? raw - 1.f : raw);
}
__device__ __forceinline__ float dact_density(float raw, int k) {                      // trunc_exp: backward clamped at 15
    if (k == 0) return __expf(fminf(raw - 1.f, 15.f));
    return 1.f / (1.f + __expf(-(k == 2 ? raw - 1.f : raw)));
}
// second derivative w.r.t. raw, from the first (d1); trunc_exp: the clamped branch of its backward has none
__device__ __forceinline__ float d2act_density(float raw, float d1, int k) { return k == 0 ? ((raw - 1.f) < 15.f ? d1 : 0.f) : d1 * (1.f - d1); }
__device__ __forceinline__ float act_radiance(float z, int k) { return k == 1 ? 1.f / (1.f + __expf(-z)) : softplus1(z); }
__device__ __forceinline__ float dact_radiance(float y, int k) { return k == 1 ? y * (1.f - y) : dsoftplus_from_out(y, 1.f); }
__device__ __forceinline__ float d2act_radiance(float y, float s, int k) { return k == 1 ? s * (1.f - 2.f * y) : (1.f - s) * s; }

// bf16 MLP mode (BASELINE configs[2]: "bf16 MLP with fp32 composite"): every nn.Linear sees bf16-rounded
// inputs and bf16-rounded weights (the host passes a rounded copy of the parameter block), products are
// exact in fp32 and accumulate in fp32 -- numerically what a bf16 MFMA with fp32 accumulation computes, here
// still issued on the f32 pipe (true bf16 MFMA kernels are a later round).  The backward pass is the exact
// derivative of that forward (straight-through rounding).
template <bool RB>
__device__ __forceinline__ float lin_in(float v) {
    if (!RB) return v;
    return (float)(__bf16)v;                                   // round-to-nearest-even, v_cvt_pk_bf16_f32
}

__device__ void fill_base(float *lds, const float *__restrict__ P, int oW1, int oW2, int oB1, int oB2) {
    const int t = threadIdx.x, nt = blockDim.x;
    for (int i = t; i < 64 * 32; i += nt) lds[oW1 + (i >> 5) * 33 + (i & 31)] = P[P_BW0 + i];
    for (int i = t; i < 32 * 64; i += nt) {
        const int o = i >> 6, k = i & 63;
        lds[oW2 + o * 65 + k] = o < 16 ? P[P_BWO + o * 64 + k] : 0.f;
    }
    for (int i = t; i < 64; i += nt) lds[oB1 + i] = P[P_BB0 + i];
    for (int i = t; i < 32; i += nt) lds[oB2 + i] = i < 16 ? P[P_BBO + i] : 0.f;
}

__device__ void fill_head(float *lds, const float *__restrict__ P, int C, int oWH1, int oWH2, int oWH3,
                          int oBH1, int oBH2, int oBH3) {
    const int t = threadIdx.x, nt = blockDim.x;
    for (int i = t; i < 64 * 32; i += nt) {
        const int o = i >> 5, v = i & 31;
        float w;
        if (v == 0) w = 0.f;                                   // sigma_raw is not a head input (ngp.py:244-246)
        else if (v < 16) w = P[P_HW0 + o * 31 + 15 + v];       // geo feature v-1 -> column 16 + (v-1)
        else w = P[P_HW0 + o * 31 + (v - 16)];                 // SH component v-16 -> column v-16 (ngp.py:259)
        lds[oWH1 + o * 33 + v] = w;
    }
    for (int i = t; i < 64 * 64; i += nt) lds[oWH2 + (i >> 6) * 65 + (i & 63)] = P[P_HW1 + i];
    for (int i = t; i < 64 * C; i += nt) lds[oWH3 + i] = P[P_HWO + i];
    for (int i = t; i < 64; i += nt) { lds[oBH1 + i] = P[P_HB0 + i]; lds[oBH2 + i] = P[P_HB1 + i]; }
    for (int i = t; i < C; i += nt) lds[oBH3 + i] = P[P_HWO + 64 * C + i];
}

// compact LDS images for the two backward kernels
constexpr int LH_WH1 = 0, LH_WH2 = LH_WH1 + 64 * 33, LH_WH3 = LH_WH2 + 64 * 65, LH_BH1 = LH_WH3 + 192,
              LH_BH2 = LH_BH1 + 64, LH_BH3 = LH_BH2 + 64, LH_END = LH_BH3 + 4;       // 6596 floats
constexpr int LB_W1 = 0, LB_W2 = LB_W1 + 64 * 33, LB_B1 = LB_W2 + 32 * 65, LB_B2 = LB_B1 + 64,
              LB_END = LB_B2 + 32;                                                     // 4288 floats

// real SH degree 4, tcnn sign convention (external/sh_encoder.py:56-77); returns the 8
// components of parity `hi` (component 2j+hi in out[j]).
__device__ __forceinline__ void sh4_select(float x, float y, float z, int hi, float *out) {
    const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
    float s[16];
    s[0] = 0.28209479177387814f;
    s[1] = -0.48860251190291987f * y;
    s[2] = 0.48860251190291987f * z;
    s[3] = -0.48860251190291987f * x;
    s[4] = 1.0925484305920792f * xy;
    s[5] = -1.0925484305920792f * yz;
    s[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
    s[7] = -1.0925484305920792f * xz;
    s[8] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
    s[9] = 0.59004358992664352f * y * (-3.0f * x2 + y2);
    s[10] = 2.8906114426405538f * xy * z;
    s[11] = 0.45704579946446572f * y * (1.0f - 5.0f * z2);
    s[12] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f);
    s[13] = 0.45704579946446572f * x * (1.0f - 5.0f * z2);
    s[14] = 1.4453057213202769f * z * (x2 - y2);
    s[15] = 0.59004358992664352f * x * (-x2 + 3.0f * y2);
    // every component stays a scalar value: the SLP vectoriser otherwise packs pairs of them into v_pk_*_f32 with op_sel
    // (a source's HIGH half feeding the LOW result lane) -- the one instruction form that returns wrong results for an aligned
    // group of 16 lanes beside bf16-MFMA waves on MI355X (tools/pkf32_hazard_repro.hip; `build.py --audit` keeps it out)
#pragma unroll
    for (int k = 1; k < 16; ++k) asm volatile("" : "+v"(s[k]));
#pragma unroll
    for (int j = 0; j < 8; ++j) out[j] = hi ? s[2 * j + 1] : s[2 * j];
}

struct SampleSrc {
    const float *x_world, *dirs;                 // per-sample (seam API) -- or --
    const float *rays_o, *rays_d;                // packed stream
    const int32_t *ray_indices;
    const float *t_starts, *t_ends;
};

// position (contracted -> selector) and view direction of sample i
__device__ __forceinline__ void sample_geom(const SampleSrc &s, const ren_scene_dev &sc, int64_t i,
                                            bool &sel, float &dx, float &dy, float &dz) {
    float x, y, z;
    if (s.ray_indices) {
        int ray;
        ren_sample_pos(s.rays_o, s.rays_d, s.ray_indices, s.t_starts, s.t_ends, i, x, y, z, ray);
        const float *d = s.rays_d + 3 * (int64_t)ray;
        dx = d[0]; dy = d[1]; dz = d[2];
    } else {
        x = s.x_world[3 * i]; y = s.x_world[3 * i + 1]; z = s.x_world[3 * i + 2];
        if (s.dirs) { dx = s.dirs[3 * i]; dy = s.dirs[3 * i + 1]; dz = s.dirs[3 * i + 2]; }
        else { dx = 0.f; dy = 0.f; dz = 1.f; }
    }
    float ux, uy, uz;
    ren_contract(sc, x, y, z, ux, uy, uz);
    sel = ux > 0.f && ux < 1.f && uy > 0.f && uy < 1.f && uz > 0.f && uz < 1.f;   // ngp.py:238
}

// ---- split-bf16 arithmetic shared by ren_mlp_x.hip and ren_dense.hip ------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MFMAB(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

// ---- split / pack ---------------------------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ void split(float v, __bf16 (&t)[3]) {
    t[0] = (__bf16)v;
    if (NT > 1) {
        const float r = v - (float)t[0];
        t[1] = (__bf16)r;
        if (NT > 2) t[2] = (__bf16)(r - (float)t[1]);
    }
}

// 8 fp32 values -> NT bf16x8 operands
#ifndef REN_SPLIT_PAIRS
#define REN_SPLIT_PAIRS 1
#endif
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
template <int NT>
__device__ __forceinline__ void split8(const float *v, bf16x8 (&out)[3]) {
#if REN_SPLIT_PAIRS
    // two values at a time: ONE v_cvt_pk_bf16_f32 per piece and pair, the residuals as packed subtractions (the plain / neg
    // forms of v_pk_add_f32 only: lanes stay in their halves, nothing for op_sel to do -- tools/pkf32_hazard_repro.hip)
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        f32x2 x = {v[j], v[j + 1]};
        const bf16x2 t0 = __builtin_convertvector(x, bf16x2);
        out[0][j] = t0[0]; out[0][j + 1] = t0[1];
        if (NT > 1) {
            x -= __builtin_convertvector(t0, f32x2);
            const bf16x2 t1 = __builtin_convertvector(x, bf16x2);
            out[1][j] = t1[0]; out[1][j + 1] = t1[1];
            if (NT > 2) {
                x -= __builtin_convertvector(t1, f32x2);
                const bf16x2 t2 = __builtin_convertvector(x, bf16x2);
                out[2][j] = t2[0]; out[2][j + 1] = t2[1];
            }
        }
    }
#else
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        __bf16 t[3];
        split<NT>(v[j], t);
#pragma unroll
        for (int k = 0; k < NT; ++k) out[k][j] = t[k];
    }
#endif
}

// (weight term, activation term) pairs, smallest products first
template <int MODE> struct Pairs;
template <> struct Pairs<1> { static constexpr int N = 1, NT = 1; static constexpr int W[1] = {0}, A[1] = {0}; };
template <> struct Pairs<6> {
    static constexpr int N = 6, NT = 3;
    static constexpr int W[6] = {2, 0, 1, 1, 0, 0}, A[6] = {0, 2, 1, 0, 1, 0};
};
// MODE 3: two pieces per fp32 value, three products a1 b2 + a2 b1 + a1 b1 ("each float32 as the sum of two bfloat16":
// what torch.set_float32_matmul_precision("high") names -- scripts/run.py:34-35, the YAMLs' float32_matmul_precision):
// ~16 significant bits per product (error <= 2^-16 |a b|), half of MODE 6's matrix-pipe time
template <> struct Pairs<3> {
    static constexpr int N = 3, NT = 2;
    static constexpr int W[3] = {1, 0, 0}, A[3] = {0, 1, 0};
};

// grad[j] += sum_w slab[w * len + j], deterministic (fixed summation tree, no atomics).  16 parameters x 16
// slab groups per workgroup: with one thread per parameter walking all ~1 024 slabs the kernel was a 0.1 ms
// latency chain, five of them per training step.
constexpr int RS_P = 16, RS_Q = 16;
__global__ __launch_bounds__(RS_P * RS_Q) void reduce_slabs_kernel(const float *__restrict__ slab, int n_slabs, int len,
                                                                   float *__restrict__ grad, int64_t stride) {
    __shared__ float part[RS_Q][RS_P + 1];
    const int p = threadIdx.x % RS_P, q = threadIdx.x / RS_P;
    const int j = blockIdx.x * RS_P + p;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if (j < len) {
        int w = q;
        for (; w + 3 * RS_Q < n_slabs; w += 4 * RS_Q) {
#pragma unroll
            for (int u = 0; u < 4; ++u) s[u] += slab[(int64_t)(w + u * RS_Q) * stride + j];
        }
        for (; w < n_slabs; w += RS_Q) s[0] += slab[(int64_t)w * stride + j];
    }
    part[q][p] = (s[0] + s[1]) + (s[2] + s[3]);
    __syncthreads();
    if (q == 0 && j < len) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < RS_Q; ++k) t += part[k][p];
        grad[j] += t;
    }
}

// up to three slab sets in ONE launch (the MLP backward's head / base sets, the tangent backward's three): the same sums in the
// same order as reduce_slabs_kernel per set, a third of the launches (each is a ~13 us latency chain at any size)
struct SlabSet { const float *slab; float *grad; int n_slabs, len, first_block; };
struct SlabSets { SlabSet s[3]; int n; };
__global__ __launch_bounds__(RS_P * RS_Q) void reduce_slab_sets_kernel(SlabSets sets) {
    __shared__ float part[RS_Q][RS_P + 1];
    int k = 0;
    if (sets.n > 1 && (int)blockIdx.x >= sets.s[1].first_block) k = 1;
    if (sets.n > 2 && (int)blockIdx.x >= sets.s[2].first_block) k = 2;
    const SlabSet &S = sets.s[k];
    const int p = threadIdx.x % RS_P, q = threadIdx.x / RS_P;
    const int j = ((int)blockIdx.x - S.first_block) * RS_P + p;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (j < S.len) {
        int w = q;
        for (; w + 3 * RS_Q < S.n_slabs; w += 4 * RS_Q) {
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] += S.slab[(int64_t)(w + u * RS_Q) * S.len + j];
        }
        for (; w < S.n_slabs; w += RS_Q) acc[0] += S.slab[(int64_t)w * S.len + j];
    }
    part[q][p] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    __syncthreads();
    if (q == 0 && j < S.len) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < RS_Q; ++i) t += part[i][p];
        S.grad[j] += t;
    }
}
inline void launch_reduce_slab_sets(SlabSets sets, hipStream_t st) {
    int blocks = 0;
    for (int k = 0; k < sets.n; ++k) { sets.s[k].first_block = blocks; blocks += (sets.s[k].len + RS_P - 1) / RS_P; }
    hipLaunchKernelGGL(reduce_slab_sets_kernel, dim3(blocks), dim3(RS_P * RS_Q), 0, st, sets);
}

// stride: floats between consecutive slabs (0: len, the slabs are dense)
inline void launch_reduce_slabs(const float *slab, int n_slabs, int len, float *grad, hipStream_t st, int64_t stride = 0) {
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((len + RS_P - 1) / RS_P), dim3(RS_P * RS_Q), 0, st, slab, n_slabs, len, grad,
                       stride ? stride : (int64_t)len);
}

}  // namespace
