// Value + tangent helpers shared by the tangent MLP kernels (ren_mlp_jvp.hip: exact-f32 MFMA, ren_mlp_jvp_x.hip:
// bf16 matrix cores).  Internal.
#pragma once
#include "ren_mlp_common.h"

namespace {

__device__ __forceinline__ float d2softplus_from_s(float s, float beta) { return beta * (1.f - s) * s; }

__device__ __forceinline__ float mul_scalar(float a, float b) {
    float r;
    asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// SH degree 4 and its directional derivative along dd; component 2j+hi -> out[j]
__device__ __forceinline__ void sh4_jvp_select(float x, float y, float z, float xd, float yd, float zd, int hi,
                                               float *out, float *outd) {
    // the pairwise products as explicit scalar multiplies (exactly x * y): left to the SLP vectoriser they become v_pk_mul_f32
    // with op_sel (see below)
    const float xy = mul_scalar(x, y), xz = mul_scalar(x, z), yz = mul_scalar(y, z), x2 = mul_scalar(x, x), y2 = mul_scalar(y, y),
                z2 = mul_scalar(z, z);
    const float A = 0.48860251190291987f, Bc = 1.0925484305920792f, C6 = 0.94617469575755997f,
                E = 0.54627421529603959f, F = 0.59004358992664352f, G = 2.8906114426405538f,
                H = 0.45704579946446572f, K = 0.3731763325901154f, M = 1.4453057213202769f;
    float s[16], t[16];
    s[0] = 0.28209479177387814f;                 t[0] = 0.f;
    s[1] = -A * y;                               t[1] = -A * yd;
    s[2] = A * z;                                t[2] = A * zd;
    s[3] = -A * x;                               t[3] = -A * xd;
    s[4] = Bc * xy;                              t[4] = Bc * fmaf(x, yd, mul_scalar(xd, y));
    s[5] = -Bc * yz;                             t[5] = -Bc * fmaf(y, zd, mul_scalar(yd, z));
    s[6] = C6 * z2 - 0.31539156525251999f;       t[6] = 2.f * C6 * z * zd;
    s[7] = -Bc * xz;                             t[7] = -Bc * fmaf(x, zd, mul_scalar(xd, z));
    const float xxd = mul_scalar(x, xd), yyd = mul_scalar(y, yd);
    s[8] = E * x2 - E * y2;                      t[8] = 2.f * E * (xxd - yyd);
    s[9] = F * y * (-3.f * x2 + y2);             t[9] = F * (yd * (-3.f * x2 + y2) + y * (-6.f * xxd + 2.f * yyd));
    s[10] = G * xy * z;                          t[10] = G * (xd * yz + x * yd * z + xy * zd);
    s[11] = H * y * (1.f - 5.f * z2);            t[11] = H * (yd * (1.f - 5.f * z2) - 10.f * y * z * zd);
    s[12] = K * z * (5.f * z2 - 3.f);            t[12] = K * zd * (15.f * z2 - 3.f);
    s[13] = H * x * (1.f - 5.f * z2);            t[13] = H * (xd * (1.f - 5.f * z2) - 10.f * x * z * zd);
    s[14] = M * z * (x2 - y2);                   t[14] = M * (zd * (x2 - y2) + z * (2.f * xxd - 2.f * yyd));
    s[15] = F * x * (-x2 + 3.f * y2);            t[15] = F * (xd * (-x2 + 3.f * y2) + x * (-2.f * xxd + 6.f * yyd));
    // scalar values, not SLP pairs: see sh4_select (packed FP32 with op_sel, tools/pkf32_hazard_repro.hip)
#pragma unroll
    for (int k = 1; k < 16; ++k) { asm volatile("" : "+v"(s[k])); asm volatile("" : "+v"(t[k])); }
#pragma unroll
    for (int j = 0; j < 8; ++j) { out[j] = hi ? s[2 * j + 1] : s[2 * j]; outd[j] = hi ? t[2 * j + 1] : t[2 * j]; }
}

struct RaySrc {                                   // packed sample stream with ray tangents
    const float *rays_o, *rays_d, *rays_dd;
    const int32_t *ray_indices;
    const float *t_starts, *t_ends;
};

__device__ __forceinline__ void geom_jvp(const RaySrc &s, const ren_scene_dev &sc, int64_t i, bool &sel, float *d,
                                         float *dd) {
    float x, y, z; int ray;
    ren_sample_pos(s.rays_o, s.rays_d, s.ray_indices, s.t_starts, s.t_ends, i, x, y, z, ray);
#pragma unroll
    for (int k = 0; k < 3; ++k) { d[k] = s.rays_d[3 * (int64_t)ray + k]; dd[k] = s.rays_dd[3 * (int64_t)ray + k]; }
    float ux, uy, uz;
    ren_contract(sc, x, y, z, ux, uy, uz);
    sel = ux > 0.f && ux < 1.f && uy > 0.f && uy < 1.f && uz > 0.f && uz < 1.f;
}

}  // namespace
