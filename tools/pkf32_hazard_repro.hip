// Stand-alone reproducer attempt for the "packed-FP32 beside matrix-core waves" hazard of round 4 (profiles/NOTES.md):
// one .hip, hipcc, no torch, no library.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/pkf32_hazard_repro.hip -o /tmp/pkrepro            (packed code from the SLP vectoriser)
//   hipcc ... -fno-slp-vectorize ...                                                                    (control: no v_pk_*_f32 in victim C)
//   /tmp/pkrepro [launches per cell, default 300] [path/to/libren_amd.so]
// With a library path (a build of the library WITH packed code in ren_pose.hip, i.e. that file compiled without
// -fno-slp-vectorize) the library's own kernels join through its C ABI -- still no torch, hipMalloc'd buffers, two plain HIP
// streams: victim L = ren_pose_rays_fwd, aggressor "lib mlp_fwd_x" = ren_mlp_fwd_x (mode 6, 2 M samples).  The 2 x 2 matrix
// {clone, library} victim x {synthetic, library} aggressor tells which side carries the effect.
// VICTIMS (one thread per element, 256-thread workgroups, launched on a second, high-priority stream while the aggressor runs;
// every launch's output is compared bit for bit with the same kernel's output on an idle chip):
//   A  a dependent chain of 64 v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 per thread (inline asm: the instruction is certain)
//   B  the same arithmetic as scalar v_fma_f32 / v_mul_f32 / v_add_f32 (control)
//   C  a verbatim clone of the library's pose_rays_kernel (csrc/ren_pose.hip: binary search over int64 timestamps in f64, LERP,
//      quaternion SLERP with atan2f / sinf / cosf, rotation matrix, ray) -- the kernel that went wrong in the library.  Whether
//      it contains packed code depends on the build flags above.
//   D  victim A with an s_nop 4 in front of every packed instruction
// AGGRESSORS (persistent grid: 2 workgroups of 8 waves per CU, like mlp_fwd_x):
//   none | mfma: one 64 -> 64 split-bf16 layer per iteration (48 v_mfma_f32_32x32x16_bf16, weights via ds_read_b128, softplus +
//   3-piece split on the VALU -- the loop of tools/phase_overlap_bench.hip) | mfma_only: the 48 MFMAs without the VALU phase |
//   valu: the VALU phase without MFMAs | mem: a streaming copy
// Printed per (victim, aggressor): launches with at least one wrong element, wrong elements, and how many of the wrong elements
// sit in fully-wrong ALIGNED groups of 16 consecutive threads (the signature seen in the library: one 16-lane pass of a wave).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <dlfcn.h>
#include "../include/ren_amd.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MFMAB(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

// ---------------------------------------------------------------------------------------------------- aggressors
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

template <bool DO_M, bool DO_V>
__global__ __launch_bounds__(512, 4) void aggr_layer(float *out, int iters) {
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
                    a0 = MFMAB(W[((0 * 4 + c) * 3 + W6[k]) * 64 + lane], b[c][A6[k]], a0);
                    a1 = MFMAB(W[((1 * 4 + c) * 3 + W6[k]) * 64 + lane], b[c][A6[k]], a1);
                }
        } else {
#pragma unroll
            for (int g = 0; g < 16; ++g) { a0[g] += (float)b[g & 3][0][g >> 1]; a1[g] -= (float)b[g & 3][1][g >> 1]; }
        }
        if (DO_V) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float y[16];
#pragma unroll
                for (int g = 0; g < 16; ++g) y[g] = softplus100(t ? a1[g] : a0[g]);
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
}

__global__ void aggr_mem(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n4, int passes) {
    for (int p = 0; p < passes; ++p)
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

// ---------------------------------------------------------------------------------------------------- victims A / B / D
// 64 dependent steps on a pair (x, y): every step one packed (or two scalar) fma / mul / add with per-thread constants.
template <int KIND>      // 0: packed (A), 1: scalar (B), 2: packed with s_nop 4 in front of each (D)
__global__ void victim_chain(const float *__restrict__ in, float *__restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    v2f acc = {in[4 * i], in[4 * i + 1]}, m = {in[4 * i + 2], in[4 * i + 3]}, c = {0.75f, -0.5f}, h = {0.5f, 0.25f};
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        if (KIND == 1) {
            float ax = acc.x, ay = acc.y;          // scalar instructions, whatever the vectoriser would like
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(ax) : "v"(m.x), "v"(c.x)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(ay) : "v"(m.y), "v"(c.y));
            asm volatile("v_mul_f32 %0, %0, %1" : "+v"(ax) : "v"(h.x));               asm volatile("v_mul_f32 %0, %0, %1" : "+v"(ay) : "v"(h.y));
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(ax) : "v"(m.x));               asm volatile("v_add_f32 %0, %0, %1" : "+v"(ay) : "v"(m.y));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(ax) : "v"(h.x), "v"(m.x)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(ay) : "v"(h.y), "v"(m.y));
            acc.x = ax; acc.y = ay;
        } else {
            if (KIND == 2) asm volatile("s_nop 4");
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc) : "v"(m), "v"(c));
            if (KIND == 2) asm volatile("s_nop 4");
            asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(acc) : "v"(h));
            if (KIND == 2) asm volatile("s_nop 4");
            asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc) : "v"(m));
            if (KIND == 2) asm volatile("s_nop 4");
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc) : "v"(h), "v"(m));
        }
    }
    out[2 * i] = acc.x;
    out[2 * i + 1] = acc.y;
}

// E: packed instructions with an SGPR PAIR as a source (what hipcc makes of `Kinv[...] * u` in victim C: the matrix sits in SGPRs);
// F: the same with op_sel / op_sel_hi on the SGPR pair; G: op_sel / op_sel_hi / neg modifiers on VGPR sources only
template <int KIND>      // 0: E, 1: F, 2: G, 3 / 4 / 5: G's modifiers one kind at a time
__global__ void victim_chain_mod(const float *__restrict__ in, float *__restrict__ out, int n, v2f k1, v2f k2, v2f k3) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    v2f acc = {in[4 * i], in[4 * i + 1]}, m = {in[4 * i + 2], in[4 * i + 3]};
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        if (KIND == 0) {
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc) : "s"(k1), "v"(m));
            asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(acc) : "s"(k2));
            asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(acc) : "s"(k3));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc) : "s"(k2), "v"(m));
        } else if (KIND == 1) {
            asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[1,0] op_sel_hi:[0,1]" : "+v"(acc) : "s"(k2));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc) : "s"(k1), "v"(m));
            asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(acc) : "s"(k2));
            asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(acc) : "s"(k3));
        } else if (KIND == 3) {                    // op_sel only: the LOW result half reads a source's HIGH half
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,1,0]" : "+v"(acc) : "v"(m), "v"(m));
            asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[1,0]" : "+v"(acc) : "v"(m));
            asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1]" : "+v"(acc) : "v"(m));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,0,1]" : "+v"(acc) : "v"(m), "v"(m));
        } else if (KIND == 4) {                    // op_sel_hi only: the HIGH result half reads a source's LOW half (broadcast)
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(m), "v"(m));
            asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[0,1]" : "+v"(acc) : "v"(m));
            asm volatile("v_pk_add_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(acc) : "v"(m));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(m), "v"(m));
        } else if (KIND == 5) {                    // neg modifiers only
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "+v"(acc) : "v"(m), "v"(m));
            asm volatile("v_pk_mul_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(acc) : "v"(m));
            asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(acc) : "v"(m));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "+v"(acc) : "v"(m), "v"(m));
        } else {
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,1,0] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "+v"(acc) : "v"(m), "v"(m));
            asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[1,0] op_sel_hi:[0,1]" : "+v"(acc) : "v"(m));
            asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(acc) : "v"(m));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,0,1] op_sel_hi:[1,1,0]" : "+v"(acc) : "v"(m), "v"(m));
        }
        acc.x = __builtin_amdgcn_fmed3f(acc.x, -4.f, 4.f); acc.y = __builtin_amdgcn_fmed3f(acc.y, -4.f, 4.f);      // keep it finite
    }
    out[2 * i] = acc.x;
    out[2 * i + 1] = acc.y;
}

// ---------------------------------------------------------------------------------------------------- victim C
// clone of csrc/ren_pose.hip (pose_eval + ray_eval + pose_rays_kernel), unchanged arithmetic
struct Quat { float x, y, z, w; };
__device__ __forceinline__ Quat qmul(const Quat &p, const Quat &q) {
    Quat r;
    r.x = p.w * q.x + q.w * p.x + (p.y * q.z - p.z * q.y);
    r.y = p.w * q.y + q.w * p.y + (p.z * q.x - p.x * q.z);
    r.z = p.w * q.z + q.w * p.z + (p.x * q.y - p.y * q.x);
    r.w = p.w * q.w - (p.x * q.x + p.y * q.y + p.z * q.z);
    return r;
}
__device__ __forceinline__ float lerpf(float a, float b, float w) { return fabsf(w) < 0.5f ? a + w * (b - a) : b - (b - a) * (1.f - w); }
__device__ __forceinline__ void pose_eval(double t, const int64_t *__restrict__ tab_ts, const float *__restrict__ tab_pos,
                                          const float *__restrict__ tab_quat, int64_t C, float *p, float *R) {
    int64_t lo = 0, hi = C;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if ((double)tab_ts[mid] < t) lo = mid + 1; else hi = mid;
    }
    int64_t right = lo < C ? lo : C - 1;
    int64_t left = (t == (double)tab_ts[0]) ? right : right - 1;
    if (left < 0) left = 0;
    int64_t wbin = left < C - 1 ? left : C - 2;
    const float w = (float)((t - (double)tab_ts[left]) / (double)(tab_ts[wbin + 1] - tab_ts[wbin]));
    for (int k = 0; k < 3; ++k) p[k] = lerpf(tab_pos[3 * left + k], tab_pos[3 * right + k], w);
    Quat q0 = {tab_quat[4 * left], tab_quat[4 * left + 1], tab_quat[4 * left + 2], tab_quat[4 * left + 3]};
    Quat q1 = {tab_quat[4 * right], tab_quat[4 * right + 1], tab_quat[4 * right + 2], tab_quat[4 * right + 3]};
    float dot = q0.x * q1.x + q0.y * q1.y + q0.z * q1.z + q0.w * q1.w;
    if (dot < 0.f) { q1.x = -q1.x; q1.y = -q1.y; q1.z = -q1.z; q1.w = -q1.w; }
    Quat c0 = {-q0.x, -q0.y, -q0.z, q0.w};
    Quat rel = qmul(c0, q1);
    float vn = sqrtf(rel.x * rel.x + rel.y * rel.y + rel.z * rel.z);
    float angle = 2.f * atan2f(vn, rel.w);
    float a2 = angle * angle;
    float scale = fabsf(angle) <= 1e-3f ? 2.f + a2 / 12.f + 7.f * a2 * a2 / 2880.f : angle / sinf(angle * 0.5f);
    float rx = w * scale * rel.x, ry = w * scale * rel.y, rz = w * scale * rel.z;
    float th = sqrtf(rx * rx + ry * ry + rz * rz);
    float t2 = th * th;
    float s = th <= 1e-3f ? 0.5f - t2 / 48.f + t2 * t2 / 3840.f : sinf(th * 0.5f) / th;
    Quat rq = {s * rx, s * ry, s * rz, cosf(th * 0.5f)};
    Quat q = qmul(q0, rq);
    float x2 = q.x * q.x, y2 = q.y * q.y, z2 = q.z * q.z, w2 = q.w * q.w;
    float xy = q.x * q.y, zw = q.z * q.w, xz = q.x * q.z, yw = q.y * q.w, yz = q.y * q.z, xw = q.x * q.w;
    R[0] = x2 - y2 - z2 + w2; R[1] = 2.f * (xy - zw);     R[2] = 2.f * (xz + yw);
    R[3] = 2.f * (xy + zw);   R[4] = -x2 + y2 - z2 + w2;  R[5] = 2.f * (yz - xw);
    R[6] = 2.f * (xz - yw);   R[7] = 2.f * (yz + xw);     R[8] = -x2 - y2 + z2 + w2;
}
__device__ __forceinline__ void ray_eval(const float *__restrict__ Kinv, float u, float v, const float *p, const float *R, float *o, float *d) {
    float k0 = Kinv[0] * u + Kinv[1] * v + Kinv[2];
    float k1 = Kinv[3] * u + Kinv[4] * v + Kinv[5];
    float k2 = Kinv[6] * u + Kinv[7] * v + Kinv[8];
    float dx = R[0] * k0 + R[1] * k1 + R[2] * k2;
    float dy = R[3] * k0 + R[4] * k1 + R[5] * k2;
    float dz = R[6] * k0 + R[7] * k1 + R[8] * k2;
    float inv = 1.f / sqrtf(dx * dx + dy * dy + dz * dz);
    d[0] = dx * inv; d[1] = dy * inv; d[2] = dz * inv;
    o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
}
// KV: the camera matrix goes through VGPRs (opaque copies) instead of the SGPRs its uniform loads land in
template <bool KV>
__global__ void victim_pose(const double *__restrict__ ts, int64_t R_, const float *__restrict__ px, int64_t px_rows,
                            const float *__restrict__ Kinv, const int64_t *__restrict__ tab_ts, const float *__restrict__ tab_pos,
                            const float *__restrict__ tab_quat, int64_t C, float *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R_) return;
    float p[3], R[9];
    pose_eval(ts[i], tab_ts, tab_pos, tab_quat, C, p, R);
    const int64_t j = i % px_rows;
    if (KV) {
        float K[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) { K[k] = Kinv[k]; asm volatile("" : "+v"(K[k])); }
        ray_eval(K, px[2 * j], px[2 * j + 1], p, R, out + 6 * i, out + 6 * i + 3);
    } else {
        ray_eval(Kinv, px[2 * j], px[2 * j + 1], p, R, out + 6 * i, out + 6 * i + 3);
    }
}

// ---------------------------------------------------------------------------------------------------- host
static uint32_t rng = 12345u;
static float frand() { rng = rng * 1664525u + 1013904223u; return (rng >> 8) * (1.f / 16777216.f); }

struct Tally { int bad_launches = 0; long wrong = 0, wrong_in_full16 = 0; };
static void compare(const float *got, const float *ref, int n_thr, int per, Tally &t) {
    long wrong = 0, in16 = 0;
    for (int g = 0; g < n_thr; g += 16) {
        int w = 0;
        for (int i = g; i < g + 16 && i < n_thr; ++i) w += memcmp(got + (size_t)i * per, ref + (size_t)i * per, per * 4) != 0;
        wrong += w;
        if (w == 16) in16 += 16;
    }
    t.bad_launches += wrong > 0;
    t.wrong += wrong;
    t.wrong_in_full16 += in16;
}

int main(int argc, char **argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 300;
    hipStream_t sa, sv;
    int lo_p, hi_p;
    CK(hipDeviceGetStreamPriorityRange(&lo_p, &hi_p));
    CK(hipStreamCreateWithPriority(&sa, hipStreamNonBlocking, lo_p));
    CK(hipStreamCreateWithPriority(&sv, hipStreamNonBlocking, hi_p));
    // ---- aggressor buffers
    float *aout; CK(hipMalloc(&aout, sizeof(float) * 512 * 512));
    const size_t n4 = (size_t)64 << 20;                         // 1 GiB copy
    float4 *msrc, *mdst; CK(hipMalloc(&msrc, n4 * 16)); CK(hipMalloc(&mdst, n4 * 16)); CK(hipMemset(msrc, 1, n4 * 16));
    const size_t lds = 2 * 4 * 3 * 64 * 16;
    // ---- victims A / B / D
    const int NT = 8192;
    std::vector<float> h_in(4 * NT);
    for (auto &v : h_in) v = frand() * 1.5f - 0.75f;
    float *d_in, *d_out; CK(hipMalloc(&d_in, 4 * NT * 4)); CK(hipMalloc(&d_out, 6 * NT * 4));
    CK(hipMemcpy(d_in, h_in.data(), 4 * NT * 4, hipMemcpyHostToDevice));
    // ---- victim C: a trajectory of 1 000 control poses, 8 192 timestamps, 4 096 pixels
    const int CP = 1000;
    std::vector<int64_t> tab_ts(CP); std::vector<float> tab_pos(3 * CP), tab_quat(4 * CP), px(2 * 4096), Kinv = {0.004f, 0, -0.69f, 0, 0.004f, -0.52f, 0, 0, 1};
    std::vector<double> ts(NT);
    for (int i = 0; i < CP; ++i) {
        tab_ts[i] = (int64_t)i * 1000000 + (int64_t)(frand() * 1000);
        const double a = 0.002 * i;
        tab_pos[3 * i] = 4 * cos(a); tab_pos[3 * i + 1] = 4 * sin(a); tab_pos[3 * i + 2] = 0.5f * sinf(3 * a);
        float q[4] = {0.3f * sinf(a), 0.2f * cosf(2 * a), sinf(0.5f * a), cosf(0.5f * a)}, nq = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        for (int k = 0; k < 4; ++k) tab_quat[4 * i + k] = q[k] / nq;
    }
    for (auto &t : ts) t = frand() * (double)tab_ts[CP - 1];
    for (int i = 0; i < 4096; ++i) { px[2 * i] = floorf(frand() * 346); px[2 * i + 1] = floorf(frand() * 260); }
    int64_t *d_tts; float *d_tp, *d_tq, *d_px, *d_K; double *d_ts;
    CK(hipMalloc(&d_tts, CP * 8)); CK(hipMalloc(&d_tp, CP * 12)); CK(hipMalloc(&d_tq, CP * 16)); CK(hipMalloc(&d_px, 4096 * 8)); CK(hipMalloc(&d_K, 36)); CK(hipMalloc(&d_ts, NT * 8));
    CK(hipMemcpy(d_tts, tab_ts.data(), CP * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_tp, tab_pos.data(), CP * 12, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_tq, tab_quat.data(), CP * 16, hipMemcpyHostToDevice)); CK(hipMemcpy(d_px, px.data(), 4096 * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_K, Kinv.data(), 36, hipMemcpyHostToDevice)); CK(hipMemcpy(d_ts, ts.data(), NT * 8, hipMemcpyHostToDevice));

    // ---- optional: the library's kernels through its C ABI
    typedef int (*pose_fn)(const double *, int64_t, const float *, int64_t, const float *, const int64_t *, const float *, const float *, int64_t, float *, float *, void *);
    typedef int (*mlp_fn)(const float *, int32_t, int32_t, int32_t, const float *, const ren_scene_desc *, const float *, const float *, const float *, const float *,
                          const int32_t *, const float *, const float *, int64_t, int32_t, float *, float *, float *, float *, void *);
    pose_fn lib_pose = nullptr; mlp_fn lib_mlp = nullptr;
    const int64_t LN = 2097152;
    float *l_par = nullptr, *l_feat = nullptr, *l_x = nullptr, *l_d = nullptr, *l_rgb = nullptr, *l_sig = nullptr, *l_base = nullptr;
    ren_scene_desc scene = {{-1.5f, -1.5f, -1.5f, 1.5f, 1.5f, 1.5f}, 0};
    if (argc > 2) {
        void *h = dlopen(argv[2], RTLD_NOW);
        if (!h) { printf("dlopen %s failed: %s\n", argv[2], dlerror()); return 1; }
        lib_pose = (pose_fn)dlsym(h, "ren_pose_rays_fwd"); lib_mlp = (mlp_fn)dlsym(h, "ren_mlp_fwd_x");
        if (!lib_pose || !lib_mlp) { printf("symbols missing\n"); return 1; }
        auto fill = [&](float **d, size_t n, float lo, float hi) {
            std::vector<float> hbuf(n);
            for (auto &v : hbuf) v = lo + (hi - lo) * frand();
            CK(hipMalloc(d, n * 4)); CK(hipMemcpy(*d, hbuf.data(), n * 4, hipMemcpyHostToDevice));
        };
        fill(&l_par, 9425, -0.25f, 0.25f); fill(&l_feat, LN / 32 * 1024, -0.5f, 0.5f); fill(&l_x, LN * 3, -1.3f, 1.3f);
        {
            std::vector<float> hd(LN * 3);
            for (int64_t i = 0; i < LN; ++i) {
                float a = frand() * 2 - 1, b = frand() * 2 - 1, c = frand() * 2 - 1, nrm = sqrtf(a * a + b * b + c * c) + 1e-6f;
                hd[3 * i] = a / nrm; hd[3 * i + 1] = b / nrm; hd[3 * i + 2] = c / nrm;
            }
            CK(hipMalloc(&l_d, LN * 12)); CK(hipMemcpy(l_d, hd.data(), LN * 12, hipMemcpyHostToDevice));
        }
        CK(hipMalloc(&l_rgb, LN * 4)); CK(hipMalloc(&l_sig, LN * 4)); CK(hipMalloc(&l_base, LN / 32 * 512 * 4));
    }

    auto run_victim = [&](int v) {
        const dim3 g((NT + 255) / 256), b(256);
        switch (v) {
            case 0: hipLaunchKernelGGL(victim_chain<0>, g, b, 0, sv, d_in, d_out, NT); break;
            case 1: hipLaunchKernelGGL(victim_chain<1>, g, b, 0, sv, d_in, d_out, NT); break;
            case 2: hipLaunchKernelGGL(victim_pose<false>, g, b, 0, sv, d_ts, (int64_t)NT, d_px, (int64_t)4096, d_K, d_tts, d_tp, d_tq, (int64_t)CP, d_out); break;
            case 3: hipLaunchKernelGGL(victim_chain<2>, g, b, 0, sv, d_in, d_out, NT); break;
            case 5: hipLaunchKernelGGL(victim_chain_mod<0>, g, b, 0, sv, d_in, d_out, NT, v2f{0.75f, -0.5f}, v2f{0.5f, 0.25f}, v2f{0.125f, -0.25f}); break;
            case 6: hipLaunchKernelGGL(victim_chain_mod<1>, g, b, 0, sv, d_in, d_out, NT, v2f{0.75f, -0.5f}, v2f{0.5f, 0.25f}, v2f{0.125f, -0.25f}); break;
            case 7: hipLaunchKernelGGL(victim_chain_mod<2>, g, b, 0, sv, d_in, d_out, NT, v2f{0.75f, -0.5f}, v2f{0.5f, 0.25f}, v2f{0.125f, -0.25f}); break;
            case 9: hipLaunchKernelGGL(victim_chain_mod<3>, g, b, 0, sv, d_in, d_out, NT, v2f{0.75f, -0.5f}, v2f{0.5f, 0.25f}, v2f{0.125f, -0.25f}); break;
            case 10: hipLaunchKernelGGL(victim_chain_mod<4>, g, b, 0, sv, d_in, d_out, NT, v2f{0.75f, -0.5f}, v2f{0.5f, 0.25f}, v2f{0.125f, -0.25f}); break;
            case 11: hipLaunchKernelGGL(victim_chain_mod<5>, g, b, 0, sv, d_in, d_out, NT, v2f{0.75f, -0.5f}, v2f{0.5f, 0.25f}, v2f{0.125f, -0.25f}); break;
            case 8: hipLaunchKernelGGL(victim_pose<true>, g, b, 0, sv, d_ts, (int64_t)NT, d_px, (int64_t)4096, d_K, d_tts, d_tp, d_tq, (int64_t)CP, d_out); break;
            case 4: if (!lib_pose) break; if (lib_pose(d_ts, NT, d_px, 4096, d_K, d_tts, d_tp, d_tq, CP, d_out, d_out + 3 * NT, sv) != 0) { printf("ren_pose_rays_fwd failed\n"); exit(1); } break;
        }
    };
    auto run_aggr = [&](int a) {
        switch (a) {
            case 0: break;
            case 1: hipLaunchKernelGGL((aggr_layer<true, true>), dim3(512), dim3(512), lds, sa, aout, 1500); break;
            case 2: hipLaunchKernelGGL((aggr_layer<true, false>), dim3(512), dim3(512), lds, sa, aout, 2500); break;
            case 3: hipLaunchKernelGGL((aggr_layer<false, true>), dim3(512), dim3(512), lds, sa, aout, 2500); break;
            case 4: hipLaunchKernelGGL(aggr_mem, dim3(2048), dim3(256), 0, sa, msrc, mdst, n4, 2); break;
            case 5: for (int k = 0; k < 6; ++k)
                        if (lib_mlp(l_par, 1, 0, 6, l_feat, &scene, l_x, l_d, nullptr, nullptr, nullptr, nullptr, nullptr, LN, 0, l_rgb, l_sig, l_base, nullptr, sa) != 0) {
                            printf("ren_mlp_fwd_x failed\n"); exit(1);
                        }
                    break;
        }
    };
    const char *vn[12] = {"A packed chain (asm v_pk_*_f32)", "B scalar chain (control)", "C pose_rays clone", "D packed chain + s_nop 4", "L library ren_pose_rays_fwd",
                         "E packed chain, SGPR-pair sources", "F packed, SGPR pair + op_sel", "G packed, VGPR op_sel/neg", "K pose clone, Kinv via VGPRs",
                         "G1 packed, op_sel only", "G2 packed, op_sel_hi only", "G3 packed, neg only"};
    const char *an[6] = {"none", "mfma layer (MFMA + VALU)", "mfma only", "valu only", "memory copy", "lib mlp_fwd_x"};
    const int per[12] = {2, 2, 6, 2, 6, 2, 2, 2, 6, 2, 2, 2};
    const int n_v = 12, n_a = lib_mlp ? 6 : 5;
    std::vector<float> ref(6 * NT), got(6 * NT);
    printf("%d launches per cell; victim = %d threads\n", launches, NT);
    for (int v = 0; v < n_v; ++v) {
        if (v == 4 && !lib_pose) continue;
        if (getenv("VICTIMS") && !strchr(getenv("VICTIMS"), "ABCDLEFGK123"[v])) continue;
        CK(hipDeviceSynchronize());
        run_victim(v);
        CK(hipStreamSynchronize(sv));
        CK(hipMemcpy(ref.data(), d_out, (size_t)per[v] * NT * 4, hipMemcpyDeviceToHost));
        for (int a = 0; a < n_a; ++a) {
            Tally t;
            for (int it = 0; it < launches; ++it) {
                if (it % 4 == 0) { CK(hipStreamSynchronize(sa)); run_aggr(a); }          // the aggressor runs for a few ms: several victims per run
                CK(hipMemsetAsync(d_out, 0xff, (size_t)per[v] * NT * 4, sv));
                run_victim(v);
                CK(hipMemcpyAsync(got.data(), d_out, (size_t)per[v] * NT * 4, hipMemcpyDeviceToHost, sv));
                CK(hipStreamSynchronize(sv));
                if (v == 4) {                                  // o and d are separate (NT, 3) arrays
                    Tally to, td;
                    compare(got.data(), ref.data(), NT, 3, to); compare(got.data() + 3 * NT, ref.data() + 3 * NT, NT, 3, td);
                    t.bad_launches += (to.bad_launches | td.bad_launches); t.wrong += to.wrong + td.wrong; t.wrong_in_full16 += to.wrong_in_full16 + td.wrong_in_full16;
                } else compare(got.data(), ref.data(), NT, per[v], t);
            }
            CK(hipDeviceSynchronize());
            printf("victim %-34s beside %-26s: %4d of %d launches wrong, %6ld wrong elements, %6ld of them in fully wrong aligned 16-groups\n",
                   vn[v], an[a], t.bad_launches, launches, t.wrong, t.wrong_in_full16);
            fflush(stdout);
        }
    }
    return 0;
}
