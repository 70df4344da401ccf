// Internal helpers shared by the gfx950 kernels.  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <stdio.h>
#include "../../include/ren_amd.h"

#define REN_WAVE 64

// hipGetLastError() is sticky per thread: PyTorch's own runtime calls may leave a benign error
// behind, so clear it right before every launch and report only what our launch produced.
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...)            \
    do {                                                                        \
        (void)hipGetLastError();                                                \
        kernel<<<(grid), (block), (shmem), (stream)>>>(__VA_ARGS__);            \
    } while (0)

#define REN_CHECK_LAUNCH()                                   \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess)                               \
            fprintf(stderr, "[ren_amd] %s:%d launch error: %s\n", __FILE__, __LINE__, hipGetErrorString(e__)); \
        return e__ == hipSuccess ? REN_OK : REN_ERR_LAUNCH;  \
    } while (0)

static inline int ren_blocks(int64_t n, int threads) { return (int)((n + threads - 1) / threads); }

// Verification / tuning knobs (include/ren_amd.h: ren_set_knob).  One process-wide atomic per knob, initialised ONCE from
// the environment variable of the same purpose: no getenv on the launch path (it is not thread-safe against setenv, and
// the data-parallel step launches from two streams).  None of them changes results.
#include <atomic>
#include <cstdlib>
inline std::atomic<int> *ren_knob_store() {
    static std::atomic<int> knobs[REN_KNOB_COUNT];
    static const bool init = [] {
        const char *names[REN_KNOB_COUNT] = {"REN_HGB_NO_PAIRS", "REN_HGB_HALVE_REGIONS", "REN_MARCH_SEQUENTIAL", "REN_HG_VARIANT",
                                           "REN_VFIELD_PLAIN", "REN_HGB_SUBREGION"};
        const int defaults[REN_KNOB_COUNT] = {0, 0, 0, 2, 0, 1};
        for (int k = 0; k < REN_KNOB_COUNT; ++k) {
            const char *e = getenv(names[k]);
            knobs[k].store(e ? atoi(e) : defaults[k], std::memory_order_relaxed);
        }
        return true;
    }();
    (void)init;
    return knobs;
}
static inline int ren_knob(int k) { return ren_knob_store()[k].load(std::memory_order_relaxed); }

struct ren_scene_dev {
    float lo[3], inv_ext[3];   // (x - lo) * inv_ext is NOT used where parity needs the division
    float hi[3];
    int ct;
};

static inline ren_scene_dev ren_make_scene(const ren_scene_desc *s) {
    ren_scene_dev d;
    for (int k = 0; k < 3; ++k) {
        d.lo[k] = s->aabb[k];
        d.hi[k] = s->aabb[3 + k];
        d.inv_ext[k] = 1.0f / (s->aabb[3 + k] - s->aabb[k]);
    }
    d.ct = s->contraction_type;
    return d;
}

// World -> unit cube, robust_e_nerf/external/ngp.py:230-237 (+68-106).
__device__ __forceinline__ void ren_contract(const ren_scene_dev &sc, float x, float y, float z,
                                             float &ux, float &uy, float &uz) {
    ux = (x - sc.lo[0]) / (sc.hi[0] - sc.lo[0]);
    uy = (y - sc.lo[1]) / (sc.hi[1] - sc.lo[1]);
    uz = (z - sc.lo[2]) / (sc.hi[2] - sc.lo[2]);
    if (sc.ct == REN_CT_SPHERE) {
        ux = ux * 2.f - 1.f; uy = uy * 2.f - 1.f; uz = uz * 2.f - 1.f;
        float mag = sqrtf(ux * ux + uy * uy + uz * uz);
        if (mag > 1.f) {
            float s = (2.f - 1.f / mag) / mag;
            ux *= s; uy *= s; uz *= s;
        }
        ux = ux * 0.25f + 0.5f; uy = uy * 0.25f + 0.5f; uz = uz * 0.25f + 0.5f;
    } else if (sc.ct == REN_CT_TANH) {
        ux = (tanhf(ux - 0.5f) + 1.f) * 0.5f;
        uy = (tanhf(uy - 0.5f) + 1.f) * 0.5f;
        uz = (tanhf(uz - 0.5f) + 1.f) * 0.5f;
    }
}

// Sample position of packed sample i: o + d * (t0 + t1) / 2  (external/utils.py:68-72)
__device__ __forceinline__ void ren_sample_pos(const float *__restrict__ rays_o,
                                               const float *__restrict__ rays_d,
                                               const int32_t *__restrict__ ray_indices,
                                               const float *__restrict__ t_starts,
                                               const float *__restrict__ t_ends, int64_t i,
                                               float &x, float &y, float &z, int &ray) {
    ray = ray_indices[i];
    float tm = (t_starts[i] + t_ends[i]) * 0.5f;
    const float *o = rays_o + 3 * (int64_t)ray, *d = rays_d + 3 * (int64_t)ray;
    x = o[0] + d[0] * tm;
    y = o[1] + d[1] * tm;
    z = o[2] + d[2] * tm;
}

// Device-side sample count (ABI 24, `n_dev` arguments): the host passes the CAPACITY of the per-sample arrays as `n` and, in
// `n_dev`, where the device keeps the number of samples that exist (written by the sampler's scan / ren_count_guard): the
// kernels of a render are enqueued without the host reading that number back (models/nerf.py:279-286 and external/utils.py:
// 106-119 read it).  NULL = `n` is the count, as before.
__device__ __forceinline__ int64_t ren_eff_n(int64_t n, const int64_t *__restrict__ n_dev) {
    if (!n_dev) return n;
    const int64_t m = *n_dev;
    return m < n ? (m < 0 ? 0 : m) : n;
}

__device__ __forceinline__ float ren_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
