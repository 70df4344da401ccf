/* Oracle (TEST INFRASTRUCTURE ONLY): sequential CPU restatement of the ray-sampling
 * half of the hot path -- nerfacc==0.3.1 `ray_aabb_intersect`, `ray_marching` (two
 * passes), and `render_visibility`, which the reference reaches through
 *   robust_e_nerf/external/utils.py:106-119   (ray_marching(...) call, sigma_fn pre-pass)
 *   robust_e_nerf/models/nerf.py:248-251      (scene_aabb only for ContractionType.AABB)
 * nerfacc is an un-vendored dependency (environment.yml:30); this follows its published
 * CUDA kernels (csrc/intersection.cu, csrc/ray_marching.cu, csrc/render_transmittance.cu)
 * as recalled in SURVEY.md App. A.1.  PARITY UNPINNED (the reference holds no test or
 * golden vector for it).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC (oracle/build.py).
 * All arithmetic is float32, un-contracted, in the operation order the HIP kernels use,
 * so sample counts and interval endpoints are expected to match bit-for-bit.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

enum { CT_AABB = 0, CT_TANH = 1, CT_SPHERE = 2 };

static inline float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* nerfacc csrc/intersection.cu: slab test; miss => near = far = 1e10; near clamped >= 0 */
void orc_ray_aabb_intersect(int64_t n, const float *o, const float *d, const float *aabb,
                            float *t_min, float *t_max) {
    for (int64_t i = 0; i < n; ++i) {
        const float *ro = o + 3 * i, *rd = d + 3 * i;
        float tmin = (aabb[0] - ro[0]) / rd[0], tmax = (aabb[3] - ro[0]) / rd[0];
        if (tmin > tmax) { float t = tmin; tmin = tmax; tmax = t; }
        float tymin = (aabb[1] - ro[1]) / rd[1], tymax = (aabb[4] - ro[1]) / rd[1];
        if (tymin > tymax) { float t = tymin; tymin = tymax; tymax = t; }
        int miss = 0;
        if (tmin > tymax || tymin > tmax) miss = 1;
        if (!miss) {
            if (tymin > tmin) tmin = tymin;
            if (tymax < tmax) tmax = tymax;
            float tzmin = (aabb[2] - ro[2]) / rd[2], tzmax = (aabb[5] - ro[2]) / rd[2];
            if (tzmin > tzmax) { float t = tzmin; tzmin = tzmax; tzmax = t; }
            if (tmin > tzmax || tzmin > tmax) miss = 1;
            if (!miss) {
                if (tzmin > tmin) tmin = tzmin;
                if (tzmax < tmax) tmax = tzmax;
            }
        }
        if (miss) { t_min[i] = 1e10f; t_max[i] = 1e10f; }
        else { t_min[i] = tmin > 0.f ? tmin : 0.f; t_max[i] = tmax; }
    }
}

static inline void roi_to_unit(const float *p, const float *roi, float *u) {
    for (int k = 0; k < 3; ++k) u[k] = (p[k] - roi[k]) / (roi[3 + k] - roi[k]);
}

static inline void apply_contraction(const float *p, const float *roi, int type, float *u) {
    roi_to_unit(p, roi, u);
    if (type == CT_SPHERE) {
        for (int k = 0; k < 3; ++k) u[k] = u[k] * 2.f - 1.f;
        float norm = sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
        if (norm > 1.f) {
            float s = (2.f - 1.f / norm);
            for (int k = 0; k < 3; ++k) u[k] = s * (u[k] / norm);
        }
        for (int k = 0; k < 3; ++k) u[k] = u[k] * 0.25f + 0.5f;
    } else if (type == CT_TANH) {
        for (int k = 0; k < 3; ++k) u[k] = tanhf(u[k] - 0.5f) * 0.5f + 0.5f;
    }
}

static inline int grid_occupied_at(const float *p, const float *roi, int type, const int *res,
                                   const uint8_t *binary) {
    if (type == CT_AABB) {
        for (int k = 0; k < 3; ++k)
            if (p[k] < roi[k] || p[k] > roi[3 + k]) return 0;
    }
    float u[3];
    apply_contraction(p, roi, type, u);
    int idx = 0;
    for (int k = 0; k < 3; ++k) {
        int c = (int)(u[k] * (float)res[k]);
        c = c < 0 ? 0 : (c > res[k] - 1 ? res[k] - 1 : c);
        idx = idx * res[k] + c;               /* x-major: ix*ry*rz + iy*rz + iz */
    }
    return binary[idx] != 0;
}

static inline float sgnf(float v) { return (v > 0.f) - (v < 0.f); }

static inline float distance_to_next_voxel(const float *p, const float *dir, const float *inv_dir,
                                           const float *roi, const int *res) {
    float u[3], t = 1e30f;
    roi_to_unit(p, roi, u);
    for (int k = 0; k < 3; ++k) {
        float r = (float)res[k];
        float x = u[k] * r;
        float tx = ((floorf(x + 0.5f + 0.5f * sgnf(dir[k])) - x) * inv_dir[k]) / r * (roi[3 + k] - roi[k]);
        if (tx < t) t = tx;
    }
    return t > 0.f ? t : 0.f;
}

static inline float calc_dt(float t, float cone_angle, float dt_min, float dt_max) {
    return clampf(t * cone_angle, dt_min, dt_max);
}

/* nerfacc csrc/ray_marching.cu.  counts[i] = number of samples of ray i.  When
 * t_starts != NULL this is the write pass and `offsets` (exclusive cumsum of counts)
 * gives each ray's first packed slot.
 * mode 0 = occupancy-grid marching (reference semantics);
 * mode 1 = uniform comb: exactly n_uniform intervals of width delta=(far-near)/n_uniform
 *          starting at near + jitter*delta (the build's fixed-S sampler, SURVEY 8(d)
 *          mode (i)); rays with near >= far (AABB miss) get no samples. */
void orc_ray_march(int64_t n_rays, const float *o, const float *d, const float *t_min,
                   const float *t_max, const float *roi, const int *res, const uint8_t *binary,
                   int type, float step_size, float cone_angle, int mode, int n_uniform,
                   const float *jitter, const int64_t *offsets, int32_t *counts, float *t_starts, float *t_ends,
                   int32_t *ray_indices) {
    for (int64_t i = 0; i < n_rays; ++i) {
        const float *ro = o + 3 * i, *rd = d + 3 * i;
        float inv_dir[3] = {1.f / rd[0], 1.f / rd[1], 1.f / rd[2]};
        const float near = t_min[i], far = t_max[i];
        int64_t base = t_starts ? offsets[i] : 0;
        int j = 0;
        if (mode == 1) {
            if (near < far) {
                float delta = (far - near) / (float)n_uniform;
                float first = jitter ? near + jitter[i] * delta : near;
                for (j = 0; j < n_uniform; ++j) {
                    if (t_starts) {
                        float t0 = first + (float)j * delta;
                        t_starts[base + j] = t0;
                        t_ends[base + j] = t0 + delta;
                        ray_indices[base + j] = (int32_t)i;
                    }
                }
            }
            if (!t_starts) counts[i] = j;
            continue;
        }
        const float dt_min = step_size, dt_max = 1e10f;
        float t0 = near;
        float dt = calc_dt(t0, cone_angle, dt_min, dt_max);
        float t1 = t0 + dt;
        float t_mid = (t0 + t1) * 0.5f;
        while (t_mid < far) {
            float p[3] = {ro[0] + t_mid * rd[0], ro[1] + t_mid * rd[1], ro[2] + t_mid * rd[2]};
            if (grid_occupied_at(p, roi, type, res, binary)) {
                if (t_starts) {
                    t_starts[base + j] = t0;
                    t_ends[base + j] = t1;
                    ray_indices[base + j] = (int32_t)i;
                }
                ++j;
                t0 = t1;
                t1 = t0 + calc_dt(t0, cone_angle, dt_min, dt_max);
                t_mid = (t0 + t1) * 0.5f;
            } else if (type == CT_AABB) {
                float t_target = t_mid + distance_to_next_voxel(p, rd, inv_dir, roi, res);
                do { t_mid += dt_min; } while (t_mid < t_target);
                dt = calc_dt(t_mid, cone_angle, dt_min, dt_max);
                t0 = t_mid - dt * 0.5f;
                t1 = t_mid + dt * 0.5f;
            } else {
                t0 = t1;
                t1 = t0 + calc_dt(t0, cone_angle, dt_min, dt_max);
                t_mid = (t0 + t1) * 0.5f;
            }
        }
        if (!t_starts) counts[i] = j;
    }
}

/* nerfacc render_visibility: T = exclusive cumprod(1 - alpha) per ray,
 * keep = T >= early_stop_eps (& alpha >= alpha_thre when alpha_thre > 0).
 * alpha = 1 - exp(-sigma * (t_end - t_start)) is formed here in float32. */
void orc_visibility(int64_t n_rays, const int64_t *offsets, const int32_t *counts,
                    const float *sigmas, const float *t_starts, const float *t_ends,
                    float early_stop_eps, float alpha_thre, uint8_t *keep) {
    for (int64_t i = 0; i < n_rays; ++i) {
        float T = 1.f;
        for (int64_t j = offsets[i]; j < offsets[i] + counts[i]; ++j) {
            float alpha = 1.f - expf(-sigmas[j] * (t_ends[j] - t_starts[j]));
            int k = T >= early_stop_eps;
            if (alpha_thre > 0.f) k = k && (alpha >= alpha_thre);
            keep[j] = (uint8_t)k;
            T = T * (1.f - alpha);
        }
    }
}
