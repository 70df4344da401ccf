// Ray / AABB intersection, occupancy-grid ray marching (two passes), visibility filtering and
// packed-stream bookkeeping.  Replaces nerfacc==0.3.1 `ray_marching` (+ `ray_aabb_intersect`,
// `render_visibility`, pack/unpack_info) as called at robust_e_nerf/external/utils.py:106-119.
//
// Compiled with -ffp-contract=off: interval endpoints and sample counts are a pure function of
// un-fused float32 arithmetic so they match the sequential oracle (oracle/csrc/march.c) bit for
// bit.  One thread per ray: rays are independent, the 2 MiB (128^3) / 16 MiB (256^3) occupancy
// grid is L2 / Infinity-Cache resident, and the march is latency- not bandwidth-bound.
#include "ren_common.h"
#include <cstdlib>

namespace {

struct MarchArgs {
    float roi[6];
    int res[3];
    int type;
    float step_size, cone_angle;
    int mode, n_uniform;
    float rinv[3];                                     // 1 / res[k] where res[k] is a power of two (exact), else 0
    float rext[3];                                     // RN(1 / (roi[3 + k] - roi[k])): the verified division (fastdiv)
    int fastdiv;                                       // REN_MARCH_VERIFIED_DIV: the caller ran ren_march_div_check on this roi
};

__device__ __forceinline__ float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

__global__ void ray_aabb_kernel(const float *__restrict__ o, const float *__restrict__ d, int64_t n,
                                float a0, float a1, float a2, float a3, float a4, float a5,
                                float near_plane, float far_plane,
                                float *__restrict__ t_min, float *__restrict__ t_max) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *ro = o + 3 * i, *rd = d + 3 * i;
    float tmin = (a0 - ro[0]) / rd[0], tmax = (a3 - ro[0]) / rd[0];
    if (tmin > tmax) { float t = tmin; tmin = tmax; tmax = t; }
    float tymin = (a1 - ro[1]) / rd[1], tymax = (a4 - ro[1]) / rd[1];
    if (tymin > tymax) { float t = tymin; tymin = tymax; tymax = t; }
    bool miss = (tmin > tymax) || (tymin > tmax);
    if (!miss) {
        if (tymin > tmin) tmin = tymin;
        if (tymax < tmax) tmax = tymax;
        float tzmin = (a2 - ro[2]) / rd[2], tzmax = (a5 - ro[2]) / rd[2];
        if (tzmin > tzmax) { float t = tzmin; tzmin = tzmax; tzmax = t; }
        miss = (tmin > tzmax) || (tzmin > tmax);
        if (!miss) {
            if (tzmin > tmin) tmin = tzmin;
            if (tzmax < tmax) tmax = tzmax;
        }
    }
    float lo, hi;
    if (miss) { lo = 1e10f; hi = 1e10f; }
    else { lo = tmin > 0.f ? tmin : 0.f; hi = tmax; }
    if (near_plane == near_plane) lo = lo < near_plane ? near_plane : lo;   // torch.clamp(min=near)
    if (far_plane == far_plane) hi = hi > far_plane ? far_plane : hi;       // torch.clamp(max=far)
    t_min[i] = lo;
    t_max[i] = hi;
}

// a / b for a FIXED divisor b with y = RN(1 / b): q0 = a y, r = a - q0 b (exact, one fma), q = q0 + r y (Markstein).  For a
// given b this is bit-identical to the IEEE division for every a it has been CHECKED on: ren_march_div_check() runs all ~3.4e9
// values of 2^-100 < |a| < 2^100 against the three extents of a scene box once (the caller then sets REN_MARCH_VERIFIED_DIV).
// The numerators here are p - roi_lo with a box corner of normal magnitude (2^-60 <= |roi_lo|, also checked there): such a
// difference is 0 (0 both ways) or at least half an ulp of the corner, i.e. > 2^-100; above 2^100 lie only positions no finite
// ray of a scene reaches (a run-time guard for them cost the spec kernels the whole gain).  Three dependent operations instead
// of the ~11 of the division sequence, once per visited cell: -8 .. -13 % on the count passes (bound by their VALU chain).
__device__ __forceinline__ float div_verified(float a, float b, float y) {
    const float q0 = a * y;
    return __fmaf_rn(__fmaf_rn(-q0, b, a), y, q0);
}
__device__ __forceinline__ void roi_to_unit(const float *p, const MarchArgs &a, float *u) {
    float num[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) num[k] = p[k] - a.roi[k];
    if (a.fastdiv) {                                             // (uniform)
#pragma unroll
        for (int k = 0; k < 3; ++k) u[k] = div_verified(num[k], a.roi[3 + k] - a.roi[k], a.rext[k]);
        return;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) u[k] = num[k] / (a.roi[3 + k] - a.roi[k]);
}

// `u0` = roi_to_unit(p): computed ONCE per visited position by the caller and shared with distance_to_next_voxel (with the
// branch inside roi_to_unit the compiler no longer merges the two computations by itself)
__device__ __forceinline__ bool grid_occupied_at(const float *p, const float *u0, const MarchArgs &a,
                                                 const uint8_t *__restrict__ binary) {
    if (a.type == REN_CT_AABB) {
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (p[k] < a.roi[k] || p[k] > a.roi[3 + k]) return false;
    }
    float u[3] = {u0[0], u0[1], u0[2]};
    if (a.type == REN_CT_SPHERE) {
#pragma unroll
        for (int k = 0; k < 3; ++k) u[k] = u[k] * 2.f - 1.f;
        float norm = sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
        if (norm > 1.f) {
            float s = (2.f - 1.f / norm);
#pragma unroll
            for (int k = 0; k < 3; ++k) u[k] = s * (u[k] / norm);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) u[k] = u[k] * 0.25f + 0.5f;
    } else if (a.type == REN_CT_TANH) {
#pragma unroll
        for (int k = 0; k < 3; ++k) u[k] = tanhf(u[k] - 0.5f) * 0.5f + 0.5f;
    }
    int idx = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        int c = (int)(u[k] * (float)a.res[k]);
        c = c < 0 ? 0 : (c > a.res[k] - 1 ? a.res[k] - 1 : c);
        idx = idx * a.res[k] + c;
    }
    return binary[idx] != 0;
}

__device__ __forceinline__ float sgnf(float v) { return (float)((v > 0.f) - (v < 0.f)); }

__device__ __forceinline__ float distance_to_next_voxel(const float *u, const float *dir,
                                                        const float *inv_dir, const MarchArgs &a) {
    float t = 1e30f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float r = (float)a.res[k];
        float x = u[k] * r;
        // (a division by a power of two IS the multiplication by its reciprocal, bit for bit: 128^3 / 256^3 grids skip the
        // IEEE division sequence here, a quarter of an empty-cell step)
        const float q = (floorf(x + 0.5f + 0.5f * sgnf(dir[k])) - x) * inv_dir[k];
        float tx = (a.rinv[k] != 0.f ? q * a.rinv[k] : q / r) * (a.roi[3 + k] - a.roi[k]);
        if (tx < t) t = tx;
    }
    return t > 0.f ? t : 0.f;
}

__device__ __forceinline__ float calc_dt(float t, float cone_angle, float dt_min, float dt_max) {
    return clampf(t * cone_angle, dt_min, dt_max);
}

template <bool WRITE>
__global__ void ray_march_kernel(const float *__restrict__ o, const float *__restrict__ d,
                                 const float *__restrict__ t_min, const float *__restrict__ t_max,
                                 const float *__restrict__ jitter, int64_t n_rays, MarchArgs a,
                                 const uint8_t *__restrict__ binary,
                                 const int64_t *__restrict__ offsets, int32_t *__restrict__ counts,
                                 int32_t *__restrict__ ray_indices, float *__restrict__ t_starts,
                                 float *__restrict__ t_ends, float2 *__restrict__ cache, int cache_cap) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rays) return;
    // write pass with an interval cache: rays whose intervals the count pass kept are copied by
    // cached_write_kernel; only the ones that overflowed the cache are marched again
    if (WRITE && cache && counts[i] <= cache_cap) return;
    // a ray without samples writes nothing: also what keeps a render that ren_scan_guard / ren_count_guard cleared (counts all
    // zero, offsets meaningless) from storing past its capacity-sized arrays when there is no interval cache
    if (WRITE && counts && counts[i] == 0) return;
    const float ro[3] = {o[3 * i], o[3 * i + 1], o[3 * i + 2]};
    const float rd[3] = {d[3 * i], d[3 * i + 1], d[3 * i + 2]};
    float near = t_min[i];
    const float far = t_max[i];
    const int64_t base = WRITE ? offsets[i] : 0;
    int j = 0;
    if (a.mode == 1) {
        if (near < far) {
            float delta = (far - near) / (float)a.n_uniform;
            float first = jitter ? near + jitter[i] * delta : near;
            if (WRITE) {
                for (j = 0; j < a.n_uniform; ++j) {
                    float t0 = first + (float)j * delta;
                    t_starts[base + j] = t0;
                    t_ends[base + j] = t0 + delta;
                    ray_indices[base + j] = (int32_t)i;
                }
            }
            j = a.n_uniform;
        }
        if (!WRITE) counts[i] = j;
        return;
    }
    if (jitter) near = near + jitter[i] * a.step_size;      // stratified: one uniform per ray
    const float inv_dir[3] = {1.f / rd[0], 1.f / rd[1], 1.f / rd[2]};
    const float dt_min = a.step_size, dt_max = 1e10f;
    float t0 = near;
    float dt = calc_dt(t0, a.cone_angle, dt_min, dt_max);
    float t1 = t0 + dt;
    float t_mid = (t0 + t1) * 0.5f;
    while (t_mid < far) {
        float p[3] = {ro[0] + t_mid * rd[0], ro[1] + t_mid * rd[1], ro[2] + t_mid * rd[2]}, u[3];
        roi_to_unit(p, a, u);
        if (grid_occupied_at(p, u, a, binary)) {
            if (WRITE) {
                t_starts[base + j] = t0;
                t_ends[base + j] = t1;
                ray_indices[base + j] = (int32_t)i;
            } else if (cache && j < cache_cap) {
                cache[i * cache_cap + j] = make_float2(t0, t1);
            }
            ++j;
            t0 = t1;
            t1 = t0 + calc_dt(t0, a.cone_angle, dt_min, dt_max);
            t_mid = (t0 + t1) * 0.5f;
        } else if (a.type == REN_CT_AABB) {
            float t_target = t_mid + distance_to_next_voxel(u, rd, inv_dir, a);
            do { t_mid += dt_min; } while (t_mid < t_target);
            dt = calc_dt(t_mid, a.cone_angle, dt_min, dt_max);
            t0 = t_mid - dt * 0.5f;
            t1 = t_mid + dt * 0.5f;
        } else {
            t0 = t1;
            t1 = t0 + calc_dt(t0, a.cone_angle, dt_min, dt_max);
            t_mid = (t0 + t1) * 0.5f;
        }
    }
    if (!WRITE) counts[i] = j;
}

// Speculative marching, SPEC lanes per ray.  The march is a dependent chain (position -> occupancy byte -> next
// position) of ~1 000 steps per ray, i.e. ~0.45 ms of pure load latency however few rays there are.  Lane l of a
// ray's group assumes the next l cells are all occupied, forms ITS state with the same float operations in the
// same order the sequential loop would use (l "advance one interval" transitions from the group's state), and all
// lanes test their cell at once.  The leading run of occupied cells is emitted, the first empty (or out-of-range)
// lane decides how the group continues, exactly as the sequential loop does at that step.  In contracted space an
// empty cell advances the state like an occupied one, so all SPEC lanes always count.  Output is bit-identical to
// ray_march_kernel (same tests); dense rays need 1/SPEC of the load round trips (SPEC 8 -> 16: 96 -> 77 us for 16 k rays at configs[4] settings).
// SPEC lanes per ray: every iteration costs SPEC - 1 redundant state transitions of VALU work per lane and saves up to SPEC - 1
// load round trips, so the best width falls as the ray count (= the waves per SIMD that hide the round trips anyway) grows.
// Count passes of a training step, ms (round 5, 128^3 grid, bench.py --sampler occgrid; sequential / 2 / 4 / 8 / 16 lanes):
//   12 k + 6 k rays   0.72 / 0.67 / 0.52 / 0.44 / 0.44        33 k + 16 k rays  0.80 / 0.71 / 0.58 / 0.60 / 0.90
//   66 k rays         0.43 / 0.36 / 0.34 / 0.45 / 0.61        131 k rays        0.42 / 0.45 / 0.55 / 0.72 / 1.05
template <bool WRITE, int SPEC>
__global__ __launch_bounds__(256) void ray_march_spec_kernel(
    const float *__restrict__ o, const float *__restrict__ d, const float *__restrict__ t_min,
    const float *__restrict__ t_max, const float *__restrict__ jitter, int64_t n_rays, MarchArgs a,
    const uint8_t *__restrict__ binary, const int64_t *__restrict__ offsets, int32_t *__restrict__ counts,
    int32_t *__restrict__ ray_indices, float *__restrict__ t_starts, float *__restrict__ t_ends,
    float2 *__restrict__ cache, int cache_cap) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / SPEC;
    const int l = threadIdx.x % SPEC, lane = threadIdx.x & 63, g0 = lane - l;   // g0: first lane of the group
    bool done = i >= n_rays;
    if (!done && WRITE && cache && counts[i] <= cache_cap) done = true;          // copied by cached_write_kernel
    if (!done && WRITE && counts && counts[i] == 0) done = true;                 // no samples (or a cleared render): nothing to write
    float ro[3] = {0.f, 0.f, 0.f}, rd[3] = {1.f, 1.f, 1.f};
    float near = 0.f, far = 0.f;
    int64_t base = 0;
    if (!done) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { ro[k] = o[3 * i + k]; rd[k] = d[3 * i + k]; }
        near = t_min[i]; far = t_max[i];
        if (WRITE) base = offsets[i];
        if (jitter) near = near + jitter[i] * a.step_size;
    }
    const float inv_dir[3] = {1.f / rd[0], 1.f / rd[1], 1.f / rd[2]};
    const float dt_min = a.step_size, dt_max = 1e10f;
    float t0 = near;
    float t1 = t0 + calc_dt(t0, a.cone_angle, dt_min, dt_max);
    float t_mid = (t0 + t1) * 0.5f;
    int j = 0;
    const bool is_aabb = a.type == REN_CT_AABB;
    if (!(t_mid < far)) done = true;
    // the sequential loop's step through an EMPTY cell (aabb only): skip to the next voxel in dt_min steps
    auto empty_step = [&](float tm, float &n0, float &n1, float &nm) {
        const float p[3] = {ro[0] + tm * rd[0], ro[1] + tm * rd[1], ro[2] + tm * rd[2]};
        float u[3];
        roi_to_unit(p, a, u);
        const float t_target = tm + distance_to_next_voxel(u, rd, inv_dir, a);
        do { tm += dt_min; } while (tm < t_target);
        const float dt = calc_dt(tm, a.cone_angle, dt_min, dt_max);
        n0 = tm - dt * 0.5f;
        n1 = tm + dt * 0.5f;
        nm = tm;
    };
    // aabb rays cross long runs of empty cells too (one load round trip per voxel when only occupied runs are speculated:
    // the count pass was 0.5 ms for 16 k rays of the training configuration).  After an empty cell the group speculates
    // an EMPTY run instead: lane l forms the state l empty steps further on -- the same float operations, in the same
    // order, that the sequential loop applies -- all lanes test their cell at once, the leading run of empty cells is
    // skipped in one go, and the first occupied (or out-of-range) lane's state is where the loop would be.
    bool emode = false;                                          // per group (every lane of a group holds the same value)
    while (__any(!done)) {
        if (emode) {
            float e0 = t0, e1 = t1, em = t_mid;
#pragma unroll 1
            for (int k = 0; k < SPEC - 1; ++k)
                if (k < l && !done && em < far) empty_step(em, e0, e1, em);
            const bool in = !done && em < far;
            bool occ = false;
            if (in) {
                const float p[3] = {ro[0] + em * rd[0], ro[1] + em * rd[1], ro[2] + em * rd[2]};
                float u[3];
                roi_to_unit(p, a, u);
                occ = grid_occupied_at(p, u, a, binary);
            }
            const unsigned in_bits = (unsigned)(__ballot(in) >> g0) & ((1u << SPEC) - 1u);
            const unsigned occ_bits = (unsigned)(__ballot(occ) >> g0) & ((1u << SPEC) - 1u);
            const int e = __builtin_ctz((occ_bits | ~in_bits) | (1u << SPEC));   // leading in-range empty lanes = steps taken
            const int src = g0 + (e < SPEC ? e : SPEC - 1);
            const float b0 = __shfl(e0, src, 64), b1 = __shfl(e1, src, 64), bm = __shfl(em, src, 64);
            if (!done) {
                if (e == SPEC) {                                 // all empty: one more step from the last lane's state
                    empty_step(bm, t0, t1, t_mid);
                } else {                                         // lane e: occupied -> carry on from there; or past the far end
                    t0 = b0; t1 = b1; t_mid = bm;
                    emode = false;
                }
                if (!(t_mid < far)) done = true;
            }
            continue;
        }
        float a0 = t0, a1 = t1, am = t_mid;                      // this lane's state: l intervals further on
#pragma unroll
        for (int k = 0; k < SPEC - 1; ++k)
            if (k < l) { a0 = a1; a1 = a0 + calc_dt(a0, a.cone_angle, dt_min, dt_max); am = (a0 + a1) * 0.5f; }
        const bool in = !done && am < far;
        bool occ = false;
        if (in) {
            const float p[3] = {ro[0] + am * rd[0], ro[1] + am * rd[1], ro[2] + am * rd[2]};
            float u[3];
            roi_to_unit(p, a, u);
            occ = grid_occupied_at(p, u, a, binary);
        }
        const unsigned in_bits = (unsigned)(__ballot(in) >> g0) & ((1u << SPEC) - 1u);
        const unsigned occ_bits = (unsigned)(__ballot(occ) >> g0) & ((1u << SPEC) - 1u);
        // lanes [0, f) are steps the sequential loop takes with the state this lane assumed
        const int n_in = __builtin_ctz(~in_bits | (1u << SPEC));                 // leading lanes still inside [near, far)
        const int f = is_aabb ? __builtin_ctz(~occ_bits | (1u << SPEC)) : n_in;  // aabb: stop at the first empty cell
        const bool emit = !done && l < f && occ;
        if (emit) {
            const int at = j + __popc(occ_bits & ((1u << l) - 1u));
            if (WRITE) {
                t_starts[base + at] = a0;
                t_ends[base + at] = a1;
                ray_indices[base + at] = (int32_t)i;
            } else if (cache && at < cache_cap) {
                cache[i * cache_cap + at] = make_float2(a0, a1);
            }
        }
        const int src = g0 + (f < SPEC ? f : SPEC - 1);
        const float b0 = __shfl(a0, src, 64), b1 = __shfl(a1, src, 64), bm = __shfl(am, src, 64);
        if (!done) {
            j += __popc(occ_bits & ((1u << f) - 1u));
            if (f == SPEC) {                                     // whole group consumed: one more transition
                t0 = b1;
                t1 = t0 + calc_dt(t0, a.cone_angle, dt_min, dt_max);
                t_mid = (t0 + t1) * 0.5f;
            } else if (f >= n_in) {                              // lane f is past the far end: the loop ends
                done = true;
            } else {                                             // aabb only: lane f sits in an empty cell
                empty_step(bm, t0, t1, t_mid);
                emode = true;                                    // more empty cells are likely to follow
            }
            if (!(t_mid < far)) done = true;
        }
    }
    if (!WRITE && l == 0 && i < n_rays) counts[i] = j;
}

// Write pass from the interval cache of the count pass: 8 lanes per ray copy its (t0, t1) pairs into the packed
// streams.  The march itself is a ~1 000-step dependent chain per ray (0.45 ms however few rays there are); with
// the cache it runs once instead of twice per render, with bit-identical output.
__global__ __launch_bounds__(256) void cached_write_kernel(const float2 *__restrict__ cache, int cache_cap, int64_t n_rays,
                                                           const int64_t *__restrict__ offsets,
                                                           const int32_t *__restrict__ counts,
                                                           int32_t *__restrict__ ray_indices, float *__restrict__ t_starts,
                                                           float *__restrict__ t_ends) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = g >> 3;
    if (i >= n_rays) return;
    const int c = counts[i];
    if (c > cache_cap) return;
    const int64_t base = offsets[i];
    const float2 *src = cache + i * cache_cap;
    for (int j = (int)(g & 7); j < c; j += 8) {
        const float2 v = src[j];
        t_starts[base + j] = v.x;
        t_ends[base + j] = v.y;
        ray_indices[base + j] = (int32_t)i;
    }
}

// Fixed-S stratified sampler, write pass: one thread per (ray, sample) so the three packed streams are written
// with coalesced stores (one thread per ray wrote 128 samples 512 B apart: 0.54 ms for 200 MB; this: ~0.06 ms).
// Same expressions as the per-ray loop above.
__global__ __launch_bounds__(256) void uniform_write_kernel(const float *__restrict__ t_min, const float *__restrict__ t_max,
                                                            const float *__restrict__ jitter, int64_t n_rays, int n_uniform,
                                                            const int64_t *__restrict__ offsets,
                                                            int32_t *__restrict__ ray_indices, float *__restrict__ t_starts,
                                                            float *__restrict__ t_ends) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = g / n_uniform;
    const int j = (int)(g - i * n_uniform);
    if (i >= n_rays) return;
    const float near = t_min[i], far = t_max[i];
    if (!(near < far)) return;
    const float delta = (far - near) / (float)n_uniform;
    const float first = jitter ? near + jitter[i] * delta : near;
    const float t0 = first + (float)j * delta;
    const int64_t at = offsets[i] + j;
    t_starts[at] = t0;
    t_ends[at] = t0 + delta;
    ray_indices[at] = (int32_t)i;
}

// Exclusive scan of the per-ray sample counts, up to SCAN_TILE * 1024 elements, in three coalesced stages:
// scan inside 1024-element tiles; one workgroup scans the tile sums (tile sum = local offset + count of the
// tile's last element) into the caller's int64[1024] scratch; add the tile bases.  (One workgroup walking
// 131 072 counts serially took 0.22 ms; this takes ~15 us.)
constexpr int SCAN_TILE = 1024;

__global__ __launch_bounds__(1024) void scan_tiles_kernel(const int32_t *__restrict__ counts, int64_t n,
                                                          int64_t *__restrict__ offsets) {
    __shared__ int64_t wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t i = (int64_t)blockIdx.x * SCAN_TILE + tid;
    const int64_t c = i < n ? counts[i] : 0;
    int64_t v = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int64_t t = __shfl_up(v, off, 64);
        if (lane >= off) v += t;
    }
    if (lane == 63) wsum[wave] = v;
    __syncthreads();
    int64_t base = 0;
    for (int w = 0; w < wave; ++w) base += wsum[w];
    if (i < n) offsets[i] = base + v - c;
}

// one workgroup: exclusive scan of the tile sums (tile sum = local offset + count of the tile's last element)
__global__ __launch_bounds__(1024) void scan_sums_kernel(const int32_t *__restrict__ counts, int64_t n, int n_tiles,
                                                         const int64_t *__restrict__ offsets,
                                                         int64_t *__restrict__ tile_base, int64_t *__restrict__ total) {
    __shared__ int64_t part[1024];
    const int tid = threadIdx.x;
    int64_t s = 0;
    if (tid < n_tiles) {
        const int64_t last = ((int64_t)tid + 1) * SCAN_TILE - 1 < n ? ((int64_t)tid + 1) * SCAN_TILE - 1 : n - 1;
        s = offsets[last] + counts[last];
    }
    part[tid] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int64_t v = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    if (tid < n_tiles) tile_base[tid] = part[tid] - s;
    if (tid == 1023 && total) total[0] = part[1023];
}

__global__ __launch_bounds__(1024) void scan_add_kernel(int64_t n, const int64_t *__restrict__ tile_base,
                                                        int64_t *__restrict__ offsets) {
    const int64_t i = (int64_t)blockIdx.x * SCAN_TILE + threadIdx.x;
    if (i < n) offsets[i] += tile_base[blockIdx.x];
}

// Single-workgroup exclusive scan: fallback above SCAN_TILE * 1024 elements.
__global__ __launch_bounds__(1024) void exclusive_scan_kernel(const int32_t *__restrict__ counts, int64_t n,
                                                              int64_t *__restrict__ offsets,
                                                              int64_t *__restrict__ total) {
    __shared__ int64_t part[1024];
    const int tid = threadIdx.x;
    const int64_t per = (n + 1023) / 1024;
    const int64_t b = tid * per, e = (b + per < n) ? b + per : n;
    int64_t s = 0;
    for (int64_t i = b; i < e; ++i) s += counts[i];
    part[tid] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        int64_t v = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int64_t run = part[tid] - s;
    for (int64_t i = b; i < e; ++i) { offsets[i] = run; run += counts[i]; }
    if (tid == 1023 && total) total[0] = part[1023];
}

// One wavefront per ray (a thread per ray walked ~200 samples as a dependent chain of loads and expf: 0.15 ms
// for any ray count).  The 64 alphas of a chunk are formed in parallel; the transmittance is still multiplied up
// in sample order (readlane loop), so every T -- and with it every keep decision -- has the sequential rounding.
__global__ __launch_bounds__(256) void visibility_kernel(const int64_t *__restrict__ offsets, const int32_t *__restrict__ counts,
                                  int64_t n_rays, const float *__restrict__ sigmas,
                                  const float *__restrict__ t_starts, const float *__restrict__ t_ends,
                                  float eps, float alpha_thre, uint8_t *__restrict__ keep,
                                  int32_t *__restrict__ kept_counts) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (i >= n_rays) return;
    const int64_t b = offsets[i];
    const int cnt = counts[i];
    float T = 1.f;                                                 // wave-uniform
    int kept = 0;
    for (int c0 = 0; c0 < cnt; c0 += 64) {
        const bool valid = c0 + lane < cnt;
        const int64_t j = b + c0 + lane;
        float alpha = 0.f;
        if (valid) alpha = 1.f - expf(-sigmas[j] * (t_ends[j] - t_starts[j]));
        const float om = 1.f - alpha;                              // exactly 1 on the padding lanes
        float t_here = 0.f;
#pragma unroll
        for (int l = 0; l < 64; ++l) {
            if (lane == l) t_here = T;
            T = T * __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, om), l));
        }
        bool k = valid && t_here >= eps;
        if (alpha_thre > 0.f) k = k && (alpha >= alpha_thre);
        if (valid) keep[j] = (uint8_t)k;
        kept += __popcll(__ballot(k));
    }
    if (lane == 0) kept_counts[i] = kept;
}

__global__ __launch_bounds__(256) void compact_kernel(const int64_t *__restrict__ offsets, const int32_t *__restrict__ counts,
                               const int64_t *__restrict__ new_offsets, int64_t n_rays,
                               const uint8_t *__restrict__ keep, const float *__restrict__ t_starts,
                               const float *__restrict__ t_ends, int32_t *__restrict__ out_ri,
                               float *__restrict__ out_ts, float *__restrict__ out_te) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (i >= n_rays) return;
    const int64_t b = offsets[i];
    const int cnt = counts[i];
    int64_t w = new_offsets[i];
    for (int c0 = 0; c0 < cnt; c0 += 64) {
        const bool valid = c0 + lane < cnt;
        const int64_t j = b + c0 + lane;
        const bool k = valid && keep[j];
        const unsigned long long m = __ballot(k);
        if (k) {
            const int64_t at = w + __popcll(m & ((1ull << lane) - 1ull));
            out_ri[at] = (int32_t)i;
            out_ts[at] = t_starts[j];
            out_te[at] = t_ends[j];
        }
        w += __popcll(m);
    }
}

// Per-sample feature vectors (fragment layout, 32 floats per sample) of the samples that survive the visibility
// test, moved to their compacted positions: the density pre-pass already encoded every marched sample, so the
// differentiable pass need not gather the hash table again for the survivors.
__global__ __launch_bounds__(256) void compact_features_kernel(const int64_t *__restrict__ offsets, const int32_t *__restrict__ counts,
                                        const int64_t *__restrict__ new_offsets, int64_t n_rays,
                                        const uint8_t *__restrict__ keep, const float *__restrict__ feat_in,
                                        float *__restrict__ feat_out) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (i >= n_rays) return;
    const int64_t b = offsets[i];
    const int cnt = counts[i];
    int64_t w = new_offsets[i];
    for (int c0 = 0; c0 < cnt; c0 += 64) {
        const bool valid = c0 + lane < cnt;
        const int64_t j = b + c0 + lane;
        const bool k = valid && keep[j];
        const unsigned long long m = __ballot(k);
        if (k) {
            const int64_t at = w + __popcll(m & ((1ull << lane) - 1ull));
            const float *src = feat_in + (j >> 5) * 1024 + (j & 31);
            float *dst = feat_out + (at >> 5) * 1024 + (at & 31);
            float v[32];
#pragma unroll
            for (int q = 0; q < 32; ++q) v[q] = src[q * 32];
#pragma unroll
            for (int q = 0; q < 32; ++q) dst[q * 32] = v[q];
        }
        w += __popcll(m);
    }
}

__global__ void zero_i32_kernel(int32_t *p, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0;
}

__global__ void pack_info_kernel(const int32_t *__restrict__ ri, int64_t n, int64_t *__restrict__ offsets,
                                 int32_t *__restrict__ counts) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t r = ri[i];
    if (i == 0 || ri[i - 1] != r) {
        // first sample of ray r: find run length by scanning forward is O(run); instead let the
        // last sample of the run write the count.
        offsets[r] = i;
    }
    if (i == n - 1 || ri[i + 1] != r) {
        // run end: count = end - start; start is found by walking back (runs are <= 1024 long)
        int64_t s = i;
        while (s > 0 && ri[s - 1] == r) --s;
        counts[r] = (int32_t)(i + 1 - s);
    }
}

__global__ void fill_empty_offsets_kernel(int64_t *offsets, const int32_t *counts, int64_t n_rays) {
    // rays without samples get offset 0 (never dereferenced since count == 0)
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_rays && counts[i] == 0) offsets[i] = 0;
}

}  // namespace

extern "C" int ren_ray_aabb_intersect(const float *rays_o, const float *rays_d, int64_t n_rays,
                                      const float *aabb, float near_plane, float far_plane,
                                      float *t_min, float *t_max, void *stream) {
    if (!rays_o || !rays_d || !aabb || !t_min || !t_max || n_rays < 0) return REN_ERR_BAD_ARG;
    if (n_rays == 0) return REN_OK;
    hipLaunchKernelGGL(ray_aabb_kernel, dim3(ren_blocks(n_rays, 256)), dim3(256), 0, (hipStream_t)stream,
                       rays_o, rays_d, n_rays, aabb[0], aabb[1], aabb[2], aabb[3], aabb[4], aabb[5],
                       near_plane, far_plane, t_min, t_max);
    REN_CHECK_LAUNCH();
}

// Exhaustive check of div_verified() against the division for the three extents of a box: every float a with
// 2^-100 < |a| < 2^100 (exponent fields 27 .. 226, both signs: 3.4e9 values per extent, ~20 ms in all).  mismatches[0] = how
// many (a, extent) pairs differ in any bit; 0 = the caller may set REN_MARCH_VERIFIED_DIV for this roi.
__global__ __launch_bounds__(256) void march_div_check_kernel(float b0, float b1, float b2, float y0, float y1, float y2,
                                                              unsigned long long *__restrict__ mismatches) {
    const float b[3] = {b0, b1, b2}, y[3] = {y0, y1, y2};
    unsigned bad = 0;
    // mantissa + low exponent bits from the thread, the rest of the exponent range from the loop
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;       // 2^24 threads: 23 mantissa bits + sign
    const uint32_t low = (t & 0x7fffffu) | ((t >> 23) << 31);
    for (uint32_t e = 27; e <= 226; ++e) {
        const float a = __uint_as_float(low | (e << 23));
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float q = a / b[k], f = div_verified(a, b[k], y[k]);
            bad += __float_as_uint(q) != __float_as_uint(f);
        }
    }
    if (bad) atomicAdd(mismatches, (unsigned long long)bad);
}

extern "C" int ren_march_div_check(const float *roi, int64_t *mismatches, void *stream) {
    if (!roi || !mismatches) return REN_ERR_BAD_ARG;
    float b[3], y[3];
    for (int k = 0; k < 3; ++k) {
        b[k] = roi[3 + k] - roi[k];
        if (!(b[k] > 0x1p-60f) || !(b[k] < 0x1p60f)) return REN_ERR_BAD_ARG;
        if (!(fabsf(roi[k]) >= 0x1p-60f) || !(fabsf(roi[k]) < 0x1p60f)) return REN_ERR_UNSUPPORTED;    // (see div_verified)
        y[k] = 1.f / b[k];
    }
    (void)hipMemsetAsync(mismatches, 0, sizeof(int64_t), (hipStream_t)stream);
    hipLaunchKernelGGL(march_div_check_kernel, dim3(1 << 16), dim3(256), 0, (hipStream_t)stream, b[0], b[1], b[2], y[0], y[1], y[2],
                       reinterpret_cast<unsigned long long *>(mismatches));
    REN_CHECK_LAUNCH();
}

extern "C" int ren_ray_march(const float *rays_o, const float *rays_d, const float *t_min,
                             const float *t_max, const float *jitter, int64_t n_rays,
                             const float *roi, const int32_t *res, const uint8_t *binary,
                             int32_t contraction_type, float step_size, float cone_angle,
                             int32_t mode, int32_t n_uniform, const int64_t *offsets, int32_t *counts,
                             int32_t *ray_indices, float *t_starts, float *t_ends, float *interval_cache,
                             int32_t cache_cap, void *stream) {
    if (!rays_o || !rays_d || !t_min || !t_max || n_rays < 0) return REN_ERR_BAD_ARG;
    const bool verified_div = (mode & REN_MARCH_VERIFIED_DIV) != 0;
    mode &= ~REN_MARCH_VERIFIED_DIV;
    if (mode != 0 && mode != 1) return REN_ERR_BAD_ARG;
    if (mode == 0 && (!roi || !res || !binary || step_size <= 0.f)) return REN_ERR_BAD_ARG;
    if (mode == 1 && n_uniform <= 0) return REN_ERR_BAD_ARG;
    if (contraction_type < 0 || contraction_type > 2) return REN_ERR_BAD_ARG;
    const bool write = t_starts != nullptr;
    if (write && (!offsets || !t_ends || !ray_indices)) return REN_ERR_BAD_ARG;
    if (!write && !counts) return REN_ERR_BAD_ARG;
    if (interval_cache && (cache_cap <= 0 || !counts)) return REN_ERR_BAD_ARG;
    if (n_rays == 0) return REN_OK;
    float2 *cache = mode == 0 ? reinterpret_cast<float2 *>(interval_cache) : nullptr;
    MarchArgs a;
    for (int k = 0; k < 6; ++k) a.roi[k] = roi ? roi[k] : (k < 3 ? -1e10f : 1e10f);
    for (int k = 0; k < 3; ++k) a.res[k] = res ? res[k] : 1;
    for (int k = 0; k < 3; ++k) a.rinv[k] = (a.res[k] > 0 && (a.res[k] & (a.res[k] - 1)) == 0) ? 1.f / (float)a.res[k] : 0.f;
    a.type = contraction_type; a.step_size = step_size; a.cone_angle = cone_angle;
    a.mode = mode; a.n_uniform = n_uniform;
    for (int k = 0; k < 3; ++k) a.rext[k] = 1.f / (a.roi[3 + k] - a.roi[k]);
    a.fastdiv = verified_div && roi ? 1 : 0;
    dim3 grid(ren_blocks(n_rays, 64)), block(64);   // short blocks: ray lengths vary a lot
    // tuning / verification knob: 1 = sequential kernel only; 2, 4, 8, 16 = that many speculative lanes per ray whatever the size
    const int kn = ren_knob(REN_KNOB_MARCH_SEQUENTIAL);
    // speculation spends SPEC lanes per ray to cut load round trips: it wins while the launch is latency-bound, and the
    // width that wins shrinks as the rays alone fill the chip (see the table at the kernel)
    int width = n_rays <= 10240 ? 16 : n_rays <= 24576 ? 8 : n_rays <= 81920 ? 4 : 0;
    if (kn == 1) width = 0;
    else if (kn == 2 || kn == 4 || kn == 8 || kn == 16) width = kn;
    if (mode != 0) width = 0;
    const dim3 sgrid(ren_blocks(n_rays * (width ? width : 1), 256));
#define REN_MARCH_SPEC(WRITE)                                                                                              \
    do {                                                                                                                   \
        if (width == 16)                                                                                                   \
            hipLaunchKernelGGL((ray_march_spec_kernel<WRITE, 16>), sgrid, dim3(256), 0, (hipStream_t)stream, rays_o, rays_d, \
                               t_min, t_max, jitter, n_rays, a, binary, offsets, counts, ray_indices, t_starts, t_ends,    \
                               cache, cache_cap);                                                                          \
        else if (width == 8)                                                                                               \
            hipLaunchKernelGGL((ray_march_spec_kernel<WRITE, 8>), sgrid, dim3(256), 0, (hipStream_t)stream, rays_o, rays_d, \
                               t_min, t_max, jitter, n_rays, a, binary, offsets, counts, ray_indices, t_starts, t_ends,    \
                               cache, cache_cap);                                                                          \
        else if (width == 4)                                                                                               \
            hipLaunchKernelGGL((ray_march_spec_kernel<WRITE, 4>), sgrid, dim3(256), 0, (hipStream_t)stream, rays_o, rays_d, \
                               t_min, t_max, jitter, n_rays, a, binary, offsets, counts, ray_indices, t_starts, t_ends,    \
                               cache, cache_cap);                                                                          \
        else                                                                                                               \
            hipLaunchKernelGGL((ray_march_spec_kernel<WRITE, 2>), sgrid, dim3(256), 0, (hipStream_t)stream, rays_o, rays_d, \
                               t_min, t_max, jitter, n_rays, a, binary, offsets, counts, ray_indices, t_starts, t_ends,    \
                               cache, cache_cap);                                                                          \
    } while (0)
    if (write && mode == 1)
        hipLaunchKernelGGL(uniform_write_kernel, dim3(ren_blocks(n_rays * n_uniform, 256)), dim3(256), 0,
                           (hipStream_t)stream, t_min, t_max, jitter, n_rays, n_uniform, offsets, ray_indices,
                           t_starts, t_ends);
    else if (write) {
        if (cache)
            hipLaunchKernelGGL(cached_write_kernel, dim3(ren_blocks(n_rays * 8, 256)), dim3(256), 0, (hipStream_t)stream,
                               cache, cache_cap, n_rays, offsets, counts, ray_indices, t_starts, t_ends);
        if (width)
            REN_MARCH_SPEC(true);
        else
            hipLaunchKernelGGL(ray_march_kernel<true>, grid, block, 0, (hipStream_t)stream, rays_o, rays_d,
                               t_min, t_max, jitter, n_rays, a, binary, offsets, counts, ray_indices,
                               t_starts, t_ends, cache, cache_cap);
    } else if (width)
        REN_MARCH_SPEC(false);
    else
        hipLaunchKernelGGL(ray_march_kernel<false>, grid, block, 0, (hipStream_t)stream, rays_o, rays_d,
                           t_min, t_max, jitter, n_rays, a, binary, offsets, counts, ray_indices,
                           t_starts, t_ends, cache, cache_cap);
#undef REN_MARCH_SPEC
    REN_CHECK_LAUNCH();
}

// ---- uniform random numbers for the stratified-sampling jitter (models/nerf.py passes torch.rand; any iid U[0,1) stream is
// the same sampler): Philox4x32-10, counter = (index of the group of four, offset), key = seed
__global__ __launch_bounds__(256) void philox_uniform_kernel(uint64_t seed, uint64_t offset, int64_t n, float *__restrict__ out) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (4 * g >= n) return;
    uint32_t c[4] = {(uint32_t)g, (uint32_t)((uint64_t)g >> 32), (uint32_t)offset, (uint32_t)(offset >> 32)};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        c[1] = (uint32_t)p1; c[3] = (uint32_t)p0; c[0] = n0; c[2] = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (4 * g + j < n) out[4 * g + j] = (float)(c[j] >> 8) * 5.9604644775390625e-8f;       // 24 bits: [0, 1)
}

extern "C" int ren_uniform(uint64_t seed, uint64_t offset, int64_t n, float *out, void *stream) {
    if (!out || n < 0) return REN_ERR_BAD_ARG;
    if (n == 0) return REN_OK;
    hipLaunchKernelGGL(philox_uniform_kernel, dim3(ren_blocks((n + 3) / 4, 256)), dim3(256), 0, (hipStream_t)stream, seed, offset, n, out);
    REN_CHECK_LAUNCH();
}

// Up to SCAN_ONE elements (a training batch's rays) in ONE launch of one workgroup: thread t owns a run of `per` consecutive
// counts (per a multiple of 4: 16-byte loads), the 1 024 run sums are scanned with a wave scan + 16 partials, and -- for the
// device-side sample counts -- the guard of ren_count_guard runs in the same launch: the total is known to the whole
// workgroup, so an overflowing count clears the counts right here.  At the reference's 2^20-sample budget a render has
// 10-20 k rays and its two scans were six launches (+ two guards) of a launch-bound step.
constexpr int SCAN_ONE = 65536;
__global__ __launch_bounds__(1024) void scan_guard_kernel(int32_t *__restrict__ counts, int32_t *__restrict__ counts_also, int64_t n,
                                                          int seg, int64_t *__restrict__ offsets, int64_t *__restrict__ total,
                                                          int guard, int64_t capacity, int64_t *__restrict__ n_out,
                                                          int64_t *__restrict__ stats) {
    // wave w owns the `seg` counts from w * seg on (seg: a multiple of 256) and walks them in pieces of 256 = one 16-byte load
    // per lane: fully coalesced loads AND stores, no barrier inside the walks.  (Round 5's first version gave every THREAD a
    // contiguous run: `per` dependent scalar loads and 8-byte stores 8 * per bytes apart -- 19 us at 16 k counts, 92 us at 65 k.)
    __shared__ int64_t wsum[17];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t w0 = (int64_t)wave * seg;
    auto load4 = [&](int64_t at) {
        int4 v = make_int4(0, 0, 0, 0);
        if (at + 3 < n) v = *reinterpret_cast<const int4 *>(counts + at);
        else if (at < n) {
            v.x = counts[at];
            if (at + 1 < n) v.y = counts[at + 1];
            if (at + 2 < n) v.z = counts[at + 2];
        }
        return v;
    };
    int64_t s = 0;
    for (int k = lane * 4; k < seg; k += 256) {
        const int4 v = load4(w0 + k);
        s += (int64_t)v.x + v.y + v.z + v.w;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) wsum[wave] = s;
    __syncthreads();
    if (tid == 0) {
        int64_t acc = 0;
        for (int w = 0; w < 16; ++w) { const int64_t t = wsum[w]; wsum[w] = acc; acc += t; }
        wsum[16] = acc;
    }
    __syncthreads();
    int64_t carry = wsum[wave];
    const int64_t tot = wsum[16];
    for (int k = lane * 4; k < seg; k += 256) {              // (k - 4 lane is wave-uniform: all lanes take the same trips)
        const int64_t at = w0 + k;
        const int4 v = load4(at);                            // (the wave's own counts again: L1 / L2 hits)
        const int64_t t = (int64_t)v.x + v.y + v.z + v.w;
        int64_t inc = t;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int64_t u = __shfl_up(inc, off, 64);
            if (lane >= off) inc += u;
        }
        const int64_t o0 = carry + inc - t, o1 = o0 + v.x, o2 = o1 + v.y, o3 = o2 + v.z;
        if (at + 3 < n) {
            *reinterpret_cast<longlong2 *>(offsets + at) = make_longlong2(o0, o1);
            *reinterpret_cast<longlong2 *>(offsets + at + 2) = make_longlong2(o2, o3);
        } else if (at < n) {
            offsets[at] = o0;
            if (at + 1 < n) offsets[at + 1] = o1;
            if (at + 2 < n) offsets[at + 2] = o2;
        }
        carry += __shfl(inc, 63, 64);
    }
    if (tid == 0 && total) total[0] = tot;
    if (!guard) return;
    const bool over = tot > capacity;
    if (tid == 0) {
        n_out[0] = over ? 0 : tot;
        if (stats) { stats[0] = tot; stats[1] = over ? 1 : 0; }
    }
    if (over) {
        __syncthreads();                                         // every wave is done re-reading its counts (second walk)
        for (int64_t k = tid; k < n; k += 1024) {
            counts[k] = 0;
            if (counts_also) counts_also[k] = 0;
        }
    }
}

static bool scan_one_launch(int32_t *counts, int32_t *counts_also, int64_t n, int64_t *offsets, int64_t *total, int guard,
                            int64_t capacity, int64_t *n_out, int64_t *stats, hipStream_t st) {
    if (n > SCAN_ONE || n < 1 || ((uintptr_t)counts & 15) || ((uintptr_t)offsets & 15)) return false;
    int seg = (int)((n + 15) / 16);
    seg = (seg + 255) & ~255;
    hipLaunchKernelGGL(scan_guard_kernel, dim3(1), dim3(1024), 0, st, counts, counts_also, n, seg, offsets, total, guard, capacity,
                       n_out, stats);
    return true;
}

extern "C" int ren_exclusive_scan(const int32_t *counts, int64_t n, int64_t *offsets, int64_t *total,
                                  int64_t *scratch1024, void *stream) {
    if (!counts || !offsets || n < 0) return REN_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (scan_one_launch(const_cast<int32_t *>(counts), nullptr, n, offsets, total, 0, 0, nullptr, nullptr, st)) { REN_CHECK_LAUNCH(); }
    const int64_t n_tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    if (scratch1024 && n_tiles >= 2 && n_tiles <= 1024) {
        hipLaunchKernelGGL(scan_tiles_kernel, dim3((unsigned)n_tiles), dim3(SCAN_TILE), 0, st, counts, n, offsets);
        hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(1024), 0, st, counts, n, (int)n_tiles, offsets, scratch1024, total);
        hipLaunchKernelGGL(scan_add_kernel, dim3((unsigned)n_tiles), dim3(SCAN_TILE), 0, st, n, scratch1024, offsets);
    } else {
        hipLaunchKernelGGL(exclusive_scan_kernel, dim3(1), dim3(1024), 0, st, counts, n, offsets, total);
    }
    REN_CHECK_LAUNCH();
}

// ---- device-side sample counts (ABI 24): the scan's total stays on the device; this guard compares it with the capacity the
// host allocated the per-sample arrays for.  Fits: n_out = total.  Does not fit: every ray's count is cleared (the per-ray
// kernels then write nothing and the render is empty), n_out = 0 and the overflow word is raised -- the host reads `stats`
// AFTER it has enqueued the step (no queue drain) and repeats an overflowed step with larger arrays.
__global__ __launch_bounds__(256) void count_guard_kernel(int32_t *__restrict__ counts, int32_t *__restrict__ counts_also,
                                                          int64_t n_rays, const int64_t *__restrict__ total, int64_t capacity,
                                                          int64_t *__restrict__ n_out, int64_t *__restrict__ stats) {
    const int64_t t = total[0];
    const bool over = t > capacity;
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (over && r < n_rays) {
        counts[r] = 0;
        if (counts_also) counts_also[r] = 0;
    }
    if (r == 0) {
        n_out[0] = over ? 0 : t;
        if (stats) { stats[0] = t; stats[1] = over ? 1 : 0; }
    }
}

extern "C" int ren_count_guard(int32_t *counts, int32_t *counts_also, int64_t n_rays, const int64_t *total, int64_t capacity,
                               int64_t *n_out, int64_t *stats, void *stream) {
    if (!counts || !total || !n_out || n_rays < 0 || capacity < 0) return REN_ERR_BAD_ARG;
    hipLaunchKernelGGL(count_guard_kernel, dim3(ren_blocks(n_rays > 0 ? n_rays : 1, 256)), dim3(256), 0, (hipStream_t)stream,
                       counts, counts_also, n_rays, total, capacity, n_out, stats);
    REN_CHECK_LAUNCH();
}

// ren_exclusive_scan + ren_count_guard: one launch for up to 65 536 rays, otherwise the two calls
extern "C" int ren_scan_guard(int32_t *counts, int32_t *counts_also, int64_t n_rays, int64_t *offsets, int64_t *total,
                              int64_t capacity, int64_t *n_out, int64_t *stats, int64_t *scratch1024, void *stream) {
    if (!counts || !offsets || !total || !n_out || n_rays < 0 || capacity < 0) return REN_ERR_BAD_ARG;
    if (scan_one_launch(counts, counts_also, n_rays, offsets, total, 1, capacity, n_out, stats, (hipStream_t)stream)) { REN_CHECK_LAUNCH(); }
    const int rc = ren_exclusive_scan(counts, n_rays, offsets, total, scratch1024, stream);
    if (rc != REN_OK) return rc;
    return ren_count_guard(counts, counts_also, n_rays, total, capacity, n_out, stats, stream);
}

// fragment-layout feature block (16 levels x 2 x 32 lanes per 32 samples): zero the lanes of the LAST block that lie beyond
// the device-side count (what ops.compact_features does with a host count): the MLP kernels multiply dead lanes by zero
// gradients, which only works for finite values
__global__ __launch_bounds__(64) void frag_zero_tail_kernel(float *__restrict__ feat, int64_t capacity, const int64_t *__restrict__ n_dev) {
    const int64_t n = ren_eff_n(capacity, n_dev);
    const int lane = threadIdx.x & 31, half = threadIdx.x >> 5;
    if ((n & 31) == 0 || lane < (n & 31)) return;
    float *b = feat + (n >> 5) * (int64_t)(REN_MAX_LEVELS * 64) + lane;
    for (int l = 0; l < REN_MAX_LEVELS; ++l) b[l * 64 + half * 32] = 0.f;
}

extern "C" int ren_frag_zero_tail(float *feat, int64_t capacity, const int64_t *n_dev, void *stream) {
    if (!feat || !n_dev || capacity < 0) return REN_ERR_BAD_ARG;
    if (capacity == 0) return REN_OK;
    hipLaunchKernelGGL(frag_zero_tail_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, feat, capacity, n_dev);
    REN_CHECK_LAUNCH();
}

extern "C" int ren_visibility(const int64_t *offsets, const int32_t *counts, int64_t n_rays,
                              const float *sigmas, const float *t_starts, const float *t_ends,
                              float early_stop_eps, float alpha_thre, uint8_t *keep,
                              int32_t *kept_counts, void *stream) {
    if (!offsets || !counts || !sigmas || !t_starts || !t_ends || !keep || !kept_counts || n_rays < 0)
        return REN_ERR_BAD_ARG;
    if (n_rays == 0) return REN_OK;
    hipLaunchKernelGGL(visibility_kernel, dim3(ren_blocks(n_rays * 64, 256)), dim3(256), 0, (hipStream_t)stream,
                       offsets, counts, n_rays, sigmas, t_starts, t_ends, early_stop_eps, alpha_thre, keep,
                       kept_counts);
    REN_CHECK_LAUNCH();
}

extern "C" int ren_compact_samples(const int64_t *offsets, const int32_t *counts,
                                   const int64_t *new_offsets, int64_t n_rays, const uint8_t *keep,
                                   const float *t_starts, const float *t_ends,
                                   int32_t *out_ray_indices, float *out_t_starts, float *out_t_ends,
                                   void *stream) {
    if (!offsets || !counts || !new_offsets || !keep || !t_starts || !t_ends || !out_ray_indices ||
        !out_t_starts || !out_t_ends || n_rays < 0)
        return REN_ERR_BAD_ARG;
    if (n_rays == 0) return REN_OK;
    hipLaunchKernelGGL(compact_kernel, dim3(ren_blocks(n_rays * 64, 256)), dim3(256), 0, (hipStream_t)stream,
                       offsets, counts, new_offsets, n_rays, keep, t_starts, t_ends, out_ray_indices,
                       out_t_starts, out_t_ends);
    REN_CHECK_LAUNCH();
}

extern "C" int ren_compact_features(const int64_t *offsets, const int32_t *counts, const int64_t *new_offsets,
                                    int64_t n_rays, const uint8_t *keep, const float *feat_in, float *feat_out,
                                    void *stream) {
    if (!offsets || !counts || !new_offsets || !keep || !feat_in || !feat_out || n_rays < 0) return REN_ERR_BAD_ARG;
    if (n_rays == 0) return REN_OK;
    hipLaunchKernelGGL(compact_features_kernel, dim3(ren_blocks(n_rays * 64, 256)), dim3(256), 0, (hipStream_t)stream,
                       offsets, counts, new_offsets, n_rays, keep, feat_in, feat_out);
    REN_CHECK_LAUNCH();
}

extern "C" int ren_pack_info(const int32_t *ray_indices, int64_t n, int64_t n_rays, int64_t *offsets,
                             int32_t *counts, void *stream) {
    if (!offsets || !counts || n < 0 || n_rays < 0 || (n > 0 && !ray_indices)) return REN_ERR_BAD_ARG;
    if (n_rays == 0) return REN_OK;
    hipLaunchKernelGGL(zero_i32_kernel, dim3(ren_blocks(n_rays, 256)), dim3(256), 0, (hipStream_t)stream,
                       counts, n_rays);
    if (n > 0)
        hipLaunchKernelGGL(pack_info_kernel, dim3(ren_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream,
                           ray_indices, n, offsets, counts);
    hipLaunchKernelGGL(fill_empty_offsets_kernel, dim3(ren_blocks(n_rays, 256)), dim3(256), 0,
                       (hipStream_t)stream, offsets, counts, n_rays);
    REN_CHECK_LAUNCH();
}
