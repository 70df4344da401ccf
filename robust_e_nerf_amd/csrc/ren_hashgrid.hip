// Multi-resolution hash-grid encoding (tcnn HashGrid, Linear interpolation, F=2), forward and
// parameter backward.  Replaces tcnn.Encoding as instantiated at
// robust_e_nerf/external/ngp.py:166-170 (config configs/train/synthetic.yaml:62-69).
//
// Work decomposition: one thread per (sample, level); blockIdx.y = level so that all
// workgroups in flight gather from the same level's <= 4 MiB table slice (one XCD L2 holds it),
// blockIdx.x walks 256 consecutive packed samples, which along a ray are spatially adjacent, so
// coarse levels hit L1/L2.  Per sample-level: 8 x 8-byte random gathers (forward) or 16 f32
// atomics (backward); the sample position is recomputed from the ray (24 B, cache resident) and
// its (t0,t1) instead of being staged through HBM.
//
// Output layout 1 ("fragment order") is what the fused MLP kernels consume directly as MFMA
// B-operands: feat[((i>>5)*16 + level)*64 + f*32 + (i&31)].
#include "ren_hashgrid_common.h"
#include <stdlib.h>

namespace {

// Block -> (level, sample chunk).  XCD-affine mode: workgroup b runs on XCD b % 8 (observed
// dispatch order; used for speed only), so giving XCD x the levels {x, x+8} keeps each level's
// <= 4 MiB table slice in ONE XCD's 4 MiB L2: gathers hit that L2 and -- more importantly --
// atomics on a cache line are never issued from two different (mutually incoherent) L2s.
__device__ __forceinline__ bool block_to_work(int xcd_affine, int n_levels, int64_t n_chunks, int &lvl,
                                              int64_t &chunk) {
    if (!xcd_affine) { lvl = blockIdx.y; chunk = blockIdx.x; return true; }
    const int64_t b = blockIdx.x;
    // probe (REN_KNOB_HG_VARIANT bit 2, round 6): LEVEL-INNER order -- consecutive workgroups are the 16 levels of one sample
    // chunk, so all 16 table slices (48 MB) are in flight at once: the access pattern a fused per-sample encode + MLP kernel has
    if (xcd_affine & 4) { lvl = (int)(b % n_levels); chunk = b / n_levels; return chunk < n_chunks; }
    const int64_t j = b >> 3;
    const int slot = (int)(j / n_chunks);
    chunk = j - slot * n_chunks;
    lvl = (int)(b & 7) + 8 * slot;
    return lvl < n_levels;
}

template <int LAYOUT, bool FROM_RAYS>
__global__ __launch_bounds__(256) void hashgrid_fwd_kernel(
    GridDev g, const float2 *__restrict__ table, const float *__restrict__ x_unit, ren_scene_dev sc,
    const float *__restrict__ rays_o, const float *__restrict__ rays_d,
    const int32_t *__restrict__ ray_indices, const float *__restrict__ t_starts,
    const float *__restrict__ t_ends, int64_t n, int64_t n_pad, float *__restrict__ feat, int xcd_affine,
    const int64_t *__restrict__ n_dev) {
    int lvl; int64_t chunk;
    if (!block_to_work(xcd_affine, g.n_levels, (n_pad + 255) / 256, lvl, chunk)) return;      // (the launch geometry is the capacity's)
    const int64_t i = chunk * blockDim.x + threadIdx.x;
    if (n_dev) {                                             // device-side count: the fragment padding follows the real count
        n = ren_eff_n(n, n_dev);
        if (LAYOUT == 1) { const int64_t p = ((n + 31) >> 5) << 5; n_pad = p < n_pad ? p : n_pad; } else n_pad = n;
    }
    if (i >= n_pad) return;
    float f0 = 0.f, f1 = 0.f;
    if (i < n) {
        float ux, uy, uz;
        if (FROM_RAYS) {
            float x, y, z; int ray;
            ren_sample_pos(rays_o, rays_d, ray_indices, t_starts, t_ends, i, x, y, z, ray);
            ren_contract(sc, x, y, z, ux, uy, uz);
        } else {
            ux = x_unit[3 * i]; uy = x_unit[3 * i + 1]; uz = x_unit[3 * i + 2];
        }
        const LevelPos p = level_pos(ux, uy, uz, g.scale[lvl]);
        const uint32_t res = g.res[lvl], size = g.size[lvl];
        const bool hashed = g.hashed[lvl] != 0;
        const float2 *tab = table + g.offset[lvl];
        float2 v[8];
        uint32_t idx[8];
        corner_indices8(p.c[0], p.c[1], p.c[2], res, size, hashed, idx);
        gather_corners8(tab, idx, v);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float wx = (c & 1) ? p.w[0] : 1.f - p.w[0];
            const float wy = (c & 2) ? p.w[1] : 1.f - p.w[1];
            const float wz = (c & 4) ? p.w[2] : 1.f - p.w[2];
            const float w = wx * wy * wz;
            f0 += w * v[c].x;
            f1 += w * v[c].y;
        }
    }
    if (LAYOUT == 0) {
        if (i < n) reinterpret_cast<float2 *>(feat)[i * g.n_levels + lvl] = make_float2(f0, f1);
    } else {
        const int64_t b = ((i >> 5) * REN_MAX_LEVELS + lvl) * 64 + (i & 31);
        __builtin_nontemporal_store(f0, feat + b);           // streamed once: keep the level's table slice in L2 instead
        __builtin_nontemporal_store(f1, feat + b + 32);
    }
}

template <int LAYOUT, bool FROM_RAYS, bool PAIR>
__global__ __launch_bounds__(256) void hashgrid_bwd_kernel(
    GridDev g, float *__restrict__ grad_table, const float *__restrict__ x_unit, ren_scene_dev sc,
    const float *__restrict__ rays_o, const float *__restrict__ rays_d,
    const int32_t *__restrict__ ray_indices, const float *__restrict__ t_starts,
    const float *__restrict__ t_ends, int64_t n, const float *__restrict__ dfeat, int xcd_affine) {
    int lvl; int64_t chunk;
    if (PAIR) {
        if (!block_to_work(xcd_affine, g.n_levels, (n + 127) / 128, lvl, chunk)) return;
    } else {
        if (!block_to_work(xcd_affine, g.n_levels, (n + 255) / 256, lvl, chunk)) return;
    }
    // PAIR: two adjacent lanes share a sample and take one feature each, so one atomic
    // instruction covers 32 (sample, corner) pairs x 2 features with 8-byte contiguous targets.
    const int64_t i = PAIR ? chunk * 128 + (threadIdx.x >> 1) : chunk * blockDim.x + threadIdx.x;
    const int feat_sel = PAIR ? (threadIdx.x & 1) : 0;
    if (i >= n) return;
    float d0, d1;
    if (LAYOUT == 0) {
        const float2 d = reinterpret_cast<const float2 *>(dfeat)[i * g.n_levels + lvl];
        d0 = d.x; d1 = d.y;
    } else {
        const int64_t b = ((i >> 5) * REN_MAX_LEVELS + lvl) * 64 + (i & 31);
        d0 = dfeat[b];
        d1 = dfeat[b + 32];
    }
    if (PAIR) { d0 = feat_sel ? d1 : d0; if (d0 == 0.f) return; }
    else if (d0 == 0.f && d1 == 0.f) return;
    float ux, uy, uz;
    if (FROM_RAYS) {
        float x, y, z; int ray;
        ren_sample_pos(rays_o, rays_d, ray_indices, t_starts, t_ends, i, x, y, z, ray);
        ren_contract(sc, x, y, z, ux, uy, uz);
    } else {
        ux = x_unit[3 * i]; uy = x_unit[3 * i + 1]; uz = x_unit[3 * i + 2];
    }
    const LevelPos p = level_pos(ux, uy, uz, g.scale[lvl]);
    const uint32_t res = g.res[lvl], size = g.size[lvl];
    const bool hashed = g.hashed[lvl] != 0;
    float *gt = grad_table + 2 * (size_t)g.offset[lvl];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const uint32_t idx = corner_index(p.c[0] + (c & 1), p.c[1] + ((c >> 1) & 1), p.c[2] + (c >> 2),
                                          res, size, hashed);
        const float wx = (c & 1) ? p.w[0] : 1.f - p.w[0];
        const float wy = (c & 2) ? p.w[1] : 1.f - p.w[1];
        const float wz = (c & 4) ? p.w[2] : 1.f - p.w[2];
        const float w = wx * wy * wz;
        if (PAIR) {
            atomicAdd(gt + 2 * (size_t)idx + feat_sel, w * d0);
        } else {
            atomicAdd(gt + 2 * (size_t)idx, w * d0);
            atomicAdd(gt + 2 * (size_t)idx + 1, w * d1);
        }
    }
}

// Tuning knob (REN_KNOB_HG_VARIANT): bit 0 = XCD-affine level
// mapping, bit 1 = lane-pair feature split in the backward kernel.
int hg_variant() { return ren_knob(REN_KNOB_HG_VARIANT); }

}  // namespace

extern "C" int ren_hashgrid_fwd(const ren_grid_desc *grid, const float *table, const float *x_unit,
                                const ren_scene_desc *scene, const float *rays_o, const float *rays_d,
                                const int32_t *ray_indices, const float *t_starts, const float *t_ends,
                                int64_t n, int32_t layout, float *feat, const int64_t *n_dev, void *stream) {
    GridDev g;
    int rc = make_grid(grid, g);
    if (rc) return rc;
    if (!table || !feat || n < 0 || (layout != 0 && layout != 1)) return REN_ERR_BAD_ARG;
    const bool from_rays = x_unit == nullptr;
    if (from_rays && (!scene || !rays_o || !rays_d || !ray_indices || !t_starts || !t_ends)) return REN_ERR_BAD_ARG;
    if (layout == 1 && g.n_levels != REN_MAX_LEVELS) return REN_ERR_UNSUPPORTED;
    if (n == 0) return REN_OK;
    ren_scene_dev sc = {};
    if (scene) sc = ren_make_scene(scene);
    const int64_t n_pad = layout == 1 ? ((n + 31) / 32) * 32 : n;
    const int variant = hg_variant();
    const int affine = variant & 5;
    const int64_t n_chunks = (n_pad + 255) / 256;
    dim3 grd = (affine & 4) ? dim3((unsigned)(g.n_levels * n_chunks))
             : affine ? dim3((unsigned)(8 * ((g.n_levels + 7) / 8) * n_chunks)) : dim3((unsigned)n_chunks, g.n_levels);
    dim3 blk(256);
    const float2 *tab = reinterpret_cast<const float2 *>(table);
#define LAUNCH(L, R)                                                                                     \
    hipLaunchKernelGGL((hashgrid_fwd_kernel<L, R>), grd, blk, 0, (hipStream_t)stream, g, tab, x_unit, sc, \
                       rays_o, rays_d, ray_indices, t_starts, t_ends, n, n_pad, feat, affine, n_dev)
    if (layout == 0) { if (from_rays) LAUNCH(0, true); else LAUNCH(0, false); }
    else             { if (from_rays) LAUNCH(1, true); else LAUNCH(1, false); }
#undef LAUNCH
    REN_CHECK_LAUNCH();
}

extern "C" int ren_hashgrid_bwd(const ren_grid_desc *grid, float *grad_table, const float *x_unit,
                                const ren_scene_desc *scene, const float *rays_o, const float *rays_d,
                                const int32_t *ray_indices, const float *t_starts, const float *t_ends,
                                int64_t n, int32_t layout, const float *dfeat, void *stream) {
    GridDev g;
    int rc = make_grid(grid, g);
    if (rc) return rc;
    if (!grad_table || !dfeat || n < 0 || (layout != 0 && layout != 1)) return REN_ERR_BAD_ARG;
    const bool from_rays = x_unit == nullptr;
    if (from_rays && (!scene || !rays_o || !rays_d || !ray_indices || !t_starts || !t_ends)) return REN_ERR_BAD_ARG;
    if (layout == 1 && g.n_levels != REN_MAX_LEVELS) return REN_ERR_UNSUPPORTED;
    if (n == 0) return REN_OK;
    ren_scene_dev sc = {};
    if (scene) sc = ren_make_scene(scene);
    const int variant = hg_variant();
    const int affine = variant & 1;
    const bool pair = (variant & 2) != 0;
    const int64_t n_chunks = pair ? (n + 127) / 128 : (n + 255) / 256;
    dim3 grd = affine ? dim3((unsigned)(8 * ((g.n_levels + 7) / 8) * n_chunks)) : dim3((unsigned)n_chunks, g.n_levels);
    dim3 blk(256);
#define LAUNCH(L, R, P)                                                                                          \
    hipLaunchKernelGGL((hashgrid_bwd_kernel<L, R, P>), grd, blk, 0, (hipStream_t)stream, g, grad_table, x_unit, \
                       sc, rays_o, rays_d, ray_indices, t_starts, t_ends, n, dfeat, affine)
#define LAUNCH2(L, R) do { if (pair) LAUNCH(L, R, true); else LAUNCH(L, R, false); } while (0)
    if (layout == 0) { if (from_rays) LAUNCH2(0, true); else LAUNCH2(0, false); }
    else             { if (from_rays) LAUNCH2(1, true); else LAUNCH2(1, false); }
#undef LAUNCH2
#undef LAUNCH
    REN_CHECK_LAUNCH();
}
