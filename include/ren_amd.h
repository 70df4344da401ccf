/* ren_amd.h -- C ABI of the MI355X-native (gfx950) Robust e-NeRF volume-rendering path.
 *
 * One shared object (robust_e_nerf_amd/csrc/libren_amd.so, built by
 * `hipcc --offload-arch=gfx950 -shared -fPIC`), `extern "C"` only, POD arguments.
 *
 * Conventions:
 *  - every `*_d` / unnamed array pointer is a DEVICE pointer owned by the caller
 *    (PyTorch-ROCm allocations); pointers documented "host" are read on the host
 *    before the launch (small descriptors: aabb, level table, scalars);
 *  - the library never allocates, frees or synchronises; kernels are enqueued on the
 *    `stream` argument (a hipStream_t passed as void*, 0 = default stream) and outputs
 *    are valid in stream order;
 *  - return value: REN_OK (0) or a negative ren_status; nothing throws across the ABI;
 *  - no mutable global state: the library is re-entrant, one process per GPU.
 *
 * Each entry point cites the reference interface (file:line under the reference repo)
 * it replaces.  Third-party ops the reference imports (nerfacc 0.3.1, tinycudann,
 * roma 1.2.7) are cited through their reference call sites.
 */
#ifndef REN_AMD_H
#define REN_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum ren_status {
    REN_OK = 0,
    REN_ERR_BAD_ARG = -1,      /* null pointer / negative size / bad enum            */
    REN_ERR_UNSUPPORTED = -2,  /* configuration outside what the kernels implement   */
    REN_ERR_LAUNCH = -3        /* hipGetLastError() != hipSuccess after the launch   */
} ren_status;

/* nerfacc.ContractionType (robust_e_nerf/models/robust_e_nerf.py:214-218) */
enum { REN_CT_AABB = 0, REN_CT_TANH = 1, REN_CT_SPHERE = 2 };

#define REN_MAX_LEVELS 16

/* Multi-resolution hash grid level table (tcnn HashGrid; robust_e_nerf/external/ngp.py:166-170,
 * configs/train/synthetic.yaml:62-69).  Host struct, passed by pointer, copied into the
 * kernel arguments.  n_features_per_level is fixed at 2. */
typedef struct ren_grid_desc {
    int32_t  n_levels;                 /* <= REN_MAX_LEVELS; fused MLP kernels need 16     */
    float    scale[REN_MAX_LEVELS];    /* exp2f(l*log2f(b))*N_min - 1                      */
    uint32_t res[REN_MAX_LEVELS];      /* ceilf(scale)+1                                   */
    uint32_t size[REN_MAX_LEVELS];     /* entries in the level                             */
    uint32_t offset[REN_MAX_LEVELS];   /* first entry of the level                         */
    uint32_t hashed[REN_MAX_LEVELS];   /* 1 = spatial hash, 0 = dense index                */
} ren_grid_desc;

/* Scene / field geometry shared by the field kernels.  Host struct. */
typedef struct ren_scene_desc {
    float   aabb[6];                   /* NGPradianceField.aabb (ngp.py:152)               */
    int32_t contraction_type;          /* REN_CT_*  (ngp.py:231-237)                       */
} ren_scene_desc;

/* Packed sample stream (nerfacc packed layout; external/utils.py:106-119):
 *   ray_indices[n] int32, t_starts[n], t_ends[n] float32, sorted by ray.
 * rays_o / rays_d are (n_rays,3) float32.  Sample position = o + d*(t0+t1)/2
 * (external/utils.py:68-72). */

/* ---- library info ------------------------------------------------------------------ */
int ren_abi_version(void);                       /* bumps when a signature changes      */
/* Verification / tuning knobs: process-wide, read by the launch functions, initialised once from the environment
 * variable of the same name (REN_HGB_NO_PAIRS, ...).  None changes results.
 *   REN_KNOB_HGB_NO_PAIRS      1: single-update records on the hashed levels of the binned scatter as well
 *   REN_KNOB_HGB_HALVE_REGIONS 1: halve the bin regions so the overflow path (global atomics) runs
 *   REN_KNOB_MARCH_SEQUENTIAL  1: sequential occupancy marcher instead of the speculative one; 2 / 4 / 8 / 16: that many speculative
 *                              lanes per ray whatever the ray count (default: by ray count, csrc/ren_sampling.hip)
 *   REN_KNOB_HG_VARIANT        atomic hash-grid backward: bit 0 XCD-affine level mapping, bit 1 lane-pair atomics (default 2)
 *   REN_KNOB_VFIELD_PLAIN      1: arch mlp, bf16 mode: the forward / backward kernels without the software-pipelined epilogue
 *   REN_KNOB_HGB_SUBREGION     binned scatter, which of a pair bin's 8 sub-regions a workgroup appends to: 1 (default) = the one
 *                              of the XCD it runs on, 0 = (workgroup index / 8) % 8, i.e. every sub-region written from all XCDs
 *                              (the A/B of the per-XCD layout: same code, same cursors, only the line sharing differs) */
enum { REN_KNOB_HGB_NO_PAIRS = 0, REN_KNOB_HGB_HALVE_REGIONS = 1, REN_KNOB_MARCH_SEQUENTIAL = 2, REN_KNOB_HG_VARIANT = 3,
       REN_KNOB_VFIELD_PLAIN = 4, REN_KNOB_HGB_SUBREGION = 5, REN_KNOB_COUNT = 6 };
/* `activations` argument of the ren_mlp_* / ren_vanilla_* entry points (ABI 24; until ABI 23 a process-wide knob): the
 * activation alternatives of the YAML (models/nerf.py:8-29), i.e. MODEL configuration, passed per call so that renderers
 * with different sets can launch from different host threads / streams of one process.  Code: bits 0-1 base hidden layers
 * (0 softplus beta 100, 1 relu), bits 2-3 density (0 shifted_trunc_exp, 1 softplus, 2 shifted_softplus), bits 4-5 head hidden
 * layers (0 | 1 as the base), bits 6-7 radiance (0 softplus, 1 sigmoid).  0 = every shipped config.  The exact-f32 kernels
 * (ren_mlp_fwd/bwd[_save/_saved/_bf16], ren_mlp_fwd/bwd_jvp, ren_mlp_fwd_jvp2) take every code, the arch-mlp output heads
 * (ren_vanilla_heads_*) its density / radiance kinds; the bf16-matrix-core kernels (ren_mlp_*_x) and the fused arch-mlp field
 * (ren_vanilla_fwd / _bwd) implement code 0 only and return REN_ERR_UNSUPPORTED otherwise. */
int ren_set_knob(int32_t knob, int32_t value);    /* REN_OK or REN_ERR_BAD_ARG */
int ren_get_knob(int32_t knob);
const char *ren_build_info(void);                /* "gfx950 ..."                        */

/* ---- pose interpolation + ray generation ------------------------------------------
 * Replaces LinearTrajectory.forward (robust_e_nerf/models/trajectories.py:30-91, with
 * utils/tensor_ops.py:83-180 and roma 1.2.7) and NeRF.pixel_params_to_ray
 * (robust_e_nerf/models/nerf.py:206-228).
 * ts[B] float64 ns; tab_ts[C] int64; tab_pos[C,3], tab_quat[C,4] (XYZW) float32.
 * Outputs (any may be NULL): pos[B,3], rot[B,9] row-major. */
int ren_trajectory_fwd(const double *ts, int64_t B, const int64_t *tab_ts, const float *tab_pos,
                       const float *tab_quat, int64_t C, float *pos, float *rot, void *stream);
/* Kinv[9] device, px[B,2], pos[B,3], rot[B,9] -> rays_o[B,3], rays_d[B,3] */
int ren_raygen_fwd(const float *Kinv, const float *px, const float *pos, const float *rot,
                   int64_t B, float *rays_o, float *rays_d, void *stream);
/* both in one launch: rays_o/rays_d[R,3] of the poses at ts[R] through pixels px[i % px_rows] (px[px_rows,2]; the start
 * and end renders of a training step share the events' pixels: R = 2B, px_rows = B).  Same arithmetic as the two calls. */
int ren_pose_rays_fwd(const double *ts, int64_t R, const float *px, int64_t px_rows, const float *Kinv,
                      const int64_t *tab_ts, const float *tab_pos, const float *tab_quat, int64_t C,
                      float *rays_o, float *rays_d, void *stream);

/* ---- sampling ---------------------------------------------------------------------
 * nerfacc.ray_marching as called at robust_e_nerf/external/utils.py:106-119. */
/* ray_aabb_intersect: aabb host[6]; miss => t_min=t_max=1e10; t_min clamped >= 0.
 * near/far: pass NaN for "None" (external/utils.py:112-113).                          */
int ren_ray_aabb_intersect(const float *rays_o, const float *rays_d, int64_t n_rays,
                           const float *aabb_host, float near_plane, float far_plane,
                           float *t_min, float *t_max, void *stream);
/* Two-pass marching.  mode 0: occupancy-grid marching (reference semantics), mode 1:
 * uniform comb of n_uniform intervals over [t_min,t_max) (fixed-S sampler).
 * jitter[n_rays] (may be NULL): mode 0 -> t_min += u*step (stratified);
 *                               mode 1 -> comb shifted by u*delta.
 * binary: uint8/bool grid (res[0]*res[1]*res[2]), x-major.  roi host[6], res host[3].
 * Pass 1 (t_starts==NULL): writes counts[n_rays].  Pass 2: offsets[n_rays] (int64,
 * exclusive cumsum of counts) given, writes ray_indices/t_starts/t_ends.
 * interval_cache (optional, mode 0): float[n_rays * cache_cap * 2].  Pass 1 keeps the first cache_cap
 * intervals of every ray in it; pass 2 (same cache, and counts given as well) copies them instead of marching
 * again and re-marches only the rays with more than cache_cap samples.  Same output either way.
 * mode | REN_MARCH_VERIFIED_DIV: the three divisions by the roi's extents per visited cell run as a multiply-add sequence
 * that ren_march_div_check() has shown to be bit-identical to the division for THIS roi (same output; -15 % on the pass). */
#define REN_MARCH_VERIFIED_DIV 0x100
int ren_ray_march(const float *rays_o, const float *rays_d, const float *t_min,
                  const float *t_max, const float *jitter, int64_t n_rays,
                  const float *roi_host, const int32_t *res_host, const uint8_t *binary,
                  int32_t contraction_type, float step_size, float cone_angle,
                  int32_t mode, int32_t n_uniform,
                  const int64_t *offsets, int32_t *counts,
                  int32_t *ray_indices, float *t_starts, float *t_ends, float *interval_cache,
                  int32_t cache_cap, void *stream);

/* Exhaustive check behind REN_MARCH_VERIFIED_DIV: for each of the three extents b = roi[3 + k] - roi[k] (host floats, 2^-60 < b <
 * 2^60) every float a with 2^-100 < |a| < 2^100 is divided both ways -- a / b and q0 = a y, q = fma(fma(-q0, b, a), y, q0) with
 * y = RN(1 / b) -- and mismatches[0] (device) receives the number of pairs that differ in any bit (~1e10 divisions, ~20 ms; once
 * per scene box).  REN_ERR_UNSUPPORTED for a box with a corner coordinate of 0 (or below 2^-60): position differences could then
 * fall below the checked range.  The flag is for finite rays with |position| < 2^100. */
int ren_march_div_check(const float *roi, int64_t *mismatches, void *stream);
/* exclusive cumsum of counts[n] (int32) -> offsets[n] (int64), total[1] (int64) */
/* out[n] = iid U[0, 1) floats (24 bits), Philox4x32-10 keyed by `seed` at stream position `offset` (e.g. the step number): the
 * per-ray jitter of stratified sampling without a framework RNG launch (models/nerf.py:209-215 draws it with torch.rand). */
int ren_uniform(uint64_t seed, uint64_t offset, int64_t n, float *out, void *stream);
int ren_exclusive_scan(const int32_t *counts, int64_t n, int64_t *offsets, int64_t *total,
                       int64_t *scratch1024 /* int64[1024] device scratch, may be NULL (slow path) */, void *stream);

/* ---- device-side sample counts (ABI 24) ------------------------------------------------------------
 * The reference reads the number of marched / visible samples back to the host in the middle of every render
 * (external/utils.py:106-119, models/nerf.py:279-286) because it sizes the next tensors.  Entry points with an `n_dev`
 * argument take instead: n = the CAPACITY of the per-sample arrays (it sizes the launch), n_dev = device address of the real
 * count (int64); kernels work on min(n, *n_dev) samples.  n_dev == NULL: n is the count, as before.
 * ren_count_guard: after ren_exclusive_scan left the total on the device, compare it with the capacity.  Fits: *n_out =
 * total.  Does not fit: every counts[r] = 0 (the per-ray kernels write nothing: an empty render), *n_out = 0;
 * counts_also (may be NULL) is cleared with it -- the guard after the visibility pass clears the MARCHED counts as well,
 * which is what ren_compact_samples / ren_compact_features walk.
 * stats (may be NULL): stats[0] = the total as found, stats[1] = 1 if it did not fit, else 0.  The host reads `stats`
 * after it has enqueued the rest of the step -- a wait for the sampling kernels only -- and repeats an overflowed step
 * with larger arrays before the optimiser runs (engine.py: RenderCfg.device_counts).  ren_frag_zero_tail: zero the lanes beyond the count in the last 32-sample block of
 * a fragment-layout feature array (what the host-count path does before ren_compact_features). */
int ren_count_guard(int32_t *counts, int32_t *counts_also, int64_t n_rays, const int64_t *total, int64_t capacity,
                    int64_t *n_out, int64_t *stats, void *stream);
/* ren_exclusive_scan followed by ren_count_guard; ONE launch (one workgroup) for up to 65 536 rays */
int ren_scan_guard(int32_t *counts, int32_t *counts_also, int64_t n_rays, int64_t *offsets, int64_t *total, int64_t capacity,
                   int64_t *n_out, int64_t *stats, int64_t *scratch1024, void *stream);
int ren_frag_zero_tail(float *feat, int64_t capacity, const int64_t *n_dev, void *stream);

/* nerfacc.render_visibility inside ray_marching (sigma_fn branch): per ray
 * T = excl. cumprod(1-alpha); keep = T >= early_stop_eps (& alpha >= alpha_thre if >0).
 * Writes keep[n] (uint8) and kept_counts[n_rays]. */
int ren_visibility(const int64_t *offsets, const int32_t *counts, int64_t n_rays,
                   const float *sigmas, const float *t_starts, const float *t_ends,
                   float early_stop_eps, float alpha_thre, uint8_t *keep, int32_t *kept_counts,
                   void *stream);
/* stream compaction of the packed sample arrays by keep[] using new_offsets (exclusive
 * cumsum of kept_counts) */
int ren_compact_samples(const int64_t *offsets, const int32_t *counts, const int64_t *new_offsets,
                        int64_t n_rays, const uint8_t *keep, const float *t_starts,
                        const float *t_ends, int32_t *out_ray_indices, float *out_t_starts,
                        float *out_t_ends, void *stream);
/* Fragment-layout features (32 floats per sample, see ren_hashgrid_fwd layout 1) of the kept samples, moved to
 * their compacted positions (same offsets/counts/new_offsets/keep as ren_compact_samples).  feat_out must hold
 * ceil(n_kept/32) blocks; the caller zeroes its last block (padding lanes). */
int ren_compact_features(const int64_t *offsets, const int32_t *counts, const int64_t *new_offsets, int64_t n_rays,
                         const uint8_t *keep, const float *feat_in, float *feat_out, void *stream);
/* (offsets, counts) from sorted ray_indices[n] (nerfacc unpack_info inverse) */
int ren_pack_info(const int32_t *ray_indices, int64_t n, int64_t n_rays, int64_t *offsets,
                  int32_t *counts, void *stream);

/* ---- multi-resolution hash grid ------------------------------------------------------
 * tcnn.Encoding HashGrid/Linear (robust_e_nerf/external/ngp.py:166-170,240).
 * Input either x_unit[n,3] (unit-cube positions; seam API) or, when x_unit==NULL, the
 * packed sample stream (positions formed and contracted in-kernel, ngp.py:231-237).
 * layout 0: feat[n, 2L] row-major (tcnn output);  layout 1: MFMA fragment order
 *   feat[((i>>5)*16 + l)*64 + f*32 + (i&31)]  (i sample, l level, f feature), n padded to 32. */
int ren_hashgrid_fwd(const ren_grid_desc *grid, const float *table, const float *x_unit,
                     const ren_scene_desc *scene, const float *rays_o, const float *rays_d,
                     const int32_t *ray_indices, const float *t_starts, const float *t_ends,
                     int64_t n, int32_t layout, float *feat, const int64_t *n_dev, void *stream);
/* d(table) += scatter of dfeat (same layouts); atomics, non-deterministic order */
int ren_hashgrid_bwd(const ren_grid_desc *grid, float *grad_table, const float *x_unit,
                     const ren_scene_desc *scene, const float *rays_o, const float *rays_d,
                     const int32_t *ray_indices, const float *t_starts, const float *t_ends,
                     int64_t n, int32_t layout, const float *dfeat, void *stream);

/* Same result as ren_hashgrid_bwd (grad_table += scatter of dfeat) but WITHOUT per-update global
 * atomics: the updates are counting-sorted by 16 384-entry table bin into `workspace` (HBM) and
 * each bin is accumulated in LDS, then added to the table with plain coalesced read-modify-writes
 * (see csrc/ren_hashgrid_binned.hip).  ~5x faster than the atomic scatter on MI355X, where global
 * atomics execute at the memory side.  workspace: ren_hashgrid_bwd_binned_workspace_bytes(n)
 * bytes of device scratch (1 280 B per sample + 64 KiB).  Needs level sizes <= 2^19. */
int64_t ren_hashgrid_bwd_binned_workspace_bytes(int64_t n);
int ren_hashgrid_bwd_binned(const ren_grid_desc *grid, float *grad_table, const float *x_unit,
                            const ren_scene_desc *scene, const float *rays_o, const float *rays_d,
                            const int32_t *ray_indices, const float *t_starts, const float *t_ends,
                            int64_t n, int32_t layout, const float *dfeat, void *workspace,
                            const int64_t *n_dev, void *stream);

/* ---- fused NGP MLPs --------------------------------------------------------------------
 * NGPradianceField.query_density / _query_rgb / forward (robust_e_nerf/external/ngp.py:230-280)
 * with MLP.forward (external/mlp.py:99-113), SHEncoder degree 4 (external/sh_encoder.py:28-93),
 * Softplus(beta=100) hidden, shifted_trunc_exp density, Softplus(beta=1) radiance
 * (models/nerf.py:17-29; configs/train/synthetic.yaml:72-84).
 * mlp_params: one device array of float32 in torch nn.Linear layout, concatenated:
 *   base.w0[64,32] base.b0[64] base.wo[16,64] base.bo[16]
 *   head.w0[64,31] head.b0[64] head.w1[64,64] head.b1[64] head.wo[C,64] head.bo[C]
 * feat: fragment layout (layout 1 above).  Per-sample geometry (selector ngp.py:238, view
 * direction) comes either from x_world[n,3] (+ dirs[n,3], may be NULL when density_only) -- the
 * field(x, dirs) seam -- or, when x_world==NULL, from the packed sample stream.
 * density_only!=0: only sigma is produced (sigma_fn pre-pass, external/utils.py:68-81;
 * occ_eval_fn, models/nerf.py:197-198).
 * base_out (may be NULL): raw base-MLP outputs, fragment layout [((i>>5)*8+g)*64+lane]
 * (ceil(n/32)*512 floats), saved for the backward pass. */
int ren_mlp_fwd(const float *mlp_params, int32_t radiance_dim, int32_t activations, const float *feat,
                const ren_scene_desc *scene, const float *x_world, const float *dirs,
                const float *rays_o, const float *rays_d, const int32_t *ray_indices,
                const float *t_starts, const float *t_ends, int64_t n, int32_t density_only,
                float *rgb, float *sigma, float *base_out, void *stream);
/* floats of device scratch ren_mlp_bwd needs for the per-wave weight-gradient slabs */
int64_t ren_mlp_bwd_workspace_floats(int32_t radiance_dim);
/* backward of ren_mlp_fwd: d_rgb[n,C], d_sigma[n] -> dfeat (fragment layout, ceil(n/32)*1024
 * floats) and grad_mlp_params (+=, same concatenated layout).  rgb = forward output;
 * d_base: scratch of ceil(n/32)*512 floats (gradient w.r.t. base_out, fragment layout);
 * workspace: ren_mlp_bwd_workspace_floats() floats.  Deterministic (no atomics). */
int ren_mlp_bwd(const float *mlp_params, int32_t radiance_dim, int32_t activations, const float *feat,
                const float *base_out, const ren_scene_desc *scene,
                const float *x_world, const float *dirs,
                const float *rays_o, const float *rays_d, const int32_t *ray_indices,
                const float *t_starts, const float *t_ends, int64_t n,
                const float *rgb, const float *d_rgb, const float *d_sigma,
                float *d_base, float *dfeat, float *grad_mlp_params, float *workspace, void *stream);

/* ---- volume rendering (packed) ----------------------------------------------------------
 * rendering() (robust_e_nerf/external/vol_rendering.py:16-128): nerfacc
 * render_weight_from_density + 3x accumulate_along_rays + background compose.
 * Outputs colors[n_rays,C], opacities[n_rays], depths[n_rays] (un-normalised, :111-126);
 * weights[n] and trans[n] (may be NULL) are saved for the backward pass / seam API.
 * bkgd[C] device (NULL = no background). */
int ren_composite_fwd(const int64_t *offsets, const int32_t *counts, int64_t n_rays,
                      const float *t_starts, const float *t_ends, const float *sigmas,
                      const float *rgbs, int32_t C, const float *bkgd,
                      float *colors, float *opacities, float *depths,
                      float *weights, float *trans, void *stream);
/* backward: g_colors[n_rays,C], g_opac[n_rays] (NULL=0), g_depth[n_rays] (NULL=0) and/or
 * g_weights[n] (NULL=0: per-sample upstream gradient of the weights themselves, i.e. the backward of
 * nerfacc.render_weight_from_density used on its own; then rgbs/g_colors/d_rgbs may be NULL)
 * -> d_sigmas[n], d_rgbs[n,C], d_bkgd_per_ray[n_rays,C] (NULL ok).
 * ren_composite_fwd with rgbs==NULL computes weights/trans/opacities/depths only. */
int ren_composite_bwd(const int64_t *offsets, const int32_t *counts, int64_t n_rays,
                      const float *t_starts, const float *t_ends, const float *sigmas,
                      const float *rgbs, int32_t C, const float *bkgd,
                      const float *weights, const float *trans, const float *opacities,
                      const float *g_colors, const float *g_opac, const float *g_depth,
                      const float *g_weights,
                      float *d_sigmas, float *d_rgbs, float *d_bkgd_per_ray, void *stream);

/* ---- event loss ------------------------------------------------------------------------------
 * ContrastThreshold.forward (models/event_generation_params.py:72-84), Loss.log_intensity_diff
 * (loss_metric/loss.py:32-74) and the 1/C^k normalisation + weighting of
 * RobustENeRF.training_step (models/robust_e_nerf.py:432-443,470-486), fused with its own
 * backward.  Inputs per event i: intensity_start/end[B] (render + min_modeled_intensity,
 * :867), target[B] (= ts_diff * ev_log_diff/(end-start), float32), valid[B] (uint8, NULL =
 * all valid).  err_fn: 0 l1, 1 mse, 2 mape.
 * fwd: loss_sum[2] (device) = {sum of errors over valid events, number of valid events}; the
 *      mean loss is loss_sum[0]/loss_sum[1] (an empty mask gives 0/0 = NaN like the reference).
 * bwd: g_start[B], g_end[B] = d(scale * mean error)/d(intensity); reads the count from
 *      loss_sum[1] on the device (no host sync); scale = param weight * loss weight. */
int ren_event_loss_fwd(const float *intensity_start, const float *intensity_end, const float *target,
                       const uint8_t *valid, int64_t B, int32_t err_fn, float *loss_sum, void *stream);
int ren_event_loss_bwd(const float *intensity_start, const float *intensity_end, const float *target,
                       const uint8_t *valid, int64_t B, int32_t err_fn, float scale,
                       const float *loss_sum, float *g_start, float *g_end, void *stream);

/* The same loss straight from the render outputs of the batched start / end pass (colors[2B, C], opacities[2B]; rows
 * [0, B) = start render): the intensity epilogue of RobustENeRF.render_pixels (models/robust_e_nerf.py:865-871: +
 * min_modeled_intensity, is_valid = opacity > 0 of either render when use_validity), `bayering` (:887-890: channel_idx[B]
 * uint8 or NULL), the log difference (:432-435) and the masked mean in ONE forward and ONE backward launch.
 * bwd writes g_colors[2B, C] (zeros in the channels an event does not see), and optionally intensity[2B], pred[B] (=
 * log I_end - log I_start), valid[B] and loss[1] = scale * loss_sum[0] / loss_sum[1] -- all on the device. */
int ren_event_diff_loss_fwd(const float *colors, const float *opacities, const uint8_t *channel_idx, int32_t C,
                            float min_intensity, const float *target, int32_t use_validity, int64_t B, int32_t err_fn,
                            float *loss_sum, void *stream);
int ren_event_diff_loss_bwd(const float *colors, const float *opacities, const uint8_t *channel_idx, int32_t C,
                            float min_intensity, const float *target, int32_t use_validity, int64_t B, int32_t err_fn,
                            float scale, const double *scale_dev, const float *loss_sum, float *g_colors, float *intensity,
                            float *pred, uint8_t *valid, float *loss, void *stream);
/* scale_dev (may be NULL): a device double that multiplies `scale` -- the 1/C^k param weight of a TRAINABLE mean contrast
 * threshold (event_params[4] = 1/C, [5] = 1/C^2 of ren_event_params_refresh), so that no kernel argument waits for a host
 * read of a parameter (robust_e_nerf.py:470-486). */

/* ---- event batch glue ---------------------------------------------------------------------------
 * ren_event_prepare: ContrastThreshold.forward and RefractoryPeriod.forward
 * (models/event_generation_params.py:72-84,196-203), the supervision timestamps of
 * RobustENeRF.training_step (models/robust_e_nerf.py:319-357) and both loss targets
 * (loss_metric/loss.py:39-42,63-66) for B events in one launch:
 *   ev = num_pos C_p - num_neg C_n (float32);  start = start_ts + tau (float64);
 *   ts_diff = (end - start) u_ts_diff;  ts_start = lerp(start, max(end - ts_diff, start), u_diff_start);
 *   ts_end = min(ts_start + ts_diff, end);  target_diff = float(ts_diff ev / (end - start));
 *   ts_grad = lerp(ts_start, ts_end, u_grad), target_grad = float(ev / (end - start))   [optional, NULL to skip];
 *   dts_* = d ts_* / d tau  [optional: only when the refractory period is trained].
 * ren_event_param_grad: d(loss term)/d(raw C_p/C_n ratio) and the direct d(loss term)/d(tau) through the loss
 * target with the rendered prediction held fixed -- the closed form of what the reference obtains by autograd
 * through event_generation_params.py:51-84 and loss.py:32-74 with the scaling of robust_e_nerf.py:470-486.
 * kind 0: pred = log I_end - log I_start against target_diff; kind 1: pred = d log I / dt against target_grad.
 * param_weight_power k: the 1/C^k normalisation (0: none).  Accumulates (+=) into ct_grad[1] (float32) and
 * tau_grad[1] (float64) on the device; either may be NULL. */
int ren_event_prepare(const int64_t *start_ts, const int64_t *end_ts, const int64_t *num_pos, const int64_t *num_neg,
                      const double *u_ts_diff, const double *u_diff_start, const double *u_grad, int64_t B, float c_p,
                      float c_n, double tau, double *ts_start, double *ts_end, float *target_diff, double *ts_grad,
                      float *target_grad, double *dts_start, double *dts_end, double *dts_grad, const double *event_params,
                      void *stream);
int ren_event_param_grad(int32_t kind, int32_t err_fn, int32_t param_weight_power, const float *pred,
                         const uint8_t *valid, const int64_t *start_ts, const int64_t *end_ts, const int64_t *num_pos,
                         const int64_t *num_neg, const double *u_ts_diff, int64_t B, float c_p, float c_n,
                         float raw_ratio, double tau, float weight, float *ct_grad, double *tau_grad,
                         const double *event_params, void *stream);
/* event_params (may be NULL: the scalar arguments count): device double[8] written by ren_event_params_refresh --
 * [0] C_p, [1] C_n, [2] raw ratio, [3] tau, [4] 1 / mean C, [5] 1 / (mean C)^2, [6] raw tau (clamped).  The kernels then take
 * C_p, C_n, raw and tau from there: with a trainable ratio / refractory period the step reads no parameter on the host.
 * ren_event_params_refresh: C_p = softplus(raw ratio) C_n (float32, event_generation_params.py:51-70); the raw refractory
 * period clamped to +-logit(1e-4) tau_max IN PLACE and tau = tau_max sigmoid(raw / tau_max) in float64 (:170-185,
 * modules.py:58-74).  ren_tau_adam_step: torch.optim.Adam (no weight decay) on that float64 scalar from d loss / d tau
 * (x d tau / d raw), state = {exp_avg, exp_avg_sq} on the device, `step` 1-based; clears tau_grad (robust_e_nerf.py:804-807). */
int ren_event_params_refresh(const float *ct_raw, float c_n, double *tau_raw, double tau_max, double *event_params,
                             void *stream);
int ren_tau_adam_step(double *tau_raw, double *tau_grad, double *state, double tau_max, double lr, double beta1, double beta2,
                      double eps, int64_t step, double grad_scale, void *stream);

/* ---- optimiser ----------------------------------------------------------------------------------
 * torch.optim.Adam step as configured by RobustENeRF.configure_optimizers
 * (models/robust_e_nerf.py:782-813): L2-style weight decay (grad += wd*p), bias correction.
 * grad_scale multiplies the gradient first (1/world_size after an all-reduce SUM).
 * zero_grad != 0: the gradient buffer is cleared in the same pass. */
int ren_adam_step(float *param, float *grad, float *exp_avg, float *exp_avg_sq, int64_t n,
                  float lr, float beta1, float beta2, float eps, float weight_decay,
                  int64_t step, float grad_scale, int32_t zero_grad, void *stream);

/* ---- optimiser state on the device (ABI 25) -- for a step captured in a hipGraph (engine.Trainer._graph_step), whose
 * launches cannot carry the step number as an argument, and whose optimiser must not run when a device-side sample count
 * (above) did not fit its arrays.  hyper = device double[8]: [REN_HY_STEP] Adam step of the float32 groups,
 * [REN_HY_SKIP] sticky skip word, [REN_HY_BC1/2] 1 - beta^step, [REN_HY_TAU_STEP], [REN_HY_TAU_BC1/2] the same for the
 * float64 group of the refractory period.  ren_step_tick: once per optimiser step, BEFORE its Adam launches -- raises skip
 * when the overflow words of stats_a / stats_b (int64[4] of ren_scan_guard: [1], [3]; either may be NULL) are set, otherwise
 * advances the step(s) and the corrections.  ren_adam_step_dev / ren_tau_adam_step_dev = ren_adam_step / ren_tau_adam_step
 * with the corrections from `hyper` (same arithmetic: robust_e_nerf.py:782-813); with skip raised they change NOTHING
 * (parameters, moments and gradients stay as they are: the host repeats the step and clears the word). */
#define REN_HY_STEP 0
#define REN_HY_SKIP 1
#define REN_HY_BC1 2
#define REN_HY_BC2 3
#define REN_HY_TAU_STEP 4
#define REN_HY_TAU_BC1 5
#define REN_HY_TAU_BC2 6
int ren_step_tick(double *hyper, double beta1, double beta2, const int64_t *stats_a, const int64_t *stats_b, int32_t tick_tau,
                  void *stream);
int ren_adam_step_dev(float *param, float *grad, float *exp_avg, float *exp_avg_sq, int64_t n,
                      float lr, float beta1, float beta2, float eps, float weight_decay,
                      const double *hyper, float grad_scale, int32_t zero_grad, void *stream);
int ren_tau_adam_step_dev(double *tau_raw, double *tau_grad, double *state, double tau_max, double lr, double beta1,
                          double beta2, double eps, const double *hyper, double grad_scale, void *stream);

/* ---- occupancy grid -------------------------------------------------------------------------------
 * nerfacc.OccupancyGrid._update as driven by NeRF.update_occ_grid (models/nerf.py:170-204). */
/* cell indices[m] (int64) + jitter[m,3] -> world positions x[m,3] (contract_inv) and
 * valid[m] (sphere contraction drops |x-0.5| >= 0.5).  roi host[6], res host[3]. */
int ren_occgrid_cell_points(const int64_t *indices, const float *jitter, int64_t m,
                            const float *roi_host, const int32_t *res_host,
                            int32_t contraction_type, float *x_world, uint8_t *valid, void *stream);
/* occs[idx] = max(occs[idx]*decay, sigma*step) for valid cells (models/nerf.py:197-198).
 * step_sizes[m] may be NULL (then `step_size` is used; cone_angle == 0). */
int ren_occgrid_ema(float *occs, const int64_t *indices, const uint8_t *valid, const float *sigma,
                    const float *step_sizes, float step_size, int64_t m, float ema_decay,
                    void *stream);
/* The same for cell samples that may hold duplicates (past warm-up nerfacc draws cells with replacement; its indexed
 * assignment keeps an arbitrary candidate): deterministic, a cell takes the largest of its new occupancies, decayed once.
 * scratch_cells: one float per grid cell (contents irrelevant). */
int ren_occgrid_ema_unique(float *occs, const int64_t *indices, const uint8_t *valid, const float *sigma,
                           const float *step_sizes, float step_size, int64_t m, float ema_decay, float *scratch_cells,
                           void *stream);
/* binary = occs > min(mean(occs), occ_thre);  scratch[2] device floats. */
int ren_occgrid_binarize(const float *occs, int64_t cells, float occ_thre, uint8_t *binary,
                         float *scratch, void *stream);

/* ---- forward-mode tangent path for the log-intensity-GRADIENT loss ---------------------------------
 * The reference computes d(log I)/d(timestamp) per ray with autograd.gradient(..., create_graph=True)
 * (robust_e_nerf/models/robust_e_nerf.py:383-409, utils/autograd.py:4-34) and back-propagates through
 * it.  Here every stage carries a tangent ("*d" arrays = d/dt, per unit of the timestamp) next to its
 * value, and the *_bwd_jvp entry points are the reverse pass over the (value, tangent) pair.  Sample
 * placement is not differentiated (external/vol_rendering.py:36-37).  Fragment layouts as above. */
/* LinearTrajectory.forward + d/dt: dpos[B,3], drot[B,9] */
int ren_trajectory_jvp(const double *ts, int64_t B, const int64_t *tab_ts, const float *tab_pos,
                       const float *tab_quat, int64_t C, float *pos, float *rot, float *dpos, float *drot,
                       void *stream);
/* pixel_params_to_ray + d/dt: rays_do[B,3], rays_dd[B,3] */
int ren_raygen_jvp(const float *Kinv, const float *px, const float *pos, const float *rot, const float *dpos,
                   const float *drot, int64_t B, float *rays_o, float *rays_d, float *rays_do, float *rays_dd,
                   void *stream);
/* hash features and their tangent (x = o + d tm, xd = od + dd tm) */
int ren_hashgrid_fwd_jvp(const ren_grid_desc *grid, const float *table, const ren_scene_desc *scene,
                         const float *rays_o, const float *rays_d, const float *rays_do, const float *rays_dd,
                         const int32_t *ray_indices, const float *t_starts, const float *t_ends, int64_t n,
                         float *feat, float *featd, const int64_t *n_dev, void *stream);
/* grad_table += w dfeat + wd dfeatd (per-update atomics) */
int ren_hashgrid_bwd_jvp(const ren_grid_desc *grid, float *grad_table, const ren_scene_desc *scene,
                         const float *rays_o, const float *rays_d, const float *rays_do, const float *rays_dd,
                         const int32_t *ray_indices, const float *t_starts, const float *t_ends, int64_t n,
                         const float *dfeat, const float *dfeatd, void *stream);
/* the same through the LDS-binned scatter (workspace as for ren_hashgrid_bwd_binned) */
int ren_hashgrid_bwd_binned_jvp(const ren_grid_desc *grid, float *grad_table, const ren_scene_desc *scene,
                                const float *rays_o, const float *rays_d, const float *rays_do,
                                const float *rays_dd, const int32_t *ray_indices, const float *t_starts,
                                const float *t_ends, int64_t n, const float *dfeat, const float *dfeatd,
                                void *workspace, void *stream);
/* Either of the two binned calls restricted to the levels whose bit is set in level_mask (rays_do / rays_dd / dfeatd all
 * NULL: value-only call; all given: the tangent call).  Data-parallel training (replaces DDP's bucketed
 * all-reduce of scripts/run.py:81-93) runs the fine levels first and all-reduces their slice of the table gradient while
 * the coarse levels are scattered. */
int ren_hashgrid_bwd_binned_levels(const ren_grid_desc *grid, float *grad_table, const float *x_unit,
                                   const ren_scene_desc *scene, const float *rays_o, const float *rays_d,
                                   const int32_t *ray_indices, const float *t_starts, const float *t_ends, int64_t n,
                                   int32_t layout, const float *dfeat, const float *rays_do, const float *rays_dd,
                                   const float *dfeatd, uint32_t level_mask, void *workspace, const int64_t *n_dev, void *stream);
/* The binned backward in phases: `begin` (clear, count, offsets -- needs the sample stream only), any number of `scatter`
 * calls over sample ranges [first, first + m) of the SAME stream (first on a 32-sample block; every call takes the whole
 * stream's n and pointers), and ONE `finish` (partition, accumulate, flush into grad_table).  Stream-ordered: begin before
 * every scatter, every scatter before finish -- the caller may run the scatters on another stream than whatever produces
 * the next range's dfeat (engine.py: MLP backward of chunk k + 1 beside the scatter of chunk k).  Same workspace and
 * results as ren_hashgrid_bwd_binned; replaces the tcnn backward at external/ngp.py:166-170. */
int ren_hashgrid_bwd_binned_begin(const ren_grid_desc *grid, const float *x_unit, const ren_scene_desc *scene,
                                  const float *rays_o, const float *rays_d, const int32_t *ray_indices,
                                  const float *t_starts, const float *t_ends, int64_t n, int32_t layout,
                                  void *workspace, void *stream);
int ren_hashgrid_bwd_binned_scatter(const ren_grid_desc *grid, float *grad_table, const float *x_unit,
                                    const ren_scene_desc *scene, const float *rays_o, const float *rays_d,
                                    const int32_t *ray_indices, const float *t_starts, const float *t_ends,
                                    int64_t n, int32_t layout, const float *dfeat, int64_t first, int64_t m,
                                    void *workspace, void *stream);
int ren_hashgrid_bwd_binned_finish(const ren_grid_desc *grid, float *grad_table, int64_t n, int32_t layout,
                                   void *workspace, void *stream);
/* fused MLPs with tangent: rgb, rgbd [n,C]; sigma, sigmad [n]; base_out, base_outd (ceil(n/32)*512 floats) */
int ren_mlp_fwd_jvp(const float *mlp_params, int32_t radiance_dim, int32_t activations, const float *feat, const float *featd,
                    const ren_scene_desc *scene, const float *rays_o, const float *rays_d, const float *rays_dd,
                    const int32_t *ray_indices, const float *t_starts, const float *t_ends, int64_t n,
                    float *rgb, float *rgbd, float *sigma, float *sigmad, float *base_out, float *base_outd,
                    void *stream);
int64_t ren_mlp_bwd_jvp_workspace_floats(int32_t radiance_dim);
/* reverse pass: (d_rgb, d_rgbd, d_sigma, d_sigmad) -> dfeat, dfeatd (fragment) and grad_mlp_params (+=).
 * scratch: ceil(n/32)*5120 floats; workspace: ren_mlp_bwd_jvp_workspace_floats() floats. */
int ren_mlp_bwd_jvp(const float *mlp_params, int32_t radiance_dim, int32_t activations, const float *feat, const float *featd,
                    const float *base_out, const float *base_outd, const ren_scene_desc *scene,
                    const float *rays_o, const float *rays_d, const float *rays_dd, const int32_t *ray_indices,
                    const float *t_starts, const float *t_ends, int64_t n, const float *rgb,
                    const float *d_rgb, const float *d_rgbd, const float *d_sigma, const float *d_sigmad,
                    float *scratch, float *dfeat, float *dfeatd, float *grad_mlp_params, float *workspace,
                    void *stream);
/* The same two calls on the bf16 matrix cores (csrc/ren_mlp_jvp_x.hip; default tangent path since ABI v15).
 * mode 6: split-bf16 products at fp32 accuracy (mode 3: two pieces, three products); mode 1: plain bf16 operands with fp32 accumulation for value AND tangent
 * of every nn.Linear (BASELINE configs[2]: bf16 MLP + fp32 composite with the log-intensity-gradient loss of
 * robust_e_nerf/models/robust_e_nerf.py:383-409 switched on).  Buffers, scratch and layouts as for the calls above. */
int ren_mlp_fwd_jvp_x(const float *mlp_params, int32_t radiance_dim, int32_t activations, int32_t mode, const float *feat, const float *featd,
                      const ren_scene_desc *scene, const float *rays_o, const float *rays_d, const float *rays_dd,
                      const int32_t *ray_indices, const float *t_starts, const float *t_ends, int64_t n,
                      float *rgb, float *rgbd, float *sigma, float *sigmad, float *base_out, float *base_outd,
                      const int64_t *n_dev, void *stream);
int64_t ren_mlp_bwd_jvp_x_workspace_floats(int32_t radiance_dim);
int ren_mlp_bwd_jvp_x(const float *mlp_params, int32_t radiance_dim, int32_t activations, int32_t mode, const float *feat, const float *featd,
                      const float *base_out, const float *base_outd, const ren_scene_desc *scene,
                      const float *rays_o, const float *rays_d, const float *rays_dd, const int32_t *ray_indices,
                      const float *t_starts, const float *t_ends, int64_t n, const float *rgb, const float *d_rgb,
                      const float *d_rgbd, const float *d_sigma, const float *d_sigmad, float *scratch, float *dfeat,
                      float *dfeatd, float *grad_mlp_params, float *workspace, const int64_t *n_dev, void *stream);
/* compositing with tangent: colors/colords [n_rays,C], opacities/opacds [n_rays]; saves weights, trans,
 * eds (exclusive prefix of sigmad*dt) [n] for the reverse pass */
int ren_composite_fwd_jvp(const int64_t *offsets, const int32_t *counts, int64_t n_rays, const float *t_starts,
                          const float *t_ends, const float *sigmas, const float *sigmads, const float *rgbs,
                          const float *rgbds, int32_t C, const float *bkgd, float *colors, float *colords,
                          float *opacities, float *opacds, float *weights, float *trans, float *eds, void *stream);
int ren_composite_bwd_jvp(const int64_t *offsets, const int32_t *counts, int64_t n_rays, const float *t_starts,
                          const float *t_ends, const float *sigmas, const float *sigmads, const float *rgbs,
                          const float *rgbds, int32_t C, const float *bkgd, const float *weights,
                          const float *trans, const float *eds, const float *opacities, const float *opacds,
                          const float *g_colors, const float *g_colords, float *d_sigmas, float *d_sigmads,
                          float *d_rgbs, float *d_rgbds, float *d_bkgd_per_ray, void *stream);
/* Epilogue of the tangent (l_grad) render in one launch (robust_e_nerf.py:390-398,865-871, `bayering` :887-890):
 * intensity = colors[:, channel] + min_modeled_intensity, intensity_dot = colords[:, channel], valid = opacity > 0 (NULL:
 * a background parameter makes every ray valid), dlog_dt = intensity_dot / intensity (NULL: not wanted).  C = 1: channel_idx
 * may be NULL. */
int ren_rate_epilogue(const float *colors, const float *colords, const float *opacities, const uint8_t *channel_idx,
                      int32_t C, int64_t n, float min_modeled_intensity, float *intensity, float *intensity_dot,
                      uint8_t *valid, float *dlog_dt, void *stream);
/* d loss / d tau through the poses (robust_e_nerf.py:340-357: tau moves every supervision timestamp):
 * tau_grad[0] += sum_i (g_a[i] x_a[i] + g_b[i] x_b[i]) dts[i], float64, one launch (g_b, x_b may both be NULL). */
int ren_tau_pose_grad(const float *g_a, const float *x_a, const float *g_b, const float *x_b, const double *dts,
                      int64_t n, double *tau_grad, void *stream);
/* torch.nn.utils.weight_norm(module) over the Linear layers of an MLP (external/ngp.py:207-228, external/mlp.py:303-319;
 * dim 0): W[r, :] = g[r] v[r, :] / ||v[r, :]||.  `raw` is the packed parameter block with v in the weight slots, `eff` the
 * block the field kernels read; `layers` (HOST memory) lists the reparametrised weights as n_layers x {weight offset, rows,
 * cols, offset of the layer's first g in `g`} (<= 16 layers); every other element of the block is copied through.
 * _bwd: d_raw = gradient w.r.t. (v, biases, plain weights) from d_eff = gradient w.r.t. the effective block, d_g[r] =
 * dW[r, :] . v / ||v||; zero_d_eff != 0 clears d_eff afterwards (the accumulator of the next step).  Two small launches each. */
int ren_weight_norm_fwd(const float *raw, const float *g, const int32_t *layers, int32_t n_layers, int64_t n_params,
                        float *eff, void *stream);
int ren_weight_norm_bwd(const float *raw, const float *g, float *d_eff, const int32_t *layers, int32_t n_layers,
                        int64_t n_params, float *d_raw, float *d_g, int32_t zero_d_eff, void *stream);
/* Loss.log_intensity_grad (loss_metric/loss.py:43-57): pred = intensity_dot / intensity vs target;
 * same loss_sum / scale conventions as ren_event_loss_fwd/bwd */
int ren_grad_loss_fwd(const float *intensity, const float *intensity_dot, const float *target,
                      const uint8_t *valid, int64_t B, int32_t err_fn, float *loss_sum, void *stream);
int ren_grad_loss_bwd(const float *intensity, const float *intensity_dot, const float *target,
                      const uint8_t *valid, int64_t B, int32_t err_fn, float scale, const double *scale_dev,
                      const float *loss_sum, float *g_intensity, float *g_intensity_dot, float *loss, void *stream);
/* scale_dev: as for ren_event_diff_loss_bwd; loss (may be NULL): loss[0] = scale * loss_sum[0] / loss_sum[1] */

/* ---- bf16 MLP mode (BASELINE configs[2]: "bf16 MLP with fp32 composite") ------------------------------ *
 * Same arguments as ren_mlp_fwd / ren_mlp_bwd.  Every linear layer sees bf16-rounded (nearest-even) inputs;
 * `mlp_params_bf16` is the caller's bf16-rounded copy of the parameter block (still f32 storage); products
 * accumulate in fp32, bias and activations stay fp32.  The backward is the exact derivative of that forward
 * (straight-through rounding) and accumulates into the fp32 master gradient. */
int ren_mlp_fwd_bf16(const float *mlp_params_bf16, int32_t C, int32_t activations, const float *feat, const ren_scene_desc *scene,
                     const float *x_world, const float *dirs, const float *rays_o, const float *rays_d,
                     const int32_t *ray_indices, const float *t_starts, const float *t_ends, int64_t n,
                     int32_t density_only, float *rgb, float *sigma, float *base_out, void *stream);
int ren_mlp_bwd_bf16(const float *mlp_params_bf16, int32_t C, int32_t activations, const float *feat, const float *base_out,
                     const ren_scene_desc *scene, const float *x_world, const float *dirs, const float *rays_o,
                     const float *rays_d, const int32_t *ray_indices, const float *t_starts,
                     const float *t_ends, int64_t n, const float *rgb, const float *d_rgb, const float *d_sigma,
                     float *d_base, float *dfeat, float *grad_mlp_params, float *workspace, void *stream);

/* ---- saved-activation variants (the training fast path) -------------------------------------------------- *
 * ren_mlp_fwd_save additionally stores the post-activation values of the three hidden layers
 * (ren_mlp_act_save_floats(n) floats, 768 B/sample); ren_mlp_bwd_saved reads them instead of recomputing
 * the forward (128 f32 MFMAs + 192 softplus per 32 samples).  bf16 != 0 selects the bf16 numerics mode
 * (mlp_params is then the rounded copy).  Otherwise the arguments of ren_mlp_fwd / ren_mlp_bwd. */
int64_t ren_mlp_act_save_floats(int64_t n);
int ren_mlp_fwd_save(const float *mlp_params, int32_t C, int32_t activations, int32_t bf16, const float *feat,
                     const ren_scene_desc *scene, const float *x_world, const float *dirs, const float *rays_o,
                     const float *rays_d, const int32_t *ray_indices, const float *t_starts,
                     const float *t_ends, int64_t n, float *rgb, float *sigma, float *base_out, float *act_save,
                     void *stream);
int ren_mlp_bwd_saved(const float *mlp_params, int32_t C, int32_t activations, int32_t bf16, const float *feat, const float *base_out,
                      const float *act_save, const ren_scene_desc *scene, const float *x_world, const float *dirs,
                      const float *rays_o, const float *rays_d, const int32_t *ray_indices, const float *t_starts,
                      const float *t_ends, int64_t n, const float *rgb, const float *d_rgb, const float *d_sigma,
                      float *d_base, float *dfeat, float *grad_mlp_params, float *workspace, void *stream);

/* ---- split-bf16 MLP kernels (csrc/ren_mlp_x.hip) ------------------------------------------------------------- *
 * The same fused MLPs on the bf16 matrix cores.  mode 6: every fp32 operand is split exactly into three bf16
 * pieces and six bf16 MFMAs per k-chunk reproduce the fp32 product to fp32 round-off (the default training
 * path: the f32 MFMA shares its pipe with the VALU on gfx950, the bf16 MFMA does not).  mode 1: plain bf16
 * operands (BASELINE configs[2]); takes the fp32 parameter block and rounds it itself.  mode 3: two bf16 pieces per
 * operand and three products a1 b2 + a2 b1 + a1 b1 -- "each float32 number as the sum of two bfloat16 numbers", i.e.
 * `float32_matmul_precision: high` of the reference's YAMLs (scripts/run.py:34-35): ~16 significant bits per product,
 * half of mode 6's matrix-pipe time.  The same three modes for the *_jvp_x / *_jvp2_x entry points.
 * Arguments as ren_mlp_fwd_save / ren_mlp_bwd_saved; act_save may be NULL in the forward (inference).
 * flags: REN_MLP_DENSITY_ONLY (base network and sigma only; rgb may be NULL), REN_MLP_SHARE_CU (one persistent
 * workgroup per CU instead of two, leaving half of each CU to a kernel running on another stream: the engine
 * runs the hash encoding of the next sample chunk beside the MLP of the current one). */
#define REN_MLP_DENSITY_ONLY 1
#define REN_MLP_SHARE_CU 2
int ren_mlp_fwd_x(const float *mlp_params, int32_t C, int32_t activations, int32_t mode, const float *feat, const ren_scene_desc *scene,
                  const float *x_world, const float *dirs, const float *rays_o, const float *rays_d,
                  const int32_t *ray_indices, const float *t_starts, const float *t_ends, int64_t n,
                  int32_t flags, float *rgb, float *sigma, float *base_out, float *act_save, const int64_t *n_dev, void *stream);
int64_t ren_mlp_bwd_x_workspace_floats(int32_t C);
int ren_mlp_bwd_x(const float *mlp_params, int32_t C, int32_t activations, int32_t mode, const float *feat, const float *base_out,
                  const float *act_save, const ren_scene_desc *scene, const float *x_world, const float *dirs,
                  const float *rays_o, const float *rays_d, const int32_t *ray_indices, const float *t_starts,
                  const float *t_ends, int64_t n, const float *rgb, const float *d_rgb, const float *d_sigma,
                  float *d_base, float *dfeat, float *grad_mlp_params, float *workspace, int32_t grid_cus, const int64_t *n_dev, void *stream);
/* grid_cus: persistent workgroups (= CUs) the two backward kernels occupy, 1 .. 255; 0 (or >= 256) = all 256.  The chunked
 * backward of engine.py lowers it while a scatter runs on the second stream (until ABI 23: knob REN_KNOB_MLP_BWD_CUS). */

/* ---- second-order forward tangent (value, d/dt, d2/dt2): d(l_grad)/d(tau) --------------------------- *
 * The gradient-loss prediction d(log I)/dt is evaluated at ts_g(tau); its derivative w.r.t. the refractory
 * period needs d2 I/dt2 per ray (models/robust_e_nerf.py:340-357,383-409 differentiated through
 * `grad.ts`; the reference obtains it as a third-order autograd graph).  Forward only, same sample
 * stream as the first-order render.  Arrays with suffix dd hold second time derivatives. */
int ren_trajectory_jvp2(const double *ts, int64_t B, const int64_t *tab_ts, const float *tab_pos,
                        const float *tab_quat, int64_t C, float *pos, float *rot, float *dpos, float *drot,
                        float *ddrot, void *stream);
int ren_raygen_jvp2(const float *Kinv, const float *px, const float *pos, const float *rot, const float *dpos,
                    const float *drot, const float *ddrot, int64_t B, float *rays_o, float *rays_d,
                    float *rays_do, float *rays_dd, float *rays_ddd, void *stream);
int ren_hashgrid_fwd_jvp2(const ren_grid_desc *grid, const float *table, const ren_scene_desc *scene,
                          const float *rays_o, const float *rays_d, const float *rays_do, const float *rays_dd,
                          const float *rays_ddd, const int32_t *ray_indices, const float *t_starts,
                          const float *t_ends, int64_t n, float *feat, float *featd, float *featdd, const int64_t *n_dev,
                          void *stream);
int ren_mlp_fwd_jvp2(const float *mlp_params, int32_t C, int32_t activations, const float *feat, const float *featd,
                     const float *featdd, const ren_scene_desc *scene, const float *rays_o, const float *rays_d,
                     const float *rays_do, const float *rays_dd, const float *rays_ddd,
                     const int32_t *ray_indices, const float *t_starts, const float *t_ends, int64_t n,
                     float *rgb, float *rgbd, float *rgbdd, float *sigma, float *sigmad, float *sigmadd,
                     void *stream);
/* ren_mlp_fwd_jvp2 on the bf16 matrix cores (csrc/ren_jvp2.hip: mlp_fwd_jvp2_x_kernel); mode 6: split-bf16 at fp32
 * accuracy (3: two pieces, three products), mode 1: plain bf16 operands -- the second-order render of a step follows the precision mode of its other
 * MLP kernels (RenderCfg.mlp_kernels = "x") */
int ren_mlp_fwd_jvp2_x(const float *mlp_params, int32_t C, int32_t activations, int32_t mode, const float *feat, const float *featd,
                       const float *featdd, const ren_scene_desc *scene, const float *rays_o, const float *rays_d,
                       const float *rays_do, const float *rays_dd, const float *rays_ddd,
                       const int32_t *ray_indices, const float *t_starts, const float *t_ends, int64_t n,
                       float *rgb, float *rgbd, float *rgbdd, float *sigma, float *sigmad, float *sigmadd,
                       const int64_t *n_dev, void *stream);
int ren_composite_fwd_jvp2(const int64_t *offsets, const int32_t *counts, int64_t n_rays, const float *t_starts,
                           const float *t_ends, const float *sigmas, const float *sigmads, const float *sigmadds,
                           const float *rgbs, const float *rgbds, const float *rgbdds, int32_t C,
                           const float *bkgd, float *colors, float *colords, float *colorsdd, void *stream);

/* ---- vanilla-NeRF field, `arch: mlp` (external/mlp.py:26-113,126-205,208-243,246-358) ----------------- *
 * Activations are row-major [n_pad][ld] f32, n_pad = n rounded up to 32, ld a multiple of 4 and at least
 * the layer width rounded up to 32, padding columns ZERO (they are read as part of the reduction).
 * Concatenations are column ranges of one buffer.  Weights are torch nn.Linear [n_out][n_in] + bias. */
#define REN_ACT_NONE 0
#define REN_ACT_SOFTPLUS100 1          /* Softplus(beta=100), models/nerf.py:17-29          */
#define REN_ACT_SOFTPLUS1 2            /* Softplus(beta=1)                                  */
#define REN_ACT_TRUNC_EXP_SEL 3        /* selector * exp(z - 1), external/ngp.py:45-65, mlp.py:343 */
/* the YAML's alternatives (models/nerf.py:8-29); backward-data takes REN_ACT_RELU as a previous-layer activation too.  The
 * output-head kernels of arch mlp (ren_vanilla_heads_*) differentiate the density / radiance kinds of REN_KNOB_ACTIVATIONS;
 * the tangent algebra (ren_act_jvp*) takes beta = 0 for relu.  The fused field (ren_vanilla_fwd / _bwd) implements the
 * shipped set only and returns REN_ERR_UNSUPPORTED under a non-zero REN_KNOB_ACTIVATIONS. */
#define REN_ACT_RELU 4
#define REN_ACT_SIGMOID 5
#define REN_ACT_SOFTPLUS1_SEL 6        /* selector * softplus(z)                            */
#define REN_ACT_SHIFTED_SOFTPLUS1_SEL 7 /* selector * softplus(z - 1), models/nerf.py:8-13  */
/* SinusoidalEncoder of the contracted position (63 features -> enc[:, :64], col 63 = 0; optional copy at
 * cat[:, cat_col:cat_col+64]) and of pi*direction (27 features, padded to 32, at view[:, view_col:]);
 * selector[i] = all(0 < contracted x < 1).  Positions from x_world/dirs or from the packed sample stream. */
int ren_freq_encode(const ren_scene_desc *scene, const float *x_world, const float *dirs, const float *rays_o,
                    const float *rays_d, const int32_t *ray_indices, const float *t_starts,
                    const float *t_ends, int64_t n, float *enc, int32_t ld_enc, float *cat, int32_t ld_cat,
                    int32_t cat_col, float *view, int32_t ld_view, int32_t view_col, uint8_t *selector,
                    void *stream);
/* Y[:, :n_out] = act(X[:, :n_in] W^T + b)  (nn.Linear + activation, mlp.py:99-113), n_out <= 256
 * Matrix-core path: OR one of these into `act` (ren_dense_fwd) / `prev_act` (ren_dense_bwd_data):
 * REN_DENSE_F32 exact v_mfma_f32_32x32x2_f32; REN_DENSE_BF16X6 split-bf16, six bf16 MFMAs per k-step at fp32
 * accuracy (2.7x fewer matrix-pipe cycles, co-issues with the VALU); REN_DENSE_BF16 plain bf16 operands. */
#define REN_DENSE_F32 0x000
#define REN_DENSE_BF16X6 0x600
#define REN_DENSE_BF16 0x100
int ren_dense_fwd(const float *X, int32_t ldx, const float *W, const float *bias, int32_t n_out, int32_t n_in,
                  int32_t act, const uint8_t *selector, float *Y, int32_t ldy, int64_t n, void *stream);
/* dX[:, :n_store] (+)= dZ[:, :n_out] W[:, :n_store], then multiplied by prev_act'(Yprev) -> the dZ of the
 * layer below (n_store <= n_in: gradients of concatenated encodings are not needed) */
int ren_dense_bwd_data(const float *dZ, int32_t ldz, const float *W, int32_t n_out, int32_t n_in, int32_t n_store,
                       int32_t prev_act, const float *Yprev, int32_t ldyp, int32_t accumulate, float *dX,
                       int32_t ldx, int64_t n, void *stream);
/* grad_w[n_out][n_in] += dZ^T X, grad_b[n_out] += column sums of dZ (slab-reduced, deterministic); grad_b may be NULL
 * (tangent stream of a layer: no bias) */
/* Activation algebra of the tangent stream of a dense layer (log-intensity-gradient loss with arch mlp): y = softplus_beta(z),
 * yd = s zd with s = 1 - exp(-beta y) taken from the output.  fwd: Yd = s Zd.  bwd: gz = gy s + gyd zd beta s (1 - s),
 * gzd = gyd s.  Row-major buffers; Y (and Yd in fwd) may be column ranges of wider buffers (ld), the others are dense
 * [rows][width]; width and every ld are multiples of 4. */
int ren_act_jvp_fwd(const float *Y, int32_t ldy, const float *Zd, int32_t ldz, float beta, float *Yd, int32_t ldyd,
                    int64_t rows, int32_t width, void *stream);
int ren_act_jvp_bwd(const float *gy, const float *gyd, const float *Y, int32_t ldy, const float *Zd, float beta, float *gz,
                    float *gzd, int64_t rows, int32_t width, void *stream);
int64_t ren_dense_bwd_weight_workspace_floats(int32_t n_out, int32_t n_in, int32_t n_splits);
/* Matrix-core path of ren_dense_bwd_weight: OR (REN_DENSE_BF16X6 or REN_DENSE_BF16) << 8 into n_splits (bits 16..23) for
 * bf16 MFMAs: REN_DENSE_BF16X6 = operands split into three bf16 pieces, six MFMAs per product pair (fp32 round-off);
 * REN_DENSE_BF16 = plain bf16 operands (one MFMA); default: exact f32 MFMA. */
int ren_dense_bwd_weight(const float *dZ, int32_t ldz, const float *X, int32_t ldx, int32_t n_out, int32_t n_in,
                         int64_t n, int32_t n_splits, float *grad_w, float *grad_b, float *workspace,
                         void *stream);
/* output activations backward: dz_rgb[n_pad][32] = g_rgb * softplus1'(rgb), dz_sigma[n_pad][32] (col 0) =
 * g_sigma * d sigma/d raw; padding zeroed */
int ren_vanilla_heads_bwd(const float *g_rgb, const float *rgb, const float *g_sigma, const float *sigma,
                          int64_t n, int32_t C, int32_t activations, float *dz_rgb, float *dz_sigma, void *stream);

/* ---- arch mlp: the whole field as one launch per pass (csrc/ren_vfield.hip) ------------------------------------------------
 * Replaces NerfMLP.forward / query_density (robust_e_nerf/external/mlp.py:126-205: eight hidden Linear + Softplus(beta = 100)
 * with the skip concatenation after layer 4, sigma layer, bottleneck, 283 -> 128 -> C colour head) and its autograd:
 * activations stay in registers from layer to layer.  `params`: the field's parameter block (reference state-dict order,
 * torch layout).  `mode`: REN_DENSE_BF16 (1: bf16 operands, saved copies in bf16), REN_DENSE_BF16X6 (6: three-piece split,
 * fp32 round-off, saved copies in fp32) or 3 (two pieces, three products = `float32_matmul_precision: high`; layouts of
 * mode 6, an image of two pieces; every ren_vanilla_* entry point, not the per-layer ren_dense_* ones).  `image` (ren_vanilla_image_bytes) is the MFMA-fragment form of the weights: rebuild
 * it with ren_vanilla_prep whenever the parameters change.  `saved` / `dz` (ren_vanilla_saved_bytes each) hold the layers'
 * activations / pre-activation gradients in the kernels' fragment layout (documented in ren_vfield.hip).
 * fwd: enc [n_pad][ld_enc >= 64], view [n_pad][ld_view >= 32], selector [n_pad] are the outputs of ren_freq_encode (rows
 * n .. n_pad zero); sigma [n_pad]; rgb4 [n_pad][4] (columns >= C zero) or NULL for the density only (then view may be NULL
 * and saved must be).  bwd: dz_rgb / dz_sigma [n_pad][32] as written by ren_vanilla_heads_bwd.  ren_vanilla_bwd_weight ADDS
 * dW_l = dz_l^T x_{l-1}, db_l = sum dz_l of all twelve layers to `grads` (layout of params), slab-reduced (deterministic).
 * Backward over a sample RANGE of a larger forward pass (bounds the memory of `dz`): row pointers (dz_rgb, dz_sigma, enc, view)
 * and `saved` advanced to the first sample of the range -- a multiple of 256 samples; saved + first_sample * 256 * element
 * size -- n = samples in the range, saved_slot_bytes = ren_vanilla_saved_bytes(mode, n_forward) / 10 (0: the pass itself). */
int64_t ren_vanilla_image_bytes(int32_t mode);
int64_t ren_vanilla_saved_bytes(int32_t mode, int64_t n);
int ren_vanilla_prep(const float *params, int32_t C, int32_t mode, void *image, void *stream);
int ren_vanilla_fwd(const float *enc, int32_t ld_enc, const float *view, int32_t ld_view, const uint8_t *selector,
                    const float *params, int32_t C, int32_t activations, const void *image, int32_t mode, int64_t n, void *saved, float *sigma,
                    float *rgb4, void *stream);
int ren_vanilla_bwd(const float *dz_rgb, const float *dz_sigma, const void *image, int32_t mode, int32_t activations, int64_t n, const void *saved,
                    int64_t saved_slot_bytes, void *dz, void *stream);
/* Value + forward-mode tangent (d/dt) of the whole field in one launch each way -- the third render of the step with the
 * log-intensity-gradient loss (models/robust_e_nerf.py:383-409 through external/mlp.py:126-205 under utils/autograd.py:4-34).
 * REN_DENSE_BF16: value and tangent are the two 32-sample "blocks" of a wave, one launch each way.  REN_DENSE_BF16X6 (fp32
 * round-off): two launches each way -- forward: the value, then the tangent with sp'(z) from the value's saved copies (saved
 * is required); reverse: the tangent side, which leaves its coupling term per layer in `coupling` (ren_vanilla_saved_bytes,
 * required in this mode), then the value side.  Other modes / activation sets: REN_ERR_UNSUPPORTED (per-layer launches).  encd / viewd:
 * d/dt of the two encodings (ren_freq_encode_jvp, same leading dimensions as enc / view); saved / savedd: the layers' value /
 * tangent activations (ren_vanilla_saved_bytes each; both or neither); zsd4 / zod4 [n_pad][4]: tangent pre-activations of the
 * sigma head (column 0) and the colour head (columns < C) for ren_vanilla_heads_jvp / ren_vanilla_heads_bwd_jvp.  bwd_jvp: the
 * four [n_pad][32] gradient buffers of ren_vanilla_heads_bwd_jvp -> dz / dzd (ren_vanilla_saved_bytes each).  The weight
 * gradients are ren_vanilla_bwd_weight(dz, saved, ..) + ren_vanilla_bwd_weight_tangent(dzd, savedd, .., encd, viewd, dzd_rgb,
 * dzd_sigma, ..): the latter adds dW_l += dzd_l^T xd_(l-1) only (the tangent stream has no bias). */
int ren_vanilla_fwd_jvp(const float *enc, int32_t ld_enc, const float *view, int32_t ld_view, const float *encd, const float *viewd,
                        const uint8_t *selector, const float *params, int32_t C, int32_t activations, const void *image, int32_t mode,
                        int64_t n, void *saved, void *savedd, float *sigma, float *rgb4, float *zsd4, float *zod4, void *stream);
int ren_vanilla_bwd_jvp(const float *dz_rgb, const float *dzd_rgb, const float *dz_sigma, const float *dzd_sigma, const void *image,
                        int32_t mode, int32_t activations, int64_t n, const void *saved, const void *savedd, int64_t saved_slot_bytes,
                        void *dz, void *dzd, void *coupling, void *stream);
int ren_vanilla_bwd_weight_tangent(const void *dzd, const void *savedd, int64_t saved_slot_bytes, const float *encd, int32_t ld_enc,
                                   const float *viewd, int32_t ld_view, const float *dzd_rgb, const float *dzd_sigma, int32_t C,
                                   int32_t mode, int64_t n, int32_t n_splits, float *grads, float *workspace, void *stream);
int64_t ren_vanilla_bwd_weight_workspace_floats(int32_t n_splits);
int ren_vanilla_bwd_weight(const void *dz, const void *saved, int64_t saved_slot_bytes, const float *enc, int32_t ld_enc,
                           const float *view, int32_t ld_view, const float *dz_rgb, const float *dz_sigma, int32_t C,
                           int32_t mode, int64_t n, int32_t n_splits, float *grads, float *workspace, void *stream);
/* Tangent streams of the vanilla field (robust_e_nerf/external/mlp.py:208-243,333-358 under
 * utils/autograd.py:4-34 for the log-intensity-gradient loss, models/robust_e_nerf.py:383-409; the second order serves
 * d loss / d tau, see ren_trajectory_jvp2).  ren_freq_encode_jvp: `order`-th time derivative (1 or 2) of ren_freq_encode's
 * position / view features for a packed sample stream whose rays carry o, d and their time derivatives do, dd, ddd
 * (o'' = 0 inside a pose segment); same destinations and layout as ren_freq_encode, rows n..n_pad written as zeros. */
int ren_freq_encode_jvp(const ren_scene_desc *scene, const float *rays_o, const float *rays_d, const float *rays_do,
                        const float *rays_dd, const float *rays_ddd, const int32_t *ray_indices, const float *t_starts,
                        const float *t_ends, int64_t n, int32_t order, float *enc, int32_t ld_enc, float *cat, int32_t ld_cat,
                        int32_t cat_col, float *view, int32_t ld_view, int32_t view_col, void *stream);
/* y = softplus_beta(z) with first and second time derivatives: Yd = s Zd, Ydd = beta s (1 - s) Zd^2 + s Zdd */
int ren_act_jvp2_fwd(const float *Y, int32_t ldy, const float *Zd, const float *Zdd, int32_t ldz, float beta, float *Yd,
                     int32_t ldyd, float *Ydd, int32_t ldydd, int64_t rows, int32_t width, void *stream);
/* Output heads of the tangent streams: rgb = softplus_1(zo), sigma = selector exp(zs - 1) (clamped derivative, ngp.py:45-65).
 * zod / zodd / zsd / zsdd: [n_pad][4] pre-activation tangents (ren_dense_fwd without bias / activation); the second-order
 * arguments (zodd, zsdd, rgbdd, sigmadd) may all be NULL.  rgb* are (n, C), sigma* (n). */
int ren_vanilla_heads_jvp(const float *rgb, const float *sigma, const float *zod, const float *zodd, const float *zsd,
                          const float *zsdd, int64_t n, int32_t C, int32_t activations, float *rgbd, float *rgbdd, float *sigmad, float *sigmadd,
                          void *stream);
/* reverse pass of (rgb, rgbd, sigma, sigmad) -> pre-activation gradients of the value and the tangent stream, each
 * [n_pad][32] zero-padded (as ren_vanilla_heads_bwd) */
int ren_vanilla_heads_bwd_jvp(const float *g_rgb, const float *g_rgbd, const float *g_sigma, const float *g_sigmad,
                              const float *rgb, const float *sigma, const float *zod, const float *zsd, int64_t n, int32_t C, int32_t activations,
                              float *dz_rgb, float *dzd_rgb, float *dz_sigma, float *dzd_sigma, void *stream);

/* ---- utilities ------------------------------------------------------------------------------------- */
/* out[c] = sum_r in[r*C + c]   (C <= 4); scratch512: 512 floats of device scratch (two-stage, deterministic) */
int ren_column_sum(const float *in, int64_t rows, int32_t C, float *out, float *scratch512, void *stream);
/* NeRF.render_bkgd as a softplus-parametrised parameter (models/nerf.py:81-88): bkgd[C] = softplus(raw[C]); and its gradient
 * from the per-ray d_bkgd of ren_composite_bwd: grad_raw[c] += sigmoid(raw[c]) * sum_r d_bkgd_per_ray[r, c] */
int ren_bkgd_param_fwd(const float *raw, int32_t C, float *bkgd, void *stream);
int ren_bkgd_param_grad(const float *d_bkgd_per_ray, int64_t rows, int32_t C, const float *raw, float *grad_raw,
                        float *scratch512, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* REN_AMD_H */
