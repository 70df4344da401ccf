// Event log-intensity-difference loss (forward + its own backward), fused Adam, and the
// occupancy-grid maintenance kernels.  All bandwidth-trivial elementwise / reduction work that the
// reference spreads over dozens of torch launches per step.
#include "ren_common.h"

namespace {

// ---------------------------------------------------------------------------------- event loss
// loss_metric/loss.py:59-74 with the error functions of :22-30 (l1 / mse / mape, modules.py:77-102)
__device__ __forceinline__ float err_fn_val(int fn, float pred, float tgt) {
    const float d = pred - tgt;
    if (fn == 0) return fabsf(d);
    if (fn == 1) return d * d;
    return fabsf(d) / fmaxf(fabsf(tgt), 2.220446049250313e-16f);
}

__device__ __forceinline__ float err_fn_grad(int fn, float pred, float tgt) {
    const float d = pred - tgt;
    const float sg = (float)((d > 0.f) - (d < 0.f));
    if (fn == 0) return sg;
    if (fn == 1) return 2.f * d;
    return sg / fmaxf(fabsf(tgt), 2.220446049250313e-16f);
}

__global__ __launch_bounds__(256) void event_loss_fwd_kernel(
    const float *__restrict__ i_start, const float *__restrict__ i_end, const float *__restrict__ target,
    const uint8_t *__restrict__ valid, int64_t B, int fn, float *__restrict__ loss_sum) {
    __shared__ float ps[4], pc[4];
    float s = 0.f, c = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < B; i += (int64_t)gridDim.x * blockDim.x) {
        if (valid && !valid[i]) continue;
        const float pred = logf(i_end[i]) - logf(i_start[i]);       // robust_e_nerf.py:432-435
        s += err_fn_val(fn, pred, target[i]);
        c += 1.f;
    }
    s = ren_wave_sum(s);
    c = ren_wave_sum(c);
    if ((threadIdx.x & 63) == 0) { ps[threadIdx.x >> 6] = s; pc[threadIdx.x >> 6] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(loss_sum, ps[0] + ps[1] + ps[2] + ps[3]);
        atomicAdd(loss_sum + 1, pc[0] + pc[1] + pc[2] + pc[3]);
    }
}

__global__ void event_loss_bwd_kernel(const float *__restrict__ i_start, const float *__restrict__ i_end,
                                      const float *__restrict__ target, const uint8_t *__restrict__ valid,
                                      int64_t B, int fn, float scale, const float *__restrict__ loss_sum,
                                      float *__restrict__ g_start, float *__restrict__ g_end) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    float g = 0.f;
    const float is = i_start[i], ie = i_end[i];
    if (!valid || valid[i]) {
        const float pred = logf(ie) - logf(is);
        g = scale / loss_sum[1] * err_fn_grad(fn, pred, target[i]);
    }
    g_end[i] = g / ie;
    g_start[i] = -g / is;
}

// ---- the same loss straight from the render outputs: one forward and one backward launch per step replace the
// intensity epilogue (+ min_modeled_intensity, Bayer channel gather), the validity mask, the two loss kernels, the
// gradient concatenation / channel scatter and the scalar loss arithmetic that used to be ~12 torch launches
struct DiffLossArgs {
    const float *colors, *opac, *target;     // (2B, C): rows [0, B) start render, [B, 2B) end render
    const uint8_t *channel;                  // (B) Bayer channel of the event, NULL: channel 0
    int64_t B;
    int C, fn, use_validity;
    float min_intensity;
};

__device__ __forceinline__ bool diff_loss_event(const DiffLossArgs &a, int64_t i, float &is, float &ie, int &ch) {
    ch = a.channel ? (int)a.channel[i] : 0;
    is = a.colors[i * a.C + ch] + a.min_intensity;                 // robust_e_nerf.py:867, 425-431
    ie = a.colors[(a.B + i) * a.C + ch] + a.min_intensity;
    return !a.use_validity || a.opac[i] > 0.f || a.opac[a.B + i] > 0.f;   // :868-871, 442-443
}

__global__ __launch_bounds__(256) void diff_loss_fwd_kernel(DiffLossArgs a, float *__restrict__ loss_sum) {
    __shared__ float ps[4], pc[4];
    float s = 0.f, c = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.B; i += (int64_t)gridDim.x * blockDim.x) {
        float is, ie; int ch;
        if (!diff_loss_event(a, i, is, ie, ch)) continue;
        s += err_fn_val(a.fn, logf(ie) - logf(is), a.target[i]);
        c += 1.f;
    }
    s = ren_wave_sum(s);
    c = ren_wave_sum(c);
    if ((threadIdx.x & 63) == 0) { ps[threadIdx.x >> 6] = s; pc[threadIdx.x >> 6] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(loss_sum, ps[0] + ps[1] + ps[2] + ps[3]);
        atomicAdd(loss_sum + 1, pc[0] + pc[1] + pc[2] + pc[3]);
    }
}

__global__ void diff_loss_bwd_kernel(DiffLossArgs a, float scale, const double *__restrict__ scale_dev,
                                     const float *__restrict__ loss_sum,
                                     float *__restrict__ g_colors, float *__restrict__ inten, float *__restrict__ pred_out,
                                     uint8_t *__restrict__ valid_out, float *__restrict__ loss_out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (scale_dev) scale = (float)((double)scale * scale_dev[0]);   // weight x param weight(mean contrast), the latter on the device
    if (i == 0 && loss_out) loss_out[0] = loss_sum[0] / loss_sum[1] * scale;   // an empty mask gives NaN like the reference
    if (i >= a.B) return;
    float is, ie; int ch;
    const bool ok = diff_loss_event(a, i, is, ie, ch);
    const float pred = logf(ie) - logf(is);
    const float g = ok ? scale / loss_sum[1] * err_fn_grad(a.fn, pred, a.target[i]) : 0.f;
    for (int c = 0; c < a.C; ++c) {                                  // zeros in the channels the event does not see
        g_colors[i * a.C + c] = c == ch ? -g / is : 0.f;
        g_colors[(a.B + i) * a.C + c] = c == ch ? g / ie : 0.f;
    }
    if (inten) { inten[i] = is; inten[a.B + i] = ie; }
    if (pred_out) pred_out[i] = pred;
    if (valid_out) valid_out[i] = ok ? 1 : 0;
}

// ---------------------------------------------------------------------------------- Adam
// torch.optim.Adam (single-tensor formulation): g += wd*p; m,v EMA; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
__global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ p, float *__restrict__ g,
                                                   float *__restrict__ m, float *__restrict__ v, int64_t n,
                                                   float lr_over_bc1, float beta1, float beta2, float eps,
                                                   float wd, float inv_sqrt_bc2, float grad_scale,
                                                   int zero_grad) {
    const int64_t n4 = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 pp = reinterpret_cast<float4 *>(p)[i];
        float4 gg = reinterpret_cast<float4 *>(g)[i];
        float4 mm = reinterpret_cast<float4 *>(m)[i];
        float4 vv = reinterpret_cast<float4 *>(v)[i];
        float *pa = &pp.x, *ga = &gg.x, *ma = &mm.x, *va = &vv.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float gr = ga[k] * grad_scale + wd * pa[k];
            ma[k] = beta1 * ma[k] + (1.f - beta1) * gr;
            va[k] = beta2 * va[k] + (1.f - beta2) * gr * gr;
            pa[k] -= lr_over_bc1 * ma[k] / (sqrtf(va[k]) * inv_sqrt_bc2 + eps);
        }
        reinterpret_cast<float4 *>(p)[i] = pp;
        reinterpret_cast<float4 *>(m)[i] = mm;
        reinterpret_cast<float4 *>(v)[i] = vv;
        if (zero_grad) reinterpret_cast<float4 *>(g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // tail
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float gr = g[i] * grad_scale + wd * p[i];
        float mi = beta1 * m[i] + (1.f - beta1) * gr;
        float vi = beta2 * v[i] + (1.f - beta2) * gr * gr;
        m[i] = mi; v[i] = vi;
        p[i] -= lr_over_bc1 * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
        if (zero_grad) g[i] = 0.f;
    }
}

// ---- optimiser state on the device (ABI 25): a captured step (hipGraph) cannot carry the step number as a launch argument.
// `hyper` = double[8] (REN_HY_*): [0] Adam step of the float32 groups, [1] skip (sticky: an overflowed device-side sample
// count of this or an earlier step -- every *_dev optimiser kernel then leaves parameters, moments AND gradients alone until
// the host has repeated the step and cleared it), [2] 1 - beta1^step, [3] 1 - beta2^step, [4] step of the tau group,
// [5] / [6] its two corrections.  step_tick_kernel advances it once per optimiser step, before the Adam launches.
__global__ void step_tick_kernel(double *__restrict__ hy, double beta1, double beta2, const int64_t *__restrict__ stats_a,
                                 const int64_t *__restrict__ stats_b, int tick_tau) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    bool skip = hy[REN_HY_SKIP] != 0.0;
    if (stats_a && (stats_a[1] | stats_a[3])) skip = true;
    if (stats_b && (stats_b[1] | stats_b[3])) skip = true;
    hy[REN_HY_SKIP] = skip ? 1.0 : 0.0;
    if (skip) return;
    // the float32 groups: ren_adam_step takes its betas as floats and forms 1 - beta^step from those in double
    const double st = hy[REN_HY_STEP] + 1.0, b1f = (double)(float)beta1, b2f = (double)(float)beta2;
    hy[REN_HY_STEP] = st;
    hy[REN_HY_BC1] = 1.0 - pow(b1f, st);
    hy[REN_HY_BC2] = 1.0 - pow(b2f, st);
    if (tick_tau) {
        const double tt = hy[REN_HY_TAU_STEP] + 1.0;
        hy[REN_HY_TAU_STEP] = tt;
        hy[REN_HY_TAU_BC1] = 1.0 - pow(beta1, tt);
        hy[REN_HY_TAU_BC2] = 1.0 - pow(beta2, tt);
    }
}

// adam_kernel with the bias corrections (and the skip word) read from `hyper`: same arithmetic -- lr / bc1 and 1 / sqrt(bc2)
// are formed in double and rounded to float exactly as ren_adam_step forms them on the host
__global__ __launch_bounds__(256) void adam_dev_kernel(float *__restrict__ p, float *__restrict__ g,
                                                       float *__restrict__ m, float *__restrict__ v, int64_t n,
                                                       float lr, float beta1, float beta2, float eps, float wd,
                                                       const double *__restrict__ hy, float grad_scale, int zero_grad) {
    if (hy[REN_HY_SKIP] != 0.0) return;
    const float lr_over_bc1 = (float)((double)lr / hy[REN_HY_BC1]);
    const float inv_sqrt_bc2 = (float)(1.0 / sqrt(hy[REN_HY_BC2]));
    const int64_t n4 = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 pp = reinterpret_cast<float4 *>(p)[i];
        float4 gg = reinterpret_cast<float4 *>(g)[i];
        float4 mm = reinterpret_cast<float4 *>(m)[i];
        float4 vv = reinterpret_cast<float4 *>(v)[i];
        float *pa = &pp.x, *ga = &gg.x, *ma = &mm.x, *va = &vv.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float gr = ga[k] * grad_scale + wd * pa[k];
            ma[k] = beta1 * ma[k] + (1.f - beta1) * gr;
            va[k] = beta2 * va[k] + (1.f - beta2) * gr * gr;
            pa[k] -= lr_over_bc1 * ma[k] / (sqrtf(va[k]) * inv_sqrt_bc2 + eps);
        }
        reinterpret_cast<float4 *>(p)[i] = pp;
        reinterpret_cast<float4 *>(m)[i] = mm;
        reinterpret_cast<float4 *>(v)[i] = vv;
        if (zero_grad) reinterpret_cast<float4 *>(g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float gr = g[i] * grad_scale + wd * p[i];
        float mi = beta1 * m[i] + (1.f - beta1) * gr;
        float vi = beta2 * v[i] + (1.f - beta2) * gr * gr;
        m[i] = mi; v[i] = vi;
        p[i] -= lr_over_bc1 * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
        if (zero_grad) g[i] = 0.f;
    }
}

// ---------------------------------------------------------------------------------- occupancy grid
struct CellArgs { float roi[6]; int res[3]; int type; };

__global__ void occgrid_cell_points_kernel(const int64_t *__restrict__ indices, const float *__restrict__ jitter,
                                           int64_t m, CellArgs a, float *__restrict__ xw,
                                           uint8_t *__restrict__ valid) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const int64_t idx = indices[i];
    const int iz = (int)(idx % a.res[2]);
    const int iy = (int)((idx / a.res[2]) % a.res[1]);
    const int ix = (int)(idx / ((int64_t)a.res[2] * a.res[1]));
    float u[3] = {((float)ix + jitter[3 * i]) / (float)a.res[0],
                  ((float)iy + jitter[3 * i + 1]) / (float)a.res[1],
                  ((float)iz + jitter[3 * i + 2]) / (float)a.res[2]};
    bool ok = true;
    if (a.type == REN_CT_SPHERE) {
        float cx = u[0] - 0.5f, cy = u[1] - 0.5f, cz = u[2] - 0.5f;
        ok = sqrtf(cx * cx + cy * cy + cz * cz) < 0.5f;
        float x = cx * 4.f, y = cy * 4.f, z = cz * 4.f;
        float mag = sqrtf(x * x + y * y + z * z);
        if (mag > 1.f) {
            float s = 1.f / fmaxf(2.f * mag - mag * mag, 1e-10f);
            x *= s; y *= s; z *= s;
        }
        u[0] = x * 0.5f + 0.5f; u[1] = y * 0.5f + 0.5f; u[2] = z * 0.5f + 0.5f;
    } else if (a.type == REN_CT_TANH) {
#pragma unroll
        for (int k = 0; k < 3; ++k)
            u[k] = atanhf(fminf(fmaxf(u[k] * 2.f - 1.f, -1.f + 1e-6f), 1.f - 1e-6f)) + 0.5f;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) xw[3 * i + k] = u[k] * (a.roi[3 + k] - a.roi[k]) + a.roi[k];
    if (valid) valid[i] = (uint8_t)ok;
}

__global__ void occgrid_ema_kernel(float *__restrict__ occs, const int64_t *__restrict__ indices,
                                   const uint8_t *__restrict__ valid, const float *__restrict__ sigma,
                                   const float *__restrict__ step_sizes, float step_size, int64_t m,
                                   float decay) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    if (valid && !valid[i]) return;
    const float occ = sigma[i] * (step_sizes ? step_sizes[i] : step_size);
    const int64_t idx = indices[i];
    occs[idx] = fmaxf(occs[idx] * decay, occ);
}

// Deterministic variant for samples with duplicate cells (past warm-up nerfacc draws cells with replacement, and its
// `occs[indices] = maximum(occs[indices] * decay, occ)` keeps an arbitrary candidate): a cell takes the LARGEST of its
// candidates, decayed once.  Three passes over a cells-sized scratch: clear the drawn cells, atomic max of the new
// occupancies (non-negative floats order like their bit patterns), then the first thread to claim a cell applies the EMA.
__global__ void occgrid_ema_clear_kernel(const int64_t *__restrict__ indices, int64_t m, float *__restrict__ tmp) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) tmp[indices[i]] = 0.f;
}
__global__ void occgrid_ema_max_kernel(const int64_t *__restrict__ indices, const uint8_t *__restrict__ valid,
                                       const float *__restrict__ sigma, const float *__restrict__ step_sizes,
                                       float step_size, int64_t m, float *__restrict__ tmp) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m || (valid && !valid[i])) return;
    const float occ = fmaxf(sigma[i] * (step_sizes ? step_sizes[i] : step_size), 0.f);
    atomicMax(reinterpret_cast<unsigned int *>(tmp + indices[i]), __float_as_uint(occ));
}
__global__ void occgrid_ema_apply_kernel(float *__restrict__ occs, const int64_t *__restrict__ indices,
                                         const uint8_t *__restrict__ valid, int64_t m, float decay, float *__restrict__ tmp) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m || (valid && !valid[i])) return;
    const int64_t idx = indices[i];
    const float best = atomicExch(tmp + idx, -1.f);              // claimed: later duplicates see -1
    if (best >= 0.f) occs[idx] = fmaxf(occs[idx] * decay, best);
}

__global__ __launch_bounds__(256) void sum_kernel(const float *__restrict__ x, int64_t n, float *__restrict__ out) {
    __shared__ float ps[4];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        s += x[i];
    s = ren_wave_sum(s);
    if ((threadIdx.x & 63) == 0) ps[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, ps[0] + ps[1] + ps[2] + ps[3]);
}

__global__ void binarize_kernel(const float *__restrict__ occs, int64_t cells, float occ_thre,
                                const float *__restrict__ sum, uint8_t *__restrict__ binary) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cells) return;
    const float thr = fminf(sum[0] / (float)cells, occ_thre);
    binary[i] = (uint8_t)(occs[i] > thr);
}

// ---------------------------------------------------------------------------------- event batch glue
// a2-a4 of SURVEY 8(a) in ONE launch instead of ~15 float64 torch kernels (at the reference's 2^20-sample
// budget the step is launch-bound on this glue): event correction (event_generation_params.py:72-84,196-203),
// supervision timestamps (robust_e_nerf.py:319-357) and both loss targets (loss.py:39-42,63-66), plus the
// derivative of every supervision timestamp w.r.t. the refractory period (chain rule of the same lines).
struct PrepArgs {
    const int64_t *start_ts, *end_ts, *num_pos, *num_neg;
    const double *u_ts_diff, *u_diff_start, *u_grad;
    int64_t B;
    float c_p, c_n;
    double tau;
    double *ts_start, *ts_end, *ts_grad, *dts_start, *dts_end, *dts_grad;
    float *target_diff, *target_grad;
    const double *ep;                        // device-resident event parameters (REN_EP_*), or NULL: the scalars above
};

// Device-resident event-generation parameters (ren_event_params_refresh): with a trainable C_p / C_n ratio or refractory
// period the values move with every optimiser step; kept on the device, no kernel argument of the next step waits for a
// host read of them (the reference reads nothing either: they are nn.Parameters, event_generation_params.py:51-84,162-203).
enum { REN_EP_CP = 0, REN_EP_CN = 1, REN_EP_RAW = 2, REN_EP_TAU = 3, REN_EP_INV_C = 4, REN_EP_INV_C2 = 5, REN_EP_TAU_RAW = 6 };

__device__ __forceinline__ double lerp64(double a, double b, double w) {      // torch.lerp's two-sided formula
    const double diff = b - a;
    return fabs(w) < 0.5 ? a + w * diff : b - diff * (1.0 - w);
}

__global__ __launch_bounds__(256) void event_prepare_kernel(PrepArgs a) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.B) return;
    if (a.ep) { a.c_p = (float)a.ep[REN_EP_CP]; a.c_n = (float)a.ep[REN_EP_CN]; a.tau = a.ep[REN_EP_TAU]; }
    const float ev = __fsub_rn(__fmul_rn((float)a.num_pos[i], a.c_p), __fmul_rn((float)a.num_neg[i], a.c_n));
    const double end = (double)a.end_ts[i];
    const double start = (double)a.start_ts[i] + a.tau;
    const double u1 = a.u_ts_diff[i], u2 = a.u_diff_start[i];
    const double span = end - start;
    const double ts_diff = span * u1;
    const bool late = end - ts_diff > start;                       // max(end - ts_diff, start)
    const double hi = late ? end - ts_diff : start;
    const double d_start = lerp64(start, hi, u2);
    const bool inside = d_start + ts_diff < end;                   // min(d_start + ts_diff, end)
    const double d_end = inside ? d_start + ts_diff : end;
    const double rate = (double)ev / span;
    a.ts_start[i] = d_start;
    a.ts_end[i] = d_end;
    a.target_diff[i] = (float)(ts_diff * rate);
    if (a.target_grad) a.target_grad[i] = (float)rate;
    // d/d tau: start' = 1, ts_diff' = -u1, hi' = late ? u1 : 1
    const double g_start = (1.0 - u2) + u2 * (late ? u1 : 1.0);
    const double g_end = inside ? g_start - u1 : 0.0;
    if (a.dts_start) { a.dts_start[i] = g_start; a.dts_end[i] = g_end; }
    if (a.ts_grad) {
        const double u3 = a.u_grad[i];
        a.ts_grad[i] = lerp64(d_start, d_end, u3);
        if (a.dts_grad) a.dts_grad[i] = (1.0 - u3) * g_start + u3 * g_end;
    }
}

// d(loss term)/d(raw contrast-threshold ratio) and the DIRECT d(loss term)/d(tau) through the loss target, with
// the rendered prediction held fixed (event_generation_params.py:51-84,196-203, loss.py:32-74,
// robust_e_nerf.py:470-486): closed form of what the reference gets from autograd, one workgroup.
//   L = w pw(C) mean_valid e(pred - target),  C = (C_p + C_n)/2,  C_p = softplus(raw) C_n,  pw = C^-k
struct ParamGradArgs {
    const float *pred;
    const uint8_t *valid;
    const int64_t *start_ts, *end_ts, *num_pos, *num_neg;
    const double *u_ts_diff;
    int64_t B;
    int kind, fn, pw_k;                                            // kind 0: diff target, 1: rate target
    float c_p, c_n, raw, weight;
    double tau;
    float *ct_grad;                                                // += dL/d raw   (NULL: skip)
    double *tau_grad;                                              // += dL/d tau   (NULL: skip)
    const double *ep;                                              // device-resident c_p, c_n, raw, tau (or NULL)
};

__global__ __launch_bounds__(1024) void event_param_grad_kernel(ParamGradArgs a) {
    __shared__ double red[4][16];
    if (a.ep) { a.c_p = (float)a.ep[REN_EP_CP]; a.c_n = (float)a.ep[REN_EP_CN]; a.raw = (float)a.ep[REN_EP_RAW]; a.tau = a.ep[REN_EP_TAU]; }
    double s_e = 0.0, s_c = 0.0, s_t = 0.0, cnt = 0.0;
    for (int64_t i = threadIdx.x; i < a.B; i += 1024) {
        if (a.valid && !a.valid[i]) continue;
        const float np = (float)a.num_pos[i], nn = (float)a.num_neg[i];
        const float ev = __fsub_rn(__fmul_rn(np, a.c_p), __fmul_rn(nn, a.c_n));
        const double span = (double)a.end_ts[i] - ((double)a.start_ts[i] + a.tau);
        const double rate = (double)ev / span;
        double tgt, t_c, t_t;                                      // target and its derivatives w.r.t. C_p, tau
        if (a.kind == 0) {
            const double u1 = a.u_ts_diff[i], ts_diff = span * u1;
            tgt = ts_diff * rate;
            t_c = ts_diff * (double)np / span;
            t_t = -u1 * rate + ts_diff * (double)ev / (span * span);
        } else {
            tgt = rate;
            t_c = (double)np / span;
            t_t = (double)ev / (span * span);
        }
        const float tf = (float)tgt, p = a.pred[i], d = p - tf;
        const float sg = (float)((d > 0.f) - (d < 0.f));
        float e, e_t;                                              // error and d error / d target
        if (a.fn == 0) { e = fabsf(d); e_t = -sg; }
        else if (a.fn == 1) { e = d * d; e_t = -2.f * d; }
        else {
            const float at = fabsf(tf), den = fmaxf(at, 2.220446049250313e-16f);
            e = fabsf(d) / den;
            e_t = -sg / den - (at > 2.220446049250313e-16f ? fabsf(d) * (float)((tf > 0.f) - (tf < 0.f)) / (den * den) : 0.f);
        }
        s_e += (double)e; s_c += (double)e_t * t_c; s_t += (double)e_t * t_t; cnt += 1.0;
    }
    double v[4] = {s_e, s_c, s_t, cnt};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_xor(v[k], off, 64);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = v[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t[4] = {0.0, 0.0, 0.0, 0.0};
        for (int k = 0; k < 4; ++k)
            for (int w = 0; w < 16; ++w) t[k] += red[k][w];
        const double C = a.ep ? 1.0 / a.ep[REN_EP_INV_C] : ((double)a.c_p + (double)a.c_n) * 0.5;
        const double pw = a.pw_k == 0 ? 1.0 : (a.pw_k == 1 ? 1.0 / C : 1.0 / (C * C));
        const double dpw = a.pw_k == 0 ? 0.0 : (a.pw_k == 1 ? -0.5 / (C * C) : -1.0 / (C * C * C));   // d pw / d C_p
        const double mean_e = t[0] / t[3], mean_c = t[1] / t[3], mean_t = t[2] / t[3];
        const double g_cp = (double)a.weight * (dpw * mean_e + pw * mean_c);
        const double sig = 1.0 / (1.0 + exp(-(double)a.raw));     // d softplus(raw) / d raw
        if (a.ct_grad) a.ct_grad[0] += (float)(g_cp * sig * (double)a.c_n);
        if (a.tau_grad) a.tau_grad[0] += (double)a.weight * pw * mean_t;
    }
}


// ---- epilogue of the tangent (l_grad) render: intensity, its time derivative, validity and d log I / dt of the event's
// colour channel in one launch (robust_e_nerf.py:390-398,865-871): I = c + min_modeled_intensity, I' = c',
// valid = opacity > 0 (no background parameter), d log I / dt = I' / I.
__global__ void rate_epilogue_kernel(const float *__restrict__ colors, const float *__restrict__ colords,
                                     const float *__restrict__ opac, const uint8_t *__restrict__ chan, int C, int64_t n,
                                     float min_i, float *__restrict__ inten, float *__restrict__ intend,
                                     uint8_t *__restrict__ valid, float *__restrict__ dlog) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = chan ? chan[i] : 0;
    const float v = colors[i * C + c] + min_i, vd = colords[i * C + c];
    inten[i] = v;
    intend[i] = vd;
    if (valid) valid[i] = opac[i] > 0.f;
    if (dlog) dlog[i] = vd / v;
}

// ---- d loss / d tau through the poses: sum_i (g_a[i] x_a[i] + g_b[i] x_b[i]) dts[i] in float64, added to out[0]
// (robust_e_nerf.py:340-357: every supervision timestamp moves with tau; g = d loss / d (intensity | its time derivative),
// x = the next time derivative of the intensity).  One workgroup: B is a batch of events.
__global__ __launch_bounds__(1024) void tau_pose_grad_kernel(const float *__restrict__ g_a, const float *__restrict__ x_a,
                                                             const float *__restrict__ g_b, const float *__restrict__ x_b,
                                                             const double *__restrict__ dts, int64_t n, double *__restrict__ out) {
    __shared__ double red[16];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
        double v = (double)g_a[i] * (double)x_a[i];
        if (g_b) v += (double)g_b[i] * (double)x_b[i];
        s += v * dts[i];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < 16; ++w) t += red[w];
        out[0] += t;
    }
}

// event parameters from their raw (trainable) forms, as RobustENeRF evaluates them at the top of every step
// (event_generation_params.py:51-70: C_p = softplus(raw ratio) C_n in float32; :170-185 + modules.py:58-74: the raw
// refractory period clamped to +-logit(1e-4) tau_max, tau = tau_max sigmoid(raw / tau_max) in float64)
__global__ void event_params_refresh_kernel(const float *__restrict__ ct_raw, float c_n, double *__restrict__ tau_raw, double tau_max,
                                            double *__restrict__ ep) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float raw = ct_raw[0];
    const float ratio = raw > 20.f ? raw : log1pf(expf(raw));         // torch softplus (beta 1, threshold 20), float32
    const double c_p = (double)ratio * (double)c_n, mean_c = (c_p + (double)c_n) * 0.5;
    const double lim = 9.21024036697585;                              // |logit(1e-4)|
    double tr = tau_raw[0] / tau_max;
    tr = tr < -lim ? -lim : (tr > lim ? lim : tr);
    tau_raw[0] = tau_max * tr;
    ep[REN_EP_CP] = c_p; ep[REN_EP_CN] = (double)c_n; ep[REN_EP_RAW] = (double)raw;
    ep[REN_EP_TAU] = tau_max * (1.0 / (1.0 + exp(-tr)));
    ep[REN_EP_INV_C] = 1.0 / mean_c; ep[REN_EP_INV_C2] = 1.0 / (mean_c * mean_c);
    ep[REN_EP_TAU_RAW] = tau_raw[0];
}

// torch.optim.Adam on the float64 raw refractory period (its own group, lr = tau_max x relative lr: robust_e_nerf.py:804-807)
// from d loss / d tau: tau = tau_max sigmoid(raw / tau_max)  =>  d tau / d raw = s (1 - s).  state = {exp_avg, exp_avg_sq}.
__global__ void tau_adam_kernel(double *__restrict__ tau_raw, double *__restrict__ tau_grad, double *__restrict__ state, double tau_max,
                                double lr, double beta1, double beta2, double eps, double bc1, double bc2, double grad_scale) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double s = 1.0 / (1.0 + exp(-tau_raw[0] / tau_max));
    const double g = tau_grad[0] * grad_scale * s * (1.0 - s);
    const double m = beta1 * state[0] + (1.0 - beta1) * g, v = beta2 * state[1] + (1.0 - beta2) * g * g;
    state[0] = m; state[1] = v;
    tau_raw[0] -= lr / bc1 * m / (sqrt(v) / sqrt(bc2) + eps);
    tau_grad[0] = 0.0;
}

__global__ void tau_adam_dev_kernel(double *__restrict__ tau_raw, double *__restrict__ tau_grad, double *__restrict__ state,
                                    double tau_max, double lr, double beta1, double beta2, double eps,
                                    const double *__restrict__ hy, double grad_scale) {
    if (threadIdx.x != 0 || blockIdx.x != 0 || hy[REN_HY_SKIP] != 0.0) return;
    const double bc1 = hy[REN_HY_TAU_BC1], bc2 = hy[REN_HY_TAU_BC2];
    const double s = 1.0 / (1.0 + exp(-tau_raw[0] / tau_max));
    const double g = tau_grad[0] * grad_scale * s * (1.0 - s);
    const double m = beta1 * state[0] + (1.0 - beta1) * g, v = beta2 * state[1] + (1.0 - beta2) * g * g;
    state[0] = m; state[1] = v;
    tau_raw[0] -= lr / bc1 * m / (sqrt(v) / sqrt(bc2) + eps);
    tau_grad[0] = 0.0;
}

}  // namespace

extern "C" int ren_event_params_refresh(const float *ct_raw, float c_n, double *tau_raw, double tau_max, double *event_params,
                                        void *stream) {
    if (!ct_raw || !tau_raw || !event_params || !(tau_max > 0.0)) return REN_ERR_BAD_ARG;
    hipLaunchKernelGGL(event_params_refresh_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ct_raw, c_n, tau_raw, tau_max,
                       event_params);
    REN_CHECK_LAUNCH();
}

extern "C" int ren_tau_adam_step(double *tau_raw, double *tau_grad, double *state, double tau_max, double lr, double beta1,
                                 double beta2, double eps, int64_t step, double grad_scale, void *stream) {
    if (!tau_raw || !tau_grad || !state || step < 1 || !(tau_max > 0.0)) return REN_ERR_BAD_ARG;
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    hipLaunchKernelGGL(tau_adam_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, tau_raw, tau_grad, state, tau_max, lr, beta1,
                       beta2, eps, bc1, bc2, grad_scale);
    REN_CHECK_LAUNCH();
}

extern "C" int ren_rate_epilogue(const float *colors, const float *colords, const float *opacities, const uint8_t *channel_idx,
                                 int32_t C, int64_t n, float min_modeled_intensity, float *intensity, float *intensity_dot,
                                 uint8_t *valid, float *dlog_dt, void *stream) {
    if (!colors || !colords || !intensity || !intensity_dot || n < 0 || C < 1 || C > 4 || (C > 1 && !channel_idx) ||
        (valid && !opacities))
        return REN_ERR_BAD_ARG;
    if (n == 0) return REN_OK;
    hipLaunchKernelGGL(rate_epilogue_kernel, dim3(ren_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, colors, colords,
                       opacities, C > 1 ? channel_idx : nullptr, (int)C, n, min_modeled_intensity, intensity, intensity_dot,
                       valid, dlog_dt);
    REN_CHECK_LAUNCH();
}

extern "C" int ren_tau_pose_grad(const float *g_a, const float *x_a, const float *g_b, const float *x_b, const double *dts,
                                 int64_t n, double *tau_grad, void *stream) {
    if (!g_a || !x_a || !dts || !tau_grad || n < 0 || ((g_b == nullptr) != (x_b == nullptr))) return REN_ERR_BAD_ARG;
    if (n == 0) return REN_OK;
    hipLaunchKernelGGL(tau_pose_grad_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, g_a, x_a, g_b, x_b, dts, n, tau_grad);
    REN_CHECK_LAUNCH();
}

extern "C" int ren_event_prepare(const int64_t *start_ts, const int64_t *end_ts, const int64_t *num_pos,
                                 const int64_t *num_neg, const double *u_ts_diff, const double *u_diff_start,
                                 const double *u_grad, int64_t B, float c_p, float c_n, double tau, double *ts_start,
                                 double *ts_end, float *target_diff, double *ts_grad, float *target_grad,
                                 double *dts_start, double *dts_end, double *dts_grad, const double *event_params,
                                 void *stream) {
    if (!start_ts || !end_ts || !num_pos || !num_neg || !u_ts_diff || !u_diff_start || !ts_start || !ts_end ||
        !target_diff || B < 0)
        return REN_ERR_BAD_ARG;
    if ((ts_grad && !u_grad) || (dts_grad && !ts_grad) || ((dts_start == nullptr) != (dts_end == nullptr)))
        return REN_ERR_BAD_ARG;
    if (B == 0) return REN_OK;
    PrepArgs a{start_ts, end_ts, num_pos, num_neg, u_ts_diff, u_diff_start, u_grad, B, c_p, c_n, tau,
               ts_start, ts_end, ts_grad, dts_start, dts_end, dts_grad, target_diff, target_grad, event_params};
    hipLaunchKernelGGL(event_prepare_kernel, dim3(ren_blocks(B, 256)), dim3(256), 0, (hipStream_t)stream, a);
    REN_CHECK_LAUNCH();
}

extern "C" int ren_event_param_grad(int32_t kind, int32_t err_fn, int32_t param_weight_power, const float *pred,
                                    const uint8_t *valid, const int64_t *start_ts, const int64_t *end_ts,
                                    const int64_t *num_pos, const int64_t *num_neg, const double *u_ts_diff, int64_t B,
                                    float c_p, float c_n, float raw_ratio, double tau, float weight, float *ct_grad,
                                    double *tau_grad, const double *event_params, void *stream) {
    if (kind < 0 || kind > 1 || err_fn < 0 || err_fn > 2 || param_weight_power < 0 || param_weight_power > 2)
        return REN_ERR_BAD_ARG;
    if (!pred || !start_ts || !end_ts || !num_pos || !num_neg || (kind == 0 && !u_ts_diff) || B < 0) return REN_ERR_BAD_ARG;
    if (B == 0 || (!ct_grad && !tau_grad)) return REN_OK;
    ParamGradArgs a{pred, valid, start_ts, end_ts, num_pos, num_neg, u_ts_diff, B, kind, err_fn, param_weight_power,
                    c_p, c_n, raw_ratio, weight, tau, ct_grad, tau_grad, event_params};
    hipLaunchKernelGGL(event_param_grad_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a);
    REN_CHECK_LAUNCH();
}

extern "C" int ren_event_loss_fwd(const float *i_start, const float *i_end, const float *target,
                                  const uint8_t *valid, int64_t B, int32_t err_fn, float *loss_sum,
                                  void *stream) {
    if (!i_start || !i_end || !target || !loss_sum || B < 0 || err_fn < 0 || err_fn > 2) return REN_ERR_BAD_ARG;
    hipError_t e = hipMemsetAsync(loss_sum, 0, 2 * sizeof(float), (hipStream_t)stream);
    if (e != hipSuccess) return REN_ERR_LAUNCH;
    if (B == 0) return REN_OK;
    int blocks = ren_blocks(B, 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(event_loss_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, i_start, i_end,
                       target, valid, B, err_fn, loss_sum);
    REN_CHECK_LAUNCH();
}

extern "C" int ren_event_loss_bwd(const float *i_start, const float *i_end, const float *target,
                                  const uint8_t *valid, int64_t B, int32_t err_fn, float scale,
                                  const float *loss_sum, float *g_start, float *g_end, void *stream) {
    if (!i_start || !i_end || !target || !loss_sum || !g_start || !g_end || B < 0 || err_fn < 0 || err_fn > 2)
        return REN_ERR_BAD_ARG;
    if (B == 0) return REN_OK;
    hipLaunchKernelGGL(event_loss_bwd_kernel, dim3(ren_blocks(B, 256)), dim3(256), 0, (hipStream_t)stream,
                       i_start, i_end, target, valid, B, err_fn, scale, loss_sum, g_start, g_end);
    REN_CHECK_LAUNCH();
}

extern "C" int ren_event_diff_loss_fwd(const float *colors, const float *opacities, const uint8_t *channel_idx, int32_t C,
                                       float min_intensity, const float *target, int32_t use_validity, int64_t B,
                                       int32_t err_fn, float *loss_sum, void *stream) {
    if (!colors || !target || !loss_sum || B < 0 || err_fn < 0 || err_fn > 2 || (C != 1 && C != 3)) return REN_ERR_BAD_ARG;
    if (use_validity && !opacities) return REN_ERR_BAD_ARG;
    if (hipMemsetAsync(loss_sum, 0, 2 * sizeof(float), (hipStream_t)stream) != hipSuccess) return REN_ERR_LAUNCH;
    if (B == 0) return REN_OK;
    int blocks = ren_blocks(B, 256);
    if (blocks > 1024) blocks = 1024;
    const DiffLossArgs a{colors, opacities, target, channel_idx, B, C, err_fn, use_validity, min_intensity};
    hipLaunchKernelGGL(diff_loss_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, loss_sum);
    REN_CHECK_LAUNCH();
}

extern "C" int ren_event_diff_loss_bwd(const float *colors, const float *opacities, const uint8_t *channel_idx, int32_t C,
                                       float min_intensity, const float *target, int32_t use_validity, int64_t B,
                                       int32_t err_fn, float scale, const double *scale_dev, const float *loss_sum,
                                       float *g_colors, float *intensity, float *pred, uint8_t *valid, float *loss,
                                       void *stream) {
    if (!colors || !target || !loss_sum || !g_colors || B < 0 || err_fn < 0 || err_fn > 2 || (C != 1 && C != 3))
        return REN_ERR_BAD_ARG;
    if (use_validity && !opacities) return REN_ERR_BAD_ARG;
    const DiffLossArgs a{colors, opacities, target, channel_idx, B, C, err_fn, use_validity, min_intensity};
    hipLaunchKernelGGL(diff_loss_bwd_kernel, dim3(ren_blocks(B > 0 ? B : 1, 256)), dim3(256), 0, (hipStream_t)stream, a, scale,
                       scale_dev, loss_sum, g_colors, intensity, pred, valid, loss);
    REN_CHECK_LAUNCH();
}

extern "C" int ren_adam_step(float *param, float *grad, float *exp_avg, float *exp_avg_sq, int64_t n,
                             float lr, float beta1, float beta2, float eps, float weight_decay,
                             int64_t step, float grad_scale, int32_t zero_grad, void *stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || n < 0 || step < 1) return REN_ERR_BAD_ARG;
    if (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15)
        return REN_ERR_BAD_ARG;                      // float4 path needs 16-byte alignment
    if (n == 0) return REN_OK;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    int64_t blocks = (n / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(adam_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                       exp_avg_sq, n, (float)(lr / bc1), beta1, beta2, eps, weight_decay,
                       (float)(1.0 / sqrt(bc2)), grad_scale, zero_grad);
    REN_CHECK_LAUNCH();
}

extern "C" int ren_step_tick(double *hyper, double beta1, double beta2, const int64_t *stats_a, const int64_t *stats_b,
                             int32_t tick_tau, void *stream) {
    if (!hyper || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0)) return REN_ERR_BAD_ARG;
    hipLaunchKernelGGL(step_tick_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, hyper, beta1, beta2, stats_a, stats_b, tick_tau);
    REN_CHECK_LAUNCH();
}

extern "C" int ren_adam_step_dev(float *param, float *grad, float *exp_avg, float *exp_avg_sq, int64_t n,
                                 float lr, float beta1, float beta2, float eps, float weight_decay,
                                 const double *hyper, float grad_scale, int32_t zero_grad, void *stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || !hyper || n < 0) return REN_ERR_BAD_ARG;
    if (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15)
        return REN_ERR_BAD_ARG;
    if (n == 0) return REN_OK;
    int64_t blocks = (n / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(adam_dev_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                       exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, hyper, grad_scale, zero_grad);
    REN_CHECK_LAUNCH();
}

extern "C" int ren_tau_adam_step_dev(double *tau_raw, double *tau_grad, double *state, double tau_max, double lr, double beta1,
                                     double beta2, double eps, const double *hyper, double grad_scale, void *stream) {
    if (!tau_raw || !tau_grad || !state || !hyper || !(tau_max > 0.0)) return REN_ERR_BAD_ARG;
    hipLaunchKernelGGL(tau_adam_dev_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, tau_raw, tau_grad, state, tau_max, lr,
                       beta1, beta2, eps, hyper, grad_scale);
    REN_CHECK_LAUNCH();
}

extern "C" int ren_occgrid_cell_points(const int64_t *indices, const float *jitter, int64_t m,
                                       const float *roi, const int32_t *res, int32_t contraction_type,
                                       float *x_world, uint8_t *valid, void *stream) {
    if (!indices || !jitter || !roi || !res || !x_world || m < 0) return REN_ERR_BAD_ARG;
    if (contraction_type < 0 || contraction_type > 2) return REN_ERR_BAD_ARG;
    if (m == 0) return REN_OK;
    CellArgs a;
    for (int k = 0; k < 6; ++k) a.roi[k] = roi[k];
    for (int k = 0; k < 3; ++k) a.res[k] = res[k];
    a.type = contraction_type;
    hipLaunchKernelGGL(occgrid_cell_points_kernel, dim3(ren_blocks(m, 256)), dim3(256), 0, (hipStream_t)stream,
                       indices, jitter, m, a, x_world, valid);
    REN_CHECK_LAUNCH();
}

extern "C" int ren_occgrid_ema(float *occs, const int64_t *indices, const uint8_t *valid, const float *sigma,
                               const float *step_sizes, float step_size, int64_t m, float ema_decay,
                               void *stream) {
    if (!occs || !indices || !sigma || m < 0) return REN_ERR_BAD_ARG;
    if (m == 0) return REN_OK;
    hipLaunchKernelGGL(occgrid_ema_kernel, dim3(ren_blocks(m, 256)), dim3(256), 0, (hipStream_t)stream, occs,
                       indices, valid, sigma, step_sizes, step_size, m, ema_decay);
    REN_CHECK_LAUNCH();
}

extern "C" int ren_occgrid_ema_unique(float *occs, const int64_t *indices, const uint8_t *valid, const float *sigma,
                                      const float *step_sizes, float step_size, int64_t m, float ema_decay,
                                      float *scratch_cells, void *stream) {
    if (!occs || !indices || !sigma || !scratch_cells || m < 0) return REN_ERR_BAD_ARG;
    if (m == 0) return REN_OK;
    const dim3 grd(ren_blocks(m, 256)), blk(256);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(occgrid_ema_clear_kernel, grd, blk, 0, st, indices, m, scratch_cells);
    hipLaunchKernelGGL(occgrid_ema_max_kernel, grd, blk, 0, st, indices, valid, sigma, step_sizes, step_size, m, scratch_cells);
    hipLaunchKernelGGL(occgrid_ema_apply_kernel, grd, blk, 0, st, occs, indices, valid, m, ema_decay, scratch_cells);
    REN_CHECK_LAUNCH();
}

extern "C" int ren_occgrid_binarize(const float *occs, int64_t cells, float occ_thre, uint8_t *binary,
                                    float *scratch, void *stream) {
    if (!occs || !binary || !scratch || cells <= 0) return REN_ERR_BAD_ARG;
    hipError_t e = hipMemsetAsync(scratch, 0, sizeof(float), (hipStream_t)stream);
    if (e != hipSuccess) return REN_ERR_LAUNCH;
    int blocks = ren_blocks(cells, 256 * 16);
    hipLaunchKernelGGL(sum_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, occs, cells, scratch);
    hipLaunchKernelGGL(binarize_kernel, dim3(ren_blocks(cells, 256)), dim3(256), 0, (hipStream_t)stream, occs,
                       cells, occ_thre, scratch, binary);
    REN_CHECK_LAUNCH();
}

// ---- torch.nn.utils.weight_norm (dim 0) over a packed parameter block (external/ngp.py:207-228, external/mlp.py:303-319) ----
// W[r, :] = g[r] v[r, :] / ||v[r, :]||.  One wave per weight row; everything that is not a listed weight (biases, layers
// without the reparametrisation) is copied through.
namespace {

constexpr int WN_MAX_LAYERS = 16;
struct WnArgs {
    int n_layers, n_rows;
    int w_off[WN_MAX_LAYERS], rows[WN_MAX_LAYERS], cols[WN_MAX_LAYERS], g_off[WN_MAX_LAYERS], row0[WN_MAX_LAYERS];
};

__device__ __forceinline__ bool wn_row(const WnArgs &a, int row, int &w_off, int &cols, int &g_idx) {
    if (row >= a.n_rows) return false;
    int l = 0;
    while (l + 1 < a.n_layers && row >= a.row0[l + 1]) ++l;
    const int r = row - a.row0[l];
    cols = a.cols[l];
    w_off = a.w_off[l] + r * cols;
    g_idx = a.g_off[l] + r;
    return true;
}

__global__ __launch_bounds__(256) void weight_norm_fwd_kernel(WnArgs a, const float *__restrict__ raw, const float *__restrict__ g,
                                                              float *__restrict__ eff) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    int w_off, cols, g_idx;
    if (!wn_row(a, row, w_off, cols, g_idx)) return;
    float ss = 0.f;
    for (int c = lane; c < cols; c += 64) { const float v = raw[w_off + c]; ss += v * v; }
    const float s = g[g_idx] / sqrtf(ren_wave_sum(ss));
    for (int c = lane; c < cols; c += 64) eff[w_off + c] = raw[w_off + c] * s;
}

// d g[r] = dW[r, :] . v / ||v||;  d v[r, :] = (g / ||v||) (dW[r, :] - d g[r] v / ||v||)
__global__ __launch_bounds__(256) void weight_norm_bwd_kernel(WnArgs a, const float *__restrict__ raw, const float *__restrict__ g,
                                                              const float *__restrict__ d_eff, float *__restrict__ d_raw,
                                                              float *__restrict__ d_g) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    int w_off, cols, g_idx;
    if (!wn_row(a, row, w_off, cols, g_idx)) return;
    float ss = 0.f, dot = 0.f;
    for (int c = lane; c < cols; c += 64) {
        const float v = raw[w_off + c];
        ss += v * v;
        dot += v * d_eff[w_off + c];
    }
    const float inv = 1.f / sqrtf(ren_wave_sum(ss));
    const float dg = ren_wave_sum(dot) * inv, s = g[g_idx] * inv;
    for (int c = lane; c < cols; c += 64) d_raw[w_off + c] = s * (d_eff[w_off + c] - dg * inv * raw[w_off + c]);
    if (lane == 0) d_g[g_idx] = dg;
}

int wn_args(const int32_t *layers, int32_t n_layers, int64_t n_params, WnArgs &a) {
    if (!layers || n_layers < 0 || n_layers > WN_MAX_LAYERS || n_params < 0) return REN_ERR_BAD_ARG;
    a.n_layers = n_layers;
    a.n_rows = 0;
    for (int l = 0; l < n_layers; ++l) {
        a.w_off[l] = layers[4 * l]; a.rows[l] = layers[4 * l + 1]; a.cols[l] = layers[4 * l + 2]; a.g_off[l] = layers[4 * l + 3];
        if (a.w_off[l] < 0 || a.rows[l] < 1 || a.cols[l] < 1 || a.g_off[l] < 0 ||
            (int64_t)a.w_off[l] + (int64_t)a.rows[l] * a.cols[l] > n_params)
            return REN_ERR_BAD_ARG;
        a.row0[l] = a.n_rows;
        a.n_rows += a.rows[l];
    }
    return REN_OK;
}

}  // namespace

extern "C" int ren_weight_norm_fwd(const float *raw, const float *g, const int32_t *layers, int32_t n_layers,
                                   int64_t n_params, float *eff, void *stream) {
    WnArgs a;
    const int rc = wn_args(layers, n_layers, n_params, a);
    if (rc) return rc;
    if (!raw || !eff || (n_layers && !g)) return REN_ERR_BAD_ARG;
    if (n_params == 0) return REN_OK;
    if (hipMemcpyAsync(eff, raw, n_params * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess)
        return REN_ERR_LAUNCH;
    if (a.n_rows)
        hipLaunchKernelGGL(weight_norm_fwd_kernel, dim3((a.n_rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, a, raw, g, eff);
    REN_CHECK_LAUNCH();
}

extern "C" int ren_weight_norm_bwd(const float *raw, const float *g, float *d_eff, const int32_t *layers,
                                   int32_t n_layers, int64_t n_params, float *d_raw, float *d_g, int32_t zero_d_eff,
                                   void *stream) {
    WnArgs a;
    const int rc = wn_args(layers, n_layers, n_params, a);
    if (rc) return rc;
    if (!raw || !d_eff || !d_raw || (n_layers && (!g || !d_g))) return REN_ERR_BAD_ARG;
    if (n_params == 0) return REN_OK;
    if (hipMemcpyAsync(d_raw, d_eff, n_params * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess)
        return REN_ERR_LAUNCH;
    if (a.n_rows)
        hipLaunchKernelGGL(weight_norm_bwd_kernel, dim3((a.n_rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, a, raw, g, d_eff,
                           d_raw, d_g);
    if (zero_d_eff && hipMemsetAsync(d_eff, 0, n_params * sizeof(float), (hipStream_t)stream) != hipSuccess) return REN_ERR_LAUNCH;
    REN_CHECK_LAUNCH();
}
