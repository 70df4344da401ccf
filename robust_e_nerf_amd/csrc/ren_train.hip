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

}  // namespace

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
