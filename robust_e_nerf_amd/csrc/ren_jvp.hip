// Forward-mode (tangent) companions of the pose, hash-grid, compositing and loss kernels for the
// log-intensity-GRADIENT loss.  The reference obtains d(log I)/d(timestamp) per ray by reverse-mode
// autograd with create_graph=True and then differentiates THROUGH that graph
// (robust_e_nerf/models/robust_e_nerf.py:383-409, utils/autograd.py:4-34): ~3-4x the cost of a render
// and the reason it uses torch MLPs, a torch SH encoder and softplus instead of fused kernels
// (external/ngp.py:5-19).  The timestamp is ONE scalar per ray, so here a tangent d/dt is carried
// through the forward kernels (value + derivative in one pass: SURVEY 7.2 H1) and ordinary
// reverse-mode is then taken over the (value, tangent) pair (ren_*_bwd_jvp).  Sample placement is not
// differentiated, as in the reference (external/vol_rendering.py:36-37).
#include "ren_hashgrid_common.h"

namespace {

// ------------------------------------------------------------------------------------------------ pose
struct Quat { float x, y, z, w; };
__device__ __forceinline__ Quat qmul(const Quat &p, const Quat &q) {
    Quat r;
    r.x = p.w * q.x + q.w * p.x + (p.y * q.z - p.z * q.y);
    r.y = p.w * q.y + q.w * p.y + (p.z * q.x - p.x * q.z);
    r.z = p.w * q.z + q.w * p.z + (p.x * q.y - p.y * q.x);
    r.w = p.w * q.w - (p.x * q.x + p.y * q.y + p.z * q.z);
    return r;
}
__device__ __forceinline__ float lerpf(float a, float b, float w) {
    return fabsf(w) < 0.5f ? a + w * (b - a) : b - (b - a) * (1.f - w);
}

// value: identical to trajectory_kernel (ren_pose.hip); tangent: d pos/dt = (p_r - p_l)/bin,
// d R/dt = R [rv]_x / bin with rv the full rotation vector q_l -> q_r (q(w) = q_l (x) exp(w rv / 2)).
__global__ void trajectory_jvp_kernel(const double *__restrict__ ts, int64_t B, const int64_t *__restrict__ tab_ts,
                                      const float *__restrict__ tab_pos, const float *__restrict__ tab_quat,
                                      int64_t C, float *__restrict__ pos, float *__restrict__ rot,
                                      float *__restrict__ dpos, float *__restrict__ drot) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    const double t = ts[i];
    int64_t lo = 0, hi = C;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if ((double)tab_ts[mid] < t) lo = mid + 1; else hi = mid;
    }
    int64_t right = lo < C ? lo : C - 1;
    int64_t left = (t == (double)tab_ts[0]) ? right : right - 1;
    if (left < 0) left = 0;
    int64_t wbin = left < C - 1 ? left : C - 2;
    const double bin = (double)(tab_ts[wbin + 1] - tab_ts[wbin]);
    const float w = (float)((t - (double)tab_ts[left]) / bin);
    const float wdot = (float)(1.0 / bin);
    for (int k = 0; k < 3; ++k) {
        const float a = tab_pos[3 * left + k], b = tab_pos[3 * right + k];
        pos[3 * i + k] = lerpf(a, b, w);
        dpos[3 * i + k] = (b - a) * wdot;
    }
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
    const float vx = scale * rel.x, vy = scale * rel.y, vz = scale * rel.z;     // rotation vector rv
    float rx = w * vx, ry = w * vy, rz = w * vz;
    float th = sqrtf(rx * rx + ry * ry + rz * rz);
    float t2 = th * th;
    float s = th <= 1e-3f ? 0.5f - t2 / 48.f + t2 * t2 / 3840.f : sinf(th * 0.5f) / th;
    Quat rq = {s * rx, s * ry, s * rz, cosf(th * 0.5f)};
    Quat q = qmul(q0, rq);
    float x2 = q.x * q.x, y2 = q.y * q.y, z2 = q.z * q.z, w2 = q.w * q.w;
    float xy = q.x * q.y, zw = q.z * q.w, xz = q.x * q.z, yw = q.y * q.w, yz = q.y * q.z, xw = q.x * q.w;
    float R[9];
    R[0] = x2 - y2 - z2 + w2; R[1] = 2.f * (xy - zw);     R[2] = 2.f * (xz + yw);
    R[3] = 2.f * (xy + zw);   R[4] = -x2 + y2 - z2 + w2;  R[5] = 2.f * (yz - xw);
    R[6] = 2.f * (xz - yw);   R[7] = 2.f * (yz + xw);     R[8] = -x2 - y2 + z2 + w2;
    // dR/dt = R * skew(rv) * wdot ; skew(v) = [[0,-vz,vy],[vz,0,-vx],[-vy,vx,0]]
    for (int r = 0; r < 3; ++r) {
        const float a = R[3 * r], b = R[3 * r + 1], c = R[3 * r + 2];
        rot[9 * i + 3 * r] = a; rot[9 * i + 3 * r + 1] = b; rot[9 * i + 3 * r + 2] = c;
        drot[9 * i + 3 * r] = (b * vz - c * vy) * wdot;
        drot[9 * i + 3 * r + 1] = (c * vx - a * vz) * wdot;
        drot[9 * i + 3 * r + 2] = (a * vy - b * vx) * wdot;
    }
}

__global__ void raygen_jvp_kernel(const float *__restrict__ Kinv, const float *__restrict__ px,
                                  const float *__restrict__ pos, const float *__restrict__ rot,
                                  const float *__restrict__ dpos, const float *__restrict__ drot, int64_t B,
                                  float *__restrict__ rays_o, float *__restrict__ rays_d,
                                  float *__restrict__ rays_do, float *__restrict__ rays_dd) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    const float u = px[2 * i], v = px[2 * i + 1];
    const float k0 = Kinv[0] * u + Kinv[1] * v + Kinv[2];
    const float k1 = Kinv[3] * u + Kinv[4] * v + Kinv[5];
    const float k2 = Kinv[6] * u + Kinv[7] * v + Kinv[8];
    const float *R = rot + 9 * i, *dR = drot + 9 * i;
    float m[3], md[3];
    for (int r = 0; r < 3; ++r) {
        m[r] = R[3 * r] * k0 + R[3 * r + 1] * k1 + R[3 * r + 2] * k2;
        md[r] = dR[3 * r] * k0 + dR[3 * r + 1] * k1 + dR[3 * r + 2] * k2;
    }
    const float inv = 1.f / sqrtf(m[0] * m[0] + m[1] * m[1] + m[2] * m[2]);
    const float d0 = m[0] * inv, d1 = m[1] * inv, d2 = m[2] * inv;
    const float dm = d0 * md[0] + d1 * md[1] + d2 * md[2];
    rays_d[3 * i] = d0; rays_d[3 * i + 1] = d1; rays_d[3 * i + 2] = d2;
    rays_dd[3 * i] = (md[0] - d0 * dm) * inv;
    rays_dd[3 * i + 1] = (md[1] - d1 * dm) * inv;
    rays_dd[3 * i + 2] = (md[2] - d2 * dm) * inv;
    for (int k = 0; k < 3; ++k) { rays_o[3 * i + k] = pos[3 * i + k]; rays_do[3 * i + k] = dpos[3 * i + k]; }
}

// ------------------------------------------------------------------------------------------------ hash grid
__global__ __launch_bounds__(256) void hashgrid_fwd_jvp_kernel(
    GridDev g, const float2 *__restrict__ table, ren_scene_dev sc, const float *__restrict__ rays_o,
    const float *__restrict__ rays_d, const float *__restrict__ rays_do, const float *__restrict__ rays_dd,
    const int32_t *__restrict__ ray_indices, const float *__restrict__ t_starts, const float *__restrict__ t_ends,
    int64_t n, int64_t n_pad, float *__restrict__ feat, float *__restrict__ featd, const int64_t *__restrict__ n_dev) {
    const int lvl = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n_dev) {
        n = ren_eff_n(n, n_dev);
        const int64_t p = ((n + 31) >> 5) << 5;
        n_pad = p < n_pad ? p : n_pad;
    }
    if (i >= n_pad) return;
    float f0 = 0.f, f1 = 0.f, g0 = 0.f, g1 = 0.f;
    if (i < n) {
        float x[3], xd[3], u[3], ud[3];
        sample_pos_jvp(rays_o, rays_d, rays_do, rays_dd, ray_indices, t_starts, t_ends, i, x, xd);
        contract_jvp(sc, x, xd, u, ud);
        const float scale = g.scale[lvl];
        const LevelPos p = level_pos(u[0], u[1], u[2], scale);
        const float wd[3] = {scale * ud[0], scale * ud[1], scale * ud[2]};
        const uint32_t res = g.res[lvl], size = g.size[lvl];
        const bool hashed = g.hashed[lvl] != 0;
        const float2 *tab = table + g.offset[lvl];
        float2 v[8];
        uint32_t idx[8];
        corner_indices8(p.c[0], p.c[1], p.c[2], res, size, hashed, idx);
        gather_corners8(tab, idx, v);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float ax = (c & 1) ? p.w[0] : 1.f - p.w[0], bx = (c & 1) ? wd[0] : -wd[0];
            const float ay = (c & 2) ? p.w[1] : 1.f - p.w[1], by = (c & 2) ? wd[1] : -wd[1];
            const float az = (c & 4) ? p.w[2] : 1.f - p.w[2], bz = (c & 4) ? wd[2] : -wd[2];
            const float w = ax * ay * az;
            const float wdc = bx * ay * az + ax * by * az + ax * ay * bz;
            f0 += w * v[c].x; f1 += w * v[c].y;
            g0 += wdc * v[c].x; g1 += wdc * v[c].y;
        }
    }
    const int64_t b = ((i >> 5) * REN_MAX_LEVELS + lvl) * 64 + (i & 31);
    __builtin_nontemporal_store(f0, feat + b); __builtin_nontemporal_store(f1, feat + b + 32);      // streamed once (see hashgrid_fwd_kernel)
    __builtin_nontemporal_store(g0, featd + b); __builtin_nontemporal_store(g1, featd + b + 32);
}

// d table += w_c * dfeat + wdot_c * dfeatd  (per-update atomics, lane pair per feature)
__global__ __launch_bounds__(256) void hashgrid_bwd_jvp_kernel(
    GridDev g, float *__restrict__ grad_table, ren_scene_dev sc, const float *__restrict__ rays_o,
    const float *__restrict__ rays_d, const float *__restrict__ rays_do, const float *__restrict__ rays_dd,
    const int32_t *__restrict__ ray_indices, const float *__restrict__ t_starts, const float *__restrict__ t_ends,
    int64_t n, const float *__restrict__ dfeat, const float *__restrict__ dfeatd) {
    const int lvl = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * 128 + (threadIdx.x >> 1);
    const int fsel = threadIdx.x & 1;
    if (i >= n) return;
    const int64_t b = ((i >> 5) * REN_MAX_LEVELS + lvl) * 64 + (i & 31) + 32 * fsel;
    const float d = dfeat[b], dd = dfeatd[b];
    if (d == 0.f && dd == 0.f) return;
    float x[3], xd[3], u[3], ud[3];
    sample_pos_jvp(rays_o, rays_d, rays_do, rays_dd, ray_indices, t_starts, t_ends, i, x, xd);
    contract_jvp(sc, x, xd, u, ud);
    const float scale = g.scale[lvl];
    const LevelPos p = level_pos(u[0], u[1], u[2], scale);
    const float wd[3] = {scale * ud[0], scale * ud[1], scale * ud[2]};
    const uint32_t res = g.res[lvl], size = g.size[lvl];
    const bool hashed = g.hashed[lvl] != 0;
    float *gt = grad_table + 2 * (size_t)g.offset[lvl];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const uint32_t idx = corner_index(p.c[0] + (c & 1), p.c[1] + ((c >> 1) & 1), p.c[2] + (c >> 2), res, size, hashed);
        const float ax = (c & 1) ? p.w[0] : 1.f - p.w[0], bx = (c & 1) ? wd[0] : -wd[0];
        const float ay = (c & 2) ? p.w[1] : 1.f - p.w[1], by = (c & 2) ? wd[1] : -wd[1];
        const float az = (c & 4) ? p.w[2] : 1.f - p.w[2], bz = (c & 4) ? wd[2] : -wd[2];
        const float w = ax * ay * az;
        const float wdc = bx * ay * az + ax * by * az + ax * ay * bz;
        atomicAdd(gt + 2 * (size_t)idx + fsel, w * d + wdc * dd);
    }
}

// ------------------------------------------------------------------------------------------------ compositing
__device__ __forceinline__ float wave_incl_scan(float v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        float t = __shfl_up(v, off, 64);
        if (lane >= off) v += t;
    }
    return v;
}
__device__ __forceinline__ float wave_incl_suffix_scan(float v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        float t = __shfl_down(v, off, 64);
        if (lane + off < 64) v += t;
    }
    return v;
}

// a = sigma dt, E = excl. prefix(a), T = exp(-E), w = T (1 - exp(-a));  tangents: ad = sigmad dt,
// Ed = excl. prefix(ad), wd = -w Ed + T_{next} ad.   color = sum w c + bk (1 - sum w),
// colord = sum (wd c + w cd) - bk sum wd.
template <int C>
__global__ __launch_bounds__(256) void composite_fwd_jvp_kernel(
    const int64_t *__restrict__ offsets, const int32_t *__restrict__ counts, int64_t n_rays,
    const float *__restrict__ t_starts, const float *__restrict__ t_ends, const float *__restrict__ sigmas,
    const float *__restrict__ sigmads, const float *__restrict__ rgbs, const float *__restrict__ rgbds,
    const float *__restrict__ bkgd, float *__restrict__ colors, float *__restrict__ colords,
    float *__restrict__ opacities, float *__restrict__ opacds, float *__restrict__ weights,
    float *__restrict__ trans, float *__restrict__ eds) {
    const int lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (ray >= n_rays) return;
    const int64_t base = offsets[ray];
    const int cnt = counts[ray];
    float carry = 0.f, carryd = 0.f, acc_o = 0.f, acc_od = 0.f, acc_c[C], acc_cd[C];
#pragma unroll
    for (int c = 0; c < C; ++c) { acc_c[c] = 0.f; acc_cd[c] = 0.f; }
    for (int s = 0; s < cnt; s += 64) {
        const int j = s + lane;
        const bool act = j < cnt;
        float dt = 0.f, sg = 0.f, sgd = 0.f;
        if (act) { dt = t_ends[base + j] - t_starts[base + j]; sg = sigmas[base + j]; sgd = sigmads[base + j]; }
        const float a = sg * dt, ad = sgd * dt;
        const float inc = wave_incl_scan(a, lane), incd = wave_incl_scan(ad, lane);
        const float E = carry + (inc - a), Ed = carryd + (incd - ad);
        const float T = expf(-E), ea = expf(-a);
        const float w = T * (1.f - ea);
        const float wd = -w * Ed + T * ea * ad;
        if (act) {
            weights[base + j] = w; trans[base + j] = T; eds[base + j] = Ed;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float col = rgbs[(base + j) * C + c], cold = rgbds[(base + j) * C + c];
                acc_c[c] += w * col;
                acc_cd[c] += wd * col + w * cold;
            }
            acc_o += w; acc_od += wd;
        }
        carry += __shfl(inc, 63, 64);
        carryd += __shfl(incd, 63, 64);
    }
    acc_o = ren_wave_sum(acc_o); acc_od = ren_wave_sum(acc_od);
#pragma unroll
    for (int c = 0; c < C; ++c) { acc_c[c] = ren_wave_sum(acc_c[c]); acc_cd[c] = ren_wave_sum(acc_cd[c]); }
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            colors[ray * C + c] = bkgd ? acc_c[c] + bkgd[c] * (1.f - acc_o) : acc_c[c];
            colords[ray * C + c] = bkgd ? acc_cd[c] - bkgd[c] * acc_od : acc_cd[c];
        }
        opacities[ray] = acc_o;
        opacds[ray] = acc_od;
    }
}

// L = L(color, colord).  With v = gC.(c - bk), u = gCd.(c - bk), y = gCd.cd, A = v + y, B = Ed u:
//   d sigma_i  = dt_i [ T_{i+1} (A_i - B_i) - sum_{j>i} w_j (A_j - B_j) - sum_{j>=i} T_{j+1} ad_j u_j ]
//   d sigmad_i = dt_i [ T_{i+1} u_i - sum_{j>i} w_j u_j ]
//   d c_i = gC w_i + gCd wd_i ,  d cd_i = gCd w_i
template <int C>
__global__ __launch_bounds__(256) void composite_bwd_jvp_kernel(
    const int64_t *__restrict__ offsets, const int32_t *__restrict__ counts, int64_t n_rays,
    const float *__restrict__ t_starts, const float *__restrict__ t_ends, const float *__restrict__ sigmas,
    const float *__restrict__ sigmads, const float *__restrict__ rgbs, const float *__restrict__ rgbds,
    const float *__restrict__ bkgd, const float *__restrict__ weights, const float *__restrict__ trans,
    const float *__restrict__ eds, const float *__restrict__ opacities, const float *__restrict__ opacds,
    const float *__restrict__ g_colors, const float *__restrict__ g_colords, float *__restrict__ d_sigmas,
    float *__restrict__ d_sigmads, float *__restrict__ d_rgbs, float *__restrict__ d_rgbds,
    float *__restrict__ d_bkgd_per_ray) {
    const int lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (ray >= n_rays) return;
    const int64_t base = offsets[ray];
    const int cnt = counts[ray];
    float gc[C], gcd[C], bk[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        gc[c] = g_colors ? g_colors[ray * C + c] : 0.f;
        gcd[c] = g_colords[ray * C + c];
        bk[c] = bkgd ? bkgd[c] : 0.f;
    }
    if (d_bkgd_per_ray && lane == 0) {
#pragma unroll
        for (int c = 0; c < C; ++c)
            d_bkgd_per_ray[ray * C + c] = gc[c] * (1.f - opacities[ray]) - gcd[c] * opacds[ray];
    }
    float cAB = 0.f, cU = 0.f, cZ = 0.f;                    // carries of the suffix sums over later chunks
    const int n_chunks = (cnt + 63) >> 6;
    for (int ch = n_chunks - 1; ch >= 0; --ch) {
        const int j = ch * 64 + lane;
        const bool act = j < cnt;
        float dt = 0.f, sg = 0.f, sgd = 0.f, w = 0.f, T = 0.f, Ed = 0.f, A = 0.f, u = 0.f;
        float Tn = 0.f, ad = 0.f, wd = 0.f;
        if (act) {
            dt = t_ends[base + j] - t_starts[base + j];
            sg = sigmas[base + j]; sgd = sigmads[base + j];
            w = weights[base + j]; T = trans[base + j]; Ed = eds[base + j];
            Tn = T * expf(-sg * dt);
            ad = sgd * dt;
            wd = -w * Ed + Tn * ad;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float col = rgbs[(base + j) * C + c] - bk[c], cold = rgbds[(base + j) * C + c];
                A += gc[c] * col + gcd[c] * cold;
                u += gcd[c] * col;
                d_rgbs[(base + j) * C + c] = gc[c] * w + gcd[c] * wd;
                d_rgbds[(base + j) * C + c] = gcd[c] * w;
            }
        }
        const float AB = A - Ed * u;
        const float wAB = w * AB, wU = w * u, Z = Tn * ad * u;
        const float sAB = wave_incl_suffix_scan(wAB, lane), sU = wave_incl_suffix_scan(wU, lane),
                    sZ = wave_incl_suffix_scan(Z, lane);
        if (act) {
            d_sigmas[base + j] = dt * (Tn * AB - (cAB + sAB - wAB) - (cZ + sZ));
            d_sigmads[base + j] = dt * (Tn * u - (cU + sU - wU));
        }
        cAB += __shfl(sAB, 0, 64); cU += __shfl(sU, 0, 64); cZ += __shfl(sZ, 0, 64);
    }
}

// ------------------------------------------------------------------------------------------------ loss
// pred = colord / intensity  (d log I / dt), robust_e_nerf.py:394-398; Loss.log_intensity_grad, loss.py:43-57
__device__ __forceinline__ float err_val(int fn, float pred, float tgt) {
    const float d = pred - tgt;
    if (fn == 0) return fabsf(d);
    if (fn == 1) return d * d;
    return fabsf(d) / fmaxf(fabsf(tgt), 2.220446049250313e-16f);
}
__device__ __forceinline__ float err_grad(int fn, float pred, float tgt) {
    const float d = pred - tgt;
    const float sg = (float)((d > 0.f) - (d < 0.f));
    if (fn == 0) return sg;
    if (fn == 1) return 2.f * d;
    return sg / fmaxf(fabsf(tgt), 2.220446049250313e-16f);
}

__global__ __launch_bounds__(256) void grad_loss_fwd_kernel(const float *__restrict__ inten, const float *__restrict__ intend,
                                                            const float *__restrict__ target,
                                                            const uint8_t *__restrict__ valid, int64_t B, int fn,
                                                            float *__restrict__ loss_sum) {
    __shared__ float ps[4], pc[4];
    float s = 0.f, c = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < B; i += (int64_t)gridDim.x * blockDim.x) {
        if (valid && !valid[i]) continue;
        s += err_val(fn, intend[i] / inten[i], target[i]);
        c += 1.f;
    }
    s = ren_wave_sum(s); c = ren_wave_sum(c);
    if ((threadIdx.x & 63) == 0) { ps[threadIdx.x >> 6] = s; pc[threadIdx.x >> 6] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(loss_sum, ps[0] + ps[1] + ps[2] + ps[3]);
        atomicAdd(loss_sum + 1, pc[0] + pc[1] + pc[2] + pc[3]);
    }
}

__global__ void grad_loss_bwd_kernel(const float *__restrict__ inten, const float *__restrict__ intend,
                                     const float *__restrict__ target, const uint8_t *__restrict__ valid, int64_t B,
                                     int fn, float scale, const double *__restrict__ scale_dev, const float *__restrict__ loss_sum,
                                     float *__restrict__ g_inten, float *__restrict__ g_intend, float *__restrict__ loss_out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (scale_dev) scale = (float)((double)scale * scale_dev[0]);   // weight x param weight(mean contrast), the latter on the device
    if (i == 0 && loss_out) loss_out[0] = loss_sum[0] / loss_sum[1] * scale;
    if (i >= B) return;
    float g = 0.f;
    const float I = inten[i], Id = intend[i];
    const float pred = Id / I;
    if (!valid || valid[i]) g = scale / loss_sum[1] * err_grad(fn, pred, target[i]);
    g_intend[i] = g / I;
    g_inten[i] = -g * pred / I;
}

}  // namespace

extern "C" int ren_trajectory_jvp(const double *ts, int64_t B, const int64_t *tab_ts, const float *tab_pos,
                                  const float *tab_quat, int64_t C, float *pos, float *rot, float *dpos,
                                  float *drot, void *stream) {
    if (!ts || !tab_ts || !tab_pos || !tab_quat || !pos || !rot || !dpos || !drot || B < 0 || C < 2)
        return REN_ERR_BAD_ARG;
    if (B == 0) return REN_OK;
    hipLaunchKernelGGL(trajectory_jvp_kernel, dim3(ren_blocks(B, 256)), dim3(256), 0, (hipStream_t)stream, ts, B,
                       tab_ts, tab_pos, tab_quat, C, pos, rot, dpos, drot);
    REN_CHECK_LAUNCH();
}

extern "C" int ren_raygen_jvp(const float *Kinv, const float *px, const float *pos, const float *rot,
                              const float *dpos, const float *drot, int64_t B, float *rays_o, float *rays_d,
                              float *rays_do, float *rays_dd, void *stream) {
    if (!Kinv || !px || !pos || !rot || !dpos || !drot || !rays_o || !rays_d || !rays_do || !rays_dd || B < 0)
        return REN_ERR_BAD_ARG;
    if (B == 0) return REN_OK;
    hipLaunchKernelGGL(raygen_jvp_kernel, dim3(ren_blocks(B, 256)), dim3(256), 0, (hipStream_t)stream, Kinv, px, pos,
                       rot, dpos, drot, B, rays_o, rays_d, rays_do, rays_dd);
    REN_CHECK_LAUNCH();
}

extern "C" int ren_hashgrid_fwd_jvp(const ren_grid_desc *grid, const float *table, const ren_scene_desc *scene,
                                    const float *rays_o, const float *rays_d, const float *rays_do,
                                    const float *rays_dd, const int32_t *ray_indices, const float *t_starts,
                                    const float *t_ends, int64_t n, float *feat, float *featd, const int64_t *n_dev,
                                    void *stream) {
    GridDev g;
    int rc = make_grid(grid, g);
    if (rc) return rc;
    if (!table || !scene || !rays_o || !rays_d || !rays_do || !rays_dd || !ray_indices || !t_starts || !t_ends ||
        !feat || !featd || n < 0)
        return REN_ERR_BAD_ARG;
    if (g.n_levels != REN_MAX_LEVELS) return REN_ERR_UNSUPPORTED;
    if (n == 0) return REN_OK;
    const int64_t n_pad = ((n + 31) / 32) * 32;
    hipLaunchKernelGGL(hashgrid_fwd_jvp_kernel, dim3(ren_blocks(n_pad, 256), g.n_levels), dim3(256), 0,
                       (hipStream_t)stream, g, reinterpret_cast<const float2 *>(table), ren_make_scene(scene), rays_o,
                       rays_d, rays_do, rays_dd, ray_indices, t_starts, t_ends, n, n_pad, feat, featd, n_dev);
    REN_CHECK_LAUNCH();
}

extern "C" int ren_hashgrid_bwd_jvp(const ren_grid_desc *grid, float *grad_table, const ren_scene_desc *scene,
                                    const float *rays_o, const float *rays_d, const float *rays_do,
                                    const float *rays_dd, const int32_t *ray_indices, const float *t_starts,
                                    const float *t_ends, int64_t n, const float *dfeat, const float *dfeatd,
                                    void *stream) {
    GridDev g;
    int rc = make_grid(grid, g);
    if (rc) return rc;
    if (!grad_table || !scene || !rays_o || !rays_d || !rays_do || !rays_dd || !ray_indices || !t_starts ||
        !t_ends || !dfeat || !dfeatd || n < 0)
        return REN_ERR_BAD_ARG;
    if (g.n_levels != REN_MAX_LEVELS) return REN_ERR_UNSUPPORTED;
    if (n == 0) return REN_OK;
    hipLaunchKernelGGL(hashgrid_bwd_jvp_kernel, dim3(ren_blocks(n, 128), g.n_levels), dim3(256), 0,
                       (hipStream_t)stream, g, grad_table, ren_make_scene(scene), rays_o, rays_d, rays_do, rays_dd,
                       ray_indices, t_starts, t_ends, n, dfeat, dfeatd);
    REN_CHECK_LAUNCH();
}

extern "C" int ren_composite_fwd_jvp(const int64_t *offsets, const int32_t *counts, int64_t n_rays,
                                     const float *t_starts, const float *t_ends, const float *sigmas,
                                     const float *sigmads, const float *rgbs, const float *rgbds, int32_t C,
                                     const float *bkgd, float *colors, float *colords, float *opacities,
                                     float *opacds, float *weights, float *trans, float *eds, void *stream) {
    if (!offsets || !counts || !colors || !colords || !opacities || !opacds || !weights || !trans || !eds || n_rays < 0)
        return REN_ERR_BAD_ARG;
    if (C != 1 && C != 3) return REN_ERR_UNSUPPORTED;
    if (n_rays == 0) return REN_OK;
    dim3 grid(ren_blocks(n_rays, 4)), block(256);
    if (C == 1)
        hipLaunchKernelGGL(composite_fwd_jvp_kernel<1>, grid, block, 0, (hipStream_t)stream, offsets, counts, n_rays,
                           t_starts, t_ends, sigmas, sigmads, rgbs, rgbds, bkgd, colors, colords, opacities, opacds,
                           weights, trans, eds);
    else
        hipLaunchKernelGGL(composite_fwd_jvp_kernel<3>, grid, block, 0, (hipStream_t)stream, offsets, counts, n_rays,
                           t_starts, t_ends, sigmas, sigmads, rgbs, rgbds, bkgd, colors, colords, opacities, opacds,
                           weights, trans, eds);
    REN_CHECK_LAUNCH();
}

extern "C" int ren_composite_bwd_jvp(const int64_t *offsets, const int32_t *counts, int64_t n_rays,
                                     const float *t_starts, const float *t_ends, const float *sigmas,
                                     const float *sigmads, const float *rgbs, const float *rgbds, int32_t C,
                                     const float *bkgd, const float *weights, const float *trans, const float *eds,
                                     const float *opacities, const float *opacds, const float *g_colors,
                                     const float *g_colords, float *d_sigmas, float *d_sigmads, float *d_rgbs,
                                     float *d_rgbds, float *d_bkgd_per_ray, void *stream) {
    if (!offsets || !counts || !weights || !trans || !eds || !g_colords || !d_sigmas || !d_sigmads || !d_rgbs ||
        !d_rgbds || n_rays < 0)
        return REN_ERR_BAD_ARG;
    if (d_bkgd_per_ray && (!opacities || !opacds)) return REN_ERR_BAD_ARG;
    if (C != 1 && C != 3) return REN_ERR_UNSUPPORTED;
    if (n_rays == 0) return REN_OK;
    dim3 grid(ren_blocks(n_rays, 4)), block(256);
    if (C == 1)
        hipLaunchKernelGGL(composite_bwd_jvp_kernel<1>, grid, block, 0, (hipStream_t)stream, offsets, counts, n_rays,
                           t_starts, t_ends, sigmas, sigmads, rgbs, rgbds, bkgd, weights, trans, eds, opacities,
                           opacds, g_colors, g_colords, d_sigmas, d_sigmads, d_rgbs, d_rgbds, d_bkgd_per_ray);
    else
        hipLaunchKernelGGL(composite_bwd_jvp_kernel<3>, grid, block, 0, (hipStream_t)stream, offsets, counts, n_rays,
                           t_starts, t_ends, sigmas, sigmads, rgbs, rgbds, bkgd, weights, trans, eds, opacities,
                           opacds, g_colors, g_colords, d_sigmas, d_sigmads, d_rgbs, d_rgbds, d_bkgd_per_ray);
    REN_CHECK_LAUNCH();
}

extern "C" int ren_grad_loss_fwd(const float *intensity, const float *intensity_dot, const float *target,
                                 const uint8_t *valid, int64_t B, int32_t err_fn, float *loss_sum, void *stream) {
    if (!intensity || !intensity_dot || !target || !loss_sum || B < 0 || err_fn < 0 || err_fn > 2) return REN_ERR_BAD_ARG;
    if (hipMemsetAsync(loss_sum, 0, 2 * sizeof(float), (hipStream_t)stream) != hipSuccess) return REN_ERR_LAUNCH;
    if (B == 0) return REN_OK;
    int blocks = ren_blocks(B, 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(grad_loss_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, intensity, intensity_dot,
                       target, valid, B, err_fn, loss_sum);
    REN_CHECK_LAUNCH();
}

extern "C" int ren_grad_loss_bwd(const float *intensity, const float *intensity_dot, const float *target,
                                 const uint8_t *valid, int64_t B, int32_t err_fn, float scale, const double *scale_dev,
                                 const float *loss_sum, float *g_intensity, float *g_intensity_dot, float *loss, void *stream) {
    if (!intensity || !intensity_dot || !target || !loss_sum || !g_intensity || !g_intensity_dot || B < 0 ||
        err_fn < 0 || err_fn > 2)
        return REN_ERR_BAD_ARG;
    if (B == 0 && !loss) return REN_OK;
    hipLaunchKernelGGL(grad_loss_bwd_kernel, dim3(ren_blocks(B > 0 ? B : 1, 256)), dim3(256), 0, (hipStream_t)stream, intensity,
                       intensity_dot, target, valid, B, err_fn, scale, scale_dev, loss_sum, g_intensity, g_intensity_dot, loss);
    REN_CHECK_LAUNCH();
}
