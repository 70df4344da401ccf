// Second-order forward-mode render: value, d/dt and d2/dt2 of the rendered intensity along the camera
// trajectory, forward only.  Needed for ONE thing: the gradient of the log-intensity-gradient loss
// w.r.t. the (scalar) refractory period tau.  That loss compares d(log I)/dt at the supervision
// timestamp ts_g(tau) with the event rate, so d loss / d tau = sum_i [dL/dI I' + dL/dI' I''] d ts_g/d tau
// (robust_e_nerf/models/robust_e_nerf.py:340-357,383-409 differentiated through `grad.ts`; the reference
// gets it as a third-order autograd graph).  Everything is pointwise in t, sample placement is held
// fixed (external/vol_rendering.py:36-37) and cell/contraction branch boundaries are not differentiated,
// exactly like autograd.  Second-order truncated Taylor arithmetic (T2) carries (v, v', v'').
#include "ren_mlp_xfrag.h"
#include "ren_hashgrid_common.h"

namespace {

struct T2 { float v, d, e; };
__device__ __forceinline__ T2 t2(float v, float d = 0.f, float e = 0.f) { return T2{v, d, e}; }
__device__ __forceinline__ T2 operator+(const T2 &a, const T2 &b) { return T2{a.v + b.v, a.d + b.d, a.e + b.e}; }
__device__ __forceinline__ T2 operator-(const T2 &a, const T2 &b) { return T2{a.v - b.v, a.d - b.d, a.e - b.e}; }
__device__ __forceinline__ T2 operator*(const T2 &a, const T2 &b) {
    return T2{a.v * b.v, a.d * b.v + a.v * b.d, a.e * b.v + 2.f * a.d * b.d + a.v * b.e};
}
__device__ __forceinline__ T2 operator*(float s, const T2 &a) { return T2{s * a.v, s * a.d, s * a.e}; }
__device__ __forceinline__ T2 operator+(const T2 &a, float s) { return T2{a.v + s, a.d, a.e}; }
__device__ __forceinline__ T2 t2_rcp(const T2 &a) {
    const float r = 1.f / a.v, r2 = r * r;
    return T2{r, -r2 * a.d, -r2 * a.e + 2.f * r2 * r * a.d * a.d};
}
__device__ __forceinline__ T2 t2_sqrt(const T2 &a) {
    const float s = sqrtf(a.v), h = 0.5f / s;
    return T2{s, h * a.d, h * a.e - 0.5f * h * a.d * a.d / a.v};
}
__device__ __forceinline__ T2 t2_tanh(const T2 &a) {
    const float th = tanhf(a.v), c = 1.f - th * th;
    return T2{th, c * a.d, c * a.e - 2.f * th * c * a.d * a.d};
}
// exp(-a)
__device__ __forceinline__ T2 t2_expneg(const T2 &a) {
    const float e = expf(-a.v);
    return T2{e, -e * a.d, e * (a.d * a.d - a.e)};
}

// ------------------------------------------------------------------------------------------------ pose
// Same searchsorted / SLERP as trajectory_jvp_kernel (ren_jvp.hip); adds R'' = R S S with S = skew(rv)/bin
// (inside a pose segment the angular velocity is constant and the position is linear: p'' = 0).
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

__global__ void trajectory_jvp2_kernel(const double *__restrict__ ts, int64_t B, const int64_t *__restrict__ tab_ts,
                                       const float *__restrict__ tab_pos, const float *__restrict__ tab_quat,
                                       int64_t C, float *__restrict__ pos, float *__restrict__ rot,
                                       float *__restrict__ dpos, float *__restrict__ drot, float *__restrict__ ddrot) {
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
    const float vx = scale * rel.x, vy = scale * rel.y, vz = scale * rel.z;
    float rx = w * vx, ry = w * vy, rz = w * vz;
    float th = sqrtf(rx * rx + ry * ry + rz * rz);
    float t2v = th * th;
    float s = th <= 1e-3f ? 0.5f - t2v / 48.f + t2v * t2v / 3840.f : sinf(th * 0.5f) / th;
    Quat rq = {s * rx, s * ry, s * rz, cosf(th * 0.5f)};
    Quat q = qmul(q0, rq);
    float x2 = q.x * q.x, y2 = q.y * q.y, z2 = q.z * q.z, w2 = q.w * q.w;
    float xy = q.x * q.y, zw = q.z * q.w, xz = q.x * q.z, yw = q.y * q.w, yz = q.y * q.z, xw = q.x * q.w;
    float R[9];
    R[0] = x2 - y2 - z2 + w2; R[1] = 2.f * (xy - zw);     R[2] = 2.f * (xz + yw);
    R[3] = 2.f * (xy + zw);   R[4] = -x2 + y2 - z2 + w2;  R[5] = 2.f * (yz - xw);
    R[6] = 2.f * (xz - yw);   R[7] = 2.f * (yz + xw);     R[8] = -x2 - y2 + z2 + w2;
    for (int r = 0; r < 3; ++r) {
        const float a = R[3 * r], b = R[3 * r + 1], c = R[3 * r + 2];
        const float da = (b * vz - c * vy) * wdot, db = (c * vx - a * vz) * wdot, dc = (a * vy - b * vx) * wdot;
        rot[9 * i + 3 * r] = a; rot[9 * i + 3 * r + 1] = b; rot[9 * i + 3 * r + 2] = c;
        drot[9 * i + 3 * r] = da; drot[9 * i + 3 * r + 1] = db; drot[9 * i + 3 * r + 2] = dc;
        ddrot[9 * i + 3 * r] = (db * vz - dc * vy) * wdot;
        ddrot[9 * i + 3 * r + 1] = (dc * vx - da * vz) * wdot;
        ddrot[9 * i + 3 * r + 2] = (da * vy - db * vx) * wdot;
    }
}

__global__ void raygen_jvp2_kernel(const float *__restrict__ Kinv, const float *__restrict__ px,
                                   const float *__restrict__ pos, const float *__restrict__ rot,
                                   const float *__restrict__ dpos, const float *__restrict__ drot,
                                   const float *__restrict__ ddrot, int64_t B, float *__restrict__ rays_o,
                                   float *__restrict__ rays_d, float *__restrict__ rays_do, float *__restrict__ rays_dd,
                                   float *__restrict__ rays_ddd) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    const float u = px[2 * i], v = px[2 * i + 1];
    const float k[3] = {Kinv[0] * u + Kinv[1] * v + Kinv[2], Kinv[3] * u + Kinv[4] * v + Kinv[5],
                        Kinv[6] * u + Kinv[7] * v + Kinv[8]};
    T2 m[3];
    for (int r = 0; r < 3; ++r) {
        const float *R = rot + 9 * i + 3 * r, *dR = drot + 9 * i + 3 * r, *eR = ddrot + 9 * i + 3 * r;
        m[r] = t2(R[0] * k[0] + R[1] * k[1] + R[2] * k[2], dR[0] * k[0] + dR[1] * k[1] + dR[2] * k[2],
                  eR[0] * k[0] + eR[1] * k[1] + eR[2] * k[2]);
    }
    const T2 inv = t2_rcp(t2_sqrt(m[0] * m[0] + m[1] * m[1] + m[2] * m[2]));
    for (int r = 0; r < 3; ++r) {
        const T2 d = m[r] * inv;
        rays_d[3 * i + r] = d.v; rays_dd[3 * i + r] = d.d; rays_ddd[3 * i + r] = d.e;
        rays_o[3 * i + r] = pos[3 * i + r]; rays_do[3 * i + r] = dpos[3 * i + r];
    }
}

// ------------------------------------------------------------------------------------------------ hash grid
struct Ray2 {                                                     // packed sample stream + ray Taylor coefficients
    const float *o, *d, *od, *dd, *ddd;
    const int32_t *ray_indices;
    const float *t_starts, *t_ends;
};

// unit-cube position with first and second time derivative (o'' = 0 inside a pose segment)
__device__ __forceinline__ void unit_pos2(const Ray2 &r, const ren_scene_dev &sc, int64_t i, T2 *u, int &ray) {
    ray = r.ray_indices[i];
    const float tm = (r.t_starts[i] + r.t_ends[i]) * 0.5f;
    T2 y[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int64_t j = 3 * (int64_t)ray + k;
        const float inv = 1.f / (sc.hi[k] - sc.lo[k]);
        y[k] = t2((r.o[j] + r.d[j] * tm - sc.lo[k]) * inv, (r.od[j] + r.dd[j] * tm) * inv, r.ddd[j] * tm * inv);
    }
    if (sc.ct == REN_CT_SPHERE) {
#pragma unroll
        for (int k = 0; k < 3; ++k) y[k] = 2.f * y[k] + (-1.f);
        const T2 m = t2_sqrt(y[0] * y[0] + y[1] * y[1] + y[2] * y[2]);
        if (m.v > 1.f) {
            const T2 im = t2_rcp(m);
            const T2 g = (2.f * im) - im * im;                     // (2 - 1/m)/m
#pragma unroll
            for (int k = 0; k < 3; ++k) y[k] = y[k] * g;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) u[k] = 0.25f * y[k] + 0.5f;
    } else if (sc.ct == REN_CT_TANH) {
#pragma unroll
        for (int k = 0; k < 3; ++k) u[k] = 0.5f * (t2_tanh(y[k] + (-0.5f)) + 1.f);
    } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) u[k] = y[k];
    }
}

__global__ __launch_bounds__(256) void hashgrid_fwd_jvp2_kernel(GridDev g, const float2 *__restrict__ table,
                                                                ren_scene_dev sc, Ray2 r, int64_t n, int64_t n_pad,
                                                                float *__restrict__ feat, float *__restrict__ featd,
                                                                float *__restrict__ featdd, const int64_t *__restrict__ n_dev) {
    const int lvl = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n_dev) {                                             // device-side sample count (ren_common.h)
        n = ren_eff_n(n, n_dev);
        const int64_t p = ((n + 31) >> 5) << 5;
        n_pad = p < n_pad ? p : n_pad;
    }
    if (i >= n_pad) return;
    T2 f0 = t2(0.f), f1 = t2(0.f);
    if (i < n) {
        T2 u[3]; int ray;
        unit_pos2(r, sc, i, u, ray);
        const float scale = g.scale[lvl];
        const LevelPos p = level_pos(u[0].v, u[1].v, u[2].v, scale);
        T2 a1[3], a0[3];                                            // weight of the upper / lower corner per axis
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            a1[k] = t2(p.w[k], scale * u[k].d, scale * u[k].e);
            a0[k] = t2(1.f - p.w[k], -a1[k].d, -a1[k].e);
        }
        const uint32_t res = g.res[lvl], size = g.size[lvl];
        const bool hashed = g.hashed[lvl] != 0;
        float2 v[8];
        uint32_t idx[8];
        corner_indices8(p.c[0], p.c[1], p.c[2], res, size, hashed, idx);
        gather_corners8(table + g.offset[lvl], idx, v);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const T2 w = ((c & 1) ? a1[0] : a0[0]) * ((c & 2) ? a1[1] : a0[1]) * ((c & 4) ? a1[2] : a0[2]);
            f0 = f0 + v[c].x * w;
            f1 = f1 + v[c].y * w;
        }
    }
    const int64_t b = ((i >> 5) * REN_MAX_LEVELS + lvl) * 64 + (i & 31);
    feat[b] = f0.v; feat[b + 32] = f1.v;
    featd[b] = f0.d; featd[b + 32] = f1.d;
    featdd[b] = f0.e; featdd[b + 32] = f1.e;
}

// ------------------------------------------------------------------------------------------------ MLPs
// real SH degree 4 along d(t) in T2 arithmetic; component 2j+hi -> out[j]
__device__ __forceinline__ void sh4_t2_select(const T2 &x, const T2 &y, const T2 &z, int hi, T2 *out) {
    const T2 xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
    const float A = 0.48860251190291987f, Bc = 1.0925484305920792f, C6 = 0.94617469575755997f,
                E = 0.54627421529603959f, F = 0.59004358992664352f, G = 2.8906114426405538f,
                H = 0.45704579946446572f, K = 0.3731763325901154f, M = 1.4453057213202769f;
    T2 s[16];
    s[0] = t2(0.28209479177387814f);
    s[1] = -A * y;
    s[2] = A * z;
    s[3] = -A * x;
    s[4] = Bc * xy;
    s[5] = -Bc * yz;
    s[6] = C6 * z2 + (-0.31539156525251999f);
    s[7] = -Bc * xz;
    s[8] = E * (x2 - y2);
    s[9] = F * (y * (y2 - 3.f * x2));
    s[10] = G * (xy * z);
    s[11] = H * (y * (t2(1.f) - 5.f * z2));
    s[12] = K * (z * (5.f * z2 + (-3.f)));
    s[13] = H * (x * (t2(1.f) - 5.f * z2));
    s[14] = M * (z * (x2 - y2));
    s[15] = F * (x * (3.f * y2 - x2));
#pragma unroll
    for (int j = 0; j < 8; ++j) out[j] = hi ? s[2 * j + 1] : s[2 * j];
}

// softplus(beta = 100) with first and second tangent, in place (z, zd, ze) -> (y, yd, ye):
// y' = s z', y'' = s z'' + beta (1 - s) s z'^2 with s = 1 - exp(-beta y)
// (k: hidden-activation kind, ren_mlp_common.h act_kinds; relu: s = [y > 0], no curvature)
__device__ __forceinline__ void act2(f32x16 (&y)[2], f32x16 (&yd)[2], f32x16 (&ye)[2], int k) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const float v = act_hidden(y[r][g], k);
            const float s = dact_hidden(v, k);
            const float zd = yd[r][g];
            ye[r][g] = s * ye[r][g] + d2act_hidden(s, k) * zd * zd;
            yd[r][g] = s * zd;
            y[r][g] = v;
        }
}

struct Fwd2Args {
    const float *params, *feat, *featd, *featdd;
    Ray2 ray;
    ren_scene_dev sc;
    int64_t n;
    float *rgb, *rgbd, *rgbdd, *sigma, *sigmad, *sigmadd;
    const int64_t *n_dev;                          // device-side sample count or NULL (x kernel)
    int act_code;                                  // f32 kernel: activation alternatives (ren_mlp_common.h); the x kernel implements 0 only
};

template <int C>
__global__ __launch_bounds__(256, 1) void mlp_fwd_jvp2_kernel(Fwd2Args a) {
    const ActKinds ak = act_kinds(a.act_code);
    extern __shared__ __attribute__((aligned(16))) float lds_base[];
    fill_base(lds_base, a.params, L_W1, L_W2, L_B1, L_B2);
    fill_head(lds_base, a.params, C, L_WH1, L_WH2, L_WH3, L_BH1, L_BH2, L_BH3);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hi = lane >> 5, sl = lane & 31;
    const int64_t n_blk = (a.n + 31) >> 5;
    for (int64_t blk = (int64_t)blockIdx.x * 4 + wave; blk < n_blk; blk += (int64_t)gridDim.x * 4) {
        int zo = 0;
        asm volatile("" : "+v"(zo));
        const float *lds = lds_base + zo;
        const float *W1 = lds + L_W1, *W2 = lds + L_W2, *WH1 = lds + L_WH1, *WH2 = lds + L_WH2;
        const int64_t i = blk * 32 + sl;
        const bool live = i < a.n;
        f32x16 h[2], hd[2], he[2];
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            h[0][g] = lds[L_B1 + rowc(g) + 4 * hi]; h[1][g] = lds[L_B1 + 32 + rowc(g) + 4 * hi];
            hd[0][g] = 0.f; hd[1][g] = 0.f; he[0][g] = 0.f; he[1][g] = 0.f;
        }
        {
            const int64_t fo = blk * (16 * 64) + lane;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const float x = a.feat[fo + s * 64], xd = a.featd[fo + s * 64], xe = a.featdd[fo + s * 64];
                const float a0 = W1[sl * 33 + 2 * s + hi], a1 = W1[(32 + sl) * 33 + 2 * s + hi];
                h[0] = MFMA(a0, x, h[0]); h[1] = MFMA(a1, x, h[1]);
                hd[0] = MFMA(a0, xd, hd[0]); hd[1] = MFMA(a1, xd, hd[1]);
                he[0] = MFMA(a0, xe, he[0]); he[1] = MFMA(a1, xe, he[1]);
            }
        }
        act2(h, hd, he, ak.bh);
        f32x16 o, od, oe;
#pragma unroll
        for (int g = 0; g < 16; ++g) { o[g] = lds[L_B2 + rowc(g) + 4 * hi]; od[g] = 0.f; oe[g] = 0.f; }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const float aw = W2[sl * 65 + 32 * r + rowc(g) + 4 * hi];
                o = MFMA(aw, h[r][g], o);
                od = MFMA(aw, hd[r][g], od);
                oe = MFMA(aw, he[r][g], oe);
            }
        bool sel = false;
        T2 dir[3] = {t2(0.f), t2(0.f), t2(1.f)};
        if (live) {
            T2 u[3]; int ray;
            unit_pos2(a.ray, a.sc, i, u, ray);
            sel = u[0].v > 0.f && u[0].v < 1.f && u[1].v > 0.f && u[1].v < 1.f && u[2].v > 0.f && u[2].v < 1.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int64_t j = 3 * (int64_t)ray + k;
                dir[k] = t2(a.ray.d[j], a.ray.dd[j], a.ray.ddd[j]);
            }
        }
        if (live && hi == 0) {
            // trunc_exp (ngp.py:45-65): value exp(x), derivative exp(min(x, 15))
            // (other densities: softplus / shifted_softplus, nerf.py:8-13,21-25)
            const float ec = dact_density(o[0], ak.dn);
            a.sigma[i] = sel ? act_density(o[0], ak.dn) : 0.f;
            a.sigmad[i] = sel ? ec * od[0] : 0.f;
            a.sigmadd[i] = sel ? ec * oe[0] + d2act_density(o[0], ec, ak.dn) * od[0] * od[0] : 0.f;
        }
        T2 shs[8];
        sh4_t2_select(dir[0], dir[1], dir[2], hi, shs);
        f32x16 p[2], pd[2], pe[2];
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            p[0][g] = lds[L_BH1 + rowc(g) + 4 * hi]; p[1][g] = lds[L_BH1 + 32 + rowc(g) + 4 * hi];
            pd[0][g] = 0.f; pd[1][g] = 0.f; pe[0][g] = 0.f; pe[1][g] = 0.f;
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int col = s < 8 ? rowc(s) + 4 * hi : 16 + 2 * (s - 8) + hi;
            const float bv = s < 8 ? o[s] : shs[s < 8 ? 0 : s - 8].v;
            const float bd = s < 8 ? od[s] : shs[s < 8 ? 0 : s - 8].d;
            const float be = s < 8 ? oe[s] : shs[s < 8 ? 0 : s - 8].e;
            const float a0 = WH1[sl * 33 + col], a1 = WH1[(32 + sl) * 33 + col];
            p[0] = MFMA(a0, bv, p[0]); p[1] = MFMA(a1, bv, p[1]);
            pd[0] = MFMA(a0, bd, pd[0]); pd[1] = MFMA(a1, bd, pd[1]);
            pe[0] = MFMA(a0, be, pe[0]); pe[1] = MFMA(a1, be, pe[1]);
        }
        act2(p, pd, pe, ak.hh);
        f32x16 q[2], qd[2], qe[2];
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            q[0][g] = lds[L_BH2 + rowc(g) + 4 * hi]; q[1][g] = lds[L_BH2 + 32 + rowc(g) + 4 * hi];
            qd[0][g] = 0.f; qd[1][g] = 0.f; qe[0][g] = 0.f; qe[1][g] = 0.f;
        }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int col = 32 * r + rowc(g) + 4 * hi;
                const float a0 = WH2[sl * 65 + col], a1 = WH2[(32 + sl) * 65 + col];
                q[0] = MFMA(a0, p[r][g], q[0]); q[1] = MFMA(a1, p[r][g], q[1]);
                qd[0] = MFMA(a0, pd[r][g], qd[0]); qd[1] = MFMA(a1, pd[r][g], qd[1]);
                qe[0] = MFMA(a0, pe[r][g], qe[0]); qe[1] = MFMA(a1, pe[r][g], qe[1]);
            }
        act2(q, qd, qe, ak.hh);
        float acc[C], accd[C], acce[C];
#pragma unroll
        for (int c = 0; c < C; ++c) { acc[c] = 0.f; accd[c] = 0.f; acce[c] = 0.f; }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int g = 0; g < 16; ++g)
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const float w3 = lds[L_WH3 + c * 64 + 32 * r + rowc(g) + 4 * hi];
                    acc[c] += q[r][g] * w3; accd[c] += qd[r][g] * w3; acce[c] += qe[r][g] * w3;
                }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float z3 = acc[c] + __shfl_xor(acc[c], 32, 64) + lds[L_BH3 + c];
            const float z3d = accd[c] + __shfl_xor(accd[c], 32, 64);
            const float z3e = acce[c] + __shfl_xor(acce[c], 32, 64);
            if (hi == 0 && live) {
                const float y = act_radiance(z3, ak.rd), s = dact_radiance(y, ak.rd);
                a.rgb[i * C + c] = y;
                a.rgbd[i * C + c] = s * z3d;
                a.rgbdd[i * C + c] = s * z3e + d2act_radiance(y, s, ak.rd) * z3d * z3d;
            }
        }
    }
}

// ---- the same forward on the bf16 matrix cores (MODE 6: split-bf16 at fp32 accuracy, MODE 1: plain bf16 operands) ----------
// Value, first and second tangent share every weight-fragment read (one ds_read_b128, three MFMAs with independent
// accumulators); activations are split into bf16 pieces one 8-wide k-chunk at a time, so only one chunk of operand
// pieces of the three streams is live.  Fragment layouts and the LDS image are those of csrc/ren_mlp_jvp_x.hip.
template <int NT> struct XL2 {
    static constexpr int F_W1 = 0;                              // 2 tiles x 2 chunks
    static constexpr int F_W2 = F_W1 + 2 * 2 * NT * 512;        // 1 x 4
    static constexpr int F_WH1 = F_W2 + 1 * 4 * NT * 512;       // 2 x 2
    static constexpr int F_WH2 = F_WH1 + 2 * 2 * NT * 512;      // 2 x 4
    static constexpr int F_END = F_WH2 + 2 * 4 * NT * 512;      // bf16 elements
    static constexpr int BYTES_F = F_END * 2;
    static constexpr int T_B1 = 0, T_B2 = 64, T_BH1 = 96, T_BH2 = 160, T_WH3 = 224, T_BH3 = 416, T_END = 420;
    static constexpr size_t BYTES = (size_t)BYTES_F + T_END * 4;
};

template <int MODE, int TILES>
__device__ __forceinline__ void mma_j3(f32x16 *a, f32x16 *d, f32x16 *e, const __bf16 *frag, int chunks, int c,
                                       const bf16x8 (&b)[3], const bf16x8 (&bd)[3], const bf16x8 (&be)[3], int lane) {
    using PR = Pairs<MODE>;
#pragma unroll
    for (int k = 0; k < PR::N; ++k)
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
            const bf16x8 w = ldfrag<PR::NT>(frag, t, chunks, c, PR::W[k], lane);
            a[t] = MFMAB(w, b[PR::A[k]], a[t]);
            d[t] = MFMAB(w, bd[PR::A[k]], d[t]);
            e[t] = MFMAB(w, be[PR::A[k]], e[t]);
        }
}

// k-chunk c (8 inputs per lane) of an activated 64-wide layer held as two accumulator tiles -> bf16 operand pieces
template <int NT>
__device__ __forceinline__ void chunk_pieces(const f32x16 (&y)[2], int c, bf16x8 (&b)[3]) {
    float v[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) v[g] = y[c >> 1][8 * (c & 1) + g];
    split8<NT>(v, b);
}

template <int C, int MODE>
__global__ __launch_bounds__(256, 2) void mlp_fwd_jvp2_x_kernel(Fwd2Args a) {
    using PR = Pairs<MODE>;
    constexpr int NT = PR::NT;
    using L = XL2<NT>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem2[];
    __bf16 *frag = reinterpret_cast<__bf16 *>(smem2);
    float *tail = reinterpret_cast<float *>(smem2 + L::BYTES_F);
    fill_frags<NT, 0>(frag + L::F_W1, a.params, 2, 2);
    fill_frags<NT, 1>(frag + L::F_W2, a.params, 1, 4);
    fill_frags<NT, 2>(frag + L::F_WH1, a.params, 2, 2);
    fill_frags<NT, 3>(frag + L::F_WH2, a.params, 2, 4);
    for (int i = threadIdx.x; i < 64; i += blockDim.x) {
        tail[L::T_B1 + i] = a.params[P_BB0 + i];
        tail[L::T_BH1 + i] = a.params[P_HB0 + i];
        tail[L::T_BH2 + i] = a.params[P_HB1 + i];
        if (i < 32) tail[L::T_B2 + i] = i < 16 ? a.params[P_BBO + i] : 0.f;
    }
    for (int i = threadIdx.x; i < 64 * C; i += blockDim.x) tail[L::T_WH3 + i] = a.params[P_HWO + i];
    if (threadIdx.x < C) tail[L::T_BH3 + threadIdx.x] = a.params[P_HWO + 64 * C + threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hi = lane >> 5, sl = lane & 31;
    const int64_t n_smp = ren_eff_n(a.n, a.n_dev), n_blk = (n_smp + 31) >> 5;
    for (int64_t blk = (int64_t)blockIdx.x * 4 + wave; blk < n_blk; blk += (int64_t)gridDim.x * 4) {
        int zo = 0;                                             // keep the (loop-invariant) LDS reads inside the loop
        asm volatile("" : "+v"(zo));
        const __bf16 *fr = frag + zo;
        const float *tl = tail + zo;
        const int64_t i = blk * 32 + sl;
        const bool live = i < n_smp;
        // ---- base layer 0 on the hash features and their two tangents
        f32x16 h[2], hd[2], he[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 16; ++g) { h[t][g] = tl[L::T_B1 + 32 * t + rowc(g) + 4 * hi]; hd[t][g] = 0.f; he[t][g] = 0.f; }
        {
            const int64_t fo = blk * (16 * 64) + lane;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                float x[8], xd[8], xe[8];
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    x[s] = a.feat[fo + (8 * c + s) * 64]; xd[s] = a.featd[fo + (8 * c + s) * 64]; xe[s] = a.featdd[fo + (8 * c + s) * 64];
                }
                bf16x8 b[3], bd[3], be[3];
                split8<NT>(x, b); split8<NT>(xd, bd); split8<NT>(xe, be);
                mma_j3<MODE, 2>(h, hd, he, fr + L::F_W1, 2, c, b, bd, be, lane);
            }
        }
        // Compiler barrier before every activation of this kernel: without it hipcc (ROCm 7.2) interleaves the activation's
        // VALU code with the tail of the MFMA group and the FIRST launch of the mode-6 kernel in a process returns wrong
        // second tangents (the stream whose MFMA issues last) for one or two half-blocks of 16 samples -- reproducible in
        // nine of ten fresh processes, never on a repeat launch, gone with the barrier (tools/jvp2_first_launch.py,
        // test_second_order_mlp_forward_matrix_core_kernel_vs_f32_kernel).
        asm volatile("" ::: "memory");
        act2(h, hd, he, 0);
        // ---- base output: 64 -> 16 (rows 16..31 of the tile are zero)
        f32x16 o, od, oe;
#pragma unroll
        for (int g = 0; g < 16; ++g) { o[g] = tl[L::T_B2 + rowc(g) + 4 * hi]; od[g] = 0.f; oe[g] = 0.f; }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            bf16x8 b[3], bd[3], be[3];
            chunk_pieces<NT>(h, c, b); chunk_pieces<NT>(hd, c, bd); chunk_pieces<NT>(he, c, be);
            mma_j3<MODE, 1>(&o, &od, &oe, fr + L::F_W2, 4, c, b, bd, be, lane);
        }
        bool sel = false;
        T2 dir[3] = {t2(0.f), t2(0.f), t2(1.f)};
        if (live) {
            T2 u[3]; int ray;
            unit_pos2(a.ray, a.sc, i, u, ray);
            sel = u[0].v > 0.f && u[0].v < 1.f && u[1].v > 0.f && u[1].v < 1.f && u[2].v > 0.f && u[2].v < 1.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int64_t j = 3 * (int64_t)ray + k;
                dir[k] = t2(a.ray.d[j], a.ray.dd[j], a.ray.ddd[j]);
            }
        }
        if (live && hi == 0) {
            // trunc_exp (ngp.py:45-65): value exp(x), derivative exp(min(x, 15))
            const float xr = o[0] - 1.f, ec = __expf(fminf(xr, 15.f));
            a.sigma[i] = sel ? __expf(xr) : 0.f;
            a.sigmad[i] = sel ? ec * od[0] : 0.f;
            a.sigmadd[i] = sel ? ec * (oe[0] + (xr < 15.f ? od[0] * od[0] : 0.f)) : 0.f;
        }
        // ---- head layer 0: [base_out(16) | SH(16)] -> 64
        f32x16 p[2], pd[2], pe[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 16; ++g) { p[t][g] = tl[L::T_BH1 + 32 * t + rowc(g) + 4 * hi]; pd[t][g] = 0.f; pe[t][g] = 0.f; }
        {
            T2 shs[8];
            sh4_t2_select(dir[0], dir[1], dir[2], hi, shs);
            float v[8], vd[8], ve[8];
            bf16x8 b[3], bd[3], be[3];
#pragma unroll
            for (int g = 0; g < 8; ++g) { v[g] = o[g]; vd[g] = od[g]; ve[g] = oe[g]; }
            split8<NT>(v, b); split8<NT>(vd, bd); split8<NT>(ve, be);
            mma_j3<MODE, 2>(p, pd, pe, fr + L::F_WH1, 2, 0, b, bd, be, lane);
#pragma unroll
            for (int g = 0; g < 8; ++g) { v[g] = shs[g].v; vd[g] = shs[g].d; ve[g] = shs[g].e; }
            split8<NT>(v, b); split8<NT>(vd, bd); split8<NT>(ve, be);
            mma_j3<MODE, 2>(p, pd, pe, fr + L::F_WH1, 2, 1, b, bd, be, lane);
        }
        asm volatile("" ::: "memory");
        act2(p, pd, pe, 0);
        // ---- head layer 1: 64 -> 64
        f32x16 q[2], qd[2], qe[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 16; ++g) { q[t][g] = tl[L::T_BH2 + 32 * t + rowc(g) + 4 * hi]; qd[t][g] = 0.f; qe[t][g] = 0.f; }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            bf16x8 b[3], bd[3], be[3];
            chunk_pieces<NT>(p, c, b); chunk_pieces<NT>(pd, c, bd); chunk_pieces<NT>(pe, c, be);
            mma_j3<MODE, 2>(q, qd, qe, fr + L::F_WH2, 4, c, b, bd, be, lane);
        }
        asm volatile("" ::: "memory");
        act2(q, qd, qe, 0);
        // ---- head output: 64 -> C on the VALU in fp32 (MODE 1: bf16-rounded operands, as every other layer)
        float acc[C], accd[C], acce[C];
#pragma unroll
        for (int c = 0; c < C; ++c) { acc[c] = 0.f; accd[c] = 0.f; acce[c] = 0.f; }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const float qa = MODE == 1 ? (float)(__bf16)q[r][g] : q[r][g];
                const float qb = MODE == 1 ? (float)(__bf16)qd[r][g] : qd[r][g];
                const float qc = MODE == 1 ? (float)(__bf16)qe[r][g] : qe[r][g];
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    float w3 = tl[L::T_WH3 + c * 64 + 32 * r + rowc(g) + 4 * hi];
                    if (MODE == 1) w3 = (float)(__bf16)w3;
                    acc[c] += qa * w3; accd[c] += qb * w3; acce[c] += qc * w3;
                }
            }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float z3 = acc[c] + __shfl_xor(acc[c], 32, 64) + tl[L::T_BH3 + c];
            const float z3d = accd[c] + __shfl_xor(accd[c], 32, 64);
            const float z3e = acce[c] + __shfl_xor(acce[c], 32, 64);
            if (hi == 0 && live) {
                const float y = softplus1(z3), s = dsoftplus_from_out(y, 1.f);
                a.rgb[i * C + c] = y;
                a.rgbd[i * C + c] = s * z3d;
                a.rgbdd[i * C + c] = s * z3e + (1.f - s) * s * z3d * z3d;
            }
        }
    }
}

template <int MODE>
int launch_fwd_jvp2_x(const Fwd2Args &a, int C, hipStream_t st) {
    using L = XL2<Pairs<MODE>::NT>;
    const int64_t n_blk = (a.n + 31) / 32;
    int64_t blocks = (n_blk + 3) / 4;
    if (blocks > 512) blocks = 512;                             // two persistent workgroups per CU
#define REN_J2X(CC)                                                                                          \
    do {                                                                                                     \
        (void)hipFuncSetAttribute((const void *)mlp_fwd_jvp2_x_kernel<CC, MODE>,                              \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)L::BYTES);                \
        hipLaunchKernelGGL((mlp_fwd_jvp2_x_kernel<CC, MODE>), dim3((int)blocks), dim3(256), L::BYTES, st, a); \
    } while (0)
    if (C == 1) REN_J2X(1); else REN_J2X(3);
#undef REN_J2X
    REN_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------ compositing
__device__ __forceinline__ float wave_incl_scan2(float v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        float t = __shfl_up(v, off, 64);
        if (lane >= off) v += t;
    }
    return v;
}

// a = sigma dt (T2), E = exclusive prefix(a), T = exp(-E), alpha = 1 - exp(-a), w = T alpha;
// color = sum w c + bk (1 - sum w): value, first and second time derivative.
template <int C>
__global__ __launch_bounds__(256) void composite_fwd_jvp2_kernel(
    const int64_t *__restrict__ offsets, const int32_t *__restrict__ counts, int64_t n_rays,
    const float *__restrict__ t_starts, const float *__restrict__ t_ends, const float *__restrict__ sg,
    const float *__restrict__ sgd, const float *__restrict__ sge, const float *__restrict__ rgb,
    const float *__restrict__ rgbd, const float *__restrict__ rgbe, const float *__restrict__ bkgd,
    float *__restrict__ colors, float *__restrict__ colords, float *__restrict__ colorsdd) {
    const int lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (ray >= n_rays) return;
    const int64_t base = offsets[ray];
    const int cnt = counts[ray];
    T2 carry = t2(0.f), acc_o = t2(0.f), acc_c[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc_c[c] = t2(0.f);
    for (int s = 0; s < cnt; s += 64) {
        const int j = s + lane;
        const bool act = j < cnt;
        T2 a = t2(0.f);
        if (act) {
            const float dt = t_ends[base + j] - t_starts[base + j];
            a = t2(sg[base + j] * dt, sgd[base + j] * dt, sge[base + j] * dt);
        }
        const T2 inc = T2{wave_incl_scan2(a.v, lane), wave_incl_scan2(a.d, lane), wave_incl_scan2(a.e, lane)};
        const T2 E = carry + (inc - a);
        const T2 T = t2_expneg(E), ea = t2_expneg(a);
        const T2 w = T * (t2(1.f) - ea);
        if (act) {
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const int64_t k = (base + j) * C + c;
                acc_c[c] = acc_c[c] + w * t2(rgb[k], rgbd[k], rgbe[k]);
            }
            acc_o = acc_o + w;
        }
        carry = carry + T2{__shfl(inc.v, 63, 64), __shfl(inc.d, 63, 64), __shfl(inc.e, 63, 64)};
    }
    acc_o = T2{ren_wave_sum(acc_o.v), ren_wave_sum(acc_o.d), ren_wave_sum(acc_o.e)};
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const T2 col = T2{ren_wave_sum(acc_c[c].v), ren_wave_sum(acc_c[c].d), ren_wave_sum(acc_c[c].e)};
        if (lane == 0) {
            const float bk = bkgd ? bkgd[c] : 0.f;
            colors[ray * C + c] = col.v + bk * (1.f - acc_o.v);
            colords[ray * C + c] = col.d - bk * acc_o.d;
            colorsdd[ray * C + c] = col.e - bk * acc_o.e;
        }
    }
}

// ------------------------------------------------------------------------------------------------ arch mlp: d/dt, d2/dt2
// of the frequency encodings (SinusoidalEncoder, robust_e_nerf/external/mlp.py:208-243,333-352) of a packed sample
// stream: the position encoding of u(t) (contracted sample midpoint) and the view encoding of d(t).  Same feature
// order as freq_encode_kernel (csrc/ren_dense.hip): [x, sin(2^k x) (k-major, then axis), sin(2^k x + pi/2)].
// `order` picks the Taylor coefficient written (1: first, 2: second derivative); rows n..n_pad are zero.
template <int deg>
__device__ __forceinline__ void sin_enc_t2(const T2 *x, int order, float *out) {
#pragma unroll
    for (int j = 0; j < 3; ++j) out[j] = order == 1 ? x[j].d : x[j].e;
#pragma unroll
    for (int k = 0; k < deg; ++k) {
        const float sc = (float)(1 << k);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float xb = x[j].v * sc, xd = x[j].d * sc, xe = x[j].e * sc;
            const float sn = sinf(xb), cs = sinf(xb + 1.5707963267948966f);
            out[3 + 3 * k + j] = order == 1 ? cs * xd : cs * xe - sn * xd * xd;
            out[3 + 3 * deg + 3 * k + j] = order == 1 ? -sn * xd : -sn * xe - cs * xd * xd;
        }
    }
}

struct EncT2Args {
    Ray2 ray;
    ren_scene_dev sc;
    int64_t n, n_pad;
    int order;
    float *enc; int ld_enc;
    float *cat; int ld_cat, cat_col;
    float *view; int ld_view, view_col;
};

__global__ __launch_bounds__(256) void freq_encode_jvp_kernel(EncT2Args a) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_pad) return;
    float e[64], v[32];
#pragma unroll
    for (int j = 0; j < 64; ++j) e[j] = 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j) v[j] = 0.f;
    if (i < a.n) {
        T2 u[3]; int ray;
        unit_pos2(a.ray, a.sc, i, u, ray);
        const float TWO_PI = 6.283185307179586f, PI = 3.141592653589793f;
        T2 p[3], c[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            p[k] = TWO_PI * (u[k] + (-0.5f));
            const int64_t j = 3 * (int64_t)ray + k;
            c[k] = t2(PI * a.ray.d[j], PI * a.ray.dd[j], PI * a.ray.ddd[j]);
        }
        sin_enc_t2<10>(p, a.order, e);
        e[63] = 0.f;
        if (a.view) sin_enc_t2<4>(c, a.order, v);
    }
    float4 *o1 = reinterpret_cast<float4 *>(a.enc + i * a.ld_enc);
#pragma unroll
    for (int j = 0; j < 16; ++j) o1[j] = make_float4(e[4 * j], e[4 * j + 1], e[4 * j + 2], e[4 * j + 3]);
    if (a.cat) {
        float4 *o2 = reinterpret_cast<float4 *>(a.cat + i * a.ld_cat + a.cat_col);
#pragma unroll
        for (int j = 0; j < 16; ++j) o2[j] = make_float4(e[4 * j], e[4 * j + 1], e[4 * j + 2], e[4 * j + 3]);
    }
    if (a.view) {
#pragma unroll
        for (int j = 27; j < 32; ++j) v[j] = 0.f;
        float4 *o3 = reinterpret_cast<float4 *>(a.view + i * a.ld_view + a.view_col);
#pragma unroll
        for (int j = 0; j < 8; ++j) o3[j] = make_float4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
    }
}


}  // namespace

extern "C" int ren_trajectory_jvp2(const double *ts, int64_t B, const int64_t *tab_ts, const float *tab_pos,
                                   const float *tab_quat, int64_t C, float *pos, float *rot, float *dpos,
                                   float *drot, float *ddrot, void *stream) {
    if (!ts || !tab_ts || !tab_pos || !tab_quat || !pos || !rot || !dpos || !drot || !ddrot || B < 0 || C < 2)
        return REN_ERR_BAD_ARG;
    if (B == 0) return REN_OK;
    hipLaunchKernelGGL(trajectory_jvp2_kernel, dim3(ren_blocks(B, 256)), dim3(256), 0, (hipStream_t)stream, ts, B,
                       tab_ts, tab_pos, tab_quat, C, pos, rot, dpos, drot, ddrot);
    REN_CHECK_LAUNCH();
}

extern "C" int ren_raygen_jvp2(const float *Kinv, const float *px, const float *pos, const float *rot,
                               const float *dpos, const float *drot, const float *ddrot, int64_t B, float *rays_o,
                               float *rays_d, float *rays_do, float *rays_dd, float *rays_ddd, void *stream) {
    if (!Kinv || !px || !pos || !rot || !dpos || !drot || !ddrot || !rays_o || !rays_d || !rays_do || !rays_dd ||
        !rays_ddd || B < 0)
        return REN_ERR_BAD_ARG;
    if (B == 0) return REN_OK;
    hipLaunchKernelGGL(raygen_jvp2_kernel, dim3(ren_blocks(B, 256)), dim3(256), 0, (hipStream_t)stream, Kinv, px, pos,
                       rot, dpos, drot, ddrot, B, rays_o, rays_d, rays_do, rays_dd, rays_ddd);
    REN_CHECK_LAUNCH();
}

extern "C" int ren_hashgrid_fwd_jvp2(const ren_grid_desc *grid, const float *table, const ren_scene_desc *scene,
                                     const float *rays_o, const float *rays_d, const float *rays_do,
                                     const float *rays_dd, const float *rays_ddd, const int32_t *ray_indices,
                                     const float *t_starts, const float *t_ends, int64_t n, float *feat,
                                     float *featd, float *featdd, const int64_t *n_dev, void *stream) {
    GridDev g;
    int rc = make_grid(grid, g);
    if (rc) return rc;
    if (!table || !scene || !rays_o || !rays_d || !rays_do || !rays_dd || !rays_ddd || !ray_indices || !t_starts ||
        !t_ends || !feat || !featd || !featdd || n < 0)
        return REN_ERR_BAD_ARG;
    if (g.n_levels != REN_MAX_LEVELS) return REN_ERR_UNSUPPORTED;
    if (n == 0) return REN_OK;
    const int64_t n_pad = ((n + 31) / 32) * 32;
    const Ray2 r{rays_o, rays_d, rays_do, rays_dd, rays_ddd, ray_indices, t_starts, t_ends};
    hipLaunchKernelGGL(hashgrid_fwd_jvp2_kernel, dim3(ren_blocks(n_pad, 256), g.n_levels), dim3(256), 0,
                       (hipStream_t)stream, g, reinterpret_cast<const float2 *>(table), ren_make_scene(scene), r, n,
                       n_pad, feat, featd, featdd, n_dev);
    REN_CHECK_LAUNCH();
}

extern "C" int ren_mlp_fwd_jvp2(const float *mlp_params, int32_t C, int32_t activations, const float *feat, const float *featd,
                                const float *featdd, const ren_scene_desc *scene, const float *rays_o,
                                const float *rays_d, const float *rays_do, const float *rays_dd,
                                const float *rays_ddd, const int32_t *ray_indices, const float *t_starts,
                                const float *t_ends, int64_t n, float *rgb, float *rgbd, float *rgbdd, float *sigma,
                                float *sigmad, float *sigmadd, void *stream) {
    if (!mlp_params || !feat || !featd || !featdd || !scene || !rays_o || !rays_d || !rays_do || !rays_dd ||
        !rays_ddd || !ray_indices || !t_starts || !t_ends || !rgb || !rgbd || !rgbdd || !sigma || !sigmad ||
        !sigmadd || n < 0)
        return REN_ERR_BAD_ARG;
    if (C != 1 && C != 3) return REN_ERR_UNSUPPORTED;
    if (n == 0) return REN_OK;
    Fwd2Args a;
    a.params = mlp_params; a.feat = feat; a.featd = featd; a.featdd = featdd;
    a.ray = Ray2{rays_o, rays_d, rays_do, rays_dd, rays_ddd, ray_indices, t_starts, t_ends};
    a.sc = ren_make_scene(scene);
    a.n = n; a.rgb = rgb; a.rgbd = rgbd; a.rgbdd = rgbdd; a.sigma = sigma; a.sigmad = sigmad; a.sigmadd = sigmadd;
    a.act_code = activations; a.n_dev = nullptr;
    const int64_t n_blk = (n + 31) / 32;
    int64_t blocks = (n_blk + 3) / 4;
    if (blocks > 256) blocks = 256;
    const size_t lds = (size_t)L_WEIGHTS_END * 4;
    if (C == 1) hipLaunchKernelGGL(mlp_fwd_jvp2_kernel<1>, dim3((int)blocks), dim3(256), lds, (hipStream_t)stream, a);
    else        hipLaunchKernelGGL(mlp_fwd_jvp2_kernel<3>, dim3((int)blocks), dim3(256), lds, (hipStream_t)stream, a);
    REN_CHECK_LAUNCH();
}

extern "C" int ren_mlp_fwd_jvp2_x(const float *mlp_params, int32_t C, int32_t activations, int32_t mode, const float *feat, const float *featd,
                                  const float *featdd, const ren_scene_desc *scene, const float *rays_o, const float *rays_d,
                                  const float *rays_do, const float *rays_dd, const float *rays_ddd,
                                  const int32_t *ray_indices, const float *t_starts, const float *t_ends, int64_t n,
                                  float *rgb, float *rgbd, float *rgbdd, float *sigma, float *sigmad, float *sigmadd,
                                  const int64_t *n_dev, void *stream) {
    if (!mlp_params || !feat || !featd || !featdd || !scene || !rays_o || !rays_d || !rays_do || !rays_dd ||
        !rays_ddd || !ray_indices || !t_starts || !t_ends || !rgb || !rgbd || !rgbdd || !sigma || !sigmad ||
        !sigmadd || n < 0)
        return REN_ERR_BAD_ARG;
    if (C != 1 && C != 3) return REN_ERR_UNSUPPORTED;
    if (mode != 1 && mode != 3 && mode != 6) return REN_ERR_UNSUPPORTED;
    if (activations != 0) return REN_ERR_UNSUPPORTED;      // activation alternatives: exact-f32 kernels only
    if (n == 0) return REN_OK;
    Fwd2Args a;
    a.act_code = 0; a.n_dev = n_dev;
    a.params = mlp_params; a.feat = feat; a.featd = featd; a.featdd = featdd;
    a.ray = Ray2{rays_o, rays_d, rays_do, rays_dd, rays_ddd, ray_indices, t_starts, t_ends};
    a.sc = ren_make_scene(scene);
    a.n = n; a.rgb = rgb; a.rgbd = rgbd; a.rgbdd = rgbdd; a.sigma = sigma; a.sigmad = sigmad; a.sigmadd = sigmadd;
    if (mode == 3) return launch_fwd_jvp2_x<3>(a, C, (hipStream_t)stream);
    return mode == 6 ? launch_fwd_jvp2_x<6>(a, C, (hipStream_t)stream) : launch_fwd_jvp2_x<1>(a, C, (hipStream_t)stream);
}

extern "C" int ren_composite_fwd_jvp2(const int64_t *offsets, const int32_t *counts, int64_t n_rays,
                                      const float *t_starts, const float *t_ends, const float *sigmas,
                                      const float *sigmads, const float *sigmadds, const float *rgbs,
                                      const float *rgbds, const float *rgbdds, int32_t C, const float *bkgd,
                                      float *colors, float *colords, float *colorsdd, void *stream) {
    if (!offsets || !counts || !t_starts || !t_ends || !sigmas || !sigmads || !sigmadds || !rgbs || !rgbds ||
        !rgbdds || !colors || !colords || !colorsdd || n_rays < 0)
        return REN_ERR_BAD_ARG;
    if (C != 1 && C != 3) return REN_ERR_UNSUPPORTED;
    if (n_rays == 0) return REN_OK;
    dim3 grid(ren_blocks(n_rays, 4)), block(256);
    if (C == 1)
        hipLaunchKernelGGL(composite_fwd_jvp2_kernel<1>, grid, block, 0, (hipStream_t)stream, offsets, counts, n_rays,
                           t_starts, t_ends, sigmas, sigmads, sigmadds, rgbs, rgbds, rgbdds, bkgd, colors, colords,
                           colorsdd);
    else
        hipLaunchKernelGGL(composite_fwd_jvp2_kernel<3>, grid, block, 0, (hipStream_t)stream, offsets, counts, n_rays,
                           t_starts, t_ends, sigmas, sigmads, sigmadds, rgbs, rgbds, rgbdds, bkgd, colors, colords,
                           colorsdd);
    REN_CHECK_LAUNCH();
}

extern "C" int ren_freq_encode_jvp(const ren_scene_desc *scene, const float *rays_o, const float *rays_d,
                                   const float *rays_do, const float *rays_dd, const float *rays_ddd,
                                   const int32_t *ray_indices, const float *t_starts, const float *t_ends, int64_t n,
                                   int32_t order, float *enc, int32_t ld_enc, float *cat, int32_t ld_cat, int32_t cat_col,
                                   float *view, int32_t ld_view, int32_t view_col, void *stream) {
    if (!scene || !rays_o || !rays_d || !rays_do || !rays_dd || !rays_ddd || !ray_indices || !t_starts || !t_ends || !enc ||
        n < 0 || (order != 1 && order != 2) || ld_enc < 64 || (ld_enc & 3))
        return REN_ERR_BAD_ARG;
    if (cat && (ld_cat < cat_col + 64 || (ld_cat & 3) || (cat_col & 3))) return REN_ERR_BAD_ARG;
    if (view && (ld_view < view_col + 32 || (ld_view & 3) || (view_col & 3))) return REN_ERR_BAD_ARG;
    if (n == 0) return REN_OK;
    EncT2Args a;
    a.ray = Ray2{rays_o, rays_d, rays_do, rays_dd, rays_ddd, ray_indices, t_starts, t_ends};
    a.sc = ren_make_scene(scene);
    a.n = n; a.n_pad = (n + 31) / 32 * 32; a.order = order;
    a.enc = enc; a.ld_enc = ld_enc; a.cat = cat; a.ld_cat = ld_cat; a.cat_col = cat_col;
    a.view = view; a.ld_view = ld_view; a.view_col = view_col;
    hipLaunchKernelGGL(freq_encode_jvp_kernel, dim3(ren_blocks(a.n_pad, 256)), dim3(256), 0, (hipStream_t)stream, a);
    REN_CHECK_LAUNCH();
}
