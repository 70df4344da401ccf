// Pose interpolation (LERP + SLERP) and ray generation for event batches.
// Replaces robust_e_nerf/models/trajectories.py:30-91 (+ utils/tensor_ops.py:83-180, roma)
// and robust_e_nerf/models/nerf.py:206-228.  ~25 tiny torch launches -> one kernel each.
#include "ren_common.h"

namespace {

struct Quat { float x, y, z, w; };

__device__ __forceinline__ Quat qmul(const Quat &p, const Quat &q) {   // roma.quat_product, XYZW
    Quat r;
    r.x = p.w * q.x + q.w * p.x + (p.y * q.z - p.z * q.y);
    r.y = p.w * q.y + q.w * p.y + (p.z * q.x - p.x * q.z);
    r.z = p.w * q.z + q.w * p.z + (p.x * q.y - p.y * q.x);
    r.w = p.w * q.w - (p.x * q.x + p.y * q.y + p.z * q.z);
    return r;
}

__device__ __forceinline__ float lerpf(float a, float b, float w) {     // torch.lerp
    return fabsf(w) < 0.5f ? a + w * (b - a) : b - (b - a) * (1.f - w);
}

// pose at time t: position p[3] (if want_pos) and rotation R[9] row-major (if want_rot)
__device__ __forceinline__ void pose_eval(double t, const int64_t *__restrict__ tab_ts, const float *__restrict__ tab_pos,
                                          const float *__restrict__ tab_quat, int64_t C, bool want_pos, bool want_rot,
                                          float *p, float *R) {
    // torch.searchsorted(side='left'): first index with tab_ts[idx] >= t   (trajectories.py:50-52)
    int64_t lo = 0, hi = C;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if ((double)tab_ts[mid] < t) lo = mid + 1; else hi = mid;
    }
    int64_t right = lo < C ? lo : C - 1;
    int64_t left = (t == (double)tab_ts[0]) ? right : right - 1;          // :53-56
    if (left < 0) left = 0;
    int64_t wbin = left < C - 1 ? left : C - 2;
    const float w = (float)((t - (double)tab_ts[left]) /
                            (double)(tab_ts[wbin + 1] - tab_ts[wbin]));    // :63-65
    if (want_pos) {
        for (int k = 0; k < 3; ++k)
            p[k] = lerpf(tab_pos[3 * left + k], tab_pos[3 * right + k], w);   // :69-73
    }
    if (!want_rot) return;
    Quat q0 = {tab_quat[4 * left], tab_quat[4 * left + 1], tab_quat[4 * left + 2], tab_quat[4 * left + 3]};
    Quat q1 = {tab_quat[4 * right], tab_quat[4 * right + 1], tab_quat[4 * right + 2], tab_quat[4 * right + 3]};
    // tensor_ops.unitquat_slerp(shortest_path=True): flip, relative rotation, full rotvec
    float dot = q0.x * q1.x + q0.y * q1.y + q0.z * q1.z + q0.w * q1.w;
    if (dot < 0.f) { q1.x = -q1.x; q1.y = -q1.y; q1.z = -q1.z; q1.w = -q1.w; }
    Quat c0 = {-q0.x, -q0.y, -q0.z, q0.w};
    Quat rel = qmul(c0, q1);
    float vn = sqrtf(rel.x * rel.x + rel.y * rel.y + rel.z * rel.z);
    float angle = 2.f * atan2f(vn, rel.w);                                  // tensor_ops.py:100
    float a2 = angle * angle;
    float scale = fabsf(angle) <= 1e-3f ? 2.f + a2 / 12.f + 7.f * a2 * a2 / 2880.f
                                        : angle / sinf(angle * 0.5f);
    float rx = w * scale * rel.x, ry = w * scale * rel.y, rz = w * scale * rel.z;
    float th = sqrtf(rx * rx + ry * ry + rz * rz);                          // roma.rotvec_to_unitquat
    float t2 = th * th;
    float s = th <= 1e-3f ? 0.5f - t2 / 48.f + t2 * t2 / 3840.f : sinf(th * 0.5f) / th;
    Quat rq = {s * rx, s * ry, s * rz, cosf(th * 0.5f)};
    Quat q = qmul(q0, rq);
    // roma.unitquat_to_rotmat (no normalisation)
    float x2 = q.x * q.x, y2 = q.y * q.y, z2 = q.z * q.z, w2 = q.w * q.w;
    float xy = q.x * q.y, zw = q.z * q.w, xz = q.x * q.z, yw = q.y * q.w, yz = q.y * q.z, xw = q.x * q.w;
    R[0] = x2 - y2 - z2 + w2; R[1] = 2.f * (xy - zw);     R[2] = 2.f * (xz + yw);
    R[3] = 2.f * (xy + zw);   R[4] = -x2 + y2 - z2 + w2;  R[5] = 2.f * (yz - xw);
    R[6] = 2.f * (xz - yw);   R[7] = 2.f * (yz + xw);     R[8] = -x2 - y2 + z2 + w2;
}

__global__ void trajectory_kernel(const double *__restrict__ ts, int64_t B,
                                  const int64_t *__restrict__ tab_ts,
                                  const float *__restrict__ tab_pos,
                                  const float *__restrict__ tab_quat, int64_t C,
                                  float *__restrict__ pos, float *__restrict__ rot) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    float p[3], R[9];
    pose_eval(ts[i], tab_ts, tab_pos, tab_quat, C, pos != nullptr, rot != nullptr, p, R);
    if (pos) for (int k = 0; k < 3; ++k) pos[3 * i + k] = p[k];
    if (rot) for (int k = 0; k < 9; ++k) rot[9 * i + k] = R[k];
}

// d = R (Kinv [u, v, 1]^T) normalised, o = p  (models/nerf.py:206-228)
__device__ __forceinline__ void ray_eval(const float *__restrict__ Kinv, float u, float v, const float *p, const float *R,
                                         float *o, float *d) {
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

__global__ void raygen_kernel(const float *__restrict__ Kinv, const float *__restrict__ px,
                              const float *__restrict__ pos, const float *__restrict__ rot,
                              int64_t B, float *__restrict__ rays_o, float *__restrict__ rays_d) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    ray_eval(Kinv, px[2 * i], px[2 * i + 1], pos + 3 * i, rot + 9 * i, rays_o + 3 * i, rays_d + 3 * i);
}

// a5 + a6 in one launch: the pose at ts[i] and the ray of pixel px[i % px_rows] (the start / end renders of a step share
// the events' pixels: px_rows = B for R = 2B timestamps, no concatenated copy of the positions)
__global__ void pose_rays_kernel(const double *__restrict__ ts, int64_t R_, const float *__restrict__ px, int64_t px_rows,
                                 const float *__restrict__ Kinv, const int64_t *__restrict__ tab_ts,
                                 const float *__restrict__ tab_pos, const float *__restrict__ tab_quat, int64_t C,
                                 float *__restrict__ rays_o, float *__restrict__ rays_d) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R_) return;
    float p[3], R[9];
    pose_eval(ts[i], tab_ts, tab_pos, tab_quat, C, true, true, p, R);
    const int64_t j = i % px_rows;
    ray_eval(Kinv, px[2 * j], px[2 * j + 1], p, R, rays_o + 3 * i, rays_d + 3 * i);
}

}  // namespace

extern "C" int ren_trajectory_fwd(const double *ts, int64_t B, const int64_t *tab_ts,
                                  const float *tab_pos, const float *tab_quat, int64_t C,
                                  float *pos, float *rot, void *stream) {
    if (!ts || !tab_ts || !tab_pos || !tab_quat || B < 0 || C < 2) return REN_ERR_BAD_ARG;
    if (B == 0) return REN_OK;
    hipLaunchKernelGGL(trajectory_kernel, dim3(ren_blocks(B, 256)), dim3(256), 0, (hipStream_t)stream,
                       ts, B, tab_ts, tab_pos, tab_quat, C, pos, rot);
    REN_CHECK_LAUNCH();
}

extern "C" int ren_raygen_fwd(const float *Kinv, const float *px, const float *pos, const float *rot,
                              int64_t B, float *rays_o, float *rays_d, void *stream) {
    if (!Kinv || !px || !pos || !rot || !rays_o || !rays_d || B < 0) return REN_ERR_BAD_ARG;
    if (B == 0) return REN_OK;
    hipLaunchKernelGGL(raygen_kernel, dim3(ren_blocks(B, 256)), dim3(256), 0, (hipStream_t)stream,
                       Kinv, px, pos, rot, B, rays_o, rays_d);
    REN_CHECK_LAUNCH();
}

extern "C" int ren_pose_rays_fwd(const double *ts, int64_t R, const float *px, int64_t px_rows, const float *Kinv,
                                 const int64_t *tab_ts, const float *tab_pos, const float *tab_quat, int64_t C,
                                 float *rays_o, float *rays_d, void *stream) {
    if (!ts || !px || !Kinv || !tab_ts || !tab_pos || !tab_quat || !rays_o || !rays_d || R < 0 || px_rows < 1 || C < 2)
        return REN_ERR_BAD_ARG;
    if (R == 0) return REN_OK;
    hipLaunchKernelGGL(pose_rays_kernel, dim3(ren_blocks(R, 256)), dim3(256), 0, (hipStream_t)stream, ts, R, px, px_rows, Kinv,
                       tab_ts, tab_pos, tab_quat, C, rays_o, rays_d);
    REN_CHECK_LAUNCH();
}
