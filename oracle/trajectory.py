"""Oracle: pose interpolation (LERP + SLERP) and ray generation.

TEST INFRASTRUCTURE ONLY.  Follows ``robust_e_nerf/models/trajectories.py:30-91``,
``robust_e_nerf/utils/tensor_ops.py:83-180`` and ``robust_e_nerf/models/nerf.py:206-228``
(all PINNED through tests/golden, where the reference's own Python runs over a stub of
RoMa built from the formulas below).  The RoMa==1.2.7 primitives (XYZW quaternions) are
third-party and restated from its published formulas: PARITY UNPINNED
for those, although the composed trajectory matched SciPy's Slerp to 6e-7 (App. C.3).
"""
from __future__ import annotations

import torch


# ---- roma 1.2.7 primitives -------------------------------------------------------------
def quat_conjugation(q):
    return torch.cat([-q[..., :3], q[..., 3:]], dim=-1)


def quat_product(p, q):
    pv, pw = p[..., :3], p[..., 3:]
    qv, qw = q[..., :3], q[..., 3:]
    v = pw * qv + qw * pv + torch.cross(pv, qv, dim=-1)
    w = pw * qw - (pv * qv).sum(-1, keepdim=True)
    return torch.cat([v, w], dim=-1)


def rotvec_to_unitquat(r):
    theta = r.norm(dim=-1, keepdim=True)
    small = theta <= 1e-3
    t2 = theta * theta
    safe = torch.where(small, torch.ones_like(theta), theta)
    s = torch.where(small, 0.5 - t2 / 48 + t2 * t2 / 3840, torch.sin(safe / 2) / safe)
    return torch.cat([s * r, torch.cos(theta / 2)], dim=-1)


def unitquat_to_rotmat(q):
    x, y, z, w = q.unbind(-1)
    x2, y2, z2, w2 = x * x, y * y, z * z, w * w
    xy, zw, xz, yw, yz, xw = x * y, z * w, x * z, y * w, y * z, x * w
    m = torch.stack([
        x2 - y2 - z2 + w2, 2 * (xy - zw), 2 * (xz + yw),
        2 * (xy + zw), -x2 + y2 - z2 + w2, 2 * (yz - xw),
        2 * (xz - yw), 2 * (yz + xw), -x2 - y2 + z2 + w2,
    ], dim=-1)
    return m.reshape(*q.shape[:-1], 3, 3)


# ---- reference tensor_ops --------------------------------------------------------------
def unitquat_to_full_rotvec(q):
    """tensor_ops.py:83-111: angle in [0, 2pi]."""
    vn = q[..., :3].norm(dim=-1)
    angle = 2 * torch.atan2(vn, q[..., 3])
    small = angle.abs() <= 1e-3
    safe = torch.where(small, torch.ones_like(angle), angle)
    scale = torch.where(small, 2 + angle ** 2 / 12 + 7 * angle ** 4 / 2880, safe / torch.sin(safe / 2))
    return scale[..., None] * q[..., :3]


def unitquat_slerp(q0, q1, steps):
    """tensor_ops.py:114-180 with shortest_path=True and one step per quaternion pair."""
    q1 = torch.where((q0 * q1).sum(-1, keepdim=True) < 0, -q1, q1)
    rel = quat_product(quat_conjugation(q0), q1)
    rv = unitquat_to_full_rotvec(rel)
    rots = rotvec_to_unitquat(steps[..., None] * rv)
    return quat_product(q0, rots)


def linear_trajectory(ts, tab_ts, tab_pos, tab_quat):
    """trajectories.py:30-91.  ts (B,) float64 [ns]; tab_ts (C,) int64; -> p (B,3), R (B,3,3)."""
    right = torch.searchsorted(tab_ts, ts)
    left = torch.where(ts == tab_ts[0], right, right - 1)
    assert bool(((left >= 0) & (right < len(tab_ts))).all())
    bin_width = tab_ts.diff()
    w = ((ts - tab_ts[left]) / bin_width[left]).to(tab_pos.dtype)
    p = torch.lerp(tab_pos[left], tab_pos[right], w[:, None])
    q = unitquat_slerp(tab_quat[left], tab_quat[right], w)
    return p, unitquat_to_rotmat(q)


def pixel_params_to_ray(Kinv, px, p, R):
    """nerf.py:206-228: d = normalize(R (K^-1 [u v 1]^T)), o = p."""
    hom = torch.cat([px, torch.ones_like(px[..., :1])], dim=-1)[..., None]
    d = (R @ (Kinv @ hom)).squeeze(-1)
    d = d / torch.linalg.vector_norm(d, dim=-1, keepdim=True)
    return p, d
