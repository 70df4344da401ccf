"""Oracle: ray/AABB intersection, occupancy-grid ray marching, visibility filtering.

TEST INFRASTRUCTURE ONLY.  Thin ctypes front-end over ``oracle/csrc/march.c`` (sequential
float32 C, see that file for the nerfacc citations) shaped like
``nerfacc.ray_marching`` as the reference calls it at
``robust_e_nerf/external/utils.py:106-119``.  PARITY UNPINNED (third-party semantics).
"""
from __future__ import annotations

import ctypes
from typing import Callable, Optional, Tuple

import numpy as np
import torch

from . import build as _build

_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(_build.build())
    return _lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


def ray_aabb_intersect(o: torch.Tensor, d: torch.Tensor, aabb: torch.Tensor):
    o_ = np.ascontiguousarray(o.detach().numpy(), dtype=np.float32)
    d_ = np.ascontiguousarray(d.detach().numpy(), dtype=np.float32)
    ab = np.ascontiguousarray(aabb.detach().numpy(), dtype=np.float32)
    n = o_.shape[0]
    tmin = np.empty(n, np.float32)
    tmax = np.empty(n, np.float32)
    lib().orc_ray_aabb_intersect(ctypes.c_int64(n), _p(o_), _p(d_), _p(ab), _p(tmin), _p(tmax))
    return torch.from_numpy(tmin), torch.from_numpy(tmax)


def march(
    o, d, t_min, t_max, roi, binary: Optional[torch.Tensor], contraction_type: int,
    step_size: float, cone_angle: float, mode: int = 0, n_uniform: int = 0,
    jitter: Optional[torch.Tensor] = None,
):
    """Two-pass packed marching -> (counts (R,) i32, offsets (R,) i64, ray_indices, t_starts, t_ends)."""
    o_ = np.ascontiguousarray(o.detach().numpy(), dtype=np.float32)
    d_ = np.ascontiguousarray(d.detach().numpy(), dtype=np.float32)
    tmin = np.ascontiguousarray(t_min.numpy(), dtype=np.float32)
    tmax = np.ascontiguousarray(t_max.numpy(), dtype=np.float32)
    roi_ = np.ascontiguousarray(roi.numpy(), dtype=np.float32)
    if binary is None:
        binary = torch.ones(1, 1, 1, dtype=torch.bool)
    res = np.asarray(binary.shape, dtype=np.int32)
    bin_ = np.ascontiguousarray(binary.numpy().astype(np.uint8))
    n = o_.shape[0]
    counts = np.zeros(n, np.int32)
    jit = None if jitter is None else np.ascontiguousarray(jitter.numpy(), dtype=np.float32)
    args = [ctypes.c_int64(n), _p(o_), _p(d_), _p(tmin), _p(tmax), _p(roi_), _p(res), _p(bin_),
            ctypes.c_int(contraction_type), ctypes.c_float(step_size), ctypes.c_float(cone_angle),
            ctypes.c_int(mode), ctypes.c_int(n_uniform), None if jit is None else _p(jit)]
    lib().orc_ray_march(*args, None, _p(counts), None, None, None)
    offsets = np.zeros(n, np.int64)
    np.cumsum(counts[:-1], out=offsets[1:])
    total = int(counts.sum())
    ts = np.empty(total, np.float32)
    te = np.empty(total, np.float32)
    ri = np.empty(total, np.int32)
    lib().orc_ray_march(*args, _p(offsets), _p(counts), _p(ts), _p(te), _p(ri))
    return (torch.from_numpy(counts), torch.from_numpy(offsets), torch.from_numpy(ri),
            torch.from_numpy(ts), torch.from_numpy(te))


def visibility(counts, offsets, sigmas, t_starts, t_ends, early_stop_eps: float, alpha_thre: float):
    n = counts.shape[0]
    keep = np.zeros(t_starts.shape[0], np.uint8)
    lib().orc_visibility(
        ctypes.c_int64(n), _p(np.ascontiguousarray(offsets.numpy())),
        _p(np.ascontiguousarray(counts.numpy())),
        _p(np.ascontiguousarray(sigmas.detach().numpy().astype(np.float32).reshape(-1))),
        _p(np.ascontiguousarray(t_starts.numpy())), _p(np.ascontiguousarray(t_ends.numpy())),
        ctypes.c_float(early_stop_eps), ctypes.c_float(alpha_thre), _p(keep))
    return torch.from_numpy(keep.astype(bool))


def ray_marching(
    rays_o: torch.Tensor, rays_d: torch.Tensor, *,
    scene_aabb: Optional[torch.Tensor] = None,
    grid_binary: Optional[torch.Tensor] = None, grid_roi: Optional[torch.Tensor] = None,
    contraction_type: int = 0,
    sigma_fn: Optional[Callable] = None,
    near_plane: Optional[float] = None, far_plane: Optional[float] = None,
    render_step_size: float = 1e-3, stratified: bool = False, cone_angle: float = 0.0,
    early_stop_eps: float = 1e-4, alpha_thre: float = 0.0,
    jitter: Optional[torch.Tensor] = None,
) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """nerfacc.ray_marching Python wrapper semantics:
    t_min/t_max from the scene AABB (else 0 / 1e10), near/far clamps, one uniform PER RAY
    shifting t_min by u*step when ``stratified`` (``jitter`` supplies the uniforms so CPU
    and GPU consume identical randomness), marching, then the no-grad sigma_fn visibility
    filter.  Returns (ray_indices (n,) int32, t_starts (n,1), t_ends (n,1))."""
    n = rays_o.shape[0]
    if scene_aabb is not None:
        t_min, t_max = ray_aabb_intersect(rays_o, rays_d, scene_aabb)
    else:
        t_min = torch.zeros(n)
        t_max = torch.full((n,), 1e10)
    if near_plane is not None:
        t_min = torch.clamp(t_min, min=near_plane)
    if far_plane is not None:
        t_max = torch.clamp(t_max, max=far_plane)
    if stratified:
        assert jitter is not None, "oracle takes the per-ray uniforms explicitly"
        t_min = t_min + jitter.to(torch.float32) * np.float32(render_step_size)
    roi = grid_roi if grid_roi is not None else torch.tensor([-1e10] * 3 + [1e10] * 3)
    counts, offsets, ri, ts, te = march(
        rays_o, rays_d, t_min, t_max, roi, grid_binary, contraction_type,
        render_step_size, cone_angle)
    if sigma_fn is not None:
        with torch.no_grad():
            sig = sigma_fn(ts[:, None], te[:, None], ri.long())
        keep = visibility(counts, offsets, sig, ts, te, early_stop_eps, alpha_thre)
        ri, ts, te = ri[keep], ts[keep], te[keep]
    return ri, ts[:, None], te[:, None]
