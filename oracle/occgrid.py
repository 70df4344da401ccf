"""Oracle: occupancy-grid maintenance (nerfacc==0.3.1 ``OccupancyGrid``).

TEST INFRASTRUCTURE ONLY.  Reference call sites: ``robust_e_nerf/models/nerf.py:98-102``
(ctor), ``:170-204`` (``update_occ_grid`` -> ``every_n_step`` with the ``occ_eval_fn`` of
``:171-198``).  The grid policy lives in nerfacc (un-vendored): restated from its
published ``OccupancyGrid._update``.  PARITY UNPINNED.

Randomness (cell choice, in-cell jitter, camera choice for the cone step) is passed in
explicitly so the CPU oracle and the HIP path consume identical numbers.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch

from .field import AABB, UN_BOUNDED_SPHERE, UN_BOUNDED_TANH


def grid_coords(res) -> torch.Tensor:
    """(cells,3) int64, x-major flattening (ix*ry*rz + iy*rz + iz)."""
    rx, ry, rz = res
    g = torch.stack(torch.meshgrid(torch.arange(rx), torch.arange(ry), torch.arange(rz), indexing="ij"), -1)
    return g.reshape(-1, 3)


def contract_inv(x_unit: torch.Tensor, roi: torch.Tensor, contraction_type: int) -> torch.Tensor:
    """Unit cube -> world (nerfacc contraction.cu ``contract_inv``)."""
    lo, hi = roi[:3], roi[3:]
    if contraction_type == AABB:
        return x_unit * (hi - lo) + lo
    if contraction_type == UN_BOUNDED_SPHERE:
        x = (x_unit - 0.5) * 4.0                            # -> [-2, 2]
        mag = x.norm(dim=-1, keepdim=True)
        x = torch.where(mag > 1, x / torch.clamp(2 * mag - mag * mag, min=1e-10), x)  # inverse of (2 - 1/m) x/m
        x = x * 0.5 + 0.5
        return x * (hi - lo) + lo
    x = torch.atanh((x_unit * 2 - 1).clamp(-1 + 1e-6, 1 - 1e-6)) + 0.5
    return x * (hi - lo) + lo


def update(
    occs: torch.Tensor, res, roi: torch.Tensor, contraction_type: int,
    indices: torch.Tensor, jitter: torch.Tensor, occ_eval_fn: Callable,
    occ_thre: float = 1e-2, ema_decay: float = 0.95,
):
    """One ``_update``: occs[idx] = max(occs[idx]*decay, occ(x)); binary = occs > min(mean, thre).

    indices: (m,) int64 cells to refresh (all cells during warm-up, else 1/4 uniform + up to
    1/4 occupied); jitter: (m,3) uniforms in [0,1).  Returns (occs_new, binary (res) bool)."""
    coords = grid_coords(res)[indices].to(torch.float32)
    x = (coords + jitter) / torch.tensor(res, dtype=torch.float32)
    if contraction_type == UN_BOUNDED_SPHERE:
        mask = (x - 0.5).norm(dim=1) < 0.5
        x, indices = x[mask], indices[mask]
    xw = contract_inv(x, roi, contraction_type)
    with torch.no_grad():
        occ = occ_eval_fn(xw).squeeze(-1)
    occs = occs.clone()
    occs[indices] = torch.maximum(occs[indices] * ema_decay, occ)
    binary = (occs > torch.clamp(occs.mean(), max=occ_thre)).view(*res)
    return occs, binary


def occ_eval(x, density_fn: Callable, render_step_size: float, cone_angle: float = 0.0,
             cam_positions: Optional[torch.Tensor] = None, cam_ids: Optional[torch.Tensor] = None,
             near: Optional[float] = None, far: Optional[float] = None):
    """nerf.py:171-198: density(x) * step(x)."""
    if cone_angle > 0.0:
        t = (cam_positions[cam_ids] - x).norm(dim=-1, keepdim=True)
        step = torch.clamp(t * cone_angle, min=render_step_size)
        if near is not None and far is not None:
            step = torch.where((t > near) & (t < far), step, torch.zeros_like(step))
    else:
        step = render_step_size
    return density_fn(x) * step
