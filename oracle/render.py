"""Oracle: packed volume rendering (weights, accumulation, background).

TEST INFRASTRUCTURE ONLY.  Restates the nerfacc==0.3.1 ops the reference calls at
``robust_e_nerf/external/vol_rendering.py:89-126`` (``render_weight_from_density``,
``accumulate_along_rays``) -- third-party, PARITY UNPINNED -- and the reference's own
``rendering`` glue (``vol_rendering.py:16-128``), which IS pinned through tests/golden.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch


def _segment_exclusive_cumsum(v: torch.Tensor, ray_indices: torch.Tensor) -> torch.Tensor:
    """Exclusive cumsum of v (n,) within runs of equal (sorted) ray index."""
    n = v.shape[0]
    if n == 0:
        return v
    inc = torch.cumsum(v.double(), dim=0)
    exc = inc - v.double()
    is_first = torch.ones(n, dtype=torch.bool)
    is_first[1:] = ray_indices[1:] != ray_indices[:-1]
    first_pos = torch.nonzero(is_first)[:, 0]
    seg_id = torch.cumsum(is_first.long(), 0) - 1
    return (exc - exc[first_pos][seg_id]).to(v.dtype)


def render_weight_from_density(t_starts, t_ends, sigmas, ray_indices, n_rays: int):
    """w_i = T_i (1 - exp(-sigma_i dt_i)), T_i = exp(-sum_{j<i} sigma_j dt_j).  (n,1) in/out."""
    sd = (sigmas * (t_ends - t_starts)).squeeze(-1)
    T = torch.exp(-_segment_exclusive_cumsum(sd, ray_indices))
    return (T * (1.0 - torch.exp(-sd)))[:, None]


def accumulate_along_rays(weights, ray_indices, values: Optional[torch.Tensor], n_rays: int):
    """zeros(n_rays, D).index_add_(0, ray_indices, w * values)  (values=None -> w)."""
    src = weights if values is None else weights * values
    out = torch.zeros(n_rays, src.shape[-1], dtype=src.dtype)
    return out.index_add(0, ray_indices.long(), src)


def rendering(
    t_starts, t_ends, ray_indices, n_rays: int, rgb_sigma_fn: Callable,
    render_bkgd: Optional[torch.Tensor] = None,
):
    """vol_rendering.py:16-128 (rgb_sigma_fn branch): colors, opacities, depths."""
    rgbs, sigmas = rgb_sigma_fn(t_starts, t_ends, ray_indices.long())
    assert rgbs.shape[-1] in (1, 3)
    assert sigmas.shape == t_starts.shape
    w = render_weight_from_density(t_starts, t_ends, sigmas, ray_indices, n_rays)
    colors = accumulate_along_rays(w, ray_indices, rgbs, n_rays)
    opac = accumulate_along_rays(w, ray_indices, None, n_rays)
    depths = accumulate_along_rays(w, ray_indices, (t_starts + t_ends) / 2.0, n_rays)
    if render_bkgd is not None:
        colors = colors + render_bkgd * (1.0 - opac)
    return colors, opac, depths
