"""TEST-ONLY driver of the op-by-op seam (``robust_e_nerf_amd.nerfacc_api`` + ``field``).

The GPU box has no /root/reference, so the parity tests need something that calls the seam in the
order the reference's ``render_image`` (external/utils.py:38-140) and ``rendering``
(external/vol_rendering.py:16-128) do: march with a density callback, query the field at the
packed samples, weights from density, three accumulations, background.  With ``nerfacc_api``
registered as ``nerfacc`` the reference's own two files do the same thing unmodified
(tests/golden/check_seam.py); this file is not part of the product package."""
from __future__ import annotations

import torch

from robust_e_nerf_amd import nerfacc_api as seam


def _composite(ts, te, ray_idx, n_rays, rgb, sigma, bkgd):
    assert rgb.shape[-1] in (1, 3) and sigma.shape == ts.shape
    w = seam.render_weight_from_density(ts, te, sigma, ray_indices=ray_idx, n_rays=n_rays)
    acc = lambda v: seam.accumulate_along_rays(w, ray_idx, values=v, n_rays=n_rays)
    color, alpha, depth = acc(rgb), acc(None), acc(0.5 * (ts + te))
    if bkgd is not None:
        color = color + bkgd * (1.0 - alpha)
    return color, alpha, depth


def render_image(radiance_field, occupancy_grid, rays_o, rays_d, scene_aabb, near_plane=None, far_plane=None,
                 render_step_size=1e-3, render_bkgd=None, cone_angle=0.0, early_stop_eps=1e-4, alpha_thre=0.0,
                 test_chunk_size=8192, jitter=None):
    """-> colors (R,C), opacities (R,1), depths (R,1), number of rendered samples."""
    R = rays_o.shape[0]
    train = radiance_field.training
    step = R if train else test_chunk_size
    parts, n_samples = [], 0
    for lo in range(0, R, step):
        o, d = rays_o[lo:lo + step], rays_d[lo:lo + step]
        mid = lambda a, b, i: o[i] + d[i] * (a + b) * 0.5
        idx, ts, te = seam.ray_marching(
            o, d, scene_aabb=scene_aabb, grid=occupancy_grid, near_plane=near_plane, far_plane=far_plane,
            sigma_fn=lambda a, b, i: radiance_field.query_density(mid(a, b, i)),
            render_step_size=render_step_size, stratified=train, cone_angle=cone_angle,
            early_stop_eps=early_stop_eps, alpha_thre=alpha_thre,
            jitter=None if jitter is None else jitter[lo:lo + step])
        il = idx.long()
        rgb, sigma = radiance_field(mid(ts, te, il), d[il])
        parts.append(_composite(ts, te, idx, o.shape[0], rgb, sigma, render_bkgd))
        n_samples += ts.shape[0]
    color, alpha, depth = (torch.cat(c, 0) for c in zip(*parts))
    return color, alpha, depth, n_samples
