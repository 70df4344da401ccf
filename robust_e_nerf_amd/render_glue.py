"""Reference-shaped render glue over the op-by-op seam (nerfacc_api / field): the build's own
counterpart of ``render_image`` (robust_e_nerf/external/utils.py:38-140) and ``rendering``
(external/vol_rendering.py:16-128), used by the tests to exercise the seam on the GPU box (where
/root/reference does not exist).  With ``nerfacc_api`` registered as ``nerfacc`` the reference's own
two files do the same thing unmodified."""
from __future__ import annotations

from typing import Optional

import torch

from . import nerfacc_api as nerfacc


def rendering(t_starts, t_ends, ray_indices, n_rays: int, rgb_sigma_fn, render_bkgd: Optional[torch.Tensor] = None):
    rgbs, sigmas = rgb_sigma_fn(t_starts, t_ends, ray_indices.long())
    if rgbs.shape[-1] not in (1, 3):
        raise AssertionError("rgbs must have 1 or 3 channels, got {}".format(rgbs.shape))
    if sigmas.shape != t_starts.shape:
        raise AssertionError("sigmas must have shape of (N, 1)! Got {}".format(sigmas.shape))
    w = nerfacc.render_weight_from_density(t_starts, t_ends, sigmas, ray_indices=ray_indices, n_rays=n_rays)
    colors = nerfacc.accumulate_along_rays(w, ray_indices, values=rgbs, n_rays=n_rays)
    opac = nerfacc.accumulate_along_rays(w, ray_indices, values=None, n_rays=n_rays)
    depth = nerfacc.accumulate_along_rays(w, ray_indices, values=(t_starts + t_ends) / 2.0, n_rays=n_rays)
    if render_bkgd is not None:
        colors = colors + render_bkgd * (1.0 - opac)
    return colors, opac, depth


def render_image(radiance_field, occupancy_grid, rays_o, rays_d, scene_aabb, near_plane=None, far_plane=None,
                 render_step_size: float = 1e-3, render_bkgd=None, cone_angle: float = 0.0,
                 early_stop_eps: float = 1e-4, alpha_thre: float = 0.0, test_chunk_size: int = 8192, jitter=None):
    """-> colors (R,C), opacities (R,1), depths (R,1), number of rendered samples."""
    n_rays = rays_o.shape[0]
    training = radiance_field.training
    chunk = n_rays if training else test_chunk_size
    outs, n_total = [], 0
    for s in range(0, n_rays, chunk):
        o, d = rays_o[s:s + chunk], rays_d[s:s + chunk]

        def pos(ts, te, ri):
            return o[ri] + d[ri] * (ts + te) / 2.0

        ri, ts, te = nerfacc.ray_marching(
            o, d, scene_aabb=scene_aabb, grid=occupancy_grid,
            sigma_fn=lambda a, b, i: radiance_field.query_density(pos(a, b, i)),
            near_plane=near_plane, far_plane=far_plane, render_step_size=render_step_size, stratified=training,
            cone_angle=cone_angle, early_stop_eps=early_stop_eps, alpha_thre=alpha_thre,
            jitter=None if jitter is None else jitter[s:s + chunk])
        c, a, z = rendering(ts, te, ri, o.shape[0], lambda a_, b_, i: radiance_field(pos(a_, b_, i), d[i]),
                            render_bkgd=render_bkgd)
        outs.append((c, a, z))
        n_total += ts.shape[0]
    colors, opac, depth = (torch.cat(x, 0) for x in zip(*outs))
    return colors, opac, depth, n_total
