"""Op-by-op seam: the subset of ``nerfacc==0.3.1`` the reference imports
(``robust_e_nerf/external/utils.py:20``, ``external/vol_rendering.py:12-13``,
``models/nerf.py:3,98-102,200-204``, ``models/robust_e_nerf.py:214-218``) with the same names,
argument meaning and return shapes, served by the HIP kernels.  With this module registered as
``nerfacc`` the reference's own ``render_image`` / ``rendering`` / ``NeRF`` glue runs unmodified.

Differences that are deliberate and documented: tensors must live on the ROCm device; randomness
(`stratified` jitter, grid-cell jitter) is drawn with ``torch.rand`` on the device unless supplied.
"""
from __future__ import annotations

import enum
import math
from typing import Callable, List, Optional, Tuple, Union

import torch

from . import ops


class ContractionType(enum.Enum):
    """Same ordering as nerfacc (AABB, UN_BOUNDED_TANH, UN_BOUNDED_SPHERE)."""
    AABB = 0
    UN_BOUNDED_TANH = 1
    UN_BOUNDED_SPHERE = 2


# ------------------------------------------------------------------------------------- occupancy grid
class OccupancyGrid(torch.nn.Module):
    """nerfacc.OccupancyGrid(roi_aabb, resolution, contraction_type) with ``every_n_step``."""

    def __init__(self, roi_aabb, resolution: Union[int, List[int]] = 128,
                 contraction_type: ContractionType = ContractionType.AABB):
        super().__init__()
        if isinstance(resolution, int):
            resolution = [resolution] * 3
        assert len(resolution) == 3
        self._res = [int(r) for r in resolution]
        self.contraction_type = contraction_type
        self.num_cells = self._res[0] * self._res[1] * self._res[2]
        self.register_buffer("_roi_aabb", torch.as_tensor(roi_aabb, dtype=torch.float32).flatten())
        # nerfacc 0.3.1 keeps the resolution as a persistent int32 buffer: the key `nerf.occupancy_grid.resolution` of a
        # reference checkpoint must load with strict=True
        self.register_buffer("resolution", torch.tensor(self._res, dtype=torch.int32))
        self.register_buffer("_binary", torch.zeros(self._res, dtype=torch.bool))
        self.register_buffer("occs", torch.zeros(self.num_cells, dtype=torch.float32))
        self.register_buffer("_scratch", torch.zeros(4, dtype=torch.float32), persistent=False)

    @property
    def binary(self):
        return self._binary

    @property
    def roi_aabb(self):
        return self._roi_aabb

    @torch.no_grad()
    def _update(self, step, occ_eval_fn, occ_thre=0.01, ema_decay=0.95, warmup_steps=256,
                indices=None, jitter=None):
        dev = self.occs.device
        if indices is None:
            if step < warmup_steps:
                indices = torch.arange(self.num_cells, device=dev)
            else:
                n = self.num_cells // 4
                uni = torch.randint(self.num_cells, (n,), device=dev)
                occ_idx = torch.nonzero(self._binary.flatten())[:, 0]
                if n < occ_idx.numel():
                    occ_idx = occ_idx[torch.randint(occ_idx.numel(), (n,), device=dev)]
                indices = torch.cat([uni, occ_idx])
        if jitter is None:
            jitter = torch.rand(indices.shape[0], 3, device=dev)
        roi = self._roi_aabb.tolist()
        x, valid = ops.occgrid_cell_points(indices.contiguous(), jitter.contiguous(), roi, self._res,
                                           self.contraction_type.value)
        occ = occ_eval_fn(x).reshape(-1).to(torch.float32).contiguous()
        ops.occgrid_ema(self.occs, indices.contiguous(), valid, occ, None, 1.0, ema_decay)
        binary_u8 = torch.empty(self.num_cells, device=dev, dtype=torch.uint8)
        ops.occgrid_binarize(self.occs, occ_thre, binary_u8, self._scratch)
        self._binary = binary_u8.view(self._res).bool()

    def every_n_step(self, step: int, occ_eval_fn: Callable, occ_thre: float = 1e-2, ema_decay: float = 0.95,
                     warmup_steps: int = 256, n: int = 16):
        if not self.training:
            raise RuntimeError("You should only call this function only during training. "
                               "Please call _update() directly if you want to update the field during inference.")
        if step % n == 0 and self.training:
            self._update(step, occ_eval_fn, occ_thre, ema_decay, warmup_steps)


# ------------------------------------------------------------------------------------- ray marching
@torch.no_grad()
def ray_marching(rays_o, rays_d, t_min=None, t_max=None, scene_aabb=None, grid: Optional[OccupancyGrid] = None,
                 sigma_fn: Optional[Callable] = None, alpha_fn: Optional[Callable] = None,
                 near_plane: Optional[float] = None, far_plane: Optional[float] = None,
                 render_step_size: float = 1e-3, stratified: bool = False, cone_angle: float = 0.0,
                 early_stop_eps: float = 1e-4, alpha_thre: float = 0.0, jitter: Optional[torch.Tensor] = None):
    """-> (ray_indices (n,) int32, t_starts (n,1), t_ends (n,1)), exactly as the reference consumes it
    at external/utils.py:106-119 (it calls ``.long()`` on the indices itself)."""
    if not rays_o.is_cuda:
        raise NotImplementedError("Only support ROCm device tensors.")
    if alpha_fn is not None and sigma_fn is not None:
        raise ValueError("Only one of `alpha_fn` and `sigma_fn` should be provided.")
    if alpha_fn is not None:
        raise NotImplementedError("alpha_fn is never passed by the reference (SURVEY App. B.5)")
    rays_o, rays_d = rays_o.contiguous().float(), rays_d.contiguous().float()
    n_rays = rays_o.shape[0]
    render_step_size = float(render_step_size)
    if t_min is None or t_max is None:
        if scene_aabb is not None:
            t_min, t_max = ops.ray_aabb_intersect(rays_o, rays_d, scene_aabb.tolist(), near_plane, far_plane)
        else:
            t_min = torch.full((n_rays,), 0.0 if near_plane is None else float(near_plane), device=rays_o.device)
            t_max = torch.full((n_rays,), 1e10 if far_plane is None else float(far_plane), device=rays_o.device)
    if stratified and jitter is None:
        jitter = torch.rand(n_rays, device=rays_o.device)
    if grid is not None:
        roi, res, binary, ct = grid._roi_aabb.tolist(), grid._res, grid._binary.contiguous().view(-1), \
            grid.contraction_type.value
    else:
        roi, res, ct = [-1e10] * 3 + [1e10] * 3, [1, 1, 1], 0
        binary = torch.ones(1, dtype=torch.bool, device=rays_o.device)
    args = (rays_o, rays_d, t_min.contiguous(), t_max.contiguous(), jitter if stratified else None, roi, res, binary,
            ct, render_step_size, cone_angle, 0, 0)
    counts = ops.ray_march_count(*args)
    offsets, total = ops.exclusive_scan(counts)
    n0 = int(total.item())
    ri, ts, te = ops.ray_march_write(*args, offsets, n0)
    if sigma_fn is not None and n0 > 0:
        sigmas = sigma_fn(ts[:, None], te[:, None], ri.long())
        assert sigmas.shape == (n0, 1), "sigmas must have shape of (N, 1)! Got {}".format(sigmas.shape)
        keep, kept = ops.visibility(offsets, counts, sigmas.reshape(-1).contiguous().float(), ts, te,
                                    early_stop_eps, alpha_thre)
        new_offsets, total2 = ops.exclusive_scan(kept)
        ri, ts, te = ops.compact_samples(offsets, counts, new_offsets, keep, ts, te, int(total2.item()))
    return ri, ts[:, None], te[:, None]


# ------------------------------------------------------------------------------------- weights
class _RenderWeightFromDensity(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t_starts, t_ends, sigmas, offsets, counts):
        ts, te, sg = (t.reshape(-1).contiguous().float() for t in (t_starts, t_ends, sigmas))
        _, opac, _, w, T = ops.composite_fwd(offsets, counts, ts, te, sg, None, 1, None, save=True)
        ctx.save_for_backward(ts, te, sg, offsets, counts, w, T)
        ctx.sigmas_in = sigmas                                   # with its graph, for the differentiable backward
        return w[:, None]

    @staticmethod
    def backward(ctx, g_w):
        ts, te, sg, offsets, counts, w, T = ctx.saved_tensors
        if torch.is_grad_enabled() and (ctx.sigmas_in.requires_grad or g_w.requires_grad):
            # create_graph=True (the log-intensity-gradient loss differentiates this backward again,
            # utils/autograd.py:4-34): the same derivative written with differentiable torch scans
            return None, None, _weights_backward_torch(ts, te, ctx.sigmas_in, offsets, counts, g_w), None, None
        d_sig, _, _ = ops.composite_bwd(offsets, counts, ts, te, sg, None, 1, None, w, T, None, None,
                                        g_weights=g_w.reshape(-1).contiguous().float())
        return None, None, d_sig[:, None], None, None


def _weights_backward_torch(ts, te, sigmas, offsets, counts, g_w):
    """d(sum g_i w_i)/d sigma_k = dt_k (g_k T_k exp(-s_k) - sum_{i>k in the ray} g_i w_i), s = sigma dt, from float64
    cumulative sums over the packed stream (differentiable to any order w.r.t. sigmas and g_w)."""
    n = ts.shape[0]
    dt = (te - ts).double()
    s = sigmas.reshape(-1).double() * dt
    g = g_w.reshape(-1).double()
    ray = torch.repeat_interleave(torch.arange(counts.shape[0], device=ts.device), counts.long(), output_size=n)
    first = offsets.long()[ray]                                   # first sample of this sample's ray
    cs = torch.cumsum(s, 0)
    excl = cs - s                                                 # sum_{j < k, whole stream}
    T = torch.exp(-(excl - excl[first]))                          # transmittance in front of sample k
    w = T * (1.0 - torch.exp(-s))
    gw = g * w
    cg = torch.cumsum(gw, 0)
    last = first + counts.long()[ray] - 1
    suffix = cg[last] - cg                                        # sum_{i > k in the ray} g_i w_i
    return ((g * T * torch.exp(-s) - suffix) * dt).to(sigmas.dtype).reshape(sigmas.shape)


def render_weight_from_density(t_starts, t_ends, sigmas, *, packed_info=None, ray_indices=None, n_rays=None):
    """w_i = T_i (1 - exp(-sigma_i dt_i)); (n,1) in, (n,1) out; differentiable w.r.t. sigmas only
    (external/vol_rendering.py:36-37,89-95)."""
    assert ray_indices is not None and n_rays is not None, "the reference always passes ray_indices and n_rays"
    offsets, counts = ops.pack_info(ray_indices.to(torch.int32).contiguous(), int(n_rays))
    return _RenderWeightFromDensity.apply(t_starts, t_ends, sigmas, offsets, counts)


def render_weight_from_alpha(alphas, *, packed_info=None, ray_indices=None, n_rays=None):
    raise NotImplementedError("rgb_alpha_fn is never passed by the reference's render_image (SURVEY App. B.5)")


def accumulate_along_rays(weights, ray_indices, values=None, n_rays=None):
    """zeros(n_rays, D).index_add_(0, ray_indices, weights * values)  (values None -> weights).
    Plain torch scatter-add on (n, D<=3) tensors: bandwidth-trivial glue kept differentiable by autograd;
    the fused path (engine.Renderer / ren_composite_fwd) does all three accumulations in one kernel."""
    assert ray_indices.dim() == 1 and weights.dim() == 2
    src = weights if values is None else weights * values
    assert n_rays is not None
    out = torch.zeros(int(n_rays), src.shape[-1], device=src.device, dtype=src.dtype)
    return out.index_add(0, ray_indices.long(), src)
