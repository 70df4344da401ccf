"""Inference render + image metrics: the step after the hot path.

Mirrors ``RobustENeRF.evaluation_step`` / ``render_pixels`` (robust_e_nerf/models/robust_e_nerf.py:
533-571, 849-885): one camera pose, a full pixel grid, rays rendered in chunks of
``test_chunk_size`` without jitter (external/utils.py:99-105,115), then the affine alignment in
log space and PSNR of ``evaluation_epoch_end`` (:634-677) / ``Metric.compute`` (loss_metric/metric.py:60-72).
The render runs on the HIP kernels; alignment / PSNR are a handful of reductions on (H*W,) tensors.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import ops
from .engine import NGPField, Renderer


def pixel_grid(height: int, width: int, device) -> torch.Tensor:
    """(H, W, 2) pixel positions (x, y), float32 — robust_e_nerf.py:110-117."""
    xs, ys = torch.meshgrid(torch.arange(width, device=device), torch.arange(height, device=device), indexing="xy")
    return torch.stack([xs, ys], dim=2).to(torch.float32)


@torch.no_grad()
def render_pixels(r: Renderer, Kinv: torch.Tensor, px: torch.Tensor, pos: torch.Tensor, rot: torch.Tensor,
                  bkgd: Optional[torch.Tensor] = None, training: bool = False, jitter=None):
    """robust_e_nerf.py:849-885 for (N,2) pixels with per-pixel poses (N,3), (N,3,3):
    -> intensity (N,) [monochrome] or (N, 3) [Bayer sensor, radiance_dim 3], opacity, depth (z-depth), samples used,
    is_valid."""
    o, d = ops.raygen(Kinv, px.contiguous(), pos.contiguous(), rot.contiguous())
    colors, opac, depth, ctx = r.forward(o, d, jitter, bkgd, training=training, save=False)
    intensity = (colors[:, 0] if colors.shape[1] == 1 else colors) + r.cfg.min_modeled_intensity
    is_valid = torch.ones_like(opac, dtype=torch.bool) if bkgd is not None else opac > 0
    depth = depth / (opac + r.cfg.opacity_eps)                         # nerf.py:279-282
    depth = depth * (d * rot[:, :, 2]).sum(-1)                        # ray distance -> z depth (:873-884)
    return intensity, opac, depth, ctx["pk"].n, is_valid


@torch.no_grad()
def render_image(r: Renderer, Kinv: torch.Tensor, cam_pos: torch.Tensor, cam_rot: torch.Tensor, height: int,
                 width: int, bkgd: Optional[torch.Tensor] = None, chunk: Optional[int] = None,
                 rows: Optional[Tuple[int, int]] = None):
    """evaluation_step: (H, W) predicted intensity ((3, H, W) for radiance_dim 3), opacity, depth for one pose.  ``chunk`` is the reference's
    ``test_chunk_size`` (16 384 there, to fit a 2080 Ti); with 288 GB a 640x480 image is one chunk, which is 4x
    faster than 19 chunks (2.6 vs 10.3 ms, tools/render_bench.py).  The result does not depend on it.  Default: one
    chunk for arch ngp (~200 B/sample of temporaries), 65 536 rays for arch mlp on the fused field, the reference's 16 384
    rays for arch mlp on the per-layer kernels (8 x 256 hidden activations are ~10 KB/sample)."""
    if chunk is None:
        if isinstance(r.field, NGPField):
            chunk = 1 << 20
        else:       # arch mlp: the fused field keeps nothing per sample in inference (~0.5 KB of encodings / outputs)
            chunk = 65536 if getattr(r, "fused_field", False) and r._dense_mode() != 0 else 16384
    dev = Kinv.device
    px = pixel_grid(height, width, dev)
    if rows is not None:                                               # a band of image rows (render_image_sharded)
        px = px[rows[0]: rows[1]]
        height = rows[1] - rows[0]
    px = px.reshape(-1, 2)
    n = px.shape[0]
    C = r.field.C
    out_i = torch.empty((n,) if C == 1 else (n, C), device=dev)
    out_o = torch.empty(n, device=dev)
    out_d = torch.empty(n, device=dev)
    for s in range(0, n, chunk):                                       # external/utils.py:99-105
        e = min(s + chunk, n)
        pos = cam_pos.reshape(1, 3).expand(e - s, 3).contiguous()
        rot = cam_rot.reshape(1, 3, 3).expand(e - s, 3, 3).contiguous()
        i, o, d, _, _ = render_pixels(r, Kinv, px[s:e], pos, rot, bkgd)
        out_i[s:e], out_o[s:e], out_d[s:e] = i, o, d
    img = out_i.view(height, width) if C == 1 else out_i.view(height, width, C).permute(2, 0, 1).contiguous()
    return img, out_o.view(height, width), out_d.view(height, width)


# ---- data-parallel evaluation: collective C3 of SURVEY 2.3 (`self.all_gather(outputs)`, robust_e_nerf.py:591) ------------
def view_shard(n_views: int, rank: int, world: int):
    """indices of the evaluation views this rank renders: torch's DistributedSampler(shuffle=False) as Lightning's DDP
    installs it on the reference's val / test loaders (scripts/run.py:81-93) -- the index list is padded to a multiple of
    the world size by wrapping around, rank r takes r, r + world, ..."""
    if n_views == 0:
        return []
    per = -(-n_views // world)
    idx = list(range(n_views))
    while len(idx) < per * world:                                      # DistributedSampler: repeat from the start
        idx += idx[: per * world - len(idx)]
    return idx[rank: per * world: world]


def gather_views(local: torch.Tensor, n_views: int, rank: int, world: int, group=None) -> torch.Tensor:
    """`local`: (len(view_shard(...)), ...) outputs of this rank's views -> (n_views, ...) on every rank, in view order
    (all_gather of equal-sized blocks; the wrap-around duplicates are dropped)."""
    if world == 1:
        return local
    import torch.distributed as dist
    blocks = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(blocks, local.contiguous(), group=group)
    out = torch.empty((n_views,) + tuple(local.shape[1:]), device=local.device, dtype=local.dtype)
    seen = set()
    for r_ in range(world):
        for j, v in enumerate(view_shard(n_views, r_, world)):
            if v not in seen:
                out[v] = blocks[r_][j]
                seen.add(v)
    return out


@torch.no_grad()
def render_image_sharded(r: Renderer, Kinv, cam_pos, cam_rot, height: int, width: int, bkgd=None, *, rank: int, world: int,
                         group=None, chunk: Optional[int] = None):
    """ONE image rendered by all ranks (BASELINE configs[4]: a 640x480 novel view on 8 GPUs): rank r renders a band of
    ceil(H / world) rows, the bands are all-gathered (3 x H x W floats over xGMI).  Same result as render_image."""
    if world == 1:
        return render_image(r, Kinv, cam_pos, cam_rot, height, width, bkgd, chunk)
    import torch.distributed as dist
    per = -(-height // world)
    lo, hi = min(rank * per, height), min((rank + 1) * per, height)
    C = r.field.C
    band = torch.zeros(C + 2, per, width, device=Kinv.device)          # intensity (C) | opacity | depth
    if hi > lo:
        img, opac, depth = render_image(r, Kinv, cam_pos, cam_rot, height, width, bkgd, chunk, rows=(lo, hi))
        band[:C, : hi - lo] = img if C > 1 else img[None]
        band[C, : hi - lo], band[C + 1, : hi - lo] = opac, depth
    bands = [torch.empty_like(band) for _ in range(world)]
    dist.all_gather(bands, band, group=group)
    full = torch.cat(bands, dim=1)[:, :height]
    return (full[0] if C == 1 else full[:C].contiguous()), full[C], full[C + 1]


def affine_align_log(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """Least-squares a, b with a*log(pred)+b ~= log(target), float64, for ONE image (or per channel of a (3, H, W) one);
    returns exp(a*log(pred)+b) in target's dtype.  The reference's epoch metric uses one fit over all views:
    ``align_and_score``; this per-image form is for displaying a single novel view (scripts/render.py)."""
    a, b = solve_affine(log_fit_sums(pred[None], target[None]))
    return apply_affine(pred, a, b).to(target.dtype)


def _channel_view(x: torch.Tensor) -> torch.Tensor:
    """(V, H, W) or (V, N) -> (V, 1, pixels);  (V, 3, H, W) -> (V, 3, pixels)"""
    return x.reshape(x.shape[0], x.shape[1], -1) if x.dim() == 4 else x.reshape(x.shape[0], 1, -1)


def log_fit_sums(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """Per-channel sums of the normal equations of  a*log(pred) + b ~= log(target)  over the given views, float64:
    (C, 5) = [sum x^2, sum x, count, sum x*y, sum y] with x = log pred, y = log target.  Additive over views and ranks,
    which is what lets every rank contribute its shard of the reference's single fit over the gathered views
    (robust_e_nerf.py:640-670) with a 5C-double all-reduce instead of gathering the images."""
    x = _channel_view(pred).log().to(torch.float64)
    y = _channel_view(target).log().to(torch.float64)
    cnt = torch.full(x.shape[1:2], float(x.shape[0] * x.shape[2]), dtype=torch.float64, device=x.device)
    return torch.stack([(x * x).sum((0, 2)), x.sum((0, 2)), cnt, (x * y).sum((0, 2)), y.sum((0, 2))], dim=1)


def solve_affine(sums: torch.Tensor):
    """(C, 5) sums -> scale a (C,), offset b (C,) of the least-squares fit (the solution ``torch.linalg.lstsq`` returns at
    robust_e_nerf.py:666-669), solved in centred form for conditioning."""
    sxx, sx, n, sxy, sy = sums.unbind(1)
    mx, my = sx / n, sy / n
    a = (sxy - n * mx * my) / (sxx - n * mx * mx)
    return a, my - a * mx


def apply_affine(pred: torch.Tensor, a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """exp(a log(pred) + b), per channel; float64 product rounded to float32 before the exp as the reference (:670-683)"""
    if a.numel() == 1:                                                 # monochrome (H, W)
        shape = (1,) * pred.dim()
    else:                                                              # (3, H, W) of a Bayer sensor
        assert pred.dim() == 3 and pred.shape[0] == a.numel()
        shape = (-1, 1, 1)
    lp = pred.log().to(torch.float64) * a.to(pred.device).view(shape) + b.to(pred.device).view(shape)
    return lp.to(torch.float32).exp()


def psnr(pred: torch.Tensor, target: torch.Tensor, data_range: float) -> float:
    """10 log10(range^2 / MSE) — torchmetrics.functional.psnr as used at metric.py:68-72."""
    mse = ((pred.double() - target.double()) ** 2).mean()
    return float(10.0 * torch.log10(torch.tensor(data_range, dtype=torch.float64) ** 2 / mse))


def l1(pred: torch.Tensor, target: torch.Tensor) -> float:
    return float((pred.double() - target.double()).abs().mean())


def align_and_score(pred: torch.Tensor, target: torch.Tensor, data_range: float, sums: Optional[torch.Tensor] = None):
    """evaluation_epoch_end's metric part (robust_e_nerf.py:634-696): `pred`, `target` (V, H, W) or (V, 3, H, W).
    ONE scale / offset per channel fitted over all views (``sums``: the fit's sums when the views of other ranks take part
    in it, default: these views only), then L1 and PSNR of every aligned view.  -> (V, 2) float64 [l1, psnr], (a, b)"""
    a, b = solve_affine(log_fit_sums(pred, target) if sums is None else sums)
    out = torch.zeros(pred.shape[0], 2, dtype=torch.float64)
    for v in range(pred.shape[0]):
        al = apply_affine(pred[v], a, b)
        out[v, 0], out[v, 1] = l1(al, target[v]), psnr(al, target[v], data_range)
    return out, (a, b)


@torch.no_grad()
def evaluate_posed_images(r: Renderer, posed: dict, bkgd: Optional[torch.Tensor] = None, rank: int = 0, world: int = 1,
                          group=None, chunk: Optional[int] = None, limit: Optional[int] = None):
    """validation / test epoch of the reference (models/robust_e_nerf.py:519-696) over data.load_posed_images(...): every
    view is rendered at its pose (views sharded over the ranks like DDP's DistributedSampler), ALL views are aligned to
    their targets by ONE affine fit in log space per channel (:634-677: the reference flattens batch x H x W before its
    lstsq) -- each rank adds the normal-equation sums of its views, a 5C-double all-reduce replaces the image gather (C3)
    -- and every view is then scored with L1 / PSNR over the target's pixel-value range (loss_metric/metric.py:60-72).
    The wrap-around duplicates DistributedSampler pads with are left out of the fit and of the means (a deviation from the
    reference under data parallelism when n_views % world != 0: its all-gather keeps them in both; documented, README).
    -> dict(l1, psnr: means over the views; per_view: (V, 2) tensor; scale, offset: the fit)"""
    dev = r.field.flat.device
    n = len(posed["sample_id"]) if limit is None else min(limit, len(posed["sample_id"]))
    if n == 0:                                             # no view at all (limit 0): nothing to fit (0 / 0 otherwise), nothing to score
        return dict(l1=float("nan"), psnr=float("nan"), per_view=torch.zeros(0, 2), n_views=0, scale=None, offset=None)
    Kinv = torch.linalg.inv(posed["intrinsics"].double()).float().contiguous().to(dev).contiguous()
    H, W = posed["img"].shape[-2:]
    rng = posed["max_normalized_pixel_value"] - posed["min_normalized_pixel_value"]
    mine = view_shard(n, rank, world)
    fresh = [j * world + rank < n for j in range(len(mine))]          # False: a padded repeat of some other rank's view
    C = 1 if posed["img"].dim() == 3 else posed["img"].shape[1]
    preds, tgts = [], []
    for v in mine:
        tgt = posed["img"][v].to(dev)
        img, _, _ = render_image(r, Kinv, posed["T_wc_position"][v].to(dev), posed["T_wc_orientation"][v].to(dev).contiguous(),
                                 H, W, bkgd, chunk)
        if img.shape != tgt.shape:
            raise ValueError(f"view {posed['sample_id'][v]}: prediction {tuple(img.shape)} vs target {tuple(tgt.shape)}")
        preds.append(img.clamp_min(1e-12))
        tgts.append(tgt)
    sums = torch.zeros(C, 5, dtype=torch.float64, device=dev)
    own = [j for j, f in enumerate(fresh) if f]
    if own:
        sums += log_fit_sums(torch.stack([preds[j] for j in own]), torch.stack([tgts[j] for j in own]))
    if world > 1:
        import torch.distributed as dist
        dist.all_reduce(sums, group=group)
    local = torch.zeros(len(mine), 2, device=dev)
    a = b = None
    if mine:
        sc, (a, b) = align_and_score(torch.stack(preds), torch.stack(tgts), rng, sums)
        local.copy_(sc)
    else:
        a, b = solve_affine(sums)
    per_view = gather_views(local, n, rank, world, group)
    return dict(l1=float(per_view[:, 0].mean()), psnr=float(per_view[:, 1].mean()), per_view=per_view.cpu(), n_views=n,
                scale=a.cpu(), offset=b.cpu())
