"""Oracle: one render / one training step of the hot path, end to end, on the CPU.

TEST INFRASTRUCTURE ONLY.  Composition follows
``robust_e_nerf/models/robust_e_nerf.py:301-517`` (training_step), ``:849-885``
(render_pixels), ``robust_e_nerf/models/nerf.py:230-286`` (NeRF.forward) and
``robust_e_nerf/external/utils.py:38-140`` (render_image, training = one chunk).
Also the ``cpu_baseline`` leg of bench.py.
"""
from __future__ import annotations

from dataclasses import dataclass, field as dc_field
from typing import Dict, Optional

import torch

from . import events, field, hashgrid, render, sampling, trajectory


@dataclass
class SceneCfg:
    """Subset of configs/train/*.yaml model.nerf.* that shapes the hot path."""
    aabb: tuple = (-1.5, -1.5, -1.5, 1.5, 1.5, 1.5)
    contraction_type: int = field.AABB
    occ_res: tuple = (128, 128, 128)
    near_plane: Optional[float] = None
    far_plane: Optional[float] = None
    render_step_size: float = 3 ** 0.5 * 3.0 / 1024        # robust_e_nerf.py:220-226 "auto"
    cone_angle: float = 0.0
    early_stop_eps: float = 1e-4
    alpha_thre: float = 0.0
    min_modeled_intensity: float = 1e-3
    opacity_eps: float = 1e-10
    bkgd_is_param: bool = True                              # alpha_over_white_bg
    sampler: str = "occgrid"                                # "occgrid" | "uniform"
    n_uniform: int = 64
    acts: Optional[dict] = None                             # activation alternatives (field.DEFAULT_ACTS keys)


def render_rays(o, d, p: Dict[str, torch.Tensor], spec, cfg: SceneCfg, *,
                binary: Optional[torch.Tensor], jitter: Optional[torch.Tensor],
                bkgd: Optional[torch.Tensor], training: bool = True):
    """NeRF.forward (nerf.py:230-286) -> radiance (R,C), opacity (R,), depth (R,), n, packed."""
    aabb = torch.tensor(cfg.aabb, dtype=torch.float32)
    n_rays = o.shape[0]

    def positions(ts, te, ri):
        return o[ri] + d[ri] * (ts + te) / 2.0             # external/utils.py:68-72

    arch_mlp = "mlp.base.hidden_layers.0.weight" in p       # `arch: mlp` parameters (oracle/vanilla.py; nerf.py:143-160)

    def sigma_fn(ts, te, ri):
        if arch_mlp:
            from . import vanilla
            return vanilla.forward(p, positions(ts, te, ri), None, aabb, cfg.contraction_type, density_only=True, acts=cfg.acts)
        return field.query_density(positions(ts, te, ri), p, spec, aabb, cfg.contraction_type, acts=cfg.acts)

    def rgb_sigma_fn(ts, te, ri):
        if arch_mlp:
            from . import vanilla
            return vanilla.forward(p, positions(ts, te, ri), d[ri], aabb, cfg.contraction_type, acts=cfg.acts)
        return field.field_forward(positions(ts, te, ri), d[ri], p, spec, aabb, cfg.contraction_type, acts=cfg.acts)

    scene_aabb = aabb if cfg.contraction_type == field.AABB else None   # nerf.py:248-251
    if cfg.sampler == "occgrid":
        ri, ts, te = sampling.ray_marching(
            o.detach(), d.detach(), scene_aabb=scene_aabb, grid_binary=binary, grid_roi=aabb,
            contraction_type=cfg.contraction_type, sigma_fn=sigma_fn,
            near_plane=cfg.near_plane, far_plane=cfg.far_plane,
            render_step_size=cfg.render_step_size, stratified=training, cone_angle=cfg.cone_angle,
            early_stop_eps=cfg.early_stop_eps, alpha_thre=cfg.alpha_thre, jitter=jitter)
    else:
        # the build's fixed-S sampler: S equal intervals over the ray/AABB slab, comb shifted
        # by u * delta when training (SURVEY 8(d) sampling mode (i)).
        t_min, t_max = sampling.ray_aabb_intersect(o.detach(), d.detach(), aabb)
        if cfg.near_plane is not None:
            t_min = t_min.clamp(min=cfg.near_plane)
        if cfg.far_plane is not None:
            t_max = t_max.clamp(max=cfg.far_plane)
        _, _, ri, ts, te = sampling.march(o.detach(), d.detach(), t_min, t_max, aabb, None,
                                          cfg.contraction_type, cfg.render_step_size, 0.0,
                                          mode=1, n_uniform=cfg.n_uniform,
                                          jitter=jitter if training else None)
        ts, te = ts[:, None], te[:, None]
    colors, opac, depth = render.rendering(ts, te, ri, n_rays, rgb_sigma_fn, render_bkgd=bkgd)
    opac = opac.squeeze(-1)
    depth = depth.squeeze(-1) / (opac + cfg.opacity_eps)    # nerf.py:279-282
    return colors, opac, depth, ts.shape[0], (ri, ts, te)


def render_pixels(Kinv, px, pos, R, p, spec, cfg: SceneCfg, channel_idx=None, **kw):
    """robust_e_nerf.py:849-885: intensity, opacity, depth*cos, n, is_valid.  channel_idx (Bayer sensor): the
    colour channel each event's pixel sees, `bayering` (:887-890) applied to the (B, 3) render."""
    o, d = trajectory.pixel_params_to_ray(Kinv, px, pos, R)
    colors, opac, depth, n, packed = render_rays(o, d, p, spec, cfg, **kw)
    intensity = colors + cfg.min_modeled_intensity                      # :867
    intensity = intensity.squeeze(-1) if channel_idx is None else intensity.gather(1, channel_idx.long()[:, None])[:, 0]
    is_valid = torch.ones_like(opac, dtype=torch.bool) if cfg.bkgd_is_param else opac > 0
    depth = depth * (d * R[..., 2]).sum(-1)
    return intensity, opac, depth, n, is_valid, packed


@dataclass
class EventBatch:
    position: torch.Tensor          # (B,2) f32
    start_ts: torch.Tensor          # (B,) i64
    end_ts: torch.Tensor            # (B,) i64
    num_pos: torch.Tensor           # (B,) i64
    num_neg: torch.Tensor           # (B,) i64
    u_ts_diff: torch.Tensor         # (B,) f64
    u_diff_start: torch.Tensor      # (B,) f64
    u_grad: torch.Tensor            # (B,) f64
    channel_idx: Optional[torch.Tensor] = None   # (B,) colour channel per event (Bayer sensor), else monochrome


def training_forward(batch: EventBatch, p, spec, cfg: SceneCfg, *, Kinv, tab_ts, tab_pos, tab_quat,
                     p2n_raw, neg_ct, tau_raw, tau_max, bkgd_raw, binary,
                     jitter_start, jitter_end, loss_cfg: Optional[dict] = None, jitter_grad=None,
                     tangent: Optional[str] = None):
    """training_step: log-intensity-difference loss (two renders) and, when loss_cfg["w_grad"] > 0,
    the log-intensity-gradient loss (a third render differentiated w.r.t. its timestamp with
    create_graph=True, robust_e_nerf.py:383-409).

    tangent: how d log I / dt of the third render is taken -- "reverse" = `torch.autograd.grad(..., create_graph=True)` as
    the reference (utils/autograd.py:4-34), "forward" = forward-mode AD of the same graph (what the HIP path computes with
    its tangent kernels; identical mathematics, and in fp32 identical to round-off -- tests/test_oracle_golden.py).  Default:
    reverse, except inside `field.bf16_linear()` where the placement of the bf16 roundings depends on the mode and the
    emulation follows the kernels.

    Returns (loss, aux) where aux holds every intermediate the parity tests compare."""
    if tangent is None:
        tangent = "forward" if field.BF16_LINEAR else "reverse"
    lc = dict(err_diff="mse", w_diff=1.0, pw_diff="mean_contrast_reciprocal_sq")
    lc.update(loss_cfg or {})
    w_grad = lc.get("w_grad", 0.0)
    c_p, c_n, mean_c = events.contrast_thresholds(p2n_raw, neg_ct)
    ev_diff = events.event_log_intensity_diff(batch.num_pos, batch.num_neg, c_p, c_n)
    tau = events.refractory_period(events.clamp_tau_raw(tau_raw, tau_max), tau_max)
    tsd = events.supervision_timestamps(batch.start_ts, batch.end_ts, batch.u_ts_diff,
                                        batch.u_diff_start, batch.u_grad, tau, want_grad=w_grad > 0)
    bkgd = torch.nn.functional.softplus(bkgd_raw) if cfg.bkgd_is_param else None
    out = {}
    grad_kw = {}
    if w_grad > 0:
        ts_g = tsd["grad_ts"]
        if not ts_g.requires_grad:
            ts_g = ts_g.detach().requires_grad_()               # robust_e_nerf.py:355
        if tangent == "forward":
            import torch.autograd.forward_ad as fwAD
            with fwAD.dual_level():
                pos, R = trajectory.linear_trajectory(fwAD.make_dual(ts_g, torch.ones_like(ts_g)), tab_ts, tab_pos, tab_quat)
                res = render_pixels(Kinv, batch.position, pos, R, p, spec, cfg, channel_idx=batch.channel_idx,
                                    binary=binary, jitter=jitter_grad, bkgd=bkgd, training=True)
                dual = fwAD.unpack_dual(res[0].log())
                dlog = dual.tangent if dual.tangent is not None else torch.zeros_like(dual.primal)
                out["grad"] = tuple(fwAD.unpack_dual(v).primal if torch.is_tensor(v) else v for v in res[:5]) + (res[5],)
            dlog = dlog.to(ts_g.dtype)
        else:
            pos, R = trajectory.linear_trajectory(ts_g, tab_ts, tab_pos, tab_quat)
            out["grad"] = render_pixels(Kinv, batch.position, pos, R, p, spec, cfg, channel_idx=batch.channel_idx,
                                        binary=binary, jitter=jitter_grad, bkgd=bkgd, training=True)
            log_g = out["grad"][0].log()
            (dlog,) = torch.autograd.grad(log_g, ts_g, torch.ones_like(log_g), create_graph=True)   # utils/autograd.py:4-34
        grad_kw = dict(pred_log_grad=dlog, grad_valid=out["grad"][4])
    for name, ts, jit in (("start", tsd["diff_start_ts"], jitter_start),
                          ("end", tsd["diff_end_ts"], jitter_end)):
        pos, R = trajectory.linear_trajectory(ts, tab_ts, tab_pos, tab_quat)
        out[name] = render_pixels(Kinv, batch.position, pos, R, p, spec, cfg, channel_idx=batch.channel_idx,
                                  binary=binary, jitter=jit, bkgd=bkgd, training=True)
    log_s, log_e = out["start"][0].log(), out["end"][0].log()
    pred = log_e - log_s
    valid = out["start"][4] | out["end"][4]
    loss, terms = events.event_loss(
        ev_diff, tsd["start_ts"], batch.end_ts, pred_log_diff=pred, ts_diff=tsd["ts_diff"],
        diff_valid=valid, mean_c=mean_c, **lc, **grad_kw)
    aux = dict(ts=tsd, grad=out.get("grad"), pred_log_grad=grad_kw.get("pred_log_grad"), intensity_start=out["start"][0], intensity_end=out["end"][0],
               opacity_start=out["start"][1], opacity_end=out["end"][1],
               n_start=out["start"][3], n_end=out["end"][3], pred_log_diff=pred, terms=terms,
               packed_start=out["start"][5], packed_end=out["end"][5])
    return loss, aux


def near_cell_face(xu: torch.Tensor, spec, k_ulp: float = 4.0) -> torch.Tensor:
    """(n, L) bool: in THIS restatement's own arithmetic (hashgrid.encode: pos = fmaf(scale, x, 0.5)) the sample sits within
    k_ulp units in the last place of pos of a cell face of level l in some dimension.  d feature / dt of a trilinear level is
    piecewise constant per cell, so a position that differs in its last bits (another summation order of o + d t) may take
    the neighbour cell's slope exactly there and nowhere else (models/robust_e_nerf.py:383-409 differentiates through it)."""
    out = []
    for lvl in range(spec.n_levels):
        pos = (xu.detach().double() * float(spec.scales[lvl]) + 0.5).float()
        w = (pos - torch.floor(pos)).double()
        ulp = torch.abs(torch.nextafter(pos, torch.full_like(pos, float("inf"))) - pos).double()
        out.append((torch.minimum(w, 1.0 - w) <= k_ulp * ulp).any(dim=-1))
    return torch.stack(out, dim=1)


def render_given(o, d, od, dd, ri, ts, te, p, spec, cfg: SceneCfg, bkgd, feat=None, featd=None, fused_position: bool = False):
    """The render of robust_e_nerf.py:383-409 and its time derivative on SOMEBODY ELSE'S rays and samples:
    rays (o, d) with their time derivatives (od, dd) and the packed sample stream (ri, ts, te) are inputs -- the HIP path's
    own, copied to the host -- so sample placement is identical by construction and what is compared is the field, the
    compositing and the forward-mode derivative (utils/autograd.py:4-34 in forward mode, as training_forward(tangent=
    "forward")).  feat / featd (n, L F): evaluate everything BEHIND the encoder on these features and their tangents instead
    of encoding (field.ENC_OVERRIDE).  fused_position: the sample position as ONE fused multiply-add o + d tm (product exact in
    float64, one rounding) instead of torch's multiply, then add -- the kernels' summation order (csrc/ren_common.h
    ren_sample_pos under -ffp-contract=fast); the two differ in the last bit of x, i.e. by up to ulp(scale x) = 2^-12 of a cell
    in the interpolation weights of the finest level.  -> dict(colors, colords, opacity, enc, encd, xu)"""
    import torch.autograd.forward_ad as fwAD
    aabb = torch.tensor(cfg.aabb, dtype=torch.float32)
    n_rays, ril = o.shape[0], ri.long()
    ts, te = ts.reshape(-1, 1), te.reshape(-1, 1)
    keep = {}
    with fwAD.dual_level():
        od_, dd_ = fwAD.make_dual(o, od), fwAD.make_dual(d, dd)

        def enc_fn(xu):
            if feat is not None:
                e = fwAD.make_dual(feat, featd)
            else:
                e = hashgrid.encode(xu, p["hash"], spec)
            u = fwAD.unpack_dual(e)
            keep["enc"], keep["encd"] = u.primal.detach(), (u.tangent if u.tangent is not None else torch.zeros_like(u.primal)).detach()
            keep["xu"] = fwAD.unpack_dual(xu).primal.detach()
            return e

        def rgb_sigma_fn(ts_, te_, ri_):
            if fused_position:
                tm = ((ts_ + te_) * 0.5).double()
                x = (od_[ri_].double() + dd_[ri_].double() * tm).float()
            else:
                x = od_[ri_] + dd_[ri_] * (ts_ + te_) / 2.0                # external/utils.py:68-72
            return field.field_forward(x, dd_[ri_], p, spec, aabb, cfg.contraction_type, acts=cfg.acts)

        field.ENC_OVERRIDE = enc_fn
        try:
            colors, opac, _ = render.rendering(ts, te, ri, n_rays, rgb_sigma_fn, render_bkgd=bkgd)
        finally:
            field.ENC_OVERRIDE = None
        c, oq = fwAD.unpack_dual(colors), fwAD.unpack_dual(opac)
        out = dict(colors=c.primal.detach(), colords=(c.tangent if c.tangent is not None else torch.zeros_like(c.primal)).detach(),
                   opacity=oq.primal.detach().squeeze(-1), **keep)
    return out
