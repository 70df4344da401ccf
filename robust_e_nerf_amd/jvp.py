"""Log-intensity-gradient supervision: render + d(render)/d(timestamp) in one forward-mode pass and
the reverse pass over the (value, tangent) pair.  Replaces the reference's
``autograd.gradient(log_intensity, ts, create_graph=True)`` second-order path
(robust_e_nerf/models/robust_e_nerf.py:383-409, utils/autograd.py:4-34).  See csrc/ren_jvp.hip.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from . import _lib, ops
from ._lib import check
from .ops import _f, _ptr, _stream


def trajectory_jvp(ts, tab_ts, tab_pos, tab_quat):
    B, dev = ts.shape[0], ts.device
    pos, dpos = torch.empty(B, 3, device=dev), torch.empty(B, 3, device=dev)
    rot, drot = torch.empty(B, 3, 3, device=dev), torch.empty(B, 3, 3, device=dev)
    check(_lib.load().ren_trajectory_jvp(_ptr(ts, torch.float64), B, _ptr(tab_ts, torch.int64), _ptr(tab_pos),
                                         _ptr(tab_quat), tab_ts.shape[0], _ptr(pos), _ptr(rot), _ptr(dpos),
                                         _ptr(drot), _stream()), "ren_trajectory_jvp")
    return pos, rot, dpos, drot


def raygen_jvp(Kinv, px, pos, rot, dpos, drot):
    B, dev = px.shape[0], px.device
    o, d, od, dd = (torch.empty(B, 3, device=dev) for _ in range(4))
    check(_lib.load().ren_raygen_jvp(_ptr(Kinv), _ptr(px), _ptr(pos), _ptr(rot), _ptr(dpos), _ptr(drot), B,
                                     _ptr(o), _ptr(d), _ptr(od), _ptr(dd), _stream()), "ren_raygen_jvp")
    return o, d, od, dd


def trajectory_jvp2(ts, tab_ts, tab_pos, tab_quat):
    """-> pos, rot, d pos/dt, d rot/dt, d2 rot/dt2 (d2 pos/dt2 = 0 inside a pose segment)."""
    B, dev = ts.shape[0], ts.device
    pos, dpos = torch.empty(B, 3, device=dev), torch.empty(B, 3, device=dev)
    rot, drot, ddrot = (torch.empty(B, 3, 3, device=dev) for _ in range(3))
    check(_lib.load().ren_trajectory_jvp2(_ptr(ts, torch.float64), B, _ptr(tab_ts, torch.int64), _ptr(tab_pos),
                                          _ptr(tab_quat), tab_ts.shape[0], _ptr(pos), _ptr(rot), _ptr(dpos),
                                          _ptr(drot), _ptr(ddrot), _stream()), "ren_trajectory_jvp2")
    return pos, rot, dpos, drot, ddrot


def raygen_jvp2(Kinv, px, pos, rot, dpos, drot, ddrot):
    B, dev = px.shape[0], px.device
    o, d, od, dd, ddd = (torch.empty(B, 3, device=dev) for _ in range(5))
    check(_lib.load().ren_raygen_jvp2(_ptr(Kinv), _ptr(px), _ptr(pos), _ptr(rot), _ptr(dpos), _ptr(drot),
                                      _ptr(ddrot), B, _ptr(o), _ptr(d), _ptr(od), _ptr(dd), _ptr(ddd), _stream()),
          "ren_raygen_jvp2")
    return o, d, od, dd, ddd


def render_forward2(r, o, d, od, dd, ddd, pk, bkgd):
    """Second-order forward render over an EXISTING sample stream `pk` (sample placement is not
    differentiated): -> colors, d colors/dt, d2 colors/dt2, each (R, C).  Forward only."""
    f, lib = r.field, _lib.load()
    n, R, dev = pk.n, o.shape[0], o.device
    if n == 0:
        z = torch.zeros(R, f.C, device=dev)
        return z + (bkgd if bkgd is not None else 0.0), z.clone(), z.clone()
    ri, ts, te = pk.ray_indices, pk.t_starts, pk.t_ends
    if hasattr(r, "_field_forward_jvp"):                   # arch mlp: value, d/dt, d2/dt2 streams through the dense layers
        rgb, rgbd, rgbdd, sg, sgd, sgdd = r._field_forward_jvp(o, d, od, dd, pk, ddd=ddd)
        colors, colords, colorsdd = (torch.empty(R, f.C, device=dev) for _ in range(3))
        check(lib.ren_composite_fwd_jvp2(_ptr(pk.offsets), _ptr(pk.counts), R, _ptr(ts), _ptr(te), _ptr(sg), _ptr(sgd),
                                         _ptr(sgdd), _ptr(rgb), _ptr(rgbd), _ptr(rgbdd), f.C, _ptr(bkgd), _ptr(colors),
                                         _ptr(colords), _ptr(colorsdd), _stream()), "ren_composite_fwd_jvp2")
        return colors, colords, colorsdd
    nb = ops.n_blocks32(n)
    feat, featd, featdd = (torch.empty(nb * 1024, device=dev) for _ in range(3))
    check(lib.ren_hashgrid_fwd_jvp2(ctypes.byref(f.grid), _ptr(f.table), ctypes.byref(r.scene), _ptr(o), _ptr(d),
                                    _ptr(od), _ptr(dd), _ptr(ddd), _ptr(ri), _ptr(ts), _ptr(te), n, _ptr(feat),
                                    _ptr(featd), _ptr(featdd), _ptr(pk.n_dev, torch.int64), _stream()), "ren_hashgrid_fwd_jvp2")
    rgb, rgbd, rgbdd = (torch.empty(n, f.C, device=dev) for _ in range(3))
    sg, sgd, sgdd = (torch.empty(n, device=dev) for _ in range(3))
    if r.cfg.mlp_kernels == "x":                            # bf16 matrix cores, in the step's precision mode
        check(lib.ren_mlp_fwd_jvp2_x(_ptr(f.mlp), f.C, r._act_code, r._xmode(), _ptr(feat), _ptr(featd), _ptr(featdd), ctypes.byref(r.scene),
                                     _ptr(o), _ptr(d), _ptr(od), _ptr(dd), _ptr(ddd), _ptr(ri), _ptr(ts), _ptr(te), n,
                                     _ptr(rgb), _ptr(rgbd), _ptr(rgbdd), _ptr(sg), _ptr(sgd), _ptr(sgdd), _ptr(pk.n_dev, torch.int64),
                                     _stream()), "ren_mlp_fwd_jvp2_x")
    else:
        check(lib.ren_mlp_fwd_jvp2(_ptr(r._mlp_params()), f.C, r._act_code, _ptr(feat), _ptr(featd), _ptr(featdd), ctypes.byref(r.scene),
                                   _ptr(o), _ptr(d), _ptr(od), _ptr(dd), _ptr(ddd), _ptr(ri), _ptr(ts), _ptr(te), n,
                                   _ptr(rgb), _ptr(rgbd), _ptr(rgbdd), _ptr(sg), _ptr(sgd), _ptr(sgdd), _stream()),
              "ren_mlp_fwd_jvp2")
    colors, colords, colorsdd = (torch.empty(R, f.C, device=dev) for _ in range(3))
    check(lib.ren_composite_fwd_jvp2(_ptr(pk.offsets), _ptr(pk.counts), R, _ptr(ts), _ptr(te), _ptr(sg), _ptr(sgd),
                                     _ptr(sgdd), _ptr(rgb), _ptr(rgbd), _ptr(rgbdd), f.C, _ptr(bkgd), _ptr(colors),
                                     _ptr(colords), _ptr(colorsdd), _stream()), "ren_composite_fwd_jvp2")
    return colors, colords, colorsdd


def render_forward(r, o, d, od, dd, jitter, bkgd, training: bool = True, pk=None, begun=None, device_counts: bool = False):
    """-> colors (R,C), colords (R,C) [d/dt], opacity (R,), ctx.  pk: the samples, when the caller has already placed them
    (engine.Trainer.grad_loss_forward_backward(early=True)); begun: Renderer.sample_begin() of these rays, already enqueued."""
    f, lib = r.field, _lib.load()
    if pk is None:
        pk = r.sample(o, d, jitter, training, begun=begun, device_counts=device_counts)
    n, R, dev = pk.n, o.shape[0], o.device               # (pk.n_dev: n is the capacity, the count is on the device)
    if n == 0:
        colors = torch.zeros(R, f.C, device=dev) + (bkgd if bkgd is not None else 0.0)
        return colors, torch.zeros(R, f.C, device=dev), torch.zeros(R, device=dev), dict(pk=pk, empty=True, bkgd=bkgd)
    ri, ts, te = pk.ray_indices, pk.t_starts, pk.t_ends
    if hasattr(r, "_field_forward_jvp"):                   # arch mlp: dense layers, value + tangent (vanilla.py)
        rgb, rgbd, sigma, sigmad, fctx = r._field_forward_jvp(o, d, od, dd, pk)
        feat = featd = base = based = None
    else:
        fctx = None
        nb = ops.n_blocks32(n)
        feat = torch.empty(nb * 1024, device=dev)
        featd = torch.empty(nb * 1024, device=dev)
        check(lib.ren_hashgrid_fwd_jvp(ctypes.byref(f.grid), _ptr(f.table), ctypes.byref(r.scene), _ptr(o), _ptr(d),
                                       _ptr(od), _ptr(dd), _ptr(ri), _ptr(ts), _ptr(te), n, _ptr(feat), _ptr(featd),
                                       _ptr(pk.n_dev, torch.int64), _stream()), "ren_hashgrid_fwd_jvp")
        rgb, rgbd = torch.empty(n, f.C, device=dev), torch.empty(n, f.C, device=dev)
        sigma, sigmad = torch.empty(n, device=dev), torch.empty(n, device=dev)
        base, based = torch.empty(nb * 512, device=dev), torch.empty(nb * 512, device=dev)
        if r.cfg.mlp_kernels == "x":                        # bf16 matrix cores: mode 6 (fp32 accuracy) / 1 (bf16 operands)
            check(lib.ren_mlp_fwd_jvp_x(_ptr(f.mlp), f.C, r._act_code, r._xmode(), _ptr(feat), _ptr(featd), ctypes.byref(r.scene), _ptr(o),
                                        _ptr(d), _ptr(dd), _ptr(ri), _ptr(ts), _ptr(te), n, _ptr(rgb), _ptr(rgbd), _ptr(sigma),
                                        _ptr(sigmad), _ptr(base), _ptr(based), _ptr(pk.n_dev, torch.int64), _stream()), "ren_mlp_fwd_jvp_x")
        else:
            check(lib.ren_mlp_fwd_jvp(_ptr(r._mlp_params()), f.C, r._act_code, _ptr(feat), _ptr(featd), ctypes.byref(r.scene), _ptr(o), _ptr(d),
                                      _ptr(dd), _ptr(ri), _ptr(ts), _ptr(te), n, _ptr(rgb), _ptr(rgbd), _ptr(sigma),
                                      _ptr(sigmad), _ptr(base), _ptr(based), _stream()), "ren_mlp_fwd_jvp")
    colors, colords = torch.empty(R, f.C, device=dev), torch.empty(R, f.C, device=dev)
    opac, opacd = torch.empty(R, device=dev), torch.empty(R, device=dev)
    w, T, eds = (torch.empty(n, device=dev) for _ in range(3))
    check(lib.ren_composite_fwd_jvp(_ptr(pk.offsets), _ptr(pk.counts), R, _ptr(ts), _ptr(te), _ptr(sigma),
                                    _ptr(sigmad), _ptr(rgb), _ptr(rgbd), f.C, _ptr(bkgd), _ptr(colors),
                                    _ptr(colords), _ptr(opac), _ptr(opacd), _ptr(w), _ptr(T), _ptr(eds), _stream()),
          "ren_composite_fwd_jvp")
    ctx = dict(pk=pk, o=o, d=d, od=od, dd=dd, feat=feat, featd=featd, rgb=rgb, rgbd=rgbd, sigma=sigma,
               sigmad=sigmad, base=base, based=based, w=w, T=T, eds=eds, opac=opac, opacd=opacd, bkgd=bkgd,
               empty=False, fctx=fctx)
    if fctx is not None:
        ctx["buffers"] = fctx["buffers"]                    # the value-only reverse pass (Renderer.backward) of arch mlp
    elif r.cfg.mlp_kernels == "x":
        # Renderer.backward on this context (l_diff with a trainable tau: the tangent is only needed forward) runs the
        # bf16-matrix-core backward in the forward's mode, recomputing the hidden activations from feat / base
        ctx["xmode"], ctx["acts"] = r._xmode(), None
    return colors, colords, opac, ctx


def render_backward(r, ctx, g_colors, g_colords, final: bool = False):
    """Accumulates into r.field.grad; returns d(bkgd) (C,) or None.  final: last backward pass of the step."""
    f, lib = r.field, _lib.load()
    if ctx["empty"]:
        if final and r.dp_early_slice() is not None:         # same collective sequence on every rank (Renderer.backward)
            r.dp_early()
        return g_colors.sum(0) if ctx.get("bkgd") is not None else None
    pk = ctx["pk"]
    n, R, dev = pk.n, ctx["o"].shape[0], ctx["o"].device
    nb = ops.n_blocks32(n)
    ri, ts, te = pk.ray_indices, pk.t_starts, pk.t_ends
    d_sig, d_sigd = torch.empty(n, device=dev), torch.empty(n, device=dev)
    d_rgb, d_rgbd = torch.empty(n, f.C, device=dev), torch.empty(n, f.C, device=dev)
    bk = ctx["bkgd"]
    d_bk = torch.empty(R, f.C, device=dev) if bk is not None else None
    check(lib.ren_composite_bwd_jvp(_ptr(pk.offsets), _ptr(pk.counts), R, _ptr(ts), _ptr(te), _ptr(ctx["sigma"]),
                                    _ptr(ctx["sigmad"]), _ptr(ctx["rgb"]), _ptr(ctx["rgbd"]), f.C, _ptr(bk),
                                    _ptr(ctx["w"]), _ptr(ctx["T"]), _ptr(ctx["eds"]), _ptr(ctx["opac"]),
                                    _ptr(ctx["opacd"]), _ptr(g_colors.contiguous()), _ptr(g_colords.contiguous()),
                                    _ptr(d_sig), _ptr(d_sigd), _ptr(d_rgb), _ptr(d_rgbd), _ptr(d_bk), _stream()),
          "ren_composite_bwd_jvp")
    if ctx.get("fctx") is not None:                        # arch mlp
        r._field_backward_jvp(ctx["fctx"], pk, ctx["rgb"], ctx["sigma"], d_rgb, d_rgbd, d_sig, d_sigd)
        return ops.column_sum(d_bk) if d_bk is not None else None
    scratch = torch.empty(nb * 5120, device=dev)
    dfeat, dfeatd = torch.empty(nb * 1024, device=dev), torch.empty(nb * 1024, device=dev)
    if r.cfg.mlp_kernels == "x":
        ws = torch.empty(int(lib.ren_mlp_bwd_jvp_x_workspace_floats(f.C)), device=dev)
        check(lib.ren_mlp_bwd_jvp_x(_ptr(f.mlp), f.C, r._act_code, r._xmode(), _ptr(ctx["feat"]), _ptr(ctx["featd"]), _ptr(ctx["base"]),
                                    _ptr(ctx["based"]), ctypes.byref(r.scene), _ptr(ctx["o"]), _ptr(ctx["d"]),
                                    _ptr(ctx["dd"]), _ptr(ri), _ptr(ts), _ptr(te), n, _ptr(ctx["rgb"]), _ptr(d_rgb),
                                    _ptr(d_rgbd), _ptr(d_sig), _ptr(d_sigd), _ptr(scratch), _ptr(dfeat), _ptr(dfeatd),
                                    _ptr(f.g_mlp), _ptr(ws), _ptr(pk.n_dev, torch.int64), _stream()), "ren_mlp_bwd_jvp_x")
    else:
        ws = torch.empty(int(lib.ren_mlp_bwd_jvp_workspace_floats(f.C)), device=dev)
        check(lib.ren_mlp_bwd_jvp(_ptr(f.mlp), f.C, r._act_code, _ptr(ctx["feat"]), _ptr(ctx["featd"]), _ptr(ctx["base"]),
                                  _ptr(ctx["based"]), ctypes.byref(r.scene), _ptr(ctx["o"]), _ptr(ctx["d"]),
                                  _ptr(ctx["dd"]), _ptr(ri), _ptr(ts), _ptr(te), n, _ptr(ctx["rgb"]), _ptr(d_rgb),
                                  _ptr(d_rgbd), _ptr(d_sig), _ptr(d_sigd), _ptr(scratch), _ptr(dfeat), _ptr(dfeatd),
                                  _ptr(f.g_mlp), _ptr(ws), _stream()), "ren_mlp_bwd_jvp")
    if r.cfg.binned_scatter:
        r._binned_workspace(n, dev)
        kw = dict(scene=r.scene, rays=(ctx["o"], ctx["d"]), samples=(ri, ts, te), n=n, layout=1,
                  tangent=(ctx["od"], ctx["dd"], dfeatd), n_dev=pk.n_dev)
        if final and r.dp_early_slice() is not None:          # see Renderer._field_backward
            lo_mask = (1 << r.cfg.dp_split_level) - 1
            ops.hashgrid_bwd_binned(f.grid, f.g_table, dfeat, r._bin_ws, level_mask=0xFFFF & ~lo_mask, **kw)
            r.dp_early()
            ops.hashgrid_bwd_binned(f.grid, f.g_table, dfeat, r._bin_ws, level_mask=lo_mask, **kw)
        else:
            ops.hashgrid_bwd_binned(f.grid, f.g_table, dfeat, r._bin_ws, **kw)
    else:
        check(lib.ren_hashgrid_bwd_jvp(ctypes.byref(f.grid), _ptr(f.g_table), ctypes.byref(r.scene), _ptr(ctx["o"]),
                                       _ptr(ctx["d"]), _ptr(ctx["od"]), _ptr(ctx["dd"]), _ptr(ri), _ptr(ts),
                                       _ptr(te), n, _ptr(dfeat), _ptr(dfeatd), _stream()), "ren_hashgrid_bwd_jvp")
    return ops.column_sum(d_bk) if d_bk is not None else None


def rate_epilogue(colors, colords, opac, chan, min_modeled_intensity: float, want_valid: bool, want_dlog: bool = True):
    """intensity, d intensity/dt, validity (None when every ray is valid) and d log I/dt of the events' channels: one launch"""
    R, C = colors.shape
    dev = colors.device
    inten, intend = torch.empty(R, device=dev), torch.empty(R, device=dev)
    valid = torch.empty(R, device=dev, dtype=torch.uint8) if want_valid else None
    dlog = torch.empty(R, device=dev) if want_dlog else None
    check(_lib.load().ren_rate_epilogue(_ptr(colors), _ptr(colords), _ptr(opac), _ptr(chan), C, R, _f(min_modeled_intensity),
                                        _ptr(inten), _ptr(intend), _ptr(valid), _ptr(dlog), _stream()), "ren_rate_epilogue")
    return inten, intend, valid, dlog


def tau_pose_grad(tau_grad, dts, g_a, x_a, g_b=None, x_b=None):
    """tau_grad[0] (f64, device) += sum (g_a x_a + g_b x_b) dts"""
    n = g_a.numel()
    check(_lib.load().ren_tau_pose_grad(_ptr(g_a, torch.float32), _ptr(x_a, torch.float32), _ptr(g_b, torch.float32),
                                        _ptr(x_b, torch.float32), _ptr(dts, torch.float64), n, _ptr(tau_grad, torch.float64),
                                        _stream()), "ren_tau_pose_grad")


def grad_loss_fwd(inten, intend, target, valid, err_fn: str):
    loss_sum = torch.empty(2, device=inten.device)
    check(_lib.load().ren_grad_loss_fwd(_ptr(inten), _ptr(intend), _ptr(target), _ptr(valid), inten.shape[0],
                                        ops.ERR_FN[err_fn], _ptr(loss_sum), _stream()), "ren_grad_loss_fwd")
    return loss_sum


def grad_loss_bwd(inten, intend, target, valid, err_fn: str, scale: float, loss_sum, scale_dev=None, want_loss: bool = False):
    """-> g_i, g_id (and the loss term, a device scalar, with want_loss); scale_dev: device double multiplying `scale`"""
    g_i, g_id = torch.empty_like(inten), torch.empty_like(intend)
    loss = torch.empty((), device=inten.device, dtype=torch.float32) if want_loss else None
    check(_lib.load().ren_grad_loss_bwd(_ptr(inten), _ptr(intend), _ptr(target), _ptr(valid), inten.shape[0],
                                        ops.ERR_FN[err_fn], _f(scale), _ptr(scale_dev, torch.float64), _ptr(loss_sum), _ptr(g_i),
                                        _ptr(g_id), _ptr(loss), _stream()), "ren_grad_loss_bwd")
    return (g_i, g_id, loss) if want_loss else (g_i, g_id)
