"""Oracle: event-generation parameters, supervision timestamps, event loss.

TEST INFRASTRUCTURE ONLY.  Follows
``robust_e_nerf/models/event_generation_params.py:51-84,162-203`` (C_p/C_n ratio,
refractory period tau), ``robust_e_nerf/utils/modules.py:38-102`` (parametrisations,
MAPELoss), ``robust_e_nerf/models/robust_e_nerf.py:322-357`` (supervision timestamps),
``robust_e_nerf/loss_metric/loss.py:32-74`` (loss) and ``robust_e_nerf.py:470-486``
(parameter normalisation + weighted sum).  PINNED through tests/golden.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F_

MAPE_EPS = float(np.finfo(np.float64).eps)                 # modules.py:86
MIN_SIGMOID_GRAD = 1e-4                                    # event_generation_params.py:92


def softplus_right_inverse(y, beta=1.0, threshold=20.0):   # modules.py:47-55
    return torch.where(y * beta > threshold, y, torch.log(torch.exp(beta * y) - 1) / beta)


def contrast_thresholds(p2n_raw: torch.Tensor, neg_ct: torch.Tensor):
    """C_p = softplus(raw) * C_n ; mean C = (C_p + C_n)/2.  event_generation_params.py:51-70."""
    ratio = F_.softplus(p2n_raw, 1.0, 20.0)
    c_p = ratio * neg_ct
    return c_p, neg_ct, (c_p + neg_ct) / 2


def event_log_intensity_diff(num_pos, num_neg, c_p, c_n):  # :72-84
    return num_pos * c_p - num_neg * c_n


def clamp_tau_raw(tau_raw: torch.Tensor, tau_max: torch.Tensor) -> torch.Tensor:
    """:170-185 -- keep sigmoid'(raw/tau_max) >= 1e-4."""
    lim = torch.tensor(MIN_SIGMOID_GRAD, dtype=tau_raw.dtype).logit().abs()
    return tau_max * (tau_raw / tau_max).clamp(min=-lim, max=lim)


def refractory_period(tau_raw: torch.Tensor, tau_max: torch.Tensor) -> torch.Tensor:
    """tau = tau_max * sigmoid(raw / tau_max)  (float64).  modules.py:58-74."""
    return tau_max * torch.sigmoid(tau_raw / tau_max)


def supervision_timestamps(start_ts, end_ts, u_ts_diff, u_diff_start, u_grad, tau,
                           want_diff: bool = True, want_grad: bool = False):
    """robust_e_nerf.py:319-357.  start/end int64 ns, u_* float64, tau float64 scalar."""
    start = start_ts + tau                                  # int64 + f64 -> f64 (:201)
    out: Dict[str, torch.Tensor] = {"start_ts": start}
    if want_diff:
        ts_diff = (end_ts - start) * u_ts_diff
        d_start = torch.lerp(start, torch.max(end_ts - ts_diff, start), u_diff_start)
        d_end = torch.min(d_start + ts_diff, end_ts.to(d_start.dtype))
        out.update(ts_diff=ts_diff, diff_start_ts=d_start, diff_end_ts=d_end)
        lo, hi = d_start, d_end
    else:
        lo, hi = start, end_ts.to(start.dtype)
    if want_grad:
        out["grad_ts"] = torch.lerp(lo, hi, u_grad)
    return out


def mape(pred, target):                                    # modules.py:77-102
    return (pred - target).abs() / target.abs().clamp(min=MAPE_EPS)


ERR = {"l1": lambda a, b: (a - b).abs(), "mse": lambda a, b: (a - b) ** 2, "mape": mape}


def event_loss(
    ev_log_diff, start_ts_f64, end_ts, *,
    pred_log_diff=None, ts_diff=None, diff_valid=None,
    pred_log_grad=None, grad_valid=None,
    err_diff="mse", err_grad="mape", w_diff=1.0, w_grad=0.0,
    pw_diff="mean_contrast_reciprocal_sq", pw_grad=None, mean_c=None,
):
    """loss.py:32-74 + robust_e_nerf.py:470-486 -> (total, {name: normalised mean loss})."""
    target_grad = ev_log_diff / (end_ts - start_ts_f64)     # f64  (loss.py:39-42)
    inv_c = 1 / mean_c
    pw = {None: 1.0, "mean_contrast_reciprocal": inv_c, "mean_contrast_reciprocal_sq": inv_c ** 2}
    terms = {}
    if w_grad > 0:
        e = ERR[err_grad](pred_log_grad, target_grad)
        terms["log_intensity_grad"] = pw[pw_grad] * e[grad_valid].mean()
    if w_diff > 0:
        tgt = (ts_diff * target_grad).to(pred_log_diff.dtype)
        e = ERR[err_diff](pred_log_diff, tgt)
        terms["log_intensity_diff"] = pw[pw_diff] * e[diff_valid].mean()
    weights = {"log_intensity_grad": w_grad, "log_intensity_diff": w_diff}
    total = sum(terms[k] * weights[k] for k in terms)
    return total, terms
