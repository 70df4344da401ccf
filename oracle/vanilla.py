"""Oracle: vanilla-NeRF radiance field (`arch: mlp`): frequency positional encoding + 8x256 MLP.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Follows ``robust_e_nerf/external/mlp.py``: ``MLP.forward`` (:99-113, skip-concat AFTER the activation of
layer ``skip_layer``), ``NerfMLP`` (:126-205: base -> sigma layer | bottleneck -> [bottleneck, view
encoding] -> 128 -> C), ``SinusoidalEncoder`` (:208-243: ``[x, sin(2^k x) k<deg, sin(2^k x + pi/2) k<deg]``,
scale-major then dimension) and ``VanillaNeRFRadianceField`` (:246-358: contraction, selector,
``x -> 2 pi (x - 1/2)``, ``dirs -> pi dirs``), with the activation table of ``models/nerf.py:8-29`` and the
defaults of ``configs/train/synthetic.yaml:85-96``.  PINNED against the reference's own Python via
``tests/golden/field_mlp_*.npz``.
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import numpy as np
import torch

from . import field as ofield
from . import hashgrid

POS_DEG, VIEW_DEG = 10, 4
POS_DIM, VIEW_DIM = 3 + 3 * 2 * POS_DEG, 3 + 3 * 2 * VIEW_DEG            # 63, 27
DEPTH, WIDTH, SKIP, WIDTH_COND = 8, 256, 4, 128


def layer_shapes(C: int = 1) -> List[Tuple[str, int, int]]:
    """(name, out_features, in_features) in parameter-block order; names = reference state-dict stems."""
    out = []
    fin = POS_DIM
    for i in range(DEPTH):
        out.append((f"mlp.base.hidden_layers.{i}", WIDTH, fin))
        fin = WIDTH + POS_DIM if (i % SKIP == 0 and i > 0) else WIDTH
    out.append(("mlp.sigma_layer.output_layer", 1, WIDTH))
    out.append(("mlp.bottleneck_layer.output_layer", WIDTH, WIDTH))
    out.append(("mlp.rgb_layer.hidden_layers.0", WIDTH_COND, WIDTH + VIEW_DIM))
    out.append(("mlp.rgb_layer.output_layer", C, WIDTH_COND))
    return out


def n_params(C: int = 1) -> int:
    return sum(o * i + o for _, o, i in layer_shapes(C))


def init_params(seed: int, C: int = 1, gain: float = 1.0) -> Dict[str, torch.Tensor]:
    """Portable deterministic parameters: U(-b, b), b = gain / sqrt(fan_in) (torch nn.Linear default
    has gain 1), from the counter-based generator used for the hash-table fixtures."""
    u = hashgrid.mix32_uniform(n_params(C), seed)
    p, off = {}, 0
    for name, o, i in layer_shapes(C):
        b = np.float32(gain / math.sqrt(i))
        p[name + ".weight"] = torch.from_numpy((u[off: off + o * i] * np.float32(2) - np.float32(1)) * b).view(o, i).clone()
        off += o * i
        p[name + ".bias"] = torch.from_numpy((u[off: off + o] * np.float32(2) - np.float32(1)) * b).clone()
        off += o
    return p


def sinusoidal(x: torch.Tensor, deg: int) -> torch.Tensor:            # mlp.py:229-243
    scales = torch.tensor([2.0 ** k for k in range(deg)], dtype=x.dtype)
    xb = (x[..., None, :] * scales[:, None]).reshape(*x.shape[:-1], deg * x.shape[-1])
    return torch.cat([x, torch.sin(torch.cat([xb, xb + 0.5 * math.pi], dim=-1))], dim=-1)


def forward(p: Dict[str, torch.Tensor], x: torch.Tensor, dirs, aabb: torch.Tensor, contraction_type: int,
            density_only: bool = False, acts=None):
    """-> (rgb (n, C), sigma (n, 1)) or sigma only.  mlp.py:321-358.  acts: activation alternatives (models/nerf.py:8-29) as
    dict(base_hidden = the one hidden activation of this architecture, density, radiance); absent keys: the shipped set."""
    a = dict(ofield.DEFAULT_ACTS, **(acts or {}))
    u = ofield.contract(x, aabb, contraction_type)
    selector = ((u > 0.0) & (u < 1.0)).all(dim=-1)
    enc = sinusoidal(2 * math.pi * (u - 0.5), POS_DEG)
    h = enc
    for i in range(DEPTH):                                             # mlp.py:99-113
        h = torch.nn.functional.linear(h, p[f"mlp.base.hidden_layers.{i}.weight"], p[f"mlp.base.hidden_layers.{i}.bias"])
        h = ofield._hidden(h, a["base_hidden"])
        if i % SKIP == 0 and i > 0:
            h = torch.cat([h, enc], dim=-1)
    raw_sigma = torch.nn.functional.linear(h, p["mlp.sigma_layer.output_layer.weight"], p["mlp.sigma_layer.output_layer.bias"])
    sigma = ofield._density(raw_sigma, a["density"]) * selector[..., None]
    if density_only:
        return sigma
    bott = torch.nn.functional.linear(h, p["mlp.bottleneck_layer.output_layer.weight"], p["mlp.bottleneck_layer.output_layer.bias"])
    cond = sinusoidal(dirs * math.pi, VIEW_DEG)
    r = torch.cat([bott, cond], dim=-1)
    r = ofield._hidden(torch.nn.functional.linear(r, p["mlp.rgb_layer.hidden_layers.0.weight"], p["mlp.rgb_layer.hidden_layers.0.bias"]), a["base_hidden"])
    raw_rgb = torch.nn.functional.linear(r, p["mlp.rgb_layer.output_layer.weight"], p["mlp.rgb_layer.output_layer.bias"])
    return ofield._radiance(raw_rgb, a["radiance"]), sigma


def weight_norm_params(p):
    """`weight_norm: true` (external/mlp.py:303-319: torch.nn.utils.weight_norm on every Linear, dim 0): a dict holding
    "<layer>.weight_g" (rows, 1) + "<layer>.weight_v" in place of "<layer>.weight" -> the dict forward() takes, with
    W = g v / ||v||_row (differentiable).  Pinned to the reference by tests/golden/field_mlp_wn.npz."""
    q = {k: v for k, v in p.items() if not (k.endswith(".weight_g") or k.endswith(".weight_v"))}
    for k, v in p.items():
        if k.endswith(".weight_v"):
            g = p[k[:-1] + "g"]
            q[k[: -len("_v")]] = v * (g.reshape(-1, 1) / v.norm(dim=1, keepdim=True))
    return q
