"""Oracle: Instant-NGP radiance field (contraction, hash grid, SH, two MLPs).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Follows ``robust_e_nerf/external/ngp.py:45-106,230-280`` (field),
``robust_e_nerf/external/mlp.py:99-113`` (MLP forward),
``robust_e_nerf/external/sh_encoder.py:28-93`` (real SH, tcnn sign convention) and the
activation tables of ``robust_e_nerf/models/nerf.py:8-29``.  PINNED against the
reference's own Python via ``tests/golden`` (hash grid excepted -- see hashgrid.py).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F_

from . import hashgrid

AABB, UN_BOUNDED_TANH, UN_BOUNDED_SPHERE = 0, 1, 2       # nerfacc.ContractionType order


# ----------------------------------------------------------------------------- activations
class _TruncExp(torch.autograd.Function):
    """exp with backward g*exp(clamp(x, max=15)) -- ngp.py:45-61."""

    @staticmethod
    def forward(x):
        return torch.exp(x)

    @staticmethod
    def setup_context(ctx, inputs, output):
        ctx.save_for_backward(inputs[0])
        ctx.save_for_forward(inputs[0])

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * torch.exp(torch.clamp(x, max=15))

    @staticmethod
    def jvp(ctx, xd):                                       # forward mode (training_forward(tangent="forward")): same clamp
        (x,) = ctx.saved_tensors
        return xd * torch.exp(torch.clamp(x, max=15))


def shifted_trunc_exp(x, shift=1.0):                       # ngp.py:64-65
    return _TruncExp.apply(x - shift)


def softplus(x, beta: float, threshold: float = 20.0):     # nerf.py:17-29 (torch softplus)
    return F_.softplus(x, beta, threshold)


# ----------------------------------------------------------------------------- bf16 MLP mode
def _bround(x):
    return x.to(torch.bfloat16).to(x.dtype)


class _RoundedMatMul(torch.autograd.Function):
    """a @ b with BOTH operands rounded to bfloat16 (nearest even) and fp32 accumulation -- and the same for every matrix
    product differentiation adds, in either mode and to any order: the tangent product of forward mode is
    round(da) @ round(b), the products of the backward pass are round(g) @ round(b)^T and round(a)^T @ round(g).  This is what
    `float32_matmul_precision: medium` (BASELINE configs[2]) does to the matmuls of the reference's nn.Linear layers, the
    ones autograd adds included, and what mode 1 of the HIP MLP kernels does."""

    @staticmethod
    def forward(a, b):
        return _bround(a) @ _bround(b)

    @staticmethod
    def setup_context(ctx, inputs, output):
        ctx.save_for_backward(*inputs)
        ctx.save_for_forward(*inputs)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        return _RoundedMatMul.apply(g, b.transpose(-1, -2)), _RoundedMatMul.apply(a.transpose(-1, -2), g)

    @staticmethod
    def jvp(ctx, ad, bd):
        a, b = ctx.saved_tensors
        out = None
        if ad is not None:
            out = _RoundedMatMul.apply(ad, b)
        if bd is not None:
            t = _RoundedMatMul.apply(a, bd)
            out = t if out is None else out + t
        return out


BF16_LINEAR = False          # BASELINE configs[2] "bf16 MLP with fp32 composite": see bf16_linear()


class bf16_linear:
    """Context manager: every matrix product of the field's nn.Linear layers -- forward, tangent and backward -- sees
    bf16-rounded operands (`_RoundedMatMul`); fp32 accumulation, bias, activations and activation derivatives.  With it
    `step.training_forward` takes d log I / dt in FORWARD mode, as the HIP path does (the rounding sits on the tangent
    operands of every layer; reverse mode would put it on the cotangents: same mathematics, other round-off)."""

    def __enter__(self):
        global BF16_LINEAR
        self._old, BF16_LINEAR = BF16_LINEAR, True

    def __exit__(self, *a):
        global BF16_LINEAR
        BF16_LINEAR = self._old


def linear(x, w, b):
    if BF16_LINEAR:
        return _RoundedMatMul.apply(x, w.T) + b
    return x @ w.T + b


# ----------------------------------------------------------------------------- contraction
def contract(x: torch.Tensor, aabb: torch.Tensor, contraction_type: int) -> torch.Tensor:
    """World -> unit cube.  ngp.py:230-237, 68-106."""
    aabb_min, aabb_max = aabb[:3], aabb[3:]
    x = (x - aabb_min) / (aabb_max - aabb_min)
    if contraction_type == UN_BOUNDED_SPHERE:              # ngp.py:76-93
        x = x * 2 - 1
        mag = x.norm(dim=-1, keepdim=True)
        x = torch.where(mag > 1, (2 - 1 / mag) * (x / mag), x)
        x = x / 4 + 0.5
    elif contraction_type == UN_BOUNDED_TANH:              # ngp.py:96-106
        x = (torch.tanh(x - 0.5) + 1) / 2
    return x


def selector(x_unit: torch.Tensor) -> torch.Tensor:        # ngp.py:238
    return ((x_unit > 0.0) & (x_unit < 1.0)).all(dim=-1)


# ----------------------------------------------------------------------------- SH encoder
def sh_encode(d: torch.Tensor, degree: int = 4) -> torch.Tensor:
    """Real spherical harmonics, degree <= 4 (16 outputs).  sh_encoder.py:28-93."""
    assert 1 <= degree <= 4, "oracle restates degrees 1..4 (configs use 4)"
    x, y, z = d.unbind(-1)
    xy, xz, yz = x * y, x * z, y * z
    x2, y2, z2 = x * x, y * y, z * z
    out = [torch.full_like(x, 0.28209479177387814)]
    if degree > 1:
        out += [-0.48860251190291987 * y, 0.48860251190291987 * z, -0.48860251190291987 * x]
    if degree > 2:
        out += [
            1.0925484305920792 * xy,
            -1.0925484305920792 * yz,
            0.94617469575755997 * z2 - 0.31539156525251999,
            -1.0925484305920792 * xz,
            0.54627421529603959 * x2 - 0.54627421529603959 * y2,
        ]
    if degree > 3:
        out += [
            0.59004358992664352 * y * (-3.0 * x2 + y2),
            2.8906114426405538 * xy * z,
            0.45704579946446572 * y * (1.0 - 5.0 * z2),
            0.3731763325901154 * z * (5.0 * z2 - 3.0),
            0.45704579946446572 * x * (1.0 - 5.0 * z2),
            1.4453057213202769 * z * (x2 - y2),
            0.59004358992664352 * x * (-x2 + 3.0 * y2),
        ]
    return torch.stack(out, dim=-1)


# ----------------------------------------------------------------------------- parameters
def init_params(
    spec: hashgrid.HashGridSpec,
    radiance_dim: int = 1,
    seed: int = 0,
    table_kind: str = "uniform",
    table_scale: float = 1e-4,
    dtype=torch.float32,
) -> Dict[str, torch.Tensor]:
    """torch.nn.Linear default init U(+-1/sqrt(fan_in)) (ngp.py:179-185,198-204 pass
    hidden_init=None...bias_init=None so the nn.Linear defaults stay)."""
    g = torch.Generator().manual_seed(seed)

    def lin(out_f, in_f):
        bound = 1.0 / math.sqrt(in_f)
        w = (torch.rand(out_f, in_f, generator=g) * 2 - 1) * bound
        b = (torch.rand(out_f, generator=g) * 2 - 1) * bound
        return w.to(dtype), b.to(dtype)

    p = {}
    p["hash"] = hashgrid.init_table(spec, seed + 1000, table_scale, table_kind).to(dtype)
    p["base.w0"], p["base.b0"] = lin(64, spec.n_output_dims)
    p["base.wo"], p["base.bo"] = lin(16, 64)
    p["head.w0"], p["head.b0"] = lin(64, 31)
    p["head.w1"], p["head.b1"] = lin(64, 64)
    p["head.wo"], p["head.bo"] = lin(radiance_dim, 64)
    return p


PARAM_ORDER = (
    "hash", "base.w0", "base.b0", "base.wo", "base.bo",
    "head.w0", "head.b0", "head.w1", "head.b1", "head.wo", "head.bo",
)


# ----------------------------------------------------------------------------- field
# Activation alternatives of the YAML (models/nerf.py:8-29): hidden {softplus (beta 100), relu}, density {shifted_trunc_exp,
# softplus (beta 1), shifted_softplus = softplus(x - 1)}, radiance {softplus (beta 1), sigmoid}.  `acts` = dict(base_hidden,
# density, head_hidden, radiance); absent keys take the shipped configs' values.
WN_WEIGHTS = ("base.w0", "base.wo", "head.w0", "head.w1", "head.wo")


def weight_norm_params(p: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """`weight_norm: true` (ngp.py:207-228: torch.nn.utils.weight_norm on every Linear of the flagged MLP, dim 0): a parameter
    dict holding "<k>_g" (rows, 1) and "<k>_v" (rows, cols) in place of "<k>" -> the dict the field functions take, with
    W = g v / ||v||_row (differentiable).  Pinned to the reference by tests/golden/field_wn.npz."""
    q = {k: v for k, v in p.items() if not (k.endswith("_g") or k.endswith("_v"))}
    for k in WN_WEIGHTS:
        if k + "_v" in p:
            v, g = p[k + "_v"], p[k + "_g"]
            q[k] = v * (g.reshape(-1, 1) / v.norm(dim=1, keepdim=True))
    return q


DEFAULT_ACTS = dict(base_hidden="softplus", density="shifted_trunc_exp", head_hidden="softplus", radiance="softplus")


def _hidden(x, name):
    if name == "softplus":
        return softplus(x, 100.0)
    if name == "relu":
        return torch.relu(x)
    raise NotImplementedError(f"hidden activation {name!r} (nerf.py:17-20)")


def _density(x, name):
    if name == "shifted_trunc_exp":
        return shifted_trunc_exp(x)
    if name == "softplus":
        return softplus(x, 1.0)
    if name == "shifted_softplus":                            # nerf.py:8-13
        return softplus(x - 1.0, 1.0)
    raise NotImplementedError(f"density activation {name!r} (nerf.py:21-25)")


def _radiance(x, name):
    if name == "softplus":
        return softplus(x, 1.0)
    if name == "sigmoid":
        return torch.sigmoid(x)
    raise NotImplementedError(f"radiance activation {name!r} (nerf.py:26-29)")


# test hook (step.render_given): a callable xu -> features that replaces hashgrid.encode -- the rest of the field is then
# evaluated on somebody else's features (the HIP encoder's), which separates encoder differences from everything behind it
ENC_OVERRIDE = None


def query_density(
    x_world: torch.Tensor, p: Dict[str, torch.Tensor], spec, aabb: torch.Tensor,
    contraction_type: int = AABB, return_feat: bool = False, acts: Optional[dict] = None,
):
    """ngp.py:230-254: sigma = density_activation(raw0) * selector (default exp(raw0 - 1)), geo = raw[1:16]."""
    a = dict(DEFAULT_ACTS, **(acts or {}))
    xu = contract(x_world, aabb, contraction_type)
    sel = selector(xu)
    enc = hashgrid.encode(xu, p["hash"], spec) if ENC_OVERRIDE is None else ENC_OVERRIDE(xu)
    h = _hidden(linear(enc, p["base.w0"], p["base.b0"]), a["base_hidden"])
    raw = linear(h, p["base.wo"], p["base.bo"])
    sigma = _density(raw[:, :1], a["density"]) * sel[:, None].to(raw.dtype)
    if return_feat:
        return sigma, raw[:, 1:]
    return sigma


def query_rgb(dirs: torch.Tensor, geo: torch.Tensor, p: Dict[str, torch.Tensor], acts: Optional[dict] = None) -> torch.Tensor:
    """ngp.py:256-267: head([SH16(dir) | geo15]), hidden activation (default softplus 100), radiance activation (softplus 1)."""
    a = dict(DEFAULT_ACTS, **(acts or {}))
    h = torch.cat([sh_encode(dirs, 4), geo], dim=-1)
    h = _hidden(linear(h, p["head.w0"], p["head.b0"]), a["head_hidden"])
    h = _hidden(linear(h, p["head.w1"], p["head.b1"]), a["head_hidden"])
    return _radiance(linear(h, p["head.wo"], p["head.bo"]), a["radiance"])


def field_forward(x_world, dirs, p, spec, aabb, contraction_type: int = AABB, acts: Optional[dict] = None):
    """ngp.py:269-280 -> (rgb (n,C), sigma (n,1))."""
    sigma, geo = query_density(x_world, p, spec, aabb, contraction_type, return_feat=True, acts=acts)
    return query_rgb(dirs, geo, p, acts), sigma
