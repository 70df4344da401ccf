"""``NGPradianceField``-shaped module (robust_e_nerf/external/ngp.py:109-280) on the fused kernels.

Same constructor arguments, same parameter / state-dict names as the reference
(``mlp_base.0.params``, ``mlp_base.1.hidden_layers.0.weight`` ... ``mlp_head.output_layer.bias``,
buffer ``aabb``), same methods (``query_density(x, return_feat)``, ``forward(positions, directions)``)
so the reference's ``render_image`` closures (``external/utils.py:68-96``) call it unchanged; the
arithmetic is two launches (hash grid, fused MLPs) instead of ~40, with a hand-written backward.
"""
from __future__ import annotations

import math
from typing import List, Union

import torch

from . import ops
from .engine import contract_points
from .nerfacc_api import ContractionType
from .tcnn_api import Encoding


class _Linear(torch.nn.Module):
    """Parameter holder with nn.Linear's names and default initialisation (ngp.py:179-185)."""

    def __init__(self, in_f: int, out_f: int):
        super().__init__()
        lin = torch.nn.Linear(in_f, out_f)
        self.weight, self.bias = lin.weight, lin.bias

    def effective_weight(self):
        """the layer's weight; under torch.nn.utils.weight_norm (ngp.py:224-228) g v / ||v|| from the CURRENT weight_g / weight_v
        (the hook that refreshes `.weight` runs on forward(), which a parameter holder never sees)"""
        if hasattr(self, "weight_g"):
            return torch._weight_norm(self.weight_v, self.weight_g, 0)
        return self.weight


class _MLPParams(torch.nn.Module):
    """Mirrors external/mlp.py:26-97's attribute layout: hidden_layers (ModuleList) + output_layer."""

    def __init__(self, dims: List[int]):
        super().__init__()
        self.hidden_layers = torch.nn.ModuleList([_Linear(dims[i], dims[i + 1]) for i in range(len(dims) - 2)])
        self.output_layer = _Linear(dims[-2], dims[-1])

    def tensors(self):
        out = []
        for layer in list(self.hidden_layers) + [self.output_layer]:
            out += [layer.effective_weight(), layer.bias]
        return out


def _act_name(v, slot: str) -> str:
    """name of an activation given as the YAML string or as the callable the reference passes (models/nerf.py:17-29 maps the
    names to torch.nn.Softplus(beta=100) / ReLU for hidden layers, shifted_trunc_exp / Softplus / shifted_softplus for the
    density, Softplus(beta=1) / Sigmoid for the radiance)"""
    if isinstance(v, str):
        return v
    if isinstance(v, torch.nn.ReLU):
        return "relu"
    if isinstance(v, torch.nn.Sigmoid):
        return "sigmoid"
    if isinstance(v, torch.nn.Softplus):
        want = 100 if slot == "hidden" else 1
        if v.beta != want:
            raise NotImplementedError(f"{slot} activation Softplus(beta={v.beta}): the kernels implement beta={want}")
        return "softplus"
    name = getattr(v, "__name__", "")
    if name in ("shifted_trunc_exp", "shifted_softplus"):
        return name
    raise NotImplementedError(f"{slot} activation {v!r}")


def matmul_mode() -> int:
    """mode of the matrix-core MLP kernels for the process-wide torch.get_float32_matmul_precision() -- the switch the
    reference sets from its YAML (scripts/run.py:34-35): highest -> 6 (every product to fp32 round-off), high -> 3 (each
    float32 as two bfloat16, three products), medium -> 1 (bf16 operands)"""
    return {"highest": 6, "high": 3, "medium": 1}[torch.get_float32_matmul_precision()]


class _FieldFn(torch.autograd.Function):
    """(x_world, dirs, table, *mlp tensors) -> (rgb (n,C), sigma (n,1)) with analytic backward to the
    table and MLP parameters (positions / directions are not differentiated on this path)."""

    @staticmethod
    def forward(ctx, x_world, dirs, table, module, density_only, *mlp_tensors):
        m = module
        x_world = x_world.contiguous().float()
        n = x_world.shape[0]
        mlp = torch.cat([t.reshape(-1) for t in mlp_tensors]).contiguous().float()
        xu = contract_points(x_world, m.aabb.tolist(), m.contraction_type.value)
        feat = ops.hashgrid_fwd(m.encoding.grid, table, x_unit=xu, n=n, layout=1)
        dirs_c = None if dirs is None else dirs.contiguous().float()
        # shipped activation set: the matrix-core kernels in the precision torch is set to; alternatives: exact-f32 kernels
        ctx.xmode = matmul_mode() if m._act_code == 0 else None
        if ctx.xmode is not None:
            rgb, sigma, base, _ = ops.mlp_fwd_x(mlp, m.radiance_dim, ctx.xmode, feat, m.scene, x_world=x_world, dirs=dirs_c, n=n,
                                                density_only=density_only, save=not density_only, save_acts=False)
        else:
            rgb, sigma, base = ops.mlp_fwd(mlp, m.radiance_dim, feat, m.scene, x_world=x_world, dirs=dirs_c, n=n,
                                           density_only=density_only, save_base=not density_only, act=m._act_code)
        ctx.module, ctx.n, ctx.density_only = m, n, density_only
        ctx.shapes = [t.shape for t in mlp_tensors]
        if not density_only:
            ctx.save_for_backward(x_world, dirs_c, xu, feat, base, rgb, mlp)
            return rgb, sigma[:, None]
        return sigma[:, None]

    @staticmethod
    def backward(ctx, *grads):
        m = ctx.module
        if ctx.density_only:
            raise NotImplementedError("query_density is used without gradients (sigma_fn / occ_eval_fn)")
        g_rgb, g_sigma = grads
        x_world, dirs, xu, feat, base, rgb, mlp = ctx.saved_tensors
        n = ctx.n
        dev = x_world.device
        g_rgb = torch.zeros_like(rgb) if g_rgb is None else g_rgb.contiguous().float()
        g_sigma = torch.zeros(n, device=dev) if g_sigma is None else g_sigma.reshape(-1).contiguous().float()
        g_mlp = torch.zeros_like(mlp)
        if ctx.xmode is not None:
            ws = torch.empty(ops.mlp_bwd_x_workspace_floats(m.radiance_dim), device=dev, dtype=torch.float32)
            dfeat = ops.mlp_bwd_x(mlp, m.radiance_dim, ctx.xmode, feat, base, None, m.scene, x_world=x_world, dirs=dirs, n=n,
                                  rgb=rgb, d_rgb=g_rgb, d_sigma=g_sigma, grad_mlp_params=g_mlp, workspace=ws)
        else:
            ws = torch.empty(ops.mlp_bwd_workspace_floats(m.radiance_dim), device=dev, dtype=torch.float32)
            dfeat = ops.mlp_bwd(mlp, m.radiance_dim, feat, base, m.scene, x_world=x_world, dirs=dirs, n=n, rgb=rgb,
                                d_rgb=g_rgb, d_sigma=g_sigma, grad_mlp_params=g_mlp, workspace=ws, act=m._act_code)
        g_table = torch.zeros(m.encoding.n_params, device=dev, dtype=torch.float32)
        ops.hashgrid_bwd_auto(m.encoding.grid, g_table, dfeat, x_unit=xu, n=n, layout=1)
        outs, off = [], 0
        for shp in ctx.shapes:
            k = math.prod(shp)
            outs.append(g_mlp[off: off + k].view(shp))
            off += k
        return (None, None, g_table, None, None, *outs)


class _TruncExp(torch.autograd.Function):
    """exp with the gradient clamp of the reference (external/ngp.py:45-65): backward g exp(min(x, 15)), written with
    differentiable ops so that create_graph=True works."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * torch.exp(x.clamp(max=15.0))


def sh4(d: torch.Tensor) -> torch.Tensor:
    """real spherical harmonics of degree 4 (16 components, tcnn sign convention: external/sh_encoder.py:56-93) in
    differentiable torch ops; the same polynomials as csrc/ren_mlp_common.h:sh4_select"""
    x, y, z = d.unbind(-1)
    xy, xz, yz, x2, y2, z2 = x * y, x * z, y * z, x * x, y * y, z * z
    return torch.stack([
        torch.full_like(x, 0.28209479177387814), -0.48860251190291987 * y, 0.48860251190291987 * z, -0.48860251190291987 * x,
        1.0925484305920792 * xy, -1.0925484305920792 * yz, 0.94617469575755997 * z2 - 0.31539156525251999,
        -1.0925484305920792 * xz, 0.54627421529603959 * x2 - 0.54627421529603959 * y2,
        0.59004358992664352 * y * (-3.0 * x2 + y2), 2.8906114426405538 * xy * z, 0.45704579946446572 * y * (1.0 - 5.0 * z2),
        0.3731763325901154 * z * (5.0 * z2 - 3.0), 0.45704579946446572 * x * (1.0 - 5.0 * z2),
        1.4453057213202769 * z * (x2 - y2), 0.59004358992664352 * x * (-x2 + 3.0 * y2)], -1)


class NGPradianceField(torch.nn.Module):
    """Instant-NGP radiance field with the reference's interface (external/ngp.py:109-280)."""

    def __init__(self, aabb: Union[torch.Tensor, List[float]], num_dim: int = 3, use_viewdirs: bool = True,
                 contraction_type: ContractionType = ContractionType.AABB, pos_encoding_config: dict = None,
                 dir_encoding_config: dict = None, mlp_base_config: dict = None, mlp_head_config: dict = None):
        super().__init__()
        assert num_dim == 3
        assert isinstance(contraction_type, ContractionType)
        pos_encoding_config = pos_encoding_config or dict(
            otype="HashGrid", n_levels=16, n_features_per_level=2, log2_hashmap_size=19, base_resolution=16,
            per_level_scale=1.4472692012786865, interpolation="Linear")
        dir_encoding_config = dir_encoding_config or dict(degree=4)
        mlp_base_config = mlp_base_config or {}
        mlp_head_config = mlp_head_config or {}
        # the fused kernels implement exactly configs/train/*.yaml's network; anything else is refused loudly
        if not use_viewdirs or dir_encoding_config.get("degree", 4) != 4:
            raise NotImplementedError("fused kernels: view directions with SH degree 4")
        if (mlp_base_config.get("n_neurons", 64), mlp_base_config.get("n_hidden_layers", 1),
                mlp_base_config.get("geo_feat_dim", 15)) != (64, 1, 15):
            raise NotImplementedError("fused kernels: base MLP 32->64->16")
        if (mlp_head_config.get("n_neurons", 64), mlp_head_config.get("n_hidden_layers", 2)) != (64, 2):
            raise NotImplementedError("fused kernels: head MLP 31->64->64->C")
        # activations: the reference hands over callables (models/nerf.py:17-29, 150-163) or their YAML names; both are
        # resolved to the names of its own tables.  Alternatives run on the exact-f32 kernels (REN_KNOB_ACTIVATIONS).
        self.acts = dict(base_hidden=_act_name(mlp_base_config.get("hidden_activation", "softplus"), "hidden"),
                         density=_act_name(mlp_base_config.get("density_activation", "shifted_trunc_exp"), "density"),
                         head_hidden=_act_name(mlp_head_config.get("hidden_activation", "softplus"), "hidden"),
                         radiance=_act_name(mlp_head_config.get("radiance_activation", "softplus"), "radiance"))
        self._act_code = ops.activation_code(**self.acts)
        self.radiance_dim = int(mlp_head_config.get("output_dim", 1))
        if self.radiance_dim not in (1, 3):
            raise NotImplementedError("radiance_dim must be 1 or 3")
        if not isinstance(aabb, torch.Tensor):
            aabb = torch.tensor(aabb, dtype=torch.float32)
        self.register_buffer("aabb", aabb)
        self.num_dim, self.use_viewdirs, self.contraction_type = 3, True, contraction_type
        self.geo_feat_dim = 15
        encoding = Encoding(3, pos_encoding_config)
        if encoding.grid.n_levels != 16:
            raise NotImplementedError("fused kernels: 16 levels x 2 features")
        # names as in the reference: mlp_base = Sequential(encoding, MLP), mlp_head = MLP
        self.mlp_base = torch.nn.Sequential(encoding, _MLPParams([32, 64, 16]))
        self.mlp_head = _MLPParams([31, 64, 64, self.radiance_dim])
        for mlp, mlp_config in ((self.mlp_base[1], mlp_base_config), (self.mlp_head, mlp_head_config)):   # ngp.py:207-228
            if mlp_config.get("weight_norm", False):
                for module in mlp.modules():
                    if isinstance(module, _Linear):
                        torch.nn.utils.weight_norm(module)
        self.scene = ops.make_scene_desc(aabb.tolist(), contraction_type.value)

    @property
    def encoding(self) -> Encoding:
        return self.mlp_base[0]

    def _mlp_tensors(self):
        return self.mlp_base[1].tensors() + self.mlp_head.tensors()

    def query_density(self, x, return_feat: bool = False):
        if return_feat:
            raise NotImplementedError("return_feat=True is internal to forward() in the fused path")
        shp = x.shape[:-1]
        with torch.no_grad():
            sigma = _FieldFn.apply(x.reshape(-1, 3), None, self.encoding.params, self, True, *self._mlp_tensors())
        return sigma.view(*shp, 1)

    def _forward_twice_differentiable(self, x, d):
        """The same field composed op by op -- contraction, HIP hash-grid encoding (tcnn_api, twice differentiable), the two
        tiny MLPs and the SH encoder in torch, as the reference does itself for this very reason (ngp.py:5-19) -- for the
        log-intensity-gradient loss, whose autograd.gradient(..., create_graph=True) w.r.t. the ray timestamps needs
        d(rgb, sigma)/d(position, direction) and their second derivatives (robust_e_nerf.py:383-409)."""
        F = torch.nn.functional
        xu = contract_points(x, self.aabb.tolist(), self.contraction_type.value)
        sel = ((xu > 0.0) & (xu < 1.0)).all(dim=-1, keepdim=True)
        feat = self.encoding(xu)
        hid = lambda v, name: F.relu(v) if name == "relu" else F.softplus(v, beta=100)
        a = self.acts
        (w0, b0, wo, bo), (hw0, hb0, hw1, hb1, hwo, hbo) = self.mlp_base[1].tensors(), self.mlp_head.tensors()
        o = F.linear(hid(F.linear(feat, w0, b0), a["base_hidden"]), wo, bo)
        raw = o[:, :1]
        dens = {"shifted_trunc_exp": lambda v: _TruncExp.apply(v - 1.0), "softplus": F.softplus,
                "shifted_softplus": lambda v: F.softplus(v - 1.0)}[a["density"]]          # nerf.py:8-13,20-24
        sigma = dens(raw) * sel
        hin = torch.cat([sh4(d), o[:, 1:]], dim=-1)                                     # ngp.py:256-260
        p = hid(F.linear(hin, hw0, hb0), a["head_hidden"])
        q = hid(F.linear(p, hw1, hb1), a["head_hidden"])
        z = F.linear(q, hwo, hbo)
        rgb = torch.sigmoid(z) if a["radiance"] == "sigmoid" else F.softplus(z, beta=1)
        return rgb, sigma

    def forward(self, positions: torch.Tensor, directions: torch.Tensor = None):
        assert directions is not None and positions.shape == directions.shape, \
            f"{positions.shape} v.s. {None if directions is None else directions.shape}"
        shp = positions.shape[:-1]
        if torch.is_grad_enabled() and (positions.requires_grad or directions.requires_grad):
            rgb, sigma = self._forward_twice_differentiable(positions.reshape(-1, 3), directions.reshape(-1, 3))
            return rgb.view(*shp, self.radiance_dim), sigma.view(*shp, 1)
        rgb, sigma = _FieldFn.apply(positions.reshape(-1, 3), directions.reshape(-1, 3), self.encoding.params, self,
                                    False, *self._mlp_tensors())
        return rgb.view(*shp, self.radiance_dim), sigma.view(*shp, 1)
