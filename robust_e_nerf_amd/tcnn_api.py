"""Op-by-op seam: ``tinycudann.Encoding`` as the reference instantiates it
(``robust_e_nerf/external/ngp.py:166-170``): ``Encoding(n_input_dims=3, encoding_config, dtype)``,
an ``nn.Module`` with one flat float32 ``params`` Parameter (state-dict key ``mlp_base.0.params``)
and ``n_output_dims``; forward (n,3) in the unit cube -> (n, 2L) row-major.
"""
from __future__ import annotations

import torch

from . import ops


def _to_rows(frag: torch.Tensor, n: int) -> torch.Tensor:
    """fragment layout [block][level][feature][32 samples] -> (n, 2L) row-major"""
    return frag.view(-1, 16, 2, 32).permute(0, 3, 1, 2).reshape(-1, 32)[:n]


def _to_frag(rows: torch.Tensor) -> torch.Tensor:
    n = rows.shape[0]
    nb = ops.n_blocks32(n)
    pad = torch.zeros(nb * 32, 32, device=rows.device, dtype=torch.float32)
    pad[:n] = rows
    return pad.view(nb, 32, 16, 2).permute(0, 2, 3, 1).contiguous().view(-1)


def _encode_with_tangent(module, params, x, v):
    """(encoding(x), d encoding / dx . v), both (n, 2L): ren_hashgrid_fwd_jvp over one zero-length "ray" per point
    (origin = the point, its time derivative = v, unit-cube scene so the contraction is the identity)."""
    import ctypes
    from . import _lib
    from .ops import _ptr, _stream
    n = x.shape[0]
    nb = ops.n_blocks32(n)
    dev = x.device
    feat, featd = torch.empty(nb * 1024, device=dev), torch.empty(nb * 1024, device=dev)
    zero3, zero1 = torch.zeros(n, 3, device=dev), torch.zeros(n, device=dev)
    ri = torch.arange(n, device=dev, dtype=torch.int32)
    _lib.check(_lib.load().ren_hashgrid_fwd_jvp(ctypes.byref(module.grid), _ptr(params), ctypes.byref(module.unit_scene), _ptr(x),
                                                _ptr(zero3), _ptr(v), _ptr(zero3), _ptr(ri), _ptr(zero1), _ptr(zero1), n,
                                                _ptr(feat), _ptr(featd), None, _stream()), "ren_hashgrid_fwd_jvp")
    return _to_rows(feat, n), _to_rows(featd, n)


class _HashGridDx(torch.autograd.Function):
    """d loss / d position = J_x^T g of the encoding, itself differentiable w.r.t. g and the table (what
    ``autograd.gradient(..., create_graph=True)`` of the reference needs, robust_e_nerf/utils/autograd.py:4-34).
    The mixed second derivatives of the trilinear interpolation w.r.t. the position are not propagated (d/dx of this
    function is zero), as in tinycudann's HashGrid double backward; the fused engine keeps them (csrc/ren_jvp2.hip)."""

    @staticmethod
    def forward(ctx, x, params, g, module):
        x, g = x.detach().contiguous().float(), g.detach().contiguous().float()
        cols = []
        for k in range(3):
            e = torch.zeros_like(x)
            e[:, k] = 1.0
            cols.append((_encode_with_tangent(module, params.detach(), x, e)[1] * g).sum(-1))
        ctx.save_for_backward(x, params.detach(), g)
        ctx.module = module
        return torch.stack(cols, -1)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, u):
        x, params, g = ctx.saved_tensors
        m = ctx.module
        u = u.contiguous().float()
        n = x.shape[0]
        g_g = g_p = None
        if ctx.needs_input_grad[2]:
            g_g = _encode_with_tangent(m, params, x, u)[1]                       # J_x u
        if ctx.needs_input_grad[1]:
            # d/d table of sum_c (dw_c/dx . u) table_c . g  =  scatter of (dw_c/dx . u) g
            if not ops.binned_supported(m.grid):
                raise NotImplementedError("second-order table gradient of a DenseGrid with a level above 2^19 entries "
                                          "(the tangent scatter exists in the binned form only)")
            g_p = torch.zeros(m.n_params, device=x.device, dtype=torch.float32)
            ws = torch.empty(ops.hashgrid_bwd_binned_workspace_bytes(n), device=x.device, dtype=torch.uint8)
            zero3, zero1 = torch.zeros(n, 3, device=x.device), torch.zeros(n, device=x.device)
            ops.hashgrid_bwd_binned(m.grid, g_p, torch.zeros(ops.n_blocks32(n) * 1024, device=x.device), ws, scene=m.unit_scene,
                                    rays=(x, zero3), samples=(torch.arange(n, device=x.device, dtype=torch.int32), zero1, zero1),
                                    n=n, layout=1, tangent=(u, zero3, _to_frag(g)))
        return None, g_p, g_g, None


class _HashGridFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, params, module):
        xc = x.detach().contiguous().float()
        n = xc.shape[0]
        out = ops.hashgrid_fwd(module.grid, params.detach(), x_unit=xc, n=n, layout=0)
        ctx.save_for_backward(x, params)
        ctx.module = module
        return out

    @staticmethod
    def backward(ctx, g_out):
        x, params = ctx.saved_tensors
        m = ctx.module
        n = x.shape[0]
        g_x = g_params = None
        if ctx.needs_input_grad[0]:                              # the log-intensity-gradient loss: d/d position, twice differentiable
            g_x = _HashGridDx.apply(x, params, g_out, m)
        if ctx.needs_input_grad[1]:
            g_params = torch.zeros(m.n_params, device=x.device, dtype=torch.float32)
            g = g_out.detach().contiguous().float()
            ops.hashgrid_bwd_auto(m.grid, g_params, g, x_unit=x.detach().contiguous().float(), n=n, layout=0)
        return g_x, g_params, None


class Encoding(torch.nn.Module):
    def __init__(self, n_input_dims: int, encoding_config: dict, dtype=torch.float32, seed: int = 1337):
        super().__init__()
        if n_input_dims != 3:
            raise NotImplementedError("n_input_dims must be 3")
        otype = encoding_config.get("otype", "HashGrid")
        if otype not in ("HashGrid", "DenseGrid", "TiledGrid"):
            raise NotImplementedError(f"otype {otype} (HashGrid, DenseGrid, TiledGrid are built)")
        if encoding_config.get("interpolation", "Linear") != "Linear":
            raise NotImplementedError("only Linear interpolation is built")
        if dtype != torch.float32:
            raise NotImplementedError("the reference forces float32 (ngp.py:169)")
        self.grid, self.n_params = ops.make_grid_desc(
            encoding_config.get("n_levels", 16), encoding_config.get("n_features_per_level", 2),
            encoding_config.get("log2_hashmap_size", 19), encoding_config.get("base_resolution", 16),
            encoding_config.get("per_level_scale", 2.0), otype)
        self.n_input_dims = 3
        self.unit_scene = ops.make_scene_desc([0.0, 0.0, 0.0, 1.0, 1.0, 1.0], ops.AABB)
        self.n_output_dims = self.grid.n_levels * 2
        g = torch.Generator().manual_seed(seed)
        # tcnn initialises grid parameters U(-1e-4, 1e-4)
        self.params = torch.nn.Parameter((torch.rand(self.n_params, generator=g) * 2 - 1) * 1e-4)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return _HashGridFn.apply(x, self.params, self)
