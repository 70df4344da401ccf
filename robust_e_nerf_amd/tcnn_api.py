"""Op-by-op seam: ``tinycudann.Encoding`` as the reference instantiates it
(``robust_e_nerf/external/ngp.py:166-170``): ``Encoding(n_input_dims=3, encoding_config, dtype)``,
an ``nn.Module`` with one flat float32 ``params`` Parameter (state-dict key ``mlp_base.0.params``)
and ``n_output_dims``; forward (n,3) in the unit cube -> (n, 2L) row-major.
"""
from __future__ import annotations

import torch

from . import ops


class _HashGridFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, params, module):
        x = x.contiguous().float()
        n = x.shape[0]
        out = ops.hashgrid_fwd(module.grid, params, x_unit=x, n=n, layout=0)
        ctx.save_for_backward(x)
        ctx.module = module
        ctx.x_needs_grad = x.requires_grad
        return out

    @staticmethod
    def backward(ctx, g_out):
        (x,) = ctx.saved_tensors
        m = ctx.module
        if ctx.needs_input_grad[0]:
            raise NotImplementedError(
                "d(encoding)/d(position) (needed only by the log-intensity-gradient loss) is not built yet")
        n = x.shape[0]
        g_params = torch.zeros(m.n_params, device=x.device, dtype=torch.float32)
        g = g_out.contiguous().float()
        ws = torch.empty(ops.hashgrid_bwd_binned_workspace_bytes(n), device=x.device, dtype=torch.uint8)
        ops.hashgrid_bwd_binned(m.grid, g_params, g, ws, x_unit=x, n=n, layout=0)
        return None, g_params, None


class Encoding(torch.nn.Module):
    def __init__(self, n_input_dims: int, encoding_config: dict, dtype=torch.float32, seed: int = 1337):
        super().__init__()
        if n_input_dims != 3:
            raise NotImplementedError("n_input_dims must be 3")
        if encoding_config.get("otype", "HashGrid") != "HashGrid":
            raise NotImplementedError(f"otype {encoding_config.get('otype')} (only HashGrid is built)")
        if encoding_config.get("interpolation", "Linear") != "Linear":
            raise NotImplementedError("only Linear interpolation is built")
        if dtype != torch.float32:
            raise NotImplementedError("the reference forces float32 (ngp.py:169)")
        self.grid, self.n_params = ops.make_grid_desc(
            encoding_config.get("n_levels", 16), encoding_config.get("n_features_per_level", 2),
            encoding_config.get("log2_hashmap_size", 19), encoding_config.get("base_resolution", 16),
            encoding_config.get("per_level_scale", 2.0))
        self.n_input_dims = 3
        self.n_output_dims = self.grid.n_levels * 2
        g = torch.Generator().manual_seed(seed)
        # tcnn initialises grid parameters U(-1e-4, 1e-4)
        self.params = torch.nn.Parameter((torch.rand(self.n_params, generator=g) * 2 - 1) * 1e-4)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return _HashGridFn.apply(x, self.params, self)
