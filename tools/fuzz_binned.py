"""Randomised cross-check of ren_hashgrid_bwd_binned against the atomic scatter: sizes around every threshold of the
region / part sizing (sampled count strides at 1024 / 2048 / 4096 count blocks), uniform and clustered points, both
feature layouts, regular and halved regions.  GPU only; prints the worst relative deviation."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robust_e_nerf_amd import ops
dev = "cuda:0"
grid, n_table = ops.make_grid_desc()
g = torch.Generator(device=dev).manual_seed(0)
sizes = [1, 31, 32, 33, 1000, 65537, 1024 * 1024 - 1, 1024 * 1024, 1024 * 1024 + 1, 2048 * 1024 + 5, 4096 * 1024 - 3,
         4096 * 1024 + 9, 3_000_001] + [int(x) for x in torch.randint(1, 3_000_000, (12,)).tolist()]
worst = 0.0
for k, n in enumerate(sizes):
    x = torch.rand(n, 3, generator=g, device=dev)
    mode = k % 3
    if mode == 1:                                   # half of the points in a 3 % corner
        x[: n // 2] = 0.6 + 0.03 * x[: n // 2]
    elif mode == 2:                                 # sorted along x: count blocks are NOT exchangeable
        x = x[torch.argsort(x[:, 0])].contiguous()
    gout = torch.randn(n, 32, generator=g, device=dev)
    ref = torch.zeros(n_table, device=dev)
    ops.hashgrid_bwd(grid, ref, gout, x_unit=x, n=n, layout=0)
    ws = torch.empty(ops.hashgrid_bwd_binned_workspace_bytes(n), device=dev, dtype=torch.uint8)
    for halve in (0, 1):
        with ops.knob("hgb_halve_regions", halve):
            out = torch.zeros(n_table, device=dev)
            ops.hashgrid_bwd_binned(grid, out, gout, ws, x_unit=x, n=n, layout=0)
            torch.cuda.synchronize()
        err = float((out - ref).abs().max() / ref.abs().max().clamp(min=1e-30))
        worst = max(worst, err)
        flag = "" if err < 2e-4 else "   <-- MISMATCH"
        print(f"n={n:8d} mode={mode} halve={halve} rel err {err:.2e}{flag}", flush=True)
print("worst", worst)
assert worst < 2e-4
