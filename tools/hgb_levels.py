import os, sys, math
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from robust_e_nerf_amd import ops, engine
dev = "cuda:0"
R, S = int(os.environ.get("RAYS", 131072)), 128
grid, n_table = ops.make_grid_desc()
g = torch.Generator().manual_seed(0)
ang = torch.rand(R, generator=g) * 2 * math.pi
o = torch.stack([4 * torch.cos(ang), 4 * torch.sin(ang), torch.rand(R, generator=g) - 0.5], -1)
d = (torch.rand(R, 3, generator=g) - 0.5) * 1.6 - o
d = d / d.norm(dim=-1, keepdim=True)
o, d = o.float().to(dev).contiguous(), d.float().to(dev).contiguous()
r = engine.Renderer(engine.NGPField(dev), engine.RenderCfg(sampler="uniform", n_uniform=S))
pk = r.sample(o, d, torch.rand(R, device=dev), True)
n = pk.n
dfeat = torch.randn(ops.n_blocks32(n) * 1024, device=dev)
gt = torch.zeros(n_table, device=dev)
ws = torch.empty(ops.hashgrid_bwd_binned_workspace_bytes(n), device=dev, dtype=torch.uint8)
kw = dict(scene=r.scene, rays=(o, d), samples=(pk.ray_indices, pk.t_starts, pk.t_ends), n=n, layout=1)
def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5
for l in range(16):
    t = timeit(lambda: ops.hashgrid_bwd_binned(grid, gt, dfeat, ws, level_mask=1 << l, **kw))
    print(f"level {l:2d} alone: {t:.3f} ms")
print("levels 0-4:", timeit(lambda: ops.hashgrid_bwd_binned(grid, gt, dfeat, ws, level_mask=0x1F, **kw)))
print("levels 5-15:", timeit(lambda: ops.hashgrid_bwd_binned(grid, gt, dfeat, ws, level_mask=0xFFE0, **kw)))
print("all levels:", timeit(lambda: ops.hashgrid_bwd_binned(grid, gt, dfeat, ws, **kw)))
print("all but 0:", timeit(lambda: ops.hashgrid_bwd_binned(grid, gt, dfeat, ws, level_mask=0xFFFE, **kw)))
print("all but 0, 1:", timeit(lambda: ops.hashgrid_bwd_binned(grid, gt, dfeat, ws, level_mask=0xFFFC, **kw)))
print("all but 0, 1, 2:", timeit(lambda: ops.hashgrid_bwd_binned(grid, gt, dfeat, ws, level_mask=0xFFF8, **kw)))
