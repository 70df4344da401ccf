"""Micro-benchmark of the hash-grid kernels at config-B size (GPU only; tuning aid)."""
import os, sys, time, math
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robust_e_nerf_amd import _lib, ops, engine
dev = "cuda:0"
R, S = 131072, 128
grid, n_table = ops.make_grid_desc()
table = ((torch.rand(ops.make_grid_desc()[1], generator=torch.Generator().manual_seed(0)) * 2 - 1) * 0.1).to(dev)
g = torch.Generator().manual_seed(0)
ang = torch.rand(R, generator=g) * 2 * math.pi
o = torch.stack([4 * torch.cos(ang), 4 * torch.sin(ang), torch.rand(R, generator=g) - 0.5], -1)
d = (torch.rand(R, 3, generator=g) - 0.5) * 1.6 - o
d = d / d.norm(dim=-1, keepdim=True)
o, d = o.float().to(dev).contiguous(), d.float().to(dev).contiguous()
cfg = engine.RenderCfg(sampler="uniform", n_uniform=S)
fld = engine.NGPField(dev)
r = engine.Renderer(fld, cfg)
pk = r.sample(o, d, torch.rand(R, device=dev), True)
n = pk.n
scene = r.scene
samples = (pk.ray_indices, pk.t_starts, pk.t_ends)
print("n", n)
dfeat = torch.randn(ops.n_blocks32(n) * 1024, device=dev)
gt = torch.zeros(n_table, device=dev)
def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3
for v in os.environ.get("VARIANTS", "0,1,2,3").split(","):
    _lib.load().ren_set_knob(ops.KNOBS["hg_variant"], int(v))
    tf = timeit(lambda: ops.hashgrid_fwd(grid, table, scene=scene, rays=(o, d), samples=samples, n=n, layout=1))
    tb = timeit(lambda: ops.hashgrid_bwd(grid, gt, dfeat, scene=scene, rays=(o, d), samples=samples, n=n, layout=1))
    print(f"variant {v}: fwd {tf:.2f} ms ({1036*n/tf/1e6:.0f} GB/s alg)  bwd {tb:.2f} ms ({2060*n/tb/1e6:.0f} GB/s alg)", flush=True)

ws = torch.empty(ops.hashgrid_bwd_binned_workspace_bytes(n), device=dev, dtype=torch.uint8)
tb = timeit(lambda: ops.hashgrid_bwd_binned(grid, gt, dfeat, ws, scene=scene, rays=(o, d), samples=samples, n=n, layout=1))
print(f"binned: bwd {tb:.2f} ms ({2060*n/tb/1e6:.0f} GB/s alg)", flush=True)
gt.zero_(); _lib.load().ren_set_knob(ops.KNOBS["hg_variant"], 2)
ops.hashgrid_bwd(grid, gt, dfeat, scene=scene, rays=(o, d), samples=samples, n=n, layout=1)
ga = gt.clone(); gt.zero_()
ops.hashgrid_bwd_binned(grid, gt, dfeat, ws, scene=scene, rays=(o, d), samples=samples, n=n, layout=1)
print("max abs diff atomic vs binned", float((ga - gt).abs().max()), "max", float(ga.abs().max()))
ops.profile_start()
for _ in range(3):
    ops.hashgrid_bwd_binned(grid, gt, dfeat, ws, scene=scene, rays=(o, d), samples=samples, n=n, layout=1)
print(ops.profile_stop())
