"""Can the persistent MLP backward kernels and the binned hash-grid backward share the GPU?  Times both alone and concurrently on
two streams (independent data).  usage: python tools/overlap_probe.py [variant.so of csrc/ren_mlp_x.hip]   env: N"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robust_e_nerf_amd import _lib, ops, engine

dev = "cuda:0"
n = int(os.environ.get("N", 8388608))
if len(sys.argv) > 1:
    var, lib = ctypes.CDLL(os.path.abspath(sys.argv[1])), _lib.load()
    for name in ("ren_mlp_bwd_x", "ren_mlp_bwd_x_workspace_floats"):
        fn = getattr(var, name)
        fn.restype, fn.argtypes = _lib.SIGNATURES[name]
        setattr(lib, name, fn)
C = 1
g = torch.Generator().manual_seed(0)
x = (torch.rand(n, 3, generator=g) * 2.6 - 1.3).to(dev)
d = torch.randn(n, 3, generator=g)
d = (d / d.norm(dim=-1, keepdim=True)).to(dev)
nb = ops.n_blocks32(n)
feat = (torch.rand(nb * 1024, generator=g) - 0.5).to(dev)
params = ((torch.rand(ops.mlp_param_count(C), generator=g) - 0.5) * 0.5).to(dev)
scene = ops.make_scene_desc([-1.5, -1.5, -1.5, 1.5, 1.5, 1.5], 0)
d_rgb = torch.randn(n, C, generator=g).to(dev)
d_sigma = (torch.randn(n, generator=g) * 0.1).to(dev)
rgb, sigma, base, _ = ops.mlp_fwd_x(params, C, 6, feat, scene, x_world=x, dirs=d, n=n, save=True, save_acts=False)
gm = torch.zeros_like(params)
ws = torch.empty(ops.mlp_bwd_x_workspace_floats(C), device=dev)
fld = engine.NGPField(dev)
grid = fld.grid
xu = ((x + 1.5) / 3.0).clamp(0, 1).contiguous()
dfeat = torch.randn(nb * 1024, generator=g).to(dev)
gt = torch.zeros_like(fld.table)
hws = torch.empty(ops.hashgrid_bwd_binned_workspace_bytes(n), device=dev, dtype=torch.uint8)

def mlp():
    ops.mlp_bwd_x(params, C, 6, feat, base, None, scene, x_world=x, dirs=d, n=n, rgb=rgb, d_rgb=d_rgb, d_sigma=d_sigma,
                  grad_mlp_params=gm, workspace=ws)

def hgb():
    ops.hashgrid_bwd_binned(grid, gt, dfeat, hws, x_unit=xu, n=n, layout=1)

def wall(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def both():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        mlp()
    with torch.cuda.stream(s2):
        hgb()
    cur.wait_stream(s1); cur.wait_stream(s2)

print(f"n {n}: mlp_bwd {wall(mlp):.2f} ms  hashgrid_bwd_binned {wall(hgb):.2f} ms  concurrent {wall(both):.2f} ms")
