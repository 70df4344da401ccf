"""Which kernel on the main stream makes `pose_rays_kernel` on a second stream return wrong rays (the packed-FP32 hazard of
round 4, profiles/NOTES.md)?  The pose kernel runs on a side stream with no dependency while ONE kind of kernel loops on the
main stream; its output is compared with the output computed on an idle chip.  Meaningful on a build WITH packed code in
ren_pose.hip (drop NO_SLP for it in robust_e_nerf_amd/build.py and rebuild); on the tree's own build every line reads 0.
GPU only.   python tools/aggressor_probe.py [iterations]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robust_e_nerf_amd import ops, engine
dev = "cuda:0"
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
n = int(os.environ.get("N", 2097152))
C = 1
gen = torch.Generator().manual_seed(0)
# ---- aggressors (independent data, as tools/overlap_probe.py)
x = (torch.rand(n, 3, generator=gen) * 2.6 - 1.3).to(dev)
dd = torch.randn(n, 3, generator=gen)
dd = (dd / dd.norm(dim=-1, keepdim=True)).to(dev)
nb = ops.n_blocks32(n)
feat = (torch.rand(nb * 1024, generator=gen) - 0.5).to(dev)
params = ((torch.rand(ops.mlp_param_count(C), generator=gen) - 0.5) * 0.5).to(dev)
scene = ops.make_scene_desc([-1.5, -1.5, -1.5, 1.5, 1.5, 1.5], 0)
d_rgb = torch.randn(n, C, generator=gen).to(dev)
d_sigma = (torch.randn(n, generator=gen) * 0.1).to(dev)
rgb, sigma, base, _ = ops.mlp_fwd_x(params, C, 6, feat, scene, x_world=x, dirs=dd, n=n, save=True, save_acts=False)
gm = torch.zeros_like(params)
ws = torch.empty(ops.mlp_bwd_x_workspace_floats(C), device=dev)
fld = engine.NGPField(dev)
xu = ((x + 1.5) / 3.0).clamp(0, 1).contiguous()
dfeat = torch.randn(nb * 1024, generator=gen).to(dev)
gt = torch.zeros_like(fld.table)
hws = torch.empty(ops.hashgrid_bwd_binned_workspace_bytes(n), device=dev, dtype=torch.uint8)
featbuf = torch.empty(nb * ops.FRAG_FLOATS_PER_BLOCK, device=dev)
AGG = {
    "nothing": lambda: None,
    "mlp_fwd_x": lambda: ops.mlp_fwd_x(params, C, 6, feat, scene, x_world=x, dirs=dd, n=n, save=True, save_acts=False),
    "mlp_bwd_x (persistent head + base kernels)": lambda: ops.mlp_bwd_x(params, C, 6, feat, base, None, scene, x_world=x, dirs=dd, n=n, rgb=rgb,
                                                                       d_rgb=d_rgb, d_sigma=d_sigma, grad_mlp_params=gm, workspace=ws),
    "hashgrid_bwd_binned": lambda: ops.hashgrid_bwd_binned(fld.grid, gt, dfeat, hws, x_unit=xu, n=n, layout=1),
    "hashgrid_fwd": lambda: ops.hashgrid_fwd(fld.grid, fld.table, x_unit=xu, n=n, layout=1, out=featbuf),
    "adam_step (table)": lambda: ops.adam_step(fld.flat, fld.grad, m_, v_, lr=0.0, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, step=1,
                                              grad_scale=1.0, zero_grad=False),
}
m_, v_ = torch.zeros_like(fld.flat), torch.zeros_like(fld.flat)
# ---- the victim: poses + rays of 8 192 timestamps
g = np.load(os.path.join(ROOT, "tests", "golden", "training_step_diff.npz"))
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
tab_ts, tab_pos, tab_quat, Kinv = T(g["tab_ts"]), T(g["tab_pos"]).float(), T(g["tab_quat"]).float(), T(g["Kinv"]).float().contiguous()
B = 4096
ts = (torch.rand(2 * B, generator=gen, dtype=torch.float64) * float(g["tab_ts"][-1])).to(dev)
px = torch.stack([torch.randint(0, 346, (B,), generator=gen), torch.randint(0, 260, (B,), generator=gen)], -1).float().to(dev).contiguous()
torch.cuda.synchronize()
o_ref, d_ref = ops.pose_rays(ts, px, Kinv, tab_ts, tab_pos, tab_quat)
torch.cuda.synchronize()
side = torch.cuda.Stream()
# ---- a second victim: the density pre-pass of a render (mlp_fwd_x, density only -- a matrix-core kernel WITH packed code of
# its own), which Trainer.step's "begun" placement runs on the side stream beside the l_diff backward
nv = 262144
xv = (torch.rand(nv, 3, generator=gen) * 2.6 - 1.3).to(dev)
featv = (torch.rand(ops.n_blocks32(nv) * 1024, generator=gen) - 0.5).to(dev)
torch.cuda.synchronize()
_, sig_ref, _, _ = ops.mlp_fwd_x(params, C, 6, featv, scene, x_world=xv, dirs=None, n=nv, density_only=True)
torch.cuda.synchronize()
for name, fn in AGG.items():
    bad = torch.zeros(1, device=dev, dtype=torch.int64)
    for it in range(iters):
        fn()
        with torch.cuda.stream(side):
            _, sg, _, _ = ops.mlp_fwd_x(params, C, 6, featv, scene, x_world=xv, dirs=None, n=nv, density_only=True)
            bad += ((sg != sig_ref).sum() > 0).to(torch.int64)
        if it % 8 == 7:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    print(f"main stream: {name:45s} side-stream density pre-pass (mlp_fwd_x) launches with wrong values: {int(bad)} of {iters}", flush=True)
for name, fn in AGG.items():
    bad_iters = torch.zeros(1, device=dev, dtype=torch.int64)
    bad_rays = torch.zeros(1, device=dev, dtype=torch.int64)
    torch.cuda.synchronize()
    for it in range(iters):
        fn()
        with torch.cuda.stream(side):
            o_, d_ = ops.pose_rays(ts, px, Kinv, tab_ts, tab_pos, tab_quat)
            wrong = ((d_ != d_ref).any(dim=1) | (o_ != o_ref).any(dim=1)).sum()
            bad_rays += wrong
            bad_iters += (wrong > 0).to(torch.int64)
        if it % 8 == 7:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    print(f"main stream: {name:45s} side-stream pose_rays launches with wrong rays: {int(bad_iters)} of {iters}, wrong rays {int(bad_rays)}", flush=True)
