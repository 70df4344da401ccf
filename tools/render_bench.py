"""Novel-view inference render (BASELINE configs[4] shape: 640x480, occupancy-grid marching, no jitter): ms per image
for different ray-chunk sizes.  Synthetic scene as bench.py.  GPU only."""
import math, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from robust_e_nerf_amd import engine, evaluation
dev = "cuda:0"
H, W = 480, 640
gen = torch.Generator().manual_seed(0)
def lin(o, i):
    b = 1 / math.sqrt(i)
    return (torch.rand(o, i, generator=gen) * 2 - 1) * b, (torch.rand(o, generator=gen) * 2 - 1) * b
p = {}
p["base.w0"], p["base.b0"] = lin(64, 32); p["base.wo"], p["base.bo"] = lin(16, 64)
p["head.w0"], p["head.b0"] = lin(64, 31); p["head.w1"], p["head.b1"] = lin(64, 64); p["head.wo"], p["head.bo"] = lin(1, 64)
p["hash"] = (torch.rand(engine.ops.make_grid_desc()[1], generator=gen) * 2 - 1) * 0.1
CONFIG_E = "--config-e" in sys.argv   # BASELINE configs[4] settings: mocap-desk2.yaml (sphere contraction, 256^3 grid, cone angle, near/far)
fld = engine.NGPField(dev); fld.load(p)
if CONFIG_E:
    aabb = bench.E_AABB
    r = engine.Renderer(fld, engine.RenderCfg(aabb=aabb, contraction_type=engine.ops.UN_BOUNDED_SPHERE, occ_res=(256,) * 3,
                                               near_plane=0.05, far_plane=3.0, render_step_size=math.sqrt(3) * 1.5 / 1024,
                                               cone_angle=0.004, sampler="occgrid"))
    g3 = np.stack(np.meshgrid(*[np.arange(256)] * 3, indexing="ij"), -1)
    r.binary.copy_(torch.from_numpy((np.linalg.norm((g3 + 0.5) / 256 - 0.5, axis=-1) < 0.1).astype(np.uint8).reshape(-1)).to(dev))
    K = np.array([[500.0, 0, W / 2 - 0.5], [0, 500.0, H / 2 - 0.5], [0, 0, 1]])
    centre = torch.tensor([1.25, -1.35, 1.1], device=dev)
    pos = centre + torch.tensor([0.85, 0.0, 0.1], device=dev)
    z = (centre - pos) / (centre - pos).norm()
    bk = None                                               # alpha_over_white_bg false: no background, is_valid = opacity > 0
else:
    aabb = (-1.5,) * 3 + (1.5,) * 3
    r = engine.Renderer(fld, engine.RenderCfg(aabb=aabb, sampler="occgrid"))
    r.binary.copy_(torch.from_numpy(bench.ball_binary(128, 0.42, aabb)).to(dev))
    K = np.array([[480.0 * W / 346, 0, W / 2 - 0.5], [0, 480.0 * W / 346, H / 2 - 0.5], [0, 0, 1]])
    pos = torch.tensor([4.0, 0.0, 0.3], device=dev)
    z = -pos / pos.norm()
    bk = torch.tensor([0.7], device=dev)
Kinv = torch.from_numpy(np.linalg.inv(K)).float().to(dev)
x = torch.linalg.cross(torch.tensor([0.0, 0.0, 1.0], device=dev), z); x = x / x.norm(); y = torch.linalg.cross(z, x)
rot = torch.stack([x, y, z], 1)
for chunk in (16384, 65536, H * W):
    out = evaluation.render_image(r, Kinv, pos, rot, H, W, bkgd=bk, chunk=chunk); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        out = evaluation.render_image(r, Kinv, pos, rot, H, W, bkgd=bk, chunk=chunk)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"chunk {chunk:7d}: {dt * 1e3:7.2f} ms / image = {H * W / dt / 1e6:6.2f} M rays/s   (mean intensity {float(out[0].mean()):.4f})")
