"""Encoder forward at config-B size under three workgroup orders (REN_KNOB_HG_VARIANT): level-major (default: every workgroup in
flight gathers from ONE level's <= 4 MiB slice), XCD-affine (level l on XCD l % 8), level-inner (the 16 levels of a sample chunk
together: what a fused per-sample encode + MLP kernel would do to the L2s).  GPU only."""
import os, sys, math
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robust_e_nerf_amd import ops, engine
dev = "cuda:0"
R, S = int(os.environ.get("RAYS", 131072)), 128
g = torch.Generator().manual_seed(0)
fld = engine.NGPField(dev)
fld.table.copy_(((torch.rand(fld.n_table, generator=g) * 2 - 1) * 0.1).to(dev))
ang = torch.rand(R, generator=g) * 2 * math.pi
o = torch.stack([4 * torch.cos(ang), 4 * torch.sin(ang), torch.rand(R, generator=g) - 0.5], -1)
d = (torch.rand(R, 3, generator=g) - 0.5) * 1.6 - o
d = d / d.norm(dim=-1, keepdim=True)
o, d = o.float().to(dev).contiguous(), d.float().to(dev).contiguous()
r = engine.Renderer(fld, engine.RenderCfg(sampler="uniform", n_uniform=S))
pk = r.sample(o, d, torch.rand(R, device=dev), True)
n = pk.n
feat = torch.empty(ops.n_blocks32(n) * ops.FRAG_FLOATS_PER_BLOCK, device=dev)
ref = None
for name, v in (("level-major (default)", 0), ("XCD-affine", 1), ("level-inner", 4), ("level-major (default)", 0), ("level-inner", 4)):
    with ops.knob("hg_variant", v):
        run = lambda: ops.hashgrid_fwd(fld.grid, fld.table, scene=r.scene, rays=(o, d), samples=(pk.ray_indices, pk.t_starts, pk.t_ends),
                                       n=n, layout=1, out=feat)
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            run()
        e1.record(); torch.cuda.synchronize()
    if ref is None:
        ref = feat.clone()
    print(f"{name:24s} n = {n / 1e6:.1f} M  {e0.elapsed_time(e1) / 5:7.3f} ms   bit-identical: {bool(torch.equal(feat, ref))}", flush=True)
