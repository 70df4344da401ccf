"""Do the two-stream chunked forward (encoder beside MLP) and the chunked backward repeat the single-stream results?
Forward outputs are compared bit for bit (no atomics in the forward), gradients to the scatter's float-atomic noise.  GPU only.
   python tools/chunk_stress.py [repeats]"""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robust_e_nerf_amd import ops, engine
dev = "cuda:0"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
R, S = 131072, 128
g = torch.Generator().manual_seed(0)
ang = torch.rand(R, generator=g) * 2 * math.pi
o = torch.stack([4 * torch.cos(ang), 4 * torch.sin(ang), torch.rand(R, generator=g) - 0.5], -1)
d = (torch.rand(R, 3, generator=g) - 0.5) * 1.6 - o
d = d / d.norm(dim=-1, keepdim=True)
o, d = o.float().to(dev).contiguous(), d.float().to(dev).contiguous()
jit = torch.rand(R, generator=g).to(dev)
gcol = torch.randn(R, 1, generator=g).to(dev)
fld = engine.NGPField(dev)
def run(fwd_chunks, bwd_chunks):
    r = engine.Renderer(fld, engine.RenderCfg(sampler="uniform", n_uniform=S, fwd_chunks=fwd_chunks, bwd_chunks=bwd_chunks))
    fld.grad.zero_()
    colors, opac, depth, ctx = r.forward(o, d, jit, None, training=True, save=True)
    r.backward(ctx, gcol, final=True)
    torch.cuda.synchronize()
    return colors.clone(), opac.clone(), fld.grad.clone()
c0, a0, g0 = run(1, 1)
c1, a1, g1 = run(1, 1)
print("single stream twice: colors equal", bool(torch.equal(c0, c1)), "grad max diff %.3e of %.3e" % (float((g0 - g1).abs().max()), float(g0.abs().max())), flush=True)
bad = 0
worst = 0.0
for k in range(reps):
    for fc, bc in ((8, 1), (8, 6)):
        c, a, gg = run(fc, bc)
        ok = torch.equal(c, c0) and torch.equal(a, a0)
        dg = float((gg - g0).abs().max())
        worst = max(worst, dg)
        if not ok or dg > 1e-4 * float(g0.abs().max()):
            bad += 1
            print("MISMATCH rep", k, "chunks", fc, bc, "colors differing:", int((c != c0).sum()), "grad diff %.3e" % dg, flush=True)
print("repeats", reps, "x 2 configurations: mismatches", bad, "| worst grad diff %.3e of %.3e" % (worst, float(g0.abs().max())))
