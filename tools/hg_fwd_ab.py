"""A/B of single-file builds of csrc/ren_hashgrid.hip: forward time at config-B size (fragment layout).  GPU only.
usage: python tools/hg_fwd_ab.py build_variants/a.so build_variants/b.so ..."""
import ctypes, os, sys, math
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robust_e_nerf_amd import _lib, ops, engine
dev = "cuda:0"
R, S = int(os.environ.get("RAYS", 131072)), 128
grid, n_table = ops.make_grid_desc()
g = torch.Generator().manual_seed(0)
table = ((torch.rand(n_table, generator=g) * 2 - 1) * 0.1).to(dev)
ang = torch.rand(R, generator=g) * 2 * math.pi
o = torch.stack([4 * torch.cos(ang), 4 * torch.sin(ang), torch.rand(R, generator=g) - 0.5], -1)
d = (torch.rand(R, 3, generator=g) - 0.5) * 1.6 - o
d = d / d.norm(dim=-1, keepdim=True)
o, d = o.float().to(dev).contiguous(), d.float().to(dev).contiguous()
r = engine.Renderer(engine.NGPField(dev), engine.RenderCfg(sampler="uniform", n_uniform=S))
pk = r.sample(o, d, torch.rand(R, device=dev), True)
n = pk.n
feat = torch.empty(ops.n_blocks32(n) * ops.FRAG_FLOATS_PER_BLOCK, device=dev)
P = ops._ptr
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
ref = None
for path in (sys.argv[1:] or [_lib.LIB_PATH]) * 2:
    lib = ctypes.CDLL(os.path.abspath(path))
    fn = lib.ren_hashgrid_fwd
    fn.restype, fn.argtypes = _lib.SIGNATURES["ren_hashgrid_fwd"]
    run = lambda: fn(ctypes.byref(grid), P(table), None, ctypes.byref(r.scene), P(o), P(d), P(pk.ray_indices), P(pk.t_starts),
                     P(pk.t_ends), n, 1, P(feat), None, st)
    assert run() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run()
    e1.record(); torch.cuda.synchronize()
    if ref is None:
        ref = feat.clone()
    print(f"{os.path.basename(path):20s} {e0.elapsed_time(e1) / 5:7.3f} ms   bit-identical to first: {bool(torch.equal(feat, ref))}", flush=True)
