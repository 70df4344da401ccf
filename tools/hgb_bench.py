"""Time variants of the binned hash-grid backward (single-file builds of csrc/ren_hashgrid_binned.hip). GPU only."""
import ctypes, os, sys, math
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robust_e_nerf_amd import _lib, ops, engine
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
P = ops._ptr
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
ref = None
# REN_AB="hgb_scatter=1,2": time the default library once per value of that knob (A/B of kernel generations inside one build)
ab = os.environ.get("REN_AB")
variants = [(p_, None) for p_ in (sys.argv[1:] or [_lib.LIB_PATH])]
if ab:
    kname, vals = ab.split("=")
    variants = [(_lib.LIB_PATH, (ops.KNOBS[kname], int(v))) for v in vals.split(",")]
for path, kv in variants:
    lib = ctypes.CDLL(os.path.abspath(path))
    if kv is not None:
        lib.ren_set_knob(kv[0], kv[1])
    for name in ("ren_hashgrid_bwd_binned", "ren_hashgrid_bwd_binned_workspace_bytes"):
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = _lib.SIGNATURES[name]
    ws = torch.empty(int(lib.ren_hashgrid_bwd_binned_workspace_bytes(n)), device=dev, dtype=torch.uint8)
    run = lambda: lib.ren_hashgrid_bwd_binned(ctypes.byref(grid), P(gt), None, ctypes.byref(r.scene), P(o), P(d),
                                              P(pk.ray_indices), P(pk.t_starts), P(pk.t_ends), n, 1, P(dfeat), P(ws), None, st)
    assert run() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        run()
    e1.record(); torch.cuda.synchronize()
    gt.zero_(); run(); torch.cuda.synchronize()
    msg = ""
    if ref is None:
        ref = gt.clone()
    else:
        msg = "  max|diff| vs first %.2e (max %.2e)" % (float((gt - ref).abs().max()), float(ref.abs().max()))
    print(f"{os.path.basename(path):24s} {'' if kv is None else 'knob=%d ' % kv[1]}{e0.elapsed_time(e1) / 3:7.2f} ms{msg}", flush=True)
    del ws
