"""Micro-benchmark of the fused-MLP kernels at config-B size (GPU only; tuning aid).

usage: python tools/mlp_bench.py [lib.so ...]   (default: the in-tree library)
Each library only has to export ren_mlp_fwd / ren_mlp_bwd / ren_mlp_bwd_workspace_floats, so
single-file builds of csrc/ren_mlp.hip with different -D knobs can be compared side by side.
"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robust_e_nerf_amd import _lib, ops

dev = "cuda:0"
n = int(os.environ.get("N", 131072 * 128))
C = int(os.environ.get("C", 1))
nb = ops.n_blocks32(n)
g = torch.Generator(device=dev).manual_seed(0)
feat = torch.randn(nb * 1024, device=dev, generator=g) * 0.1
x = torch.rand(n, 3, device=dev, generator=g) * 2 - 1
d = torch.randn(n, 3, device=dev, generator=g)
d = d / d.norm(dim=-1, keepdim=True)
params = torch.randn(9360 + 65 * C, device=dev, generator=g) * 0.15
scene = _lib.SceneDesc()
for i, v in enumerate([-1.5, -1.5, -1.5, 1.5, 1.5, 1.5]):
    scene.aabb[i] = v
scene.contraction_type = 0
P = ops._ptr
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


d_rgb, d_sig = torch.randn(n, C, device=dev, generator=g), torch.randn(n, device=dev, generator=g)
ref = None
for path in (sys.argv[1:] or [_lib.LIB_PATH]):
    lib = ctypes.CDLL(os.path.abspath(path))
    for name in ("ren_mlp_fwd", "ren_mlp_bwd", "ren_mlp_bwd_workspace_floats"):
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = _lib.SIGNATURES[name]
    rgb, sigma = torch.empty(n, C, device=dev), torch.empty(n, device=dev)
    base = torch.empty(nb * 512, device=dev)
    fwd = lambda: lib.ren_mlp_fwd(P(params), C, 0, P(feat), ctypes.byref(scene), P(x), P(d), None, None, None, None,
                                  None, n, 0, P(rgb), P(sigma), P(base), st)
    assert fwd() == 0
    tf = timeit(fwd)
    d_base, dfeat = torch.empty(nb * 512, device=dev), torch.empty(nb * 1024, device=dev)
    gp = torch.zeros_like(params)
    ws = torch.empty(int(lib.ren_mlp_bwd_workspace_floats(C)), device=dev)
    bwd = lambda: lib.ren_mlp_bwd(P(params), C, 0, P(feat), P(base), ctypes.byref(scene), P(x), P(d), None, None, None,
                                  None, None, n, P(rgb), P(d_rgb), P(d_sig), P(d_base), P(dfeat), P(gp), P(ws), st)
    assert bwd() == 0
    tb = timeit(bwd)
    gp.zero_(); bwd(); torch.cuda.synchronize()
    out = (rgb.clone(), sigma.clone(), dfeat.clone(), gp.clone())
    msg = ""
    if ref is None:
        ref = out
    else:
        rel = lambda a, b: float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
        msg = "  rel-to-first: rgb %.1e sigma %.1e dfeat %.1e gparams %.1e" % tuple(rel(a, b) for a, b in zip(out, ref))
    print(f"{os.path.basename(path):28s} fwd {tf:6.2f} ms  bwd {tb:6.2f} ms{msg}", flush=True)
