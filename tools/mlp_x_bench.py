"""Time and check variants of the matrix-core NGP MLP kernels (single-file builds of csrc/ren_mlp_x.hip) against the
exact-f32 MFMA kernels of the main library.  GPU only.
usage: python tools/mlp_x_bench.py [variant.so ...]      (default: the main library)    env: N (samples), MODE (6 | 1)"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robust_e_nerf_amd import _lib, ops

dev = "cuda:0"
n = int(os.environ.get("N", 8388608))
mode = int(os.environ.get("MODE", 6))
C = 1
g = torch.Generator().manual_seed(0)
x = (torch.rand(n, 3, generator=g) * 2.6 - 1.3).to(dev)
d = torch.randn(n, 3, generator=g)
d = (d / d.norm(dim=-1, keepdim=True)).to(dev)
nb = ops.n_blocks32(n)
feat = (torch.rand(nb * 1024, generator=g) - 0.5).to(dev)
npar = ops.mlp_param_count(C)
params = ((torch.rand(npar, generator=g) - 0.5) * 0.5).to(dev)
scene = ops.make_scene_desc([-1.5, -1.5, -1.5, 1.5, 1.5, 1.5], 0)
d_rgb = torch.randn(n, C, generator=g).to(dev)
d_sigma = (torch.randn(n, generator=g) * 0.1).to(dev)
P = ops._ptr
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

# reference: exact f32 MFMA kernels
rgb0, sigma0, base0 = ops.mlp_fwd(params, C, feat, scene, x_world=x, dirs=d, n=n, save_base=True)
g0 = torch.zeros(npar, device=dev)
ws0 = torch.empty(ops.mlp_bwd_workspace_floats(C), device=dev)
df0 = ops.mlp_bwd(params, C, feat, base0, scene, x_world=x, dirs=d, n=n, rgb=rgb0, d_rgb=d_rgb, d_sigma=d_sigma,
                  grad_mlp_params=g0, workspace=ws0)
torch.cuda.synchronize()
sl = ops.mlp_slices(C)


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for path in (sys.argv[1:] or [_lib.LIB_PATH]):
    lib = ctypes.CDLL(os.path.abspath(path))
    for name in ("ren_mlp_fwd_x", "ren_mlp_bwd_x", "ren_mlp_bwd_x_workspace_floats"):
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = _lib.SIGNATURES[name]
    rgb = torch.empty(n, C, device=dev); sigma = torch.empty(n, device=dev)
    base = torch.empty(nb * ops.BASE_FLOATS_PER_BLOCK, device=dev)
    fwd = lambda: lib.ren_mlp_fwd_x(P(params), C, 0, mode, P(feat), ctypes.byref(scene), P(x), P(d), None, None, None, None,
                                    None, n, 0, P(rgb), P(sigma), P(base), None, None, st)
    assert fwd() == 0
    t_f = timeit(fwd)
    gm = torch.zeros(npar, device=dev)
    ws = torch.empty(int(lib.ren_mlp_bwd_x_workspace_floats(C)), device=dev)
    d_base = torch.empty(nb * ops.BASE_FLOATS_PER_BLOCK, device=dev)
    dfeat = torch.empty(nb * ops.FRAG_FLOATS_PER_BLOCK, device=dev)
    bwd = lambda: lib.ren_mlp_bwd_x(P(params), C, 0, mode, P(feat), P(base), None, ctypes.byref(scene), P(x), P(d), None, None,
                                    None, None, None, n, P(rgb), P(d_rgb), P(d_sigma), P(d_base), P(dfeat), P(gm), P(ws), 0, None, st)
    assert bwd() == 0
    t_b = timeit(bwd)
    gm.zero_(); bwd(); torch.cuda.synchronize()
    import math
    errs = " ".join(f"{k}:{rel(gm[a:a + math.prod(sh)], g0[a:a + math.prod(sh)]):.1e}" for k, (a, sh) in sl.items())
    print(f"{os.path.basename(path):22s} fwd {t_f:6.3f} ms  bwd {t_b:6.3f} ms | rgb {rel(rgb, rgb0):.1e} sigma {rel(sigma, sigma0):.1e} "
          f"dfeat {rel(dfeat, df0):.1e} | dW {errs}", flush=True)
