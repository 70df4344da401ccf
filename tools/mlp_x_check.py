"""Split-bf16 MLP kernels vs the exact-fp32 kernels: accuracy and time (GPU only; tuning aid)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robust_e_nerf_amd import _lib, ops
dev = "cuda:0"
n = int(os.environ.get("N", 131072 * 128))
C = 1
nb = ops.n_blocks32(n)
g = torch.Generator(device=dev).manual_seed(0)
feat = torch.randn(nb * 1024, device=dev, generator=g) * 0.1
x = torch.rand(n, 3, device=dev, generator=g) * 2 - 1
d = torch.randn(n, 3, device=dev, generator=g); d = d / d.norm(dim=-1, keepdim=True)
params = torch.randn(9360 + 65 * C, device=dev, generator=g) * 0.15
scene = ops.make_scene_desc([-1.5] * 3 + [1.5] * 3, 0)
lib = _lib.load(); P = ops._ptr
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
def outs():
    return (torch.empty(n, C, device=dev), torch.empty(n, device=dev), torch.empty(nb * 512, device=dev),
            torch.empty(int(lib.ren_mlp_act_save_floats(n)), device=dev))
r0 = outs()
f32 = lambda: lib.ren_mlp_fwd_save(P(params), C, 0, 0, P(feat), ctypes.byref(scene), P(x), P(d), None, None, None, None, None, n,
                                   P(r0[0]), P(r0[1]), P(r0[2]), P(r0[3]), st)
assert f32() == 0
print(f"f32 MFMA fwd_save      {timeit(f32):6.2f} ms")
rel = lambda a, b: float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
for mode in (6, 1):
    r = outs()
    fx = lambda: lib.ren_mlp_fwd_x(P(params), C, 0, mode, P(feat), ctypes.byref(scene), P(x), P(d), None, None, None, None, None,
                                   n, 0, P(r[0]), P(r[1]), P(r[2]), P(r[3]), None, st)
    assert fx() == 0
    t = timeit(fx)
    print(f"bf16 MFMA mode {mode} fwd   {t:6.2f} ms   rel-to-f32: rgb %.2e sigma %.2e base %.2e acts %.2e" %
          tuple(rel(a, b) for a, b in zip(r, r0)))
    fi = lambda: lib.ren_mlp_fwd_x(P(params), C, 0, mode, P(feat), ctypes.byref(scene), P(x), P(d), None, None, None, None, None,
                                   n, 0, P(r[0]), P(r[1]), P(r[2]), None, None, st)
    print(f"   without activation save {timeit(fi):6.2f} ms")

# ---- backward
d_rgb, d_sig = torch.randn(n, C, device=dev, generator=g), torch.randn(n, device=dev, generator=g)
def bwd_outs():
    return torch.empty(nb * 512, device=dev), torch.empty(nb * 1024, device=dev), torch.zeros_like(params)
b0 = bwd_outs()
ws0 = torch.empty(int(lib.ren_mlp_bwd_workspace_floats(C)), device=dev)
fb = lambda: lib.ren_mlp_bwd_saved(P(params), C, 0, 0, P(feat), P(r0[2]), P(r0[3]), ctypes.byref(scene), P(x), P(d), None, None,
                                   None, None, None, n, P(r0[0]), P(d_rgb), P(d_sig), P(b0[0]), P(b0[1]), P(b0[2]), P(ws0), st)
assert fb() == 0
tb = timeit(fb)
b0[2].zero_(); fb(); torch.cuda.synchronize()
print(f"f32 MFMA bwd_saved     {tb:6.2f} ms")
for mode in (6, 1):
    r = outs()
    assert lib.ren_mlp_fwd_x(P(params), C, 0, mode, P(feat), ctypes.byref(scene), P(x), P(d), None, None, None, None, None,
                             n, 0, P(r[0]), P(r[1]), P(r[2]), P(r[3]), None, st) == 0
    b = bwd_outs()
    ws = torch.empty(int(lib.ren_mlp_bwd_x_workspace_floats(C)), device=dev)
    fx = lambda: lib.ren_mlp_bwd_x(P(params), C, 0, mode, P(feat), P(r[2]), P(r[3]), ctypes.byref(scene), P(x), P(d), None, None,
                                   None, None, None, n, P(r[0]), P(d_rgb), P(d_sig), P(b[0]), P(b[1]), P(b[2]), P(ws), 0, None, st)
    assert fx() == 0
    t = timeit(fx)
    b[2].zero_(); fx(); torch.cuda.synchronize()
    print(f"bf16 MFMA mode {mode} bwd   {t:6.2f} ms   rel-to-f32: d_base %.2e dfeat %.2e gparams %.2e" %
          tuple(rel(a, c) for a, c in zip(b, b0)))
    for k, (off, shape) in ops.mlp_slices(C).items():
        import math
        sl_ = slice(off, off + math.prod(shape))
        print(f"      {k:8s} %.2e" % rel(b[2][sl_], b0[2][sl_]), end="")
    print()
    # recompute variant (act_save = NULL): must equal the saved-activation backward bit for bit
    b2 = bwd_outs()
    fr = lambda: lib.ren_mlp_bwd_x(P(params), C, 0, mode, P(feat), P(r[2]), None, ctypes.byref(scene), P(x), P(d), None, None,
                                   None, None, None, n, P(r[0]), P(d_rgb), P(d_sig), P(b2[0]), P(b2[1]), P(b2[2]), P(ws), 0, None, st)
    assert fr() == 0
    t2 = timeit(fr)
    b2[2].zero_(); fr(); torch.cuda.synchronize()
    print(f"   recompute (no act load) {t2:6.2f} ms   vs saved: d_base %.2e dfeat %.2e gparams %.2e" %
          tuple(rel(a, c) for a, c in zip(b2, b)))
