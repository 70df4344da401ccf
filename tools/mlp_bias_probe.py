"""Are the element-wise errors of the matrix-core MLP kernels BIASED?  Mean and rms of (kernel - float64) for the data
gradients of the x kernels (bf16 MFMA, six-term split) and of the exact-f32 MFMA kernels.  GPU only.
A biased error of 1e-7 per element is invisible element-wise but grows like N (not sqrt N) in the sums over samples that
the bias / weight gradients are."""
import ctypes, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robust_e_nerf_amd import _lib, ops, tcnn_api

dev, n, C = "cuda:0", int(os.environ.get("N", 1 << 20)), 1
lib = _lib.load()
P = ops._ptr
gen = torch.Generator(device=dev).manual_seed(0)
nb = ops.n_blocks32(n)
feat = torch.randn(nb * 1024, device=dev, generator=gen) * 0.1
x = torch.rand(n, 3, device=dev, generator=gen) * 2 - 1
d = torch.randn(n, 3, device=dev, generator=gen); d = d / d.norm(dim=-1, keepdim=True)
params = torch.randn(9360 + 65 * C, device=dev, generator=gen) * 0.15
scene = ops.make_scene_desc([-1.5] * 3 + [1.5] * 3, 0)
st = ops._stream()
d_rgb, d_sig = torch.randn(n, C, device=dev, generator=gen), torch.randn(n, device=dev, generator=gen)


def sh16(d):
    x, y, z = d.unbind(-1); xy, xz, yz, x2, y2, z2 = x * y, x * z, y * z, x * x, y * y, z * z
    return torch.stack([torch.full_like(x, 0.28209479177387814), -0.48860251190291987 * y, 0.48860251190291987 * z,
                        -0.48860251190291987 * x, 1.0925484305920792 * xy, -1.0925484305920792 * yz,
                        0.94617469575755997 * z2 - 0.31539156525251999, -1.0925484305920792 * xz,
                        0.54627421529603959 * x2 - 0.54627421529603959 * y2, 0.59004358992664352 * y * (-3.0 * x2 + y2),
                        2.8906114426405538 * xy * z, 0.45704579946446572 * y * (1.0 - 5.0 * z2),
                        0.3731763325901154 * z * (5.0 * z2 - 3.0), 0.45704579946446572 * x * (1.0 - 5.0 * z2),
                        1.4453057213202769 * z * (x2 - y2), 0.59004358992664352 * x * (-x2 + 3.0 * y2)], -1)


sp = torch.nn.functional.softplus
W = {k: params[o: o + math.prod(s)].view(s).double() for k, (o, s) in ops.mlp_slices(C).items()}
e = tcnn_api._to_rows(feat, n).double().requires_grad_()
h = sp(e @ W["base.w0"].T + W["base.b0"], beta=100)
raw = (h @ W["base.wo"].T + W["base.bo"]).requires_grad_()
raw.retain_grad()
hin = torch.cat([sh16(d.double()), raw[:, 1:]], -1)
p = sp(hin @ W["head.w0"].T + W["head.b0"], beta=100)
q = sp(p @ W["head.w1"].T + W["head.b1"], beta=100)
rgb = sp(q @ W["head.wo"].T + W["head.bo"], beta=1)
sigma = torch.exp(raw[:, 0] - 1)
((rgb * d_rgb.double()).sum() + (sigma * d_sig.double()).sum()).backward()
ref_df, ref_db = e.grad, raw.grad                          # (n, 32), (n, 16)


def stats(name, got, ref):
    err = (got.double() - ref)
    scale = ref.abs().max()
    print(f"{name:34s} max|err|/max {float(err.abs().max() / scale):.2e}  mean(err)/rms(err) {float(err.mean() / err.pow(2).mean().sqrt()):+.3f}  "
          f"rms(err)/rms(ref) {float(err.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()):.2e}  "
          f"sum(err)/sum|ref| {float(err.sum() / ref.abs().sum()):+.2e}")
    # per-column: the bias gradient is the column sum
    cs_err, cs_ref = err.sum(0), ref.sum(0)
    print(f"{'':34s} column sums: max|err| / max|ref| {float(cs_err.abs().max() / cs_ref.abs().max()):.2e}   signs of the column-sum errors: "
          f"{int((cs_err > 0).sum())}+ / {int((cs_err < 0).sum())}-")


def base_rows(b):                                         # saved base-output layout [block][8][64] -> (n, 16)
    return b.view(-1, 8, 2, 32).permute(0, 3, 1, 2)       # [blk][sample][g][hi]: neuron rowc(g) + 4 hi, g < 8


def unperm(b):
    t = b.view(-1, 8, 2, 32)                              # [blk][g][hi][sample]
    out = torch.empty(t.shape[0], 32, 16, device=b.device)
    for g in range(8):
        for hi in range(2):
            out[:, :, (g & 3) + 8 * (g >> 2) + 4 * hi] = t[:, g, hi, :]
    return out.reshape(-1, 16)[:n]


for label, mode in (("exact-f32 MFMA kernels", 0), ("x kernels (bf16 MFMA, 6 terms)", 6)):
    rgb_k, sig_k = torch.empty(n, C, device=dev), torch.empty(n, device=dev)
    base_k = torch.empty(nb * 512, device=dev)
    acts = torch.empty(int(lib.ren_mlp_act_save_floats(n)), device=dev)
    d_base, dfeat, gm = torch.empty(nb * 512, device=dev), torch.empty(nb * 1024, device=dev), torch.zeros_like(params)
    if mode == 0:
        lib.ren_mlp_fwd_save(P(params), C, 0, 0, P(feat), ctypes.byref(scene), P(x), P(d), None, None, None, None, None, n,
                             P(rgb_k), P(sig_k), P(base_k), P(acts), st)
        ws = torch.empty(int(lib.ren_mlp_bwd_workspace_floats(C)), device=dev)
        lib.ren_mlp_bwd_saved(P(params), C, 0, 0, P(feat), P(base_k), P(acts), ctypes.byref(scene), P(x), P(d), None, None, None,
                              None, None, n, P(rgb_k), P(d_rgb), P(d_sig), P(d_base), P(dfeat), P(gm), P(ws), st)
    else:
        lib.ren_mlp_fwd_x(P(params), C, 0, 6, P(feat), ctypes.byref(scene), P(x), P(d), None, None, None, None, None, n, 0,
                          P(rgb_k), P(sig_k), P(base_k), None, None, st)
        ws = torch.empty(int(lib.ren_mlp_bwd_x_workspace_floats(C)), device=dev)
        lib.ren_mlp_bwd_x(P(params), C, 0, 6, P(feat), P(base_k), None, ctypes.byref(scene), P(x), P(d), None, None, None,
                          None, None, n, P(rgb_k), P(d_rgb), P(d_sig), P(d_base), P(dfeat), P(gm), P(ws), 0, None, st)
    torch.cuda.synchronize()
    print("==", label)
    stats("forward rgb", rgb_k, rgb.detach())
    stats("forward base_out (raw)", unperm(base_k), raw.detach())
    stats("d base_out (head data gradient)", unperm(d_base), ref_db)
    stats("d feat (base data gradient)", tcnn_api._to_rows(dfeat, n), ref_df)
