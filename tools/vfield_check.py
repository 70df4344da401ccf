"""GPU: the fused arch-mlp field kernels (csrc/ren_vfield.hip) against a float64 torch model of the twelve layers, and their
timing.  python tools/vfield_check.py [--n 100000] [--mode 6] [--C 1] [--time] [--lib single-file-build.so]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robust_e_nerf_amd import _lib, vanilla                                  # noqa: E402


def reference(fld, enc, view, sel, n):
    """float64 model; returns outputs, pre-activations (with retain_grad) and the parameter leaves"""
    F = torch.nn.functional
    names = [f"mlp.base.hidden_layers.{i}" for i in range(8)] + ["mlp.sigma_layer.output_layer", "mlp.bottleneck_layer.output_layer",
                                                                 "mlp.rgb_layer.hidden_layers.0", "mlp.rgb_layer.output_layer"]
    W = [fld.w[k].double().requires_grad_() for k in names]
    Bv = [fld.b[k].double().requires_grad_() for k in names]
    e, v = enc[:n, :63].double(), view[:n, :27].double()
    x, hs, zs = e, [], []
    for l in range(8):
        z = (torch.cat([x, e], 1) if l == 5 else x) @ W[l].T + Bv[l]
        z.retain_grad()
        x = F.softplus(z, beta=100)
        zs.append(z)
        hs.append(x)
    zsig = x @ W[8].T + Bv[8]
    zsig.retain_grad()
    sigma = torch.where(sel[:n].bool(), torch.exp(zsig[:, 0] - 1), torch.zeros_like(zsig[:, 0]))
    bott = x @ W[9].T + Bv[9]
    bott.retain_grad()
    zr = torch.cat([bott, v], 1) @ W[10].T + Bv[10]
    zr.retain_grad()
    r = F.softplus(zr, beta=100)
    zo = r @ W[11].T + Bv[11]
    zo.retain_grad()
    rgb = F.softplus(zo)
    return dict(names=names, W=W, B=Bv, hs=hs, zs=zs, zsig=zsig, sigma=sigma, bott=bott, zr=zr, r=r, zo=zo, rgb=rgb)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10000)
    ap.add_argument("--mode", type=int, default=6)
    ap.add_argument("--C", type=int, default=1)
    ap.add_argument("--time", action="store_true")
    ap.add_argument("--lib", default=None, help="single-file build of csrc/ren_vfield.hip to take the ren_vanilla_{prep,fwd,bwd,..} entry points from")
    a = ap.parse_args()
    if a.lib:
        import ctypes
        var, lib = ctypes.CDLL(os.path.abspath(a.lib)), _lib.load()
        for name, (res, args) in _lib.SIGNATURES.items():
            if hasattr(var, name):
                fn = getattr(var, name)
                fn.restype, fn.argtypes = res, args
                setattr(lib, name, fn)
    dev = "cuda:0"
    torch.manual_seed(0)
    fld = vanilla.VanillaField(dev, a.C)
    for name, o, i in fld.layers:                                            # torch nn.Linear default init range
        k = 1.0 / i ** 0.5
        fld.w[name].uniform_(-k, k)
        fld.b[name].uniform_(-k, k)
    n, mode, C = a.n, a.mode, a.C
    ff = vanilla.FusedField(fld, mode)
    ff.prep()
    B = vanilla._Buffers(n, dev, C, full=True, backward=False, fused=ff, save=True)
    B.enc.zero_(); B.view.zero_()
    B.enc[:n, :63] = torch.rand(n, 63, device=dev) * 2 - 1
    B.view[:n, :27] = torch.rand(n, 27, device=dev) * 2 - 1
    B.sel[:n] = (torch.rand(n, device=dev) < 0.8).to(torch.uint8)
    ff.forward(B, True)
    R = reference(fld, B.enc, B.view, B.sel, n)
    rel = lambda got, ref: float((got.double() - ref.detach()).abs().max() / ref.detach().abs().max())
    print(f"mode {mode} n {n} C {C}: sigma {rel(B.sigma[:n], R['sigma']):.2e} rgb {rel(B.rgb4[:n, :C], R['rgb']):.2e}"
          f" rgb4 padding {float(B.rgb4[:n, C:].abs().max()) if C < 4 else 0.0}")
    sig_only = vanilla._Buffers(n, dev, C, full=False, backward=False, fused=ff)
    sig_only.enc.copy_(B.enc); sig_only.sel.copy_(B.sel)
    ff.forward(sig_only, False)
    print(f"  density-only launch vs full: {float((sig_only.sigma[:n] - B.sigma[:n]).abs().max()):.1e}")
    acts = ff.decode(B.saved, n)
    print("  saved: " + " ".join(f"h{l} {rel(acts[l], R['hs'][l]):.1e}" for l in range(8)) +
          f" bott {rel(acts[8], R['bott']):.1e} r {rel(acts[9][:, :128], R['r']):.1e}")
    dz_rgb, dz_sig = torch.zeros(B.n_pad, 32, device=dev), torch.zeros(B.n_pad, 32, device=dev)
    dz_rgb[:n, :C] = torch.randn(n, C, device=dev)
    dz_sig[:n, 0] = torch.randn(n, device=dev)
    ((R["zo"] * dz_rgb[:n, :C].double()).sum() + (R["zsig"][:, 0] * dz_sig[:n, 0].double()).sum()).backward()
    dz = ff.new_saved(n)
    ff.backward(dz_rgb, dz_sig, B, dz)
    dzr = ff.decode(dz, n)
    print("  dz: " + " ".join(f"{l} {rel(dzr[l], R['zs'][l].grad):.1e}" for l in range(8)) +
          f" bott {rel(dzr[8], R['bott'].grad):.1e} r {rel(dzr[9][:, :128], R['zr'].grad):.1e}")
    fld.grad.zero_()
    ff.backward_weight(dz_rgb, dz_sig, B, dz)
    print("  dW/db: " + " ".join(f"{k.split('.')[1][:4]}{k.split('.')[-1] if 'hidden' in k else ''} {rel(fld.gw[k], R['W'][i].grad):.1e}/{rel(fld.gb[k], R['B'][i].grad):.1e}"
                                  for i, k in enumerate(R["names"])))
    if a.time:
        ev = lambda: torch.cuda.Event(enable_timing=True)
        nosave = vanilla._Buffers(n, dev, C, full=True, backward=False, fused=ff)
        nosave.enc.copy_(B.enc); nosave.view.copy_(B.view); nosave.sel.copy_(B.sel)
        for what, fn in (("prep", ff.prep), ("fwd", lambda: ff.forward(B, True)), ("fwd nosave", lambda: ff.forward(nosave, True)),
                         ("fwd density", lambda: ff.forward(sig_only, False)), ("bwd", lambda: ff.backward(dz_rgb, dz_sig, B, dz)),
                         ("dw", lambda: ff.backward_weight(dz_rgb, dz_sig, B, dz))):
            for _ in range(2):
                fn()
            s, t = ev(), ev()
            s.record()
            for _ in range(5):
                fn()
            t.record()
            torch.cuda.synchronize()
            ms = s.elapsed_time(t) / 5
            print(f"  {what}: {ms:.3f} ms  ({n / ms / 1e3:.1f} M samples/s)")


if __name__ == "__main__":
    main()
