"""GPU: the fused arch-mlp trunk kernels (csrc/ren_trunk.hip) against a float64 torch model of the eight layers, and their
timing.  python tools/trunk_check.py [--n 100000] [--mode 6] [--time]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robust_e_nerf_amd import _lib, vanilla                                  # noqa: E402
from robust_e_nerf_amd._lib import check                                     # noqa: E402
from robust_e_nerf_amd.ops import _ptr, _stream                              # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10000)
    ap.add_argument("--mode", type=int, default=6)
    ap.add_argument("--time", action="store_true")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--lib", default=None, help="single-file build of csrc/ren_trunk.hip to take the ren_trunk_* entry points from")
    a = ap.parse_args()
    if a.lib:
        import ctypes
        var, lib = ctypes.CDLL(os.path.abspath(a.lib)), _lib.load()
        for name, (res, args) in _lib.SIGNATURES.items():
            if name.startswith("ren_trunk_"):
                fn = getattr(var, name)
                fn.restype, fn.argtypes = res, args
                setattr(lib, name, fn)
    dev = "cuda:0"
    torch.manual_seed(0)
    fld = vanilla.VanillaField(dev)
    for name, o, i in fld.layers:                                            # torch nn.Linear default init range
        k = 1.0 / i ** 0.5
        fld.w[name].uniform_(-k, k).mul_(a.scale)
        fld.b[name].uniform_(-k, k)
    n, mode = a.n, a.mode
    n_pad = (n + 31) // 32 * 32
    enc = torch.zeros(n_pad, 64, device=dev)
    enc[:, :63] = torch.rand(n_pad, 63, device=dev) * 2 - 1
    tr = vanilla.Trunk(fld, mode)
    tr.prep()
    saved = tr.new_saved(n)
    h7 = torch.empty(n_pad, 256, device=dev)
    tr.forward(enc, n, saved, h7)
    # float64 model
    W = [fld.w[f"mlp.base.hidden_layers.{i}"].double().requires_grad_() for i in range(8)]
    Bv = [fld.b[f"mlp.base.hidden_layers.{i}"].double().requires_grad_() for i in range(8)]
    e = enc[:n, :63].double()
    x, hs, zs = e, [], []
    for l in range(8):
        inp = torch.cat([x, e], 1) if l == 5 else x
        z = inp @ W[l].T + Bv[l]
        z.retain_grad()
        x = torch.nn.functional.softplus(z, beta=100)
        zs.append(z)
        hs.append(x)
    rel = lambda got, ref: float((got.double() - ref).abs().max() / ref.abs().max())
    print(f"mode {mode} n {n}: h7 max err / max {rel(h7[:n], hs[7]):.2e}")
    acts = tr.decode(saved, n)
    for l in range(8):
        print(f"  saved h{l}: {rel(acts[l], hs[l]):.2e}", end="")
    print()
    G = torch.zeros(n_pad, 256, device=dev)
    G[:n] = torch.randn(n, 256, device=dev)
    (zs[7] * G[:n].double()).sum().backward()
    dz = tr.new_saved(n)
    tr.backward(G, n, saved, dz)
    dzr = tr.decode(dz, n)
    for l in range(8):
        print(f"  dz{l}: {rel(dzr[l], zs[l].grad):.2e}", end="")
    print()
    fld.grad.zero_()
    tr.backward_weight(dz, saved, enc, n)
    for l in range(8):
        name = f"mlp.base.hidden_layers.{l}"
        print(f"  dW{l}: {rel(fld.gw[name], W[l].grad):.2e} db{l}: {rel(fld.gb[name], Bv[l].grad):.2e}")
    if a.time:
        ev = lambda: torch.cuda.Event(enable_timing=True)
        for what, fn in (("prep", tr.prep), ("fwd", lambda: tr.forward(enc, n, saved, h7)), ("fwd nosave", lambda: tr.forward(enc, n, None, h7)),
                         ("bwd", lambda: tr.backward(G, n, saved, dz)), ("dw", lambda: tr.backward_weight(dz, saved, enc, n))):
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
