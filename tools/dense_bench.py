"""Times ren_dense_fwd / ren_dense_bwd_data / ren_dense_bwd_weight on one 256 -> 256 trunk layer (arch mlp) and checks
them against float64 matmuls.   python tools/dense_bench.py [--n 1048576] [--mode 6|1|0]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from robust_e_nerf_amd import _lib                                          # noqa: E402
from robust_e_nerf_amd.ops import _ptr, _stream                             # noqa: E402
from robust_e_nerf_amd._lib import check                                    # noqa: E402


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1 << 20)
    ap.add_argument("--mode", type=int, default=6)
    ap.add_argument("--shapes", default="256x256,256x320,128x288,256x64")
    a = ap.parse_args()
    lib = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    n = a.n
    for shp in a.shapes.split(","):
        o, i = (int(v) for v in shp.split("x"))
        ldx = (i + 31) // 32 * 32
        X = torch.zeros(n, ldx)
        X[:, :i] = torch.randn(n, i, generator=g)
        W = torch.randn(o, i, generator=g) / i ** 0.5
        b = torch.randn(o, generator=g) * 0.1
        X, W, b = X.to(dev), W.to(dev), b.to(dev)
        Y = torch.empty(n, o, device=dev)
        fwd = lambda: check(lib.ren_dense_fwd(_ptr(X), ldx, _ptr(W), _ptr(b), o, i, 1 | (a.mode << 8), None, _ptr(Y), o, n,
                                              _stream()), "fwd")
        t_f = timeit(fwd)
        m = min(n, 4096)
        z = X[:m, :i].double() @ W.double().t() + b.double()
        ref = torch.where(z * 100 > 20, z, torch.log1p(torch.exp(z * 100)) / 100)
        err_f = float((Y[:m].double() - ref).abs().max() / ref.abs().max())
        dZ = torch.randn(n, o, generator=g).to(dev)
        ldo = (o + 31) // 32 * 32
        if ldo != o:
            dZp = torch.zeros(n, ldo, device=dev); dZp[:, :o] = dZ; dZ = dZp
        i4 = i // 4 * 4
        dX = torch.empty(n, i4, device=dev)
        Yp = torch.rand(n, i4, device=dev) * 0.05
        bwd = lambda: check(lib.ren_dense_bwd_data(_ptr(dZ), ldo, _ptr(W), o, i, i4, 1 | (a.mode << 8), _ptr(Yp), i4, 0, _ptr(dX),
                                                   i4, n, _stream()), "bwd")
        t_b = timeit(bwd)
        refb = (dZ[:m, :o].double() @ W.double())[:, :i4] * (1.0 - torch.exp(-100.0 * Yp[:m].double()))
        err_b = float((dX[:m].double() - refb).abs().max() / refb.abs().max())
        splits = int(os.environ.get("DW_SPLITS", 256))
        ws = torch.empty(int(lib.ren_dense_bwd_weight_workspace_floats(o, i, splits)), device=dev)
        gW, gb = torch.zeros(o, i, device=dev), torch.zeros(o, device=dev)
        dw = lambda: check(lib.ren_dense_bwd_weight(_ptr(dZ), ldo, _ptr(X), ldx, o, i, n, splits | (a.mode << 16), _ptr(gW), _ptr(gb),
                                                    _ptr(ws), _stream()), "dw")
        gW.zero_(); gb.zero_(); dw(); torch.cuda.synchronize()
        refw = dZ[:, :o].double().t() @ X[:, :i].double()
        err_w = float((gW.double() - refw).abs().max() / refw.abs().max())
        t_w = timeit(dw)
        fl = 2.0 * n * o * i
        print(f"{o}x{i} n={n} mode={a.mode}: fwd {t_f:.3f} ms ({fl / t_f / 1e9:.0f} TF/s alg, err {err_f:.1e})  "
              f"bwd_data {t_b:.3f} ms ({fl / t_b / 1e9:.0f} TF/s, err {err_b:.1e})  bwd_weight {t_w:.3f} ms ({fl / t_w / 1e9:.0f} TF/s, "
              f"err {err_w:.1e})", flush=True)


if __name__ == "__main__":
    main()
