"""Randomised bit-exactness check of the speculative marcher (and the interval cache) against the sequential kernel:
random occupancy grids of several densities and resolutions, all contraction types, cone angles, near/far planes."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robust_e_nerf_amd import _lib, ops
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
aabb = (-1.5, -1.5, -1.5, 1.5, 1.5, 1.5)
bad = 0
for trial in range(48):
    R = int(torch.randint(1, 6000, (1,), generator=g))
    res = (16, 32, 128)[trial % 3]
    ct = trial % 3 if trial % 2 else 0
    dens = (0.02, 0.3, 0.9, 1.0)[trial % 4]
    binary = (torch.rand(res ** 3, generator=g) < dens).to(torch.uint8).to(dev)
    ang = torch.rand(R, generator=g) * 2 * math.pi
    o = torch.stack([4 * torch.cos(ang), 4 * torch.sin(ang), torch.rand(R, generator=g) * 2 - 1], -1).float()
    d = (torch.rand(R, 3, generator=g) - 0.5) * 2.5 - o
    d = (d / d.norm(dim=-1, keepdim=True)).float()
    o, d = o.to(dev).contiguous(), d.to(dev).contiguous()
    cone = (0.0, 0.004, 0.01)[trial % 3]
    step = math.sqrt(3) * 3 / 1024
    if ct == 0:
        tmin, tmax = ops.ray_aabb_intersect(o, d, aabb, None, None)
    else:
        tmin, tmax = torch.full((R,), 0.05, device=dev), torch.full((R,), 8.0, device=dev)
    jit = torch.rand(R, generator=g).to(dev)
    args = (o, d, tmin, tmax, jit, aabb, (res,) * 3, binary, ct, step, cone, 0, 0)
    args_fast = args[:11] + (ops.MARCH_VERIFIED_DIV if os.environ.get("FUZZ_FASTDIV") == "1" and ops.march_div_check(aabb, dev) else 0,) + args[12:]
    out = {}
    for seq in ("1", "0"):                 # sequential kernel | FUZZ_WIDTH (0: the default policy; 2 / 4 / 8 / 16 lanes; 32: look-ahead)
        _lib.load().ren_set_knob(ops.KNOBS["march_sequential"], 1 if seq == "1" else int(os.environ.get("FUZZ_WIDTH", "0")))
        ar = args if seq == "1" else args_fast                  # (the reference run: sequential kernel, IEEE divisions)
        counts = ops.ray_march_count(*ar)
        offsets, total = ops.exclusive_scan(counts)
        n = int(total)
        ri, ts, te = ops.ray_march_write(*ar, offsets, n)
        cache = torch.empty(R, 16, 2, device=dev)
        counts_c = ops.ray_march_count(*ar, cache=cache)
        ri_c, ts_c, te_c = ops.ray_march_write(*ar, offsets, n, counts=counts_c, cache=cache)
        torch.cuda.synchronize()
        same_c = torch.equal(counts_c, counts) and torch.equal(ri_c, ri) and torch.equal(ts_c, ts) and torch.equal(te_c, te)
        out[seq] = (counts, ri, ts, te, same_c)
    _lib.load().ren_set_knob(ops.KNOBS["march_sequential"], 0)
    a, b = out["1"], out["0"]
    ok = all(torch.equal(a[k], b[k]) for k in range(4)) and a[4] and b[4]
    bad += not ok
    print(f"trial {trial:2d} R={R:5d} res={res:3d} ct={ct} density={dens} cone={cone}: samples {int(a[0].sum()):8d} {'ok' if ok else 'MISMATCH'}", flush=True)
assert bad == 0
print("all identical")
