"""Where do three prefetched steps first differ from three plain ones? (diagnosis of a flaky test; GPU only)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as T
from oracle import hashgrid
from robust_e_nerf_amd import ops, engine
g = T.load_golden("training_step_diff")
table = hashgrid.init_table(hashgrid.make_spec(), int(g["table_seed"]), float(g["table_scale"]), "mix32")
sampler = sys.argv[1] if len(sys.argv) > 1 else "occgrid"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
B = 4096
dev = T.dev
def run(use):
    tr, _ = T._trainer_from_golden(engine, g, table, sampler=sampler)
    tr.r.cfg.n_uniform = 32
    gen = torch.Generator().manual_seed(5)
    steps = []
    for i in range(4):
        nb = T._config_batch(B, 30 + i, int(g["tab_ts"][-1]))
        steps.append(({k: dev(v) for k, v in nb.items()}, dev(torch.rand(B, generator=gen)), dev(torch.rand(B, generator=gen))))
    torch.cuda.synchronize()
    res = []
    for i in range(3):
        pf = tr._prefetched
        fr = None
        def flat(front):
            st = front["begun"]
            d_ = dict(o=front["o"], d=front["d"], jitter=front["jitter"], ts=front["prep"]["ts"], target=front["prep"]["target_diff"],
                      counts=st["counts"], offsets=st["offsets"], total=st["total"], t_min=st["args"][2], t_max=st["args"][3])
            if st["cache"] is not None:
                d_["cache"] = st["cache"]
            return d_
        if pf is not None:
            pf[2].synchronize()
            live = flat(pf[1])
            fr = {k: v.clone() for k, v in live.items()}
            torch.cuda.synchronize()
        else:
            # the same front, computed in order on the main stream (reference for the prefetched one)
            f0 = tr._front(*steps[i])
            f0["begun"] = tr.r.sample_begin(f0["o"], f0["d"], f0["jitter"], True)
            fr = {k: v.clone() for k, v in flat(f0).items()}
            live = None
        loss, aux = tr.forward_backward(*steps[i])
        if live is not None:
            torch.cuda.synchronize()
            for k, v in live.items():
                if not torch.equal(v, fr[k]):
                    print("   step", i, "front tensor", k, "CHANGED after it was produced:", int((v != fr[k]).sum()), "of", v.numel(), flush=True)
        if use:
            assert tr.prefetch(*steps[i + 1])
        rec = dict(loss=float(loss), n=aux["n"], grad=tr.r.field.grad.clone(), small_grad=tr.small_grad.clone(), front=fr,
                   steps=steps, tr=tr)
        tr.optimizer_step()
        rec["flat"] = tr.r.field.flat.clone()
        rec["m"], rec["v"] = tr.m.clone(), tr.v.clone()
        res.append(rec)
    torch.cuda.synchronize()
    return res
ref = run(False)
ref2 = run(False)
def cmp(a, b, tag):
    for i in range(3):
        if a[i]["front"] is not None and b[i]["front"] is not None:
            for k in a[i]["front"]:
                x, y = a[i]["front"][k], b[i]["front"][k]
                if k == "cache":
                    continue
                if k == "d" and x.shape == y.shape and not torch.equal(x, y):
                    bad = (x != y).any(dim=1).nonzero().flatten()
                    print("      wrong rays:", bad.tolist()[:40], "| prefetched d:", x[bad[0]].tolist(), "in-order d:", y[bad[0]].tolist(),
                          "| o equal:", bool(torch.equal(a[i]["front"]["o"], b[i]["front"]["o"])), "ts equal:", bool(torch.equal(a[i]["front"]["ts"], b[i]["front"]["ts"])), flush=True)
                    # which pixel would give the prefetched direction?  compare with the direction of every other ray of the batch
                    tr_, steps_ = a[i]["tr"], a[i]["steps"]
                    ts_ = a[i]["front"]["ts"]
                    for kk in range(4):
                        for name in ("position",):
                            o_k, d_k = ops.pose_rays(ts_, steps_[kk][0][name].contiguous(), tr_.Kinv, tr_.tab_ts, tr_.tab_pos, tr_.tab_quat)
                            print("      d recomputed with steps[%d].position: matches prefetched on the bad rays: %s, on all rays: %s" % (
                                kk, bool(torch.equal(d_k[bad], x[bad])), bool(torch.equal(d_k, x))), flush=True)
                    _, rot_ = ops.trajectory(ts_.contiguous(), tr_.tab_ts, tr_.tab_pos, tr_.tab_quat)
                    K_ = torch.linalg.inv(tr_.Kinv.double().reshape(3, 3))
                    def implied(dd):
                        k = torch.einsum("nji,nj->ni", rot_[bad].double(), dd[bad].double())      # R^T d
                        q = k @ K_.T
                        return (q[:, :2] / q[:, 2:3])
                    pos_i = steps_[i][0]["position"]
                    true_px = pos_i[bad % pos_i.shape[0]]
                    print("      true px      :", [[round(v, 3) for v in r] for r in true_px.tolist()][:16], flush=True)
                    print("      implied (bad):", [[round(v, 3) for v in r] for r in implied(x).tolist()][:16], flush=True)
                    print("      implied (ok) :", [[round(v, 3) for v in r] for r in implied(y).tolist()][:4], flush=True)
                    pxs = steps_[i][0]["position"]
                    print("      address of this step's position tensor: %#x, bad px byte offset %d..%d; all position ptrs: %s" % (
                        pxs.data_ptr(), int(bad[0] % pxs.shape[0]) * 8, int(bad[-1] % pxs.shape[0]) * 8 + 8,
                        [hex(st_[0]["position"].data_ptr()) for st_ in steps_]), flush=True)
                    dots = (y @ x[bad[0]]).abs()
                    print("      closest in-order ray to the wrong direction: ray", int(dots.argmax()), "cos %.9f" % float(dots.max()), flush=True)
                if x.shape != y.shape or not torch.equal(x, y):
                    print("   ", tag, "step", i, "front", k, "differs from the in-order front:", "shape" if x.shape != y.shape else int((x != y).sum()), flush=True)
        dg = (a[i]["grad"] - b[i]["grad"]).abs()
        dp = (a[i]["flat"] - b[i]["flat"]).abs()
        nt = a[i]["grad"].numel() - 9500
        print(tag, "step", i, "loss %.7f %.7f n %d %d" % (a[i]["loss"], b[i]["loss"], a[i]["n"], b[i]["n"]),
              "| grad: max diff %.3e (max %.3e), entries differing %d (table %d), zero in one only %d" % (
                  float(dg.max()), float(b[i]["grad"].abs().max()), int((dg > 0).sum()), int((dg[:nt] > 0).sum()),
                  int(((a[i]["grad"] == 0) != (b[i]["grad"] == 0)).sum())),
              "| params after Adam: max diff %.3e, > 1e-4: %d" % (float(dp.max()), int((dp > 1e-4).sum())), flush=True)
cmp(ref2, ref, "plain/plain  ")
for k in range(reps):
    r = run(True)
    cmp(r, ref, "prefetch/plain")
