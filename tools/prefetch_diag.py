"""Repeatability of three consecutive training steps with and without Trainer.prefetch (diagnosis of a flaky test). GPU only.
   python tools/prefetch_diag.py [uniform|occgrid] [repeats]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as T
from oracle import hashgrid
from robust_e_nerf_amd import ops, engine
g = T.load_golden("training_step_diff")
table = hashgrid.init_table(hashgrid.make_spec(), int(g["table_seed"]), float(g["table_scale"]), "mix32")
sampler = sys.argv[1] if len(sys.argv) > 1 else "uniform"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
B = 4096
VAR = os.environ.get("DIAG_VAR", "")
if VAR == "prio0":
    engine.Trainer.side_stream = property(lambda self: self.__dict__.setdefault("_side0", torch.cuda.Stream()))
dev = T.dev
def run(use, sync_each=False):
    tr, _ = T._trainer_from_golden(engine, g, table, sampler=sampler)
    tr.r.cfg.n_uniform = 32
    if VAR == "unordered":          # Trainer.prefetch as built in round 3: its side-stream work NOT ordered after the backward
        tr.side_stream.wait_stream = lambda *a, **k: None
    gen = torch.Generator().manual_seed(5)
    steps = []
    for i in range(4):
        nb = T._config_batch(B, 30 + i, int(g["tab_ts"][-1]))
        steps.append(({k: dev(v) for k, v in nb.items()}, dev(torch.rand(B, generator=gen)), dev(torch.rand(B, generator=gen))))
    torch.cuda.synchronize()
    if VAR == "m2":       # inputs rewritten by a kernel (not the copy engine)
        steps = [({k: v.clone() for k, v in b.items()}, j0.clone(), j1.clone()) for b, j0, j1 in steps]
        torch.cuda.synchronize()
    res = []
    for i in range(3):
        ready = torch.cuda.Event(); ready.record()
        loss, aux = tr.forward_backward(*steps[i])
        if use:
            if VAR == "sync_before":
                torch.cuda.synchronize()
            if VAR == "m1":
                tr.side_stream.wait_event(ready)
            assert tr.prefetch(*steps[i + 1])
            if VAR == "sync_after":
                torch.cuda.synchronize()
        res.append((float(loss), aux["n"], float(tr.r.field.grad.double().abs().sum()), float(tr.small_grad[0])))
        if sync_each:
            torch.cuda.synchronize()
        tr.optimizer_step()
    return res
for k in range(reps):
    for use in (True,):
        r = run(use)
        print(VAR, sampler, "prefetch" if use else "plain   ", " | ".join("%.7f n=%d |g|=%.6e gb=%.4e" % x for x in r), flush=True)
