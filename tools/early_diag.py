"""Repeatability of three consecutive l_diff + l_grad steps, third render placed in order / early / begun (GPU only)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as T
from oracle import hashgrid
from robust_e_nerf_amd import ops, engine
g = T.load_golden("training_step_grad")
table = hashgrid.init_table(hashgrid.make_spec(), int(g["table_seed"]), float(g["table_scale"]), "mix32")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
B = 4096
dev = T.dev
def run(mode):
    tr, _ = T._trainer_from_golden(engine, g, table, sampler="occgrid")
    tr.t.w_grad, tr.t.err_grad, tr.t.pw_grad = float(g["w_grad"]), "mape", None
    gen = torch.Generator().manual_seed(5)
    steps = []
    for i in range(3):
        nb = T._config_batch(B, 30 + i, int(g["tab_ts"][-1]))
        nb["u_grad"] = torch.rand(B, generator=gen, dtype=torch.float64).numpy()
        steps.append(({k: dev(v) for k, v in nb.items()}, dev(torch.rand(B, generator=gen)), dev(torch.rand(B, generator=gen)),
                      dev(torch.rand(B, generator=gen))))
    torch.cuda.synchronize()
    res = []
    for i in range(3):
        b, j0, j1, j2 = steps[i]
        if mode == "begun":
            tr.begin_grad_sampling(b, j2)
        loss, aux = tr.forward_backward(b, j0, j1)
        lg, aux_g = tr.grad_loss_forward_backward(b, j2, early={"inorder": False, "begun": True, "early": "all"}[mode])
        tr.optimizer_step()
        res.append("%.7f %.7f n=%d ng=%d" % (float(loss), float(lg), aux["n"], aux_g["n"]))
    return " | ".join(res)
from collections import Counter
for mode in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("inorder", "early", "begun")):
    c = Counter(run(mode) for _ in range(reps))
    for k, v in c.items():
        print(mode, v, "x", k, flush=True)
