import sys; sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import torch, os
if os.environ.get('NANFILL'):
    torch.use_deterministic_algorithms(True, warn_only=True); torch.utils.deterministic.fill_uninitialized_memory = True
from test_gpu_device_counts import _run
from conftest import load_golden
from oracle import hashgrid
from robust_e_nerf_amd import engine
g = load_golden("training_step_grad")
table = hashgrid.init_table(hashgrid.make_spec(), int(g["table_seed"]), float(g["table_scale"]), "mix32")
runs = [(_run(engine, g, table, dc, w_grad=1e-3, trainable=True)[0]) for dc in (False, False, None, None)]
def diff(a, b):
    return [(abs(x["tau"] - y["tau"]) / abs(y["tau"]), float((x["table"] - y["table"]).abs().max()), abs(x["loss"]-y["loss"])/abs(y["loss"]), x["n"] == y["n"]) for x, y in zip(a, b)]
for name, (i, j) in {"host vs host": (0, 1), "dev vs dev": (2, 3), "host vs dev": (0, 2)}.items():
    print(name, [tuple(f"{v:.1e}" if not isinstance(v, bool) else v for v in t) for t in diff(runs[i], runs[j])])
