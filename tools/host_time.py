import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.argv = ["bench.py", "--no-cpu-baseline", "--steps", "10", "--warmup", "3"] + sys.argv[1:]
from robust_e_nerf_amd import engine, parallel
import torch
acc = {}
def wrap(cls, name):
    f = getattr(cls, name)
    def g(self, *a, **k):
        t = time.perf_counter(); r = f(self, *a, **k); acc.setdefault(name, []).append(time.perf_counter() - t); return r
    setattr(cls, name, g)
wrap(engine.Trainer, "optimizer_step"); wrap(engine.Trainer, "forward_backward"); wrap(parallel.GradSync, "finish"); wrap(parallel.GradSync, "early")
wrap(engine.Renderer, "backward"); wrap(engine.Renderer, "forward"); wrap(engine.Trainer, "grad_loss_forward_backward"); wrap(engine.Renderer, "sample")
import bench
bench.main()
for k, v in acc.items():
    v = v[3:]
    print(k, "host ms avg %.3f max %.3f" % (1e3 * sum(v) / len(v), 1e3 * max(v)), file=sys.stderr)
