"""Is a step graph captured LATER (other graphs already in the pool, an lr change) slower than the first one?  (e2e: 2.70 -> 2.06 M rays/s
after the capture at an lr milestone.)  GPU only."""
import os, sys, time, subprocess
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py"]
import bench, numpy as np, math
from robust_e_nerf_amd import engine, ops

dev = "cuda:0"
B = int(os.environ.get("EVENTS", 6144))
ts, pos, quat, Kinv = bench.synthetic_scene()
T = torch.from_numpy
gen = torch.Generator().manual_seed(0)
fld = engine.NGPField(dev)
fld.flat.copy_((torch.rand(fld.flat.shape, generator=gen) * 2 - 1).to(dev) * 0.1)
cfg = engine.RenderCfg(sampler="occgrid")
r = engine.Renderer(fld, cfg)
r.binary.copy_(T(bench.ball_binary(128, float(os.environ.get("BALL", 1.2)), cfg.aabb)).to(dev))
tr = engine.Trainer(r, engine.TrainCfg(w_grad=1e-3), Kinv=T(Kinv), tab_ts=T(ts), tab_pos=T(pos), tab_quat=T(quat),
                    p2n_raw=torch.tensor(0.5413), neg_ct=torch.tensor(0.25), tau_raw=torch.tensor(0.0, dtype=torch.float64),
                    tau_max=torch.tensor(1e5), bkgd_raw=torch.tensor([0.5413]))
tr.use_graph = True
batches = [{k: T(v).to(dev).contiguous() for k, v in bench.synthetic_events(B, int(ts[-1]), seed=1 + b).items()} for b in range(4)]


def run(steps, tag):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    for i in range(steps):
        j = torch.rand(3, B, device=dev)
        loss, aux = tr.step(batches[i % 4], j[0], j[1], jitter_grad=j[2], global_step=None)
        n += aux["n"] + aux["grad"]["n"]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{tag:44s} {dt / steps * 1e3:7.3f} ms/step  {n / dt / 1e6:7.1f} M samples/s  replays {tr.graph_replays} captures {tr.graph_captures} "
          f"overflows {tr.device_count_overflows} mem {torch.cuda.memory_allocated() / 2**30:.1f}/{torch.cuda.memory_reserved() / 2**30:.1f} GiB", flush=True)


if os.environ.get("STREAM"):                       # all of it on a non-default stream (another hardware queue for the launches)
    _s = torch.cuda.Stream(priority=-1 if os.environ["STREAM"] == "high" else 0)
    torch.cuda.set_stream(_s)
if os.environ.get("SIDE_PRIO"):                    # the trainer's side stream at normal priority
    tr._side = torch.cuda.Stream()
run(6, "warm-up (host counts, capture)")
if os.environ.get("PHASE") == "A":
    run(100, "graph 1")
    sys.exit(0)
if os.environ.get("PHASE") == "C":                 # second capture at ANOTHER event count
    run(30, "graph 1")
    B = B + 64
    batches = [{k: T(v).to(dev).contiguous() for k, v in bench.synthetic_events(B, int(ts[-1]), seed=1 + b).items()} for b in range(4)]
    run(6, "other event count: capture 2")
    run(60, "graph 2")
    print("keys", [(k[0], k[1]) for k in tr._graphs], "overflows", tr.device_count_overflows)
    sys.exit(0)
if os.environ.get("PHASE") == "D":                 # same event count, graph 1 kept, capture 2 forced through a dummy key change
    run(30, "graph 1")
    print("keys", [(k[0], k[1]) for k in tr._graphs], "overflows", tr.device_count_overflows, "spr", tr.r._spr)
    tr.t.w_diff = 1.0000001
    tr.lr_scale = 0.5
    run(6, "lr change: capture 2")
    run(60, "graph 2")
    print("keys", [(k[0], k[1]) for k in tr._graphs], "overflows", tr.device_count_overflows, "spr", tr.r._spr)
    tr.lr_scale = 1.0
    run(60, "graph 1 again")
    sys.exit(0)
if os.environ.get("PHASE") == "E":                 # many captures in a row (an lr schedule with many milestones, batch sizes that come and go)
    run(30, "graph 1")
    for k in range(2, 10):
        tr.lr_scale = 1.0 / k
        run(4, f"capture {k}")
        run(30, f"graph {k}")
    tr.lr_scale = 1.0
    run(30, "graph 1 again")
    sys.exit(0)
if os.environ.get("PHASE") == "B":
    tr._graphs.clear()
    run(6, "cleared: capture 2")
    run(100, "graph 2 (the only one)")
    sys.exit(0)
run(50, "graph 1")
tr.use_graph = False
run(50, "eager")
tr.use_graph = True
run(50, "graph 1 again")
tr.lr_scale = 0.33
run(6, "lr change: capture 2")
run(50, "graph 2")
tr._graphs.clear()
run(6, "cleared: capture 3")
run(50, "graph 3 (the only one)")
tr.lr_scale = 0.1
run(6, "lr change: capture 4")
run(50, "graph 4")
tr.use_graph = False
run(50, "eager")
