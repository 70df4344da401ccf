"""hipGraph feasibility probe for the small-batch step: does torch.cuda.graph capture the library's ctypes
launches (they go to torch's current raw stream), what does a replay cost per node against eager launches from Python, do
forked side streams and pinned-memory copy nodes work, and can the host poll pinned memory a running graph writes.

    python tools/graph_probe.py            (on the GPU box)
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robust_e_nerf_amd import ops  # noqa: E402

dev = "cuda:0"
torch.cuda.set_device(0)
N_NODES = int(os.environ.get("NODES", 80))
n = 4096
bufs = [torch.empty(n, device=dev) for _ in range(4)]


def chain(k):
    for i in range(k):
        ops.uniform(n, 7, i, device=dev, out=bufs[i % 4])


def timeit(fn, reps=50):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6, t_host / reps * 1e6


# 1. eager
eager, eager_host = timeit(lambda: chain(N_NODES))
print(f"eager : {N_NODES} launches  {eager:8.1f} us per pass ({eager / N_NODES:.2f} us / launch), host enqueue {eager_host:.1f} us")

# 2. captured
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    chain(N_NODES)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    chain(N_NODES)
graph, graph_host = timeit(g.replay)
print(f"graph : {N_NODES} nodes     {graph:8.1f} us per replay ({graph / N_NODES:.2f} us / node), host launch {graph_host:.1f} us")
ref = bufs[(N_NODES - 1) % 4].clone()
g.replay()
torch.cuda.synchronize()
assert torch.equal(ref, bufs[(N_NODES - 1) % 4])

# 3. fork / join on a side stream + allocation inside the capture + pinned copy node
side = torch.cuda.Stream()
pinned = torch.zeros(4, dtype=torch.int64).pin_memory()
seq_dev = torch.zeros(4, dtype=torch.int64, device=dev)
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    cur = torch.cuda.current_stream()
    a = ops.uniform(n, 1, 0, device=dev)
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        b = ops.uniform(n, 2, 0, device=dev)
        seq_dev += 1
        pinned.copy_(seq_dev, non_blocking=True)
    chain(40)
    cur.wait_stream(side)
    c = a + b
for _ in range(3):
    g2.replay()
torch.cuda.synchronize()
print("fork/join + alloc + pinned copy node: seq on host =", pinned.tolist(), " c ok:", bool(torch.isfinite(c).all()))

# 4. host polls pinned memory while the graph runs
t_seen = []
for rep in range(5):
    want = int(pinned[0]) + 1
    t0 = time.perf_counter()
    g2.replay()
    while int(pinned[0]) < want:
        pass
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    t_seen.append(((t1 - t0) * 1e6, (t2 - t0) * 1e6))
print("poll: (us until the host saw the side stream's copy, us until the whole graph was done):", [(round(a), round(b)) for a, b in t_seen])

# 5. back-to-back replays: is there idle between graphs?
reps = 20
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    g.replay()
e1.record()
torch.cuda.synchronize()
print(f"{reps} replays back to back: {e0.elapsed_time(e1) / reps * 1e3:.1f} us each (GPU time)")
