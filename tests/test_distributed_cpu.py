"""CPU, world_size 2 over gloo: the data-parallel path (shard -> local gradient -> flat all-reduce
(SUM) -> Adam with grad_scale 1/world) reproduces the single-process full-batch update.

The gradients here come from the oracle (this is a test); what is under test is
robust_e_nerf_amd.parallel and the gradient-scaling convention the HIP Adam kernel implements."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO

WORLD = 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _small_problem():
    """Tiny field (hashmap 2^12) + 64 rays so the oracle backward takes a second."""
    from oracle import field, hashgrid
    spec = hashgrid.make_spec(log2_hashmap_size=12)
    p = field.init_params(spec, seed=3, table_kind="normal", table_scale=0.3)
    g = torch.Generator().manual_seed(5)
    R = 64
    ang = torch.rand(R, generator=g) * 6.28
    o = torch.stack([4 * torch.cos(ang), 4 * torch.sin(ang), torch.rand(R, generator=g) - 0.5], -1)
    d = (torch.rand(R, 3, generator=g) - 0.5) - o
    d = d / d.norm(dim=-1, keepdim=True)
    target = torch.rand(R, generator=g)
    return spec, p, o.float(), d.float(), target, torch.rand(R, generator=g)


def _flat_grad(spec, p, o, d, target, jitter, lo, hi):
    """mean-squared rendering loss on rays [lo, hi) -> flat gradient in engine order [hash | MLP]."""
    from oracle import step as ostep
    from conftest import FIELD_KEYS
    q = {k: v.clone().requires_grad_() for k, v in p.items()}
    cfg = ostep.SceneCfg(sampler="uniform", n_uniform=16)
    colors, _, _, _, _ = ostep.render_rays(o[lo:hi], d[lo:hi], q, spec, cfg, binary=None, jitter=jitter[lo:hi],
                                           bkgd=torch.tensor([1.0]))
    loss = ((colors[:, 0] - target[lo:hi]) ** 2).mean()
    loss.backward()
    return torch.cat([q["hash"].grad.reshape(-1)] + [q[k].grad.reshape(-1) for k in FIELD_KEYS]), float(loss)


def _adam_ref(param, grad, steps=1):
    p = param.clone().requires_grad_()
    opt = torch.optim.Adam([p], lr=0.01, weight_decay=1e-6)
    for _ in range(steps):
        p.grad = grad.clone()
        opt.step()
    return p.detach()


def _worker(rank, port, out_dir):
    import sys
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(WORLD))
    torch.set_num_threads(2)
    from robust_e_nerf_amd import parallel
    r, w = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, WORLD)
    spec, p, o, d, target, jitter = _small_problem()
    lo, hi = parallel.shard_bounds(o.shape[0], rank, WORLD)
    grad, loss = _flat_grad(spec, p, o, d, target, jitter, lo, hi)
    small = torch.tensor([float(rank + 1), 0.0, 0.0, 0.0])
    parallel.allreduce_sum_([grad, small])
    assert small[0] == 3.0
    mean_s = parallel.allgather_mean(10.0 * (rank + 1))
    assert abs(mean_s - 15.0) < 1e-12
    assert parallel.rank_seed(7, rank) == 7 + rank and parallel.per_rank_budget(1 << 20, WORLD) == 1 << 19
    # ---- the packed buffer [table | MLP | pad | aux]: one collective; with an early slice three, same sums
    n_tab, n_pad = 1000, 1040
    buf = torch.arange(n_pad + parallel.AUX_FLOATS, dtype=torch.float32) * (rank + 1)
    buf[n_pad + parallel.AUX_TAU_HI] = 1e8 * (rank + 1)
    buf[n_pad + parallel.AUX_TAU_LO] = 0.25
    want = torch.arange(n_pad + parallel.AUX_FLOATS, dtype=torch.float32) * 3
    want[n_pad + parallel.AUX_TAU_HI], want[n_pad + parallel.AUX_TAU_LO] = 3e8, 0.5
    sync = parallel.GradSync(None, WORLD)
    b1 = buf.clone()
    sync.finish(b1)
    assert sync.reset_count() == 1 and torch.equal(b1, want)
    b2 = buf.clone()
    sync.early(b2, 400, n_tab)                              # the fine levels' slice, launched from the backward pass
    sync.finish(b2)
    assert sync.reset_count() == 3 and torch.equal(b2, want)
    # d loss / d tau travels as (hi, lo) float pair: the sum keeps what a float32 alone would lose
    tau = float(b2[n_pad + parallel.AUX_TAU_HI].double() + b2[n_pad + parallel.AUX_TAU_LO].double())
    assert tau == 3e8 + 0.5
    # bf16-compressed parameter gradients: the sum agrees to bf16 precision, the aux tail (tau pair, counters) stays exact
    csync = parallel.GradSync(None, WORLD, compress="bf16")
    b3 = buf.clone()
    csync.early(b3, 400, n_tab)
    csync.finish(b3)
    assert csync.reset_count() == 4                          # early slice, the two remaining ranges, the fp32 aux block
    assert torch.equal(b3[n_pad:], want[n_pad:])
    assert float(((b3[:n_pad] - want[:n_pad]).abs() / want[:n_pad].clamp_min(1.0)).max()) < 2 ** -7
    # ---- evaluation: views sharded like DistributedSampler(shuffle=False), outputs all-gathered (C3, robust_e_nerf.py:591)
    from robust_e_nerf_amd import evaluation
    for n_views in (1, 4, 5):
        mine = evaluation.view_shard(n_views, rank, WORLD)
        local = torch.stack([torch.full((2, 3), float(v)) for v in mine])
        allv = evaluation.gather_views(local, n_views, rank, WORLD)
        assert allv.shape == (n_views, 2, 3) and torch.equal(allv[:, 0, 0], torch.arange(n_views, dtype=torch.float32))
    if rank == 0:
        torch.save({"grad_sum": grad, "loss0": loss}, os.path.join(out_dir, "r0.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_matches_full_batch(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    got = torch.load(os.path.join(tmp_path, "r0.pt"))
    spec, p, o, d, target, jitter = _small_problem()
    full, _ = _flat_grad(spec, p, o, d, target, jitter, 0, o.shape[0])
    # DDP semantics: mean over ranks of per-rank mean-loss gradients == full-batch gradient (equal shards)
    avg = got["grad_sum"] / WORLD
    err = float((avg - full).abs().max() / full.abs().max())
    assert err < 1e-5, err
    # and the optimiser convention: Adam(grad_sum, grad_scale=1/world) == Adam(full-batch grad)
    from conftest import FIELD_KEYS
    flat = torch.cat([p["hash"].reshape(-1)] + [p[k].reshape(-1) for k in FIELD_KEYS])
    a = _adam_ref(flat, got["grad_sum"] * (1.0 / WORLD))
    b = _adam_ref(flat, full)
    # Adam normalises each coordinate by sqrt(v): compare only coordinates with a non-negligible gradient
    m = full.abs() > 1e-3 * full.abs().max()
    assert float((a - b)[m].abs().max()) < 1e-4 * 0.01 * 10


def test_view_shard_is_distributed_sampler_without_shuffle():
    from torch.utils.data.distributed import DistributedSampler
    from robust_e_nerf_amd import evaluation
    for n in (1, 2, 5, 8, 9):
        for w in (1, 2, 3, 8):
            for r in range(w):
                ref = list(DistributedSampler(list(range(n)), num_replicas=w, rank=r, shuffle=False))
                assert evaluation.view_shard(n, r, w) == ref, (n, w, r)


def test_shard_bounds_cover_everything():
    from robust_e_nerf_amd import parallel
    for n in (0, 1, 7, 64, 65):
        for w in (1, 2, 3, 8):
            pieces = [parallel.shard_bounds(n, r, w) for r in range(w)]
            assert pieces[0][0] == 0 and pieces[-1][1] == n
            assert all(pieces[i][1] == pieces[i + 1][0] for i in range(w - 1))
