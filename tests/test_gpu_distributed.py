"""Two ranks through the HIP path on ONE GPU (gloo process group, both processes on cuda:0): what a 1-GPU lease can check of
the data-parallel step that replaces Lightning DDP (scripts/run.py:81-93; models/robust_e_nerf.py:63-66,916-919):

* the sum of the ranks' shard gradients equals the single-process gradient of the whole batch -- with gradient
  accumulation (2 ranks x 2 micro-batches == 1 rank x 4 micro-batches), i.e. the early slice of the table gradient is
  reduced exactly once per optimiser step;
* a rank WITHOUT a single sample issues the same collectives as its peer;
* after K steps with an occupancy refresh and a dynamic-batch-size change the replicated parameters, Adam state and
  occupancy grids are bit-identical on both ranks.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import REPO

pytestmark = pytest.mark.gpu
WORLD = 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_trainer(world, sampler="occgrid", w_grad=1e-3, dp_overlap=True, dev="cuda:0"):
    import bench
    from robust_e_nerf_amd import engine
    ts, pos, quat, Kinv = bench.synthetic_scene(201)
    T = torch.from_numpy
    gen = torch.Generator(device=dev).manual_seed(0)
    fld = engine.NGPField(dev)
    fld.flat.copy_((torch.rand(fld.flat.shape, device=dev, generator=gen) * 2 - 1) * 0.1)
    cfg = engine.RenderCfg(sampler=sampler, n_uniform=24, occ_thre=1e-5, dp_overlap=dp_overlap)
    r = engine.Renderer(fld, cfg)
    r.binary.copy_(T(bench.ball_binary(128, 0.5, cfg.aabb)).to(dev))
    tr = engine.Trainer(r, engine.TrainCfg(w_grad=w_grad, train_contrast_threshold=True), Kinv=T(Kinv), tab_ts=T(ts),
                        tab_pos=T(pos), tab_quat=T(quat), p2n_raw=torch.tensor(0.5413), neg_ct=torch.tensor(0.25),
                        tau_raw=torch.tensor(0.0, dtype=torch.float64), tau_max=torch.tensor(1e5),
                        bkgd_raw=torch.tensor([0.5413]), world_size=world)
    return tr, int(ts[-1])


def _events(B, t_end, seed, dev="cuda:0"):
    import bench
    ev = bench.synthetic_events(B, t_end, seed=seed)
    b = {k: torch.from_numpy(v).to(dev).contiguous() for k, v in ev.items()}
    g = torch.Generator().manual_seed(seed)
    return b, torch.rand(3, B, generator=g).to(dev)


def _shard(b, jit, lo, hi):
    return {k: v[lo:hi].contiguous() for k, v in b.items()}, jit[:, lo:hi].contiguous()


def _worker(rank, port, out_dir, backend="gloo", one_gpu_per_rank=False):
    """backend gloo + both ranks on cuda:0 (this file), or backend nccl (= RCCL) + one GPU per rank
    (tests/test_gpu_rccl_multirank.py, needs >= 2 GPUs): the same checks either way"""
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(WORLD),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from robust_e_nerf_amd import parallel
    dev = f"cuda:{rank}" if one_gpu_per_rank else "cuda:0"
    torch.cuda.set_device(dev)
    parallel.init_from_env(backend=backend)
    assert dist.get_world_size() == WORLD and dist.get_backend() == backend
    res = {"backend": dist.get_backend(), "world": dist.get_world_size(), "device": dev}
    import functools
    mk, evs = functools.partial(_make_trainer, dev=dev), functools.partial(_events, dev=dev)   # this rank's device
    # ---- 1. gradient accumulation under data parallelism: 2 ranks x 2 micro-batches, ONE optimiser step
    tr, t_end = mk(WORLD)
    B = 256                                                   # events per rank and micro-batch
    for bi in range(2):
        b, jit = evs(2 * B, t_end, seed=10 + bi)
        sb, sj = _shard(b, jit, rank * B, (rank + 1) * B)
        tr.step(sb, sj[0], sj[1], global_step=0, jitter_grad=sj[2], batch_index=bi, accumulate_grad_batches=2)
    res["collectives"] = tr.last_collectives
    res["m_accum"] = tr.m.clone()
    res["ct_m"] = tr.ct_m.clone()
    # ---- 2. a rank without samples (empty occupancy grid on rank 1 only) still matches its peer's collectives
    tr2, _ = mk(WORLD, w_grad=0.0)
    if rank == 1:
        tr2.r.binary.zero_()
    b, jit = evs(2 * B, t_end, seed=20)
    sb, sj = _shard(b, jit, rank * B, (rank + 1) * B)
    loss, aux = tr2.step(sb, sj[0], sj[1])
    res["empty_n"] = aux["n"]
    res["empty_collectives"] = tr2.last_collectives
    res["empty_m_nonzero"] = int((tr2.m != 0).sum())
    del tr2
    # ---- 3. K steps: occupancy refresh inside, dynamic batch size; replicas must stay identical
    tr3, _ = mk(WORLD)
    Bk, sizes = 256, []
    for k, gs in enumerate((15, 16, 17, 32, 33)):              # refreshes at 16 and 32
        b, jit = evs(2 * Bk, t_end, seed=30 + k)
        sb, sj = _shard(b, jit, rank * Bk, (rank + 1) * Bk)
        loss, aux = tr3.step(sb, sj[0], sj[1], global_step=gs, jitter_grad=sj[2])
        new = tr3.update_train_batch_size(aux, eff_ray_sample_batch_size=1 << 14)
        sizes.append(new)
        Bk = max(64, min(512, new))
    res["sizes"] = sizes
    torch.cuda.synchronize()
    mine = [tr3.r.field.flat, tr3.m, tr3.v, tr3.small, tr3.ct, tr3.r.binary.float(), tr3.r.occs]
    same = []
    for t in mine:
        tc = t.detach().clone() if backend == "nccl" else t.detach().cpu()
        both = [torch.empty_like(tc) for _ in range(WORLD)]
        dist.all_gather(both, tc)
        same.append(bool(torch.equal(both[0], both[1])))
    res["identical"] = same
    res["binary_cells"] = int(tr3.r.binary.sum())
    # ---- 4. one evaluation image rendered by both ranks (row bands + all-gather, C3) == the single-rank render
    from robust_e_nerf_amd import evaluation, ops
    ts_q = torch.tensor([7.3e6], dtype=torch.float64, device=dev)
    pos, rot = ops.trajectory(ts_q, tr3.tab_ts, tr3.tab_pos, tr3.tab_quat)
    bk = torch.nn.functional.softplus(tr3.small[:1])
    full = evaluation.render_image(tr3.r, tr3.Kinv, pos[0], rot[0], 50, 64, bk)
    shard = evaluation.render_image_sharded(tr3.r, tr3.Kinv, pos[0], rot[0], 50, 64, bk, rank=rank, world=WORLD)
    res["render_equal"] = [bool(torch.equal(a, b)) for a, b in zip(full, shard)]
    # ---- 5. device-side sample counts under data parallelism: four steps with the counts on the device,
    # capacities made four times too small on RANK 1 ONLY at step 2 -- that rank repeats its passes by itself (no collective
    # inside a pass: one all-reduce per step at the settle point, on every rank) -- against the host-count run
    def dp_run(dc, squeeze, graph=False):
        tr5, _ = mk(WORLD)
        tr5.device_counts, tr5.use_graph = dc, graph
        out = []
        for k in range(4):
            b5, jit5 = evs(2 * B, t_end, seed=50 + k)
            sb5, sj5 = _shard(b5, jit5, rank * B, (rank + 1) * B)
            if squeeze and k == 2 and rank == 1 and tr5.r._spr is not None:
                tr5.r._spr = tuple(0.25 * s for s in tr5.r._spr)
                tr5._graphs.clear()                           # (at 256 events a cached graph's capacities would still be taken)
            loss5, aux5 = tr5.step(sb5, sj5[0], sj5[1], jitter_grad=sj5[2])
            out.append((float(loss5), int(aux5["n"]), int(aux5["grad"]["n"]), tr5.last_collectives))
        return out, tr5
    ref5, trh = dp_run(False, False)
    got5, trd = dp_run(None, True)
    overs = [None] * WORLD
    dist.all_gather_object(overs, int(trd.device_count_overflows))
    res["dc_overflows"] = overs
    res["dc_counts_equal"] = [a[1:3] == b[1:3] for a, b in zip(got5, ref5)]
    res["dc_loss_err"] = max(abs(a[0] - b[0]) / abs(b[0]) for a, b in zip(got5, ref5))
    res["dc_collectives"] = ([a[3] for a in ref5], [a[3] for a in got5])
    res["dc_param_err"] = float((trd.r.field.flat - trh.r.field.flat).abs().max() / trh.r.field.flat.abs().max())
    res["dc_on"] = bool(trd.device_counts_ok() and not trh.device_counts_ok())
    # the same with the loss passes of every step replayed from a hipGraph (round 6): under data parallelism the captured part ends
    # where the gradient exchange begins -- the all-reduce and the optimiser launches follow the replay eagerly -- and the rank whose
    # count overflows repeats its passes before it enters the collective its peer is waiting in
    got6, trg = dp_run(None, True, graph=True)
    overs6, reps6 = [None] * WORLD, [None] * WORLD
    dist.all_gather_object(overs6, int(trg.device_count_overflows))
    dist.all_gather_object(reps6, int(trg.graph_replays))
    res["dpg_overflows"], res["dpg_replays"] = overs6, reps6
    res["dpg_counts_equal"] = [a[1:3] == b[1:3] for a, b in zip(got6, ref5)]
    res["dpg_loss_err"] = max(abs(a[0] - b[0]) / abs(b[0]) for a, b in zip(got6, ref5))
    res["dpg_collectives"] = [a[3] for a in got6]
    res["dpg_param_err"] = float((trg.r.field.flat - trh.r.field.flat).abs().max() / trh.r.field.flat.abs().max())
    views = evaluation.view_shard(3, rank, WORLD)
    local = torch.stack([evaluation.render_image(tr3.r, tr3.Kinv, pos[0] + 0.01 * v, rot[0], 20, 24, bk)[0] for v in views])
    allv = evaluation.gather_views(local, 3, rank, WORLD)
    ref = torch.stack([evaluation.render_image(tr3.r, tr3.Kinv, pos[0] + 0.01 * v, rot[0], 20, 24, bk)[0] for v in range(3)])
    res["views_equal"] = bool(torch.equal(allv, ref))
    if rank == 0:
        torch.save({k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in res.items()}, os.path.join(out_dir, "r0.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_through_the_hip_path_on_one_gpu(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    check_two_rank_results(torch.load(os.path.join(tmp_path, "r0.pt")))


def check_two_rank_results(got):
    # 1. one rank, the same four micro-batches, accumulate 4: the mean gradient (Adam's first moment after step 1) must agree
    tr, t_end = _make_trainer(1)
    B, bi = 256, 0
    for mb in range(2):
        b, jit = _events(2 * B, t_end, seed=10 + mb)
        for r in range(WORLD):
            sb, sj = _shard(b, jit, r * B, (r + 1) * B)
            tr.step(sb, sj[0], sj[1], global_step=0, jitter_grad=sj[2], batch_index=bi, accumulate_grad_batches=4)
            bi += 1
    m_ref, m_dp = tr.m.cpu(), got["m_accum"]
    n_table = tr.r.field.n_table
    for name, sl in (("table", slice(0, n_table)), ("mlp", slice(n_table, tr.r.field.n_params))):
        err = float((m_dp[sl] - m_ref[sl]).abs().max() / m_ref[sl].abs().max())
        print(f"2 ranks x 2 micro-batches vs 1 rank x 4: first moment of the {name} gradient, max rel err {err:.2e}")
        assert err < 2e-5, (name, err)
    assert float((got["ct_m"] - tr.ct_m.cpu()).abs().max()) <= 1e-5 * float(tr.ct_m.abs().max())
    # with the sample counts on the device (the default for the occupancy sampler) no pass contains a collective: ONE
    # all-reduce of the packed buffer per optimiser step, at the settle point (engine.Trainer.device_counts_ok)
    assert got["collectives"] == 1, got["collectives"]
    # 2. the empty rank issued the same collectives and received its peer's gradient
    assert got["empty_n"] > 0 and got["empty_collectives"] == 1 and got["empty_m_nonzero"] > 0
    # 3. replicas identical after refreshes and batch-size changes
    assert all(got["identical"]), got["identical"]
    assert got["binary_cells"] > 0 and len(set(got["sizes"])) > 1, (got["binary_cells"], got["sizes"])
    assert all(got["render_equal"]) and got["views_equal"]
    # 5. device-side counts under data parallelism, one rank overflowing: the same counts, losses and parameters as the
    # host-count run (whose steps keep the early slice: 3 collectives), one collective per step, rank 1 repeated by itself
    assert got["dc_on"] and got["dc_overflows"] == [0, 1], got["dc_overflows"]
    assert all(got["dc_counts_equal"]) and got["dc_loss_err"] < 2e-5 and got["dc_param_err"] < 2e-5, \
        (got["dc_counts_equal"], got["dc_loss_err"], got["dc_param_err"])
    assert got["dc_collectives"] == ([3] * 4, [1] * 4), got["dc_collectives"]
    # ... and with the passes replayed from a captured graph (steps 1-3; step 0 learns the capacities with host counts)
    assert got["dpg_overflows"] == [0, 1] and min(got["dpg_replays"]) >= 2, (got["dpg_overflows"], got["dpg_replays"])
    assert all(got["dpg_counts_equal"]) and got["dpg_loss_err"] < 2e-5 and got["dpg_param_err"] < 1e-3, \
        (got["dpg_counts_equal"], got["dpg_loss_err"], got["dpg_param_err"])
    assert got["dpg_collectives"] == [1] * 4, got["dpg_collectives"]
