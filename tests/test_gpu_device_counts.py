"""Device-side sample counts (RenderCfg.device_counts): the reference reads the number of marched and of
visible samples back to the host in the middle of every render (external/utils.py:106-119, models/nerf.py:279-286).  The
trainer's renders keep both on the device and enqueue the rest of the step over arrays of a learnt capacity; these tests
hold that path to the host-count path: the SAME sample counts, losses and gradients, step after step -- also when a count
does not fit and the step is repeated."""
import numpy as np
import pytest
import torch

from test_gpu_parity import DEV, _config_batch, _trainer_from_golden, amd, dev, load_golden      # noqa: F401  (amd: fixture)

pytestmark = pytest.mark.gpu


def _run(engine, g, table, device_counts, steps=5, B=2048, w_grad=0.0, trainable=False, squeeze_at=None, events=None):
    tr, _ = _trainer_from_golden(engine, g, table)
    tr.device_counts = device_counts
    if w_grad:
        tr.t.w_grad, tr.t.err_grad, tr.t.pw_grad = w_grad, "mape", None
    if trainable:
        tr.t.train_contrast_threshold = tr.t.train_refractory_period = True
    gen = torch.Generator().manual_seed(5)
    out = []
    for i in range(steps):
        b = B if events is None else events[i]
        nb = _config_batch(b, 30 + i, int(g["tab_ts"][-1]))
        nb["u_grad"] = torch.rand(b, generator=gen, dtype=torch.float64).numpy()
        batch = {k: dev(v) for k, v in nb.items()}
        j = [dev(torch.rand(b, generator=gen)) for _ in range(3)]
        if squeeze_at is not None and i == squeeze_at and tr.r._spr is not None:
            tr.r._spr = tuple(0.25 * s for s in tr.r._spr)          # arrays four times too small: both guards must trip
        loss, aux = tr.step(batch, j[0], j[1], jitter_grad=j[2] if w_grad else None)
        rec = dict(loss=float(loss), n=int(aux["n"]), n_marched=int(aux["n_marched"]),
                   n_grad=int(aux["grad"]["n"]) if w_grad else 0, table=tr.r.field.table.clone(), mlp=tr.r.field.mlp.clone(),
                   small=tr.small.clone(), ct=tr.ct.clone(), tau=float(tr.tau))
        out.append(rec)
    return out, tr


def _same(a, b, tol=2e-5, ptol=None):     # (parameters after Adam: the scatter's float-atomic noise, amplified where a gradient is ~0:
    ptol = tol if ptol is None else ptol   #  Adam moves such a parameter by lr x g / (|g| + eps) whatever |g| is -- ptol bounds the parameters)
    for i, (x, y) in enumerate(zip(a, b)):
        assert (x["n"], x["n_marched"], x["n_grad"]) == (y["n"], y["n_marched"], y["n_grad"]), (i, x["n"], y["n"])
        assert abs(x["loss"] - y["loss"]) <= tol * abs(y["loss"]), (i, x["loss"], y["loss"])
        for k in ("table", "mlp", "small", "ct"):
            d = float((x[k] - y[k]).abs().max())
            assert d <= ptol * max(float(y[k].abs().max()), 1e-30) + 1e-12, (i, k, d)
        # tau: Adam's m / sqrt(v) carries the gradients' 1e-6 float-atomic noise, times lr = 50 tau_max = 5e6 per step
        assert abs(x["tau"] - y["tau"]) <= 1e-3 * abs(y["tau"]) + 1e-300, (i, x["tau"], y["tau"])


@pytest.mark.parametrize("w_grad,trainable", [(0.0, False), (1e-3, False), (1e-3, True)], ids=["l_diff", "l_diff+l_grad", "C_p,tau"])
def test_device_counts_repeat_the_host_count_steps(amd, full_table_cache, w_grad, trainable):
    """five optimiser steps (2 048 events, occupancy sampler): exact sample counts of every
    render, losses and parameters after every step equal to the host-count run's"""
    ops, engine = amd
    g = load_golden("training_step_grad")
    table = full_table_cache(g["table_seed"], g["table_scale"])
    # trainable tau: its Adam group (lr = 50 tau_max, a division by sqrt(v)) amplifies the scatter's float-atomic noise
    # chaotically -- two HOST-count runs of the same five steps part by 1e-6 .. 1e-3 in tau from the third or fourth step on
    # (tools/dc_noise_probe.py, also with NaN-filled allocations: no uninitialised read) -- so that case is held over three steps
    steps = 3 if trainable else 5
    ref, tr0 = _run(engine, g, table, False, steps=steps, w_grad=w_grad, trainable=trainable)
    got, tr1 = _run(engine, g, table, None, steps=steps, w_grad=w_grad, trainable=trainable)
    assert tr1.device_counts_ok() and tr1.r._spr is not None
    assert getattr(tr1, "device_count_overflows", 0) == 0
    _same(got, ref, tol=2e-4 if trainable else 2e-5)


def test_device_counts_follow_a_changing_batch_size(amd, full_table_cache):
    """update_train_batch_size changes the ray count from step to step: the capacities are per ray"""
    ops, engine = amd
    g = load_golden("training_step_grad")
    table = full_table_cache(g["table_seed"], g["table_scale"])
    ev = [2048, 1536, 3000, 777, 4096, 2048]
    ref, _ = _run(engine, g, table, False, steps=6, events=ev, w_grad=1e-3)
    got, tr = _run(engine, g, table, None, steps=6, events=ev, w_grad=1e-3)
    assert getattr(tr, "device_count_overflows", 0) == 0
    _same(got, ref)


def test_overflowed_device_counts_repeat_the_step(amd, full_table_cache):
    """arrays made four times too small at step 2: the guards clear the renders, the optimiser does not see their
    gradients, the step is repeated with host counts -- same counts, losses and parameters as the host-count run, and the
    capacities recover"""
    ops, engine = amd
    g = load_golden("training_step_grad")
    table = full_table_cache(g["table_seed"], g["table_scale"])
    ref, _ = _run(engine, g, table, False, w_grad=1e-3)
    got, tr = _run(engine, g, table, None, w_grad=1e-3, squeeze_at=2)
    assert tr.device_count_overflows == 1
    _same(got, ref)


def test_count_guard_and_frag_zero_tail(amd):
    ops, _ = amd
    counts = torch.tensor([3, 0, 5, 2], dtype=torch.int32, device=DEV)
    also = torch.tensor([9, 9, 9, 9], dtype=torch.int32, device=DEV)
    offs, total = ops.exclusive_scan(counts)
    n_out, stats = torch.full((1,), -1, dtype=torch.int64, device=DEV), torch.full((2,), -1, dtype=torch.int64, device=DEV)
    ops.count_guard(counts, total, 10, n_out, stats, counts_also=also)
    assert n_out.item() == 10 and stats.tolist() == [10, 0] and counts.tolist() == [3, 0, 5, 2] and also.tolist() == [9] * 4
    ops.count_guard(counts, total, 9, n_out, stats, counts_also=also)
    assert n_out.item() == 0 and stats.tolist() == [10, 1] and counts.tolist() == [0] * 4 and also.tolist() == [0] * 4
    feat = torch.ones(3 * ops.FRAG_FLOATS_PER_BLOCK, device=DEV)
    n_dev = torch.tensor([37], dtype=torch.int64, device=DEV)
    ops.frag_zero_tail(feat, 96, n_dev)
    f = feat.view(3, 16, 2, 32)
    assert float(f[0].min()) == 1.0 and float(f[2].min()) == 1.0
    assert float(f[1, :, :, :5].min()) == 1.0 and float(f[1, :, :, 5:].abs().max()) == 0.0


def test_one_launch_scan_and_guard(amd):
    """ren_exclusive_scan / ren_scan_guard below 65 536 elements (one workgroup, one launch) against torch.cumsum, ragged
    sizes included; the guard clears both count arrays when the total does not fit"""
    ops, _ = amd
    gen = torch.Generator().manual_seed(1)
    for n in (1, 3, 4, 1023, 1024, 1025, 4097, 16384, 40000, 65536, 65537, 200000):
        c = torch.randint(0, 300, (n,), generator=gen, dtype=torch.int32)
        ref = torch.cumsum(c.long(), 0) - c.long()
        cd = c.to(DEV)
        offs, total = ops.exclusive_scan(cd)
        assert torch.equal(offs.cpu(), ref) and int(total) == int(c.sum()), n
        for cap, over in ((int(c.sum()), False), (int(c.sum()) - 1, True)):
            cd, also = c.to(DEV), torch.ones(n, dtype=torch.int32, device=DEV)
            n_out, stats = torch.empty(1, dtype=torch.int64, device=DEV), torch.empty(2, dtype=torch.int64, device=DEV)
            offs, total = ops.scan_guard(cd, cap, n_out, stats, counts_also=also)
            assert torch.equal(offs.cpu(), ref) and int(total) == int(c.sum()) and stats.tolist() == [int(c.sum()), int(over)], (n, cap)
            assert int(n_out) == (0 if over else int(c.sum()))
            assert int(cd.abs().sum()) == (0 if over else int(c.sum())) and int(also.sum()) == (0 if over else n)


def test_an_overflowed_pass_leaves_the_table_and_mlp_gradients_alone(amd, full_table_cache):
    """the exact repeat of an overflowed pass rests on every kernel of the pass honouring the cleared counts (n_dev = 0):
    run the loss passes ONCE with capacities four times too small and without the repeat (Trainer._forward_backward /
    _grad_loss_forward_backward directly, counts on the device) -- with and without the marcher's interval cache -- and look at the
    gradient buffers: the table and MLP gradients must still be exactly zero (only the 64-byte block of scalar-parameter gradients,
    which _dc_pass restores from its snapshot, may have moved), and no per-sample array was written past its capacity (the step
    after it, with the capacities back, equals the host-count run)."""
    ops, engine = amd
    g = load_golden("training_step_grad")
    table = full_table_cache(g["table_seed"], g["table_scale"])
    for march_cache in (512, 0):
        tr, _ = _trainer_from_golden(engine, g, table)
        tr.r.cfg.march_cache = march_cache
        tr.t.w_grad, tr.t.err_grad, tr.t.pw_grad = 1e-3, "mape", None
        tr.use_graph = False
        gen = torch.Generator().manual_seed(5)
        nb = _config_batch(2048, 30, int(g["tab_ts"][-1]))
        nb["u_grad"] = torch.rand(2048, generator=gen, dtype=torch.float64).numpy()
        batch = {k: dev(v) for k, v in nb.items()}
        j = [dev(torch.rand(2048, generator=gen)) for _ in range(3)]
        tr.step(batch, j[0], j[1], jitter_grad=j[2])                      # host counts: the capacities learn
        assert tr.r._spr is not None
        good = tr.r._spr
        tr.r._spr = tuple(0.25 * s for s in good)
        f = tr.r.field
        assert float(f.grad_all.abs().max()) == 0.0                      # (Adam cleared them)
        _, _, log = tr._forward_backward(batch, j[0], j[1], False, True)
        _, _, log_g = tr._grad_loss_forward_backward(batch, j[2], True, False, True)
        assert log.overflowed and log_g.overflowed
        torch.cuda.synchronize()
        assert float(f.g_table.abs().max()) == 0.0 and float(f.g_mlp.abs().max()) == 0.0, march_cache
        tr._gs.zero_()
        tr.r._spr = good
    ref, _ = _run(engine, g, table, False, steps=2, w_grad=1e-3)
    got, _ = _run(engine, g, table, None, steps=2, w_grad=1e-3)
    _same(got, ref)
