"""The training step as ONE hipGraph launch (Trainer._graph_step).  A step whose shape repeats -- event
count, capacities of its renders' device-side sample counts, learning-rate factor -- is captured once and replayed: the same
launches in the same order, so the same sample counts, losses and parameters as the eager step, step after step; the
optimiser's step numbers live on the device (ren_step_tick / ren_adam_step_dev, ABI 25), and a count that does not fit its
arrays raises the skip word there: the captured optimiser launches then change nothing and the host repeats the step with
host-side counts (the reference's own placement of the reads: external/utils.py:106-119, models/nerf.py:279-286)."""
import numpy as np
import pytest
import torch

from test_gpu_parity import DEV, _config_batch, _trainer_from_golden, amd, dev, load_golden      # noqa: F401  (amd: fixture)
from test_gpu_device_counts import _same
# (ptol = 1e-3 below: after a few Adam steps a parameter whose gradient is ~0 has moved by lr x g / (|g| + eps) = +-0.01 per step
# whatever |g| is, so another summation order of the MLP weight gradients -- a replayed step, or `side_cus` CUs left free: another
# slab partition -- shows as 1e-5 .. 3e-4 of the largest parameter; a pass whose gradient were wrong would show as >= 1e-2.  Sample
# counts are compared exactly and every step's loss to 2e-5.)

pytestmark = pytest.mark.gpu


def _run(engine, g, table, use_graph, steps=6, B=2048, w_grad=0.0, trainable=False, squeeze_at=None):
    tr, _ = _trainer_from_golden(engine, g, table)
    tr.use_graph = use_graph
    if w_grad:
        tr.t.w_grad, tr.t.err_grad, tr.t.pw_grad = w_grad, "mape", None
    if trainable:
        tr.t.train_contrast_threshold = tr.t.train_refractory_period = True
    gen = torch.Generator().manual_seed(5)
    out = []
    for i in range(steps):
        nb = _config_batch(B, 30 + i, int(g["tab_ts"][-1]))
        nb["u_grad"] = torch.rand(B, generator=gen, dtype=torch.float64).numpy()
        batch = {k: dev(v) for k, v in nb.items()}
        j = [dev(torch.rand(B, generator=gen)) for _ in range(3)]
        if squeeze_at is not None and i == squeeze_at and tr.r._spr is not None:
            tr.r._spr = tuple(0.25 * s for s in tr.r._spr)          # arrays four times too small: both guards must trip
        loss, aux = tr.step(batch, j[0], j[1], jitter_grad=j[2] if w_grad else None)
        rec = dict(loss=float(loss), n=int(aux["n"]), n_marched=int(aux["n_marched"]),
                   n_grad=int(aux["grad"]["n"]) if w_grad else 0, table=tr.r.field.table.clone(), mlp=tr.r.field.mlp.clone(),
                   small=tr.small.clone(), ct=tr.ct.clone(), tau=float(tr.tau))
        out.append(rec)
    return out, tr


@pytest.mark.parametrize("w_grad,trainable", [(0.0, False), (1e-3, False), (1e-3, True)], ids=["l_diff", "l_diff+l_grad", "C_p,tau"])
def test_replayed_steps_repeat_the_eager_steps(amd, full_table_cache, w_grad, trainable):
    ops, engine = amd
    g = load_golden("training_step_grad")
    table = full_table_cache(g["table_seed"], g["table_scale"])
    steps = 3 if trainable else 6                  # (trainable tau: see test_device_counts_repeat_the_host_count_steps)
    ref, tr0 = _run(engine, g, table, False, steps=steps, w_grad=w_grad, trainable=trainable)
    got, tr1 = _run(engine, g, table, True, steps=steps, w_grad=w_grad, trainable=trainable)
    assert tr0.graph_replays == 0 and tr1.graph_replays >= steps - 2, (tr1.graph_replays, tr1.graph_captures)
    assert tr1.device_count_overflows == 0 and tr1.step_count == tr0.step_count == steps
    hy = tr1._hyper.tolist()
    assert hy[ops.HY_STEP] == steps and hy[ops.HY_SKIP] == 0.0
    _same(got, ref, tol=2e-4 if trainable else 2e-5, ptol=1e-3)


def test_auto_mode_captures_a_shape_that_repeats(amd, full_table_cache):
    """use_graph = None (the default): a step shape that occurs three times in a row is captured, the capture is measured against
    the eager steps timed just before it (three replays with the skip word raised: parameters untouched, gradients cleared) and
    kept only if it is faster (profiles/NOTES.md: which hardware queue a forked graph's internal stream lands on decides that)
    -- kept or rejected, the steps are the eager steps"""
    ops, engine = amd
    g = load_golden("training_step_grad")
    table = full_table_cache(g["table_seed"], g["table_scale"])
    got, tr = _run(engine, g, table, None, steps=8, w_grad=1e-3)
    assert tr.graph_captures >= 1 and tr.step_count == 8
    kept = [sg for sg in tr._graphs.values() if "ms" in sg]
    assert (len(kept) >= 1 and tr.graph_replays >= 1) or sum(tr._graph_bad.values()) >= 1
    for sg in kept:
        assert sg["ms"][0] <= 0.97 * sg["ms"][1]
    hy = tr._hyper.tolist()
    assert hy[ops.HY_STEP] == 8 and hy[ops.HY_SKIP] == 0.0
    ref, _ = _run(engine, g, table, False, steps=8, w_grad=1e-3)
    _same(got, ref, ptol=1e-3)


def test_overflow_inside_a_replayed_step_is_repeated_exactly(amd, full_table_cache):
    """capacities made four times too small at step 3: the new shape is captured, its replay overflows, the captured optimiser
    launches see the skip word and change nothing, the host clears the gradients and repeats the step with host counts"""
    ops, engine = amd
    g = load_golden("training_step_grad")
    table = full_table_cache(g["table_seed"], g["table_scale"])
    ref, _ = _run(engine, g, table, False, w_grad=1e-3)
    got, tr = _run(engine, g, table, True, w_grad=1e-3, squeeze_at=3)
    assert tr.device_count_overflows == 1 and tr.step_count == 6
    hy = tr._hyper.tolist()
    assert hy[ops.HY_STEP] == 6 and hy[ops.HY_SKIP] == 0.0
    _same(got, ref, ptol=1e-3)


def test_device_side_optimiser_state_matches_the_host_side_calls(amd):
    """ren_step_tick + ren_adam_step_dev / ren_tau_adam_step_dev against ren_adam_step / ren_tau_adam_step (torch.optim.Adam's
    arithmetic, test_adam_matches_torch) over five steps; a raised skip word freezes parameters, moments AND gradients"""
    ops, _ = amd
    gen = torch.Generator().manual_seed(0)
    n = 4099
    p0 = torch.randn(n + 1, generator=gen)[:n].to(DEV)
    pa, pb = p0.clone(), p0.clone()
    ma, va, mb, vb = (torch.zeros(n, device=DEV) for _ in range(4))
    hy = torch.zeros(8, dtype=torch.float64, device=DEV)
    ta, tb = (torch.tensor([0.3], dtype=torch.float64, device=DEV) for _ in range(2))
    sa, sb = (torch.zeros(2, dtype=torch.float64, device=DEV) for _ in range(2))
    for step in range(1, 6):
        gr = torch.randn(n, generator=gen).to(DEV)
        ga, gb = gr.clone(), gr.clone()
        tg_a, tg_b = (torch.tensor([0.7 * step], dtype=torch.float64, device=DEV) for _ in range(2))
        ops.adam_step(pa, ga, ma, va, lr=1e-2, weight_decay=1e-6, step=step, grad_scale=0.5)
        ops.tau_adam_step(ta, tg_a, sa, 1e5, lr=5.0, step=step, grad_scale=0.5)
        ops.step_tick(hy, tick_tau=True)
        ops.adam_step_dev(pb, gb, mb, vb, hy, lr=1e-2, weight_decay=1e-6, grad_scale=0.5)
        ops.tau_adam_step_dev(tb, tg_b, sb, 1e5, hy, lr=5.0, grad_scale=0.5)
        assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb) and float(gb.abs().max()) == 0.0
        assert torch.equal(ta, tb) and torch.equal(sa, sb) and float(tg_b) == 0.0
    assert hy[ops.HY_STEP].item() == 5 and hy[ops.HY_TAU_STEP].item() == 5
    over = torch.tensor([10, 1, 0, 0], dtype=torch.int64, device=DEV)
    gr = torch.randn(n, generator=gen).to(DEV)
    keep = (pb.clone(), mb.clone(), vb.clone(), gr.clone())
    ops.step_tick(hy, stats=[over])
    ops.adam_step_dev(pb, gr, mb, vb, hy, lr=1e-2)
    assert hy[ops.HY_SKIP].item() == 1.0 and hy[ops.HY_STEP].item() == 5
    assert all(torch.equal(a, b) for a, b in zip(keep, (pb, mb, vb, gr)))
    ops.step_tick(hy)                                      # sticky until the host clears it
    assert hy[ops.HY_SKIP].item() == 1.0 and hy[ops.HY_STEP].item() == 5


@pytest.mark.parametrize("trainable", [False, True], ids=["frozen", "C_p,tau"])
def test_merged_march_of_the_three_renders_repeats_the_separate_passes(amd, full_table_cache, trainable):
    """Trainer.grad_sampling = "merged" (the placement of every captured step): the third render's rays go through ONE ray / box
    test and ONE march count pass together with the l_diff renders' (Renderer.sample_begin_merged), everything else in order on
    one stream.  A ray's march does not depend on its neighbours: the same sample counts, losses and parameters as the
    "begun" placement, with host-side and with device-side counts."""
    ops, engine = amd
    g = load_golden("training_step_grad")
    table = full_table_cache(g["table_seed"], g["table_scale"])
    steps = 3 if trainable else 5

    def run(merged, dc):
        tr, _ = _trainer_from_golden(engine, g, table)
        tr.use_graph, tr.device_counts = False, dc
        tr.grad_sampling = "merged" if merged else None
        tr.t.w_grad, tr.t.err_grad, tr.t.pw_grad = 1e-3, "mape", None
        if trainable:
            tr.t.train_contrast_threshold = tr.t.train_refractory_period = True
        gen = torch.Generator().manual_seed(5)
        out = []
        for i in range(steps):
            nb = _config_batch(1536, 30 + i, int(g["tab_ts"][-1]))
            nb["u_grad"] = torch.rand(1536, generator=gen, dtype=torch.float64).numpy()
            batch = {k: dev(v) for k, v in nb.items()}
            j = [dev(torch.rand(1536, generator=gen)) for _ in range(3)]
            assert tr.grad_sampling_mode() == ("merged" if merged else "begun")
            loss, aux = tr.step(batch, j[0], j[1], jitter_grad=j[2])
            out.append(dict(loss=float(loss), n=int(aux["n"]), n_marched=int(aux["n_marched"]), n_grad=int(aux["grad"]["n"]),
                            table=tr.r.field.table.clone(), mlp=tr.r.field.mlp.clone(), small=tr.small.clone(), ct=tr.ct.clone(),
                            tau=float(tr.tau)))
        return out
    ref = run(False, False)
    for dc in (False, None):             # (parameters after five Adam steps at 1 536 events: the scatter's float-atomic noise reaches 5e-6
        _same(run(True, dc), ref, tol=2e-4 if trainable else 2e-5, ptol=1e-3)     # of the largest MLP parameter in one run out of six: _same)


def test_packed_batch_feeds_a_replayed_step_with_one_copy(amd, full_table_cache):
    """engine.pack_batch: every field of the event batch a view into one device buffer (same keys, shapes, dtypes, values), so the
    replayed step takes a new batch with ONE copy launch -- same steps as with separate tensors"""
    ops, engine = amd
    g = load_golden("training_step_grad")
    table = full_table_cache(g["table_seed"], g["table_scale"])
    nb = _config_batch(512, 3, int(g["tab_ts"][-1]))
    b = {k: dev(v) for k, v in nb.items()}
    pb = engine.pack_batch(b)
    assert set(pb) == set(b) | {"_pack"}
    for k, v in b.items():
        assert pb[k].dtype == v.dtype and pb[k].shape == v.shape and torch.equal(pb[k], v) and pb[k].data_ptr() % 16 == 0
        assert pb["_pack"].data_ptr() <= pb[k].data_ptr() < pb["_pack"].data_ptr() + pb["_pack"].numel()
    outs = []
    for packed in (False, True):
        tr, _ = _trainer_from_golden(engine, g, table)
        tr.use_graph = True
        tr.t.w_grad, tr.t.err_grad, tr.t.pw_grad = 1e-3, "mape", None
        gen = torch.Generator().manual_seed(5)
        out = []
        for i in range(5):
            nb = _config_batch(2048, 30 + i, int(g["tab_ts"][-1]))
            nb["u_grad"] = torch.rand(2048, generator=gen, dtype=torch.float64).numpy()
            batch = {k: dev(v) for k, v in nb.items()}
            if packed:
                batch = engine.pack_batch(batch)
            j = [dev(torch.rand(2048, generator=gen)) for _ in range(3)]
            loss, aux = tr.step(batch, j[0], j[1], jitter_grad=j[2])
            out.append(dict(loss=float(loss), n=int(aux["n"]), n_marched=int(aux["n_marched"]), n_grad=int(aux["grad"]["n"]),
                            table=tr.r.field.table.clone(), mlp=tr.r.field.mlp.clone(), small=tr.small.clone(), ct=tr.ct.clone(),
                            tau=float(tr.tau)))
        assert tr.graph_replays >= 3
        outs.append(out)
    _same(outs[1], outs[0], ptol=1e-3)
