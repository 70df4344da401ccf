"""robust_e_nerf_amd/lightning.py executed over a stand-in ``pytorch_lightning`` (PL is absent from this image): the
adapter must turn the reference's batch layout (``batch["event"]`` / ``batch["normalized"]`` with a
leading dim of 1, robust_e_nerf/data/datamodule.py:198-230) into one fused step, follow Lightning's
``accumulate_grad_batches`` / ``global_step`` / ``current_epoch``, and write the dynamic batch size back into the data
module exactly where the reference does (robust_e_nerf/models/robust_e_nerf.py:941-948).
CPU: a recording stand-in for engine.Trainer.  GPU: the real Trainer, against Trainer.step itself."""
import sys
import types

import numpy as np
import pytest
import torch


class _StubLightningModule(torch.nn.Module):
    """the slice of pl.LightningModule the adapter touches"""

    def __init__(self):
        super().__init__()
        self.logged, self.trainer, self.global_step, self.current_epoch = {}, None, 0, 0

    def log(self, name, value):
        self.logged[name] = float(value)


@pytest.fixture()
def stub_pl(monkeypatch):
    pl = types.ModuleType("pytorch_lightning")
    pl.LightningModule = _StubLightningModule
    monkeypatch.setitem(sys.modules, "pytorch_lightning", pl)
    return pl


def _reference_batch(B, seed=0, bayer=False):
    g = np.random.default_rng(seed)
    end = g.integers(20_000_000, 190_000_000, B)
    ev = {"position": torch.from_numpy(g.integers(0, 64, (1, B, 2)).astype(np.float32)),
          "start_ts": torch.from_numpy((end - g.integers(200_000, 5_000_000, B))[None]),
          "end_ts": torch.from_numpy(end[None]),
          "num_pos": torch.from_numpy((g.random(B) < 0.5).astype(np.int64)[None]),
          "num_neg": torch.zeros(1, B, dtype=torch.int64)}
    ev["num_neg"] = 1 - ev["num_pos"]
    if bayer:
        ev["channel_idx"] = torch.from_numpy(g.integers(0, 3, (1, B)).astype(np.uint8))
    nm = {"ts_diff": torch.ones(1, B, dtype=torch.float64), "diff_start_ts": torch.from_numpy(g.random((1, B))),
          "grad_ts": torch.from_numpy(g.random((1, B)))}
    return {"event": ev, "normalized": nm}


class _FakeDataModule:
    def __init__(self, B):
        self.train_dataset = types.SimpleNamespace(batch_size=B)
        self.train_normalized_sampler = types.SimpleNamespace(datasets=[types.SimpleNamespace(size=B) for _ in range(3)])


def test_lightning_adapter_over_stub_pl_cpu(stub_pl):
    from robust_e_nerf_amd import lightning

    class Recorder:                                      # duck-typed engine.Trainer
        def __init__(self):
            self.calls = []
            self.r = types.SimpleNamespace(field=types.SimpleNamespace(flat=torch.zeros(1)))

        def set_epoch(self, epoch, milestones, gamma):
            self.calls.append(("set_epoch", epoch, milestones, gamma))

        def step(self, b, j0, j1, **kw):
            self.calls.append(("step", b, j0, j1, kw))
            return torch.tensor(0.25), {"n": 640, "rays": 2 * b["position"].shape[0]}

        def update_train_batch_size(self, aux, budget, accum, bi):
            self.calls.append(("batch_size", budget, accum, bi))
            return 77

    rec = Recorder()
    mod = lightning.make_module(rec, dict(milestones=(2, 3), gamma=0.5, eff_ray_sample_batch_size=1 << 16))
    assert mod.automatic_optimization is False
    assert isinstance(mod.configure_optimizers(), torch.optim.Optimizer)
    B = 32
    dm = _FakeDataModule(B)
    mod.trainer = types.SimpleNamespace(accumulate_grad_batches=2, datamodule=dm)
    mod.current_epoch, mod.global_step = 3, 48
    mod.on_train_epoch_start()
    assert rec.calls[-1] == ("set_epoch", 3, (2, 3), 0.5)
    batch = _reference_batch(B, bayer=True)
    out = mod.training_step(batch, 5)
    assert float(out) == 0.25 and not out.requires_grad
    _, b, j0, j1, kw = rec.calls[-2]
    assert set(b) == {"position", "start_ts", "end_ts", "num_pos", "num_neg", "channel_idx", "u_ts_diff", "u_diff_start", "u_grad"}
    assert b["position"].shape == (B, 2) and b["position"].dtype == torch.float32 and b["start_ts"].dtype == torch.int64
    assert torch.equal(b["u_diff_start"], batch["normalized"]["diff_start_ts"][0]) and b["u_ts_diff"].dtype == torch.float64
    assert j0.shape == (B,) and j1.shape == (B,) and kw["jitter_grad"].shape == (B,)
    assert kw["global_step"] == 48 and kw["batch_index"] == 5 and kw["accumulate_grad_batches"] == 2
    assert rec.calls[-1] == ("batch_size", 1 << 16, 2, 5)
    assert mod.logged["train/loss"] == 0.25 and mod.logged["train/mean_num_samples_per_ray"] == 640 / (2 * B)
    # robust_e_nerf.py:941-948: the new size goes to the event dataset and to every normalized sampler
    assert dm.train_dataset.batch_size == 77 and [s.size for s in dm.train_normalized_sampler.datasets] == [77] * 3


def test_make_module_without_lightning_raises(monkeypatch):
    from robust_e_nerf_amd import lightning
    monkeypatch.setitem(sys.modules, "pytorch_lightning", None)
    with pytest.raises(ImportError):
        lightning.make_module(object())


@pytest.mark.gpu
def test_lightning_adapter_runs_the_fused_step_gpu(stub_pl):
    """the adapter over a real engine.Trainer: same loss and same parameters as Trainer.step on the same batch"""
    import bench
    from robust_e_nerf_amd import engine, lightning
    dev = "cuda:0"
    ts, pos, quat, Kinv = bench.synthetic_scene(201)
    T = torch.from_numpy

    def make():
        torch.manual_seed(0)
        fld = engine.NGPField(dev)
        fld.flat.uniform_(-0.1, 0.1)
        r = engine.Renderer(fld, engine.RenderCfg(sampler="uniform", n_uniform=16))
        return engine.Trainer(r, engine.TrainCfg(w_grad=1e-3), Kinv=T(Kinv), tab_ts=T(ts), tab_pos=T(pos), tab_quat=T(quat),
                              p2n_raw=torch.tensor(0.5413), neg_ct=torch.tensor(0.25),
                              tau_raw=torch.tensor(0.0, dtype=torch.float64), tau_max=torch.tensor(1e5),
                              bkgd_raw=torch.tensor([0.5413]))
    B = 256
    batch = _reference_batch(B, seed=3)
    tr_a, tr_b = make(), make()
    mod = lightning.make_module(tr_a, dict(eff_ray_sample_batch_size=1 << 14))
    dm = _FakeDataModule(B)
    mod.trainer = types.SimpleNamespace(accumulate_grad_batches=1, datamodule=dm)
    mod.global_step = 16
    mod.on_train_epoch_start()
    torch.manual_seed(5)
    loss_a = mod.training_step(batch, 0)
    # the same step by hand
    ev, nm = batch["event"], batch["normalized"]
    b = {k: ev[k][0].to(dev).contiguous() for k in ("position", "start_ts", "end_ts", "num_pos", "num_neg")}
    b["u_ts_diff"], b["u_diff_start"], b["u_grad"] = nm["ts_diff"][0].to(dev), nm["diff_start_ts"][0].to(dev), nm["grad_ts"][0].to(dev)
    torch.manual_seed(5)
    j = torch.rand(3, B, device=dev)
    loss_b, aux = tr_b.step(b, j[0], j[1], global_step=16, jitter_grad=j[2], batch_index=0, accumulate_grad_batches=1)
    assert abs(float(loss_a) - float(loss_b)) <= 1e-6 * abs(float(loss_b))
    # same parameters after the step (not bit for bit: the rare float atomics of the binned scatter's overflow path make
    # two runs differ in the last bit of a few table gradients, and Adam's first step is lr * sign-like)
    diff = (tr_a.r.field.flat - tr_b.r.field.flat).abs()
    assert float((diff > 1e-6).float().mean()) < 1e-4, float(diff.max())
    assert dm.train_dataset.batch_size == tr_b.update_train_batch_size(aux, 1 << 14, 1, 0)
