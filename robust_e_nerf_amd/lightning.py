"""Optional PyTorch-Lightning adapter: the reference drives training with a ``LightningModule``
(``robust_e_nerf/models/robust_e_nerf.py:16-517``, fitted by ``scripts/run.py:66-97``).  Where ``pytorch_lightning`` is
installed, ``make_module(trainer)`` wraps an ``engine.Trainer`` in a LightningModule with manual optimisation, so the
reference's data module, loggers and ``ModelCheckpoint`` keep working; where it is not (this image), importing this file
is harmless and ``make_module`` raises ImportError.  ``scripts/train.py`` is the dependency-free loop.
"""
from __future__ import annotations


def make_module(trainer, train_cfg: dict | None = None):
    """trainer: engine.Trainer.  -> a LightningModule whose training_step runs one fused HIP step (forward, loss,
    backward, all-reduce, Adam) on the reference's batch layout (App. B.1: ``batch["event"]`` / ``batch["normalized"]``
    with a leading dim of 1)."""
    try:
        import pytorch_lightning as pl
    except ImportError as e:                                      # pragma: no cover - PL is absent from the build image
        raise ImportError("pytorch_lightning is not installed; use scripts/train.py (same YAML schema)") from e
    import torch

    class RobustENeRFHip(pl.LightningModule):
        def __init__(self):
            super().__init__()
            self.automatic_optimization = False                   # the fused step owns backward + optimiser
            self.hip = trainer
            self._dummy = torch.nn.Parameter(torch.zeros(1))      # Lightning wants at least one parameter / optimiser

        def configure_optimizers(self):
            return torch.optim.SGD([self._dummy], lr=0.0)

        def on_train_epoch_start(self):
            ms = (train_cfg or {}).get("milestones", (20, 30, 36))
            self.hip.set_epoch(self.current_epoch, tuple(ms), float((train_cfg or {}).get("gamma", 0.33)))

        def training_step(self, batch, batch_index):
            ev, nm = batch["event"], batch["normalized"]
            dev = self.hip.r.field.flat.device
            b = {k: ev[k][0].to(dev).contiguous() for k in ("position", "start_ts", "end_ts", "num_pos", "num_neg")}
            b["position"] = b["position"].to(torch.float32)
            if "channel_idx" in ev:
                b["channel_idx"] = ev["channel_idx"][0].to(dev)
            b["u_ts_diff"], b["u_diff_start"] = nm["ts_diff"][0].to(dev), nm["diff_start_ts"][0].to(dev)
            if "grad_ts" in nm:
                b["u_grad"] = nm["grad_ts"][0].to(dev)
            B = b["position"].shape[0]
            j = torch.rand(3, B, device=dev)
            loss, aux = self.hip.step(b, j[0], j[1], global_step=self.global_step, jitter_grad=j[2], batch_index=batch_index,
                                      accumulate_grad_batches=self.trainer.accumulate_grad_batches)
            self.log("train/loss", loss.detach())
            self.log("train/mean_num_samples_per_ray", aux["n"] / max(aux["rays"], 1))
            new = self.hip.update_train_batch_size(aux, (train_cfg or {}).get("eff_ray_sample_batch_size", 1 << 20),
                                                   self.trainer.accumulate_grad_batches, batch_index)
            if new is not None:                                   # robust_e_nerf.py:941-948
                dm = self.trainer.datamodule
                dm.train_dataset.batch_size = new
                for smp in dm.train_normalized_sampler.datasets:
                    smp.size = new
            return loss.detach()

    return RobustENeRFHip()
