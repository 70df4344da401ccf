#!/usr/bin/env python
"""Minimal trainer shell over the HIP hot path, driven by the reference's YAML schema.

    python scripts/train.py --config /path/to/configs/train/synthetic.yaml [--dataset-dir DIR] [--out DIR]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 scripts/train.py --config ...

Replaces ``scripts/run.py`` + the PyTorch-Lightning fit loop for training (robust_e_nerf/models/robust_e_nerf.py:
301-517,782-950): epochs x ``limit_train_batches`` steps, MultiStepLR stepped per epoch, occupancy-grid update
every step, dynamic event batch size from the ray-sample budget, one process per GPU with an RCCL all-reduce of
the flat gradient.  Checkpoints carry the reference's state-dict key names so weights move both ways.
``--synthetic N`` trains on the benchmark's synthetic orbit instead of a dataset directory.
"""
import argparse
import math
import os
import sys
import time

import numpy as np
import torch
import yaml

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

NGP_KEYS = {"hash": "mlp_base.0.params", "base.w0": "mlp_base.1.hidden_layers.0.weight",
            "base.b0": "mlp_base.1.hidden_layers.0.bias", "base.wo": "mlp_base.1.output_layer.weight",
            "base.bo": "mlp_base.1.output_layer.bias", "head.w0": "mlp_head.hidden_layers.0.weight",
            "head.b0": "mlp_head.hidden_layers.0.bias", "head.w1": "mlp_head.hidden_layers.1.weight",
            "head.b1": "mlp_head.hidden_layers.1.bias", "head.wo": "mlp_head.output_layer.weight",
            "head.bo": "mlp_head.output_layer.bias"}
PREFIX = "nerf.radiance_field."                       # RobustENeRF.nerf (models/nerf.py) . radiance_field
OCC = "nerf.occupancy_grid."                          # NeRF.occupancy_grid = nerfacc.OccupancyGrid (models/nerf.py:98-102)
CT_KEY = "contrast_threshold.parametrizations.p2n_contrast_threshold_ratio.original"
TAU_KEY = "refractory_period.parametrizations._refractory_period.original"


def softplus_inv(y):
    return float(y + math.log(-math.expm1(-y)))


def field_state_dict(fld, arch, aabb):
    """parameters + persistent buffers of the reference's radiance field module (ngp.py:152, mlp.py:217-219,269)"""
    buf = {PREFIX + "aabb": torch.tensor(aabb, dtype=torch.float32)}
    if arch == "mlp":
        buf[PREFIX + "posi_encoder.scales"] = torch.tensor([2 ** i for i in range(10)])
        buf[PREFIX + "view_encoder.scales"] = torch.tensor([2 ** i for i in range(4)])
        return dict(buf, **{PREFIX + k: v.detach().cpu().clone() for k, v in fld.state_dict(trainable=True).items()})
    sd = dict(buf, **{PREFIX + NGP_KEYS["hash"]: fld.table.detach().cpu().clone()})
    for k, v in fld.trainable_views().items():               # weight_norm: "<layer>.weight_g" / ".weight_v" (ngp.py:207-228)
        name = NGP_KEYS[k[:-2]] + k[-2:] if k[-2:] in ("_g", "_v") else NGP_KEYS[k]
        sd[PREFIX + name] = v.detach().cpu().clone()
    return sd


def load_field_state_dict(fld, arch, sd):
    sd = {k[len(PREFIX):]: v for k, v in sd.items() if k.startswith(PREFIX)}
    if arch == "mlp":
        fld.load(sd)
    else:
        p = {}
        for ours, theirs in NGP_KEYS.items():
            if theirs in sd:
                p[ours] = sd[theirs]
            else:                                            # a weight-normalised layer (ngp.py:207-228)
                p[ours + "_g"], p[ours + "_v"] = sd[theirs + "_g"], sd[theirs + "_v"]
        fld.load(p)


SUPPORTED = {  # what the fused kernels implement = what every shipped configs/train/*.yaml selects
    "ngp": {"dir_encoding": {"degree": 4},
            "mlp_base": {"hidden_activation": "softplus", "density_activation": "shifted_trunc_exp", "n_neurons": 64,
                         "n_hidden_layers": 1, "geo_feat_dim": 15, "weight_norm": False},
            "mlp_head": {"hidden_activation": "softplus", "radiance_activation": "softplus", "n_neurons": 64,
                         "n_hidden_layers": 2, "weight_norm": False}},
    "mlp": {"net_depth": 8, "net_width": 256, "skip_layer": 4, "net_depth_condition": 1, "net_width_condition": 128,
            "hidden_activation": "softplus", "density_activation": "shifted_trunc_exp", "radiance_activation": "softplus",
            "pos_encoder_max_deg": 10, "view_encoder_max_deg": 4, "weight_norm": False},
}


# arch ngp: the activation alternatives of the YAML (models/nerf.py:8-29) run on the exact-f32 fused MLP kernels
NGP_ACTIVATIONS = {("mlp_base", "hidden_activation"): ("softplus", "relu"),
                   ("mlp_base", "density_activation"): ("shifted_trunc_exp", "softplus", "shifted_softplus"),
                   ("mlp_head", "hidden_activation"): ("softplus", "relu"),
                   ("mlp_head", "radiance_activation"): ("softplus", "sigmoid")}
# arch mlp: one hidden activation for the whole MLP (external/mlp.py:258); alternatives run on the per-layer launches
MLP_ACTIVATIONS = {("hidden_activation",): ("softplus", "relu"),
                   ("density_activation",): ("shifted_trunc_exp", "softplus", "shifted_softplus"),
                   ("radiance_activation",): ("softplus", "sigmoid")}
ACTIVATIONS = {"ngp": NGP_ACTIVATIONS, "mlp": MLP_ACTIVATIONS}


def activation_fields(ncfg, arch) -> dict:
    """RenderCfg fields for model.nerf.ngp.mlp_base / mlp_head (arch mlp: model.nerf.mlp) activations (absent keys: the
    shipped values)"""
    if arch == "mlp":
        m = ncfg.get("mlp") or {}
        hid = m.get("hidden_activation", "softplus")
        return dict(base_hidden_activation=hid, head_hidden_activation=hid,
                    density_activation=m.get("density_activation", "shifted_trunc_exp"),
                    radiance_activation=m.get("radiance_activation", "softplus"))
    g = ncfg.get("ngp") or {}
    b, h = g.get("mlp_base") or {}, g.get("mlp_head") or {}
    return dict(base_hidden_activation=b.get("hidden_activation", "softplus"),
                density_activation=b.get("density_activation", "shifted_trunc_exp"),
                head_hidden_activation=h.get("hidden_activation", "softplus"),
                radiance_activation=h.get("radiance_activation", "softplus"))


def weight_norm_flags(ncfg, arch):
    """(mlp_base.weight_norm, mlp_head.weight_norm) of model.nerf.ngp; arch mlp: model.nerf.mlp.weight_norm (one flag)"""
    if arch == "mlp":
        return bool((ncfg.get("mlp") or {}).get("weight_norm", False))
    g = ncfg.get("ngp") or {}
    return (bool((g.get("mlp_base") or {}).get("weight_norm", False)), bool((g.get("mlp_head") or {}).get("weight_norm", False)))


def check_supported(ncfg, arch):
    """Fail loudly on hyper-parameters the HIP kernels do not implement (no silent fallback)."""
    def walk(want, got, path):
        for k, v in want.items():
            if k not in got:
                continue                                        # absent key = the reference default = supported value
            if isinstance(v, dict):
                walk(v, got[k] or {}, path + [k])
            elif tuple(path[1:] + [k]) in ACTIVATIONS[arch]:
                if got[k] not in ACTIVATIONS[arch][tuple(path[1:] + [k])]:
                    raise NotImplementedError(f"model.nerf.{'.'.join(path + [k])} = {got[k]!r}: one of "
                                              f"{ACTIVATIONS[arch][tuple(path[1:] + [k])]} (models/nerf.py:17-29)")
            elif k == "weight_norm" and isinstance(got[k], bool):
                continue                                        # a reparametrisation of the trainable block (NGPField / VanillaField)
            elif got[k] != v:
                raise NotImplementedError(f"model.nerf.{'.'.join(path + [k])} = {got[k]!r}: the MI355X kernels implement {v!r} only")
    walk(SUPPORTED[arch], ncfg.get(arch) or {}, [arch])
    pe = (ncfg.get("ngp") or {}).get("pos_encoding") or {}
    if arch == "ngp" and (pe.get("otype", "HashGrid") not in ("HashGrid", "DenseGrid", "TiledGrid")
                          or pe.get("interpolation", "Linear") != "Linear"
                          or pe.get("n_features_per_level", 2) != 2 or pe.get("n_levels", 16) != 16):
        raise NotImplementedError(f"model.nerf.ngp.pos_encoding {pe}: HashGrid / DenseGrid / TiledGrid, Linear interpolation, "
                                  "16 levels x 2 features only")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", required=True)
    ap.add_argument("--dataset-dir")
    ap.add_argument("--synthetic", type=int, default=0, help="train on N synthetic events (no dataset directory)")
    ap.add_argument("--out", default="runs/train")
    ap.add_argument("--max-epochs", type=int)
    ap.add_argument("--limit-train-batches", type=int)
    ap.add_argument("--resume")
    ap.add_argument("--mlp-bf16", action="store_true",
                    help="bf16 MLP operands, fp32 accumulate / composite (same as float32_matmul_precision: medium)")
    ap.add_argument("--accumulate-grad-batches", type=int, help="overrides trainer.accumulate_grad_batches")
    ap.add_argument("--batch-size-quantum", type=int, default=1,
                    help="round the dynamic batch size (robust_e_nerf.py:907-950) DOWN to a multiple of this many events: step "
                         "shapes then repeat and engine.Trainer replays a captured hipGraph instead of enqueuing ~80 launches "
                         "(1 = the reference's exact int(budget / mean samples per ray), the default)")
    ap.add_argument("--no-validation", action="store_true", help="skip the validation epochs over views/transforms_val.json")
    ap.add_argument("--limit-val-batches", type=int, help="validate on the first N views only")
    args = ap.parse_args()
    cfg = yaml.safe_load(open(args.config))
    rank, local_rank, world = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("LOCAL_RANK", 0), ("WORLD_SIZE", 1)))
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl")
    dev = f"cuda:{local_rank}"
    torch.cuda.set_device(local_rank)
    seed = cfg.get("seed") or 0
    torch.manual_seed(seed)

    from robust_e_nerf_amd import data, engine, ops

    # ---- data: event table in HBM, poses, calibration --------------------------------------------------
    dcfg, mcfg, ncfg = cfg["data"], cfg["model"], cfg["model"]["nerf"]
    if args.synthetic:
        import bench
        tab_ts, tab_pos, tab_quat, Kinv = (torch.from_numpy(a) for a in bench.synthetic_scene())
        ev = bench.synthetic_events(args.synthetic, int(tab_ts[-1]), seed=1)
        events = {k: torch.from_numpy(ev[k]) for k in ("position", "start_ts", "end_ts", "num_pos", "num_neg")}
        pos_ct, neg_ct, tau0, tau_max = 0.25, 0.25, 0.0, torch.tensor(1e5, dtype=torch.float64)
    else:
        root = args.dataset_dir or dcfg["dataset_directory"]
        events = data.load_events(root, dcfg.get("train_dataset_perm_seed"), device=dev)
        tab_ts, tab_pos, tab_quat = data.load_camera_poses(root)
        calib = data.load_calibration(root)
        Kinv = calib["Kinv"]
        pos_ct, neg_ct = float(calib["pos_contrast_threshold"]), float(calib["neg_contrast_threshold"])
        tau_max = data.load_max_refractory_period(root).to(torch.float64)
        tau0 = float(calib["refractory_period"])
        if not (0 <= tau0 < float(tau_max)):               # event_generation_params.py:89,113-130
            import warnings
            warnings.warn(f"Calibrated refractory period ({tau0}) is not in [0, max. refractory period = {float(tau_max)}): "
                          f"redefining it to 0.999 of the max. refractory period")
            tau0 = 0.999 * float(tau_max)
    budget = int(dcfg["train_eff_ray_sample_batch_size"])
    batch_size = max(1, int(dcfg["train_init_eff_batch_size"]) // world)
    batcher = data.EventBatcher(events, batch_size, dev, seed=seed, rank=rank,
                                dataset_ratio=dcfg.get("train_dataset_ratio", 1.0))

    # ---- model ----------------------------------------------------------------------------------------------
    aabb = ncfg["aabb"]
    if aabb == "auto":                                       # robust_e_nerf.py:206-212
        aabb = torch.cat([tab_pos.min(0).values, tab_pos.max(0).values]).tolist()
    ct = {"aabb": ops.AABB, "tanh": ops.UN_BOUNDED_TANH, "sphere": ops.UN_BOUNDED_SPHERE}[ncfg["contraction_type"]]
    step_size = ncfg["render_step_size"]
    if step_size == "auto":                                  # robust_e_nerf.py:220-226
        ext = max(aabb[3 + k] - aabb[k] for k in range(3))
        step_size = ext * math.sqrt(3) / 1024
    og = ncfg["occ_grid"]
    # float32_matmul_precision (scripts/run.py:34-35, torch.set_float32_matmul_precision): "highest" = fp32 products,
    # "medium" = bf16 operands with fp32 accumulation = the bf16 matrix-core mode of the fused MLPs (BASELINE configs[2]);
    # "high" = "each float32 as the sum of two bfloat16" (torch's wording): three bf16 products per fp32 product instead of six
    # (the matrix-core MLP kernels of arch ngp and the fused field of arch mlp)
    precision = cfg.get("float32_matmul_precision", "highest")
    if precision not in ("highest", "high", "medium"):
        raise ValueError(f"float32_matmul_precision: {precision!r} (highest | high | medium)")
    mlp_bf16 = args.mlp_bf16 or precision == "medium"
    rcfg = engine.RenderCfg(aabb=tuple(float(v) for v in aabb), contraction_type=ct, occ_res=(int(og["resolution"]),) * 3,
                            near_plane=ncfg.get("near_plane"), far_plane=ncfg.get("far_plane"),
                            render_step_size=float(step_size), cone_angle=float(ncfg["cone_angle"]),
                            early_stop_eps=float(ncfg["early_stop_eps"]), alpha_thre=float(ncfg["alpha_thre"]),
                            min_modeled_intensity=float(mcfg["min_modeled_intensity"]), occ_thre=float(og["occ_thre"]),
                            ema_decay=float(og["ema_decay"]), warmup_steps=int(og["warmup_steps"]), occ_n=int(og["n"]),
                            mlp_bf16=mlp_bf16, mlp_precision="medium" if mlp_bf16 else precision)
    arch = ncfg.get("arch", "ngp")
    check_supported(ncfg, arch)
    for k_, v_ in activation_fields(ncfg, arch).items():
        setattr(rcfg, k_, v_)
    gen = torch.Generator().manual_seed(seed)

    def lin(o, i):                                           # nn.Linear default init (hidden_init=None, ngp.py:179-185)
        b = 1 / math.sqrt(i)
        return (torch.rand(o, i, generator=gen) * 2 - 1) * b, (torch.rand(o, generator=gen) * 2 - 1) * b
    C = 3 if "channel_idx" in events else 1                 # Bayer sensor -> radiance_dim 3 (robust_e_nerf.py:230-233)
    if arch == "mlp":
        from robust_e_nerf_amd import vanilla
        fld = vanilla.VanillaField(dev, C, weight_norm=weight_norm_flags(ncfg, arch))
        fld.load({k: v for name, o, i in vanilla.layer_shapes(C) for k, v in zip((name + ".weight", name + ".bias"), lin(o, i))})
        renderer = vanilla.VanillaRenderer(fld, rcfg)
    else:
        fld = engine.NGPField(dev, C, ncfg.get("ngp", {}).get("pos_encoding"), weight_norm=weight_norm_flags(ncfg, arch))
        p = {"hash": (torch.rand(fld.n_table, generator=gen) * 2 - 1) * 1e-4}          # tcnn grid init U(+-1e-4)
        for k, (o, i) in {"base.w0": (64, 32), "base.wo": (16, 64), "head.w0": (64, 31), "head.w1": (64, 64), "head.wo": (C, 64)}.items():
            p[k], p[k.replace(".w", ".b")] = lin(o, i)
        fld.load(p)
        renderer = engine.Renderer(fld, rcfg)
    lcfg, ocfg = cfg["loss"], cfg["optimizer"]
    tcfg = engine.TrainCfg(
        err_diff=lcfg["error_fn"]["log_intensity_diff"], w_diff=float(lcfg["weight"]["log_intensity_diff"]),
        pw_diff=lcfg["param_weight"].get("log_intensity_diff"), err_grad=lcfg["error_fn"]["log_intensity_grad"],
        w_grad=float(lcfg["weight"]["log_intensity_grad"]), pw_grad=lcfg["param_weight"].get("log_intensity_grad"),
        lr=float(ocfg["lr"]["default"]), weight_decay=float(lcfg["weight"]["nerf_mlp_weight_decay"]),
        # render_bkgd is a parameter only when alpha_over_white_bg (robust_e_nerf.py:154-159); otherwise no background is
        # composited and the loss is masked with is_valid = opacity > 0 (:868-871): mocap-*, office-maze
        bkgd_is_param=data.alpha_over_white_bg_of(dcfg),
        train_contrast_threshold=not mcfg["contrast_threshold"]["freeze"],
        lr_contrast_threshold=float(ocfg["lr"]["contrast_threshold"]),
        train_refractory_period=not mcfg["refractory_period"]["freeze"],
        relative_lr_refractory_period=float(ocfg["relative_lr"]["refractory_period"]))
    tau_raw = float(tau_max) * torch.logit(torch.tensor(max(tau0, 1e-9) / float(tau_max), dtype=torch.float64)) if tau0 > 0 \
        else torch.tensor(-1e30, dtype=torch.float64)
    tr = engine.Trainer(renderer, tcfg, Kinv=Kinv, tab_ts=tab_ts, tab_pos=tab_pos, tab_quat=tab_quat,
                        p2n_raw=torch.tensor(softplus_inv(pos_ct / neg_ct)), neg_ct=torch.tensor(neg_ct),
                        tau_raw=tau_raw, tau_max=tau_max, bkgd_raw=torch.tensor([softplus_inv(1.0)] * fld.C),
                        world_size=world, process_group=None)
    start_epoch, start_step = 0, 0
    resume_rng = None
    if args.resume:
        ck = torch.load(args.resume, map_location="cpu", weights_only=False)
        rsd = ck["state_dict"]
        load_field_state_dict(fld, arch, rsd)
        if tcfg.bkgd_is_param:
            tr.small[: fld.C] = rsd["nerf.parametrizations.render_bkgd.original"].to(dev, torch.float32).reshape(-1)
        if OCC + "_binary" not in rsd or OCC + "occs" not in rsd:
            raise KeyError(f"{args.resume}: no occupancy grid ({OCC}occs / {OCC}_binary) in the checkpoint")
        renderer.occs.copy_(rsd[OCC + "occs"].to(dev).reshape(-1))
        renderer.binary.copy_(rsd[OCC + "_binary"].reshape(-1).to(torch.uint8).to(dev))
        tr.load_event_params(rsd.get(CT_KEY), rsd.get(TAU_KEY))
        if "optimizer_state" in ck:                          # absent in a reference (PL) checkpoint: fresh moments then
            tr.load_optimizer_state_dict(ck["optimizer_state"])
            start_epoch, start_step = int(ck["epoch"]) + 1, int(ck["global_step"])
            if rank == 0:
                print(f"resumed {args.resume}: epoch {start_epoch}, global_step {start_step}, raw C_p/C_n ratio "
                      f"{float(tr.ct[0]):.4f}, tau {tr.tau:.6g}", flush=True)
            if "batch_size" in ck:
                batch_size = int(ck["batch_size"])
                batcher.set_batch_size(batch_size)
            resume_rng = ck.get("rng_state")

    # ---- fit loop ---------------------------------------------------------------------------------------------
    tcf, sched = cfg["trainer"], cfg["lr_scheduler"]["multi_step_lr"]
    max_epochs = args.max_epochs or int(tcf["max_epochs"])
    per_epoch = args.limit_train_batches or int(tcf["limit_train_batches"])
    log_every = int(tcf.get("log_every_n_steps", 100))
    accum = int(args.accumulate_grad_batches or tcf.get("accumulate_grad_batches", 1) or 1)
    # a new batch size takes effect two batches later (one batch is already pre-fetched, robust_e_nerf.py:925-931)
    from collections import deque
    pending = deque([batch_size])
    jgen = torch.Generator(device=dev).manual_seed(seed + 17 + rank)
    if resume_rng is not None:
        # continue the random streams (event indices / normalized samplers, ray jitters, occupancy-grid refresh) and the
        # in-flight batch-size queue where the checkpointed run left them: a resumed run then equals the uninterrupted one
        # instead of replaying epoch 0's draws.  Every rank's generator states are in the checkpoint (gathered to rank 0 when
        # it was written); a checkpoint written by a different world size re-seeds the ranks it has no state for.
        per_rank = resume_rng.get("per_rank")
        if per_rank is None:                                    # round-3 checkpoints: rank 0's states only
            per_rank = [{"batcher": resume_rng["batcher"], "jitter": resume_rng["jitter"]}]
        if rank < len(per_rank):
            batcher.gen.set_state(per_rank[rank]["batcher"])
            jgen.set_state(per_rank[rank]["jitter"])
        else:
            batcher.gen.manual_seed(seed + rank + 7919 * start_epoch)
            jgen.manual_seed(seed + 17 + rank + 7919 * start_epoch)
            print(f"rank {rank}: no random-stream state in {args.resume} (written by {len(per_rank)} rank(s)): re-seeded",
                  flush=True)
        if resume_rng.get("occ") is not None:                   # identical on every rank by construction
            renderer._occ_gen = torch.Generator(device=dev)
            renderer._occ_gen.set_state(resume_rng["occ"])
        if resume_rng.get("pending"):
            pending = deque(int(b) for b in resume_rng["pending"])
    os.makedirs(args.out, exist_ok=True)
    val_views, val_every = None, int(tcf.get("check_val_every_n_epoch", 1) or 1)
    if not args.synthetic and not args.no_validation and \
            data.has_posed_images(root, data.eval_transforms_stage("val", cfg.get("eval_target"))):
        # datamodule.py:100-134: eval_target's views, eval_dataset_perm_seed, first val_dataset_ratio (x val_eff_batch_size)
        val_views = data.load_eval_views(root, "val", dcfg, cfg.get("eval_target"))
        if rank == 0:
            print(f"validation: {len(val_views['sample_id'])} posed views every {val_every} epoch(s)", flush=True)
    step, t0, rays = start_step, time.perf_counter(), 0
    for epoch in range(start_epoch, max_epochs):
        tr.set_epoch(epoch, tuple(sched["milestones"]), float(sched["gamma"]))
        for bi in range(per_epoch):
            batch = batcher.next()
            B = batch["position"].shape[0]
            j = torch.rand(3, B, device=dev, generator=jgen)
            loss, aux = tr.step(batch, j[0], j[1], global_step=step, jitter_grad=j[2], batch_index=bi,
                                accumulate_grad_batches=accum)
            rays += (3 if tcfg.w_grad > 0 else 2) * B
            nb = tr.update_train_batch_size(aux, budget, accum, bi)              # robust_e_nerf.py:907-950
            if nb is not None:
                pending.append(nb)
            nxt = pending.popleft() if len(pending) > 1 else pending[0]
            if args.batch_size_quantum > 1:
                nxt = max(args.batch_size_quantum, nxt // args.batch_size_quantum * args.batch_size_quantum)
            batcher.set_batch_size(nxt)
            step += (bi + 1) % accum == 0                                        # global_step counts optimiser steps
            if rank == 0 and (bi + 1) % accum == 0 and step % log_every == 0:
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                print(f"epoch {epoch} step {step}  loss {float(loss):.5f}  batch {B}  samples/ray {aux['n'] / max(aux['rays'], 1):.1f}"
                      f"  {rays * world / dt / 1e6:.2f} M rays/s  mem {torch.cuda.memory_allocated() / 2**30:.1f}/"
                      f"{torch.cuda.memory_reserved() / 2**30:.1f} GiB" +
                      (f"  device counts ({tr.device_count_overflows} repeated passes)" if tr.device_counts_ok() else "") +
                      (f"  graph replays {tr.graph_replays} / captures {tr.graph_captures}" if tr.graph_replays else ""), flush=True)
                t0, rays = time.perf_counter(), 0
        # ---- validation epoch (trainer.check_val_every_n_epoch, synthetic.yaml:152-154; robust_e_nerf.py:519-571):
        # the dataset's posed validation views, rendered by all ranks, aligned and scored as the reference does
        if val_views is not None and (epoch + 1) % val_every == 0:
            from robust_e_nerf_amd import evaluation
            bk = torch.nn.functional.softplus(tr.small[: fld.C]) if tcfg.bkgd_is_param else None
            vm = evaluation.evaluate_posed_images(renderer, val_views, bk, rank, world, limit=args.limit_val_batches)
            if rank == 0:
                print(f"epoch {epoch} validation over {vm['n_views']} views: val/l1 {vm['l1']:.5f}  val/psnr {vm['psnr']:.3f} dB", flush=True)
        mine_rng = {"batcher": batcher.gen.get_state().cpu(), "jitter": jgen.get_state().cpu()}
        rank_rng = [mine_rng]
        if world > 1:                                           # every rank's generator states travel to rank 0's file
            rank_rng = [None] * world
            dist.all_gather_object(rank_rng, mine_rng)
        if rank == 0:
            sd = field_state_dict(fld, arch, rcfg.aabb)
            sd[CT_KEY] = tr.ct[:1].detach().cpu().clone()
            sd[TAU_KEY] = tr.tau_raw.detach().clone()
            if tcfg.bkgd_is_param:                      # models/nerf.py:81-88 (softplus-parametrised parameter)
                sd["nerf.parametrizations.render_bkgd.original"] = tr.small[: fld.C].detach().cpu().clone()
            # nerfacc.OccupancyGrid persistent buffers (models/nerf.py:98-102; nerfacc 0.3.x registers _roi_aabb, _binary,
            # resolution, occs -- grid_coords / grid_indices are non-persistent)
            sd[OCC + "_roi_aabb"] = torch.tensor(rcfg.aabb, dtype=torch.float32)
            sd[OCC + "_binary"] = renderer.binary.detach().cpu().bool().view(*rcfg.occ_res)
            sd[OCC + "resolution"] = torch.tensor(rcfg.occ_res, dtype=torch.int32)
            sd[OCC + "occs"] = renderer.occs.detach().cpu().clone()
            rng = {"per_rank": rank_rng, "pending": list(pending),
                   "occ": renderer._occ_gen.get_state() if renderer._occ_gen is not None else None}
            torch.save({"state_dict": sd, "epoch": epoch, "global_step": step, "optimizer_state": tr.optimizer_state_dict(),
                        "batch_size": batcher.batch_size, "rng_state": rng}, os.path.join(args.out, "last.ckpt"))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
