"""Render views from a checkpoint: the inference side of the reference's `evaluation_step` / `evaluation_epoch_end`
(robust_e_nerf/models/robust_e_nerf.py:533-571, 634-677) for a checkpoint written by scripts/train.py or by the
reference (same state-dict keys).

    python scripts/render.py --config <YAML> --ckpt runs/train/last.ckpt --out renders/ \
        [--dataset-dir DIR | --synthetic] [--every 50] [--height 260 --width 346] [--gt-dir DIR]

Poses come from the dataset's camera_poses.npz (every `--every`-th pose) or from the synthetic benchmark orbit; the
intrinsics from camera_calibration.npz.  Each view is written as <index>.png (8-bit, intensity clipped to [0, 1] after an
optional gain) and all of them as views.npz (float32 intensity, opacity, z-depth).  With --gt-dir (files <index>.npy:
linear intensity images of the same size) the prediction is aligned to the ground truth by the reference's affine fit in
log space and the PSNR of every view and their mean are printed (metric.py:60-72).
"""
import argparse
import math
import os
import sys

import numpy as np
import torch
import yaml

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "scripts"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", required=True)
    ap.add_argument("--ckpt", required=True)
    ap.add_argument("--out", default="renders")
    ap.add_argument("--dataset-dir")
    ap.add_argument("--synthetic", action="store_true", help="poses / intrinsics of the synthetic benchmark orbit")
    ap.add_argument("--every", type=int, default=100, help="render every N-th pose of the trajectory")
    ap.add_argument("--height", type=int)
    ap.add_argument("--width", type=int)
    ap.add_argument("--gain", type=float, default=1.0, help="multiplies the intensity before the 8-bit PNG is written")
    ap.add_argument("--gt-dir", help="<index>.npy ground-truth intensity images: aligned PSNR is reported")
    ap.add_argument("--stage", choices=["val", "test"],
                    help="evaluate on the dataset's posed images (views/transforms_<stage>.json, the reference's PosedImage "
                         "layout) instead of rendering along the trajectory: aligned L1 / PSNR per view and their means, as "
                         "the reference's validation / test epochs (robust_e_nerf.py:519-696)")
    ap.add_argument("--chunk", type=int, help="rays per render call (default: whole image for arch ngp, 16 384 for arch mlp)")
    args = ap.parse_args()

    import train as cli
    from robust_e_nerf_amd import data, engine, evaluation, ops
    cfg = yaml.safe_load(open(args.config))
    dev = "cuda:0"
    torch.cuda.set_device(0)
    dcfg, mcfg, ncfg = cfg["data"], cfg["model"], cfg["model"]["nerf"]
    if args.synthetic:
        import bench
        tab_ts, tab_pos, tab_quat, Kinv = (torch.from_numpy(a) for a in bench.synthetic_scene())
        height, width = args.height or 260, args.width or 346
    else:
        root = args.dataset_dir or dcfg["dataset_directory"]
        tab_ts, tab_pos, tab_quat = data.load_camera_poses(root)
        calib = data.load_calibration(root)
        Kinv = calib["Kinv"]
        raw = np.load(os.path.join(root, data.CAMERA_CALIBRATION))
        height, width = args.height or int(raw["img_height"]), args.width or int(raw["img_width"])
    aabb = ncfg["aabb"]
    if aabb == "auto":
        aabb = torch.cat([tab_pos.min(0).values, tab_pos.max(0).values]).tolist()
    ct = {"aabb": ops.AABB, "tanh": ops.UN_BOUNDED_TANH, "sphere": ops.UN_BOUNDED_SPHERE}[ncfg["contraction_type"]]
    step_size = ncfg["render_step_size"]
    if step_size == "auto":
        step_size = max(aabb[3 + k] - aabb[k] for k in range(3)) * math.sqrt(3) / 1024
    og = ncfg["occ_grid"]
    rcfg = engine.RenderCfg(aabb=tuple(float(v) for v in aabb), contraction_type=ct, occ_res=(int(og["resolution"]),) * 3,
                            near_plane=ncfg.get("near_plane"), far_plane=ncfg.get("far_plane"), render_step_size=float(step_size),
                            cone_angle=float(ncfg["cone_angle"]), early_stop_eps=float(ncfg["early_stop_eps"]),
                            alpha_thre=float(ncfg["alpha_thre"]), min_modeled_intensity=float(mcfg["min_modeled_intensity"]),
                            mlp_precision=cfg.get("float32_matmul_precision", "highest"))
    arch = ncfg.get("arch", "ngp")
    cli.check_supported(ncfg, arch)
    for k_, v_ in cli.activation_fields(ncfg, arch).items():
        setattr(rcfg, k_, v_)
    sd = torch.load(args.ckpt, map_location="cpu", weights_only=False)["state_dict"]
    C = int(sd[cli.PREFIX + ("mlp.rgb_layer.output_layer.bias" if arch == "mlp" else cli.NGP_KEYS["head.bo"])].numel())
    if arch == "mlp":
        from robust_e_nerf_amd import vanilla
        fld = vanilla.VanillaField(dev, C, weight_norm=cli.weight_norm_flags(ncfg, arch))
        cli.load_field_state_dict(fld, arch, sd)
        r = vanilla.VanillaRenderer(fld, rcfg)
    else:
        fld = engine.NGPField(dev, C, ncfg.get("ngp", {}).get("pos_encoding"), weight_norm=cli.weight_norm_flags(ncfg, arch))
        cli.load_field_state_dict(fld, arch, sd)
        r = engine.Renderer(fld, rcfg)
    r.binary.copy_(sd[cli.OCC + "_binary"].reshape(-1).to(torch.uint8).to(dev))
    bk_key = "nerf.parametrizations.render_bkgd.original"
    bkgd = torch.nn.functional.softplus(sd[bk_key].to(dev, torch.float32).reshape(-1)) if bk_key in sd else None

    os.makedirs(args.out, exist_ok=True)
    from PIL import Image
    if args.stage:
        if args.synthetic:
            raise SystemExit("--stage needs a dataset directory with a views/ folder")
        posed = data.load_eval_views(root, args.stage, dcfg, cfg.get("eval_target"))      # datamodule.py:100-134
        m = evaluation.evaluate_posed_images(r, posed, bkgd, chunk=args.chunk)
        for sid, (l1v, ps) in zip(posed["sample_id"], m["per_view"].tolist()):
            print(f"{args.stage} view {sid}: l1 {l1v:.5f}  psnr {ps:.2f} dB")
        np.savez(os.path.join(args.out, f"{args.stage}_metrics.npz"), sample_id=np.array(posed["sample_id"]),
                 l1_psnr=m["per_view"].numpy())
        print(f"{args.stage}: {m['n_views']} views, mean l1 {m['l1']:.5f}, mean PSNR {m['psnr']:.2f} dB", flush=True)
        return
    Kinv_d = Kinv.to(dev, torch.float32)
    idx = list(range(0, tab_ts.shape[0], max(1, args.every)))
    pos_all, rot_all = ops.trajectory(tab_ts[idx].to(dev, torch.float64), tab_ts.to(dev), tab_pos.to(dev), tab_quat.to(dev))
    imgs, opacs, depths, scores = [], [], [], []
    for k, i in enumerate(idx):
        img, opac, depth = evaluation.render_image(r, Kinv_d, pos_all[k], rot_all[k], height, width, bkgd=bkgd, chunk=args.chunk)
        imgs.append(img.cpu()); opacs.append(opac.cpu()); depths.append(depth.cpu())
        shown = img
        if args.gt_dir:
            gt = torch.from_numpy(np.load(os.path.join(args.gt_dir, f"{i}.npy"))).to(dev, torch.float32)
            shown = evaluation.affine_align_log(img, gt + rcfg.min_modeled_intensity)
            scores.append(evaluation.psnr(shown, gt + rcfg.min_modeled_intensity, 1.0))
            print(f"view {i}: PSNR {scores[-1]:.2f} dB", flush=True)
        u8 = (shown * args.gain).clamp(0, 1).mul(255).round().byte().cpu().numpy()
        if u8.ndim == 3:
            u8 = np.transpose(u8, (1, 2, 0))
        Image.fromarray(u8, mode="L" if u8.ndim == 2 else "RGB").save(os.path.join(args.out, f"{i}.png"))
    np.savez(os.path.join(args.out, "views.npz"), index=np.array(idx), intensity=torch.stack(imgs).numpy(),
             opacity=torch.stack(opacs).numpy(), depth=torch.stack(depths).numpy())
    msg = f"{len(idx)} views of {height} x {width} written to {args.out}"
    if scores:
        msg += f"; mean PSNR {sum(scores) / len(scores):.2f} dB"
    print(msg, flush=True)


if __name__ == "__main__":
    main()
