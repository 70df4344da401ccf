"""End-to-end check on a simulated event sequence with a known scene (GPU only).

1. simulate: an analytic volumetric scene (three textured blobs) is rendered at every pose of the benchmark
   orbit (346 x 260, 1 ms spacing) and turned into events by per-pixel log-intensity thresholding with
   linearly interpolated timestamps (noise-free ESIM-style), written in the reference's on-disk format
   (raw_events.npz, camera_poses.npz, camera_calibration.npz; data/datasets.py:14-62);
2. train: scripts/train.py on that directory with the reference's YAML schema (occupancy-grid sampling,
   l_diff + l_grad, dynamic batch size, MultiStepLR, Adam);
3. evaluate: novel views off the training orbit, affine log-intensity alignment + PSNR against the analytic
   scene (robust_e_nerf.py:634-677, metric.py:68-72).

    python tools/e2e_synthetic.py --out gpurun_out/e2e [--epochs 6 --steps-per-epoch 500]
"""
import argparse, json, math, os, subprocess, sys, time
import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench
from robust_e_nerf_amd import engine, evaluation, ops

DEV = "cuda:0"
AABB = (-1.5, -1.5, -1.5, 1.5, 1.5, 1.5)
H, W = 260, 346
BKGD, CT = 0.55, 0.25


def scene(x):
    """analytic density and emitted intensity at world points (n, 3)"""
    def blob(c, r, amp):
        return amp * torch.sigmoid((r - (x - torch.tensor(c, device=x.device)).norm(dim=-1)) / 0.02)
    sigma = blob((0.0, 0.0, 0.0), 0.55, 40.0) + blob((0.75, 0.35, 0.25), 0.28, 40.0) + blob((-0.5, -0.7, -0.3), 0.33, 40.0)
    tex = torch.sin(9.0 * x[:, 0] + 0.5) * torch.sin(9.0 * x[:, 1] + 1.0) * torch.sin(9.0 * x[:, 2] + 2.0)
    return sigma, 0.12 + 0.75 * (0.5 + 0.5 * tex)


def render_scene(Kinv, pos, rot, S=192):
    """(H, W) intensity of the analytic scene from one pose: fixed-S midpoint quadrature through the aabb."""
    px = evaluation.pixel_grid(H, W, DEV).reshape(-1, 2)
    n = px.shape[0]
    o, d = ops.raygen(Kinv, px.contiguous(), pos.reshape(1, 3).expand(n, 3).contiguous(),
                      rot.reshape(1, 3, 3).expand(n, 3, 3).contiguous())
    t0, t1 = ops.ray_aabb_intersect(o, d, AABB, None, None)
    hit = t0 < t1
    k = (torch.arange(S, device=DEV) + 0.5) / S
    tm = t0[:, None] + (t1 - t0)[:, None] * k[None]
    dt = ((t1 - t0) / S).clamp(min=0)[:, None]
    x = o[:, None, :] + tm[..., None] * d[:, None, :]
    sigma, c = scene(x.reshape(-1, 3))
    sigma, c = sigma.view(n, S) * hit[:, None], c.view(n, S)
    tau = sigma * dt
    T = torch.exp(-(torch.cumsum(tau, 1) - tau))
    w = T * (1 - torch.exp(-tau))
    return ((w * c).sum(1) + BKGD * (1 - w.sum(1))).view(H, W)


def rotmats(quat_xyzw):
    from scipy.spatial.transform import Rotation
    return torch.from_numpy(Rotation.from_quat(quat_xyzw).as_matrix().astype(np.float32))


def write_val_views(out_dir, Kinv, n=6, bit_depth=16):
    """the novel views as the reference's posed-image layout (data/datasets.py:376-690): views/transforms_val.json with
    OpenGL camera-to-world matrices and explicit intrinsics, images as 16-bit grey PNGs (a real-capture style dataset:
    quantized, no renderer_params.npz)"""
    import json
    from PIL import Image
    os.makedirs(os.path.join(out_dir, "views", "val"), exist_ok=True)
    Kinv_d = torch.from_numpy(Kinv).to(DEV) if not torch.is_tensor(Kinv) else Kinv
    frames, levels = [], 2 ** bit_depth
    for j, (pos, rot) in enumerate(novel_views(n)):
        img = render_scene(Kinv_d, pos, rot)                                 # linear intensity in (0, 1)
        q = (img * levels).floor().clamp(0, levels - 1).to(torch.int32).cpu().numpy().astype(np.uint16)
        Image.fromarray(q).save(os.path.join(out_dir, "views", "val", f"r_{j}.png"))
        T = np.eye(4)
        T[:3, :3] = rot.cpu().numpy().astype(np.float64) @ np.diag([1.0, -1.0, -1.0])      # common -> OpenGL camera frame
        T[:3, 3] = pos.cpu().numpy()
        frames.append(dict(file_path=f"./val/r_{j}", transform_matrix=T.tolist()))
    K = np.linalg.inv(Kinv_d.cpu().numpy().astype(np.float64))
    json.dump(dict(intrinsics=K.tolist(), frames=frames), open(os.path.join(out_dir, "views", "transforms_val.json"), "w"))


def simulate(out_dir, n_poses=2001, val_views=6):
    tab_ts, tab_pos, tab_quat, Kinv = bench.synthetic_scene(n_poses)
    Kinv_d, rot = torch.from_numpy(Kinv).to(DEV), rotmats(tab_quat).to(DEV)
    pos = torch.from_numpy(tab_pos).to(DEV)
    eps = 1e-3
    pix = torch.arange(H * W, device=DEV)
    ev_pix, ev_t, ev_pol = [], [], []
    L_prev = L_ref = None
    for k in range(len(tab_ts)):
        L = (render_scene(Kinv_d, pos[k], rot[k]) + eps).log().reshape(-1).double()
        if k == 0:
            L_prev, L_ref = L, L.clone()
            continue
        tp, tk = float(tab_ts[k - 1]), float(tab_ts[k])
        while True:
            up, dn = L - L_ref >= CT, L_ref - L >= CT
            m = up | dn
            if not bool(m.any()):
                break
            level = torch.where(up, L_ref + CT, L_ref - CT)
            frac = ((level - L_prev) / (L - L_prev)).clamp(0, 1)
            ev_pix.append(pix[m]); ev_t.append((tp + frac[m] * (tk - tp)).round().long()); ev_pol.append(up[m])
            L_ref = torch.where(m, level, L_ref)
        L_prev = L
    p, t, pol = torch.cat(ev_pix), torch.cat(ev_t), torch.cat(ev_pol)
    order = torch.argsort(t, stable=True)
    p, t, pol = p[order].cpu().numpy(), t[order].cpu().numpy(), pol[order].cpu().numpy()
    os.makedirs(out_dir, exist_ok=True)
    np.savez(os.path.join(out_dir, "raw_events.npz"), position=np.stack([p % W, p // W], -1).astype(np.uint16),
             timestamp=t.astype(np.int64), polarity=pol)
    np.savez(os.path.join(out_dir, "camera_poses.npz"), T_wc_timestamp=tab_ts, T_wc_position=tab_pos, T_wc_orientation=tab_quat)
    np.savez(os.path.join(out_dir, "camera_calibration.npz"), intrinsics=np.linalg.inv(Kinv.astype(np.float64)),
             img_width=W, img_height=H, distortion_params=np.zeros(4), distortion_model="plumb_bob", bayer_pattern="",
             pos_contrast_threshold=CT, neg_contrast_threshold=CT, refractory_period=0.0)
    if val_views:
        write_val_views(out_dir, Kinv_d, val_views)
    return len(t), Kinv_d


def novel_views(n=6):
    """poses off the training orbit (other heights and radii), looking at the origin"""
    out = []
    for j in range(n):
        a = 2 * math.pi * (j + 0.37) / n
        p = np.array([3.6 * math.cos(a), 3.6 * math.sin(a), (-1.0, 1.2)[j % 2]])
        f = -p / np.linalg.norm(p)
        r = np.cross(f, [0.0, 0.0, 1.0]); r /= np.linalg.norm(r)
        out.append((torch.tensor(p, dtype=torch.float32, device=DEV),
                    torch.tensor(np.stack([r, np.cross(f, r), f], -1), dtype=torch.float32, device=DEV)))
    return out


def evaluate(ckpt, Kinv_d, cfg, png=None):
    sys.path.insert(0, os.path.join(REPO, "scripts"))
    import train as cli
    sd = torch.load(ckpt, map_location="cpu")["state_dict"]
    arch = cfg["model"]["nerf"].get("arch", "ngp")
    rcfg = engine.RenderCfg(aabb=AABB, sampler="occgrid", render_step_size=3 * math.sqrt(3) / 1024)
    if arch == "mlp":
        from robust_e_nerf_amd import vanilla
        fld = vanilla.VanillaField(DEV, 1)
        cli.load_field_state_dict(fld, "mlp", sd)
        r = vanilla.VanillaRenderer(fld, rcfg)
    else:
        fld = engine.NGPField(DEV, 1)
        cli.load_field_state_dict(fld, "ngp", sd)
        r = engine.Renderer(fld, rcfg)
    r.binary.copy_(sd[cli.OCC + "_binary"].reshape(-1).to(torch.uint8).to(DEV))
    bk = torch.nn.functional.softplus(sd["nerf.parametrizations.render_bkgd.original"].to(DEV))
    gts, preds = [], []
    for pos, rot in novel_views():
        gts.append(render_scene(Kinv_d, pos, rot) + 1e-3)
        preds.append(evaluation.render_image(r, Kinv_d, pos, rot, H, W, bkgd=bk)[0])
    # the reference's epoch metric: ONE affine fit in log space over all views, then per-view PSNR (robust_e_nerf.py:634-696)
    gts, preds = torch.stack(gts), torch.stack(preds)
    per_view, (a, b) = evaluation.align_and_score(preds, gts, 1.0)
    scores = per_view[:, 1].tolist()
    tiles = [torch.cat([gts[v], evaluation.apply_affine(preds[v], a, b)], 1) for v in range(len(gts))]
    if png:                                                        # rows: views; left = analytic scene, right = prediction
        from PIL import Image
        img = (torch.cat(tiles[:3], 0).clamp(0, 1) * 255).round().byte().cpu().numpy()
        Image.fromarray(img, mode="L").save(png)
    return scores


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/e2e")
    ap.add_argument("--epochs", type=int, default=6)
    ap.add_argument("--steps-per-epoch", type=int, default=500)
    ap.add_argument("--arch", default="ngp", choices=["ngp", "mlp"])
    ap.add_argument("--budget", type=int, default=1 << 20, help="train_eff_ray_sample_batch_size")
    ap.add_argument("--batch-size-quantum", type=int, default=1, help="scripts/train.py --batch-size-quantum")
    ap.add_argument("--precision", default="highest", choices=["highest", "high", "medium"], help="float32_matmul_precision of the YAML")
    args = ap.parse_args()
    data_dir = os.path.join(args.out, "dataset")
    t0 = time.perf_counter()
    n_events, Kinv_d = simulate(data_dir)
    t_sim = time.perf_counter() - t0
    print(f"simulated {n_events} events in {t_sim:.1f} s", flush=True)
    import yaml
    cfg = yaml.safe_load(open(os.path.join(REPO, "configs", "synthetic_smoke.yaml")))
    cfg["data"].update(dataset_directory=data_dir, train_init_eff_batch_size=65536 if args.arch == "ngp" else 2048, train_eff_ray_sample_batch_size=args.budget)
    cfg["model"]["nerf"]["arch"] = args.arch
    cfg["float32_matmul_precision"] = args.precision
    cfg["trainer"].update(max_epochs=args.epochs, limit_train_batches=args.steps_per_epoch, log_every_n_steps=100)
    cfg["lr_scheduler"]["multi_step_lr"]["milestones"] = [max(1, args.epochs // 2), max(2, 3 * args.epochs // 4), max(3, 9 * args.epochs // 10)]
    cfg_path = os.path.join(args.out, "train.yaml")
    yaml.safe_dump(cfg, open(cfg_path, "w"))
    init = subprocess.run([sys.executable, os.path.join(REPO, "scripts", "train.py"), "--config", cfg_path, "--out",
                           os.path.join(args.out, "init"), "--max-epochs", "1", "--limit-train-batches", "1"],
                          capture_output=True, text=True)
    if init.returncode:
        raise SystemExit(init.stderr[-2000:])
    scores0 = evaluate(os.path.join(args.out, "init", "last.ckpt"), Kinv_d, cfg)
    t0 = time.perf_counter()
    log = subprocess.run([sys.executable, os.path.join(REPO, "scripts", "train.py"), "--config", cfg_path, "--out",
                          os.path.join(args.out, "run"), "--batch-size-quantum", str(args.batch_size_quantum)], capture_output=True, text=True)
    t_train = time.perf_counter() - t0
    print(log.stdout[-3000:], log.stderr[-2000:] if log.returncode else "", flush=True)
    if log.returncode:
        open(os.path.join(args.out, "train_stderr.txt"), "w").write(log.stderr)
        raise SystemExit("training failed")
    scores = evaluate(os.path.join(args.out, "run", "last.ckpt"), Kinv_d, cfg, png=os.path.join(args.out, "e2e_novel_views.png"))
    # the same views through the reference's dataset layout: views/transforms_val.json (16-bit PNGs) -> scripts/render.py --stage val
    rv = subprocess.run([sys.executable, os.path.join(REPO, "scripts", "render.py"), "--config", cfg_path, "--ckpt",
                         os.path.join(args.out, "run", "last.ckpt"), "--out", os.path.join(args.out, "val_render"), "--stage", "val"],
                        capture_output=True, text=True)
    val_line = [l for l in rv.stdout.splitlines() if l.startswith("val:")]
    val_epochs = [l for l in log.stdout.splitlines() if "val/psnr" in l]
    res = {"what": "tools/e2e_synthetic.py: simulated events of an analytic scene -> scripts/train.py -> novel-view PSNR "
                   "after affine log alignment (left/right halves of e2e_novel_views.png: analytic scene / prediction)",
           "batch_size_quantum": args.batch_size_quantum, "mean_psnr_db_after_1_step": float(np.mean(scores0)), "events": n_events, "simulate_s": t_sim, "train_s": t_train, "epochs": args.epochs,
           "steps": args.epochs * args.steps_per_epoch, "novel_view_psnr_db": scores, "mean_psnr_db": float(np.mean(scores)),
           "posed_image_validation (scripts/render.py --stage val)": val_line[0] if val_line else rv.stderr[-500:],
           "validation_epochs (scripts/train.py)": val_epochs,
           "train_log_tail": log.stdout.strip().splitlines()[-6:]}
    json.dump(res, open(os.path.join(args.out, "e2e_result.json"), "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
