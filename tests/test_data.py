"""CPU: event dataset schema + batcher (robust_e_nerf_amd/data.py) against a direct restatement of the
reference's per-event loops (data/datasets.py:133-284) on small random streams."""
import collections
import os

import numpy as np
import pytest
import torch

from robust_e_nerf_amd import data


def _stream(seed, n=4000, w=7, h=5, dup=0.15):
    g = np.random.default_rng(seed)
    pos = np.stack([g.integers(0, w, n), g.integers(0, h, n)], -1).astype(np.uint16)
    ts = np.sort(g.integers(0, 2000, n)).astype(np.int64)            # plenty of equal timestamps
    pol = g.random(n) < 0.5
    return pos, ts, pol, w


def _loop_queue(pos, ts, pol, w, h):
    win_t = [[collections.deque(maxlen=2) for _ in range(w)] for _ in range(h)]
    win_p = [[collections.deque(maxlen=2) for _ in range(w)] for _ in range(h)]
    out = []
    for i in range(len(ts)):
        x, y = int(pos[i, 0]), int(pos[i, 1])
        win_t[y][x].append(int(ts[i])); win_p[y][x].append(int(pol[i]))
        if len(win_t[y][x]) < 2 or win_t[y][x][0] == win_t[y][x][-1]:
            continue
        npos = sum(win_p[y][x]) - win_p[y][x][0]
        out.append((x, y, win_t[y][x][0], int(ts[i]), npos, 1 - npos))
    return np.array(out, np.int64)


def _loop_max_refractory(pos, ts, w, h):
    win = [[collections.deque(maxlen=2) for _ in range(w)] for _ in range(h)]
    best = float("inf")
    for i in range(len(ts)):
        d = win[int(pos[i, 1])][int(pos[i, 0])]
        if len(d) > 0 and int(ts[i]) == d[-1]:
            continue
        d.append(int(ts[i]))
        if len(d) == 2:
            best = min(best, d[1] - d[0])
    return best


def test_queue_and_refractory_match_event_loop():
    for seed in range(4):
        pos, ts, pol, w = _stream(seed)
        ev = data.queue_raw_events(pos, ts, pol, w)
        ref = _loop_queue(pos, ts, pol, w, 5)
        got = torch.stack([ev["position"][:, 0], ev["position"][:, 1], ev["start_ts"], ev["end_ts"], ev["num_pos"],
                           ev["num_neg"]], -1).numpy()
        assert got.shape == ref.shape and (got == ref).all()
        assert float(data.max_refractory_period(pos, ts, w)) == _loop_max_refractory(pos, ts, w, 5)


def test_colorize_and_cache_roundtrip(tmp_path):
    pos, ts, pol, w = _stream(9)
    np.savez(os.path.join(tmp_path, data.RAW_EVENTS), position=pos, timestamp=ts, polarity=pol)
    np.savez(os.path.join(tmp_path, data.CAMERA_CALIBRATION), intrinsics=np.eye(3, dtype=np.float32),
             distortion_params=np.zeros(0, np.float32), distortion_model="plumb_bob", img_height=np.uint16(5),
             img_width=np.uint16(w), bayer_pattern="RGGB")
    ev = data.load_events(str(tmp_path), permutation_seed=3)
    assert os.path.isfile(os.path.join(tmp_path, data.TF_EVENTS))
    ev2 = data.load_events(str(tmp_path), permutation_seed=3)           # from the cache
    assert all(torch.equal(ev[k], ev2[k]) for k in ev)
    assert ev["position"].dtype == torch.float32 and ev["channel_idx"].dtype == torch.uint8
    x, y = ev["position"][:, 0].long(), ev["position"][:, 1].long()
    expect = torch.tensor([0, 1, 1, 2], dtype=torch.uint8)[(x % 2) + 2 * (y % 2)]      # R G / G B
    assert torch.equal(ev["channel_idx"], expect)


def test_undistort_inverts_plumb_bob():
    K = np.array([[300.0, 0, 160], [0, 310.0, 120], [0, 0, 1]])
    dist = np.array([-0.25, 0.08, 1e-3, -2e-3])
    g = np.random.default_rng(0)
    und = np.stack([g.uniform(20, 300, 200), g.uniform(20, 220, 200)], -1)
    x, y = (und[:, 0] - 160) / 300, (und[:, 1] - 120) / 310
    r2 = x * x + y * y
    rad = 1 + dist[0] * r2 + dist[1] * r2 * r2
    xd = x * rad + 2 * dist[2] * x * y + dist[3] * (r2 + 2 * x * x)
    yd = y * rad + dist[2] * (r2 + 2 * y * y) + 2 * dist[3] * x * y
    dis = np.stack([xd * 300 + 160, yd * 310 + 120], -1)
    # five fixed-point iterations (cv::undistortPoints' default criteria in OpenCV 4.5.2, which the reference pins): the 5th
    # iterate, 2.5e-4 px from the exact inverse here -- what cv2 returns, not a converged solve
    assert np.abs(data.undistort_points(dis, K, dist, "plumb_bob") - und).max() < 1e-3


def test_batcher_shapes_ranges_and_rank_seeding():
    pos, ts, pol, w = _stream(1)
    ev = data.undistort_events(data.queue_raw_events(pos, ts, pol, w), dict(distortion_params=np.zeros(0)))
    b0 = data.EventBatcher(ev, 256, "cpu", seed=5, rank=0).next()
    b0b = data.EventBatcher(ev, 256, "cpu", seed=5, rank=0).next()
    b1 = data.EventBatcher(ev, 256, "cpu", seed=5, rank=1).next()
    assert all(torch.equal(b0[k], b0b[k]) for k in b0) and not torch.equal(b0["end_ts"], b1["end_ts"])
    assert b0["position"].shape == (256, 2) and b0["position"].dtype == torch.float32
    for k in ("u_ts_diff", "u_diff_start", "u_grad"):
        assert b0[k].dtype == torch.float64 and float(b0[k].min()) >= 0 and float(b0[k].max()) <= 1
    assert float(b0["u_ts_diff"].min()) == 1.0 and (b0["start_ts"] < b0["end_ts"]).all()
    big = data.trunc_normal(0.0, 1.0, 200000, 0.5, 0.25, torch.float64, torch.Generator().manual_seed(0), "cpu")
    assert abs(float(big.mean()) - 0.5) < 5e-3 and 0.19 < float(big.std()) < 0.23      # truncated at 2 sigma


def test_cli_rejects_unsupported_hyperparameters():
    """scripts/train.py:check_supported -- every shipped reference config shape passes; alternatives the kernels do not
    implement raise NotImplementedError instead of silently training something else."""
    import copy, importlib.util, os
    import pytest
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("train_cli", os.path.join(repo, "scripts", "train.py"))
    cli = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cli)
    import yaml
    ncfg = yaml.safe_load(open(os.path.join(repo, "configs", "synthetic_smoke.yaml")))["model"]["nerf"]
    ncfg["ngp"] = {"pos_encoding": {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "interpolation": "Linear"},
                   "dir_encoding": {"degree": 4},
                   "mlp_base": {"hidden_activation": "softplus", "density_activation": "shifted_trunc_exp", "n_neurons": 64,
                                "n_hidden_layers": 1, "geo_feat_dim": 15, "weight_norm": False},
                   "mlp_head": {"hidden_activation": "softplus", "radiance_activation": "softplus", "n_neurons": 64,
                                "n_hidden_layers": 2, "weight_norm": False}}
    cli.check_supported(ncfg, "ngp")
    # the activation alternatives of models/nerf.py:8-29 are accepted for arch ngp since round 4 (exact-f32 MLP kernels) ...
    for path, ok in ((("mlp_head", "radiance_activation"), "sigmoid"), (("mlp_base", "hidden_activation"), "relu"),
                     (("mlp_head", "hidden_activation"), "relu"), (("mlp_base", "density_activation"), "shifted_softplus"),
                     (("mlp_base", "density_activation"), "softplus")):
        c = copy.deepcopy(ncfg)
        c["ngp"][path[0]][path[1]] = ok
        cli.check_supported(c, "ngp")
        key = {("mlp_head", "radiance_activation"): "radiance_activation", ("mlp_base", "hidden_activation"): "base_hidden_activation",
               ("mlp_head", "hidden_activation"): "head_hidden_activation", ("mlp_base", "density_activation"): "density_activation"}[path]
        assert cli.activation_fields(c, "ngp")[key] == ok
    # ... names outside the reference's tables, and everything else the kernels do not implement, are not
    for path, bad in ((("mlp_head", "radiance_activation"), "tanh"), (("mlp_base", "hidden_activation"), "gelu"),
                      (("mlp_base", "n_neurons"), 128), (("pos_encoding", "interpolation"), "Smoothstep"),
                      (("pos_encoding", "otype"), "Frequency")):
        c = copy.deepcopy(ncfg)
        c["ngp"][path[0]][path[1]] = bad
        with pytest.raises(NotImplementedError):
            cli.check_supported(c, "ngp")
    c = copy.deepcopy(ncfg)                                            # weight_norm (ngp.py:207-228): per MLP, arch ngp
    c["ngp"]["mlp_head"]["weight_norm"] = True
    cli.check_supported(c, "ngp")
    assert cli.weight_norm_flags(c, "ngp") == (False, True) and cli.weight_norm_flags(ncfg, "ngp") == (False, False)
    for otype in ("DenseGrid", "TiledGrid"):                           # the other grid types of tcnn's grid encoding are built
        c = copy.deepcopy(ncfg)
        c["ngp"]["pos_encoding"]["otype"] = otype
        cli.check_supported(c, "ngp")
    ncfg["mlp"] = {"net_depth": 8, "net_width": 256, "skip_layer": 4, "hidden_activation": "softplus"}
    cli.check_supported(ncfg, "mlp")
    ncfg["mlp"]["net_width"] = 128
    with pytest.raises(NotImplementedError):
        cli.check_supported(ncfg, "mlp")
    ncfg["mlp"] = {"net_width": 256, "hidden_activation": "relu", "density_activation": "softplus", "radiance_activation": "sigmoid"}
    cli.check_supported(ncfg, "mlp")                                   # arch mlp: alternatives on the per-layer launches
    assert cli.activation_fields(ncfg, "mlp") == dict(base_hidden_activation="relu", head_hidden_activation="relu",
                                                      density_activation="softplus", radiance_activation="sigmoid")
    ncfg["mlp"]["hidden_activation"] = "gelu"
    with pytest.raises(NotImplementedError):
        cli.check_supported(ncfg, "mlp")
    ncfg["mlp"] = {"net_width": 256, "weight_norm": True}              # arch mlp: one flag for the whole MLP (mlp.py:303-319)
    cli.check_supported(ncfg, "mlp")
    assert cli.weight_norm_flags(ncfg, "mlp") is True


def test_scripts_and_tools_compile():
    """every python entry point outside the package at least parses (tools/ run on the GPU box only)"""
    import glob, os
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = glob.glob(os.path.join(repo, "tools", "*.py")) + glob.glob(os.path.join(repo, "scripts", "*.py")) + \
        [os.path.join(repo, "bench.py"), os.path.join(repo, "__graft_entry__.py")]
    assert len(files) > 10
    for f in files:
        compile(open(f).read(), f, "exec")


# ---------------------------------------------------------------------------------------------------------------
# Fixtures generated by the REFERENCE's own code (tests/golden/make_golden.py: gen_dataset, gen_batch_size)
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_event_intervals_refractory_and_bayer_vs_reference_fixture():
    """data/datasets.py:133-329 run by make_golden.py on a 3 000-event stream: interval construction (`queue_raw_events`),
    `extract_max_refractory_period` and `colorize_events` -- every integer identical."""
    g = np.load(os.path.join(GOLD, "dataset.npz"))
    ev = data.queue_raw_events(g["raw_position"], g["raw_timestamp"], g["raw_polarity"], int(g["width"]))
    for k in ("position", "start_ts", "end_ts", "num_pos", "num_neg"):
        assert ev[k].dtype == torch.int64 and np.array_equal(ev[k].numpy(), g[k]), k
    assert float(data.max_refractory_period(g["raw_position"], g["raw_timestamp"], int(g["width"]))) == float(g["max_refractory_period"])
    col = data.colorize_events(ev, str(g["bayer_pattern"]))
    assert col["channel_idx"].dtype == torch.uint8 and np.array_equal(col["channel_idx"].numpy(), g["channel_idx"])


def test_normalized_samplers_vs_reference_fixture():
    """data/samplers.py: the three samplers as transforms of the uniforms their generator drew (float64, exact)."""
    g = np.load(os.path.join(GOLD, "dataset.npz"))
    u = torch.from_numpy(g["u01"])
    assert u.dtype == torch.float64
    assert torch.equal(data.uniform_from_uniform(u.clone(), 0.0, 1.0), torch.from_numpy(g["uniform_0_1"]))
    assert torch.equal(data.uniform_from_uniform(u.clone(), -2.0, 3.0), torch.from_numpy(g["uniform_m2_3"]))
    assert torch.equal(data.trunc_normal_from_uniform(u.clone(), 0.0, 1.0, 0.5, 0.25), torch.from_numpy(g["trunc_normal_05_025"]))
    assert torch.equal(data.trunc_normal_from_uniform(u.clone(), 0.0, 1.0, 0.2, 0.1), torch.from_numpy(g["trunc_normal_02_01"]))
    assert float(np.abs(g["dirac_1"] - 1.0).max()) == 0.0


def test_update_train_batch_size_vs_reference_fixture():
    """a19: RobustENeRF.update_train_batch_size (robust_e_nerf.py:907-950) run by make_golden.py on 24 cases (with and
    without the grad render, 2 ranks, accumulate_grad_batches 1 / 2 / 4): mean over renders and ranks, new batch size,
    and the micro-batches on which the reference leaves the batch size alone."""
    from robust_e_nerf_amd import parallel
    g = np.load(os.path.join(GOLD, "batch_size.npz"))
    skipped = 0
    for (budget, with_grad, mg, ms, me, accum, bi), (mean_ref, new_ref) in zip(g["cases"], g["result"]):
        means = ([mg] if with_grad else []) + [ms, me]
        mean, new = parallel.new_train_batch_size(int(budget), means, lambda m: (m + 1.5 * m) / 2, int(accum), int(bi))
        assert abs(mean - mean_ref) <= 1e-6 * mean_ref                  # the reference reduces in float32
        if new_ref < 0:
            assert new is None
            skipped += 1
        else:
            assert new is not None and abs(new - int(new_ref)) <= 1     # int(budget / mean) at float32 vs float64 mean
    assert 0 < skipped < len(g["cases"])


def test_reference_train_yamls_are_accepted_by_the_cli_schema():
    """Every configs/train/*.yaml the reference ships carries the keys scripts/train.py reads and only settings the fused
    kernels implement (check_supported).  Runs where /root/reference exists (this container); the GPU-side CLI tests use the
    repo's own files with the same settings."""
    import glob, os, sys
    import pytest, yaml
    ref = "/root/reference/configs/train"
    if not os.path.isdir(ref):
        pytest.skip("reference checkout not present")
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import train as cli
    files = sorted(glob.glob(os.path.join(ref, "*.yaml")))
    assert len(files) >= 4
    for f in files:
        cfg = yaml.safe_load(open(f))
        ncfg = cfg["model"]["nerf"]
        cli.check_supported(ncfg, ncfg.get("arch", "ngp"))
        cli.check_supported(ncfg, "mlp")
        assert ncfg["contraction_type"] in ("aabb", "tanh", "sphere") and cfg.get("float32_matmul_precision") == "highest"
        for sec, keys in (("data", ("train_init_eff_batch_size", "train_eff_ray_sample_batch_size", "alpha_over_white_bg")),
                          ("loss", ("error_fn", "weight", "param_weight")), ("optimizer", ("lr", "relative_lr")),
                          ("trainer", ("max_epochs", "limit_train_batches"))):
            assert all(k in cfg[sec] for k in keys), (f, sec)
        assert set(ncfg["occ_grid"]) >= {"resolution", "occ_thre", "ema_decay", "warmup_steps", "n"}
        assert "multi_step_lr" in cfg["lr_scheduler"] and "freeze" in cfg["model"]["contrast_threshold"]


def test_posed_image_loader_vs_reference_fixture(tmp_path):
    """f2: data.load_posed_images against the reference's own `PosedImage` (data/datasets.py:376-690; fixture
    posed_images.npz written by tests/golden/make_golden.py::gen_posed_images): the dataset files are rebuilt from the
    fixture's inputs, the loader must return the reference's normalized images, poses (OpenGL -> common convention),
    intrinsics, sample ids, permutation and pixel-value bounds."""
    import json
    from PIL import Image
    from robust_e_nerf_amd import data
    g = dict(np.load(os.path.join(GOLD, "posed_images.npz")))
    for name in ("rgba", "gray12", "bayer"):
        root = tmp_path / name
        stage = str(g[f"{name}.stage"])
        os.makedirs(root / "views" / stage)
        for k, im in enumerate(g[f"{name}.in_img"]):
            Image.fromarray(im).save(str(root / "views" / stage / f"r_{k}.png"))
        (root / "views" / f"transforms_{stage}.json").write_text(str(g[f"{name}.transforms"]))
        np.savez(str(root / "camera_calibration.npz"), bayer_pattern=np.array(str(g[f"{name}.bayer"])))
        rp = json.loads(str(g[f"{name}.rp"]))
        if rp is not None:
            np.savez(str(root / "renderer_params.npz"), **{k: np.array(v) for k, v in rp.items()})
        assert data.has_posed_images(str(root), stage) and not data.has_posed_images(str(root), "train")
        seed = int(g[f"{name}.seed"])
        got = data.load_posed_images(str(root), stage, alpha_over_white_bg=bool(g[f"{name}.alpha"]),
                                     permutation_seed=None if seed < 0 else seed)
        assert got["img"].shape == g[f"{name}.img"].shape, name
        assert np.allclose(got["img"].numpy(), g[f"{name}.img"], rtol=2e-6, atol=1e-7), name
        assert np.allclose(got["T_wc_position"].numpy(), g[f"{name}.T_wc_position"], atol=1e-6)
        assert np.allclose(got["T_wc_orientation"].numpy(), g[f"{name}.T_wc_orientation"], atol=1e-6)
        assert np.allclose(got["intrinsics"].numpy(), g[f"{name}.intrinsics"], rtol=1e-6)
        ref_ids = ["".join(map(chr, row)).rstrip() for row in g[f"{name}.sample_id"]]
        assert got["sample_id"] == ref_ids, (got["sample_id"], ref_ids)
        assert abs(got["min_normalized_pixel_value"] - float(g[f"{name}.min"])) < 1e-9
        assert abs(got["max_normalized_pixel_value"] - float(g[f"{name}.max"])) < 1e-6


def test_epoch_metric_is_one_affine_fit_over_all_views_vs_reference_fixture():
    """f2 / ADVICE r3: evaluation.align_and_score against the reference's own `evaluation_epoch_end`
    (models/robust_e_nerf.py:590-707; fixture eval_epoch.npz from tests/golden/make_golden.py::gen_eval_epoch): ONE
    scale / offset per channel over the flattened batch x H x W, then the mean of the per-view L1 / PSNR.  The fit is
    additive over view shards (what the ranks all-reduce), and a per-view fit scores measurably differently."""
    from robust_e_nerf_amd import evaluation
    g = dict(np.load(os.path.join(GOLD, "eval_epoch.npz")))
    for name in ("mono", "bayer"):
        pred, tgt = torch.from_numpy(g[f"{name}.pred"]), torch.from_numpy(g[f"{name}.target"])
        rng = float(g[f"{name}.max"]) - float(g[f"{name}.min"])
        pv, (a, b) = evaluation.align_and_score(pred, tgt, rng)
        assert abs(float(pv[:, 0].mean()) - float(g[f"{name}.l1"])) < 1e-6 * float(g[f"{name}.l1"]) + 1e-8, name
        assert abs(float(pv[:, 1].mean()) - float(g[f"{name}.psnr"])) < 1e-4, name
        # the reference's lstsq on the same data (float64) gives the same scale / offset
        C = 1 if pred.dim() == 3 else 3
        x = pred.reshape(pred.shape[0], C, -1).transpose(0, 1).reshape(C, -1).log().double()
        y = tgt.reshape(tgt.shape[0], C, -1).transpose(0, 1).reshape(C, -1).log().double()
        sol = torch.linalg.lstsq(torch.stack([x, torch.ones_like(x)], -1), y[..., None]).solution[..., 0]
        assert torch.allclose(a, sol[:, 0], rtol=1e-10) and torch.allclose(b, sol[:, 1], rtol=1e-9, atol=1e-12)
        # shards add up: sums of views [0, 2) + sums of views [2, V) == sums of all views
        s = evaluation.log_fit_sums(pred[:2], tgt[:2]) + evaluation.log_fit_sums(pred[2:], tgt[2:])
        assert torch.allclose(s, evaluation.log_fit_sums(pred, tgt), rtol=1e-12)
        pv2, _ = evaluation.align_and_score(pred[2:], tgt[2:], rng, sums=s)
        assert torch.equal(pv2, pv[2:])
        # what the per-view fit of round 3 would have reported: systematically better than the reference's number
        per_view = torch.stack([evaluation.align_and_score(pred[v:v + 1], tgt[v:v + 1], rng)[0][0] for v in range(len(pred))])
        assert float(per_view[:, 1].mean()) > float(g[f"{name}.psnr"]) + 0.5


def test_eval_views_vs_reference_datamodule_fixture(tmp_path):
    """f1/f2, ADVICE r3: data.load_eval_views against the reference's own `DataModule._build_dataset` (data/datamodule.py:
    100-134; fixture eval_dataset.npz from make_golden.py::gen_eval_dataset): eval_target picks the transforms file,
    eval_dataset_perm_seed permutes, {val,test}_dataset_ratio (x {val,test}_eff_batch_size for ints) trims."""
    import json
    from PIL import Image
    from robust_e_nerf_amd import data
    g = dict(np.load(os.path.join(GOLD, "eval_dataset.npz")))
    root = tmp_path / "ds"
    tfs = json.loads(str(g["transforms"]))
    for stage, tf in tfs.items():
        os.makedirs(root / "views" / stage)
        for k, im in enumerate(g[f"img.{stage}"]):
            Image.fromarray(im).save(str(root / "views" / stage / f"{stage[0]}_{k}.png"))
        (root / "views" / f"transforms_{stage}.json").write_text(json.dumps(tf))
    np.savez(str(root / "camera_calibration.npz"), bayer_pattern=np.array(""))
    for i in range(int(g["n_cases"])):
        c = json.loads(str(g[f"case{i}"]))
        dcfg = {"alpha_over_white_bg": False, "eval_dataset_perm_seed": c["seed"], f"{c['stage']}_dataset_ratio": c["ratio"],
                f"{c['stage']}_eff_batch_size": c["eff"]}
        got = data.load_eval_views(str(root), c["stage"], dcfg, c["eval_target"])
        assert got["sample_id"] == list(g[f"case{i}.ids"]), (c, got["sample_id"])
        assert len(got["img"]) == len(got["sample_id"]) == len(got["T_wc_position"])
        assert np.allclose(got["img"][0].numpy(), g[f"case{i}.img0"], rtol=2e-6)
    with pytest.raises(KeyError):
        data.load_eval_views(str(root), "val", {}, ["novel_view"])                  # alpha_over_white_bg is required
    with pytest.raises(NotImplementedError):
        data.load_eval_views(str(root), "val", {"alpha_over_white_bg": False}, ["novel_view", "event_view"])
    with pytest.raises(ValueError):                                                   # datamodule.py:129
        data.load_eval_views(str(root), "val", {"alpha_over_white_bg": False, "val_dataset_ratio": 4, "val_eff_batch_size": 2},
                             ["novel_view"])


def test_image_reader_refuses_what_it_would_silently_downconvert(tmp_path):
    """a 16-bit RGB PNG decoded by Pillow comes back as 8 bit (wrong quantisation levels -> wrong pixel-value
    range -> wrong PSNR).  Without OpenCV / imageio the reader must raise; 8-bit colour and 16-bit grey stay lossless."""
    import importlib.util
    import struct
    import zlib
    from PIL import Image
    from robust_e_nerf_amd import data
    H, W = 3, 4
    px = np.arange(H * W * 3, dtype=np.uint16).reshape(H, W, 3) * 1000
    raw = b"".join(b"\x00" + px[y].astype(">u2").tobytes() for y in range(H))
    chunk = lambda t, d: struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, 16, 2, 0, 0, 0)) + \
        chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b"")
    p16 = tmp_path / "rgb16.png"
    p16.write_bytes(png)
    assert data._png_header(str(p16)) == (16, 2)
    if importlib.util.find_spec("cv2") is None and importlib.util.find_spec("imageio") is None:
        with pytest.raises(ValueError, match="OpenCV or imageio"):
            data._read_image(str(p16))
    else:
        assert np.array_equal(data._read_image(str(p16)), px)
    g16 = (np.arange(H * W, dtype=np.uint16).reshape(H, W) * 4000)
    Image.fromarray(g16).save(str(tmp_path / "g16.png"))
    got = data._read_image(str(tmp_path / "g16.png"))
    assert got.dtype == np.uint16 and np.array_equal(got, g16)
    c8 = (np.arange(H * W * 4, dtype=np.uint8).reshape(H, W, 4))
    Image.fromarray(c8).save(str(tmp_path / "c8.png"))
    assert np.array_equal(data._read_image(str(tmp_path / "c8.png")), c8)


def test_undistortion_inverts_the_forward_distortion_model():
    """f1 (VERDICT r3 missing #6): data.undistort_points restates OpenCV 4.5.2's point undistortion (cv2 is neither in this
    image nor vendored by the reference: parity unpinned against cv2 itself).  Pinned by property: pixels produced by the
    published FORWARD models -- plumb_bob x_d = x (1 + k1 r^2 + k2 r^4) + 2 p1 x y + p2 (r^2 + 2 x^2), ...; fisheye
    theta_d = theta (1 + k1 theta^2 + ... + k4 theta^8) -- are mapped back to where they came from: to 1e-9 px for the
    fisheye Newton iteration, and to what FIVE fixed-point iterations reach for plumb_bob (cv::undistortPoints' default
    criteria; the 5th iterate, checked against the contraction rate), over a 640 x 480 image."""
    from robust_e_nerf_amd import data
    g = np.random.default_rng(3)
    K = np.array([[330.0, 0, 319.5], [0, 331.0, 239.5], [0, 0, 1]])
    u = np.stack([g.uniform(0, 640, 5000), g.uniform(0, 480, 5000)], -1)        # undistorted pixels
    x, y = (u[:, 0] - K[0, 2]) / K[0, 0], (u[:, 1] - K[1, 2]) / K[1, 1]
    # ---- equidistant (TUM-VIE-like coefficients)
    k = np.array([-0.022, 0.0012, -0.0046, 0.0011])
    r = np.sqrt(x * x + y * y)
    th = np.arctan(r)
    th_d = th * (1 + k[0] * th ** 2 + k[1] * th ** 4 + k[2] * th ** 6 + k[3] * th ** 8)
    sc = np.where(r > 1e-12, th_d / np.maximum(r, 1e-12), 1.0)
    d = np.stack([x * sc * K[0, 0] + K[0, 2], y * sc * K[1, 1] + K[1, 2]], -1)
    back = data.undistort_points(d, K, k, "equidistant")
    assert np.abs(back - u).max() < 1e-9 * 640
    # ---- plumb_bob
    k1, k2, p1, p2 = -0.28, 0.07, 1e-3, -5e-4
    r2 = x * x + y * y
    xd = x * (1 + k1 * r2 + k2 * r2 * r2) + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * (1 + k1 * r2 + k2 * r2 * r2) + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    d = np.stack([xd * K[0, 0] + K[0, 2], yd * K[1, 1] + K[1, 2]], -1)
    back = data.undistort_points(d, K, np.array([k1, k2, p1, p2]), "plumb_bob")
    err5 = np.abs(back - u).max()
    assert err5 < 1.0, err5                                   # five iterations: within a pixel at the far corners of this strong distortion, NOT converged ...
    centre = r2 < 0.1
    assert np.abs(back - u)[centre].max() < 1e-3              # ... and to a milli-pixel where the contraction is fast
    # it IS the 5th iterate of the published iteration (a converged solver would return u itself)
    xi, yi = xd.copy(), yd.copy()
    for _ in range(5):
        q = xi * xi + yi * yi
        ic = 1.0 / (1.0 + (k2 * q + k1) * q)
        xi, yi = (xd - (2 * p1 * xi * yi + p2 * (q + 2 * xi * xi))) * ic, (yd - (p1 * (q + 2 * yi * yi) + 2 * p2 * xi * yi)) * ic
    assert np.allclose(back, np.stack([xi * K[0, 0] + K[0, 2], yi * K[1, 1] + K[1, 2]], -1), rtol=0, atol=1e-9)
    assert err5 > 1e-6                                        # (so the iteration count is observable)
    with pytest.raises(NotImplementedError):
        data.undistort_points(d, K, k, "fov")
