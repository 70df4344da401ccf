"""`weight_norm: true` of mlp_base / mlp_head (external/ngp.py:207-228: torch.nn.utils.weight_norm on every Linear of the
flagged MLP; VERDICT r3 missing #5): the trainable block holds (v, g), the field kernels read W = g v / ||v|| produced by
ren_weight_norm_fwd, gradients are folded back by ren_weight_norm_bwd before the optimiser.  Pinned to the reference module's
own forward and its gradients w.r.t. weight_g / weight_v (fixture field_wn.npz)."""
import math

import numpy as np
import pytest
import torch

from conftest import FIELD_KEYS, load_golden, rel_err, t

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def dev(x):
    return torch.as_tensor(x).to(DEV).contiguous()


@pytest.fixture(scope="module")
def amd():
    from robust_e_nerf_amd import engine, ops, _lib
    _lib.load()
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    return ops, engine


def _raw_params(g):
    return {k: t(v) for k, v in g.items() if k.split("_")[0] in FIELD_KEYS and not k.startswith("g")}


@pytest.mark.parametrize("tag", ["both", "head"])
def test_weight_norm_field_vs_reference_golden(amd, tag, full_table_cache):
    from oracle import field as ofield
    from robust_e_nerf_amd.engine import contract_points
    ops, engine = amd
    g0 = load_golden("field_wn")
    g = {k[len(tag) + 1:]: v for k, v in g0.items() if k.startswith(tag + ".")}
    flags = tuple(bool(v) for v in g["flags"])
    table = full_table_cache(g0["table_seed"], g0["table_scale"])
    raw = _raw_params(g)
    fld = engine.NGPField(DEV, 1, weight_norm=flags)
    assert fld.n_wn_g == (16 + 64 if flags[0] else 0) + 64 + 64 + 1 and fld.n_params == fld.n_table + fld.n_mlp + fld.n_wn_g
    fld.load(dict(raw, hash=table))
    eff = ofield.weight_norm_params(raw)                                  # the effective weights, by the pinned oracle helper
    for k, v in fld.mlp_views().items():
        assert rel_err(v.cpu(), eff[k]) < 1e-6, k
    for k, v in fld.trainable_views().items():
        assert torch.equal(v.cpu(), raw[k]), k
    aabb = [float(v) for v in g0["aabb"]]
    x, d = dev(g["x"]), dev(g["d"])
    n = x.shape[0]
    scene = ops.make_scene_desc(aabb, 0)
    xu = contract_points(x, aabb, 0)
    feat = ops.hashgrid_fwd(fld.grid, fld.table, x_unit=xu, n=n, layout=1)
    rgb, sigma, base = ops.mlp_fwd(fld.mlp, 1, feat, scene, x_world=x, dirs=d, n=n, save_base=True)
    assert rel_err(rgb.cpu(), g["rgb"]) < 1e-4 and rel_err(sigma.cpu()[:, None], g["sigma"]) < 1e-4
    ws = torch.empty(ops.mlp_bwd_workspace_floats(1), device=DEV)
    for rep in range(2):                                                  # twice: fold_grads() leaves a clean accumulator
        dfeat = ops.mlp_bwd(fld.mlp, 1, feat, base, scene, x_world=x, dirs=d, n=n, rgb=rgb, d_rgb=dev(g["g_rgb"]),
                            d_sigma=dev(g["g_sigma"]).reshape(-1).contiguous(), grad_mlp_params=fld.g_mlp, workspace=ws)
        fld.fold_grads()
        assert float(fld.g_mlp.abs().max()) == 0.0
        for k, v in fld.trainable_views(grad=True).items():
            assert rel_err(v.cpu(), g["g." + k]) < 1e-3, (rep, k)
    gt = torch.zeros_like(fld.table)
    ops.hashgrid_bwd(fld.grid, gt, dfeat, x_unit=xu, n=n, layout=1)
    assert rel_err(gt.cpu()[t(g["g_table_idx"])], g["g_table_val"]) < 1e-3


def test_weight_norm_training_steps_vs_oracle(amd, full_table_cache):
    """Whole steps (l_diff + l_grad) on a head-and-base weight-normalised field: gradients w.r.t. (g, v) vs the oracle's
    autograd through its reparametrisation helper, then three optimiser steps -- (g, v) move exactly as torch.optim.Adam
    moves them, the effective block follows, the loss goes down; checkpoint keys weight_g / weight_v round-trip."""
    from oracle import field as ofield, hashgrid, step as ostep
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import train as cli
    ops, engine = amd
    spec = hashgrid.make_spec()
    g = load_golden("training_step_grad")
    table = full_table_cache(g["table_seed"], g["table_scale"])
    gen = torch.Generator().manual_seed(4)
    raw = {}
    for k in FIELD_KEYS:
        w = t(g[k])
        if ".w" in k:
            raw[k + "_v"] = w * (0.5 + torch.rand(w.shape[0], 1, generator=gen))
            raw[k + "_g"] = w.norm(dim=1, keepdim=True) * (0.8 + 0.4 * torch.rand(w.shape[0], 1, generator=gen))
        else:
            raw[k] = w
    occ_res = int(g["occ_res"])
    cfg = engine.RenderCfg(occ_res=(occ_res,) * 3, render_step_size=float(g["render_step_size"]), sampler="occgrid")
    fld = engine.NGPField(DEV, 1, weight_norm=(True, True))
    fld.load(dict(raw, hash=table))
    r = engine.Renderer(fld, cfg)
    r.binary.copy_(dev(np.unpackbits(g["binary"])[: occ_res ** 3].astype(np.uint8)))
    tr = engine.Trainer(r, engine.TrainCfg(), Kinv=t(g["Kinv"]), tab_ts=t(g["tab_ts"]), tab_pos=t(g["tab_pos"]),
                        tab_quat=t(g["tab_quat"]), p2n_raw=t(g["p2n_raw"]), neg_ct=t(g["neg_ct"]),
                        tau_raw=t(g["tau_raw"]), tau_max=t(g["tau_max"]), bkgd_raw=t(g["bkgd_raw"]))
    batch = dict(position=dev(g["position"]), start_ts=dev(g["start_ts"]), end_ts=dev(g["end_ts"]),
                 num_pos=dev(g["num_pos"]), num_neg=dev(g["num_neg"]), u_ts_diff=dev(g["u_ts_diff"]),
                 u_diff_start=dev(g["u_diff_start"]), u_grad=dev(g["u_grad"]))
    w_grad = float(g["w_grad"])
    tr.t.w_grad, tr.t.err_grad, tr.t.pw_grad = w_grad, "mape", None
    jit = t(g["jitters"])

    def hip_step():
        loss_d, _ = tr.forward_backward(batch, dev(jit[1]), dev(jit[2]))
        loss_g, _ = tr.grad_loss_forward_backward(batch, dev(jit[0]))
        return float(loss_d) + float(loss_g)

    loss = hip_step()
    binary = t(np.unpackbits(g["binary"])[: occ_res ** 3].astype(bool)).view(occ_res, occ_res, occ_res)
    ocfg = ostep.SceneCfg(occ_res=(occ_res,) * 3, render_step_size=float(g["render_step_size"]))
    ob = ostep.EventBatch(t(g["position"]), t(g["start_ts"]), t(g["end_ts"]), t(g["num_pos"]), t(g["num_neg"]),
                          t(g["u_ts_diff"]), t(g["u_diff_start"]), t(g["u_grad"]))
    po = {k: v.clone().requires_grad_() for k, v in raw.items()}
    po["hash"] = table.clone().requires_grad_()
    loss_o, _ = ostep.training_forward(
        ob, ofield.weight_norm_params(po), spec, ocfg, Kinv=t(g["Kinv"]), tab_ts=t(g["tab_ts"]), tab_pos=t(g["tab_pos"]),
        tab_quat=t(g["tab_quat"]), p2n_raw=t(g["p2n_raw"]), neg_ct=t(g["neg_ct"]), tau_raw=t(g["tau_raw"]), tau_max=t(g["tau_max"]),
        bkgd_raw=t(g["bkgd_raw"]), binary=binary, jitter_start=jit[1], jitter_end=jit[2], jitter_grad=jit[0],
        loss_cfg=dict(w_grad=w_grad, err_grad="mape", pw_grad=None))
    loss_o.backward()
    assert abs(loss - float(loss_o)) < 1e-4 * abs(float(loss_o)), (loss, float(loss_o))
    fld.fold_grads(zero=False)
    errs = {k: rel_err(v.cpu(), po[k].grad) for k, v in fld.trainable_views(grad=True).items()}
    print("weight_norm whole step, gradients w.r.t. (g, v, b) vs oracle: worst", max(errs, key=errs.get), f"{max(errs.values()):.2e}")
    assert max(errs.values()) < 5e-3, errs
    # optimiser: Adam over [table | v, biases | g] -- the MLP part against torch.optim.Adam given the same gradients
    n0, n1 = fld.n_table, fld.n_params
    p_ref = fld.flat[n0: n1].detach().cpu().clone().requires_grad_()
    opt = torch.optim.Adam([p_ref], lr=0.01, weight_decay=1e-6)
    losses = [loss]
    for it in range(3):
        if it:
            losses.append(hip_step())
        fld.fold_grads(zero=False)                                        # (optimizer_step folds again: same values)
        p_ref.grad = fld.grad[n0: n1].detach().cpu().clone()
        opt.step()
        tr.optimizer_step()
        assert float(fld.grad.abs().max()) == 0.0 and float(fld.g_mlp.abs().max()) == 0.0
        assert rel_err(fld.flat[n0: n1].cpu(), p_ref.detach()) < 1e-5
        eff = ofield.weight_norm_params({k: v.cpu() for k, v in fld.trainable_views().items()})
        for k, v in fld.mlp_views().items():
            assert rel_err(v.cpu(), eff[k]) < 1e-6, k
    assert losses[-1] < losses[0], losses
    # checkpoint keys as the reference module's state_dict has them under weight_norm, and back
    sd = cli.field_state_dict(fld, "ngp", [-1.5] * 3 + [1.5] * 3)
    assert cli.PREFIX + "mlp_head.hidden_layers.1.weight_g" in sd and cli.PREFIX + "mlp_head.hidden_layers.1.weight" not in sd
    assert tuple(sd[cli.PREFIX + "mlp_base.1.output_layer.weight_g"].shape) == (16, 1)
    fld2 = engine.NGPField(DEV, 1, weight_norm=(True, True))
    cli.load_field_state_dict(fld2, "ngp", sd)
    assert torch.equal(fld2.flat, fld.flat) and torch.equal(fld2.mlp, fld.mlp)


@pytest.mark.parametrize("fused", [True, False])
def test_weight_norm_vanilla_field_vs_reference_golden(amd, fused):
    """arch mlp, weight_norm=True (mlp.py:303-319): VanillaField holds (v, biases | g); the fused field (default) and the
    per-layer dense launches read the effective block; gradients folded to (g, v) vs the reference module (field_mlp_wn.npz)."""
    import types
    from oracle import vanilla as ovan
    from test_oracle_golden import _vanilla_wn_raw
    from robust_e_nerf_amd import vanilla
    ops, engine = amd
    g = load_golden("field_mlp_wn")
    raw = _vanilla_wn_raw(g)
    fld = vanilla.VanillaField(DEV, 1, weight_norm=True)
    assert fld.n_wn_g == 8 * 256 + 1 + 256 + 128 + 1 and fld.n_params == fld.n_block + fld.n_wn_g
    fld.load(raw)
    eff = ovan.weight_norm_params(raw)
    for k, v in fld.state_dict().items():
        assert rel_err(v.cpu(), eff[k]) < 1e-6, k
    for k, v in fld.state_dict(trainable=True).items():
        assert torch.equal(v.cpu(), raw[k]), k
    cfg = engine.RenderCfg(aabb=tuple(float(v) for v in g["aabb"]), contraction_type=0)
    r = vanilla.VanillaRenderer(fld, cfg)
    r.fused_field = fused
    x, d = dev(g["x"]), dev(g["d"])
    rgb, sigma, B = r.query(x, d)
    assert rel_err(rgb.cpu(), g["rgb"]) < 2e-5 and rel_err(sigma.cpu(), g["sigma"][:, 0]) < 2e-5
    ctx = dict(buffers=B, pk=types.SimpleNamespace(n=x.shape[0]), rgb=rgb, sigma=sigma)
    r._field_backward(ctx, dev(g["g_rgb"]), dev(g["g_sigma"])[:, 0].contiguous())
    fld.fold_grads()
    assert float(fld.g_eff.abs().max()) == 0.0
    for k, v in fld.state_dict(grad=True, trainable=True).items():
        gr = v.reshape(-1).cpu()
        ref, idx = t(g["gv." + k]), t(g["gi." + k])
        scale = float(g["gs." + k]) / gr.numel() + 1e-30
        err = float((gr[idx] - ref).abs().max())
        assert err < 3e-4 * max(float(ref.abs().max()), scale), (k, err, float(ref.abs().max()))
        assert abs(float(gr.double().abs().sum()) - float(g["gs." + k])) < 3e-4 * float(g["gs." + k]) + 1e-12, k


def test_weight_norm_vanilla_training_steps(amd):
    """Trainer over a weight-normalised VanillaField: (v, b, g) move as torch.optim.Adam moves them given the folded
    gradients, the effective block follows, the loss goes down."""
    from oracle import vanilla as ovan
    from robust_e_nerf_amd import vanilla
    ops, engine = amd
    g = load_golden("training_step_diff")
    occ_res = int(g["occ_res"])
    cfg = engine.RenderCfg(occ_res=(occ_res,) * 3, render_step_size=float(g["render_step_size"]), sampler="occgrid")
    fld = vanilla.VanillaField(DEV, 1, weight_norm=True)
    fld.load(ovan.init_params(5, 1, 1.0))                             # plain weights: v = W, g = ||W|| as weight_norm() starts
    r = vanilla.VanillaRenderer(fld, cfg)
    r.binary.copy_(dev(np.unpackbits(g["binary"])[: occ_res ** 3].astype(np.uint8)))
    tr = engine.Trainer(r, engine.TrainCfg(lr=1e-3), Kinv=t(g["Kinv"]), tab_ts=t(g["tab_ts"]), tab_pos=t(g["tab_pos"]),
                        tab_quat=t(g["tab_quat"]), p2n_raw=t(g["p2n_raw"]), neg_ct=t(g["neg_ct"]),
                        tau_raw=t(g["tau_raw"]), tau_max=t(g["tau_max"]), bkgd_raw=t(g["bkgd_raw"]))
    batch = dict(position=dev(g["position"]), start_ts=dev(g["start_ts"]), end_ts=dev(g["end_ts"]), num_pos=dev(g["num_pos"]),
                 num_neg=dev(g["num_neg"]), u_ts_diff=dev(g["u_ts_diff"]), u_diff_start=dev(g["u_diff_start"]))
    jit = t(g["jitters"])
    n = fld.n_params
    p_ref = fld.flat[:n].detach().cpu().clone().requires_grad_()
    opt = torch.optim.Adam([p_ref], lr=1e-3, weight_decay=1e-6)
    losses = []
    for _ in range(4):
        loss, _ = tr.forward_backward(batch, dev(jit[-2]), dev(jit[-1]))
        losses.append(float(loss))
        fld.fold_grads(zero=False)
        p_ref.grad = fld.grad[:n].detach().cpu().clone()
        opt.step()
        tr.optimizer_step()
        assert float(fld.grad.abs().max()) == 0.0 and float(fld.g_eff.abs().max()) == 0.0
        assert rel_err(fld.flat[:n].cpu(), p_ref.detach()) < 1e-5
        eff = ovan.weight_norm_params({k: v.cpu() for k, v in fld.state_dict(trainable=True).items()})
        for k, v in fld.state_dict().items():
            assert rel_err(v.cpu(), eff[k]) < 1e-6, k
    assert losses[-1] < losses[0], losses


_NAMES = {"base.w0": "mlp_base.1.hidden_layers.0.weight", "base.b0": "mlp_base.1.hidden_layers.0.bias",
          "base.wo": "mlp_base.1.output_layer.weight", "base.bo": "mlp_base.1.output_layer.bias",
          "head.w0": "mlp_head.hidden_layers.0.weight", "head.b0": "mlp_head.hidden_layers.0.bias",
          "head.w1": "mlp_head.hidden_layers.1.weight", "head.b1": "mlp_head.hidden_layers.1.bias",
          "head.wo": "mlp_head.output_layer.weight", "head.bo": "mlp_head.output_layer.bias"}


@pytest.mark.parametrize("fixture,tag", [("field_wn", "both"), ("field_wn", "head"), ("field_acts", "a"), ("field_acts", "b")])
def test_module_seam_alternatives_vs_reference_golden(amd, fixture, tag, full_table_cache):
    """robust_e_nerf_amd.field.NGPradianceField -- the module with the reference's constructor, attribute names and
    state-dict keys (external/ngp.py:109-228) -- built with weight_norm flags / activation alternatives given the way
    models/nerf.py:150-163 passes them (callables): state-dict keys as the reference module's, forward + every parameter
    gradient on the fused path, and the same values on the op-by-op twice-differentiable path."""
    import json
    from robust_e_nerf_amd import field as fld_mod, nerfacc_api
    from oracle import field as ofield
    g0 = load_golden(fixture)
    g = {k[len(tag) + 1:]: v for k, v in g0.items() if k.startswith(tag + ".")}
    table = full_table_cache(g0["table_seed"], g0["table_scale"])
    base_cfg, head_cfg = {}, dict(output_dim=1)
    if fixture == "field_wn":
        base_cfg["weight_norm"], head_cfg["weight_norm"] = bool(g["flags"][0]), bool(g["flags"][1])
    else:
        acts = json.loads(str(g0["combos"]))[tag]
        hid = {"relu": torch.nn.ReLU(), "softplus": torch.nn.Softplus(beta=100)}
        def shifted_softplus(x):
            return torch.nn.functional.softplus(x - 1)
        dens = {"softplus": torch.nn.Softplus(), "shifted_softplus": shifted_softplus, "shifted_trunc_exp": ofield.shifted_trunc_exp}
        rad = {"softplus": torch.nn.Softplus(beta=1), "sigmoid": torch.nn.Sigmoid()}
        base_cfg.update(hidden_activation=hid[acts["base_hidden"]], density_activation=dens[acts["density"]])
        head_cfg.update(hidden_activation=hid[acts["head_hidden"]], radiance_activation=rad[acts["radiance"]])
    rf = fld_mod.NGPradianceField([float(v) for v in g0["aabb"]], contraction_type=nerfacc_api.ContractionType.AABB,
                                  mlp_base_config=base_cfg, mlp_head_config=head_cfg).to(DEV)
    if fixture == "field_acts":
        assert rf.acts == acts
    sd, want_grad = {"mlp_base.0.params": table, "aabb": t(g0["aabb"])}, {}
    for k, name in _NAMES.items():
        if k + "_v" in g:
            sd[name + "_g"], sd[name + "_v"] = t(g[k + "_g"]), t(g[k + "_v"])
            want_grad[name + "_g"], want_grad[name + "_v"] = g["g." + k + "_g"], g["g." + k + "_v"]
        else:
            sd[name] = t(g[k])
            want_grad[name] = g["g." + k]
    assert set(rf.state_dict().keys()) == set(sd.keys()), sorted(set(rf.state_dict()) ^ set(sd))
    rf.load_state_dict(sd)
    x, d = dev(g["x"]), dev(g["d"])
    rgb, sigma = rf(x, d)                                                 # fused path
    assert rel_err(rgb.cpu(), g["rgb"]) < 1e-4 and rel_err(sigma.cpu(), g["sigma"]) < 1e-4
    assert rel_err(rf.query_density(x).cpu(), g["sigma"]) < 1e-4
    ((rgb * dev(g["g_rgb"])).sum() + (sigma * dev(g["g_sigma"])).sum()).backward()
    params = dict(rf.named_parameters())
    for name, ref in want_grad.items():
        assert rel_err(params[name].grad.cpu(), ref) < 1e-3, name
    assert rel_err(params["mlp_base.0.params"].grad.cpu()[t(g["g_table_idx"])], g["g_table_val"]) < 1e-3
    rf.zero_grad()
    xg = x.clone().requires_grad_()                                       # positions that need gradients: op-by-op path
    rgb2, sigma2 = rf(xg, d)
    assert rel_err(rgb2.detach().cpu(), g["rgb"]) < 1e-4 and rel_err(sigma2.detach().cpu(), g["sigma"]) < 1e-4
    ((rgb2 * dev(g["g_rgb"])).sum() + (sigma2 * dev(g["g_sigma"])).sum()).backward()
    for name, ref in want_grad.items():
        assert rel_err(params[name].grad.cpu(), ref) < 1e-3, name


def test_weight_norm_fold_is_linear_in_the_block_gradient(amd):
    """Data parallelism sums the FOLDED gradients of the ranks (Trainer.optimizer_step: fold_grads() before the all-reduce):
    that equals the fold of the summed block gradient because the fold is linear for fixed (v, g) -- checked on two random
    block gradients, for both architectures' layer tables."""
    from robust_e_nerf_amd import vanilla
    ops, engine = amd
    gen = torch.Generator(device=DEV).manual_seed(9)
    for fld, raw, eff_g, folded in (
            (engine.NGPField(DEV, 3, weight_norm=(True, True)), "mlp_raw", "g_mlp", lambda f: f.grad[f.n_table: f.n_params]),
            (vanilla.VanillaField(DEV, 1, weight_norm=True), "raw", "g_eff", lambda f: f.grad[: f.n_params])):
        getattr(fld, raw).copy_(torch.randn(getattr(fld, raw).shape, device=DEV, generator=gen))
        fld.wn_g.copy_(torch.rand(fld.wn_g.shape, device=DEV, generator=gen) + 0.5)
        fld.refresh()
        ge = getattr(fld, eff_g)[: getattr(fld, raw).numel()]               # (the vanilla block is padded to a multiple of 4)
        g1, g2 = (torch.randn(ge.shape, device=DEV, generator=gen) for _ in range(2))
        outs = []
        for gin in (g1, g2, g1 + g2):
            ge.copy_(gin)
            fld.fold_grads()
            assert float(ge.abs().max()) == 0.0
            outs.append(folded(fld).clone())
        assert rel_err(outs[0] + outs[1], outs[2]) < 1e-5
        assert float(outs[2].abs().max()) > 0.0


@pytest.mark.parametrize("arch", ["ngp", "mlp"])
def test_train_cli_with_hyperparameter_alternatives(tmp_path, arch):
    """scripts/train.py -> checkpoint -> scripts/render.py -> resume, on the reference's YAML schema with the alternatives of its
    [H] comments switched on: weight_norm (checkpoint keys weight_g / weight_v), relu hidden layers, a shifted_softplus density,
    a sigmoid radiance, and (arch ngp) a TiledGrid position encoding; gradient accumulation folds the weight-norm gradient once."""
    import math, os, subprocess, sys, yaml
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = yaml.safe_load(open(os.path.join(repo, "configs", "synthetic_smoke.yaml")))
    nerf = cfg["model"]["nerf"]
    nerf["arch"] = arch
    if arch == "ngp":                                                     # (absent keys = the reference's defaults)
        nerf["ngp"] = {"pos_encoding": {"otype": "TiledGrid"},
                       "mlp_base": dict(hidden_activation="relu", density_activation="shifted_softplus", weight_norm=True),
                       "mlp_head": dict(hidden_activation="relu", radiance_activation="sigmoid", weight_norm=True)}
    else:
        nerf["mlp"] = dict(hidden_activation="relu", density_activation="shifted_softplus", radiance_activation="sigmoid",
                           weight_norm=True)
        cfg["data"]["train_eff_ray_sample_batch_size"] = 65536
        cfg["optimizer"]["lr"]["default"] = 5.0e-4
    cfg["trainer"]["max_epochs"], cfg["trainer"]["limit_train_batches"] = 2, 6
    cfg["trainer"]["accumulate_grad_batches"], cfg["trainer"]["log_every_n_steps"] = 2, 1
    path = os.path.join(tmp_path, "cfg.yaml")
    yaml.safe_dump(cfg, open(path, "w"))
    train = [sys.executable, os.path.join(repo, "scripts", "train.py"), "--config", path, "--synthetic", "60000", "--out", str(tmp_path)]
    out = subprocess.run(train, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if "M rays/s" in l]
    assert lines and all(math.isfinite(float(l.split("loss")[1].split()[0])) for l in lines), out.stdout[-1500:]
    ck = torch.load(os.path.join(tmp_path, "last.ckpt"), map_location="cpu", weights_only=False)
    sd = ck["state_dict"]
    assert ck["global_step"] == 6                                         # 12 micro-batches, 2 per optimiser step
    stem = "nerf.radiance_field." + ("mlp_head.hidden_layers.1" if arch == "ngp" else "mlp.base.hidden_layers.3")
    assert stem + ".weight_g" in sd and stem + ".weight_v" in sd and stem + ".weight" not in sd
    g, v = sd[stem + ".weight_g"], sd[stem + ".weight_v"]
    assert tuple(g.shape) == (v.shape[0], 1) and float((g.reshape(-1) - v.norm(dim=1)).abs().max()) > 0   # g moved away from ||v||
    if arch == "ngp":
        assert sd["nerf.radiance_field.mlp_base.0.params"].numel() == 16 * 4096 * 2      # TiledGrid: 16 levels x 16^3 entries
    rd = os.path.join(tmp_path, "renders")
    out_r = subprocess.run([sys.executable, os.path.join(repo, "scripts", "render.py"), "--config", path, "--ckpt",
                            os.path.join(tmp_path, "last.ckpt"), "--synthetic", "--every", "1000", "--out", rd],
                           capture_output=True, text=True, timeout=900)
    assert out_r.returncode == 0, out_r.stderr[-2000:]
    views = np.load(os.path.join(rd, "views.npz"))
    assert np.isfinite(views["intensity"]).all() and views["intensity"].min() > 0
    out2 = subprocess.run(train + ["--resume", os.path.join(tmp_path, "last.ckpt"), "--max-epochs", "3"],
                          capture_output=True, text=True, timeout=900)
    assert out2.returncode == 0, out2.stderr[-2000:]
    assert "resumed" in out2.stdout
    ck2 = torch.load(os.path.join(tmp_path, "last.ckpt"), map_location="cpu", weights_only=False)
    assert ck2["global_step"] == 9 and stem + ".weight_g" in ck2["state_dict"]
