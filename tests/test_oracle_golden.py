"""CPU: the oracle against golden vectors produced by the reference's own Python
(tests/golden/make_golden.py).  This is what PINS the oracle."""
import numpy as np
import pytest
import torch

from conftest import FIELD_KEYS, field_params_from, load_golden, rel_err, t
from oracle import events, field, hashgrid, render, step, trajectory

SPEC = hashgrid.make_spec()


def test_level_table_matches_survey():
    assert SPEC.resolutions == (16, 24, 34, 49, 71, 102, 148, 213, 308, 446, 646, 934, 1352, 1956, 2831, 4096)
    assert SPEC.sizes[:5] == (4096, 13824, 39304, 117656, 357912) and all(s == 524288 for s in SPEC.sizes[5:])
    assert SPEC.hashed == (False,) * 5 + (True,) * 11
    assert SPEC.n_params == 12_599_920


def test_sh4():
    g = load_golden("sh4")
    assert rel_err(field.sh_encode(t(g["d"]), 4), g["out"]) < 1e-6


def test_rendering_glue():
    g = load_golden("rendering")
    c, o, z = render.rendering(t(g["t_starts"]), t(g["t_ends"]), t(g["ray_indices"]), int(g["n_rays"]),
                               lambda a, b, i: (t(g["rgb"]), t(g["sigma"])), render_bkgd=t(g["bkgd"]))
    assert rel_err(c, g["colors"]) < 1e-6 and rel_err(o, g["opacities"]) < 1e-6 and rel_err(z, g["depths"]) < 1e-6


def test_trajectory_and_raygen():
    g = load_golden("trajectory")
    ts = t(g["ts"]).requires_grad_()
    p, R = trajectory.linear_trajectory(ts, t(g["tab_ts"]), t(g["tab_pos"]), t(g["tab_quat"]))
    assert rel_err(p, g["p"]) < 1e-6 and rel_err(R, g["R"]) < 1e-6
    o, d = trajectory.pixel_params_to_ray(t(g["Kinv"]), t(g["px"]), p, R)
    assert rel_err(o, g["o"]) < 1e-6 and rel_err(d, g["d"]) < 1e-6
    (dts,) = torch.autograd.grad((d * t(g["wd"])).sum() + (o * t(g["wp"])).sum(), ts)
    assert dts.dtype == torch.float64 and rel_err(dts, g["dts"]) < 1e-5


def test_event_params_and_loss():
    g = load_golden("events")
    c_p, c_n, mean_c = events.contrast_thresholds(t(g["p2n_raw"]), t(g["neg_ct"]))
    assert rel_err(c_p, g["c_p"]) < 1e-6 and rel_err(mean_c, g["mean_c"]) < 1e-6
    tau_raw = events.clamp_tau_raw(t(g["tau_raw"]), t(g["tau_max"]))
    tau = events.refractory_period(tau_raw, t(g["tau_max"]))
    assert tau.dtype == torch.float64 and rel_err(tau, g["tau"]) < 1e-12
    ev = events.event_log_intensity_diff(t(g["num_pos"]), t(g["num_neg"]), c_p, c_n)
    assert ev.dtype == torch.float32 and rel_err(ev, g["ev_log_diff"]) < 1e-6
    start = t(g["start_ts"]) + tau
    assert rel_err(start, g["ev_start_ts"]) < 1e-15
    valid = t(g["valid"])
    for fn in ("l1", "mse", "mape"):
        _, terms = events.event_loss(
            ev, start, t(g["end_ts"]), pred_log_diff=t(g["pred_diff"]), ts_diff=t(g["ts_diff"]),
            diff_valid=valid, pred_log_grad=t(g["pred_grad"]), grad_valid=valid,
            err_diff=fn, err_grad=fn, w_diff=1.0, w_grad=1.0, pw_diff=None, pw_grad=None, mean_c=mean_c)
        assert rel_err(terms["log_intensity_diff"], g[f"loss_diff_{fn}"]) < 1e-6
        assert rel_err(terms["log_intensity_grad"], g[f"loss_grad_{fn}"]) < 1e-10


@pytest.mark.parametrize("ct", ["aabb", "sphere", "tanh"])
def test_field_forward_backward(ct, full_table_cache):
    g = load_golden(f"field_{ct}")
    table = full_table_cache(g["table_seed"], g["table_scale"]).clone().requires_grad_()
    p = field_params_from(g, table)
    for k in FIELD_KEYS:
        p[k].requires_grad_()
    rgb, sigma = field.field_forward(t(g["x"]), t(g["d"]), p, SPEC, t(g["aabb"]), int(g["contraction_type"]))
    assert rel_err(rgb, g["rgb"]) < 1e-5 and rel_err(sigma, g["sigma"]) < 1e-5
    ((rgb * t(g["g_rgb"])).sum() + (sigma * t(g["g_sigma"])).sum()).backward()
    for k in FIELD_KEYS:
        assert rel_err(p[k].grad, g["g." + k]) < 1e-4, k
    assert rel_err(table.grad[t(g["g_table_idx"])], g["g_table_val"]) < 1e-4
    assert abs(float(table.grad.double().abs().sum()) - float(g["g_table_abs"])) < 1e-4 * float(g["g_table_abs"])


def _run_training_step(g, table, with_grad, config_e=False, params=None, tangent=None):
    p = params if params is not None else field_params_from(g, table)
    for v in p.values():
        v.requires_grad_()
    occ_res = int(g["occ_res"])
    binary = t(np.unpackbits(g["binary"])[: occ_res ** 3].astype(bool)).view(occ_res, occ_res, occ_res)
    cfg = step.SceneCfg(occ_res=(occ_res,) * 3, render_step_size=float(g["render_step_size"]), sampler="occgrid")
    if config_e:                                             # configs/train/mocap-desk2.yaml:38-51
        from oracle import field as ofield
        cfg = step.SceneCfg(aabb=tuple(float(v) for v in g["aabb"]), contraction_type=ofield.UN_BOUNDED_SPHERE,
                            occ_res=(occ_res,) * 3, near_plane=float(g["near_plane"]), far_plane=float(g["far_plane"]),
                            render_step_size=float(g["render_step_size"]), cone_angle=float(g["cone_angle"]),
                            bkgd_is_param=False, sampler="occgrid")
    batch = step.EventBatch(t(g["position"]), t(g["start_ts"]), t(g["end_ts"]), t(g["num_pos"]), t(g["num_neg"]),
                            t(g["u_ts_diff"]), t(g["u_diff_start"]), t(g["u_grad"]))
    bkgd_raw = t(g["bkgd_raw"]).requires_grad_()
    jit = t(g["jitters"])
    extra = {}
    lc = None
    if with_grad:
        extra = dict(p2n_raw=t(g["p2n_raw"]).requires_grad_(), tau_raw=t(g["tau_raw"]).requires_grad_())
        lc = dict(w_grad=float(g["w_grad"]), err_grad="mape", pw_grad=None)
    kw = dict(p2n_raw=t(g["p2n_raw"]), tau_raw=t(g["tau_raw"]))
    kw.update(extra)
    loss, aux = step.training_forward(
        batch, p, SPEC, cfg, Kinv=t(g["Kinv"]), tab_ts=t(g["tab_ts"]), tab_pos=t(g["tab_pos"]),
        tab_quat=t(g["tab_quat"]), neg_ct=t(g["neg_ct"]),
        tau_max=t(g["tau_max"]), bkgd_raw=bkgd_raw, binary=binary,
        jitter_start=jit[-2], jitter_end=jit[-1], jitter_grad=jit[0] if with_grad else None, loss_cfg=lc, tangent=tangent, **kw)
    aux["leaves"] = kw
    return loss, aux, p, bkgd_raw


def test_training_step_diff(full_table_cache):
    """The reference's real training_step (l_diff only) vs the oracle's composition."""
    g = load_golden("training_step_diff")
    table = full_table_cache(g["table_seed"], g["table_scale"]).clone()
    loss, aux, p, bkgd_raw = _run_training_step(g, table, False)
    assert rel_err(loss, g["loss"]) < 1e-5
    logged = dict(zip(g["logged_keys"].tolist(), g["logged_vals"].tolist()))
    mean_s = (aux["n_start"] + aux["n_end"]) / 2 / len(g["start_ts"])
    assert abs(mean_s - logged["train/mean_num_samples_per_ray"]) < 1e-3
    loss.backward()
    for k in FIELD_KEYS:
        assert rel_err(p[k].grad, g["g." + k]) < 2e-4, k
    assert rel_err(bkgd_raw.grad, g["g_bkgd_raw"]) < 1e-4
    assert rel_err(p["hash"].grad[t(g["g_table_idx"])], g["g_table_val"]) < 2e-4


def test_training_step_grad(full_table_cache):
    """The reference's real training_step with l_diff + l_grad (C_p and tau trainable): the second-order
    path (autograd.gradient(..., create_graph=True), robust_e_nerf.py:395-398) through the oracle."""
    g = load_golden("training_step_grad")
    table = full_table_cache(g["table_seed"], g["table_scale"]).clone()
    loss, aux, p, bkgd_raw = _run_training_step(g, table, True)
    assert rel_err(loss, g["loss"]) < 1e-5
    loss.backward()
    for k in FIELD_KEYS:
        assert rel_err(p[k].grad, g["g." + k]) < 5e-4, k
    assert rel_err(bkgd_raw.grad, g["g_bkgd_raw"]) < 1e-4
    assert rel_err(p["hash"].grad[t(g["g_table_idx"])], g["g_table_val"]) < 5e-4
    assert rel_err(aux["leaves"]["p2n_raw"].grad, g["g_p2n_raw"]) < 1e-4
    assert rel_err(aux["leaves"]["tau_raw"].grad, g["g_tau_raw"]) < 1e-3


def test_training_step_grad_forward_mode(full_table_cache):
    """d log I / dt taken in FORWARD mode (step.training_forward(tangent="forward"): what the HIP path computes, and what
    the bf16 emulation of BASELINE configs[2] uses) reproduces the reference's reverse-mode `autograd.gradient` step:
    the same golden loss and gradients, incl. d loss / d tau through the second derivative."""
    from oracle import field as ofield
    g = load_golden("training_step_grad")
    table = full_table_cache(g["table_seed"], g["table_scale"]).clone()
    loss, aux, p, bkgd_raw = _run_training_step(g, table, True, tangent="forward")
    assert rel_err(loss, g["loss"]) < 1e-5
    loss.backward()
    for k in FIELD_KEYS:
        assert rel_err(p[k].grad, g["g." + k]) < 5e-4, k
    assert rel_err(p["hash"].grad[t(g["g_table_idx"])], g["g_table_val"]) < 5e-4
    assert rel_err(aux["leaves"]["p2n_raw"].grad, g["g_p2n_raw"]) < 1e-4
    assert rel_err(aux["leaves"]["tau_raw"].grad, g["g_tau_raw"]) < 1e-3
    # the rounded matrix product of the bf16 emulation: operands rounded in the forward, tangent AND backward products
    a, b = torch.randn(7, 5, dtype=torch.float64, requires_grad=True), torch.randn(5, 3, dtype=torch.float64, requires_grad=True)
    R = lambda x: x.to(torch.bfloat16).to(x.dtype)
    y = ofield._RoundedMatMul.apply(a, b)
    assert torch.equal(y, R(a) @ R(b))
    gy = torch.randn(7, 3, dtype=torch.float64)
    ga, gb = torch.autograd.grad(y, (a, b), gy)
    assert torch.equal(ga, R(gy) @ R(b).T) and torch.equal(gb, R(a).T @ R(gy))
    import torch.autograd.forward_ad as fwAD
    da = torch.randn(7, 5, dtype=torch.float64)
    with fwAD.dual_level():
        yd = fwAD.unpack_dual(ofield._RoundedMatMul.apply(fwAD.make_dual(a.detach(), da), b.detach())).tangent
    assert torch.equal(yd, R(da) @ R(b.detach()))


def test_training_step_arch_mlp():
    """The reference's training_step with `arch: mlp` (VanillaNeRFRadianceField), l_diff + l_grad, C_p and tau trainable,
    through the oracle's restatement of the vanilla field: loss, all 24 parameter gradients, C_p and tau gradients."""
    from oracle import vanilla
    g = load_golden("training_step_mlp")
    params = vanilla.init_params(int(g["param_seed"]), 1, float(g["param_gain"]))
    loss, aux, p, bkgd_raw = _run_training_step(g, None, True, params=params)
    assert rel_err(loss, g["loss"]) < 1e-5
    loss.backward()
    for k, v in p.items():
        got = v.grad.reshape(-1)
        assert rel_err(got[t(g["gi." + k]).long()], g["gv." + k]) < 5e-4, k
        assert abs(float(got.double().abs().sum()) - float(g["gs." + k])) < 5e-4 * float(g["gs." + k]), k
    assert rel_err(bkgd_raw.grad, g["g_bkgd_raw"]) < 1e-4
    assert rel_err(aux["leaves"]["p2n_raw"].grad, g["g_p2n_raw"]) < 1e-4
    assert rel_err(aux["leaves"]["tau_raw"].grad, g["g_tau_raw"]) < 1e-3


def test_training_step_config_e(full_table_cache):
    """The reference's real training_step at the settings of configs/train/mocap-desk2.yaml (BASELINE configs[4]):
    sphere contraction with scene_aabb=None (rays march near -> far), cone angle, no background parameter (is_valid =
    opacity > 0), l_grad, C_p / tau trainable -- and the occupancy refresh inside that step with the cone-angle step
    sizes of nerf.py:175-193 (warm-up policy, one random camera per cell)."""
    from oracle import field as ofield, occgrid
    g = load_golden("training_step_e")
    table = full_table_cache(g["table_seed"], g["table_scale"]).clone()
    # ---- occupancy refresh that ran inside the reference step
    occ_res = int(g["occ_res"])
    p0 = field_params_from(g, table)
    aabb = t(g["aabb"])
    cells = occ_res ** 3
    cam_ids, tab_pos = t(g["occ_cam_ids"]).long(), t(g["tab_pos"])
    step_size = float(g["render_step_size"])

    def occ_eval_fn(x):
        dens = ofield.query_density(x, p0, SPEC, aabb, ofield.UN_BOUNDED_SPHERE)
        return occgrid.occ_eval(x, lambda _: dens, step_size, float(g["cone_angle"]), tab_pos, cam_ids,
                                float(g["near_plane"]), float(g["far_plane"]))
    occs, binary = occgrid.update(torch.zeros(cells), (occ_res,) * 3, aabb, ofield.UN_BOUNDED_SPHERE, torch.arange(cells),
                                  t(g["occ_jitter"]).float(), occ_eval_fn, 1e-2, 0.95)
    assert rel_err(occs, g["occ_occs_after"]) < 1e-5
    gold_bin = np.unpackbits(g["binary"])[:cells].astype(bool)
    assert (binary.reshape(-1).numpy() != gold_bin).mean() < 1e-4
    # ---- the step itself
    loss, aux, p, bkgd_raw = _run_training_step(g, table, True, config_e=True)
    assert rel_err(loss, g["loss"]) < 1e-5
    logged = dict(zip(g["logged_keys"].tolist(), g["logged_vals"].tolist()))
    assert abs(aux["n_start"] + aux["n_end"] + aux["grad"][3] - 3 * 96 * logged["train/mean_num_samples_per_ray"]) < 1.0
    loss.backward()
    for k in FIELD_KEYS:
        assert rel_err(p[k].grad, g["g." + k]) < 5e-4, k
    assert rel_err(p["hash"].grad[t(g["g_table_idx"])], g["g_table_val"]) < 5e-4
    assert rel_err(aux["leaves"]["p2n_raw"].grad, g["g_p2n_raw"]) < 1e-4
    assert rel_err(aux["leaves"]["tau_raw"].grad, g["g_tau_raw"]) < 1e-3


@pytest.mark.parametrize("ct", ["aabb", "sphere"])
def test_vanilla_field_forward_backward(ct):
    """`arch: mlp` oracle (frequency encoding + 8x256 MLP with skip) vs the reference's
    VanillaNeRFRadianceField run on seed-generated parameters."""
    from oracle import vanilla
    g = load_golden(f"field_mlp_{ct}")
    p = {k: v.requires_grad_() for k, v in vanilla.init_params(int(g["param_seed"]), 1, float(g["param_gain"])).items()}
    aabb = t(g["aabb"])
    rgb, sigma = vanilla.forward(p, t(g["x"]), t(g["d"]), aabb, int(g["contraction_type"]))
    assert rel_err(rgb, g["rgb"]) < 2e-6 and rel_err(sigma, g["sigma"]) < 2e-6
    dens = vanilla.forward(p, t(g["x"]), None, aabb, int(g["contraction_type"]), density_only=True)
    assert rel_err(dens, g["density"]) < 2e-6
    ((rgb * t(g["g_rgb"])).sum() + (sigma * t(g["g_sigma"])).sum()).backward()
    for k, v in p.items():
        gr = v.grad.reshape(-1)
        assert rel_err(gr[t(g["gi." + k])], g["gv." + k]) < 1e-5, k
        assert abs(float(gr.double().abs().sum()) - float(g["gs." + k])) < 1e-5 * float(g["gs." + k]) + 1e-12, k


@pytest.mark.parametrize("tag", ["a", "b"])
def test_field_activation_alternatives(tag, full_table_cache):
    """The YAML's activation alternatives (models/nerf.py:8-29: relu hidden layers, softplus / shifted_softplus densities,
    sigmoid radiance) through the oracle vs the reference's own NGPradianceField built from its own activation tables
    (fixture field_acts.npz): forward and every parameter gradient."""
    import json
    g = load_golden("field_acts")
    acts = json.loads(str(g["combos"]))[tag]
    table = full_table_cache(g["table_seed"], g["table_scale"]).clone().requires_grad_()
    p = {k: t(g[f"{tag}.{k}"]).requires_grad_() for k in FIELD_KEYS}
    p["hash"] = table
    rgb, sigma = field.field_forward(t(g[f"{tag}.x"]), t(g[f"{tag}.d"]), p, SPEC, t(g["aabb"]), 0, acts=acts)
    assert rel_err(rgb, g[f"{tag}.rgb"]) < 1e-5 and rel_err(sigma, g[f"{tag}.sigma"]) < 1e-5
    ((rgb * t(g[f"{tag}.g_rgb"])).sum() + (sigma * t(g[f"{tag}.g_sigma"])).sum()).backward()
    for k in FIELD_KEYS:
        assert rel_err(p[k].grad, g[f"{tag}.g.{k}"]) < 1e-4, k
    assert rel_err(table.grad[t(g[f"{tag}.g_table_idx"])], g[f"{tag}.g_table_val"]) < 1e-4
    assert abs(float(table.grad.double().abs().sum()) - float(g[f"{tag}.g_table_abs"])) < 1e-4 * float(g[f"{tag}.g_table_abs"])


@pytest.mark.parametrize("kw", [dict(otype="TiledGrid"), dict(otype="TiledGrid", base_resolution=4),
                                dict(otype="DenseGrid", per_level_scale=1.1), dict(otype="HashGrid")])
def test_grid_types_level_table_and_index(kw):
    """DenseGrid / TiledGrid (configs/train/synthetic.yaml:63): level sizes by tcnn's rule, the host descriptor the kernels take
    (ops.make_grid_desc) agrees with the oracle's level table, and the vectorised encoder agrees with a literal, per-point
    transcription of tcnn's grid_index loop (index += cell_d * stride WHILE stride <= level size)."""
    from robust_e_nerf_amd import ops
    spec = hashgrid.make_spec(**kw)
    grid, n_params = ops.make_grid_desc(**kw)
    assert n_params == spec.n_params
    for l in range(16):
        assert (grid.res[l], grid.size[l], grid.offset[l], bool(grid.hashed[l])) == \
               (spec.resolutions[l], spec.sizes[l], spec.offsets[l], spec.hashed[l])
        assert abs(grid.scale[l] - spec.scales[l]) == 0.0
        full = (spec.resolutions[l] ** 3 + 7) // 8 * 8
        want = {"HashGrid": min(full, 1 << 19), "DenseGrid": full,
                "TiledGrid": min(full, spec.base_resolution ** 3)}[spec.otype]
        assert spec.sizes[l] == want
    if spec.otype != "HashGrid":
        assert not any(spec.hashed)
    table = hashgrid.init_table(spec, 3, 0.5, "mix32")
    gen = torch.Generator().manual_seed(0)
    x = torch.rand(24, 3, generator=gen)
    x[0] = torch.tensor([1.0, 1.0, 1.0])
    got = hashgrid.encode(x, table, spec).numpy()
    tab = table.numpy().reshape(-1, 2)
    M = 0xFFFFFFFF
    for i in range(x.shape[0]):
        for l in range(16):
            res, size = spec.resolutions[l], spec.sizes[l]
            pos = (np.float64(np.float32(spec.scales[l])) * x[i].numpy().astype(np.float64) + 0.5).astype(np.float32)   # fmaf
            cell = np.floor(pos)
            w = (pos - cell).astype(np.float32)
            acc = np.zeros(2, np.float64)
            for c in range(8):
                pg = [(int(cell[d]) + ((c >> d) & 1)) & M for d in range(3)]
                stride, idx = 1, 0
                for d in range(3):
                    if stride > size:
                        break
                    idx = (idx + pg[d] * stride) & M
                    stride = (stride * res) & M
                if spec.otype == "HashGrid" and size < stride:
                    idx = (pg[0] ^ ((pg[1] * 2654435761) & M) ^ ((pg[2] * 805459861) & M)) & M
                wc = np.prod([w[d] if (c >> d) & 1 else 1 - w[d] for d in range(3)])
                acc += wc * tab[spec.offsets[l] + idx % size]
            assert np.allclose(acc, got[i, 2 * l: 2 * l + 2], rtol=1e-5, atol=1e-6), (i, l)


@pytest.mark.parametrize("tag", ["both", "head"])
def test_field_weight_norm(tag, full_table_cache):
    """weight_norm: true (ngp.py:207-228) -- the oracle's reparametrisation helper vs the reference module's own forward and
    its gradients w.r.t. weight_g / weight_v / biases / table (fixture field_wn.npz)."""
    g0 = load_golden("field_wn")
    g = {k[len(tag) + 1:]: v for k, v in g0.items() if k.startswith(tag + ".")}
    table = full_table_cache(g0["table_seed"], g0["table_scale"])
    raw = {k: t(v).clone().requires_grad_() for k, v in g.items()
           if k.split("_")[0] in FIELD_KEYS and not k.startswith("g")}
    raw["hash"] = table.clone().requires_grad_()
    assert sum(k.endswith("_g") for k in raw) == (5 if tag == "both" else 3)
    p = field.weight_norm_params(raw)
    assert set(p) == set(FIELD_KEYS) | {"hash"}
    rgb, sigma = field.field_forward(t(g["x"]), t(g["d"]), p, SPEC, t(g0["aabb"]), 0)
    assert rel_err(rgb, g["rgb"]) < 1e-5 and rel_err(sigma, g["sigma"]) < 1e-5
    ((rgb * t(g["g_rgb"])).sum() + (sigma * t(g["g_sigma"])).sum()).backward()
    for k, v in raw.items():
        if k != "hash":
            assert rel_err(v.grad, g["g." + k]) < 1e-4, k
    idx = t(g["g_table_idx"])
    assert rel_err(raw["hash"].grad[idx], g["g_table_val"]) < 1e-4


def _vanilla_wn_raw(g):
    """seeded v + the fixture's g -> the weight-normalised parameter dict (reference state-dict names)"""
    from oracle import vanilla
    raw = {}
    for k, v in vanilla.init_params(int(g["param_seed"]), 1, float(g["param_gain"])).items():
        if k.endswith(".weight"):
            raw[k + "_v"], raw[k + "_g"] = v, t(g["wg." + k[: -len(".weight")]])
        else:
            raw[k] = v
    return raw


def test_vanilla_field_weight_norm():
    """arch mlp with weight_norm=True (mlp.py:303-319): the oracle's helper vs the reference module (fixture field_mlp_wn.npz)"""
    from oracle import vanilla
    g = load_golden("field_mlp_wn")
    raw = {k: v.clone().requires_grad_() for k, v in _vanilla_wn_raw(g).items()}
    rgb, sigma = vanilla.forward(vanilla.weight_norm_params(raw), t(g["x"]), t(g["d"]), t(g["aabb"]), 0)
    assert rel_err(rgb, g["rgb"]) < 2e-6 and rel_err(sigma, g["sigma"]) < 2e-6
    ((rgb * t(g["g_rgb"])).sum() + (sigma * t(g["g_sigma"])).sum()).backward()
    for k, v in raw.items():
        gr = v.grad.reshape(-1)
        assert rel_err(gr[t(g["gi." + k])], g["gv." + k]) < 2e-5, k
        assert abs(float(gr.double().abs().sum()) - float(g["gs." + k])) < 2e-5 * float(g["gs." + k]) + 1e-12, k


@pytest.mark.parametrize("tag", ["a", "b"])
def test_vanilla_field_activation_alternatives(tag):
    """arch mlp with the YAML's activation alternatives (models/nerf.py:8-29) -- oracle vs the reference's own
    VanillaNeRFRadianceField built from its own activation tables (fixture field_mlp_acts.npz)."""
    import json
    from oracle import vanilla
    g0 = load_golden("field_mlp_acts")
    acts = json.loads(str(g0["combos"]))[tag]
    g = {k[len(tag) + 1:]: v for k, v in g0.items() if k.startswith(tag + ".")}
    p = {k: v.requires_grad_() for k, v in vanilla.init_params(int(g0["param_seed"]), 1, float(g0["param_gain"])).items()}
    aabb = t(g0["aabb"])
    rgb, sigma = vanilla.forward(p, t(g["x"]), t(g["d"]), aabb, 0, acts=acts)
    assert rel_err(rgb, g["rgb"]) < 2e-6 and rel_err(sigma, g["sigma"]) < 2e-6
    assert rel_err(vanilla.forward(p, t(g["x"]), None, aabb, 0, density_only=True, acts=acts), g["density"]) < 2e-6
    ((rgb * t(g["g_rgb"])).sum() + (sigma * t(g["g_sigma"])).sum()).backward()
    for k, v in p.items():
        gr = v.grad.reshape(-1)
        assert rel_err(gr[t(g["gi." + k])], g["gv." + k]) < 1e-5, k
        assert abs(float(gr.double().abs().sum()) - float(g["gs." + k])) < 1e-5 * float(g["gs." + k]) + 1e-12, k


def test_render_given_reproduces_the_pinned_third_render():
    """oracle.step.render_given (the ray-by-ray checker of tests/test_gpu_parity.py: a render and its time derivative on GIVEN
    rays, ray tangents and samples) against the pinned path: on the rays and samples of training_forward(tangent="forward")
    itself it returns that render's intensity and d log I / dt bit for bit, also when the encoder is replaced by its own
    features (field.ENC_OVERRIDE); near_cell_face flags about 1.5 % of the samples of the fixture step at 4 ulp."""
    import torch.autograd.forward_ad as fwAD
    from oracle import step as ostep, hashgrid, trajectory
    g = load_golden("training_step_grad")
    spec = hashgrid.make_spec()
    table = hashgrid.init_table(spec, int(g["table_seed"]), float(g["table_scale"]), "mix32")
    p = field_params_from(g, table)
    occ_res = int(g["occ_res"])
    binary = t(np.unpackbits(g["binary"])[: occ_res ** 3].astype(bool)).view(occ_res, occ_res, occ_res)
    cfg = ostep.SceneCfg(occ_res=(occ_res,) * 3, render_step_size=float(g["render_step_size"]))
    keys = ("position", "start_ts", "end_ts", "num_pos", "num_neg", "u_ts_diff", "u_diff_start", "u_grad")
    ob = ostep.EventBatch(*(t(g[k]) for k in keys))
    jit = t(g["jitters"])
    _, aux = ostep.training_forward(
        ob, p, spec, cfg, Kinv=t(g["Kinv"]), tab_ts=t(g["tab_ts"]), tab_pos=t(g["tab_pos"]), tab_quat=t(g["tab_quat"]),
        p2n_raw=t(g["p2n_raw"]), neg_ct=t(g["neg_ct"]), tau_raw=t(g["tau_raw"]), tau_max=t(g["tau_max"]), bkgd_raw=t(g["bkgd_raw"]),
        binary=binary, jitter_start=jit[1], jitter_end=jit[2], jitter_grad=jit[0],
        loss_cfg=dict(w_grad=float(g["w_grad"]), err_grad="mape", pw_grad=None), tangent="forward")
    ri, ts, te = aux["grad"][5]
    ts_g = aux["ts"]["grad_ts"].detach()
    with fwAD.dual_level():
        pos, R = trajectory.linear_trajectory(fwAD.make_dual(ts_g, torch.ones_like(ts_g)), t(g["tab_ts"]), t(g["tab_pos"]), t(g["tab_quat"]))
        o, d = trajectory.pixel_params_to_ray(t(g["Kinv"]), ob.position, pos, R)
        uo, ud = fwAD.unpack_dual(o), fwAD.unpack_dual(d)
        o0, od, d0, dd = uo.primal.detach(), uo.tangent.detach().float(), ud.primal.detach(), ud.tangent.detach().float()
    bk = torch.nn.functional.softplus(t(g["bkgd_raw"]))
    out = ostep.render_given(o0, d0, od, dd, ri, ts.reshape(-1), te.reshape(-1), p, spec, cfg, bk)
    inten = out["colors"][:, 0] + cfg.min_modeled_intensity
    assert torch.equal(inten, aux["grad"][0].detach())
    assert torch.equal((out["colords"][:, 0] / inten).double(), aux["pred_log_grad"].detach().double())
    again = ostep.render_given(o0, d0, od, dd, ri, ts.reshape(-1), te.reshape(-1), p, spec, cfg, bk, feat=out["enc"], featd=out["encd"])
    assert torch.equal(again["colors"], out["colors"]) and torch.equal(again["colords"], out["colords"])
    near = ostep.near_cell_face(out["xu"], spec, 4.0)
    assert near.shape == (ts.shape[0], 16) and 0.002 < float(near.any(1).float().mean()) < 0.05
