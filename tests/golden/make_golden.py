"""Generate golden vectors from the REFERENCE'S OWN PYTHON (build container only).

Usage (in the build container, where /root/reference exists):
    python tests/golden/make_golden.py

The reference package cannot be imported as shipped (pytorch_lightning / nerfacc /
tinycudann / roma / easydict / cv2 / lpips / torchmetrics are absent and CUDA-only), so
small stand-in modules are injected into ``sys.modules``: the
third-party *kernels* are served by this repo's CPU oracle, everything else that runs is
the reference's unmodified code imported from /root/reference:
``NGPradianceField`` (MLPs, SH, contraction, trunc_exp), ``NeRF`` (pixel_params_to_ray,
forward, update_occ_grid), ``render_image`` / ``rendering`` glue, ``LinearTrajectory`` +
``unitquat_slerp``, ``ContrastThreshold``, ``RefractoryPeriod``, ``Loss`` and the real
``RobustENeRF.training_step`` / ``render_pixels``.

Outputs are small ``.npz`` files (inputs, seeds, expected outputs) committed next to this
script.  Nothing from /root/reference is copied: fixtures are data only.  The 50 MB hash
table is regenerated from a seed (``oracle.hashgrid.init_table(kind='mix32')``).
"""
import enum
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
REF = "/root/reference"

from oracle import events as o_events  # noqa: E402
from oracle import field as o_field  # noqa: E402
from oracle import hashgrid, occgrid, render, sampling, trajectory  # noqa: E402

JITTER_LOG = []          # per-ray uniforms consumed by the nerfacc stub, in call order
TABLE_SEED, TABLE_SCALE = 7, 0.5


# --------------------------------------------------------------------------- stubs
class EasyDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        d = {} if d is None else dict(d)
        d.update(kw)
        for k, v in d.items():
            setattr(self, k, v)

    def __setattr__(self, k, v):
        if isinstance(v, (list, tuple)):
            v = type(v)(EasyDict(x) if isinstance(x, dict) and not isinstance(x, EasyDict) else x for x in v)
        elif isinstance(v, dict) and not isinstance(v, EasyDict):
            v = EasyDict(v)
        super().__setattr__(k, v)
        super().__setitem__(k, v)

    __setitem__ = __setattr__

    def pop(self, k, *a):
        if k in self.__dict__:
            super().__delattr__(k)
        return super().pop(k, *a)

    def update(self, e=None, **f):
        d = dict(e or {})
        d.update(f)
        for k, v in d.items():
            setattr(self, k, v)


def install_stubs():
    ed = types.ModuleType("easydict")
    ed.EasyDict = EasyDict
    sys.modules["easydict"] = ed

    # ---- nerfacc ---------------------------------------------------------------
    na = types.ModuleType("nerfacc")

    class ContractionType(enum.Enum):
        AABB = 0
        UN_BOUNDED_TANH = 1
        UN_BOUNDED_SPHERE = 2

    class OccupancyGrid(torch.nn.Module):
        def __init__(self, roi_aabb, resolution=128, contraction_type=ContractionType.AABB):
            super().__init__()
            if isinstance(resolution, int):
                resolution = [resolution] * 3
            self.resolution = list(resolution)
            self.contraction_type = contraction_type
            self.register_buffer("_roi_aabb", torch.tensor(roi_aabb, dtype=torch.float32))
            self.register_buffer("_binary", torch.zeros(self.resolution, dtype=torch.bool))
            self.register_buffer("occs", torch.zeros(int(np.prod(self.resolution))))
            self.updates = []

        def every_n_step(self, step, occ_eval_fn, occ_thre=1e-2, ema_decay=0.95, warmup_steps=256, n=16):
            if not self.training:
                raise RuntimeError("not training")
            if step % n == 0:
                cells = self.occs.numel()
                if step < warmup_steps:
                    idx = torch.arange(cells)
                else:
                    # nerfacc 0.3.x OccupancyGrid._sample_uniform_and_occupied_cells(cells // 4) (restated: SURVEY App. A.1)
                    k = cells // 4
                    uni = torch.randint(cells, (k,))
                    occ_idx = torch.nonzero(self._binary.flatten())[:, 0]
                    if k < len(occ_idx):
                        occ_idx = occ_idx[torch.randint(len(occ_idx), (k,))]
                    idx = torch.cat([uni, occ_idx])
                jit = torch.rand(idx.shape[0], 3).half().float()   # fp16-representable: fixtures store it in 2 bytes
                drawn = []
                orig_randint = torch.randint

                def logging_randint(*a, **kw):          # the reference's occ_eval_fn draws one camera per point (nerf.py:177-181)
                    out = orig_randint(*a, **kw)
                    drawn.append(out.clone())
                    return out
                torch.randint = logging_randint
                try:
                    self.occs, b = occgrid.update(self.occs, tuple(self.resolution), self._roi_aabb,
                                                  self.contraction_type.value, idx, jit, occ_eval_fn,
                                                  occ_thre, ema_decay)
                finally:
                    torch.randint = orig_randint
                self.updates.append(dict(indices=idx, jitter=jit, cam_ids=drawn[0] if drawn else None, step=step))
                self._binary = b

    def ray_marching(rays_o, rays_d, t_min=None, t_max=None, scene_aabb=None, grid=None,
                     sigma_fn=None, alpha_fn=None, near_plane=None, far_plane=None,
                     render_step_size=1e-3, stratified=False, cone_angle=0.0,
                     early_stop_eps=1e-4, alpha_thre=0.0):
        jit = None
        if stratified:
            jit = torch.rand(rays_o.shape[0])
            JITTER_LOG.append(jit.clone())
        ri, ts, te = sampling.ray_marching(
            rays_o.detach(), rays_d.detach(), scene_aabb=scene_aabb,
            grid_binary=None if grid is None else grid._binary,
            grid_roi=None if grid is None else grid._roi_aabb,
            contraction_type=0 if grid is None else grid.contraction_type.value,
            sigma_fn=sigma_fn, near_plane=near_plane, far_plane=far_plane,
            render_step_size=float(render_step_size), stratified=stratified,
            cone_angle=cone_angle, early_stop_eps=early_stop_eps, alpha_thre=alpha_thre,
            jitter=jit)
        return ri, ts, te

    na.ContractionType = ContractionType
    na.OccupancyGrid = OccupancyGrid
    na.ray_marching = ray_marching
    na.render_weight_from_density = lambda t_starts, t_ends, sigmas, ray_indices=None, n_rays=None: \
        render.render_weight_from_density(t_starts, t_ends, sigmas, ray_indices, n_rays)
    na.render_weight_from_alpha = None
    na.accumulate_along_rays = lambda weights, ray_indices, values=None, n_rays=None: \
        render.accumulate_along_rays(weights, ray_indices, values, n_rays)
    sys.modules["nerfacc"] = na

    # ---- tinycudann --------------------------------------------------------------
    tc = types.ModuleType("tinycudann")

    class Encoding(torch.nn.Module):
        def __init__(self, n_input_dims, encoding_config, dtype=torch.float32):
            super().__init__()
            assert encoding_config["otype"] == "HashGrid" and encoding_config["interpolation"] == "Linear"
            self.spec = hashgrid.make_spec(
                encoding_config["n_levels"], encoding_config["n_features_per_level"],
                encoding_config["log2_hashmap_size"], encoding_config["base_resolution"],
                encoding_config["per_level_scale"])
            self.n_output_dims = self.spec.n_output_dims
            self.params = torch.nn.Parameter(
                hashgrid.init_table(self.spec, TABLE_SEED, TABLE_SCALE, "mix32"))

        def forward(self, x):
            return hashgrid.encode(x, self.params, self.spec)

    tc.Encoding = Encoding
    sys.modules["tinycudann"] = tc

    # ---- roma --------------------------------------------------------------------
    ro = types.ModuleType("roma")
    ro.quat_conjugation = trajectory.quat_conjugation
    ro.quat_product = trajectory.quat_product
    ro.rotvec_to_unitquat = trajectory.rotvec_to_unitquat
    ro.unitquat_to_rotmat = trajectory.unitquat_to_rotmat
    internal = types.ModuleType("roma.internal")

    def flatten_batch_dims(t, end_dim):
        batch_shape = t.shape[: end_dim + 1]
        return t.reshape(-1, *t.shape[end_dim + 1:]) if len(batch_shape) > 0 else t.unsqueeze(0), batch_shape

    def unflatten_batch_dims(t, batch_shape):
        return t.reshape(*batch_shape, *t.shape[1:]) if len(batch_shape) > 0 else t.squeeze(0)

    internal.flatten_batch_dims = flatten_batch_dims
    internal.unflatten_batch_dims = unflatten_batch_dims
    ro.internal = internal
    sys.modules["roma"] = ro
    sys.modules["roma.internal"] = internal

    # ---- pytorch_lightning & eval-only deps ------------------------------------------
    pl = types.ModuleType("pytorch_lightning")
    pl.LightningModule = torch.nn.Module
    pl.LightningDataModule = object
    sys.modules["pytorch_lightning"] = pl
    for name in ("cv2", "lpips", "tqdm"):
        if name not in sys.modules:
            try:
                __import__(name)
            except ImportError:
                sys.modules[name] = types.ModuleType(name)
    tm = types.ModuleType("torchmetrics")
    tm.functional = types.ModuleType("torchmetrics.functional")
    sys.modules["torchmetrics"] = tm
    sys.modules["torchmetrics.functional"] = tm.functional
    sys.path.insert(0, REF)


# --------------------------------------------------------------------------- synthetic inputs
def orbit_poses(n_poses=201, radius=4.0, period_ns=1_000_000, seed=0):
    """Camera on a circular orbit looking at the origin; x-right / y-down / z-forward."""
    k = np.arange(n_poses)
    ang = 2 * np.pi * k / (n_poses - 1) * 0.35
    pos = np.stack([radius * np.cos(ang), radius * np.sin(ang), 0.6 * np.sin(3 * ang)], -1)
    fwd = -pos / np.linalg.norm(pos, axis=-1, keepdims=True)
    up = np.array([0.0, 0.0, 1.0])
    right = np.cross(fwd, up)
    right /= np.linalg.norm(right, axis=-1, keepdims=True)
    down = np.cross(fwd, right)
    Rm = np.stack([right, down, fwd], -1)                   # columns = camera axes in world
    from scipy.spatial.transform import Rotation
    quat = Rotation.from_matrix(Rm).as_quat()               # XYZW
    ts = (k * period_ns).astype(np.int64)
    return ts, pos.astype(np.float32), quat.astype(np.float32)


def make_events(B, t_end_ns, seed=1, width=346, height=260):
    g = np.random.default_rng(seed)
    px = np.stack([g.integers(0, width, B), g.integers(0, height, B)], -1).astype(np.float32)
    end = g.integers(20_000_000, t_end_ns, B).astype(np.int64)
    delta = np.exp(g.uniform(np.log(2e5), np.log(2e7), B)).astype(np.int64)
    start = end - delta
    pol = g.random(B) < 0.5
    return px, start, end, pol.astype(np.int64), (~pol).astype(np.int64)


def ball_binary(res, radius=1.0, aabb=(-1.5, -1.5, -1.5, 1.5, 1.5, 1.5)):
    lo, hi = np.array(aabb[:3]), np.array(aabb[3:])
    g = np.stack(np.meshgrid(*[np.arange(res)] * 3, indexing="ij"), -1)
    c = (g + 0.5) / res * (hi - lo) + lo
    return torch.from_numpy(np.linalg.norm(c, axis=-1) < radius)


NGP_CFG = dict(
    pos_encoding=dict(otype="HashGrid", n_levels=16, n_features_per_level=2, log2_hashmap_size=19,
                      base_resolution=16, per_level_scale=1.4472692012786865, interpolation="Linear"),
    dir_encoding=dict(degree=4),
    mlp_base=dict(hidden_activation="softplus", density_activation="shifted_trunc_exp",
                  n_neurons=64, n_hidden_layers=1, geo_feat_dim=15, weight_norm=False),
    mlp_head=dict(hidden_activation="softplus", radiance_activation="softplus",
                  n_neurons=64, n_hidden_layers=2, weight_norm=False),
)


def field_params_np(rf):
    sd = rf.state_dict()
    return {
        "base.w0": sd["mlp_base.1.hidden_layers.0.weight"], "base.b0": sd["mlp_base.1.hidden_layers.0.bias"],
        "base.wo": sd["mlp_base.1.output_layer.weight"], "base.bo": sd["mlp_base.1.output_layer.bias"],
        "head.w0": sd["mlp_head.hidden_layers.0.weight"], "head.b0": sd["mlp_head.hidden_layers.0.bias"],
        "head.w1": sd["mlp_head.hidden_layers.1.weight"], "head.b1": sd["mlp_head.hidden_layers.1.bias"],
        "head.wo": sd["mlp_head.output_layer.weight"], "head.bo": sd["mlp_head.output_layer.bias"],
    }


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


# --------------------------------------------------------------------------- generators
def gen_field(ngp, nerfacc):
    """Reference NGPradianceField forward + first derivatives, three contraction types."""
    torch.manual_seed(10)
    for ct_name, ct, aabb in (("aabb", nerfacc.ContractionType.AABB, [-1.5] * 3 + [1.5] * 3),
                              ("sphere", nerfacc.ContractionType.UN_BOUNDED_SPHERE, [0.5, -2.1, 0.6, 2.0, -0.6, 1.6]),
                              ("tanh", nerfacc.ContractionType.UN_BOUNDED_TANH, [-1.0] * 3 + [1.0] * 3)):
        torch.manual_seed(11)
        act = dict(softplus=torch.nn.Softplus(beta=100))
        base = EasyDict(NGP_CFG["mlp_base"])
        base.hidden_activation = act["softplus"]
        base.density_activation = ngp.shifted_trunc_exp
        head = EasyDict(NGP_CFG["mlp_head"])
        head.hidden_activation = act["softplus"]
        head.radiance_activation = torch.nn.Softplus(beta=1)
        head.output_dim = 1
        rf = ngp.NGPradianceField(aabb=aabb, num_dim=3, use_viewdirs=True, contraction_type=ct,
                                  pos_encoding_config=NGP_CFG["pos_encoding"],
                                  dir_encoding_config=NGP_CFG["dir_encoding"],
                                  mlp_base_config=base, mlp_head_config=head)
        n = 384
        lo, hi = torch.tensor(aabb[:3]), torch.tensor(aabb[3:])
        span = 1.2 if ct_name == "aabb" else 3.0            # some points outside the box
        x = (torch.rand(n, 3) - 0.5) * span * (hi - lo) + (hi + lo) / 2
        d = torch.randn(n, 3)
        d = d / d.norm(dim=-1, keepdim=True)
        rgb, sigma = rf(x, d)
        g_rgb, g_sig = torch.randn_like(rgb), torch.randn_like(sigma)
        rf.zero_grad()
        ((rgb * g_rgb).sum() + (sigma * g_sig).sum()).backward()
        sd = dict(rf.named_parameters())
        grads = {"g." + k: dict(rf.named_parameters())[v].grad for k, v in {
            "base.w0": "mlp_base.1.hidden_layers.0.weight", "base.b0": "mlp_base.1.hidden_layers.0.bias",
            "base.wo": "mlp_base.1.output_layer.weight", "base.bo": "mlp_base.1.output_layer.bias",
            "head.w0": "mlp_head.hidden_layers.0.weight", "head.b0": "mlp_head.hidden_layers.0.bias",
            "head.w1": "mlp_head.hidden_layers.1.weight", "head.b1": "mlp_head.hidden_layers.1.bias",
            "head.wo": "mlp_head.output_layer.weight", "head.bo": "mlp_head.output_layer.bias"}.items()}
        gt = sd["mlp_base.0.params"].grad
        nz = torch.nonzero(gt)[:, 0]
        pick = nz[torch.linspace(0, len(nz) - 1, 256).long()]
        save(f"field_{ct_name}", aabb=np.array(aabb, np.float32), contraction_type=ct.value,
             table_seed=TABLE_SEED, table_scale=TABLE_SCALE, x=x, d=d, rgb=rgb, sigma=sigma,
             g_rgb=g_rgb, g_sigma=g_sig, g_table_sum=gt.double().sum(), g_table_abs=gt.double().abs().sum(),
             g_table_idx=pick, g_table_val=gt[pick], **field_params_np(rf), **grads)


def gen_field_acts(ngp, nerf_mod, nerfacc):
    """Reference NGPradianceField with the YAML's activation ALTERNATIVES, taken from the reference's own tables
    (models/nerf.py:17-29 NeRF.HIDDEN / DENSITY / RADIANCE_ACTIVATION_NAME_TO_FN): forward + first derivatives for two
    combinations that together use every alternative (relu; softplus and shifted_softplus densities; sigmoid)."""
    N = nerf_mod.NeRF
    aabb = [-1.5] * 3 + [1.5] * 3
    combos = {"a": dict(base_hidden="relu", density="softplus", head_hidden="relu", radiance="sigmoid"),
              "b": dict(base_hidden="softplus", density="shifted_softplus", head_hidden="relu", radiance="softplus")}
    out = {"combos": np.array(__import__("json").dumps(combos)), "aabb": np.array(aabb, np.float32),
           "table_seed": np.array(TABLE_SEED), "table_scale": np.array(TABLE_SCALE)}
    for tag, c in combos.items():
        torch.manual_seed(21)
        base = EasyDict(NGP_CFG["mlp_base"])
        base.hidden_activation = N.HIDDEN_ACTIVATION_NAME_TO_FN[c["base_hidden"]]
        base.density_activation = N.DENSITY_ACTIVATION_NAME_TO_FN[c["density"]]
        head = EasyDict(NGP_CFG["mlp_head"])
        head.hidden_activation = N.HIDDEN_ACTIVATION_NAME_TO_FN[c["head_hidden"]]
        head.radiance_activation = N.RADIANCE_ACTIVATION_NAME_TO_FN[c["radiance"]]
        head.output_dim = 1
        rf = ngp.NGPradianceField(aabb=aabb, num_dim=3, use_viewdirs=True, contraction_type=nerfacc.ContractionType.AABB,
                                  pos_encoding_config=NGP_CFG["pos_encoding"], dir_encoding_config=NGP_CFG["dir_encoding"],
                                  mlp_base_config=base, mlp_head_config=head)
        n = 384
        x = (torch.rand(n, 3) - 0.5) * 1.2 * 3.0
        d = torch.randn(n, 3)
        d = d / d.norm(dim=-1, keepdim=True)
        rgb, sigma = rf(x, d)
        g_rgb, g_sig = torch.randn_like(rgb), torch.randn_like(sigma)
        rf.zero_grad()
        ((rgb * g_rgb).sum() + (sigma * g_sig).sum()).backward()
        sd = dict(rf.named_parameters())
        grads = {f"{tag}.g." + k: sd[v].grad for k, v in {
            "base.w0": "mlp_base.1.hidden_layers.0.weight", "base.b0": "mlp_base.1.hidden_layers.0.bias",
            "base.wo": "mlp_base.1.output_layer.weight", "base.bo": "mlp_base.1.output_layer.bias",
            "head.w0": "mlp_head.hidden_layers.0.weight", "head.b0": "mlp_head.hidden_layers.0.bias",
            "head.w1": "mlp_head.hidden_layers.1.weight", "head.b1": "mlp_head.hidden_layers.1.bias",
            "head.wo": "mlp_head.output_layer.weight", "head.bo": "mlp_head.output_layer.bias"}.items()}
        gt = sd["mlp_base.0.params"].grad
        nz = torch.nonzero(gt)[:, 0]
        pick = nz[torch.linspace(0, len(nz) - 1, 256).long()]
        out.update({f"{tag}.x": x, f"{tag}.d": d, f"{tag}.rgb": rgb, f"{tag}.sigma": sigma, f"{tag}.g_rgb": g_rgb, f"{tag}.g_sigma": g_sig,
                    f"{tag}.g_table_abs": gt.double().abs().sum(), f"{tag}.g_table_idx": pick, f"{tag}.g_table_val": gt[pick]})
        out.update({f"{tag}." + k: v for k, v in field_params_np(rf).items()})
        out.update(grads)
        print(f"field_acts {tag}: {c}  rgb [{float(rgb.min()):.3f}, {float(rgb.max()):.3f}]  sigma max {float(sigma.max()):.3f}")
    save("field_acts", **out)


NGP_LINEARS = {"base.w0": "mlp_base.1.hidden_layers.0", "base.wo": "mlp_base.1.output_layer", "head.w0": "mlp_head.hidden_layers.0",
               "head.w1": "mlp_head.hidden_layers.1", "head.wo": "mlp_head.output_layer"}


def gen_field_wn(ngp, nerfacc):
    """Reference NGPradianceField with `weight_norm: true` (external/ngp.py:207-228: torch.nn.utils.weight_norm on every
    Linear of the flagged MLP): forward + gradients w.r.t. weight_g / weight_v / biases / table, for both MLPs flagged and
    for the head alone.  weight_g is scaled away from its initial ||v|| so that the two factors are told apart."""
    aabb = [-1.5] * 3 + [1.5] * 3
    out = {"aabb": np.array(aabb, np.float32), "table_seed": np.array(TABLE_SEED), "table_scale": np.array(TABLE_SCALE)}
    for tag, (wb, wh) in {"both": (True, True), "head": (False, True)}.items():
        torch.manual_seed(31)
        base = EasyDict(NGP_CFG["mlp_base"])
        base.hidden_activation = torch.nn.Softplus(beta=100)
        base.density_activation = ngp.shifted_trunc_exp
        base.weight_norm = wb
        head = EasyDict(NGP_CFG["mlp_head"])
        head.hidden_activation = torch.nn.Softplus(beta=100)
        head.radiance_activation = torch.nn.Softplus(beta=1)
        head.output_dim = 1
        head.weight_norm = wh
        rf = ngp.NGPradianceField(aabb=aabb, num_dim=3, use_viewdirs=True, contraction_type=nerfacc.ContractionType.AABB,
                                  pos_encoding_config=NGP_CFG["pos_encoding"], dir_encoding_config=NGP_CFG["dir_encoding"],
                                  mlp_base_config=base, mlp_head_config=head)
        params = dict(rf.named_parameters())
        with torch.no_grad():
            for k, v in params.items():
                if k.endswith("weight_g"):
                    v.mul_(0.6 + 0.8 * torch.rand_like(v))
        n = 384
        x = (torch.rand(n, 3) - 0.5) * 1.2 * 3.0
        d = torch.randn(n, 3)
        d = d / d.norm(dim=-1, keepdim=True)
        rgb, sigma = rf(x, d)
        g_rgb, g_sig = torch.randn_like(rgb), torch.randn_like(sigma)
        rf.zero_grad()
        ((rgb * g_rgb).sum() + (sigma * g_sig).sum()).backward()
        for k, mod in NGP_LINEARS.items():
            flagged = wb if k.startswith("base") else wh
            names = {k + "_g": mod + ".weight_g", k + "_v": mod + ".weight_v"} if flagged else {k: mod + ".weight"}
            names[k.replace("w", "b")] = mod + ".bias"
            for ours, theirs in names.items():
                out[f"{tag}.{ours}"] = params[theirs].detach().clone()
                out[f"{tag}.g.{ours}"] = params[theirs].grad
        gt = params["mlp_base.0.params"].grad
        nz = torch.nonzero(gt)[:, 0]
        pick = nz[torch.linspace(0, len(nz) - 1, 256).long()]
        out.update({f"{tag}.x": x, f"{tag}.d": d, f"{tag}.rgb": rgb, f"{tag}.sigma": sigma, f"{tag}.g_rgb": g_rgb, f"{tag}.g_sigma": g_sig,
                    f"{tag}.g_table_abs": gt.double().abs().sum(), f"{tag}.g_table_idx": pick, f"{tag}.g_table_val": gt[pick],
                    f"{tag}.flags": np.array([wb, wh])})
        print(f"field_wn {tag}: rgb [{float(rgb.min()):.3f}, {float(rgb.max()):.3f}]  sigma max {float(sigma.max()):.3f}")
    save("field_wn", **out)


def gen_field_mlp(mlp_mod, ngp, nerfacc):
    """Reference VanillaNeRFRadianceField (`arch: mlp`) forward + parameter gradients.  The 593 k parameters
    are regenerated from a seed (oracle.vanilla.init_params) instead of being stored."""
    from oracle import vanilla
    for ct_name, ct, aabb in (("aabb", nerfacc.ContractionType.AABB, [-1.5] * 3 + [1.5] * 3),
                              ("sphere", nerfacc.ContractionType.UN_BOUNDED_SPHERE, [0.5, -2.1, 0.6, 2.0, -0.6, 1.6])):
        torch.manual_seed(12)
        rf = mlp_mod.VanillaNeRFRadianceField(
            aabb=aabb, num_dim=3, contraction_type=ct, radiance_dim=1, hidden_activation=torch.nn.Softplus(beta=100),
            density_activation=ngp.shifted_trunc_exp, radiance_activation=torch.nn.Softplus(beta=1),
            net_depth=8, net_width=256, skip_layer=4, net_depth_condition=1, net_width_condition=128,
            pos_encoder_max_deg=10, view_encoder_max_deg=4, weight_norm=False)
        seed = 21
        params = vanilla.init_params(seed, C=1, gain=1.6)
        sd = rf.state_dict()
        assert set(params) == {k for k in sd if k.startswith("mlp.")}, sorted(set(sd) ^ set(params))
        rf.load_state_dict({**sd, **params})
        n = 160
        lo, hi = torch.tensor(aabb[:3]), torch.tensor(aabb[3:])
        span = 1.2 if ct_name == "aabb" else 3.0
        x = (torch.rand(n, 3) - 0.5) * span * (hi - lo) + (hi + lo) / 2
        d = torch.randn(n, 3)
        d = d / d.norm(dim=-1, keepdim=True)
        rgb, sigma = rf(x, d)
        dens = rf.query_density(x)
        g_rgb, g_sig = torch.randn_like(rgb), torch.randn_like(sigma)
        rf.zero_grad()
        ((rgb * g_rgb).sum() + (sigma * g_sig).sum()).backward()
        grads, gsum = {}, {}
        for k, v in rf.named_parameters():
            g = v.grad.reshape(-1)
            pick = torch.linspace(0, g.numel() - 1, min(64, g.numel())).long()
            grads["gi." + k] = pick
            grads["gv." + k] = g[pick]
            gsum["gs." + k] = g.double().abs().sum()
        save(f"field_mlp_{ct_name}", aabb=np.array(aabb, np.float32), contraction_type=ct.value, param_seed=seed,
             param_gain=1.6, x=x, d=d, rgb=rgb, sigma=sigma, density=dens, g_rgb=g_rgb, g_sigma=g_sig, **grads, **gsum)


def gen_field_mlp_acts(mlp_mod, nerf_mod, nerfacc):
    """Reference VanillaNeRFRadianceField with the YAML's activation ALTERNATIVES taken from the reference's own tables
    (models/nerf.py:17-29): two combinations that together use every alternative; seeded parameters, forward + gradients."""
    from oracle import vanilla
    N = nerf_mod.NeRF
    aabb = [-1.5] * 3 + [1.5] * 3
    combos = {"a": dict(base_hidden="relu", density="softplus", radiance="sigmoid"),
              "b": dict(base_hidden="softplus", density="shifted_softplus", radiance="softplus")}
    out = {"combos": np.array(__import__("json").dumps(combos)), "aabb": np.array(aabb, np.float32), "param_seed": np.array(23),
           "param_gain": np.array(1.6)}
    for tag, c in combos.items():
        torch.manual_seed(14)
        rf = mlp_mod.VanillaNeRFRadianceField(
            aabb=aabb, num_dim=3, contraction_type=nerfacc.ContractionType.AABB, radiance_dim=1,
            hidden_activation=N.HIDDEN_ACTIVATION_NAME_TO_FN[c["base_hidden"]],
            density_activation=N.DENSITY_ACTIVATION_NAME_TO_FN[c["density"]],
            radiance_activation=N.RADIANCE_ACTIVATION_NAME_TO_FN[c["radiance"]], net_depth=8, net_width=256, skip_layer=4,
            net_depth_condition=1, net_width_condition=128, pos_encoder_max_deg=10, view_encoder_max_deg=4, weight_norm=False)
        params = vanilla.init_params(23, C=1, gain=1.6)
        rf.load_state_dict({**rf.state_dict(), **params})
        n = 160
        x = (torch.rand(n, 3) - 0.5) * 1.2 * 3.0
        d = torch.randn(n, 3)
        d = d / d.norm(dim=-1, keepdim=True)
        rgb, sigma = rf(x, d)
        dens = rf.query_density(x)
        g_rgb, g_sig = torch.randn_like(rgb), torch.randn_like(sigma)
        rf.zero_grad()
        ((rgb * g_rgb).sum() + (sigma * g_sig).sum()).backward()
        for k, v in rf.named_parameters():
            g = v.grad.reshape(-1)
            pick = torch.linspace(0, g.numel() - 1, min(64, g.numel())).long()
            out[f"{tag}.gi." + k], out[f"{tag}.gv." + k], out[f"{tag}.gs." + k] = pick, g[pick], g.double().abs().sum()
        out.update({f"{tag}.x": x, f"{tag}.d": d, f"{tag}.rgb": rgb, f"{tag}.sigma": sigma, f"{tag}.density": dens,
                    f"{tag}.g_rgb": g_rgb, f"{tag}.g_sigma": g_sig})
        print(f"field_mlp_acts {tag}: {c}  rgb [{float(rgb.min()):.3f}, {float(rgb.max()):.3f}]  sigma max {float(sigma.max()):.3f}")
    save("field_mlp_acts", **out)


def gen_field_mlp_wn(mlp_mod, ngp, nerfacc):
    """Reference VanillaNeRFRadianceField with weight_norm=True (external/mlp.py:303-319): v from the seeded parameters, g =
    ||v||_row scaled by a stored factor; forward + gradients w.r.t. weight_g (whole) / weight_v, bias (samples + |.| sums)."""
    from oracle import vanilla
    aabb = [-1.5] * 3 + [1.5] * 3
    torch.manual_seed(13)
    rf = mlp_mod.VanillaNeRFRadianceField(
        aabb=aabb, num_dim=3, contraction_type=nerfacc.ContractionType.AABB, radiance_dim=1,
        hidden_activation=torch.nn.Softplus(beta=100), density_activation=ngp.shifted_trunc_exp,
        radiance_activation=torch.nn.Softplus(beta=1), net_depth=8, net_width=256, skip_layer=4, net_depth_condition=1,
        net_width_condition=128, pos_encoder_max_deg=10, view_encoder_max_deg=4, weight_norm=True)
    seed = 22
    params = vanilla.init_params(seed, C=1, gain=1.6)
    sd = rf.state_dict()
    new, g_all = {}, {}
    for k, v in params.items():
        if k.endswith(".weight"):
            g = v.norm(dim=1, keepdim=True) * (0.7 + 0.6 * torch.rand(v.shape[0], 1))
            new[k + "_v"], new[k + "_g"] = v, g
            g_all["wg." + k[: -len(".weight")]] = g
        else:
            new[k] = v
    assert set(new) == {k for k in sd if k.startswith("mlp.")}, sorted(set(sd) ^ set(new))
    rf.load_state_dict({**sd, **new})
    n = 160
    x = (torch.rand(n, 3) - 0.5) * 1.2 * 3.0
    d = torch.randn(n, 3)
    d = d / d.norm(dim=-1, keepdim=True)
    rgb, sigma = rf(x, d)
    g_rgb, g_sig = torch.randn_like(rgb), torch.randn_like(sigma)
    rf.zero_grad()
    ((rgb * g_rgb).sum() + (sigma * g_sig).sum()).backward()
    grads, gsum = {}, {}
    for k, v in rf.named_parameters():
        g = v.grad.reshape(-1)
        pick = torch.arange(g.numel()) if k.endswith("weight_g") else torch.linspace(0, g.numel() - 1, min(64, g.numel())).long()
        grads["gi." + k], grads["gv." + k] = pick, g[pick]
        gsum["gs." + k] = g.double().abs().sum()
    save("field_mlp_wn", aabb=np.array(aabb, np.float32), contraction_type=0, param_seed=seed, param_gain=1.6, x=x, d=d,
         rgb=rgb, sigma=sigma, g_rgb=g_rgb, g_sigma=g_sig, **g_all, **grads, **gsum)


def gen_sh(sh_encoder):
    torch.manual_seed(20)
    d = torch.randn(256, 3)
    d = d / d.norm(dim=-1, keepdim=True)
    save("sh4", d=d, out=sh_encoder.SHEncoder(3, 4)(d))


def gen_rendering(vol_rendering):
    """Reference `rendering` glue over the packed ops (pins the glue + monochrome branch)."""
    torch.manual_seed(30)
    n_rays = 48
    counts = torch.randint(0, 24, (n_rays,))
    counts[5] = 0
    ri = torch.repeat_interleave(torch.arange(n_rays), counts).int()
    n = int(counts.sum())
    ts = torch.rand(n, 1) * 0.01 + torch.cat([torch.arange(c) for c in counts])[:, None] * 0.02 + 2.0
    te = ts + 0.015
    rgb_v, sig_v = torch.rand(n, 1), torch.rand(n, 1) * 30
    bk = torch.tensor([0.7])
    c, o, z = vol_rendering.rendering(ts, te, ri, n_rays, rgb_sigma_fn=lambda a, b, i: (rgb_v, sig_v),
                                      render_bkgd=bk)
    save("rendering", ray_indices=ri, t_starts=ts, t_ends=te, rgb=rgb_v, sigma=sig_v, bkgd=bk,
         colors=c, opacities=o, depths=z, n_rays=n_rays)


def gen_trajectory(trajectories, nerf_mod):
    ts_tab, pos, quat = orbit_poses()
    cam = EasyDict(camera_poses=EasyDict(
        T_wc_position=torch.from_numpy(pos), T_wc_orientation=torch.from_numpy(quat),
        T_wc_timestamp=torch.from_numpy(ts_tab)))
    traj = trajectories.LinearTrajectory(cam)
    g = np.random.default_rng(40)
    ts = g.uniform(0, ts_tab[-1], 300)
    ts[:4] = [0.0, float(ts_tab[-1]), float(ts_tab[7]), float(ts_tab[7]) + 0.5]
    ts = torch.from_numpy(ts).requires_grad_()
    p, R = traj(ts)
    K = torch.tensor([[480.0, 0, 172.5], [0, 480.0, 129.5], [0, 0, 1]])
    px = torch.from_numpy(np.stack([g.integers(0, 346, 300), g.integers(0, 260, 300)], -1).astype(np.float32))
    o, d = nerf_mod.NeRF.pixel_params_to_ray(torch.linalg.inv(K), px, p, R)
    wd = torch.from_numpy(g.standard_normal((300, 3)).astype(np.float32))
    wp = torch.from_numpy(g.standard_normal((300, 3)).astype(np.float32))
    (dts,) = torch.autograd.grad((d * wd).sum() + (o * wp).sum(), ts)
    save("trajectory", tab_ts=ts_tab, tab_pos=pos, tab_quat=quat, ts=ts, p=p, R=R, Kinv=torch.linalg.inv(K),
         px=px, o=o, d=d, wd=wd, wp=wp, dts=dts)
    return ts_tab, pos, quat, K


E_AABB = [0.5, -2.1, 0.6, 2.0, -0.6, 1.6]                  # configs/train/mocap-desk2.yaml:38-39


MLP_CFG = dict(net_depth=8, net_width=256, skip_layer=4, net_depth_condition=1, net_width_condition=128,
               hidden_activation="softplus", density_activation="shifted_trunc_exp", radiance_activation="softplus",
               pos_encoder_max_deg=10, view_encoder_max_deg=4, weight_norm=False)      # configs/train/synthetic.yaml:85-96


def gen_training_step(mods, with_grad_loss: bool, config_e: bool = False, arch_mlp: bool = False):
    """The reference's real RobustENeRF.training_step over the oracle-backed stubs.

    arch_mlp: `arch: mlp` (VanillaNeRFRadianceField, parameters regenerated from a seed as in gen_field_mlp) with l_diff +
    l_grad and C_p, tau trainable: the third-order autograd graph of d l_grad / d tau through the vanilla field.

    config_e: the settings of configs/train/mocap-desk2.yaml (BASELINE configs[4]): sphere contraction (=> scene_aabb=None,
    rays march near -> far, nerf.py:248-251), cone angle 0.004, near / far planes, no background parameter
    (alpha_over_white_bg false => is_valid = opacity > 0), l_grad on, C_p and tau trainable, and an occupancy-grid update
    inside the step (global_step 16) with the cone-angle step sizes of nerf.py:175-193."""
    rmod, nerf_mod, trajectories, egp, loss_mod, nerfacc = mods
    JITTER_LOG.clear()
    torch.manual_seed(50 + int(with_grad_loss) + 7 * int(config_e) + 13 * int(arch_mlp))
    ts_tab, pos, quat = orbit_poses()
    if config_e:                                            # orbit inside the room, around the centre of the AABB
        centre = np.array([1.25, -1.35, 1.1], np.float32)
        pos = (pos * np.array([0.22, 0.22, 0.3], np.float32) + centre).astype(np.float32)
    K = np.array([[480.0, 0, 172.5], [0, 480.0, 129.5], [0, 0, 1]], np.float32)
    tmp = tempfile.mkdtemp()
    np.savez(os.path.join(tmp, "camera_calibration.npz"), intrinsics=K,
             pos_contrast_threshold=np.float32(0.3), neg_contrast_threshold=np.float32(0.25),
             refractory_period=np.float32(0.0 if not with_grad_loss else 3.0e4), bayer_pattern="")
    torch.save(torch.tensor(2.0e5, dtype=torch.float32), os.path.join(tmp, "max_refractory_period.pt"))

    B, occ_res = 96, 32
    m = rmod.RobustENeRF.__new__(rmod.RobustENeRF)
    torch.nn.Module.__init__(m)
    w_grad = 1.0e-3 if with_grad_loss else 0.0
    m.hparams = EasyDict(
        min_modeled_intensity=1e-3,
        loss=dict(weight=dict(log_intensity_grad=w_grad, log_intensity_diff=1.0, nerf_mlp_weight_decay=1e-6),
                  error_fn=dict(log_intensity_grad="mape", log_intensity_diff="mse"),
                  param_weight=dict(log_intensity_grad=None, log_intensity_diff="mean_contrast_reciprocal_sq")),
        contrast_threshold=dict(freeze=not with_grad_loss), refractory_period=dict(freeze=not with_grad_loss))
    m.has_bayer_filter = False
    m.register_buffer("train_intrinsics_inv", torch.linalg.inv(torch.from_numpy(K)), persistent=False)
    m.render_bkgd = None if config_e else "parameter"      # robust_e_nerf.py:154-159
    m.contrast_threshold = egp.ContrastThreshold(tmp)
    m.refractory_period = egp.RefractoryPeriod(tmp)
    occ_cfg = EasyDict(resolution=occ_res, occ_thre=1e-2, ema_decay=0.95, warmup_steps=256, n=16)
    step = 3 ** 0.5 * (1.5 if config_e else 3.0) / 1024      # "auto": robust_e_nerf.py:220-226
    if config_e:
        m.nerf = nerf_mod.NeRF(E_AABB, nerfacc.ContractionType.UN_BOUNDED_SPHERE, occ_cfg, 0.05, 3.0, step,
                               None, 0.004, 1e-4, 0.0, 16384, "ngp", EasyDict(NGP_CFG), 3, 1)
    elif arch_mlp:
        from oracle import vanilla
        m.nerf = nerf_mod.NeRF([-1.5] * 3 + [1.5] * 3, nerfacc.ContractionType.AABB, occ_cfg, None, None, step,
                               "parameter", 0.0, 1e-4, 0.0, 16384, "mlp", EasyDict(MLP_CFG), 3, 1)
        m.nerf.occupancy_grid._binary = ball_binary(occ_res)
        sd = m.nerf.radiance_field.state_dict()
        m.nerf.radiance_field.load_state_dict({**sd, **vanilla.init_params(23, C=1, gain=1.6)})
    else:
        m.nerf = nerf_mod.NeRF([-1.5] * 3 + [1.5] * 3, nerfacc.ContractionType.AABB, occ_cfg, None, None, step,
                               "parameter", 0.0, 1e-4, 0.0, 16384, "ngp", EasyDict(NGP_CFG), 3, 1)
        m.nerf.occupancy_grid._binary = ball_binary(occ_res)
    cam = EasyDict(camera_poses=EasyDict(
        T_wc_position=torch.from_numpy(pos), T_wc_orientation=torch.from_numpy(quat),
        T_wc_timestamp=torch.from_numpy(ts_tab)))
    m.trajectory = trajectories.LinearTrajectory(cam)
    m.loss = loss_mod.Loss(m.hparams.loss.weight, m.hparams.loss.error_fn)
    m.train_ray_sample_batch_size = 1 << 14
    ns = types.SimpleNamespace
    m.trainer = ns(accumulate_grad_batches=1,
                   datamodule=ns(train_dataset=ns(batch_size=B), train_normalized_sampler=ns(datasets=[])))
    m.global_step = 16 if config_e else 1                  # 1: not a multiple of n=16, no grid refresh
    logged = {}
    m.log = lambda k, v, **kw: logged.__setitem__(k, float(v))
    m.all_gather = lambda t: t.unsqueeze(0)
    m.train()

    px, start, end, npos, nneg = make_events(B, int(ts_tab[-1]))
    g = np.random.default_rng(51)
    u1 = np.ones(B) if not with_grad_loss else g.uniform(0.3, 1.0, B)
    u2, u3 = g.uniform(0, 1, B), g.uniform(0, 1, B)
    batch = {"event": {"position": torch.from_numpy(px)[None], "start_ts": torch.from_numpy(start)[None],
                       "end_ts": torch.from_numpy(end)[None], "num_pos": torch.from_numpy(npos)[None],
                       "num_neg": torch.from_numpy(nneg)[None]},
             "normalized": {"ts_diff": torch.from_numpy(u1)[None], "diff_start_ts": torch.from_numpy(u2)[None],
                            "grad_ts": torch.from_numpy(u3)[None]}}
    loss = m.training_step(batch, 0)
    m.zero_grad()
    loss.backward()
    named = dict(m.named_parameters())
    rf = m.nerf.radiance_field
    if arch_mlp:
        grads = {}
        for k, v in rf.named_parameters():
            g = v.grad.reshape(-1)
            pick = torch.linspace(0, g.numel() - 1, min(64, g.numel())).long()
            grads["gi." + k], grads["gv." + k], grads["gs." + k] = pick, g[pick], g.double().abs().sum()
        save("training_step_mlp", param_seed=23, param_gain=1.6, occ_res=occ_res,
             binary=np.packbits(m.nerf.occupancy_grid._binary.numpy().reshape(-1)),
             tab_ts=ts_tab, tab_pos=pos, tab_quat=quat, Kinv=m.train_intrinsics_inv,
             position=px, start_ts=start, end_ts=end, num_pos=npos, num_neg=nneg,
             u_ts_diff=u1, u_diff_start=u2, u_grad=u3, jitters=torch.stack(JITTER_LOG), render_step_size=step,
             p2n_raw=named["contrast_threshold.parametrizations.p2n_contrast_threshold_ratio.original"],
             neg_ct=m.contrast_threshold.neg_contrast_threshold,
             tau_raw=named["refractory_period.parametrizations._refractory_period.original"],
             tau_max=m.refractory_period.max_refractory_period,
             bkgd_raw=named["nerf.parametrizations.render_bkgd.original"], loss=loss, w_grad=w_grad,
             logged_keys=np.array(sorted(logged)), logged_vals=np.array([logged[k] for k in sorted(logged)]),
             g_bkgd_raw=named["nerf.parametrizations.render_bkgd.original"].grad,
             g_p2n_raw=named["contrast_threshold.parametrizations.p2n_contrast_threshold_ratio.original"].grad,
             g_tau_raw=named["refractory_period.parametrizations._refractory_period.original"].grad, **grads)
        return
    gmap = {"base.w0": "mlp_base.1.hidden_layers.0.weight", "base.b0": "mlp_base.1.hidden_layers.0.bias",
            "base.wo": "mlp_base.1.output_layer.weight", "base.bo": "mlp_base.1.output_layer.bias",
            "head.w0": "mlp_head.hidden_layers.0.weight", "head.b0": "mlp_head.hidden_layers.0.bias",
            "head.w1": "mlp_head.hidden_layers.1.weight", "head.b1": "mlp_head.hidden_layers.1.bias",
            "head.wo": "mlp_head.output_layer.weight", "head.bo": "mlp_head.output_layer.bias"}
    rfp = dict(rf.named_parameters())
    grads = {"g." + k: rfp[v].grad for k, v in gmap.items()}
    gt = rfp["mlp_base.0.params"].grad
    nz = torch.nonzero(gt)[:, 0]
    pick = nz[torch.linspace(0, len(nz) - 1, 256).long()]
    extra = {}
    if with_grad_loss:
        extra = dict(
            g_p2n_raw=named["contrast_threshold.parametrizations.p2n_contrast_threshold_ratio.original"].grad,
            g_tau_raw=named["refractory_period.parametrizations._refractory_period.original"].grad)
    if config_e:
        up = m.nerf.occupancy_grid.updates[0]
        assert len(m.nerf.occupancy_grid.updates) == 1 and up["cam_ids"] is not None
        extra.update(aabb=np.array(E_AABB, np.float32), near_plane=0.05, far_plane=3.0, cone_angle=0.004,
                     occ_step=up["step"], occ_jitter=up["jitter"].numpy().astype(np.float16),
                     occ_cam_ids=up["cam_ids"].reshape(-1).numpy().astype(np.uint8),
                     occ_occs_after=m.nerf.occupancy_grid.occs)
    save("training_step_e" if config_e else "training_step_grad" if with_grad_loss else "training_step_diff",
         table_seed=TABLE_SEED, table_scale=TABLE_SCALE, occ_res=occ_res,
         binary=np.packbits(m.nerf.occupancy_grid._binary.numpy().reshape(-1)),
         tab_ts=ts_tab, tab_pos=pos, tab_quat=quat, Kinv=m.train_intrinsics_inv,
         position=px, start_ts=start, end_ts=end, num_pos=npos, num_neg=nneg,
         u_ts_diff=u1, u_diff_start=u2, u_grad=u3,
         jitters=torch.stack(JITTER_LOG), render_step_size=step,
         p2n_raw=named["contrast_threshold.parametrizations.p2n_contrast_threshold_ratio.original"],
         neg_ct=m.contrast_threshold.neg_contrast_threshold,
         tau_raw=named["refractory_period.parametrizations._refractory_period.original"],
         tau_max=m.refractory_period.max_refractory_period,
         bkgd_raw=named.get("nerf.parametrizations.render_bkgd.original", torch.zeros(1)),
         loss=loss, w_grad=w_grad,
         logged_keys=np.array(sorted(logged)), logged_vals=np.array([logged[k] for k in sorted(logged)]),
         g_bkgd_raw=named["nerf.parametrizations.render_bkgd.original"].grad if not config_e else torch.zeros(1),
         g_table_sum=gt.double().sum(), g_table_abs=gt.double().abs().sum(), g_table_idx=pick,
         g_table_val=gt[pick], **field_params_np(rf), **grads, **extra)


def gen_events(egp, loss_mod):
    """ContrastThreshold / RefractoryPeriod / Loss on their own (incl. l1 + mape)."""
    tmp = tempfile.mkdtemp()
    np.savez(os.path.join(tmp, "camera_calibration.npz"),
             pos_contrast_threshold=np.float32(0.31), neg_contrast_threshold=np.float32(0.22),
             refractory_period=np.float32(1.0e4), bayer_pattern="")
    torch.save(torch.tensor(5.0e4, dtype=torch.float32), os.path.join(tmp, "max_refractory_period.pt"))
    ct, rp = egp.ContrastThreshold(tmp), egp.RefractoryPeriod(tmp)
    g = np.random.default_rng(60)
    B = 200
    _, start, end, npos, nneg = make_events(B, 200_000_000, seed=61)
    ev = EasyDict(start_ts=torch.from_numpy(start), end_ts=torch.from_numpy(end),
                  num_pos=torch.from_numpy(npos), num_neg=torch.from_numpy(nneg))
    ev = rp(ct(ev))
    pred_diff = torch.from_numpy(g.standard_normal(B).astype(np.float32)) * 0.3
    pred_grad = torch.from_numpy(g.standard_normal(B)) * 1e-8
    ts_diff = (ev.end_ts - ev.start_ts) * torch.from_numpy(g.uniform(0.2, 1, B))
    valid = torch.from_numpy(g.random(B) < 0.8)
    out = {}
    for fn in ("l1", "mse", "mape"):
        L = loss_mod.Loss(EasyDict(log_intensity_grad=1.0, log_intensity_diff=1.0),
                          EasyDict(log_intensity_grad=fn, log_intensity_diff=fn))
        r = L.compute(EasyDict(ev), EasyDict(log_intensity_grad=pred_grad, is_valid=valid),
                      EasyDict(log_intensity_diff=pred_diff, ts_diff=ts_diff, is_valid=valid))
        out[f"loss_diff_{fn}"], out[f"loss_grad_{fn}"] = r.log_intensity_diff, r.log_intensity_grad
    named = dict(list(ct.named_parameters()) + list(rp.named_parameters()))
    save("events", start_ts=start, end_ts=end, num_pos=npos, num_neg=nneg,
         p2n_raw=named["parametrizations.p2n_contrast_threshold_ratio.original"], neg_ct=ct.neg_contrast_threshold,
         tau_raw=named["parametrizations._refractory_period.original"], tau_max=rp.max_refractory_period,
         c_p=ct.pos_contrast_threshold, mean_c=ct.mean_contrast_threshold, tau=rp.refractory_period,
         ev_log_diff=ev.log_intensity_diff, ev_start_ts=ev.start_ts,
         pred_diff=pred_diff, pred_grad=pred_grad, ts_diff=ts_diff, valid=valid, **out)


def gen_dataset(datasets_mod, samplers_mod):
    """f1: the reference's own `Event` interval construction, maximum refractory period, Bayer colourisation
    (data/datasets.py:133-329) on a small raw stream with repeated timestamps, and its three normalized samplers
    (data/samplers.py) together with the uniforms their generator produced."""
    g = np.random.default_rng(70)
    W, H, N = 24, 18, 3000
    pos = np.stack([g.integers(0, W, N), g.integers(0, H, N)], -1).astype(np.uint16)
    ts = np.sort(g.integers(0, 400, N)).astype(np.int64) * 1000      # many equal timestamps, also at the same pixel
    pol = g.random(N) < 0.5
    tmp = tempfile.mkdtemp()
    np.savez(os.path.join(tmp, "raw_events.npz"), position=pos, timestamp=ts, polarity=pol)
    np.savez(os.path.join(tmp, "camera_calibration.npz"), img_height=np.uint16(H), img_width=np.uint16(W),
             bayer_pattern="RGGB", intrinsics=np.eye(3, dtype=np.float32), distortion_model="",
             distortion_params=np.zeros(4, np.float32))
    Ev = datasets_mod.Event
    calib = Ev.load_camera_calibration(tmp)
    q = Ev.queue_raw_events(tmp, calib)
    tau_max = Ev.extract_max_refractory_period(Ev.load_raw_events(tmp), calib)
    q = Ev.colorize_events(q, calib)
    out = dict(raw_position=pos, raw_timestamp=ts, raw_polarity=pol, width=W, height=H, bayer_pattern="RGGB",
               position=q.position, start_ts=q.start_ts, end_ts=q.end_ts, num_pos=q.num_pos, num_neg=q.num_neg,
               channel_idx=q.channel_idx, max_refractory_period=tau_max)
    # samplers: float64, the generator's uniforms next to what the sampler made of them
    n = 512
    gen = lambda: torch.Generator().manual_seed(71)
    out["u01"] = torch.rand(n, dtype=torch.float64, generator=gen())
    out["uniform_0_1"] = next(iter(samplers_mod.UniformSampler(0.0, 1.0, n, torch.float64, gen())))
    out["uniform_m2_3"] = next(iter(samplers_mod.UniformSampler(-2.0, 3.0, n, torch.float64, gen())))
    out["trunc_normal_05_025"] = next(iter(samplers_mod.TruncatedNormalSampler(0.0, 1.0, n, 0.5, 0.25, torch.float64, gen())))
    out["trunc_normal_02_01"] = next(iter(samplers_mod.TruncatedNormalSampler(0.0, 1.0, n, 0.2, 0.1, torch.float64, gen())))
    out["dirac_1"] = next(iter(samplers_mod.DiracDeltaSampler(1.0, n, torch.float64)))
    save("dataset", **out)


def gen_batch_size(rmod):
    """a19: the reference's RobustENeRF.update_train_batch_size (robust_e_nerf.py:907-950) for a table of cases:
    (ray-sample budget, mean samples per ray of the grad / start / end renders, accumulate_grad_batches, batch index)
    -> returned mean, new per-device batch size (unchanged when the accumulation rule skips the update)."""
    ns = types.SimpleNamespace
    cases, res = [], []
    g = np.random.default_rng(80)
    for k in range(24):
        budget = int(2 ** g.integers(14, 21)) // int(g.choice([1, 2, 8]))
        with_grad = bool(k % 2)
        ms = g.uniform(3.0, 700.0, 3)
        accum = int(g.choice([1, 1, 2, 4]))
        bi = int(g.integers(0, 8))
        m = rmod.RobustENeRF.__new__(rmod.RobustENeRF)
        torch.nn.Module.__init__(m)
        m.train_ray_sample_batch_size = budget
        ds = ns(batch_size=-1)
        samp = [ns(size=-1), ns(size=-1)]
        m.trainer = ns(accumulate_grad_batches=accum, datamodule=ns(train_dataset=ds, train_normalized_sampler=ns(datasets=samp)))
        m.all_gather = lambda t: torch.stack([t, t * 1.5])          # two ranks: the other one saw 1.5x the samples
        bg = EasyDict(mean_num_samples_per_ray=float(ms[0])) if with_grad else None
        bd = EasyDict(start_mean_num_samples_per_ray=float(ms[1]), end_mean_num_samples_per_ray=float(ms[2]))
        mean = m.update_train_batch_size(bg, bd, bi)
        assert samp[0].size == ds.batch_size
        cases.append([budget, float(with_grad), ms[0], ms[1], ms[2], accum, bi])
        res.append([float(mean), ds.batch_size])
    save("batch_size", cases=np.array(cases, np.float64), result=np.array(res, np.float64))


def gen_occgrid_post_warmup(nerf_mod, nerfacc):
    """a21 past warm-up: NeRF.update_occ_grid (nerf.py:170-204) at step 272 > warmup_steps = 256: the reference's
    occ_eval_fn over the cells nerfacc's OccupancyGrid samples then (1/4 uniform + up to 1/4 of the occupied ones;
    policy restated from nerfacc 0.3.x -- parity unpinned)."""
    torch.manual_seed(90)
    # the sample holds duplicate cells (drawn with replacement); the indexed assignment that applies the update keeps an
    # arbitrary candidate for them when it runs multi-threaded: one thread, so that this fixture regenerates bit for bit
    n_threads = torch.get_num_threads()
    torch.set_num_threads(1)
    occ_res = 32
    occ_cfg = EasyDict(resolution=occ_res, occ_thre=1e-2, ema_decay=0.95, warmup_steps=256, n=16)
    step = 3 ** 0.5 * 3.0 / 1024
    nerf = nerf_mod.NeRF([-1.5] * 3 + [1.5] * 3, nerfacc.ContractionType.AABB, occ_cfg, None, None, step,
                         "parameter", 0.0, 1e-4, 0.0, 16384, "ngp", EasyDict(NGP_CFG), 3, 1)
    nerf.train()
    grid = nerf.occupancy_grid
    grid._binary = ball_binary(occ_res, 0.9)
    grid.occs = torch.rand(occ_res ** 3) * 3e-3 * grid._binary.flatten().float()
    occs_before = grid.occs.clone()
    _, pos, _ = orbit_poses()
    nerf.update_occ_grid(272, torch.from_numpy(pos))
    up = grid.updates[0]
    save("occgrid_post_warmup", table_seed=TABLE_SEED, table_scale=TABLE_SCALE, occ_res=occ_res, step=272,
         render_step_size=step, occs_before=occs_before, binary_before=np.packbits(ball_binary(occ_res, 0.9).numpy().reshape(-1)),
         indices=up["indices"].numpy().astype(np.int32), jitter=up["jitter"].numpy().astype(np.float16),
         occs_after=grid.occs, binary_after=np.packbits(grid._binary.numpy().reshape(-1)),
         **field_params_np(nerf.radiance_field))
    torch.set_num_threads(n_threads)


def _install_cv2_stub():
    """cv2 is absent from this image: the three calls `PosedImage` makes (imread IMREAD_UNCHANGED, cvtColor BGR2RGB /
    BGR2GRAY on float32) over PIL / numpy -- OpenCV's documented BGR channel order and its BT.601 grey weights."""
    from PIL import Image
    cv2 = sys.modules["cv2"]
    cv2.IMREAD_UNCHANGED, cv2.COLOR_BGR2RGB, cv2.COLOR_BGR2GRAY = -1, 4, 6

    def imread(path, flags=None):
        im = Image.open(path)
        a = np.asarray(im).astype(np.uint16) if im.mode.startswith("I") else np.asarray(im)
        if a.ndim == 3:
            a = a[..., [2, 1, 0] + ([3] if a.shape[2] == 4 else [])]          # RGB(A) -> BGR(A)
        return np.ascontiguousarray(a)

    def cvtColor(img, code):
        if code == cv2.COLOR_BGR2RGB:
            return np.ascontiguousarray(img[..., ::-1])
        assert code == cv2.COLOR_BGR2GRAY
        return (np.float32(0.114) * img[..., 0] + np.float32(0.587) * img[..., 1] + np.float32(0.299) * img[..., 2]).astype(np.float32)
    cv2.imread, cv2.cvtColor = imread, cvtColor


def gen_posed_images(datasets_mod):
    """f2: the reference's own `PosedImage` (data/datasets.py:376-690) on three tiny datasets written here: quantized
    synthetic BGRA display renders composited over white, a 12-bit grey real capture with explicit intrinsics, and a
    colour (Bayer) sensor.  The fixture keeps the input files' contents and the loader's outputs."""
    import json
    from PIL import Image
    _install_cv2_stub()
    g = np.random.default_rng(77)
    out = {}

    def pose(k):
        a = 0.3 * k
        R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]])
        T = np.eye(4)
        T[:3, :3], T[:3, 3] = R, [4 * np.cos(a), 4 * np.sin(a), 0.5 * k]
        return T.tolist()
    cases = {
        "rgba": dict(img=g.integers(0, 256, (3, 6, 8, 4), dtype=np.uint8), stage="val", alpha=True, bayer="", seed=3,
                     tf=dict(camera_angle_x=0.69), rp=dict(interm_color_space="display", log_eps=1e-3)),
        "gray12": dict(img=g.integers(0, 4096, (2, 5, 7), dtype=np.uint16), stage="test", alpha=False, bayer="", seed=None,
                       tf=dict(intrinsics=[[300.0, 0, 3.2], [0, 301.0, 2.1], [0, 0, 1]], bit_depth=12), rp=None),
        "bayer": dict(img=g.integers(0, 256, (2, 4, 6, 3), dtype=np.uint8), stage="val", alpha=False, bayer="RGGB", seed=None,
                      tf=dict(camera_angle_x=0.8), rp=None),
    }
    for name, c in cases.items():
        root = tempfile.mkdtemp()
        os.makedirs(os.path.join(root, "views", c["stage"]))
        frames = []
        for k, im in enumerate(c["img"]):
            Image.fromarray(im).save(os.path.join(root, "views", c["stage"], f"r_{k}.png"))
            frames.append(dict(file_path=f"./{c['stage']}/r_{k}", transform_matrix=pose(k)))
        tf = dict(c["tf"], frames=frames)
        json.dump(tf, open(os.path.join(root, "views", f"transforms_{c['stage']}.json"), "w"))
        np.savez(os.path.join(root, "camera_calibration.npz"), bayer_pattern=np.array(c["bayer"]))
        if c["rp"] is not None:
            np.savez(os.path.join(root, "renderer_params.npz"), **{k: np.array(v) for k, v in c["rp"].items()})
        ds = datasets_mod.PosedImage(root, c["stage"], c["seed"], alpha_over_white_bg=c["alpha"])
        pi = ds.posed_imgs
        out.update({f"{name}.in_img": c["img"], f"{name}.transforms": np.array(json.dumps(tf)), f"{name}.stage": np.array(c["stage"]),
                    f"{name}.alpha": np.array(c["alpha"]), f"{name}.bayer": np.array(c["bayer"]),
                    f"{name}.seed": np.array(-1 if c["seed"] is None else c["seed"]),
                    f"{name}.rp": np.array(json.dumps(c["rp"])),
                    f"{name}.img": pi.img, f"{name}.T_wc_position": pi.T_wc_position, f"{name}.T_wc_orientation": pi.T_wc_orientation,
                    f"{name}.intrinsics": pi.intrinsics, f"{name}.sample_id": pi.sample_id,
                    f"{name}.min": np.array(ds.min_normalized_pixel_value), f"{name}.max": np.array(ds.max_normalized_pixel_value)})
    save("posed_images", **out)


def gen_eval_epoch(rmod):
    """The reference's own ``RobustENeRF.evaluation_epoch_end`` (models/robust_e_nerf.py:590-707) run on given
    prediction / target images: ONE affine fit in log space per channel over ALL views, then per-view L1 / PSNR
    (loss_metric/metric.py:60-72; torchmetrics' psnr restated as 10 log10(range^2 / mse) per sample) averaged over
    the views.  Monochrome (5 views) and Bayer-sensor (3 views x 3 channels) cases; the views have different gains so
    that a per-view fit would score differently."""
    import torchmetrics.functional as tmf
    from robust_e_nerf.loss_metric import metric as metric_mod

    def psnr(preds, target, data_range, reduction="elementwise_mean", dim=None):
        mse = ((preds - target) ** 2).mean(dim=dim)
        return (10.0 * torch.log10(torch.as_tensor(data_range, dtype=preds.dtype) ** 2 / mse)).mean()
    tmf.psnr = psnr
    tmf.ssim = lambda preds, target, data_range, reduction: torch.full((), float("nan"))
    m = metric_mod.Metric.__new__(metric_mod.Metric)
    torch.nn.Module.__init__(m)
    object.__setattr__(m, "lpips", lambda in0, in1: torch.zeros(1))
    g = torch.Generator().manual_seed(11)
    out = {}
    for name, V, shape, bayer in (("mono", 5, (24, 32), False), ("bayer", 3, (3, 16, 20), True)):
        lo, hi = 0.5 / 256, 1 - 0.5 / 256
        target = (torch.rand((V,) + shape, generator=g) * (hi - lo) + lo).float()
        gain = torch.linspace(0.6, 1.7, V).view((V,) + (1,) * len(shape))
        pred = (target.log() * gain * 0.8 + 0.3 + 0.05 * torch.randn((V,) + shape, generator=g)).exp().float()
        logged = {}
        stub = types.SimpleNamespace(
            all_gather=lambda o: o, unicode_code_pt_tensor_to_str=rmod.RobustENeRF.unicode_code_pt_tensor_to_str,
            trainer=types.SimpleNamespace(log_dir=None, is_global_zero=True), has_bayer_filter=bayer, metric=m,
            device=torch.device("cpu"), current_epoch=0, logger=None, eval_save_pred_intensity_img=False,
            log=lambda k, v, **kw: logged.__setitem__(k, float(v)))
        sid = torch.tensor([[ord(c) for c in f"r_{v}".ljust(16)] for v in range(V)])
        outputs = [dict(sample_id=sid[v:v + 1], pred_intensity_img=pred[v:v + 1], target_intensity_img=target[v:v + 1])
                   for v in range(V)]
        stage = EasyDict(name="val", min_normalized_pixel_value=lo, max_normalized_pixel_value=hi)
        rmod.RobustENeRF.evaluation_epoch_end(stub, outputs, stage)
        out.update({f"{name}.pred": pred, f"{name}.target": target, f"{name}.min": np.array(lo), f"{name}.max": np.array(hi),
                    f"{name}.l1": np.array(logged["val/l1"]), f"{name}.psnr": np.array(logged["val/psnr"])})
        print(f"eval_epoch {name}: l1 {logged['val/l1']:.6f} psnr {logged['val/psnr']:.4f}")
    save("eval_epoch", **out)


def gen_eval_dataset(datasets_mod):
    """The reference's own `DataModule._build_dataset("val" | "test")` (data/datamodule.py:100-134) on a tiny dataset with
    train / val / test transforms: which views, in which order, a validation / test epoch sees for the YAML's
    eval_target, eval_dataset_perm_seed and {val,test}_dataset_ratio x {val,test}_eff_batch_size."""
    import json
    from PIL import Image
    from robust_e_nerf.data import datamodule as dm_mod
    _install_cv2_stub()
    g = np.random.default_rng(5)
    root = tempfile.mkdtemp()
    counts = dict(train=7, val=6, test=5)
    tfs, imgs = {}, {}
    for stage, n in counts.items():
        os.makedirs(os.path.join(root, "views", stage))
        imgs[stage] = g.integers(0, 256, (n, 4, 6), dtype=np.uint8)
        frames = []
        for k in range(n):
            Image.fromarray(imgs[stage][k]).save(os.path.join(root, "views", stage, f"{stage[0]}_{k}.png"))
            T = np.eye(4)
            T[:3, 3] = [k, 0.5 * k, 2.0]
            frames.append(dict(file_path=f"./{stage}/{stage[0]}_{k}", transform_matrix=T.tolist()))
        tfs[stage] = dict(camera_angle_x=0.7, frames=frames)
        json.dump(tfs[stage], open(os.path.join(root, "views", f"transforms_{stage}.json"), "w"))
    np.savez(os.path.join(root, "camera_calibration.npz"), bayer_pattern=np.array(""))
    dm_mod.DataModule.save_hyperparameters = lambda self, *a: setattr(self, "hparams", EasyDict(train_init_eff_batch_size=4))
    cases = [dict(stage="val", eval_target=["novel_view"], seed=2, ratio=1.0, eff=1),
             dict(stage="val", eval_target=["novel_view"], seed=3, ratio=2, eff=2),
             dict(stage="test", eval_target=["novel_view"], seed=None, ratio=0.5, eff=1),
             dict(stage="val", eval_target=["event_view"], seed=3, ratio=0.75, eff=1)]
    out = {"n_cases": np.array(len(cases)), "transforms": np.array(json.dumps(tfs))}
    for stage in counts:
        out[f"img.{stage}"] = imgs[stage]
    for i, c in enumerate(cases):
        dm = dm_mod.DataModule(0, c["eval_target"], 1, None, root, 1.0, c["ratio"] if c["stage"] == "val" else 1.0,
                               c["ratio"] if c["stage"] == "test" else 1.0, None, c["seed"], False, 4, 1024,
                               c["eff"], c["eff"], 0)
        ds = dm._build_dataset(c["stage"])
        ids = ["".join(map(chr, ds[k]["sample_id"])).rstrip() for k in range(len(ds))]
        out[f"case{i}"] = np.array(json.dumps(c))
        out[f"case{i}.ids"] = np.array(ids)
        out[f"case{i}.img0"] = ds[0]["img"]
        print(f"eval_dataset case {i}: {c} -> {ids}")
    save("eval_dataset", **out)


def main():
    assert os.path.isdir(REF), "golden vectors can only be regenerated where /root/reference exists"
    install_stubs()
    import nerfacc
    from robust_e_nerf.external import ngp, sh_encoder, vol_rendering
    from robust_e_nerf.loss_metric import loss as loss_mod
    from robust_e_nerf.models import event_generation_params as egp
    from robust_e_nerf.models import nerf as nerf_mod
    from robust_e_nerf.models import robust_e_nerf as rmod
    from robust_e_nerf.models import trajectories

    if sys.argv[1:] == ["eval_epoch"]:                 # regenerate one fixture only
        return gen_eval_epoch(rmod)
    if sys.argv[1:] == ["field_acts"]:
        return gen_field_acts(ngp, nerf_mod, nerfacc)
    if sys.argv[1:] == ["field_wn"]:
        from robust_e_nerf.external import mlp as mlp_mod
        gen_field_mlp_wn(mlp_mod, ngp, nerfacc)
        return gen_field_wn(ngp, nerfacc)
    if sys.argv[1:] == ["field_mlp_acts"]:
        from robust_e_nerf.external import mlp as mlp_mod
        return gen_field_mlp_acts(mlp_mod, nerf_mod, nerfacc)
    if sys.argv[1:] == ["eval_dataset"]:
        from robust_e_nerf.data import datasets as datasets_mod
        return gen_eval_dataset(datasets_mod)
    gen_sh(sh_encoder)
    gen_rendering(vol_rendering)
    gen_trajectory(trajectories, nerf_mod)
    gen_events(egp, loss_mod)
    gen_field(ngp, nerfacc)
    gen_field_acts(ngp, nerf_mod, nerfacc)
    gen_field_wn(ngp, nerfacc)
    from robust_e_nerf.external import mlp as mlp_mod
    gen_field_mlp(mlp_mod, ngp, nerfacc)
    gen_field_mlp_wn(mlp_mod, ngp, nerfacc)
    gen_field_mlp_acts(mlp_mod, nerf_mod, nerfacc)
    mods = (rmod, nerf_mod, trajectories, egp, loss_mod, nerfacc)
    gen_training_step(mods, with_grad_loss=False)
    gen_training_step(mods, with_grad_loss=True)
    gen_training_step(mods, with_grad_loss=True, config_e=True)
    gen_training_step(mods, with_grad_loss=True, arch_mlp=True)
    from robust_e_nerf.data import datasets as datasets_mod, samplers as samplers_mod
    gen_dataset(datasets_mod, samplers_mod)
    gen_batch_size(rmod)
    gen_occgrid_post_warmup(nerf_mod, nerfacc)
    gen_posed_images(datasets_mod)
    gen_eval_epoch(rmod)
    gen_eval_dataset(datasets_mod)


if __name__ == "__main__":
    main()
