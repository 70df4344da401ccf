"""Seam conformance against the REFERENCE'S OWN PYTHON (build container only, CPU, no GPU needed).

    python tests/golden/check_seam.py

The drop-in claim of this repository is that the reference's modules run unchanged with
``nerfacc -> robust_e_nerf_amd.nerfacc_api`` and ``tinycudann -> robust_e_nerf_amd.tcnn_api``.
This script imports the reference from /root/reference with exactly that mapping (every other absent
third-party package is stubbed as in make_golden.py) and checks, per reference train YAML:

1. the reference's ``NeRF(...)`` (models/nerf.py:31-142, built as ``RobustENeRF._build_nerf`` does,
   models/robust_e_nerf.py:207-259) CONSTRUCTS over the seam modules, for ``arch: ngp`` and ``arch: mlp``;
2. every keyword the reference passes at its nerfacc / tinycudann call sites (parsed from the reference's sources
   with ``ast``) is accepted by the seam function of the same name;
3. its state-dict keys and shapes equal the ones ``scripts/train.py`` writes into a checkpoint, and a checkpoint
   state dict written by ``scripts/train.py``'s own code loads with ``strict=True`` into the reference module;
4. the seam ``OccupancyGrid`` carries nerfacc 0.3.1's persistent buffers (``_roi_aabb``, ``_binary``,
   ``resolution``, ``occs``).

Nothing of the reference travels: this file contains no reference source, it only imports it where it lies.
"""
import ast
import inspect
import os
import sys

import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(REPO, "scripts"))
REF = "/root/reference"


def install():
    import make_golden
    make_golden.install_stubs()                       # easydict, roma, pytorch_lightning, ... (+ oracle-backed nerfacc / tcnn)
    from robust_e_nerf_amd import nerfacc_api, tcnn_api
    sys.modules["nerfacc"] = nerfacc_api              # the seam under test replaces the oracle-backed stand-ins
    sys.modules["tinycudann"] = tcnn_api
    return make_golden.EasyDict, nerfacc_api, tcnn_api


def call_site_keywords(func_names):
    """{function name: set of keyword names the reference passes}, from every .py under robust_e_nerf/"""
    found = {n: set() for n in func_names}
    for root, _, files in os.walk(os.path.join(REF, "robust_e_nerf")):
        for fn in files:
            if not fn.endswith(".py"):
                continue
            tree = ast.parse(open(os.path.join(root, fn)).read())
            for node in ast.walk(tree):
                if isinstance(node, ast.Call):
                    f = node.func
                    name = f.attr if isinstance(f, ast.Attribute) else getattr(f, "id", None)
                    if name in found:
                        found[name] |= {k.arg for k in node.keywords if k.arg}
    return found


def build_reference_nerf(EasyDict, nerfacc, cfg, arch, radiance_dim=1):
    """what RobustENeRF._build_nerf does (models/robust_e_nerf.py:207-259), for the given arch"""
    import math
    from robust_e_nerf.models import nerf as rnerf
    n = EasyDict(cfg["model"]["nerf"])
    aabb = n.aabb if n.aabb != "auto" else [-1.5, -1.5, -1.5, 1.5, 1.5, 1.5]
    ct = {"aabb": nerfacc.ContractionType.AABB, "sphere": nerfacc.ContractionType.UN_BOUNDED_SPHERE,
          "tanh": nerfacc.ContractionType.UN_BOUNDED_TANH}[n.contraction_type]
    step = n.render_step_size
    if step == "auto":
        step = math.sqrt(3) * max(aabb[3 + k] - aabb[k] for k in range(3)) / 1024
    bkgd = "parameter" if cfg["model"].get("alpha_over_white_bg", True) is not None else None
    return rnerf.NeRF(aabb, ct, n.occ_grid, n.near_plane, n.far_plane, step, bkgd, n.cone_angle, n.early_stop_eps,
                      n.alpha_thre, n.test_chunk_size, arch, arch_config=n[arch], num_dim=3, radiance_dim=radiance_dim), aabb


def main():
    if not os.path.isdir(REF):
        raise SystemExit("build container only: /root/reference is absent")
    EasyDict, nerfacc_api, tcnn_api = install()
    import train as cli                                # scripts/train.py
    from robust_e_nerf_amd import engine, vanilla
    failures = []

    def check(ok, msg):
        print(("ok    " if ok else "FAIL  ") + msg)
        if not ok:
            failures.append(msg)

    # ---- 2. keywords at the reference's call sites
    seam = {"ray_marching": nerfacc_api.ray_marching, "OccupancyGrid": nerfacc_api.OccupancyGrid.__init__,
            "every_n_step": nerfacc_api.OccupancyGrid.every_n_step,
            "render_weight_from_density": nerfacc_api.render_weight_from_density,
            "render_weight_from_alpha": nerfacc_api.render_weight_from_alpha,
            "accumulate_along_rays": nerfacc_api.accumulate_along_rays, "Encoding": tcnn_api.Encoding.__init__}
    for name, kws in call_site_keywords(seam).items():
        params = inspect.signature(seam[name]).parameters
        var_kw = any(p.kind == p.VAR_KEYWORD for p in params.values())
        missing = sorted(k for k in kws if k not in params and not var_kw)
        check(not missing, f"{name}: reference call sites pass {sorted(kws)}" + (f" -- NOT accepted: {missing}" if missing else ""))

    # ---- 1, 3, 4 per YAML and arch
    for y in sorted(os.listdir(os.path.join(REF, "configs", "train"))):
        cfg = yaml.safe_load(open(os.path.join(REF, "configs", "train", y)))
        for arch in ("ngp", "mlp"):
            tag = f"{y} arch={arch}"
            try:
                ref, aabb = build_reference_nerf(EasyDict, nerfacc_api, cfg, arch)
            except Exception as e:                                                        # noqa: BLE001
                check(False, f"{tag}: reference NeRF over the seam failed to construct: {type(e).__name__}: {e}")
                continue
            check(True, f"{tag}: reference NeRF constructs over nerfacc_api / tcnn_api")
            rsd = {"nerf." + k: v for k, v in ref.state_dict().items()}
            # what scripts/train.py writes for the same configuration (its own code, on the CPU)
            res = cfg["model"]["nerf"]["occ_grid"]["resolution"]
            res = (res,) * 3 if isinstance(res, int) else tuple(res)
            if arch == "ngp":
                fld = engine.NGPField("cpu", 1, cfg["model"]["nerf"]["ngp"]["pos_encoding"])
                fld.flat.normal_()
            else:
                fld = vanilla.VanillaField("cpu", 1)
            sd = cli.field_state_dict(fld, arch, aabb)
            sd["nerf.parametrizations.render_bkgd.original"] = torch.ones(1)
            sd[cli.OCC + "_roi_aabb"] = torch.tensor(aabb, dtype=torch.float32)
            sd[cli.OCC + "_binary"] = torch.zeros(res, dtype=torch.bool)
            sd[cli.OCC + "resolution"] = torch.tensor(res, dtype=torch.int32)
            sd[cli.OCC + "occs"] = torch.zeros(res[0] * res[1] * res[2])
            only_ref, only_cli = sorted(set(rsd) - set(sd)), sorted(set(sd) - set(rsd))
            check(not only_ref and not only_cli, f"{tag}: state-dict keys equal ({len(rsd)} keys)" +
                  (f" -- only in the reference: {only_ref}; only in the CLI checkpoint: {only_cli}" if only_ref or only_cli else ""))
            bad = [k for k in set(rsd) & set(sd) if tuple(rsd[k].shape) != tuple(sd[k].shape)]
            check(not bad, f"{tag}: state-dict shapes equal" + (f" -- differ: {[(k, tuple(rsd[k].shape), tuple(sd[k].shape)) for k in bad]}" if bad else ""))
            try:
                ref.load_state_dict({k[len('nerf.'):]: v for k, v in sd.items()}, strict=True)
                check(True, f"{tag}: a scripts/train.py checkpoint loads strict=True into the reference module")
            except Exception as e:                                                        # noqa: BLE001
                check(False, f"{tag}: strict load failed: {e}")
        og = nerfacc_api.OccupancyGrid([-1.0] * 3 + [1.0] * 3, 8)
        check(set(og.state_dict()) == {"_roi_aabb", "_binary", "resolution", "occs"},
              f"{y}: seam OccupancyGrid persistent buffers = {sorted(og.state_dict())}")
    print(f"\n{'ALL OK' if not failures else str(len(failures)) + ' FAILURE(S)'}")
    raise SystemExit(1 if failures else 0)


if __name__ == "__main__":
    main()
