"""ctypes binding of the C-ABI HIP library (include/ren_amd.h).

The product path fails loudly when ``csrc/libren_amd.so`` is missing or a symbol is absent:
there is NO CPU fallback and the oracle is never imported from here.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_uint32, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "csrc", "libren_amd.so")
MAX_LEVELS = 16

REN_OK, REN_ERR_BAD_ARG, REN_ERR_UNSUPPORTED, REN_ERR_LAUNCH = 0, -1, -2, -3


class RenError(RuntimeError):
    pass


class GridDesc(Structure):
    _fields_ = [("n_levels", c_int32), ("scale", c_float * MAX_LEVELS), ("res", c_uint32 * MAX_LEVELS),
                ("size", c_uint32 * MAX_LEVELS), ("offset", c_uint32 * MAX_LEVELS),
                ("hashed", c_uint32 * MAX_LEVELS)]


class SceneDesc(Structure):
    _fields_ = [("aabb", c_float * 6), ("contraction_type", c_int32)]


P = c_void_p
# name -> (restype, argtypes); mirrors include/ren_amd.h one to one
SIGNATURES = {
    "ren_abi_version": (c_int, []),
    "ren_build_info": (c_char_p, []),
    "ren_set_knob": (c_int, [c_int32, c_int32]),
    "ren_get_knob": (c_int, [c_int32]),
    "ren_trajectory_fwd": (c_int, [P, c_int64, P, P, P, c_int64, P, P, P]),
    "ren_raygen_fwd": (c_int, [P, P, P, P, c_int64, P, P, P]),
    "ren_pose_rays_fwd": (c_int, [P, c_int64, P, c_int64, P, P, P, P, c_int64, P, P, P]),
    "ren_ray_aabb_intersect": (c_int, [P, P, c_int64, POINTER(c_float), c_float, c_float, P, P, P]),
    "ren_march_div_check": (c_int, [POINTER(c_float), P, P]),
    "ren_ray_march": (c_int, [P, P, P, P, P, c_int64, POINTER(c_float), POINTER(c_int32), P, c_int32,
                              c_float, c_float, c_int32, c_int32, P, P, P, P, P, P, c_int32, P]),
    "ren_compact_features": (c_int, [P, P, P, c_int64, P, P, P, P]),
    "ren_uniform": (c_int, [ctypes.c_uint64, ctypes.c_uint64, c_int64, P, P]),
    "ren_exclusive_scan": (c_int, [P, c_int64, P, P, P, P]),
    "ren_count_guard": (c_int, [P, P, c_int64, P, c_int64, P, P, P]),
    "ren_scan_guard": (c_int, [P, P, c_int64, P, P, c_int64, P, P, P, P]),
    "ren_frag_zero_tail": (c_int, [P, c_int64, P, P]),
    "ren_visibility": (c_int, [P, P, c_int64, P, P, P, c_float, c_float, P, P, P]),
    "ren_compact_samples": (c_int, [P, P, P, c_int64, P, P, P, P, P, P, P]),
    "ren_pack_info": (c_int, [P, c_int64, c_int64, P, P, P]),
    "ren_hashgrid_fwd": (c_int, [POINTER(GridDesc), P, P, POINTER(SceneDesc), P, P, P, P, P, c_int64, c_int32, P, P, P]),
    "ren_hashgrid_bwd": (c_int, [POINTER(GridDesc), P, P, POINTER(SceneDesc), P, P, P, P, P, c_int64, c_int32, P, P]),
    "ren_hashgrid_bwd_binned_workspace_bytes": (c_int64, [c_int64]),
    "ren_hashgrid_bwd_binned": (c_int, [POINTER(GridDesc), P, P, POINTER(SceneDesc), P, P, P, P, P, c_int64, c_int32, P, P, P, P]),
    "ren_hashgrid_bwd_binned_begin": (c_int, [POINTER(GridDesc), P, POINTER(SceneDesc), P, P, P, P, P, c_int64, c_int32, P, P]),
    "ren_hashgrid_bwd_binned_scatter": (c_int, [POINTER(GridDesc), P, P, POINTER(SceneDesc), P, P, P, P, P, c_int64, c_int32, P,
                                                c_int64, c_int64, P, P]),
    "ren_hashgrid_bwd_binned_finish": (c_int, [POINTER(GridDesc), P, c_int64, c_int32, P, P]),
    "ren_mlp_fwd": (c_int, [P, c_int32, c_int32, P, POINTER(SceneDesc), P, P, P, P, P, P, P, c_int64, c_int32, P, P, P, P]),
    "ren_mlp_bwd_workspace_floats": (c_int64, [c_int32]),
    "ren_mlp_bwd": (c_int, [P, c_int32, c_int32, P, P, POINTER(SceneDesc), P, P, P, P, P, P, P, c_int64, P, P, P, P, P, P, P, P]),
    "ren_mlp_fwd_bf16": (c_int, [P, c_int32, c_int32, P, POINTER(SceneDesc), P, P, P, P, P, P, P, c_int64, c_int32, P, P, P, P]),
    "ren_mlp_bwd_bf16": (c_int, [P, c_int32, c_int32, P, P, POINTER(SceneDesc), P, P, P, P, P, P, P, c_int64, P, P, P, P, P, P, P, P]),
    "ren_mlp_act_save_floats": (c_int64, [c_int64]),
    "ren_mlp_fwd_save": (c_int, [P, c_int32, c_int32, c_int32, P, POINTER(SceneDesc), P, P, P, P, P, P, P, c_int64, P, P, P, P, P]),
    "ren_mlp_bwd_saved": (c_int, [P, c_int32, c_int32, c_int32, P, P, P, POINTER(SceneDesc), P, P, P, P, P, P, P, c_int64, P, P, P,
                                  P, P, P, P, P]),
    "ren_act_jvp_fwd": (c_int, [P, c_int32, P, c_int32, c_float, P, c_int32, c_int64, c_int32, P]),
    "ren_act_jvp_bwd": (c_int, [P, P, P, c_int32, P, c_float, P, P, c_int64, c_int32, P]),
    "ren_event_prepare": (c_int, [P, P, P, P, P, P, P, c_int64, c_float, c_float, c_double, P, P, P, P, P, P, P, P, P, P]),
    "ren_event_params_refresh": (c_int, [P, c_float, P, c_double, P, P]),
    "ren_tau_adam_step": (c_int, [P, P, P, c_double, c_double, c_double, c_double, c_double, c_int64, c_double, P]),
    "ren_event_param_grad": (c_int, [c_int32, c_int32, c_int32, P, P, P, P, P, P, P, c_int64, c_float, c_float, c_float,
                                     c_double, c_float, P, P, P, P]),
    "ren_mlp_fwd_x": (c_int, [P, c_int32, c_int32, c_int32, P, POINTER(SceneDesc), P, P, P, P, P, P, P, c_int64, c_int32, P, P, P, P, P, P]),
    "ren_mlp_bwd_x_workspace_floats": (c_int64, [c_int32]),
    "ren_mlp_bwd_x": (c_int, [P, c_int32, c_int32, c_int32, P, P, P, POINTER(SceneDesc), P, P, P, P, P, P, P, c_int64, P, P, P, P, P,
                              P, P, c_int32, P, P]),
    "ren_composite_fwd": (c_int, [P, P, c_int64, P, P, P, P, c_int32, P, P, P, P, P, P, P]),
    "ren_composite_bwd": (c_int, [P, P, c_int64, P, P, P, P, c_int32, P, P, P, P, P, P, P, P, P, P, P, P]),
    "ren_event_loss_fwd": (c_int, [P, P, P, P, c_int64, c_int32, P, P]),
    "ren_event_loss_bwd": (c_int, [P, P, P, P, c_int64, c_int32, c_float, P, P, P, P]),
    "ren_event_diff_loss_fwd": (c_int, [P, P, P, c_int32, c_float, P, c_int32, c_int64, c_int32, P, P]),
    "ren_event_diff_loss_bwd": (c_int, [P, P, P, c_int32, c_float, P, c_int32, c_int64, c_int32, c_float, P, P, P, P, P, P, P, P]),
    "ren_bkgd_param_fwd": (c_int, [P, c_int32, P, P]),
    "ren_bkgd_param_grad": (c_int, [P, c_int64, c_int32, P, P, P, P]),
    "ren_step_tick": (c_int, [P, c_double, c_double, P, P, c_int32, P]),
    "ren_adam_step_dev": (c_int, [P, P, P, P, c_int64, c_float, c_float, c_float, c_float, c_float, P, c_float, c_int32, P]),
    "ren_tau_adam_step_dev": (c_int, [P, P, P, c_double, c_double, c_double, c_double, c_double, P, c_double, P]),
    "ren_adam_step": (c_int, [P, P, P, P, c_int64, c_float, c_float, c_float, c_float, c_float, c_int64,
                              c_float, c_int32, P]),
    "ren_occgrid_cell_points": (c_int, [P, P, c_int64, POINTER(c_float), POINTER(c_int32), c_int32, P, P, P]),
    "ren_occgrid_ema": (c_int, [P, P, P, P, P, c_float, c_int64, c_float, P]),
    "ren_occgrid_ema_unique": (c_int, [P, P, P, P, P, c_float, c_int64, c_float, P, P]),
    "ren_occgrid_binarize": (c_int, [P, c_int64, c_float, P, P, P]),
    "ren_column_sum": (c_int, [P, c_int64, c_int32, P, P, P]),
    "ren_trajectory_jvp": (c_int, [P, c_int64, P, P, P, c_int64, P, P, P, P, P]),
    "ren_raygen_jvp": (c_int, [P, P, P, P, P, P, c_int64, P, P, P, P, P]),
    "ren_hashgrid_fwd_jvp": (c_int, [POINTER(GridDesc), P, POINTER(SceneDesc), P, P, P, P, P, P, P, c_int64, P, P, P, P]),
    "ren_hashgrid_bwd_jvp": (c_int, [POINTER(GridDesc), P, POINTER(SceneDesc), P, P, P, P, P, P, P, c_int64, P, P, P]),
    "ren_hashgrid_bwd_binned_jvp": (c_int, [POINTER(GridDesc), P, POINTER(SceneDesc), P, P, P, P, P, P, P, c_int64, P, P, P, P]),
    "ren_hashgrid_bwd_binned_levels": (c_int, [POINTER(GridDesc), P, P, POINTER(SceneDesc), P, P, P, P, P, c_int64, c_int32, P,
                                               P, P, P, c_uint32, P, P, P]),
    "ren_mlp_fwd_jvp": (c_int, [P, c_int32, c_int32, P, P, POINTER(SceneDesc), P, P, P, P, P, P, c_int64, P, P, P, P, P, P, P]),
    "ren_mlp_bwd_jvp_workspace_floats": (c_int64, [c_int32]),
    "ren_mlp_bwd_jvp": (c_int, [P, c_int32, c_int32, P, P, P, P, POINTER(SceneDesc), P, P, P, P, P, P, c_int64, P, P, P, P, P,
                                P, P, P, P, P, P]),
    "ren_mlp_fwd_jvp_x": (c_int, [P, c_int32, c_int32, c_int32, P, P, POINTER(SceneDesc), P, P, P, P, P, P, c_int64, P, P, P, P, P, P, P, P]),
    "ren_mlp_bwd_jvp_x_workspace_floats": (c_int64, [c_int32]),
    "ren_mlp_bwd_jvp_x": (c_int, [P, c_int32, c_int32, c_int32, P, P, P, P, POINTER(SceneDesc), P, P, P, P, P, P, c_int64, P, P, P, P, P,
                                  P, P, P, P, P, P, P]),
    "ren_composite_fwd_jvp": (c_int, [P, P, c_int64, P, P, P, P, P, P, c_int32, P, P, P, P, P, P, P, P, P]),
    "ren_composite_bwd_jvp": (c_int, [P, P, c_int64, P, P, P, P, P, P, c_int32, P, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    "ren_trajectory_jvp2": (c_int, [P, c_int64, P, P, P, c_int64, P, P, P, P, P, P]),
    "ren_raygen_jvp2": (c_int, [P, P, P, P, P, P, P, c_int64, P, P, P, P, P, P]),
    "ren_hashgrid_fwd_jvp2": (c_int, [POINTER(GridDesc), P, POINTER(SceneDesc), P, P, P, P, P, P, P, P, c_int64, P, P, P, P, P]),
    "ren_mlp_fwd_jvp2": (c_int, [P, c_int32, c_int32, P, P, P, POINTER(SceneDesc), P, P, P, P, P, P, P, P, c_int64, P, P, P, P, P, P, P]),
    "ren_mlp_fwd_jvp2_x": (c_int, [P, c_int32, c_int32, c_int32, P, P, P, POINTER(SceneDesc), P, P, P, P, P, P, P, P, c_int64, P, P, P, P, P, P,
                                   P, P]),
    "ren_composite_fwd_jvp2": (c_int, [P, P, c_int64, P, P, P, P, P, P, P, P, c_int32, P, P, P, P, P]),
    "ren_freq_encode": (c_int, [POINTER(SceneDesc), P, P, P, P, P, P, P, c_int64, P, c_int32, P, c_int32, c_int32, P,
                                c_int32, c_int32, P, P]),
    "ren_dense_fwd": (c_int, [P, c_int32, P, P, c_int32, c_int32, c_int32, P, P, c_int32, c_int64, P]),
    "ren_dense_bwd_data": (c_int, [P, c_int32, P, c_int32, c_int32, c_int32, c_int32, P, c_int32, c_int32, P, c_int32,
                                   c_int64, P]),
    "ren_dense_bwd_weight_workspace_floats": (c_int64, [c_int32, c_int32, c_int32]),
    "ren_dense_bwd_weight": (c_int, [P, c_int32, P, c_int32, c_int32, c_int32, c_int64, c_int32, P, P, P, P]),
    "ren_vanilla_heads_bwd": (c_int, [P, P, P, P, c_int64, c_int32, c_int32, P, P, P]),
    "ren_vanilla_image_bytes": (c_int64, [c_int32]),
    "ren_vanilla_saved_bytes": (c_int64, [c_int32, c_int64]),
    "ren_vanilla_prep": (c_int, [P, c_int32, c_int32, P, P]),
    "ren_vanilla_fwd": (c_int, [P, c_int32, P, c_int32, P, P, c_int32, c_int32, P, c_int32, c_int64, P, P, P, P]),
    "ren_vanilla_bwd": (c_int, [P, P, P, c_int32, c_int32, c_int64, P, c_int64, P, P]),
    "ren_vanilla_bwd_weight_workspace_floats": (c_int64, [c_int32]),
    "ren_vanilla_bwd_weight": (c_int, [P, P, c_int64, P, c_int32, P, c_int32, P, P, c_int32, c_int32, c_int64, c_int32, P, P, P]),
    "ren_vanilla_bwd_weight_tangent": (c_int, [P, P, c_int64, P, c_int32, P, c_int32, P, P, c_int32, c_int32, c_int64, c_int32, P, P, P]),
    "ren_vanilla_fwd_jvp": (c_int, [P, c_int32, P, c_int32, P, P, P, P, c_int32, c_int32, P, c_int32, c_int64, P, P, P, P, P, P, P]),
    "ren_vanilla_bwd_jvp": (c_int, [P, P, P, P, P, c_int32, c_int32, c_int64, P, P, c_int64, P, P, P, P]),
    "ren_freq_encode_jvp": (c_int, [POINTER(SceneDesc), P, P, P, P, P, P, P, P, c_int64, c_int32, P, c_int32, P, c_int32, c_int32,
                                    P, c_int32, c_int32, P]),
    "ren_act_jvp2_fwd": (c_int, [P, c_int32, P, P, c_int32, c_float, P, c_int32, P, c_int32, c_int64, c_int32, P]),
    "ren_vanilla_heads_jvp": (c_int, [P, P, P, P, P, P, c_int64, c_int32, c_int32, P, P, P, P, P]),
    "ren_vanilla_heads_bwd_jvp": (c_int, [P, P, P, P, P, P, P, P, c_int64, c_int32, c_int32, P, P, P, P, P]),
    "ren_rate_epilogue": (c_int, [P, P, P, P, c_int32, c_int64, c_float, P, P, P, P, P]),
    "ren_tau_pose_grad": (c_int, [P, P, P, P, P, c_int64, P, P]),
    "ren_weight_norm_fwd": (c_int, [P, P, P, c_int32, c_int64, P, P]),
    "ren_weight_norm_bwd": (c_int, [P, P, P, P, c_int32, c_int64, P, P, c_int32, P]),
    "ren_grad_loss_fwd": (c_int, [P, P, P, P, c_int64, c_int32, P, P]),
    "ren_grad_loss_bwd": (c_int, [P, P, P, P, c_int64, c_int32, c_float, P, P, P, P, P, P]),
}

_lib = None


def load() -> ctypes.CDLL:
    """Load libren_amd.so, bind every symbol of include/ren_amd.h, fail loudly otherwise."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RenError(
            f"{LIB_PATH} is missing: build it with `python -m robust_e_nerf_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    # The library must be the one these sources produce (content stamp, robust_e_nerf_amd/build.py): a stale .so -- sources
    # edited or reverted without a rebuild -- measures and tests something else than the tree says (it happened in round 4).
    from . import build as _build
    if os.environ.get("REN_ALLOW_STALE_LIB") != "1" and not _build.is_current():
        if not os.path.exists(_build.STAMP):          # a library that travelled without its stamp cannot be checked: say so, load it
            import sys
            print(f"[ren_amd] warning: {_build.STAMP} is missing, cannot verify that the library matches the sources", file=sys.stderr)
        else:
            raise RenError(f"{LIB_PATH} is STALE: it was not built from the sources next to it; run "
                           "`python -m robust_e_nerf_amd.build` (or set REN_ALLOW_STALE_LIB=1 to load it anyway)")
    # PyTorch-ROCm bundles its own libamdhip64; it must be in the process BEFORE this library so
    # both resolve to the SAME HIP runtime (otherwise torch's device pointers are foreign to our
    # launches and every kernel fails with a launch error).
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RenError(f"{LIB_PATH} does not export {name}; rebuild the library") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


_ERR = {REN_ERR_BAD_ARG: (ValueError, "bad argument"),
        REN_ERR_UNSUPPORTED: (NotImplementedError, "unsupported configuration"),
        REN_ERR_LAUNCH: (RenError, "HIP launch failed")}


def check(rc: int, what: str):
    """Map C error codes to the Python exception types the reference raises."""
    if rc != REN_OK:
        exc, msg = _ERR.get(rc, (RenError, f"error {rc}"))
        raise exc(f"{what}: {msg} (ren_status {rc})")
