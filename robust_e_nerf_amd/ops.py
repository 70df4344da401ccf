"""Thin tensor-level wrappers over the C ABI: torch owns memory and streams, the HIP library does
the arithmetic.  Every function enqueues on ``torch.cuda.current_stream()`` and never syncs.
"""
from __future__ import annotations

import ctypes
import math
from typing import Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import GridDesc, SceneDesc, check

AABB, UN_BOUNDED_TANH, UN_BOUNDED_SPHERE = 0, 1, 2
FRAG_FLOATS_PER_BLOCK = 16 * 64          # hash features, fragment layout, per 32 samples
BASE_FLOATS_PER_BLOCK = 8 * 64           # saved base-MLP outputs, per 32 samples

_DT = {torch.float32: 4, torch.float64: 8, torch.int32: 4, torch.int64: 8, torch.uint8: 1, torch.bool: 1}


KNOBS = {"hgb_no_pairs": 0, "hgb_halve_regions": 1, "march_sequential": 2, "hg_variant": 3, "vfield_plain": 4, "hgb_subregion": 5}     # include/ren_amd.h REN_KNOB_*


class knob:
    """`with ops.knob("hgb_halve_regions", 1): ...` -- set a verification / tuning knob of the library for a block"""

    def __init__(self, name: str, value: int):
        self.k, self.v = KNOBS[name], int(value)

    def __enter__(self):
        lib = _lib.load()
        self.old = lib.ren_get_knob(self.k)
        check(lib.ren_set_knob(self.k, self.v), "ren_set_knob")

    def __exit__(self, *a):
        _lib.load().ren_set_knob(self.k, self.old)


# Activation alternatives of the YAML (robust_e_nerf/models/nerf.py:8-29) -> the `activations` argument of the ren_mlp_* /
# ren_vanilla_* entry points (include/ren_amd.h); every wrapper below takes it as `act` (0 = the shipped configs)
HIDDEN_ACTS = {"softplus": 0, "relu": 1}
DENSITY_ACTS = {"shifted_trunc_exp": 0, "softplus": 1, "shifted_softplus": 2}
RADIANCE_ACTS = {"softplus": 0, "sigmoid": 1}


def activation_code(base_hidden="softplus", density="shifted_trunc_exp", head_hidden="softplus", radiance="softplus") -> int:
    """0 = the shipped configs; unknown names raise NotImplementedError as the reference's table lookups would KeyError"""
    try:
        return (HIDDEN_ACTS[base_hidden] | DENSITY_ACTS[density] << 2 | HIDDEN_ACTS[head_hidden] << 4 |
                RADIANCE_ACTS[radiance] << 6)
    except KeyError as e:
        raise NotImplementedError(f"activation {e.args[0]!r} (models/nerf.py:17-29)") from e


def _ptr(t: Optional[torch.Tensor], dtype=None):
    if t is None:
        return None
    if not t.is_cuda:
        raise ValueError("robust_e_nerf_amd kernels take device (ROCm) tensors; got a CPU tensor")
    if not t.is_contiguous():
        raise ValueError("tensor must be contiguous")
    if dtype is not None and t.dtype not in (dtype if isinstance(dtype, tuple) else (dtype,)):
        raise ValueError(f"expected dtype {dtype}, got {t.dtype}")
    return ctypes.c_void_p(t.data_ptr())


try:                                     # the raw handle of torch's current stream: two C calls instead of the ~8 us of Python that
    _raw_stream, _cur_device = torch._C._cuda_getCurrentRawStream, torch._C._cuda_getDevice   # torch.cuda.current_stream() costs per launch
except AttributeError:                   # (a torch build without them)
    _raw_stream = None


def _stream():
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(_cur_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f(x):
    return ctypes.c_float(float(x))


NAN = float("nan")


# ------------------------------------------------------------------------------- descriptors
def make_grid_desc(n_levels=16, n_features_per_level=2, log2_hashmap_size=19, base_resolution=16,
                   per_level_scale=1.4472692012786865, otype="HashGrid") -> Tuple[GridDesc, int]:
    """tcnn grid level table (float32 arithmetic) -> (descriptor, number of float params).  otype (the YAML's
    pos_encoding.otype, configs/train/synthetic.yaml:63): HashGrid (level size capped at 2^log2_hashmap_size, spatial hash
    beyond), DenseGrid (no cap, never hashed), TiledGrid (capped at base_resolution^3, the dense index wraps)."""
    import numpy as np
    if otype not in ("HashGrid", "DenseGrid", "TiledGrid"):
        raise NotImplementedError(f"pos_encoding.otype={otype}")
    if n_features_per_level != 2:
        raise NotImplementedError("n_features_per_level must be 2")
    if n_levels > _lib.MAX_LEVELS:
        raise NotImplementedError("n_levels must be <= 16")
    g = GridDesc()
    g.n_levels = n_levels
    log2_pls = np.log2(np.float32(per_level_scale)).astype(np.float32)
    offset = 0
    for lvl in range(n_levels):
        scale = np.float32(np.exp2(np.float32(lvl) * log2_pls).astype(np.float32) * np.float32(base_resolution)
                           - np.float32(1.0))
        res = int(np.ceil(scale)) + 1
        dense = res ** 3
        size = (min(dense, 2 ** 31 - 1) + 7) // 8 * 8
        if otype == "HashGrid":
            size = min(size, 1 << log2_hashmap_size)
        elif otype == "TiledGrid":
            size = min(size, base_resolution ** 3)
        if offset + size >= 1 << 32:
            raise NotImplementedError("grid with more than 2^32 entries")
        g.scale[lvl] = float(scale)
        g.res[lvl] = res
        g.size[lvl] = size
        g.offset[lvl] = offset
        g.hashed[lvl] = 1 if (otype == "HashGrid" and dense > size) else 0
        offset += size
    return g, offset * 2


def make_scene_desc(aabb: Sequence[float], contraction_type: int) -> SceneDesc:
    s = SceneDesc()
    for k in range(6):
        s.aabb[k] = float(aabb[k])
    s.contraction_type = int(contraction_type)
    return s


def mlp_param_count(radiance_dim: int = 1) -> int:
    return 9360 + 65 * radiance_dim


MLP_SLICES = {  # name -> (offset, shape) for radiance_dim C (filled by mlp_slices)
}


def mlp_slices(C: int = 1):
    """Views of the concatenated MLP parameter block in torch nn.Linear layout."""
    return {
        "base.w0": (0, (64, 32)), "base.b0": (2048, (64,)), "base.wo": (2112, (16, 64)), "base.bo": (3136, (16,)),
        "head.w0": (3152, (64, 31)), "head.b0": (5136, (64,)), "head.w1": (5200, (64, 64)), "head.b1": (9296, (64,)),
        "head.wo": (9360, (C, 64)), "head.bo": (9360 + 64 * C, (C,)),
    }


# ---- torch.nn.utils.weight_norm over the Linear layers of an MLP (ngp.py:207-228, mlp.py:303-319) -------------------
def weight_norm_layers(layers):
    """layers: [(weight offset, rows, cols, first g index)] -> the host int32 table ren_weight_norm_* take."""
    import ctypes
    flat = [int(v) for layer in layers for v in layer]
    return (ctypes.c_int32 * len(flat))(*flat), len(layers)


def weight_norm_fwd(raw, g, table, eff):
    """eff = raw with W[r, :] = g[r] v[r, :] / ||v[r, :]|| in the listed layers"""
    import ctypes
    arr, n_layers = table
    check(_lib.load().ren_weight_norm_fwd(_ptr(raw), _ptr(g), ctypes.cast(arr, ctypes.c_void_p), n_layers, raw.numel(),
                                          _ptr(eff), _stream()), "ren_weight_norm_fwd")
    return eff


def weight_norm_bwd(raw, g, d_eff, table, d_raw, d_g, zero_d_eff: bool = False):
    """(d_raw, d_g) = gradient w.r.t. (v / biases / plain weights, g) from d_eff; zero_d_eff: clear d_eff afterwards"""
    import ctypes
    arr, n_layers = table
    check(_lib.load().ren_weight_norm_bwd(_ptr(raw), _ptr(g), _ptr(d_eff), ctypes.cast(arr, ctypes.c_void_p), n_layers,
                                          raw.numel(), _ptr(d_raw), _ptr(d_g), int(zero_d_eff), _stream()),
          "ren_weight_norm_bwd")


def n_blocks32(n: int) -> int:
    return (n + 31) // 32


# ------------------------------------------------------------------------------- pose / rays
def trajectory(ts: torch.Tensor, tab_ts, tab_pos, tab_quat):
    B = ts.shape[0]
    pos = torch.empty(B, 3, device=ts.device, dtype=torch.float32)
    rot = torch.empty(B, 3, 3, device=ts.device, dtype=torch.float32)
    check(_lib.load().ren_trajectory_fwd(_ptr(ts, torch.float64), B, _ptr(tab_ts, torch.int64),
                                         _ptr(tab_pos, torch.float32), _ptr(tab_quat, torch.float32),
                                         tab_ts.shape[0], _ptr(pos), _ptr(rot), _stream()), "ren_trajectory_fwd")
    return pos, rot


def raygen(Kinv, px, pos, rot):
    B = px.shape[0]
    o = torch.empty(B, 3, device=px.device, dtype=torch.float32)
    d = torch.empty(B, 3, device=px.device, dtype=torch.float32)
    check(_lib.load().ren_raygen_fwd(_ptr(Kinv, torch.float32), _ptr(px, torch.float32), _ptr(pos, torch.float32),
                                     _ptr(rot, torch.float32), B, _ptr(o), _ptr(d), _stream()), "ren_raygen_fwd")
    return o, d


def pose_rays(ts, px, Kinv, tab_ts, tab_pos, tab_quat):
    """trajectory() + raygen() in one launch; px (B, 2) is reused cyclically when ts has a multiple of B entries"""
    R = ts.shape[0]
    o = torch.empty(R, 3, device=ts.device, dtype=torch.float32)
    d = torch.empty(R, 3, device=ts.device, dtype=torch.float32)
    check(_lib.load().ren_pose_rays_fwd(_ptr(ts, torch.float64), R, _ptr(px, torch.float32), px.shape[0], _ptr(Kinv, torch.float32),
                                        _ptr(tab_ts, torch.int64), _ptr(tab_pos, torch.float32), _ptr(tab_quat, torch.float32),
                                        tab_ts.shape[0], _ptr(o), _ptr(d), _stream()), "ren_pose_rays_fwd")
    return o, d


# ------------------------------------------------------------------------------- sampling
def ray_aabb_intersect(o, d, aabb: Sequence[float], near: Optional[float] = None, far: Optional[float] = None):
    n = o.shape[0]
    tmin = torch.empty(n, device=o.device, dtype=torch.float32)
    tmax = torch.empty(n, device=o.device, dtype=torch.float32)
    ab = (ctypes.c_float * 6)(*[float(v) for v in aabb])
    check(_lib.load().ren_ray_aabb_intersect(_ptr(o, torch.float32), _ptr(d, torch.float32), n, ab,
                                             _f(NAN if near is None else near), _f(NAN if far is None else far),
                                             _ptr(tmin), _ptr(tmax), _stream()), "ren_ray_aabb_intersect")
    return tmin, tmax


def uniform(n: int, seed: int, offset: int = 0, device="cuda:0", out=None):
    """n iid U[0, 1) floats from the library's Philox stream (seed, offset): jitter without a torch RNG launch"""
    if out is None:
        out = torch.empty(n, device=device, dtype=torch.float32)
    check(_lib.load().ren_uniform(int(seed) & (2 ** 64 - 1), int(offset) & (2 ** 64 - 1), n, _ptr(out), _stream()), "ren_uniform")
    return out


def exclusive_scan(counts: torch.Tensor):
    n = counts.shape[0]
    offsets = torch.empty(n, device=counts.device, dtype=torch.int64)
    total = torch.empty(1, device=counts.device, dtype=torch.int64)
    scratch = torch.empty(1024, device=counts.device, dtype=torch.int64) if n > 65536 else None
    check(_lib.load().ren_exclusive_scan(_ptr(counts, torch.int32), n, _ptr(offsets), _ptr(total), _ptr(scratch),
                                         _stream()), "ren_exclusive_scan")
    return offsets, total


MARCH_VERIFIED_DIV = 0x100          # include/ren_amd.h REN_MARCH_VERIFIED_DIV: OR into `mode` of ray_march_count / _write


def march_div_check(roi, device) -> bool:
    """True when the marcher's multiply-add division is bit-identical to the IEEE division for the extents of `roi` (every
    float numerator of 2^-100 < |a| < 2^100 tried on the device, ~20 ms): the caller may then pass mode | MARCH_VERIFIED_DIV"""
    ext = [float(roi[3 + k]) - float(roi[k]) for k in range(3)]
    if not all(2.0 ** -59 < e < 2.0 ** 59 for e in ext) or not all(2.0 ** -59 < abs(float(roi[k])) < 2.0 ** 59 for k in range(3)):
        return False                                  # (a box corner at 0: differences p - corner could leave the checked range)
    roi_c = (ctypes.c_float * 6)(*[float(v) for v in roi])
    out = torch.zeros(1, dtype=torch.int64, device=device)
    check(_lib.load().ren_march_div_check(roi_c, _ptr(out, torch.int64), _stream()), "ren_march_div_check")
    return int(out.item()) == 0


def ray_march_count(o, d, t_min, t_max, jitter, roi, res, binary, ct, step, cone, mode, n_uniform, cache=None):
    """cache: optional float32 (n_rays, cap, 2) interval cache filled here and consumed by ray_march_write"""
    n = o.shape[0]
    counts = torch.empty(n, device=o.device, dtype=torch.int32)
    roi_c = (ctypes.c_float * 6)(*[float(v) for v in roi])
    res_c = (ctypes.c_int32 * 3)(*[int(v) for v in res])
    check(_lib.load().ren_ray_march(_ptr(o), _ptr(d), _ptr(t_min), _ptr(t_max), _ptr(jitter), n, roi_c, res_c,
                                    _ptr(binary, (torch.uint8, torch.bool)), ct, _f(step), _f(cone), mode, n_uniform,
                                    None, _ptr(counts), None, None, None, _ptr(cache, torch.float32),
                                    0 if cache is None else cache.shape[1], _stream()), "ren_ray_march(count)")
    return counts


def ray_march_write(o, d, t_min, t_max, jitter, roi, res, binary, ct, step, cone, mode, n_uniform,
                    offsets, n_total: int, out=None, counts=None, cache=None):
    n = o.shape[0]
    if out is None:
        ri = torch.empty(n_total, device=o.device, dtype=torch.int32)
        ts = torch.empty(n_total, device=o.device, dtype=torch.float32)
        te = torch.empty(n_total, device=o.device, dtype=torch.float32)
    else:
        ri, ts, te = out
    if n_total == 0:                                  # no ray meets an occupied cell: nothing to write
        return ri, ts, te
    roi_c = (ctypes.c_float * 6)(*[float(v) for v in roi])
    res_c = (ctypes.c_int32 * 3)(*[int(v) for v in res])
    check(_lib.load().ren_ray_march(_ptr(o), _ptr(d), _ptr(t_min), _ptr(t_max), _ptr(jitter), n, roi_c, res_c,
                                    _ptr(binary, (torch.uint8, torch.bool)), ct, _f(step), _f(cone), mode, n_uniform,
                                    _ptr(offsets, torch.int64), _ptr(counts, torch.int32), _ptr(ri), _ptr(ts), _ptr(te),
                                    _ptr(cache, torch.float32) if counts is not None else None,
                                    0 if (cache is None or counts is None) else cache.shape[1], _stream()),
          "ren_ray_march(write)")
    return ri, ts, te


def count_guard(counts, total, capacity: int, n_out, stats=None, counts_also=None):
    """device-side sample count: n_out[0] = total if it fits `capacity`, else 0 with every ray's count (and counts_also's)
    cleared; stats = (total as found, 1 if it did not fit else 0); see include/ren_amd.h ("device-side sample counts")"""
    check(_lib.load().ren_count_guard(_ptr(counts, torch.int32), _ptr(counts_also, torch.int32), counts.shape[0],
                                      _ptr(total, torch.int64), int(capacity), _ptr(n_out, torch.int64), _ptr(stats, torch.int64),
                                      _stream()), "ren_count_guard")


def scan_guard(counts, capacity: int, n_out, stats=None, counts_also=None):
    """exclusive_scan + count_guard (one launch for up to 65 536 rays) -> offsets, total"""
    n = counts.shape[0]
    offsets = torch.empty(n, device=counts.device, dtype=torch.int64)
    total = torch.empty(1, device=counts.device, dtype=torch.int64)
    scratch = torch.empty(1024, device=counts.device, dtype=torch.int64) if n > 65536 else None
    check(_lib.load().ren_scan_guard(_ptr(counts, torch.int32), _ptr(counts_also, torch.int32), n, _ptr(offsets), _ptr(total),
                                     int(capacity), _ptr(n_out, torch.int64), _ptr(stats, torch.int64), _ptr(scratch), _stream()),
          "ren_scan_guard")
    return offsets, total


def frag_zero_tail(feat, capacity: int, n_dev):
    check(_lib.load().ren_frag_zero_tail(_ptr(feat, torch.float32), int(capacity), _ptr(n_dev, torch.int64), _stream()),
          "ren_frag_zero_tail")


def visibility(offsets, counts, sigmas, ts, te, eps: float, alpha_thre: float):
    n_rays = counts.shape[0]
    keep = torch.empty(ts.shape[0], device=ts.device, dtype=torch.uint8)
    kept = torch.empty(n_rays, device=ts.device, dtype=torch.int32)
    check(_lib.load().ren_visibility(_ptr(offsets, torch.int64), _ptr(counts, torch.int32), n_rays,
                                     _ptr(sigmas, torch.float32), _ptr(ts), _ptr(te), _f(eps), _f(alpha_thre),
                                     _ptr(keep), _ptr(kept), _stream()), "ren_visibility")
    return keep, kept


def compact_samples(offsets, counts, new_offsets, keep, ts, te, n_new: int):
    n_rays = counts.shape[0]
    ri2 = torch.empty(n_new, device=ts.device, dtype=torch.int32)
    ts2 = torch.empty(n_new, device=ts.device, dtype=torch.float32)
    te2 = torch.empty(n_new, device=ts.device, dtype=torch.float32)
    if n_new == 0:                                    # everything culled by the visibility test
        return ri2, ts2, te2
    check(_lib.load().ren_compact_samples(_ptr(offsets), _ptr(counts), _ptr(new_offsets), n_rays, _ptr(keep),
                                          _ptr(ts), _ptr(te), _ptr(ri2), _ptr(ts2), _ptr(te2), _stream()),
          "ren_compact_samples")
    return ri2, ts2, te2


def compact_features(offsets, counts, new_offsets, keep, feat, n_new: int, n_dev=None):
    """fragment-layout features of the kept samples at their compacted positions (padding lanes zero).
    n_dev: n_new is the capacity, the number of kept samples is on the device"""
    out = torch.empty(n_blocks32(n_new) * FRAG_FLOATS_PER_BLOCK, device=feat.device, dtype=torch.float32)
    if n_new == 0:
        return out
    if n_dev is None:
        out[-FRAG_FLOATS_PER_BLOCK:].zero_()
    check(_lib.load().ren_compact_features(_ptr(offsets, torch.int64), _ptr(counts, torch.int32), _ptr(new_offsets, torch.int64),
                                           counts.shape[0], _ptr(keep), _ptr(feat, torch.float32), _ptr(out), _stream()),
          "ren_compact_features")
    if n_dev is not None:
        frag_zero_tail(out, n_new, n_dev)            # lanes beyond the count in its last 32-sample block
    return out


def pack_info(ray_indices: torch.Tensor, n_rays: int):
    n = ray_indices.shape[0]
    offsets = torch.empty(n_rays, device=ray_indices.device, dtype=torch.int64)
    counts = torch.empty(n_rays, device=ray_indices.device, dtype=torch.int32)
    check(_lib.load().ren_pack_info(_ptr(ray_indices, torch.int32), n, n_rays, _ptr(offsets), _ptr(counts),
                                    _stream()), "ren_pack_info")
    return offsets, counts


# ------------------------------------------------------------------------------- hash grid
def hashgrid_fwd(grid: GridDesc, table, *, x_unit=None, scene: Optional[SceneDesc] = None, rays=None,
                 samples=None, n: int, layout: int, out=None, n_dev=None):
    L = grid.n_levels
    dev = table.device
    if out is None:
        out = (torch.empty(n, 2 * L, device=dev, dtype=torch.float32) if layout == 0
               else torch.empty(n_blocks32(n) * FRAG_FLOATS_PER_BLOCK, device=dev, dtype=torch.float32))
    o, d = rays if rays is not None else (None, None)
    ri, ts, te = samples if samples is not None else (None, None, None)
    check(_lib.load().ren_hashgrid_fwd(ctypes.byref(grid), _ptr(table, torch.float32), _ptr(x_unit),
                                       ctypes.byref(scene) if scene is not None else None,
                                       _ptr(o), _ptr(d), _ptr(ri), _ptr(ts), _ptr(te), n, layout, _ptr(out), _ptr(n_dev, torch.int64),
                                       _stream()), "ren_hashgrid_fwd")
    return out


def hashgrid_bwd(grid: GridDesc, grad_table, dfeat, *, x_unit=None, scene=None, rays=None, samples=None,
                 n: int, layout: int):
    o, d = rays if rays is not None else (None, None)
    ri, ts, te = samples if samples is not None else (None, None, None)
    check(_lib.load().ren_hashgrid_bwd(ctypes.byref(grid), _ptr(grad_table, torch.float32), _ptr(x_unit),
                                       ctypes.byref(scene) if scene is not None else None,
                                       _ptr(o), _ptr(d), _ptr(ri), _ptr(ts), _ptr(te), n, layout,
                                       _ptr(dfeat, torch.float32), _stream()), "ren_hashgrid_bwd")


def hashgrid_bwd_binned_workspace_bytes(n: int) -> int:
    return int(_lib.load().ren_hashgrid_bwd_binned_workspace_bytes(n))


def hashgrid_bwd_binned(grid: GridDesc, grad_table, dfeat, workspace, *, x_unit=None, scene=None, rays=None,
                        samples=None, n: int, layout: int, level_mask: Optional[int] = None, tangent=None, n_dev=None):
    """LDS-binned (atomic-free) variant of hashgrid_bwd; workspace: uint8 tensor of
    hashgrid_bwd_binned_workspace_bytes(n) bytes.  level_mask: only the levels whose bit is set; tangent: (rays_do,
    rays_dd, dfeatd) of the log-intensity-gradient render."""
    if workspace.numel() * workspace.element_size() < hashgrid_bwd_binned_workspace_bytes(n):
        raise ValueError("hashgrid_bwd_binned: workspace too small")
    o, d = rays if rays is not None else (None, None)
    ri, ts, te = samples if samples is not None else (None, None, None)
    if level_mask is not None or tangent is not None:
        do, dd, dfd = tangent if tangent is not None else (None, None, None)
        check(_lib.load().ren_hashgrid_bwd_binned_levels(
            ctypes.byref(grid), _ptr(grad_table, torch.float32), _ptr(x_unit), ctypes.byref(scene) if scene is not None else None,
            _ptr(o), _ptr(d), _ptr(ri), _ptr(ts), _ptr(te), n, layout, _ptr(dfeat, torch.float32), _ptr(do), _ptr(dd), _ptr(dfd),
            0xFFFFFFFF if level_mask is None else int(level_mask), _ptr(workspace), _ptr(n_dev, torch.int64), _stream()),
              "ren_hashgrid_bwd_binned_levels")
        return
    check(_lib.load().ren_hashgrid_bwd_binned(ctypes.byref(grid), _ptr(grad_table, torch.float32), _ptr(x_unit),
                                              ctypes.byref(scene) if scene is not None else None,
                                              _ptr(o), _ptr(d), _ptr(ri), _ptr(ts), _ptr(te), n, layout,
                                              _ptr(dfeat, torch.float32), _ptr(workspace), _ptr(n_dev, torch.int64), _stream()),
          "ren_hashgrid_bwd_binned")


BINNED_MAX_LEVEL_ENTRIES = 1 << 19          # 64 bins of 8 192 entries: what ren_hashgrid_bwd_binned takes per level


def binned_supported(grid: GridDesc) -> bool:
    """False for a DenseGrid with a level above 2^19 entries: such a grid takes the per-update atomic scatter"""
    return max(int(grid.size[l]) for l in range(grid.n_levels)) <= BINNED_MAX_LEVEL_ENTRIES


def hashgrid_bwd_auto(grid: GridDesc, grad_table, dfeat, *, x_unit=None, scene=None, rays=None, samples=None, n: int,
                      layout: int):
    """parameter gradient of the module seams (field.NGPradianceField, tcnn_api.Encoding): the binned scatter with a
    workspace of its own, or the atomic scatter for grids the binned one refuses (as engine.Renderer chooses)"""
    if binned_supported(grid):
        ws = torch.empty(hashgrid_bwd_binned_workspace_bytes(n), device=dfeat.device, dtype=torch.uint8)
        hashgrid_bwd_binned(grid, grad_table, dfeat, ws, x_unit=x_unit, scene=scene, rays=rays, samples=samples, n=n,
                            layout=layout)
    else:
        hashgrid_bwd(grid, grad_table, dfeat, x_unit=x_unit, scene=scene, rays=rays, samples=samples, n=n, layout=layout)


def hashgrid_bwd_binned_begin(grid: GridDesc, workspace, *, scene, rays, samples, n: int, layout: int = 1):
    """phase 1 of hashgrid_bwd_binned: clear + count + offsets (sample stream only)"""
    if workspace.numel() * workspace.element_size() < hashgrid_bwd_binned_workspace_bytes(n):
        raise ValueError("hashgrid_bwd_binned: workspace too small")
    (o, d), (ri, ts, te) = rays, samples
    check(_lib.load().ren_hashgrid_bwd_binned_begin(ctypes.byref(grid), None, ctypes.byref(scene), _ptr(o), _ptr(d), _ptr(ri),
                                                    _ptr(ts), _ptr(te), n, layout, _ptr(workspace), _stream()),
          "ren_hashgrid_bwd_binned_begin")


def hashgrid_bwd_binned_scatter(grid: GridDesc, grad_table, dfeat, workspace, *, scene, rays, samples, n: int, first: int,
                                m: int, layout: int = 1):
    """phase 2: scatter samples [first, first + m) of the stream (dfeat, samples: the WHOLE stream's tensors)"""
    (o, d), (ri, ts, te) = rays, samples
    check(_lib.load().ren_hashgrid_bwd_binned_scatter(ctypes.byref(grid), _ptr(grad_table, torch.float32), None, ctypes.byref(scene),
                                                      _ptr(o), _ptr(d), _ptr(ri), _ptr(ts), _ptr(te), n, layout,
                                                      _ptr(dfeat, torch.float32), first, m, _ptr(workspace), _stream()),
          "ren_hashgrid_bwd_binned_scatter")


def hashgrid_bwd_binned_finish(grid: GridDesc, grad_table, workspace, *, n: int, layout: int = 1):
    """phase 3: partition + accumulate + flush"""
    check(_lib.load().ren_hashgrid_bwd_binned_finish(ctypes.byref(grid), _ptr(grad_table, torch.float32), n, layout,
                                                     _ptr(workspace), _stream()), "ren_hashgrid_bwd_binned_finish")


# ------------------------------------------------------------------------------- fused MLPs
def mlp_fwd(mlp_params, C: int, feat, scene: SceneDesc, *, x_world=None, dirs=None, rays=None, samples=None,
            n: int, density_only: bool = False, save_base: bool = False, out=None, bf16: bool = False, act: int = 0):
    """bf16=True: `mlp_params` must be the bf16-rounded copy of the block (see ren_mlp_fwd_bf16)."""
    dev = feat.device
    o, d = rays if rays is not None else (None, None)
    ri, ts, te = samples if samples is not None else (None, None, None)
    if out is None:
        sigma = torch.empty(n, device=dev, dtype=torch.float32)
        rgb = None if density_only else torch.empty(n, C, device=dev, dtype=torch.float32)
        base = torch.empty(n_blocks32(n) * BASE_FLOATS_PER_BLOCK, device=dev, dtype=torch.float32) if save_base else None
    else:
        rgb, sigma, base = out
    fn = _lib.load().ren_mlp_fwd_bf16 if bf16 else _lib.load().ren_mlp_fwd
    check(fn(_ptr(mlp_params, torch.float32), C, int(act), _ptr(feat, torch.float32), ctypes.byref(scene),
             _ptr(x_world), _ptr(dirs), _ptr(o), _ptr(d), _ptr(ri), _ptr(ts), _ptr(te), n,
             1 if density_only else 0, _ptr(rgb), _ptr(sigma), _ptr(base), _stream()),
          "ren_mlp_fwd")
    return rgb, sigma, base


def mlp_fwd_save(mlp_params, C: int, feat, scene: SceneDesc, *, rays=None, samples=None, x_world=None, dirs=None,
                 n: int, bf16: bool = False, act: int = 0):
    """Training forward: also saves base outputs and the hidden activations (768 B/sample) for mlp_bwd_saved.
    -> rgb, sigma, base, acts"""
    dev = feat.device
    o, d = rays if rays is not None else (None, None)
    ri, ts, te = samples if samples is not None else (None, None, None)
    sigma = torch.empty(n, device=dev, dtype=torch.float32)
    rgb = torch.empty(n, C, device=dev, dtype=torch.float32)
    base = torch.empty(n_blocks32(n) * BASE_FLOATS_PER_BLOCK, device=dev, dtype=torch.float32)
    lib = _lib.load()
    acts = torch.empty(int(lib.ren_mlp_act_save_floats(n)), device=dev, dtype=torch.float32)
    check(lib.ren_mlp_fwd_save(_ptr(mlp_params, torch.float32), C, int(act), 1 if bf16 else 0, _ptr(feat, torch.float32),
                               ctypes.byref(scene), _ptr(x_world), _ptr(dirs), _ptr(o), _ptr(d), _ptr(ri), _ptr(ts),
                               _ptr(te), n, _ptr(rgb), _ptr(sigma), _ptr(base), _ptr(acts), _stream()), "ren_mlp_fwd_save")
    return rgb, sigma, base, acts


def mlp_bwd_saved(mlp_params, C: int, feat, base_out, acts, scene: SceneDesc, *, rays=None, samples=None, x_world=None,
                  dirs=None, n: int, rgb, d_rgb, d_sigma, grad_mlp_params, workspace, bf16: bool = False, act: int = 0):
    dev = feat.device
    o, d = rays if rays is not None else (None, None)
    ri, ts, te = samples if samples is not None else (None, None, None)
    d_base = torch.empty(n_blocks32(n) * BASE_FLOATS_PER_BLOCK, device=dev, dtype=torch.float32)
    dfeat = torch.empty(n_blocks32(n) * FRAG_FLOATS_PER_BLOCK, device=dev, dtype=torch.float32)
    check(_lib.load().ren_mlp_bwd_saved(_ptr(mlp_params, torch.float32), C, int(act), 1 if bf16 else 0, _ptr(feat), _ptr(base_out),
                                        _ptr(acts), ctypes.byref(scene), _ptr(x_world), _ptr(dirs), _ptr(o), _ptr(d),
                                        _ptr(ri), _ptr(ts), _ptr(te), n, _ptr(rgb), _ptr(d_rgb), _ptr(d_sigma),
                                        _ptr(d_base), _ptr(dfeat), _ptr(grad_mlp_params, torch.float32), _ptr(workspace),
                                        _stream()), "ren_mlp_bwd_saved")
    return dfeat


def mlp_fwd_x(mlp_params, C: int, mode: int, feat, scene: SceneDesc, *, rays=None, samples=None, x_world=None, dirs=None,
              n: int, density_only: bool = False, save: bool = False, out=None, share_cu: bool = False, save_acts: bool = True,
              act: int = 0, n_dev=None):
    """Split-bf16 matrix-core kernels (csrc/ren_mlp_x.hip).  mode 6: fp32 accuracy; mode 3: two bf16 pieces / three products
    (float32_matmul_precision "high"); mode 1: plain bf16 operands.
    -> rgb, sigma, base, acts (base/acts None unless save; acts None with save_acts=False: mlp_bwd_x then recomputes
    the hidden activations); out: preallocated (rgb, sigma, base, acts) views"""
    dev = feat.device
    o, d = rays if rays is not None else (None, None)
    ri, ts, te = samples if samples is not None else (None, None, None)
    lib = _lib.load()
    if out is not None:
        rgb, sigma, base, acts = out
    else:
        sigma = torch.empty(n, device=dev, dtype=torch.float32)
        rgb = None if density_only else torch.empty(n, C, device=dev, dtype=torch.float32)
        base = torch.empty(n_blocks32(n) * BASE_FLOATS_PER_BLOCK, device=dev, dtype=torch.float32) if save else None
        acts = torch.empty(int(lib.ren_mlp_act_save_floats(n)), device=dev, dtype=torch.float32) if (save and save_acts) else None
    check(lib.ren_mlp_fwd_x(_ptr(mlp_params, torch.float32), C, int(act), mode, _ptr(feat, torch.float32), ctypes.byref(scene),
                            _ptr(x_world), _ptr(dirs), _ptr(o), _ptr(d), _ptr(ri), _ptr(ts), _ptr(te), n,
                            (1 if density_only else 0) | (2 if share_cu else 0), _ptr(rgb), _ptr(sigma), _ptr(base),
                            _ptr(acts), _ptr(n_dev, torch.int64), _stream()), "ren_mlp_fwd_x")
    return rgb, sigma, base, acts


def mlp_act_save_floats(n: int) -> int:
    return int(_lib.load().ren_mlp_act_save_floats(n))


def mlp_bwd_x_workspace_floats(C: int) -> int:
    return int(_lib.load().ren_mlp_bwd_x_workspace_floats(C))


def mlp_bwd_x(mlp_params, C: int, mode: int, feat, base_out, acts, scene: SceneDesc, *, rays=None, samples=None,
              x_world=None, dirs=None, n: int, rgb, d_rgb, d_sigma, grad_mlp_params, workspace, dfeat=None, d_base=None,
              act: int = 0, grid_cus: int = 0, n_dev=None):
    """grid_cus: CUs the two persistent kernels occupy (0 = all; the chunked backward leaves some to the scatter)"""
    dev = feat.device
    o, d = rays if rays is not None else (None, None)
    ri, ts, te = samples if samples is not None else (None, None, None)
    if d_base is None:
        d_base = torch.empty(n_blocks32(n) * BASE_FLOATS_PER_BLOCK, device=dev, dtype=torch.float32)
    if dfeat is None:
        dfeat = torch.empty(n_blocks32(n) * FRAG_FLOATS_PER_BLOCK, device=dev, dtype=torch.float32)
    check(_lib.load().ren_mlp_bwd_x(_ptr(mlp_params, torch.float32), C, int(act), mode, _ptr(feat), _ptr(base_out), _ptr(acts),
                                    ctypes.byref(scene), _ptr(x_world), _ptr(dirs), _ptr(o), _ptr(d), _ptr(ri), _ptr(ts),
                                    _ptr(te), n, _ptr(rgb), _ptr(d_rgb), _ptr(d_sigma), _ptr(d_base), _ptr(dfeat),
                                    _ptr(grad_mlp_params, torch.float32), _ptr(workspace), int(grid_cus), _ptr(n_dev, torch.int64), _stream()),
          "ren_mlp_bwd_x")
    return dfeat


def mlp_bwd_workspace_floats(C: int) -> int:
    return int(_lib.load().ren_mlp_bwd_workspace_floats(C))


def mlp_bwd(mlp_params, C: int, feat, base_out, scene: SceneDesc, *, x_world=None, dirs=None, rays=None,
            samples=None, n: int, rgb, d_rgb, d_sigma, grad_mlp_params, workspace, d_base=None, dfeat=None,
            bf16: bool = False, act: int = 0):
    dev = feat.device
    o, d = rays if rays is not None else (None, None)
    ri, ts, te = samples if samples is not None else (None, None, None)
    if d_base is None:
        d_base = torch.empty(n_blocks32(n) * BASE_FLOATS_PER_BLOCK, device=dev, dtype=torch.float32)
    if dfeat is None:
        dfeat = torch.empty(n_blocks32(n) * FRAG_FLOATS_PER_BLOCK, device=dev, dtype=torch.float32)
    fn = _lib.load().ren_mlp_bwd_bf16 if bf16 else _lib.load().ren_mlp_bwd
    check(fn(_ptr(mlp_params, torch.float32), C, int(act), _ptr(feat), _ptr(base_out),
                                  ctypes.byref(scene), _ptr(x_world), _ptr(dirs), _ptr(o), _ptr(d), _ptr(ri),
                                  _ptr(ts), _ptr(te), n, _ptr(rgb), _ptr(d_rgb), _ptr(d_sigma), _ptr(d_base),
                                  _ptr(dfeat), _ptr(grad_mlp_params, torch.float32), _ptr(workspace), _stream()),
          "ren_mlp_bwd")
    return dfeat


# ------------------------------------------------------------------------------- compositing
def composite_fwd(offsets, counts, ts, te, sigmas, rgbs, C: int, bkgd, save: bool = True):
    n_rays = counts.shape[0]
    dev = ts.device
    colors = torch.empty(n_rays, C, device=dev, dtype=torch.float32) if rgbs is not None else None
    opac = torch.empty(n_rays, device=dev, dtype=torch.float32)
    depth = torch.empty(n_rays, device=dev, dtype=torch.float32)
    w = torch.empty(ts.shape[0], device=dev, dtype=torch.float32) if save else None
    T = torch.empty(ts.shape[0], device=dev, dtype=torch.float32) if save else None
    check(_lib.load().ren_composite_fwd(_ptr(offsets, torch.int64), _ptr(counts, torch.int32), n_rays, _ptr(ts),
                                        _ptr(te), _ptr(sigmas), _ptr(rgbs), C, _ptr(bkgd), _ptr(colors), _ptr(opac),
                                        _ptr(depth), _ptr(w), _ptr(T), _stream()), "ren_composite_fwd")
    return colors, opac, depth, w, T


def composite_bwd(offsets, counts, ts, te, sigmas, rgbs, C: int, bkgd, w, T, opac, g_colors, g_opac=None,
                  g_depth=None, want_bkgd: bool = False, g_weights=None):
    n_rays = counts.shape[0]
    dev = ts.device
    d_sig = torch.empty(ts.shape[0], device=dev, dtype=torch.float32)
    d_rgb = torch.empty(ts.shape[0], C, device=dev, dtype=torch.float32) if rgbs is not None else None
    d_bk = torch.empty(n_rays, C, device=dev, dtype=torch.float32) if want_bkgd else None
    check(_lib.load().ren_composite_bwd(_ptr(offsets), _ptr(counts), n_rays, _ptr(ts), _ptr(te), _ptr(sigmas),
                                        _ptr(rgbs), C, _ptr(bkgd), _ptr(w), _ptr(T), _ptr(opac),
                                        _ptr(g_colors, torch.float32), _ptr(g_opac), _ptr(g_depth), _ptr(g_weights),
                                        _ptr(d_sig), _ptr(d_rgb), _ptr(d_bk), _stream()), "ren_composite_bwd")
    return d_sig, d_rgb, d_bk


def column_sum(x: torch.Tensor):
    rows, C = x.shape
    out = torch.empty(C, device=x.device, dtype=torch.float32)
    scratch = torch.empty(512, device=x.device, dtype=torch.float32)
    check(_lib.load().ren_column_sum(_ptr(x, torch.float32), rows, C, _ptr(out), _ptr(scratch), _stream()),
          "ren_column_sum")
    return out


# ------------------------------------------------------------------------------- loss / optimiser
ERR_FN = {"l1": 0, "mse": 1, "mape": 2}


PARAM_WEIGHT_POWER = {None: 0, "mean_contrast_reciprocal": 1, "mean_contrast_reciprocal_sq": 2}


def event_prepare(batch, c_p: float, c_n: float, tau: float, *, with_grad_ts: bool = False, with_dtau: bool = False, ep=None):
    """a2-a4 in one launch -> dict(ts (2B,) f64 [start | end], target_diff f32, ts_grad, target_grad, dts_start,
    dts_end, dts_grad) -- the optional entries are None unless asked for.  ep: device-resident event parameters
    (event_params_refresh); c_p / c_n / tau are then ignored."""
    st, en = batch["start_ts"], batch["end_ts"]
    B, dev = st.shape[0], st.device
    ts = torch.empty(2 * B, device=dev, dtype=torch.float64)
    target = torch.empty(B, device=dev, dtype=torch.float32)
    ts_g = torch.empty(B, device=dev, dtype=torch.float64) if with_grad_ts else None
    target_g = torch.empty(B, device=dev, dtype=torch.float32) if with_grad_ts else None
    dts = torch.empty(3 if with_grad_ts else 2, B, device=dev, dtype=torch.float64) if with_dtau else None
    check(_lib.load().ren_event_prepare(
        _ptr(st, torch.int64), _ptr(en, torch.int64), _ptr(batch["num_pos"], torch.int64), _ptr(batch["num_neg"], torch.int64),
        _ptr(batch["u_ts_diff"], torch.float64), _ptr(batch["u_diff_start"], torch.float64),
        _ptr(batch["u_grad"], torch.float64) if with_grad_ts else None, B, _f(c_p), _f(c_n), ctypes.c_double(float(tau)),
        _ptr(ts), _ptr(ts[B:]) if B else None, _ptr(target), _ptr(ts_g), _ptr(target_g),
        _ptr(dts[0]) if with_dtau else None, _ptr(dts[1]) if with_dtau else None,
        _ptr(dts[2]) if (with_dtau and with_grad_ts) else None, _ptr(ep, torch.float64), _stream()), "ren_event_prepare")
    return dict(ts=ts, target_diff=target, ts_grad=ts_g, target_grad=target_g,
                dts_start=dts[0] if with_dtau else None, dts_end=dts[1] if with_dtau else None,
                dts_pair=dts[:2].reshape(-1) if with_dtau else None,          # [d ts_start/d tau | d ts_end/d tau], (2B,) view
                dts_grad=dts[2] if (with_dtau and with_grad_ts) else None)


def event_param_grad(kind: str, err_fn: str, param_weight, pred, valid, batch, c_p: float, c_n: float, raw_ratio: float,
                     tau: float, weight: float, ct_grad=None, tau_grad=None, ep=None):
    """closed-form d(loss term)/d(raw ratio) += ct_grad[0], direct d(loss term)/d(tau) += tau_grad[0] (f64)"""
    check(_lib.load().ren_event_param_grad(
        {"diff": 0, "grad": 1}[kind], ERR_FN[err_fn], PARAM_WEIGHT_POWER[param_weight], _ptr(pred, torch.float32), _ptr(valid),
        _ptr(batch["start_ts"], torch.int64), _ptr(batch["end_ts"], torch.int64), _ptr(batch["num_pos"], torch.int64),
        _ptr(batch["num_neg"], torch.int64), _ptr(batch["u_ts_diff"], torch.float64), pred.shape[0], _f(c_p), _f(c_n),
        _f(raw_ratio), ctypes.c_double(float(tau)), _f(weight), _ptr(ct_grad, torch.float32), _ptr(tau_grad, torch.float64),
        _ptr(ep, torch.float64), _stream()), "ren_event_param_grad")


EP_CP, EP_CN, EP_RAW, EP_TAU, EP_INV_C, EP_INV_C2, EP_TAU_RAW = range(7)     # layout of the device-resident event parameters


def event_params_refresh(ct_raw, c_n: float, tau_raw, tau_max: float, ep):
    """ep (f64[8], device) <- C_p, C_n, raw ratio, tau, 1/C, 1/C^2, clamped raw tau from the raw (trainable) forms"""
    check(_lib.load().ren_event_params_refresh(_ptr(ct_raw, torch.float32), _f(c_n), _ptr(tau_raw, torch.float64),
                                               ctypes.c_double(float(tau_max)), _ptr(ep, torch.float64), _stream()),
          "ren_event_params_refresh")


def tau_adam_step(tau_raw, tau_grad, state, tau_max: float, *, lr: float, betas=(0.9, 0.999), eps: float = 1e-8, step: int,
                  grad_scale: float = 1.0):
    check(_lib.load().ren_tau_adam_step(_ptr(tau_raw, torch.float64), _ptr(tau_grad, torch.float64), _ptr(state, torch.float64),
                                        ctypes.c_double(float(tau_max)), ctypes.c_double(float(lr)), ctypes.c_double(betas[0]),
                                        ctypes.c_double(betas[1]), ctypes.c_double(eps), int(step), ctypes.c_double(grad_scale),
                                        _stream()), "ren_tau_adam_step")


def event_loss_fwd(i_start, i_end, target, valid, err_fn: str):
    if err_fn not in ERR_FN:
        raise NotImplementedError(err_fn)
    loss_sum = torch.empty(2, device=i_start.device, dtype=torch.float32)
    check(_lib.load().ren_event_loss_fwd(_ptr(i_start, torch.float32), _ptr(i_end, torch.float32),
                                         _ptr(target, torch.float32), _ptr(valid), i_start.shape[0], ERR_FN[err_fn],
                                         _ptr(loss_sum), _stream()), "ren_event_loss_fwd")
    return loss_sum


def event_loss_bwd(i_start, i_end, target, valid, err_fn: str, scale: float, loss_sum):
    g_s = torch.empty_like(i_start)
    g_e = torch.empty_like(i_end)
    check(_lib.load().ren_event_loss_bwd(_ptr(i_start), _ptr(i_end), _ptr(target), _ptr(valid), i_start.shape[0],
                                         ERR_FN[err_fn], _f(scale), _ptr(loss_sum), _ptr(g_s), _ptr(g_e), _stream()),
          "ren_event_loss_bwd")
    return g_s, g_e


def event_diff_loss(colors, opac, channel_idx, target, err_fn: str, scale: float, min_intensity: float, use_validity: bool,
                    want_pred: bool = False, scale_dev=None):
    """Loss + its gradient straight from the batched start / end render: colors (2B, C), opac (2B,) ->
    dict(loss (device scalar), loss_sum, g_colors (2B, C), intensity (2B,), pred (B,) | None, valid (B,) uint8 | None).
    Two launches (ren_event_diff_loss_fwd / _bwd) instead of the intensity epilogue, mask, loss, scatter and scalar glue."""
    if err_fn not in ERR_FN:
        raise NotImplementedError(err_fn)
    dev = colors.device
    B, C = colors.shape[0] // 2, colors.shape[1]
    lib = _lib.load()
    loss_sum = torch.empty(2, device=dev, dtype=torch.float32)
    args = (_ptr(colors, torch.float32), _ptr(opac, torch.float32), _ptr(channel_idx, torch.uint8), C, _f(min_intensity),
            _ptr(target, torch.float32), 1 if use_validity else 0, B, ERR_FN[err_fn])
    check(lib.ren_event_diff_loss_fwd(*args, _ptr(loss_sum), _stream()), "ren_event_diff_loss_fwd")
    g_colors = torch.empty_like(colors)
    inten = torch.empty(2 * B, device=dev, dtype=torch.float32)
    pred = torch.empty(B, device=dev, dtype=torch.float32) if want_pred else None
    valid = torch.empty(B, device=dev, dtype=torch.uint8) if use_validity else None
    loss = torch.empty((), device=dev, dtype=torch.float32)
    check(lib.ren_event_diff_loss_bwd(*args, _f(scale), _ptr(scale_dev, torch.float64), _ptr(loss_sum), _ptr(g_colors), _ptr(inten),
                                      _ptr(pred), _ptr(valid), _ptr(loss), _stream()), "ren_event_diff_loss_bwd")
    return dict(loss=loss, loss_sum=loss_sum, g_colors=g_colors, intensity=inten, pred=pred, valid=valid)


def bkgd_param_fwd(raw: torch.Tensor, C: int) -> torch.Tensor:
    """softplus of the background parameter (models/nerf.py:81-88), one tiny launch"""
    out = torch.empty(C, device=raw.device, dtype=torch.float32)
    check(_lib.load().ren_bkgd_param_fwd(_ptr(raw, torch.float32), C, _ptr(out), _stream()), "ren_bkgd_param_fwd")
    return out


def bkgd_param_grad(d_bkgd_per_ray: torch.Tensor, raw: torch.Tensor, grad_raw: torch.Tensor):
    """grad_raw[:C] += sigmoid(raw[:C]) * column sums of the per-ray background gradient"""
    rows, C = d_bkgd_per_ray.shape
    scratch = torch.empty(512, device=raw.device, dtype=torch.float32)
    check(_lib.load().ren_bkgd_param_grad(_ptr(d_bkgd_per_ray, torch.float32), rows, C, _ptr(raw, torch.float32),
                                          _ptr(grad_raw, torch.float32), _ptr(scratch), _stream()), "ren_bkgd_param_grad")


def adam_step(p, g, m, v, *, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, step: int, grad_scale=1.0,
              zero_grad=True):
    check(_lib.load().ren_adam_step(_ptr(p, torch.float32), _ptr(g, torch.float32), _ptr(m, torch.float32),
                                    _ptr(v, torch.float32), p.numel(), _f(lr), _f(betas[0]), _f(betas[1]), _f(eps),
                                    _f(weight_decay), int(step), _f(grad_scale), 1 if zero_grad else 0, _stream()),
          "ren_adam_step")


# ---- optimiser state on the device (ABI 25): what a captured step (engine.Trainer._graph_step) needs -- include/ren_amd.h
HY_STEP, HY_SKIP, HY_BC1, HY_BC2, HY_TAU_STEP, HY_TAU_BC1, HY_TAU_BC2 = range(7)


def step_tick(hyper, betas=(0.9, 0.999), stats=(), tick_tau: bool = False):
    """advance the device-side Adam step(s) -- or raise the sticky skip word when an overflow word of `stats` (up to two
    int64[4] blocks of scan_guard) is set.  Once per optimiser step, before adam_step_dev / tau_adam_step_dev."""
    st = [s for s in stats if s is not None]
    if len(st) > 2:
        raise ValueError("step_tick: at most two renders' stats")
    st += [None] * (2 - len(st))
    check(_lib.load().ren_step_tick(_ptr(hyper, torch.float64), ctypes.c_double(betas[0]), ctypes.c_double(betas[1]),
                                    _ptr(st[0], torch.int64), _ptr(st[1], torch.int64), 1 if tick_tau else 0, _stream()),
          "ren_step_tick")


def adam_step_dev(p, g, m, v, hyper, *, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_scale=1.0, zero_grad=True):
    check(_lib.load().ren_adam_step_dev(_ptr(p, torch.float32), _ptr(g, torch.float32), _ptr(m, torch.float32),
                                        _ptr(v, torch.float32), p.numel(), _f(lr), _f(betas[0]), _f(betas[1]), _f(eps),
                                        _f(weight_decay), _ptr(hyper, torch.float64), _f(grad_scale), 1 if zero_grad else 0,
                                        _stream()), "ren_adam_step_dev")


def tau_adam_step_dev(tau_raw, tau_grad, state, tau_max: float, hyper, *, lr: float, betas=(0.9, 0.999), eps: float = 1e-8,
                      grad_scale: float = 1.0):
    check(_lib.load().ren_tau_adam_step_dev(_ptr(tau_raw, torch.float64), _ptr(tau_grad, torch.float64),
                                            _ptr(state, torch.float64), ctypes.c_double(float(tau_max)),
                                            ctypes.c_double(float(lr)), ctypes.c_double(betas[0]), ctypes.c_double(betas[1]),
                                            ctypes.c_double(eps), _ptr(hyper, torch.float64), ctypes.c_double(grad_scale),
                                            _stream()), "ren_tau_adam_step_dev")


# ------------------------------------------------------------------------------- occupancy grid
def occgrid_cell_points(indices, jitter, roi, res, ct):
    m = indices.shape[0]
    x = torch.empty(m, 3, device=indices.device, dtype=torch.float32)
    valid = torch.empty(m, device=indices.device, dtype=torch.uint8)
    roi_c = (ctypes.c_float * 6)(*[float(v) for v in roi])
    res_c = (ctypes.c_int32 * 3)(*[int(v) for v in res])
    check(_lib.load().ren_occgrid_cell_points(_ptr(indices, torch.int64), _ptr(jitter, torch.float32), m, roi_c, res_c,
                                              ct, _ptr(x), _ptr(valid), _stream()), "ren_occgrid_cell_points")
    return x, valid


def occgrid_ema(occs, indices, valid, sigma, step_sizes, step_size: float, decay: float, scratch=None):
    """scratch (one float per cell): deterministic handling of duplicate cells in `indices` (ren_occgrid_ema_unique)"""
    if scratch is not None:
        check(_lib.load().ren_occgrid_ema_unique(_ptr(occs, torch.float32), _ptr(indices, torch.int64), _ptr(valid),
                                                 _ptr(sigma, torch.float32), _ptr(step_sizes), _f(step_size),
                                                 indices.shape[0], _f(decay), _ptr(scratch, torch.float32), _stream()),
              "ren_occgrid_ema_unique")
        return
    check(_lib.load().ren_occgrid_ema(_ptr(occs, torch.float32), _ptr(indices, torch.int64), _ptr(valid),
                                      _ptr(sigma, torch.float32), _ptr(step_sizes), _f(step_size), indices.shape[0],
                                      _f(decay), _stream()), "ren_occgrid_ema")


def occgrid_binarize(occs, occ_thre: float, binary, scratch):
    check(_lib.load().ren_occgrid_binarize(_ptr(occs, torch.float32), occs.numel(), _f(occ_thre),
                                           _ptr(binary, (torch.uint8, torch.bool)), _ptr(scratch, torch.float32),
                                           _stream()), "ren_occgrid_binarize")


# ------------------------------------------------------------------------------- per-kernel timing
# HIP-event timing of individual launches on the stream they are enqueued on (torch's current
# stream, which is the stream handed to the C ABI).  Used by bench.py for the roofline numbers;
# disabled (zero overhead) otherwise.
_PROFILE = None
_TIMED = ("ray_aabb_intersect", "ray_march_count", "ray_march_write", "exclusive_scan", "visibility",
          "compact_samples", "compact_features", "hashgrid_fwd", "hashgrid_bwd", "hashgrid_bwd_binned", "hashgrid_bwd_binned_begin",
          "hashgrid_bwd_binned_scatter", "hashgrid_bwd_binned_finish", "mlp_fwd", "mlp_bwd", "mlp_fwd_save",
          "mlp_bwd_saved", "mlp_fwd_x", "mlp_bwd_x", "composite_fwd",
          "composite_bwd", "column_sum", "event_loss_fwd", "event_loss_bwd", "adam_step", "trajectory", "raygen")


def profile_start():
    global _PROFILE
    _PROFILE = {}


def profile_stop():
    """-> {kernel family: (launches, total milliseconds)}; synchronises once."""
    global _PROFILE
    prof, _PROFILE = _PROFILE, None
    torch.cuda.synchronize()
    return {k: (len(v), sum(a.elapsed_time(b) for a, b in v)) for k, v in (prof or {}).items()}


def _wrap(name, fn):
    def timed(*a, **kw):
        if _PROFILE is None or torch.cuda.is_current_stream_capturing():    # (events inside a capture become graph nodes)
            return fn(*a, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(*a, **kw)
        e1.record()
        _PROFILE.setdefault(name, []).append((e0, e1))
        return out
    timed.__name__ = fn.__name__
    timed.__doc__ = fn.__doc__
    return timed


for _n in _TIMED:
    globals()[_n] = _wrap(_n, globals()[_n])
