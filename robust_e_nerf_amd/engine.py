"""Fused fast path: the whole render / training step as explicit HIP launches (no autograd).

This is the "one level up" seam of SURVEY.md 8b: ``Renderer.forward`` replaces
``NeRF.forward`` -> ``render_image`` -> {``ray_marching``, ``rendering``} -> field
(robust_e_nerf/models/nerf.py:230-286, external/utils.py:38-140) and ``Trainer.step`` replaces
``RobustENeRF.training_step`` + backward + optimiser step
(robust_e_nerf/models/robust_e_nerf.py:301-517, 782-813) for the log-intensity-difference loss.
Gradients are written straight into one flat buffer ([hash table | MLPs]) so the data-parallel
all-reduce is a single RCCL call and Adam a single fused pass.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field as dc_field
from typing import Dict, Optional, Sequence, Tuple

import contextlib
import dataclasses
import os
import threading
import torch

from . import ops


@dataclass
class RenderCfg:
    """model.nerf.* of configs/train/*.yaml that shapes the hot path."""
    aabb: Tuple[float, ...] = (-1.5, -1.5, -1.5, 1.5, 1.5, 1.5)
    contraction_type: int = ops.AABB
    occ_res: Tuple[int, int, int] = (128, 128, 128)
    near_plane: Optional[float] = None
    far_plane: Optional[float] = None
    render_step_size: float = math.sqrt(3) * 3.0 / 1024         # robust_e_nerf.py:220-226 ("auto")
    cone_angle: float = 0.0
    early_stop_eps: float = 1e-4
    alpha_thre: float = 0.0
    min_modeled_intensity: float = 1e-3
    opacity_eps: float = 1e-10
    radiance_dim: int = 1
    sampler: str = "occgrid"                                     # "occgrid" | "uniform"
    n_uniform: int = 128
    occ_thre: float = 1e-2
    ema_decay: float = 0.95
    warmup_steps: int = 256
    occ_n: int = 16
    occ_seed: int = 20230                # seed of the grid-refresh stream (the same on every rank)
    binned_scatter: bool = True        # LDS-binned hash-grid backward (False: per-update global atomics)
    mlp_kernels: str = "x"             # "x": split-bf16 matrix-core kernels at fp32 accuracy (csrc/ren_mlp_x.hip);
                                       # "f32": exact f32-MFMA kernels (csrc/ren_mlp.hip)
    save_activations: Optional[bool] = None   # training forward stores the hidden activations (768 B/sample) instead of
                                       # recomputing them in the backward.  Off since round 2: on the bf16 matrix cores the
                                       # recompute (96 MFMAs + 96 softplus per 32 samples) is cheaper than 13 GB of HBM traffic;
                                       # None = auto: recompute with the "x" kernels, save with the exact-f32 kernels
    march_cache: int = 512             # intervals per ray kept between the two marching passes (0: march twice)
    fwd_chunks: int = 8                # > 1: hash encoding and MLP of alternate sample chunks on two HIP streams
    bwd_chunks: int = 1                # > 1: MLP backward of chunk k + 1 beside the binned scatter of chunk k (two streams), ONE accumulate
    bwd_mlp_cus: int = 192             # CUs the MLP backward kernels occupy while a scatter runs beside them
    dp_overlap: bool = True            # data parallel: all-reduce the fine levels' table gradient beside the coarse levels' scatter
    dp_split_level: int = 8            # levels >= this one go first (8 x 4 MiB of the 50 MB buffer)
    dp_compress: Optional[str] = None  # "bf16": parameter gradients cross the links as bfloat16 (strong scaling), aux block stays fp32
    device_counts: Optional[bool] = None   # occupancy sampler: the sample counts of a training render stay on the device and the
                                       # rest of the step is enqueued over arrays of an estimated capacity (the reference reads both
                                       # counts back in the middle of every render: external/utils.py:106-119, models/nerf.py:279-286;
                                       # SURVEY 7.2 H4).  None = auto: on for the Trainer's renders on one GPU (Trainer.device_counts_ok)
    mlp_bf16: bool = False             # BASELINE configs[2]: bf16 MLP (rounded linear inputs/weights, fp32 accumulate), fp32 composite
    mlp_precision: str = "highest"     # the YAMLs' float32_matmul_precision (scripts/run.py:34-35, torch.set_float32_matmul_precision):
                                       # "highest": every MLP product to fp32 round-off (six bf16 products of three pieces);
                                       # "high": "each float32 as the sum of two bfloat16" -- three bf16 products, ~16 significant
                                       # bits per product, half of the matrix-pipe time (the "x" kernels of arch ngp and the fused
                                       # field of arch mlp; anything else stays at "highest"); "medium": bf16 operands = mlp_bf16
    # activation alternatives of the YAML (model.nerf.ngp.mlp_base / mlp_head: models/nerf.py:8-29).  Anything but the
    # shipped values runs on the exact-f32 MLP kernels (mlp_kernels is switched to "f32"): arch ngp only
    base_hidden_activation: str = "softplus"          # softplus (beta 100) | relu
    density_activation: str = "shifted_trunc_exp"     # shifted_trunc_exp | softplus | shifted_softplus
    head_hidden_activation: str = "softplus"          # softplus (beta 100) | relu
    radiance_activation: str = "softplus"             # softplus | sigmoid


class NGPField:
    """Device-resident Instant-NGP parameters: one flat float32 buffer [hash table | MLP block (| weight-norm g)]
    plus an identically shaped gradient buffer (robust_e_nerf/external/ngp.py:166-205).

    weight_norm = (mlp_base, mlp_head) (ngp.py:207-228: torch.nn.utils.weight_norm on every Linear of a flagged MLP): the
    trainable block then holds v in the weight slots and the g of every flagged row behind it; `mlp` / `g_mlp` -- what the
    field kernels read and accumulate into -- become separate buffers with the effective weights W = g v / ||v|| and their
    gradient, kept in step by refresh() (after every parameter change) and fold_grads() (before the optimiser)."""

    WN_WEIGHTS = ("base.w0", "base.wo", "head.w0", "head.w1", "head.wo")

    def __init__(self, device, radiance_dim: int = 1, pos_encoding: Optional[dict] = None,
                 weight_norm: Tuple[bool, bool] = (False, False)):
        pe = dict(n_levels=16, n_features_per_level=2, log2_hashmap_size=19, base_resolution=16,
                  per_level_scale=1.4472692012786865, otype="HashGrid")
        if pos_encoding:
            pe.update({k: v for k, v in pos_encoding.items() if k in pe})
        self.grid, self.n_table = ops.make_grid_desc(**pe)
        if self.grid.n_levels != 16:
            raise NotImplementedError("the fused MLP kernels are built for 16 levels x 2 features")
        self.C = radiance_dim
        self.n_mlp = ops.mlp_param_count(radiance_dim)
        self.weight_norm = (bool(weight_norm[0]), bool(weight_norm[1]))
        sl = ops.mlp_slices(radiance_dim)
        self.wn_layers, layers, n_g = [], [], 0                         # flagged weights, (offset, rows, cols, first g)
        for k in self.WN_WEIGHTS:
            if self.weight_norm[0 if k.startswith("base") else 1]:
                off, (rows, cols) = sl[k]
                self.wn_layers.append((k, n_g, rows))
                layers.append((off, rows, cols, n_g))
                n_g += rows
        self.n_wn_g = n_g
        n = self.n_table + self.n_mlp + n_g
        self.n_params = n
        n_pad = (n + 3) // 4 * 4
        self.flat = torch.zeros(n_pad, device=device, dtype=torch.float32)
        # gradient buffer with the data-parallel tail: [table | MLP | pad | aux] is ONE all-reduce (parallel.GradSync)
        from .parallel import AUX_FLOATS
        self.grad_all = torch.zeros(n_pad + AUX_FLOATS, device=device, dtype=torch.float32)
        self.grad = self.grad_all[:n_pad]
        self.aux = self.grad_all[n_pad:]
        self.table = self.flat[: self.n_table]
        self.g_table = self.grad[: self.n_table]
        m0, m1 = self.n_table, self.n_table + self.n_mlp
        if n_g == 0:
            self.mlp, self.g_mlp = self.flat[m0: m1], self.grad[m0: m1]
        else:
            self.mlp_raw, self.wn_g = self.flat[m0: m1], self.flat[m1: n]
            self.g_mlp_raw, self.g_wn_g = self.grad[m0: m1], self.grad[m1: n]
            self.mlp = torch.zeros(self.n_mlp, device=device, dtype=torch.float32)
            self.g_mlp = torch.zeros(self.n_mlp, device=device, dtype=torch.float32)
            self._wn_table = ops.weight_norm_layers(layers)

    def refresh(self):
        """effective MLP block from (v, g): after load() and after every optimiser step"""
        if self.n_wn_g:
            ops.weight_norm_fwd(self.mlp_raw, self.wn_g, self._wn_table, self.mlp)

    def fold_grads(self, zero: bool = True):
        """gradient w.r.t. the effective block -> gradient w.r.t. (v, g, biases) in the optimiser's buffer (linear in the
        former, so micro-batches accumulate in g_mlp and are folded once)"""
        if self.n_wn_g:
            ops.weight_norm_bwd(self.mlp_raw, self.wn_g, self.g_mlp, self._wn_table, self.g_mlp_raw, self.g_wn_g, zero_d_eff=zero)

    def load(self, p: Dict[str, torch.Tensor]):
        """p: {"hash", "base.w0", ...} (torch nn.Linear layout, the oracle's / reference's names); a weight-normalised layer
        is given as "<k>_g" (rows, 1) + "<k>_v", or as a plain "<k>" (then v = W, g = ||W||_row as weight_norm() starts)."""
        dev = self.flat.device
        self.table.copy_(p["hash"].to(dev, torch.float32).reshape(-1))
        raw = self.mlp_raw if self.n_wn_g else self.mlp
        g_of = {k: (g0, rows) for k, g0, rows in self.wn_layers}
        for k, (off, shape) in ops.mlp_slices(self.C).items():
            if k in g_of:
                g0, rows = g_of[k]
                v = (p[k + "_v"] if k + "_v" in p else p[k]).to(dev, torch.float32)
                g = p[k + "_g"].to(dev, torch.float32).reshape(-1) if k + "_g" in p else v.norm(dim=1)
                raw[off: off + math.prod(shape)].copy_(v.reshape(-1))
                self.wn_g[g0: g0 + rows].copy_(g)
            else:
                if k + "_v" in p:
                    raise ValueError(f"{k}: weight_g / weight_v given for an MLP built without weight_norm")
                raw[off: off + math.prod(shape)].copy_(p[k].to(dev, torch.float32).reshape(-1))
        self.refresh()

    def mlp_views(self, grad: bool = False) -> Dict[str, torch.Tensor]:
        """effective parameters (nn.Linear names) -- or the gradient w.r.t. them"""
        buf = self.g_mlp if grad else self.mlp
        return {k: buf[off: off + math.prod(shape)].view(shape) for k, (off, shape) in ops.mlp_slices(self.C).items()}

    def trainable_views(self, grad: bool = False) -> Dict[str, torch.Tensor]:
        """what the optimiser holds: as mlp_views(), with "<k>_g" (rows, 1) / "<k>_v" in place of a weight-normalised "<k>"
        (gradients: after fold_grads())"""
        if not self.n_wn_g:
            return self.mlp_views(grad)
        raw, gg = (self.g_mlp_raw, self.g_wn_g) if grad else (self.mlp_raw, self.wn_g)
        g_of = {k: (g0, rows) for k, g0, rows in self.wn_layers}
        out = {}
        for k, (off, shape) in ops.mlp_slices(self.C).items():
            view = raw[off: off + math.prod(shape)].view(shape)
            if k in g_of:
                g0, rows = g_of[k]
                out[k + "_g"], out[k + "_v"] = gg[g0: g0 + rows].view(rows, 1), view
            else:
                out[k] = view
        return out


@dataclass
class Packed:
    ray_indices: torch.Tensor
    t_starts: torch.Tensor
    t_ends: torch.Tensor
    offsets: torch.Tensor
    counts: torch.Tensor
    n: int
    n_marched: int = 0
    feat: Optional[torch.Tensor] = None         # hash features of these samples, when the density pre-pass made them
    n_dev: Optional[torch.Tensor] = None        # device-side counts: `n` is then the CAPACITY of the arrays (it sizes the launches),
                                                # the number of samples that exist is n_dev[0] (int64 on the device)
    log: Optional["CountLog"] = None            # where the host will find the counts (and whether they fitted)


def pack_batch(batch: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """The event batch (B.1 of SURVEY: position, start_ts, end_ts, num_pos, num_neg, u_* ...) with every field a view into ONE
    device buffer, returned under "_pack" beside them: a captured step then takes its inputs with one copy launch instead of one
    per field (eight 5-us launches are 5 % of a 1 ms step).  Same keys, shapes and dtypes; anything that is not a tensor is kept."""
    items = [(k, v) for k, v in batch.items() if isinstance(v, torch.Tensor) and k != "_pack"]
    offs, total = [], 0
    for _, v in items:
        offs.append(total)
        total += (v.numel() * v.element_size() + 15) // 16 * 16
    dev = items[0][1].device
    buf = torch.empty(max(total, 16), dtype=torch.uint8, device=dev)
    out = {k: v for k, v in batch.items() if not isinstance(v, torch.Tensor)}
    for (k, v), o in zip(items, offs):
        view = buf[o: o + v.numel() * v.element_size()].view(v.dtype).view(v.shape)
        view.copy_(v)
        out[k] = view
    out["_pack"] = buf
    return out


def _adopt(v, stream):
    """tensors allocated on another stream's pool that `stream` is about to use (after waiting for their producer): tell the
    caching allocator, so that their memory is not handed out again before `stream` is done with them"""
    if isinstance(v, torch.Tensor):
        if v.is_cuda:
            v.record_stream(stream)
    elif isinstance(v, dict):
        for x in v.values():
            _adopt(x, stream)
    elif isinstance(v, (tuple, list)):
        for x in v:
            _adopt(x, stream)
    elif isinstance(v, Packed):
        _adopt((v.ray_indices, v.t_starts, v.t_ends, v.offsets, v.counts, v.feat, v.n_dev), stream)


class CountLog:
    """The two sample counts of one render sampled with device-side counts -- marched, kept -- and whether they fitted the
    capacities the arrays were given: copied to pinned memory behind the sampling kernels.  wait() blocks the host until
    THOSE kernels are done (an event), not until the queue is empty.
    polled=True (a render inside a captured step, Trainer._graph_step): the copy is a node of the graph and no event can be
    waited for; the host arms the pinned words with -1 before every replay (arm()) and wait() spins until the copy has
    landed -- pinned host memory is coherent, and each of the four words is one aligned 8-byte store."""

    def __init__(self, n_rays: int, caps, stats: torch.Tensor, pinned: torch.Tensor, polled: bool = False):
        self.n_rays, self.caps, self.stats = n_rays, caps, stats
        pinned.copy_(stats, non_blocking=True)
        self._pinned = pinned
        self.ev = None
        if not polled:
            self.ev = torch.cuda.Event()
            self.ev.record()
        self.values = None                       # (marched, overflowed, kept, overflowed)

    def arm(self):
        self._pinned.fill_(-1)
        self.values = None

    def wait(self):
        if self.values is None:
            if self.ev is not None:
                self.ev.synchronize()
                self.values = tuple(int(v) for v in self._pinned.tolist())
            else:
                import time
                t0 = time.monotonic()
                while True:
                    v = self._pinned.tolist()
                    if min(v) >= 0:
                        break
                    if time.monotonic() - t0 > 120.0:         # (a replay that died: fail loudly instead of spinning for ever)
                        torch.cuda.synchronize()
                        raise RuntimeError("captured step: the sample counts of a replay never arrived")
                self.values = tuple(int(x) for x in v)
        return self.values

    @property
    def overflowed(self) -> bool:
        v = self.wait()
        return bool(v[1] or v[3])


_PINNED = threading.local()             # per host thread: two renderers may read counts from two threads at once


def _host_int(t: torch.Tensor) -> int:
    """one device integer on the host.  On the default stream: .item().  On a side stream: through a pinned buffer and an
    event, so that the host waits for THAT stream only (a pageable read-back waits for the whole device queue on ROCm:
    tools/sync_probe.py) -- what lets Trainer.grad_loss_forward_backward(early=True) place its samples while the previous
    backward is still running."""
    cur = torch.cuda.current_stream(t.device)
    if cur == torch.cuda.default_stream(t.device):
        return int(t.item())
    bufs = _PINNED.__dict__.setdefault("bufs", {})
    buf = bufs.get(t.dtype)
    if buf is None:
        buf = bufs[t.dtype] = torch.empty(1, dtype=t.dtype).pin_memory()
    buf.copy_(t.reshape(1), non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(cur)
    ev.synchronize()
    return int(buf[0])


class Renderer:
    def __init__(self, fld: NGPField, cfg: RenderCfg):
        self.field = fld
        self.cfg = cfg = dataclasses.replace(cfg)   # a private copy: the overrides below must not leak into the caller's object
        if os.environ.get("REN_MARCH_CACHE"):       # A/B switch for scripts that build the renderer themselves
            cfg.march_cache = int(os.environ["REN_MARCH_CACHE"])
        self.scene = ops.make_scene_desc(cfg.aabb, cfg.contraction_type)
        dev = fld.flat.device
        cells = cfg.occ_res[0] * cfg.occ_res[1] * cfg.occ_res[2]
        # nerfacc.OccupancyGrid buffers (models/nerf.py:98-102): zero until the first update
        self.occs = torch.zeros(cells, device=dev, dtype=torch.float32)
        self.binary = torch.zeros(cells, device=dev, dtype=torch.uint8)
        self._scratch = torch.zeros(4, device=dev, dtype=torch.float32)
        self._ws = torch.empty(max(ops.mlp_bwd_workspace_floats(fld.C), ops.mlp_bwd_x_workspace_floats(fld.C)),
                               device=dev, dtype=torch.float32)
        self._bin_ws, self._bin_ws_small = None, 0
        self._occ_scratch = None
        self._occ_gen = None
        self.grad_sync = None                       # parallel.GradSync of the Trainer under data parallelism
        self._fwd_streams = None
        self._bwd_stream = None
        self._reuse_prepass_feat = True             # the differentiable pass reuses the pre-pass hash features
        self._march_div_ok = None                   # ops.march_div_check(cfg.aabb), at the first occupancy-sampled render
        # device-side sample counts (RenderCfg.device_counts): samples per ray (marched, kept) the capacities are derived from,
        # learnt from the renders themselves -- the first one reads its counts on the host -- and a ring of pinned buffers
        self._spr = None
        self._count_ring, self._count_ring_at = None, 0
        self._polled_logs = None                    # a list while a step is being captured (Trainer._capture): its renders' CountLogs
        self._polled_pinned = None                  # ... and the pinned words allocated for them before the capture began
        self.dp_early_enabled = True                # the Trainer clears it when a loss pass may have to be repeated (device-side counts)
        self.bwd_side_cus = 0                       # CUs the persistent MLP backward kernels leave free (Trainer: a side stream is at work)
        self._act_code = ops.activation_code(cfg.base_hidden_activation, cfg.density_activation, cfg.head_hidden_activation,
                                             cfg.radiance_activation)
        if cfg.mlp_precision not in ("highest", "high", "medium"):
            raise ValueError(f"mlp_precision {cfg.mlp_precision!r}: highest | high | medium")
        if cfg.mlp_precision == "medium":
            cfg.mlp_bf16 = True
        if self._act_code != 0 and isinstance(fld, NGPField):
            cfg.mlp_kernels = "f32"                 # the bf16-matrix-core kernels implement the shipped activations only
        # (arch mlp: vanilla.VanillaRenderer switches to its per-layer launches, in any matrix-core mode)
        grid = getattr(fld, "grid", None)
        if grid is not None and not ops.binned_supported(grid):
            cfg.binned_scatter = False              # DenseGrid levels beyond 64 bins: the per-update atomic scatter
        # the activation set travels with every launch as an argument (`act=`; ABI 24): renderers with different sets can
        # launch from different host threads / streams of one process

    # ---- sampling (K1-K3): ray/AABB, two-pass march, no-grad density pre-pass + visibility --------
    def sample_begin(self, o, d, jitter: Optional[torch.Tensor], training: bool, device_counts=False) -> dict:
        """sample() up to (not including) the host read of the sample count: ray/AABB test, count pass, scan.  None of it
        depends on the field parameters, so Trainer.prefetch can run it for the NEXT step on a side stream."""
        return self._scan_counts(self._march_count(o, d, jitter, training), device_counts)

    def sample_begin_merged(self, rays, training: bool, device_counts=False):
        """sample_begin() of several renders with ONE ray/box test and ONE march count pass over all their rays (the march is a
        latency-bound chain of ~1 000 dependent occupancy reads per ray: 0.2 ms however few rays there are -- two renders' passes
        one after the other cost twice that, one pass over both costs it once), then a scan (and guard) per render: each
        render keeps its own packed sample stream.  rays: [(o, d, jitter), ...] -> one sample_begin() state per render, whose
        arrays are row ranges of the merged ones.  The same counts, bit for bit: a ray's march does not depend on its
        neighbours (tests/test_gpu_parity.py::test_ray_marching_bit_exact holds every speculative width to the sequential one)."""
        jit = None if any(r[2] is None for r in rays) else torch.cat([r[2] for r in rays])
        st = self._march_count(torch.cat([r[0] for r in rays]), torch.cat([r[1] for r in rays]), jit, training)
        out, a = [], 0
        for o, _, _ in rays:
            b = a + o.shape[0]
            args = tuple(v[a:b] if (k < 5 and v is not None) else v for k, v in enumerate(st["args"]))
            out.append(self._scan_counts(dict(args=args, cache=None if st["cache"] is None else st["cache"][a:b],
                                              counts=st["counts"][a:b], mode=st["mode"]), device_counts))
            a = b
        return out

    def _march_count(self, o, d, jitter, training: bool) -> dict:
        c = self.cfg
        scene_aabb = c.aabb if c.contraction_type == ops.AABB else None           # nerf.py:248-251
        if scene_aabb is not None:
            t_min, t_max = ops.ray_aabb_intersect(o, d, scene_aabb, c.near_plane, c.far_plane)
        else:
            # no scene box (contracted spaces): every ray marches near -> far (nerf.py:248-251).  One launch of the ray / box
            # kernel over an all-embracing box writes both constant vectors (t_min = max(0, near), t_max = far, the same
            # floats as torch.full); no cache of them across calls (it was keyed by the ray count, which alternates
            # between the renders of a step, and shared across streams without an event)
            if c.near_plane is not None and c.far_plane is not None:
                if c.near_plane < 0:
                    raise ValueError("near_plane < 0")             # (the box test clamps t_min at 0: ADVICE r5)
                t_min, t_max = ops.ray_aabb_intersect(o, d, (-1e30,) * 3 + (1e30,) * 3, c.near_plane, c.far_plane)
            else:
                n = o.shape[0]
                t_min = torch.full((n,), 0.0 if c.near_plane is None else c.near_plane, device=o.device)
                t_max = torch.full((n,), 1e10 if c.far_plane is None else c.far_plane, device=o.device)
        mode = 0 if c.sampler == "occgrid" else 1
        if mode == 0 and o.is_cuda:
            if self._march_div_ok is None:                            # once per renderer: the exhaustive check of this box's extents
                self._march_div_ok = ops.march_div_check(c.aabb, o.device)
        march_mode = mode | (ops.MARCH_VERIFIED_DIV if (mode == 0 and self._march_div_ok) else 0)
        jit = jitter if training else None
        args = (o, d, t_min, t_max, jit, c.aabb, c.occ_res, self.binary, c.contraction_type,
                c.render_step_size, c.cone_angle, march_mode, c.n_uniform)
        # occupancy-grid marching is a ~1 000-step dependent chain per ray: the count pass keeps the first
        # march_cache intervals of every ray, the write pass copies them (re-marching only longer rays)
        cache = torch.empty(o.shape[0], c.march_cache, 2, device=o.device) if (mode == 0 and c.march_cache > 0) else None
        counts = ops.ray_march_count(*args, cache=cache)
        return dict(args=args, cache=cache, counts=counts, mode=mode)

    def _scan_counts(self, st: dict, device_counts) -> dict:
        counts, mode, n_rays = st["counts"], st["mode"], st["counts"].shape[0]
        dcst = None
        caps = self._capacities(n_rays) if (device_counts is True and mode == 0 and self.device_counts_ok()) else None
        if caps is not None:
            # device-side counts: scan and first guard in one launch (sample() continues from here without a host read)
            stats = torch.empty(4, device=counts.device, dtype=torch.int64)
            nd = torch.empty(2, device=counts.device, dtype=torch.int64)
            offsets, total = ops.scan_guard(counts, caps[0], nd[0:1], stats[0:2])
            dcst = dict(caps=caps, stats=stats, nd=nd)
        else:
            offsets, total = ops.exclusive_scan(counts)
        return dict(st, offsets=offsets, total=total, dc=dcst)

    # ---- device-side sample counts -----------------------------------------------------------------------------------
    CAP_MARGIN, CAP_SLACK, CAP_MAX = 1.25, 4096, 1 << 23      # (from 2^23 samples on the chunked two-stream paths take over)

    def device_counts_ok(self) -> bool:
        """the kernels that take the count from the device: occupancy sampler, NGP field on the bf16-matrix-core MLP
        kernels, binned scatter"""
        c = self.cfg
        return (c.device_counts is not False and c.sampler == "occgrid" and type(self) is Renderer and c.mlp_kernels == "x" and
                c.binned_scatter and self.field.flat.is_cuda)

    def _capacities(self, n_rays: int):
        if self._spr is None:
            return None
        def bucket(x):                                   # 8 sizes per octave: the allocator sees repeating sizes while
            q = max(1024, 1 << max(0, x.bit_length() - 4))      # the ray count drifts (update_train_batch_size)
            return (x + q - 1) // q * q
        caps = tuple(bucket(int(n_rays * s * self.CAP_MARGIN) + self.CAP_SLACK) for s in self._spr)
        return None if max(caps) >= self.CAP_MAX else caps

    def _learn_counts(self, n_rays: int, marched: int, kept: int):
        """capacities follow the counts: up at once, down slowly (the occupancy grid prunes over the first epochs)"""
        m = (marched / max(n_rays, 1), kept / max(n_rays, 1))
        self._spr = m if self._spr is None else tuple(max(a, 0.75 * b + 0.25 * a) for a, b in zip(m, self._spr))

    def _count_log(self, n_rays: int, caps, stats: torch.Tensor) -> CountLog:
        if self._polled_logs is not None:                        # inside a capture: the graph owns its pinned words
            log = CountLog(n_rays, caps, stats, self._polled_pinned.pop(), polled=True)   # (pinned before the capture began)
            self._polled_logs.append(log)
            return log
        if self._count_ring is None:
            self._count_ring = [torch.empty(4, dtype=torch.int64).pin_memory() for _ in range(16)]
            self._count_logs = [None] * 16
        k = self._count_ring_at = (self._count_ring_at + 1) % 16
        if self._count_logs[k] is not None:
            self._count_logs[k].wait()                       # (its pinned buffer is about to be reused)
        log = self._count_logs[k] = CountLog(n_rays, caps, stats, self._count_ring[k])
        return log

    def _sample_device_counts(self, o, d, st, caps, keep_feat: bool) -> Packed:
        """sample() without a host read: both counts stay on the device, every array has a capacity, two guards clear the
        render (and report it) should a count not fit"""
        c = self.cfg
        args, cache, counts, offsets = st["args"], st["cache"], st["counts"], st["offsets"]
        dev = o.device
        if st.get("dc") is not None:                         # sample_begin ran the first guard with its scan
            caps, stats, nd = st["dc"]["caps"], st["dc"]["stats"], st["dc"]["nd"]
        else:
            stats = torch.empty(4, device=dev, dtype=torch.int64)
            nd = torch.empty(2, device=dev, dtype=torch.int64)
            ops.count_guard(counts, st["total"], caps[0], nd[0:1], stats[0:2])
        cap0, cap1 = caps
        n0_dev, n1_dev = nd[0:1], nd[1:2]
        ri, ts, te = ops.ray_march_write(*args, offsets, cap0, counts=counts, cache=cache)
        keep_feat = keep_feat and self._reuse_prepass_feat
        sigma = self._density_stream(o, d, (ri, ts, te), cap0, keep_feat, n_dev=n0_dev)
        feat0 = None
        if keep_feat:
            sigma, feat0 = sigma
        keep, kept = ops.visibility(offsets, counts, sigma, ts, te, c.early_stop_eps, c.alpha_thre)
        new_offsets, total2 = ops.scan_guard(kept, cap1, n1_dev, stats[2:4], counts_also=counts)
        ri2, ts2, te2 = ops.compact_samples(offsets, counts, new_offsets, keep, ts, te, cap1)
        feat1 = ops.compact_features(offsets, counts, new_offsets, keep, feat0, cap1, n_dev=n1_dev) if feat0 is not None else None
        log = self._count_log(o.shape[0], caps, stats)
        return Packed(ri2, ts2, te2, new_offsets, kept, cap1, cap0, feat1, n_dev=n1_dev, log=log)

    def sample(self, o, d, jitter: Optional[torch.Tensor], training: bool, keep_feat: bool = False,
               begun: Optional[dict] = None, device_counts=False) -> Packed:
        """device_counts: True -- the caller (Trainer) can work with counts that stay on the device (Packed.n_dev / Packed.log);
        "learn" -- host-side counts as ever, but the capacities of later renders learn from them"""
        c = self.cfg
        st = begun if begun is not None else self.sample_begin(o, d, jitter, training, device_counts=device_counts)
        learn = device_counts and st["mode"] == 0 and self.device_counts_ok()
        dc_now = learn and device_counts is True and "n0" not in st
        if st.get("dc") is not None and not dc_now:          # begun for device-side counts (a guard may have cleared it), wanted with host counts
            st = self.sample_begin(o, d, jitter, training)
        args, cache, counts, offsets, mode = st["args"], st["cache"], st["counts"], st["offsets"], st["mode"]
        if dc_now:
            caps = st["dc"]["caps"] if st.get("dc") is not None else self._capacities(o.shape[0])
            if caps is not None:
                return self._sample_device_counts(o, d, st, caps, keep_feat)
        # host sync, as in the reference (external/utils.py:106-119); `begun["n0"]`: already read back (Trainer.prefetch)
        n0 = st["n0"] if "n0" in st else _host_int(st["total"])
        ri, ts, te = ops.ray_march_write(*args, offsets, n0, counts=counts, cache=cache)
        if mode == 1 or n0 == 0:
            if learn:
                self._learn_counts(o.shape[0], n0, n0)
            return Packed(ri, ts, te, offsets, counts, n0, n0)
        # sigma_fn pre-pass (external/utils.py:68-81) + render_visibility
        keep_feat = keep_feat and self._reuse_prepass_feat and c.mlp_kernels == "x"
        sigma = self._density_stream(o, d, (ri, ts, te), n0, keep_feat)
        feat0 = None
        if keep_feat:
            sigma, feat0 = sigma
        keep, kept = ops.visibility(offsets, counts, sigma, ts, te, c.early_stop_eps, c.alpha_thre)
        new_offsets, total2 = ops.exclusive_scan(kept)
        n1 = _host_int(total2)
        ri2, ts2, te2 = ops.compact_samples(offsets, counts, new_offsets, keep, ts, te, n1)
        feat1 = None
        if feat0 is not None and n1 > 0:                     # the pre-pass already encoded every survivor
            feat1 = feat0 if n1 == n0 else ops.compact_features(offsets, counts, new_offsets, keep, feat0, n1)
        if learn:
            self._learn_counts(o.shape[0], n0, n1)
        return Packed(ri2, ts2, te2, new_offsets, kept, n1, n0, feat1)

    # ---- field evaluation over a packed sample stream (overridden by vanilla.VanillaRenderer) ----------
    def _mlp_params(self):
        """Parameter block handed to the MLP kernels: the fp32 master copy, or (bf16 mode) its bf16-rounded
        image -- 9 425 floats, re-rounded per call so it always follows the optimiser."""
        m = self.field.mlp
        return m.to(torch.bfloat16).to(torch.float32) if self.cfg.mlp_bf16 else m

    def _density_stream(self, o, d, samples, n, return_feat: bool = False, n_dev=None):
        feat = ops.hashgrid_fwd(self.field.grid, self.field.table, scene=self.scene, rays=(o, d),
                                samples=samples, n=n, layout=1, n_dev=n_dev)
        if self.cfg.mlp_kernels == "x":
            _, sigma, _, _ = ops.mlp_fwd_x(self.field.mlp, self.field.C, self._xmode(), feat, self.scene, rays=(o, d),
                                           samples=samples, n=n, density_only=True, act=self._act_code, n_dev=n_dev)
            return (sigma, feat) if return_feat else sigma
        _, sigma, _ = ops.mlp_fwd(self._mlp_params(), self.field.C, feat, self.scene, rays=(o, d),
                                  samples=samples, n=n, density_only=True, bf16=self.cfg.mlp_bf16, act=self._act_code)
        return sigma

    def _binned_workspace(self, n: int, device) -> torch.Tensor:
        """staging pool of the binned hash-grid backward (~1.4 KB per sample): grown on demand, and given back when a pass
        needs less than a quarter of it for 32 passes in a row -- the first steps of a run, before the occupancy grid has
        pruned and the batch size has adapted, can march 50 M samples (an 80 GB pool that would otherwise stay)"""
        need = ops.hashgrid_bwd_binned_workspace_bytes(n)
        ws = self._bin_ws
        if self._polled_logs is not None:                             # inside a capture: the pool was sized before it began
            if ws is None or ws.numel() < need:
                raise RuntimeError("binned-scatter workspace too small for the step being captured")
            return ws
        if ws is not None and ws.numel() >= need:
            self._bin_ws_small = self._bin_ws_small + 1 if ws.numel() > 4 * need else 0
            if self._bin_ws_small < 32:
                return ws
        shrink = ws is not None and ws.numel() >= need
        del ws
        self._bin_ws = None                                           # release before (re)allocating
        self._bin_ws_small = 0
        if shrink:
            torch.cuda.empty_cache()                                  # rare: hand the big block back to the driver as well
        self._bin_ws = torch.empty(need + need // 4, device=device, dtype=torch.uint8)
        return self._bin_ws

    def _save_acts(self) -> bool:
        c = self.cfg
        return (c.mlp_kernels != "x") if c.save_activations is None else bool(c.save_activations)

    def _xmode(self) -> int:
        return 1 if self.cfg.mlp_bf16 else 3 if self.cfg.mlp_precision == "high" else 6

    def _field_forward(self, o, d, pk, save):
        f = self.field
        samples = (pk.ray_indices, pk.t_starts, pk.t_ends)
        chunks = min(self.cfg.fwd_chunks, pk.n >> 20) if pk.n >= (1 << 23) else 1   # pays from ~8 M samples, >= 1 M per chunk
        if self.cfg.mlp_kernels == "x" and chunks > 1 and pk.feat is None and pk.n_dev is None:
            return self._field_forward_chunked(o, d, pk, save, chunks)
        feat = pk.feat if pk.feat is not None else \
            ops.hashgrid_fwd(f.grid, f.table, scene=self.scene, rays=(o, d), samples=samples, n=pk.n, layout=1, n_dev=pk.n_dev)
        if self.cfg.mlp_kernels == "x":
            rgb, sigma, base, acts = ops.mlp_fwd_x(f.mlp, f.C, self._xmode(), feat, self.scene, rays=(o, d), samples=samples,
                                                   n=pk.n, save=save, save_acts=self._save_acts(), act=self._act_code, n_dev=pk.n_dev)
            return rgb, sigma, dict(feat=feat, base=base, acts=acts, xmode=self._xmode() if save else None)
        mp = self._mlp_params()
        if save and self._save_acts():
            rgb, sigma, base, acts = ops.mlp_fwd_save(mp, f.C, feat, self.scene, rays=(o, d), samples=samples, n=pk.n,
                                                      bf16=self.cfg.mlp_bf16, act=self._act_code)
            return rgb, sigma, dict(feat=feat, base=base, mlp_params=mp, acts=acts)
        rgb, sigma, base = ops.mlp_fwd(mp, f.C, feat, self.scene, rays=(o, d), samples=samples, n=pk.n,
                                       save_base=save, bf16=self.cfg.mlp_bf16, act=self._act_code)
        return rgb, sigma, dict(feat=feat, base=base, mlp_params=mp)

    def _field_forward_chunked(self, o, d, pk, save, K):
        """Encoding (texture-addresser bound) and MLP (VALU / matrix-core bound) of different sample chunks on two
        HIP streams, so the two kernels share the CUs instead of running back to back.  Every output layout is
        local to a 32-sample block, so chunk k writes a slice of the full tensors."""
        f, n = self.field, pk.n
        dev = o.device
        lib_acts = ops.mlp_act_save_floats(32)
        nblk = ops.n_blocks32(n)
        per = -(-nblk // K)
        feat = torch.empty(nblk * ops.FRAG_FLOATS_PER_BLOCK, device=dev)
        sigma = torch.empty(n, device=dev)
        rgb = torch.empty(n, f.C, device=dev)
        base = torch.empty(nblk * ops.BASE_FLOATS_PER_BLOCK, device=dev) if save else None
        acts = torch.empty(nblk * lib_acts, device=dev) if (save and self._save_acts()) else None
        if self._fwd_streams is None:
            self._fwd_streams = (torch.cuda.Stream(dev), torch.cuda.Stream(dev))
        main = torch.cuda.current_stream()
        s_enc, s_mlp = self._fwd_streams
        s_enc.wait_stream(main)
        s_mlp.wait_stream(main)
        for k in range(K):
            b0, b1 = k * per, min(nblk, (k + 1) * per)
            if b0 >= b1:
                break
            lo, hi = b0 * 32, min(n, b1 * 32)
            smp = (pk.ray_indices[lo:hi], pk.t_starts[lo:hi], pk.t_ends[lo:hi])
            fk = feat[b0 * ops.FRAG_FLOATS_PER_BLOCK: b1 * ops.FRAG_FLOATS_PER_BLOCK]
            with torch.cuda.stream(s_enc):
                ops.hashgrid_fwd(f.grid, f.table, scene=self.scene, rays=(o, d), samples=smp, n=hi - lo, layout=1, out=fk)
                ev = torch.cuda.Event()
                ev.record(s_enc)
            with torch.cuda.stream(s_mlp):
                s_mlp.wait_event(ev)
                out = (rgb[lo:hi], sigma[lo:hi],
                       base[b0 * ops.BASE_FLOATS_PER_BLOCK: b1 * ops.BASE_FLOATS_PER_BLOCK] if save else None,
                       acts[b0 * lib_acts: b1 * lib_acts] if acts is not None else None)
                ops.mlp_fwd_x(f.mlp, f.C, self._xmode(), fk, self.scene, rays=(o, d), samples=smp, n=hi - lo, save=save,
                              out=out, share_cu=True, act=self._act_code)
        main.wait_stream(s_enc)
        main.wait_stream(s_mlp)
        return rgb, sigma, dict(feat=feat, base=base, acts=acts, xmode=self._xmode() if save else None)

    # ---- data parallelism: the slice of the packed gradient buffer that is reduced beside the rest of the backward ----
    def dp_early_slice(self) -> Optional[Tuple[int, int]]:
        """element range [start, stop) of field.grad_all whose all-reduce starts inside the step's LAST backward pass (the
        fine hash levels), or None.  A function of the configuration only, so every rank issues the same collectives."""
        if self.grad_sync is None or not self.cfg.dp_overlap or not self.cfg.binned_scatter or not self.dp_early_enabled:
            return None
        f = self.field
        return 2 * int(f.grid.offset[self.cfg.dp_split_level]), f.n_table

    def dp_early(self):
        a, b = self.dp_early_slice()
        self.grad_sync.early(self.field.grad_all, a, b)

    def _field_backward_chunked(self, ctx, d_rgb, d_sig, K: int):
        """MLP backward (VALU / matrix-core bound, persistent kernels) of sample chunk k + 1 on the current stream beside
        the binned hash-grid scatter (memory bound) of chunk k on a second stream; the bins are flushed ONCE at the end
        (ren_hashgrid_bwd_binned_begin / _scatter / _finish).  The head kernel of the MLP backward runs one wave per SIMD
        with the whole register file, so nothing can share a CU with it: while a scatter is in flight the MLP kernels are
        launched on `bwd_mlp_cus` CUs and the scatter takes the rest plus whatever frees up.  Every layout is local to a
        32-sample block, so a chunk is a slice of the full tensors; weight gradients add up over the chunks' slab
        reductions.  Same results as the single launches up to the summation order of the MLP weight gradients."""
        f, pk = self.field, ctx["pk"]
        n, dev = pk.n, d_rgb.device
        nblk = ops.n_blocks32(n)
        per = -(-nblk // K)
        FR, BA = ops.FRAG_FLOATS_PER_BLOCK, ops.BASE_FLOATS_PER_BLOCK
        dfeat = torch.empty(nblk * FR, device=dev)
        d_base = torch.empty(nblk * BA, device=dev)
        ws = self._binned_workspace(n, dev)
        rays, samples = (ctx["o"], ctx["d"]), (pk.ray_indices, pk.t_starts, pk.t_ends)
        if self._bwd_stream is None:
            self._bwd_stream = torch.cuda.Stream(dev)
        main, side = torch.cuda.current_stream(), self._bwd_stream
        ops.hashgrid_bwd_binned_begin(f.grid, ws, scene=self.scene, rays=rays, samples=samples, n=n)
        side.wait_stream(main)
        acts_per = ops.mlp_act_save_floats(32)
        for k in range(K):
            b0, b1 = k * per, min(nblk, (k + 1) * per)
            if b0 >= b1:
                break
            lo, hi = b0 * 32, min(n, b1 * 32)
            acts = ctx["acts"][b0 * acts_per: b1 * acts_per] if ctx.get("acts") is not None else None
            ops.mlp_bwd_x(f.mlp, f.C, ctx["xmode"], ctx["feat"][b0 * FR: b1 * FR], ctx["base"][b0 * BA: b1 * BA], acts,
                          self.scene, rays=rays, samples=tuple(t[lo:hi] for t in samples), n=hi - lo,
                          rgb=ctx["rgb"][lo:hi], d_rgb=d_rgb[lo:hi], d_sigma=d_sig[lo:hi], grad_mlp_params=f.g_mlp,
                          workspace=self._ws, dfeat=dfeat[b0 * FR: b1 * FR], d_base=d_base[b0 * BA: b1 * BA],
                          act=self._act_code, grid_cus=0 if k == 0 else self.cfg.bwd_mlp_cus)   # chunk 0 has the chip to itself
            ev = torch.cuda.Event()
            ev.record(main)
            with torch.cuda.stream(side):
                side.wait_event(ev)
                ops.hashgrid_bwd_binned_scatter(f.grid, f.g_table, dfeat, ws, scene=self.scene, rays=rays, samples=samples,
                                                n=n, first=lo, m=hi - lo)
        main.wait_stream(side)
        ops.hashgrid_bwd_binned_finish(f.grid, f.g_table, ws, n=n)

    def _field_backward(self, ctx, d_rgb, d_sig, final: bool = False):
        f, pk = self.field, ctx["pk"]
        K = min(self.cfg.bwd_chunks, pk.n >> 21) if pk.n >= (1 << 23) else 1     # pays from ~8 M samples, >= 2 M per chunk
        if (K > 1 and ctx.get("xmode") is not None and self.cfg.binned_scatter and pk.n_dev is None and
                not (final and self.dp_early_slice() is not None)):
            return self._field_backward_chunked(ctx, d_rgb, d_sig, K)
        samples = (pk.ray_indices, pk.t_starts, pk.t_ends)
        mp = ctx.get("mlp_params")                      # absent when the forward ran on the (fp32) tangent kernels
        if ctx.get("xmode") is not None:
            # (bwd_side_cus: the head kernel holds the whole register file of every CU it runs on, and a kernel of the side
            # stream -- the third render's one-workgroup scan -- then waits for it to END: 0.55 ms in the e2e trace of round 5)
            cus = torch.cuda.get_device_properties(d_rgb.device).multi_processor_count - self.bwd_side_cus if self.bwd_side_cus else 0
            dfeat = ops.mlp_bwd_x(f.mlp, f.C, ctx["xmode"], ctx["feat"], ctx["base"], ctx["acts"], self.scene,
                                  rays=(ctx["o"], ctx["d"]), samples=samples, n=pk.n, rgb=ctx["rgb"], d_rgb=d_rgb,
                                  d_sigma=d_sig, grad_mlp_params=f.g_mlp, workspace=self._ws, act=self._act_code, n_dev=pk.n_dev,
                                  grid_cus=max(cus, 0))
        elif ctx.get("acts") is not None:
            dfeat = ops.mlp_bwd_saved(mp, f.C, ctx["feat"], ctx["base"], ctx["acts"], self.scene, rays=(ctx["o"], ctx["d"]),
                                      samples=samples, n=pk.n, rgb=ctx["rgb"], d_rgb=d_rgb, d_sigma=d_sig,
                                      grad_mlp_params=f.g_mlp, workspace=self._ws, bf16=self.cfg.mlp_bf16, act=self._act_code)
        else:
            dfeat = ops.mlp_bwd(f.mlp if mp is None else mp, f.C, ctx["feat"], ctx["base"], self.scene,
                                rays=(ctx["o"], ctx["d"]), samples=samples, n=pk.n, rgb=ctx["rgb"], d_rgb=d_rgb,
                                d_sigma=d_sig, grad_mlp_params=f.g_mlp, workspace=self._ws,
                                bf16=self.cfg.mlp_bf16 and mp is not None, act=self._act_code)
        if self.cfg.binned_scatter:
            self._binned_workspace(pk.n, dfeat.device)
            kw = dict(scene=self.scene, rays=(ctx["o"], ctx["d"]), samples=samples, n=pk.n, layout=1, n_dev=pk.n_dev)
            if final and self.dp_early_slice() is not None:
                # last backward of the step under data parallelism: fine levels first, their slice of the table gradient
                # is all-reduced while the coarse levels are scattered
                lo_mask = (1 << self.cfg.dp_split_level) - 1
                ops.hashgrid_bwd_binned(f.grid, f.g_table, dfeat, self._bin_ws, level_mask=0xFFFF & ~lo_mask, **kw)
                self.dp_early()
                ops.hashgrid_bwd_binned(f.grid, f.g_table, dfeat, self._bin_ws, level_mask=lo_mask, **kw)
            else:
                ops.hashgrid_bwd_binned(f.grid, f.g_table, dfeat, self._bin_ws, **kw)
        else:
            ops.hashgrid_bwd(f.grid, f.g_table, dfeat, scene=self.scene, rays=(ctx["o"], ctx["d"]), samples=samples,
                             n=pk.n, layout=1)

    # ---- forward render ---------------------------------------------------------------------------
    def forward(self, o, d, jitter=None, bkgd: Optional[torch.Tensor] = None, training: bool = True,
                save: bool = True, begun: Optional[dict] = None, device_counts: bool = False):
        f = self.field
        pk = self.sample(o, d, jitter, training, keep_feat=True, begun=begun, device_counts=device_counts)
        n_rays = o.shape[0]
        if pk.n == 0:
            colors = torch.zeros(n_rays, f.C, device=o.device)
            if bkgd is not None:
                colors = colors + bkgd
            zero = torch.zeros(n_rays, device=o.device)
            return colors, zero, zero.clone(), dict(pk=pk, empty=True, bkgd=bkgd)
        rgb, sigma, fctx = self._field_forward(o, d, pk, save)
        colors, opac, depth, w, T = ops.composite_fwd(pk.offsets, pk.counts, pk.t_starts, pk.t_ends, sigma, rgb,
                                                      f.C, bkgd, save=save)
        ctx = dict(pk=pk, o=o, d=d, rgb=rgb, sigma=sigma, w=w, T=T, opac=opac, bkgd=bkgd, empty=False, **fctx)
        return colors, opac, depth, ctx

    # ---- backward: accumulates into field.grad, returns d(bkgd) -------------------------------------
    def backward(self, ctx, g_colors, g_opac=None, g_depth=None, final: bool = False, per_ray_bkgd: bool = False):
        """final: this is the last backward pass of the step (its gradients are complete when it returns).
        per_ray_bkgd: return the (R, C) per-ray background gradient instead of its column sums (Trainer folds the sum
        into ren_bkgd_param_grad)"""
        f = self.field
        if ctx["empty"]:
            # a rank without a single sample must still issue the collectives its peers issue (the slice is final here:
            # this pass adds nothing to it) -- otherwise the ranks' all-reduce sequences differ and RCCL hangs
            if final and self.dp_early_slice() is not None:
                self.dp_early()
            if ctx.get("bkgd") is None:
                return None
            return g_colors if per_ray_bkgd else g_colors.sum(0)
        pk = ctx["pk"]
        d_sig, d_rgb, d_bk = ops.composite_bwd(pk.offsets, pk.counts, pk.t_starts, pk.t_ends, ctx["sigma"],
                                               ctx["rgb"], f.C, ctx["bkgd"], ctx["w"], ctx["T"], ctx["opac"],
                                               g_colors, g_opac, g_depth, want_bkgd=ctx["bkgd"] is not None)
        self._field_backward(ctx, d_rgb, d_sig, final=final)
        if per_ray_bkgd:
            return d_bk
        return ops.column_sum(d_bk) if d_bk is not None else None

    # ---- density query (occ_eval_fn / query_density) ----------------------------------------------------
    def query_density(self, x_world: torch.Tensor) -> torch.Tensor:
        """NGPradianceField.query_density(x) (ngp.py:230-254) for arbitrary world points."""
        f = self.field
        n = x_world.shape[0]
        xu = contract_points(x_world, self.cfg.aabb, self.cfg.contraction_type)
        feat = ops.hashgrid_fwd(f.grid, f.table, x_unit=xu, n=n, layout=1)
        if self.cfg.mlp_kernels == "x":
            return ops.mlp_fwd_x(f.mlp, f.C, self._xmode(), feat, self.scene, x_world=x_world, n=n, density_only=True, act=self._act_code)[1]
        _, sigma, _ = ops.mlp_fwd(self._mlp_params(), f.C, feat, self.scene, x_world=x_world, n=n, density_only=True,
                                  bf16=self.cfg.mlp_bf16, act=self._act_code)
        return sigma

    # ---- occupancy grid (K14): nerfacc OccupancyGrid.every_n_step as driven by nerf.py:170-204 ------------
    def update_occ_grid(self, step: int, cam_positions: Optional[torch.Tensor] = None,
                        generator: Optional[torch.Generator] = None, indices=None, jitter=None, cam_ids=None):
        c = self.cfg
        if step % c.occ_n != 0:
            return False
        dev = self.occs.device
        cells = self.occs.numel()
        if generator is None:
            # every rank owns the SAME stream for the grid refresh (cell sample, in-cell jitter, camera choice), consumed
            # by nothing else: the replicated occupancy grids stay bit-identical without DDP's per-forward buffer
            # broadcast (scripts/run.py:81-93, models/nerf.py:98-102; collective C4 of SURVEY 2.3)
            if self._occ_gen is None:
                self._occ_gen = torch.Generator(device=dev).manual_seed(self.cfg.occ_seed)
            generator = self._occ_gen
        if indices is None:
            if step < c.warmup_steps:
                indices = torch.arange(cells, device=dev)
            else:
                n = cells // 4
                uni = torch.randint(cells, (n,), device=dev, generator=generator)
                occ_idx = torch.nonzero(self.binary)[:, 0]
                if n < occ_idx.numel():
                    occ_idx = occ_idx[torch.randint(occ_idx.numel(), (n,), device=dev, generator=generator)]
                indices = torch.cat([uni, occ_idx])
        if jitter is None:
            jitter = torch.rand(indices.shape[0], 3, device=dev, generator=generator)
        x, valid = ops.occgrid_cell_points(indices, jitter, c.aabb, c.occ_res, c.contraction_type)
        sigma = self.query_density(x)
        step_sizes = None
        if c.cone_angle > 0.0:                                                    # nerf.py:175-193
            if cam_ids is None:
                cam_ids = torch.randint(0, cam_positions.shape[0], (x.shape[0],), device=dev, generator=generator)
            t = (cam_positions[cam_ids] - x).norm(dim=-1)
            step_sizes = torch.clamp(t * c.cone_angle, min=c.render_step_size)
            if c.near_plane is not None and c.far_plane is not None:
                step_sizes = torch.where((t > c.near_plane) & (t < c.far_plane), step_sizes,
                                         torch.zeros_like(step_sizes))
            step_sizes = step_sizes.contiguous()
        if self._occ_scratch is None:
            self._occ_scratch = torch.empty_like(self.occs)
        ops.occgrid_ema(self.occs, indices, valid, sigma, step_sizes, c.render_step_size, c.ema_decay,
                        scratch=self._occ_scratch)
        ops.occgrid_binarize(self.occs, c.occ_thre, self.binary, self._scratch)
        self.binary_epoch = getattr(self, "binary_epoch", 0) + 1        # a march prefetched over the old grid is stale now
        return True


def contract_points(x_world: torch.Tensor, aabb: Sequence[float], ct: int) -> torch.Tensor:
    """World -> unit cube for free-standing point queries (ngp.py:230-237); plumbing-level torch."""
    ab = torch.tensor(aabb, device=x_world.device, dtype=torch.float32)
    lo, hi = ab[:3], ab[3:]
    x = (x_world - lo) / (hi - lo)
    if ct == ops.UN_BOUNDED_SPHERE:
        x = x * 2 - 1
        mag = x.norm(dim=-1, keepdim=True)
        x = torch.where(mag > 1, (2 - 1 / mag) * (x / mag), x)
        x = x / 4 + 0.5
    elif ct == ops.UN_BOUNDED_TANH:
        x = (torch.tanh(x - 0.5) + 1) / 2
    return x.contiguous()


# ===================================================================================================
@dataclass
class TrainCfg:
    """loss.* / optimizer.* of configs/train/synthetic.yaml."""
    err_diff: str = "mse"
    w_diff: float = 1.0
    pw_diff: Optional[str] = "mean_contrast_reciprocal_sq"
    lr: float = 0.01
    weight_decay: float = 1e-6
    betas: Tuple[float, float] = (0.9, 0.999)
    eps: float = 1e-8
    bkgd_is_param: bool = True
    err_grad: str = "mape"                       # loss.error_fn.log_intensity_grad
    w_grad: float = 0.0                          # loss.weight.log_intensity_grad (1e-3 in the real-data configs)
    pw_grad: Optional[str] = None                # loss.param_weight.log_intensity_grad
    train_contrast_threshold: bool = False       # model.contrast_threshold.freeze == false
    lr_contrast_threshold: float = 0.1           # optimizer.lr.contrast_threshold
    train_refractory_period: bool = False        # model.refractory_period.freeze == false
    relative_lr_refractory_period: float = 50.0  # optimizer.relative_lr.refractory_period (x tau_max)


class Trainer:
    """Training step for the log-intensity-difference loss (two renders batched into one pass)."""

    def __init__(self, renderer: Renderer, tcfg: TrainCfg, *, Kinv, tab_ts, tab_pos, tab_quat,
                 p2n_raw, neg_ct, tau_raw, tau_max, bkgd_raw, world_size: int = 1, process_group=None):
        self.r = renderer
        self.t = tcfg
        dev = renderer.field.flat.device
        self.Kinv = Kinv.to(dev, torch.float32).contiguous()
        self.tab_ts = tab_ts.to(dev).contiguous()
        self.tab_pos = tab_pos.to(dev, torch.float32).contiguous()
        self.tab_quat = tab_quat.to(dev, torch.float32).contiguous()
        # Event-generation parameters (event_generation_params.py:51-84,162-203): evaluated here on the host as the reference
        # does in float32 / float64, then DEVICE-RESIDENT (self.ep, ops.EP_*): every kernel takes C_p, C_n, tau and the 1 / C^k
        # param weight from that block, and a trainable ratio / refractory period is stepped and re-evaluated on the device
        # (ren_tau_adam_step, ren_event_params_refresh), so no launch of the next step waits for a host read of a parameter.
        # The host-side attributes (c_p, mean_c, tau, ct_raw, tau_raw) are read back on demand.
        ratio = torch.nn.functional.softplus(p2n_raw.detach().cpu().to(torch.float32))   # event_generation_params.py:51-70
        neg = neg_ct.detach().cpu().to(torch.float32)
        c_p, self.c_n = float(ratio * neg), float(neg)
        mean_c = float((ratio * neg + neg) / 2)
        traw, tmax = tau_raw.detach().cpu().to(torch.float64), tau_max.detach().cpu()
        lim = torch.tensor(1e-4, dtype=torch.float64).logit().abs()
        traw = tmax * (traw / tmax).clamp(-lim, lim)                          # :170-185
        tau = float(tmax * torch.sigmoid(traw / tmax))                        # modules.py:58-74 (float64)
        self.tau_max = tmax.to(torch.float64)
        self._tau_shape = tuple(traw.shape)
        ct_raw = float(p2n_raw.detach().reshape(-1)[0].to(torch.float32))
        self._ep_host = [c_p, self.c_n, ct_raw, tau, 1.0 / mean_c, 1.0 / mean_c ** 2, float(traw.reshape(-1)[0]), 0.0]
        self._ep_stale = False
        self.ep = torch.tensor(self._ep_host, dtype=torch.float64, device=dev)
        # trainable refractory period: a float64 scalar (event_generation_params.py:162-164); its gradient is assembled from
        # per-ray forward-mode tangents dI/dt, its Adam group (state: exp_avg, exp_avg_sq) lives on the device as well
        self._tau_raw_dev = traw.reshape(1).clone().to(dev)
        self._tau_adam, self._tau_adam_steps = torch.zeros(2, dtype=torch.float64, device=dev), 0
        # small-parameter block: [bkgd_raw (C) | pad] with its own Adam state (group "others", lr default)
        self.small = torch.zeros(4, device=dev, dtype=torch.float32)
        self.small[: renderer.field.C] = bkgd_raw.to(dev, torch.float32).reshape(-1)
        # the scalar parameters' gradients share ONE 64-byte block: [small_grad 4 f32 | ct_grad 4 f32 | d loss / d tau f64 | pad] -- a
        # pass whose sample counts stay on the device snapshots it with one copy (an empty, overflowed render still adds its
        # background-only loss terms to these three; to the table / MLP gradients it adds exactly nothing)
        self._gs = torch.zeros(8, dtype=torch.float64, device=dev)
        self.small_grad = self._gs[0:2].view(torch.float32)
        f = renderer.field
        self.m = torch.zeros_like(f.flat)
        self.v = torch.zeros_like(f.flat)
        self.sm = torch.zeros_like(self.small)
        self.sv = torch.zeros_like(self.small)
        self.step_count = 0
        self.world_size = world_size
        self.pg = process_group
        self.sync = None
        self._side, self._prefetched, self._n_host = None, None, None   # Trainer.prefetch
        self._ready_ev = None                                    # start of the last forward_backward() on its stream (early sampling)
        self.early_grad_sampling = True                          # Trainer.step: third render's samples beside the l_diff backward
        self.grad_sampling: Optional[str] = None                 # "merged": that placement also for eager steps (grad_sampling_mode)
        self.side_cus = int(os.environ.get("REN_SIDE_CUS", 4))   # CUs the l_diff pass's MLP backward leaves to the side stream ("begun")
        self._grad_begun, self._grad_pending, self._n_host_grad = None, None, None   # begin_grad_sampling
        if world_size > 1:
            from . import parallel
            self.sync = parallel.GradSync(process_group, world_size, compress=renderer.cfg.dp_compress)
            renderer.grad_sync = self.sync
        self._mean_s_dev = None                                  # samples / ray of this step, summed over ranks (device)
        # device-side sample counts (RenderCfg.device_counts): the renders of the step in flight whose counts the host has
        # not looked at yet, and how to repeat them should one not have fitted its arrays
        self._device_counts: Optional[bool] = None               # the `device_counts` property: None = auto (device_counts_ok)
        if os.environ.get("REN_DEVICE_COUNTS", "") in ("0", "off"):  # A/B switch for scripts that build the trainer themselves
            self._device_counts = False
        self._update_dp_early()
        self._dc_sync, self.device_count_overflows = False, 0
        self.keep_ctx = False                                    # tests: aux["ctx"] = the render's context (rays, samples, features)
        # optimiser state on the device (ABI 25, ops.HY_*): Adam step numbers and bias corrections, and the sticky skip word a
        # captured step raises when one of its device-side counts did not fit (Trainer._graph_step)
        self._hyper = torch.zeros(8, dtype=torch.float64, device=dev)
        self.use_graph: Optional[bool] = None                    # Trainer.step: None = auto (a step shape seen twice in a row), False = never
        if os.environ.get("REN_STEP_GRAPH", "") in ("0", "off"):
            self.use_graph = False
        self._graphs, self._graph_last_key, self._graph_pool, self._capturing, self._graph_streak = {}, None, None, False, 0
        self._graph_bad, self._graph_eager_ev = {}, None         # per step shape: captures that replayed no faster than eager steps
        self.graph_replays = self.graph_captures = 0
        self.lr_scale = 1.0
        # trainable C_p / C_n ratio (softplus-parametrised scalar, its own Adam group with lr 0.1:
        # robust_e_nerf.py:800-803).  Its loss dependence is through the per-event targets and the 1/C^k
        # normalisation only, i.e. O(B) elementwise work on already rendered predictions.
        self.ct = torch.zeros(4, device=dev, dtype=torch.float32)
        self.ct[0] = p2n_raw.detach().reshape(-1)[0].to(dev, torch.float32)
        self.ct_grad, self.ct_m, self.ct_v = self._gs[2:4].view(torch.float32), torch.zeros_like(self.ct), torch.zeros_like(self.ct)
        self._tau_grad_dev = self._gs[4:5]                       # d loss / d tau, accumulated on the device

    # ---- host views of the device-resident event parameters (a read-back when they have moved since the last look) ----
    def _ep(self):
        if self._ep_stale:
            self._ep_host, self._ep_stale = self.ep.tolist(), False
        return self._ep_host

    c_p = property(lambda self: self._ep()[ops.EP_CP])
    mean_c = property(lambda self: (self._ep()[ops.EP_CP] + self._ep()[ops.EP_CN]) / 2)
    tau = property(lambda self: self._ep()[ops.EP_TAU])
    ct_raw = property(lambda self: self._ep()[ops.EP_RAW])

    @property
    def tau_raw(self) -> torch.Tensor:
        """the raw refractory period (float64, host copy, shaped like the constructor's tau_raw)"""
        return self._tau_raw_dev.cpu().reshape(self._tau_shape)

    def _refresh_event_params(self):
        ops.event_params_refresh(self.ct, self.c_n, self._tau_raw_dev, float(self.tau_max), self.ep)
        self._ep_stale = True

    def _pw_dev(self, kind):
        """device scalar of the param weight 1 / C^k (loss.param_weight.*, robust_e_nerf.py:470-486), None for k = 0"""
        return {None: None, "mean_contrast_reciprocal": self.ep[ops.EP_INV_C: ops.EP_INV_C + 1],
                "mean_contrast_reciprocal_sq": self.ep[ops.EP_INV_C2: ops.EP_INV_C2 + 1]}[kind]

    def device_counts_ok(self) -> bool:
        """Renders of this trainer keep their sample counts on the device (what Renderer.device_counts_ok asks for).  A
        function of the configuration only.  Under data parallelism a pass that overflowed is repeated by ITS rank alone,
        so no pass may contain a collective: the early all-reduce of the fine levels' slice (RenderCfg.dp_overlap) is off
        then and the whole packed buffer is reduced at the settle point -- after the last pass of the step has been looked
        at -- by optimizer_step (every rank: the same one all-reduce per step, whatever its counts did)."""
        return self._device_counts is not False and self.r.device_counts_ok()

    @property
    def device_counts(self) -> Optional[bool]:
        """None: auto (on where device_counts_ok() says so); False: the reference's host reads"""
        return self._device_counts

    @device_counts.setter
    def device_counts(self, v: Optional[bool]):
        self._device_counts = v
        self._update_dp_early()

    def _update_dp_early(self):
        """data parallelism: the early all-reduce inside the last backward pass only while no pass can be repeated"""
        if self.world_size > 1:
            self.r.dp_early_enabled = not self.device_counts_ok()

    def _dc_mode(self):
        """what this step's renders pass to Renderer.sample: True (counts stay on the device), "learn" (host counts, but the
        capacities learn from them) or False"""
        if not self.device_counts_ok():
            return False
        return "learn" if self._dc_sync else True

    def _dc_pass(self, fn, args):
        """Run one loss pass (`fn(*args, dc)` -> loss, aux, CountLog | None) with its sample counts on the device, then look
        at them: a wait for the pass's SAMPLING kernels -- its field evaluation and backward are still in the queue.  A count
        that did not fit its arrays left the render empty: the table / MLP gradients got nothing from it, the scalar
        parameters' gradient block is put back, and the pass runs again with host-side counts."""
        dc = self._dc_mode()
        if self._capturing:                                  # _graph_step looks at the counts after every replay
            loss, aux, log = fn(*args, dc)
            if log is None:
                raise RuntimeError("a captured step needs every render on device-side counts")
            self._cap_passes.append((log, aux))
            return loss, aux
        snap = self._gs.clone() if dc is True else None
        loss, aux, log = fn(*args, dc)
        if log is None:
            return loss, aux
        v = log.wait()
        if not (v[1] or v[3]):
            self.r._learn_counts(log.n_rays, v[0], v[2])
            aux["n"] = v[2]
            if "n_marched" in aux:
                aux["n_marched"] = v[0]
            return loss, aux
        self._gs.copy_(snap)
        self.device_count_overflows += 1
        self._dc_sync, self._grad_begun, self._grad_pending = True, None, None
        try:
            loss, aux, _ = fn(*args, self._dc_mode())
        finally:
            self._dc_sync = False
        return loss, aux

    @property
    def tau_grad(self) -> torch.Tensor:
        """d loss / d tau accumulated so far (float64 scalar, host copy)"""
        return self._tau_grad_dev.cpu().reshape(())

    # ---- Bayer sensor: every event sees one colour channel of the (., 3) render (`bayering`, robust_e_nerf.py:887-890)
    def _channel_index(self, batch, repeat: int):
        if self.r.field.C == 1:
            return None
        if "channel_idx" not in batch:
            raise ValueError("radiance_dim 3 (Bayer sensor) needs batch['channel_idx'] (data.colorize_events)")
        ch = batch["channel_idx"].to(torch.int64)
        return (torch.cat([ch] * repeat) if repeat > 1 else ch)[:, None]

    @staticmethod
    def _bayer(x, ch):
        """(R, C) -> (R,): the event's channel (monochrome: the only one)"""
        return x[:, 0] if ch is None else x.gather(1, ch)[:, 0]

    @staticmethod
    def _unbayer(g, ch, C):
        """(R,) gradient -> (R, C) with zeros in the channels the event does not see"""
        if ch is None:
            return g[:, None].contiguous()
        return torch.zeros(g.shape[0], C, device=g.device, dtype=g.dtype).scatter_(1, ch, g[:, None])

    def _param_grad(self, batch, pred, kind: str, valid=None):
        """d(loss term)/d(raw ratio) and the DIRECT d(loss term)/d(tau) (through the event-rate target), with the
        rendered prediction held fixed: `ren_event_param_grad` (closed form of the reference's autograd through
        event_generation_params.py:51-84,196-203, loss.py:32-74, robust_e_nerf.py:470-486)."""
        t = self.t
        err, w, pwk = (t.err_diff, t.w_diff, t.pw_diff) if kind == "diff" else (t.err_grad, t.w_grad, t.pw_grad)
        ops.event_param_grad(kind, err, pwk, pred.contiguous(), valid, batch, 0.0, 0.0, 0.0, 0.0, w,
                             ct_grad=self.ct_grad if t.train_contrast_threshold else None,
                             tau_grad=self._tau_grad_dev if t.train_refractory_period else None, ep=self.ep)

    # ---- a2-a4: event correction + supervision timestamps: one launch (ren_event_prepare) ---------------------
    def _prepare(self, batch):
        B = batch["position"].shape[0]
        p = ops.event_prepare(batch, 0.0, 0.0, 0.0, ep=self.ep)
        return p["ts"][:B], p["ts"][B:], p["target_diff"]

    # ---- the parameter-independent front of the l_diff step, and its prefetch ------------------------------------------
    def _front(self, batch, jitter_start, jitter_end) -> dict:
        """event correction -> supervision timestamps -> poses -> rays of the start and end renders (a2-a6)"""
        t = self.t
        prep = ops.event_prepare(batch, 0.0, 0.0, 0.0, with_dtau=t.train_refractory_period, ep=self.ep)
        jitter = None
        if jitter_start is not None:
            # jitter_end None: jitter_start already holds the 2B uniforms of both renders (start rays first)
            jitter = jitter_start if jitter_end is None else torch.cat([jitter_start, jitter_end])
            jitter = jitter.to(torch.float32).contiguous()
        front = dict(prep=prep, jitter=jitter)
        if not t.train_refractory_period:
            # poses + rays of both renders in one launch; the 2B timestamps share the B event pixels
            front["o"], front["d"] = ops.pose_rays(prep["ts"], batch["position"].contiguous(), self.Kinv, self.tab_ts,
                                                   self.tab_pos, self.tab_quat)
        else:
            front["px"] = torch.cat([batch["position"], batch["position"]]).contiguous()
        return front

    @property
    def side_stream(self) -> "torch.cuda.Stream":
        if self._side is None:
            # normal priority (round 6; until then -1 = high): with a HIGH-priority stream in the process every other captured
            # step replayed 25 % slower (small kernels of alternate captures took 45-55 us each: tools/recapture_probe.py,
            # profiles/NOTES.md), and the eager lines and the e2e run are the same at either priority
            self._side = torch.cuda.Stream(priority=int(os.environ.get("REN_SIDE_PRIORITY", 0)))
        return self._side

    def prefetch(self, batch, jitter_start=None, jitter_end=None, next_global_step: Optional[int] = None) -> bool:
        """Run the front of the NEXT step's forward_backward(batch, jitter_start, jitter_end) -- event correction, poses,
        rays, ray/AABB test, march count pass, scan, and the read-back of the sample count -- on a side stream while the
        current step's backward is still on the GPU.  The one host read of a step (the packed sample count, as in the
        reference: external/utils.py:106-119) then no longer waits for the previous step, so the launch queue never runs
        dry (measured: 0.35 ms of GPU idle per 12 ms step at BASELINE configs[1]).  Only what cannot depend on this
        step's optimiser update may run early: frozen C_p / tau (they move the timestamps).  With the occupancy sampler
        the early part is the march itself (a ~1 000-step dependent chain per ray, 0.46 ms for 131 k rays: 14 % of a step at
        the reference's 2^20-sample budget) -- it reads the occupancy GRID, not the field; the density pre-pass that does
        need the updated field stays in the step.  A grid refresh between prefetch() and the step (update_occ_grid, every
        16 steps) makes the early march stale: pass `next_global_step` so that it is not started before a refresh step;
        a stale one is recognised by the grid epoch and redone.  Otherwise this is a no-op and returns False.
        `batch` / the jitters must already be complete on the device or produced on `side_stream`."""
        t = self.t
        if t.train_contrast_threshold or t.train_refractory_period:
            return False
        if self.r.cfg.sampler != "uniform" and next_global_step is not None and next_global_step % self.r.cfg.occ_n == 0:
            return False
        side = self.side_stream
        # Ordered AFTER the work already queued on this stream (round 4): run beside the persistent MLP backward kernels, the
        # pose / ray kernels of this front produced a slightly wrong rotation for an aligned group of 16 rays in ~1 step out
        # of 8 (same inputs, same kernel: tools/prefetch_diag2.py; packed-FP32 code beside matrix-core waves, profiles/NOTES.md:
        # those files are now built without it, and this ordering stays as the second line of defence), which is what made
        # test_prefetched_step_front_gives_the_same_steps fail in 40 % of its stand-alone runs.  What is left of the prefetch:
        # the front shares the chip with the optimiser step only, and the host never enqueues it on the critical path.
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            front = self._front(batch, jitter_start, jitter_end)
            st = self.r.sample_begin(front["o"], front["d"], front["jitter"], True)
            if self._n_host is None:
                self._n_host = torch.empty(1, dtype=st["total"].dtype).pin_memory()
            self._n_host.copy_(st["total"].reshape(1), non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(side)
        front["begun"] = st
        self._prefetched = (self._prefetch_key(batch, jitter_start, jitter_end) + self._grid_state(), front, ev,
                            (batch, jitter_start, jitter_end))
        return True

    def _grid_state(self):
        """what the early march of the occupancy sampler read: the grid tensor's torch version and the refresh epoch"""
        r = self.r
        return (r.binary._version, getattr(r, "binary_epoch", 0), r.cfg.sampler)

    @staticmethod
    def _prefetch_key(batch, jitter_start, jitter_end):
        """identity AND in-place version of every tensor the prefetched front was computed from: a batch or jitter tensor
        mutated between prefetch() and forward_backward() (same object ids, new contents) must not reuse the stale front"""
        ver = lambda t: (id(t), t._version) if isinstance(t, torch.Tensor) else (id(t), 0)
        return (id(batch), tuple(sorted((k, ver(v)) for k, v in batch.items())), ver(jitter_start), ver(jitter_end))

    def _take_prefetched(self, batch, jitter_start, jitter_end):
        pf, self._prefetched = self._prefetched, None
        t = self.t
        if pf is None or pf[0] != self._prefetch_key(batch, jitter_start, jitter_end) + self._grid_state() or \
                t.train_contrast_threshold or t.train_refractory_period:              # the guards of prefetch(), re-checked
            return None
        _, front, ev, _ = pf
        ev.synchronize()                                               # host: waits for the side stream's few small kernels only
        front["begun"]["n0"] = int(self._n_host[0])
        cur = torch.cuda.current_stream()
        cur.wait_event(ev)

        _adopt(front, cur)                                             # side-stream allocations now used on this stream
        return front

    def forward_backward(self, batch, jitter_start=None, jitter_end=None, final: Optional[bool] = None):
        """Loss + gradients (no optimiser step).  Returns (loss tensor (device scalar), aux).  final: this is the step's last
        backward pass (default: yes unless the log-intensity-gradient term follows)."""
        if final is None:
            final = not (self.t.w_grad > 0)
        return self._dc_pass(self._forward_backward, (batch, jitter_start, jitter_end, final))

    def _forward_backward(self, batch, jitter_start, jitter_end, final, dc):
        r, t, f = self.r, self.t, self.r.field
        B = batch["position"].shape[0]
        if f.flat.is_cuda:
            # everything enqueued so far (the last optimiser step, the occupancy-grid refresh, the batch) is what the sampling
            # of this step's third render depends on: grad_loss_forward_backward(early=True) waits for this point only
            self._ready_ev = torch.cuda.Event()
            self._ready_ev.record()
        front = self._take_prefetched(batch, jitter_start, jitter_end)
        if front is None:
            front = self._front(batch, jitter_start, jitter_end)
        prep, px, jitter = front["prep"], front.get("px"), front["jitter"]
        ts_all, target = prep["ts"], prep["target_diff"]
        bkgd = ops.bkgd_param_fwd(self.small, f.C) if t.bkgd_is_param else None              # nerf.py:81-88
        colords = None
        if t.train_refractory_period:
            # d loss/d tau needs dI/dt of both renders: carry the tangent forward (value path unchanged)
            from . import jvp
            pos, rot, dpos, drot = jvp.trajectory_jvp(ts_all, self.tab_ts, self.tab_pos, self.tab_quat)
            o, d, od, dd = jvp.raygen_jvp(self.Kinv, px, pos, rot, dpos, drot)
            begun = None
            if self._grad_pending is not None:
                begun = self._begin_with_grad(o, d, jitter, dc)
            colors, colords, opac, ctx = jvp.render_forward(r, o, d, od, dd, jitter, bkgd, training=True, begun=begun,
                                                            device_counts=dc)
        else:
            o, d = front["o"], front["d"]
            if self._grad_pending is not None:
                if front.get("begun") is None:
                    front["begun"] = self._begin_with_grad(o, d, jitter, dc)
                else:
                    self._begin_grad_now()
            colors, opac, depth, ctx = r.forward(o, d, jitter, bkgd, training=True, save=True, begun=front.get("begun"),
                                                 device_counts=dc)
        # a16 + a18: intensity epilogue, validity, Bayer channel, loss and its gradient: two launches (ren_event_diff_loss_*)
        if f.C > 1 and "channel_idx" not in batch:
            raise ValueError("radiance_dim 3 (Bayer sensor) needs batch['channel_idx'] (data.colorize_events)")
        chan = batch["channel_idx"].to(torch.uint8).contiguous() if f.C > 1 else None
        need_param = t.train_contrast_threshold or t.train_refractory_period
        # loss weight x 1 / C^k (robust_e_nerf.py:470-486); the latter from the device block
        L = ops.event_diff_loss(colors, opac, chan, target, t.err_diff, t.w_diff, r.cfg.min_modeled_intensity,
                                use_validity=not t.bkgd_is_param, want_pred=need_param, scale_dev=self._pw_dev(t.pw_diff))
        loss, g_colors, inten = L["loss"], L["g_colors"], L["intensity"]
        i_s, i_e = inten[:B], inten[B:]
        if need_param:
            self._param_grad(batch, L["pred"], "diff", L["valid"])
        if t.train_refractory_period:
            # through the poses: sum_i dL/dI_i * dI_i/dt_i * dt_i/dtau  (start and end renders)
            from . import jvp
            if f.C == 1:                                  # one launch: (2B,) gradients x tangents x [dts_start | dts_end], float64 sum
                jvp.tau_pose_grad(self._tau_grad_dev, prep["dts_pair"], g_colors, colords)
            else:
                ch = self._channel_index(batch, 2)
                idot = self._bayer(colords, ch).double()
                g_ev = self._bayer(g_colors, ch).double()
                self._tau_grad_dev += (g_ev[:B] * idot[:B] * prep["dts_start"]).sum() + (g_ev[B:] * idot[B:] * prep["dts_end"]).sum()
        d_bk = r.backward(ctx, g_colors, final=final, per_ray_bkgd=True)
        if d_bk is not None:
            ops.bkgd_param_grad(d_bk, self.small, self.small_grad)       # += sigmoid(raw) * column sums (d softplus)
        pk = ctx["pk"]                                       # (pk.log: n / n_marched are capacities here, _dc_pass puts the counts in)
        aux = dict(intensity_start=i_s, intensity_end=i_e, n=pk.n, n_marched=pk.n_marched, opacity=opac, rays=2 * B)
        if self.keep_ctx:                                    # tests: the renders' own rays and samples
            aux["ctx"] = ctx
        return loss, aux, pk.log

    def grad_sampling_mode(self) -> str:
        """where Trainer.step places the third render's samples (measured, profiles/NOTES.md):
        "inorder" on the main stream after the l_diff backward -- no l_grad term, early_grad_sampling off, or the fixed-S sampler
                  (one count read, no density pre-pass: nothing to hide, and the side stream's kernels only take CUs from the
                  backward);
        "begun"   front (timestamps, poses, rays, march count pass) enqueued on the side stream inside forward_backward() right
                  after the l_diff render's own count pass, the rest (march write pass, density pre-pass, visibility,
                  compaction, both count reads) beside the l_diff backward -- everything else.
        A third placement exists for experiments only (grad_loss_forward_backward(early=True) without begin_grad_sampling():
        ALL of it beside the l_diff forward / backward): there the pose / ray kernels run while the persistent MLP kernels own
        the chip, and in 4-8 % of such steps an aligned group of 16 rays came out with a slightly wrong rotation (same inputs,
        same kernel; traced to the packed-FP32 code the SLP vectoriser made of the pose arithmetic, which build.py now
        switches off for those files -- profiles/NOTES.md).  Trainer.step never uses it; the "begun" placement runs those
        kernels beside the l_diff render's small sampling kernels and reproduced the in-order sample counts in 1 200 of 1 200
        steps (tools/early_diag.py)."""
        if not (self.t.w_grad > 0) or not self.early_grad_sampling or self.r.cfg.sampler == "uniform" or \
                not self.r.field.flat.is_cuda:
            return "inorder"
        # "merged" (round 6, opt-in: Trainer.grad_sampling): everything in order on ONE stream, but the third render's rays join
        # the l_diff render's in one ray / box test and one march count pass (Renderer.sample_begin_merged)
        if self.grad_sampling == "merged":
            return "merged"
        return "begun"

    def _grad_front(self, batch, jitter_grad, rays_only: bool = False) -> dict:
        """the third render up to (not including) the first host read: supervision timestamps at grad.ts, poses and rays with
        their time derivatives, ray/AABB test, march count pass, scan (robust_e_nerf.py:340-357,383-409).  rays_only: without
        the last three (the "merged" placement marches these rays together with the l_diff render's)"""
        from . import jvp
        t = self.t
        prep = ops.event_prepare(batch, 0.0, 0.0, 0.0, with_grad_ts=True, with_dtau=t.train_refractory_period, ep=self.ep)
        ts_g = prep["ts_grad"]
        ddd = None
        if t.train_refractory_period:                                  # tau moves ts_g: second-order tangent
            pos, rot, dpos, drot, ddrot = jvp.trajectory_jvp2(ts_g, self.tab_ts, self.tab_pos, self.tab_quat)
            o, d, od, dd, ddd = jvp.raygen_jvp2(self.Kinv, batch["position"].contiguous(), pos, rot, dpos, drot, ddrot)
        else:
            pos, rot, dpos, drot = jvp.trajectory_jvp(ts_g, self.tab_ts, self.tab_pos, self.tab_quat)
            o, d, od, dd = jvp.raygen_jvp(self.Kinv, batch["position"].contiguous(), pos, rot, dpos, drot)
        jit = None if jitter_grad is None else jitter_grad.to(torch.float32).contiguous()
        st = None if rays_only else self.r.sample_begin(o, d, jit, True, device_counts=self._dc_mode())
        return dict(prep=prep, o=o, d=d, od=od, dd=dd, ddd=ddd, jit=jit, st=st)

    def begin_grad_sampling(self, batch, jitter_grad=None, merged: bool = False) -> bool:
        """Announce the third render of the step that is ABOUT to run -- call it before forward_backward(); the matching
        grad_loss_forward_backward(batch, jitter_grad, early=True) picks it up.  Its front (timestamps, poses, rays, march
        count pass, scan, read-back of the count into pinned memory) goes to the side stream without blocking the host,
        ordered after everything enqueued on the current stream at THIS call (the last optimiser step, the grid refresh,
        the batch, the jitter); it is enqueued from inside forward_backward(), right after the l_diff render's own front and
        count pass and before the host blocks on that render's sample count, so those kernels fill the gaps that the two
        count reads of the l_diff render leave on the main stream, and the host work of enqueuing them never delays the
        l_diff render (a step that starts on an empty queue -- trainable tau: its Adam group reads d loss / d tau back at the
        end of the step -- is host-bound right there).  Trainer.step does this when the l_grad term is on."""
        if not self.r.field.flat.is_cuda or not (self.t.w_grad > 0):
            return False
        ready = None
        if not merged:
            ready = torch.cuda.Event()
            ready.record()
        self._grad_begun, self._grad_pending = None, (batch, jitter_grad, ready)
        return True

    def _begin_with_grad(self, o, d, jitter, dc):
        """the l_diff render's sample_begin(), with the announced third render's front placed as its mode asks: "begun" -- this
        render's count pass first, then the third render's front on the side stream; "merged" -- both renders' rays through
        one ray / box test and one march count pass, in order on this stream"""
        if self._grad_pending[2] is not None:                     # (an event: "begun")
            begun = self.r.sample_begin(o, d, jitter, True, device_counts=dc)
            self._begin_grad_now()
            return begun
        (batch, jitter_grad, _), self._grad_pending = self._grad_pending, None
        fr = self._grad_front(batch, jitter_grad, rays_only=True)
        if (jitter is None) != (fr["jit"] is None):
            raise ValueError("merged sampling: jitter for both the l_diff renders and the third one, or for neither")
        begun, fr["st"] = self.r.sample_begin_merged([(o, d, jitter), (fr["o"], fr["d"], fr["jit"])], True, device_counts=dc)
        fr["n_host"], fr["key"], fr["merged"] = None, (id(batch), id(jitter_grad)), True
        self._grad_begun = fr
        return begun

    def _begin_grad_now(self):
        pend, self._grad_pending = self._grad_pending, None
        if pend is None:
            return
        batch, jitter_grad, ready = pend
        side = self.side_stream
        side.wait_event(ready)
        with torch.cuda.stream(side):
            fr = self._grad_front(batch, jitter_grad)
            if fr["st"].get("dc") is None:                  # (device-side counts: nothing to read back)
                if self._n_host_grad is None:
                    self._n_host_grad = torch.empty(1, dtype=fr["st"]["total"].dtype).pin_memory()
                self._n_host_grad.copy_(fr["st"]["total"].reshape(1), non_blocking=True)
                fr["ev"] = torch.cuda.Event()
                fr["ev"].record()
        fr["n_host"], fr["key"] = self._n_host_grad, (id(batch), id(jitter_grad))
        self._grad_begun = fr

    def grad_loss_forward_backward(self, batch, jitter_grad=None, final: bool = True, early: bool = False):
        """Log-intensity-GRADIENT loss term (robust_e_nerf.py:340-357,383-409; loss.py:43-57): a third
        render at grad.ts = lerp(diff.start, diff.end, u_grad) carrying d/dt in forward mode, compared
        with the event rate C/dt.  Accumulates gradients; returns (weighted loss term, aux).
        early: pick up the front that begin_grad_sampling() announced before the preceding forward_backward() (timestamps,
        poses, rays, march count pass: already on the side stream) and place the rest of the third render's sampling (march
        write pass, density pre-pass, visibility, compaction and the two host reads of the sample counts -- it depends on the
        field and the occupancy grid but not on the l_diff pass) on the side stream as well, i.e. beside the l_diff backward:
        the host reads then wait for a few small kernels instead of draining the queue twice.  Without a matching front the
        call runs in order.  Same arithmetic, same results.  (early="all": the experimental placement of
        grad_sampling_mode's docstring.)"""
        return self._dc_pass(self._grad_loss_forward_backward, (batch, jitter_grad, final, early))

    def _grad_loss_forward_backward(self, batch, jitter_grad, final, early, dc):
        from . import jvp
        r, t, f = self.r, self.t, self.r.field
        B = batch["position"].shape[0]
        ready, self._ready_ev = self._ready_ev, None
        begun, self._grad_begun, self._grad_pending = self._grad_begun, None, None
        if begun is not None and (not early or begun["key"] != (id(batch), id(jitter_grad))):
            begun = None                                                  # (begin_grad_sampling was for another call)
        if begun is not None and begun.get("merged"):
            early = False                                                 # merged front: the rest follows in order on this stream
        # no matching front: in order, unless the caller asks for the experimental placement (early="all": everything
        # beside the l_diff forward / backward; wrong rays observed there, see grad_sampling_mode)
        early = bool(early) and f.flat.is_cuda and (begun is not None or (early == "all" and ready is not None))
        main = torch.cuda.current_stream() if f.flat.is_cuda else None
        if early and begun is None:
            self.side_stream.wait_event(ready)
        with (torch.cuda.stream(self.side_stream) if early else contextlib.nullcontext()):
            fr = begun if begun is not None else self._grad_front(batch, jitter_grad)
            prep, o, d, od, dd, ddd, jit, st = (fr[k] for k in ("prep", "o", "d", "od", "dd", "ddd", "jit", "st"))
            target = prep["target_grad"]                                                        # loss.py:39-42
            if "ev" in fr:                                                # count pass already ran: its total is in the pinned buffer
                fr["ev"].synchronize()
                st["n0"] = int(fr["n_host"][0])
            pk = r.sample(o, d, jit, True, begun=st, device_counts=dc)
            if early:
                done = torch.cuda.Event()
                done.record()
        if early:
            main.wait_event(done)
            _adopt((prep, o, d, od, dd, ddd, jit, pk), main)
        bkgd = torch.nn.functional.softplus(self.small[: f.C]) if t.bkgd_is_param else None
        colors, colords, opac, ctx = jvp.render_forward(r, o, d, od, dd, jit, bkgd, training=True, pk=pk)
        ch = self._channel_index(batch, 1)
        chan = batch["channel_idx"].to(torch.uint8).contiguous() if f.C > 1 else None
        # a16 of the tangent render in one launch: I = c + eps, I' = c', valid = opacity > 0, d log I / dt = I' / I
        inten, intend, valid, dlog = jvp.rate_epilogue(colors, colords, opac, chan, r.cfg.min_modeled_intensity,
                                                       want_valid=not t.bkgd_is_param)
        loss_sum = jvp.grad_loss_fwd(inten, intend, target, valid, t.err_grad)
        g_i, g_id, loss = jvp.grad_loss_bwd(inten, intend, target, valid, t.err_grad, t.w_grad, loss_sum,
                                            scale_dev=self._pw_dev(t.pw_grad), want_loss=True)
        if t.train_contrast_threshold or t.train_refractory_period:
            self._param_grad(batch, dlog, "grad", valid)
        if t.train_refractory_period:
            # d L/d tau through the pose: dL/dI * dI/dt + dL/dI' * d2I/dt2, times d ts_g/d tau  (per event)
            _, _, colorsdd = jvp.render_forward2(r, o, d, od, dd, ddd, ctx["pk"], bkgd)
            jvp.tau_pose_grad(self._tau_grad_dev, prep["dts_grad"], g_i, intend, g_id, self._bayer(colorsdd, ch).contiguous())
        d_bkgd = jvp.render_backward(r, ctx, self._unbayer(g_i, ch, f.C), self._unbayer(g_id, ch, f.C), final=final)
        if d_bkgd is not None:
            self.small_grad[: f.C] += d_bkgd * torch.sigmoid(self.small[: f.C])
        aux = dict(intensity=inten, dlog_dt=dlog, n=pk.n, rays=B)
        if self.keep_ctx:                                    # tests: the render's own rays, samples, features and tangents
            aux["ctx"] = dict(ctx, colors=colors, colords=colords)
        return loss, aux, pk.log

    def optimizer_step(self, accumulate_grad_batches: int = 1, mean_samples_per_ray: Optional[float] = None):
        """Adam on [hash table | MLPs] (lr default, L2 decay 1e-6: robust_e_nerf.py:786-813) and on the
        background scalar (no decay).  Under data parallelism gradients are summed over ranks (RCCL
        all-reduce of the single flat buffer) and scaled by 1/world inside the Adam kernel."""
        f = self.r.field
        if getattr(f, "n_wn_g", 0):
            f.fold_grads()
        if self.world_size > 1:
            # ONE collective for everything a step sums over the ranks (plus the early slice when dp_overlap is on):
            # [table | MLP | pad | aux = small-parameter grads, C_p ratio grad, d loss / d tau as two floats, samples per ray]
            from . import parallel as P_
            aux = f.aux
            aux[P_.AUX_SMALL: P_.AUX_SMALL + 4] = self.small_grad
            aux[P_.AUX_CT] = self.ct_grad[0]
            hi = self._tau_grad_dev.to(torch.float32)
            aux[P_.AUX_TAU_HI] = hi[0]
            aux[P_.AUX_TAU_LO] = (self._tau_grad_dev - hi.double()).to(torch.float32)[0]
            # (fill_, not `aux[i] = python scalar`: that is a host-to-device copy which drains the whole queue on ROCm --
            # tools/sync_probe.py -- and cost the data-parallel step its launch lead)
            aux.narrow(0, P_.AUX_MEAN_S, 1).fill_(float(mean_samples_per_ray) if mean_samples_per_ray is not None else 0.0)
            self.sync.finish(f.grad_all)
            self.small_grad.copy_(aux[P_.AUX_SMALL: P_.AUX_SMALL + 4])
            self.ct_grad[0] = aux[P_.AUX_CT]
            self._tau_grad_dev[0] = aux[P_.AUX_TAU_HI].double() + aux[P_.AUX_TAU_LO].double()
            self._mean_s_dev = aux[P_.AUX_MEAN_S].clone()
            aux.zero_()
            self.last_collectives = self.sync.reset_count()
        self.step_count += 1
        gs = 1.0 / (self.world_size * accumulate_grad_batches)            # mean over ranks and accumulated micro-batches
        lr = self.t.lr * self.lr_scale
        # step numbers / bias corrections live on the device (ops.HY_*): the same launches serve the eager step and a captured
        # one, whose renders' overflow words raise the skip word instead (Trainer._graph_step)
        hy = self._hyper
        if self.t.train_refractory_period:
            self._tau_adam_steps += 1
        stats = [log.stats for log in self.r._polled_logs] if self._capturing else []
        ops.step_tick(hy, self.t.betas, stats=stats, tick_tau=self.t.train_refractory_period)
        ops.adam_step_dev(f.flat, f.grad, self.m, self.v, hy, lr=lr, betas=self.t.betas, eps=self.t.eps,
                          weight_decay=self.t.weight_decay, grad_scale=gs, zero_grad=True)
        if getattr(f, "n_wn_g", 0):
            f.refresh()
        ops.adam_step_dev(self.small, self.small_grad, self.sm, self.sv, hy, lr=lr, betas=self.t.betas, eps=self.t.eps,
                          weight_decay=0.0, grad_scale=gs, zero_grad=True)
        if self.t.train_refractory_period:
            # float64 scalar, Adam group with lr = tau_max * relative lr (robust_e_nerf.py:804-807);
            # tau = tau_max sigmoid(raw / tau_max)  =>  d tau / d raw = sigmoid'(raw / tau_max): one single-thread launch
            ops.tau_adam_step_dev(self._tau_raw_dev, self._tau_grad_dev, self._tau_adam, float(self.tau_max), hy,
                                  lr=float(self.tau_max) * self.t.relative_lr_refractory_period * self.lr_scale,
                                  betas=self.t.betas, eps=self.t.eps, grad_scale=gs)
        if self.t.train_contrast_threshold:
            ops.adam_step_dev(self.ct, self.ct_grad, self.ct_m, self.ct_v, hy, lr=self.t.lr_contrast_threshold * self.lr_scale,
                              betas=self.t.betas, eps=self.t.eps, weight_decay=0.0, grad_scale=gs, zero_grad=True)
        if self.t.train_refractory_period or self.t.train_contrast_threshold:
            self._refresh_event_params()                     # clamp (:170-185), C_p, tau, 1 / C^k for the next step's kernels

    def _sync_hyper(self):
        """device-side step numbers <- the host's (after loading a checkpoint, or a repeated step); clears the skip word"""
        h = [0.0] * 8
        h[ops.HY_STEP], h[ops.HY_TAU_STEP] = float(self.step_count), float(self._tau_adam_steps)
        self._hyper.copy_(torch.tensor(h, dtype=torch.float64))

    # ---- optimiser state for checkpoints (what ModelCheckpoint keeps under "optimizer_states": scripts/run.py:66-68) ----
    def optimizer_state_dict(self) -> Dict[str, object]:
        cpu = lambda t: t.detach().cpu().clone()
        sd = dict(step_count=self.step_count, exp_avg=cpu(self.m), exp_avg_sq=cpu(self.v), small_exp_avg=cpu(self.sm),
                  small_exp_avg_sq=cpu(self.sv), ct_exp_avg=cpu(self.ct_m), ct_exp_avg_sq=cpu(self.ct_v))
        if self._tau_adam_steps:
            m, v = self._tau_adam.tolist()
            sd["tau_adam"] = dict(step=self._tau_adam_steps, exp_avg=m, exp_avg_sq=v)
        return sd

    def load_optimizer_state_dict(self, sd: Dict[str, object]):
        self.step_count = int(sd["step_count"])
        for dst, key in ((self.m, "exp_avg"), (self.v, "exp_avg_sq"), (self.sm, "small_exp_avg"),
                         (self.sv, "small_exp_avg_sq"), (self.ct_m, "ct_exp_avg"), (self.ct_v, "ct_exp_avg_sq")):
            dst.copy_(sd[key].to(dst.device, dst.dtype))
        if "tau_adam" in sd:
            ta = sd["tau_adam"]
            if "state" in ta:                                # round <= 4 checkpoints: a torch.optim.Adam state dict
                st = next(iter(ta["state"].values()), None)
                ta = dict(step=int(st["step"]), exp_avg=float(st["exp_avg"]), exp_avg_sq=float(st["exp_avg_sq"])) if st else None
            if ta:
                self._tau_adam_steps = int(ta["step"])
                self._tau_adam.copy_(torch.tensor([ta["exp_avg"], ta["exp_avg_sq"]], dtype=torch.float64))
        self._sync_hyper()

    def load_event_params(self, p2n_raw: Optional[torch.Tensor] = None, tau_raw: Optional[torch.Tensor] = None):
        """restore the learned contrast-threshold ratio / refractory period (checkpoint resume)"""
        if p2n_raw is not None:
            self.ct[0] = p2n_raw.detach().reshape(-1)[0].to(self.ct.device, torch.float32)
        if tau_raw is not None:
            self._tau_raw_dev.copy_(tau_raw.detach().to(torch.float64).reshape(1))
        self._refresh_event_params()

    def update_train_batch_size(self, aux, eff_ray_sample_batch_size: int = 1 << 20, accumulate_grad_batches: int = 1,
                                batch_index: int = 0) -> Optional[int]:
        """Dynamic batch size (robust_e_nerf.py:907-950): keep rays x samples per render near the budget.
        mean_S = mean over this step's renders (start, end [, grad]) of n/R, averaged over ranks (C2); budget = eff //
        num_gpus (:63-66).  Returns the new per-rank event batch size, or None when the gradient-accumulation rule of the
        reference skips the update for this micro-batch."""
        from . import parallel
        means = self._render_means(aux)
        gather = None
        if self.world_size > 1:
            if self._mean_s_dev is not None and aux.get("_mean_s_synced"):
                summed = float(self._mean_s_dev)                 # rode along in the gradient all-reduce: one host read
                gather = lambda m: summed / self.world_size
            else:                                                # no optimiser step on this micro-batch (gradient accumulation)
                gather = lambda m: parallel.allgather_mean(m, self.pg)
        budget = parallel.per_rank_budget(eff_ray_sample_batch_size, self.world_size)
        mean, new = parallel.new_train_batch_size(budget, means, gather, accumulate_grad_batches, batch_index)
        return None if new is None else max(1, new)

    @staticmethod
    def _render_means(aux):
        """mean samples per ray of every render of the step (start, end [, grad]), robust_e_nerf.py:909-913"""
        # start / end renders are one batched pass of 2B rays: (n_s / B + n_e / B) / 2 = n / 2B, twice
        means = [aux["n"] / max(aux["rays"], 1)] * 2
        if aux.get("grad") is not None:
            means.append(aux["grad"]["n"] / max(aux["grad"]["rays"], 1))
        return means

    def set_epoch(self, epoch: int, milestones=(20, 30, 36), gamma: float = 0.33):
        """MultiStepLR stepped per epoch (robust_e_nerf.py:818-832, synthetic.yaml:113-128)."""
        self.lr_scale = gamma ** sum(1 for m in milestones if epoch >= m)

    def step(self, batch, jitter_start=None, jitter_end=None, global_step: Optional[int] = None, jitter_grad=None,
             batch_index: Optional[int] = None, accumulate_grad_batches: int = 1):
        """One training batch.  With gradient accumulation (PL `accumulate_grad_batches`): the occupancy grid is
        refreshed on the first micro-batch only (robust_e_nerf.py:375-379), gradients add up over the micro-batches
        and the optimiser steps on the last one with their mean.
        A step whose shape (event count, capacities of its renders, learning-rate factor) repeats is captured in a hipGraph
        and replayed as ONE launch from then on (`use_graph`, _graph_step)."""
        k = max(1, accumulate_grad_batches)
        bi = 0 if batch_index is None else batch_index
        if global_step is not None and bi % k == 0:
            self.r.update_occ_grid(global_step, self.tab_pos)
        if k == 1:
            out = self._graph_step(batch, jitter_start, jitter_end, jitter_grad)
            if out is not None:
                return out
        return self._step_passes(batch, jitter_start, jitter_end, jitter_grad, bi, k)

    def _step_passes(self, batch, jitter_start, jitter_end, jitter_grad, bi: int = 0, k: int = 1, optimizer: bool = True):
        """the loss passes of one batch and (on the last micro-batch) the optimiser step: the part of step() after the
        occupancy-grid refresh -- what a captured step consists of (optimizer=False: the passes only -- under data
        parallelism the captured part ends where the gradient exchange begins)"""
        # only the LAST backward pass of the LAST micro-batch may start the early all-reduce of the fine levels' slice:
        # an earlier pass would reduce it once per micro-batch (the rank-summed slice of micro-batch 1 would be summed
        # over the ranks again with micro-batch 2 on top) and the next scatter would write into a slice in flight
        last = (bi + 1) % k == 0
        mode = self.grad_sampling_mode()
        if mode in ("begun", "merged"):
            self.begin_grad_sampling(batch, jitter_grad, merged=mode == "merged")
        # the third render's sampling runs on the side stream beside this pass's backward: leave it a few CUs
        self.r.bwd_side_cus = self.side_cus if mode == "begun" else 0
        try:
            loss, aux = self.forward_backward(batch, jitter_start, jitter_end, final=last and not (self.t.w_grad > 0))
        finally:
            self.r.bwd_side_cus = 0
        if self.t.w_grad > 0:
            lg, aux_g = self.grad_loss_forward_backward(batch, jitter_grad, final=last, early=mode != "inorder")
            loss = loss + lg
            aux = dict(aux, grad=aux_g)
        if (bi + 1) % k == 0 and optimizer:
            self._optimizer_after_passes(aux, k)
        return loss, aux

    def _optimizer_after_passes(self, aux, k: int = 1):
        mean = None
        if self.world_size > 1:                              # (rides in the gradient all-reduce)
            means = self._render_means(aux)
            mean = sum(means) / len(means)
        self.optimizer_step(k, mean_samples_per_ray=mean)
        aux["_mean_s_synced"] = self.world_size > 1

    # ---- the whole step as one hipGraph launch (the reference's step shape: models/robust_e_nerf.py:301-517) ----
    GRAPH_CACHE = 8

    def _graph_key(self, batch, jitter_start, jitter_end, jitter_grad):
        """what a captured step is specialised to -- or None when this step cannot be captured: device-side counts with known
        capacities for every render (occupancy sampler, one GPU), no gradient accumulation.  The capacities in the key are
        those of a cached graph of the same shape that still fits the learnt counts with a margin (capturing costs ~10 steps:
        a graph is kept while the counts drift by a few per cent), otherwise what Renderer._capacities gives now."""
        r, t = self.r, self.t
        if self.use_graph is False or not r.field.flat.is_cuda or not self.device_counts_ok() or \
                r._spr is None or self._dc_sync or (jitter_end is not None and jitter_start is None):
            return None
        B = batch["position"].shape[0]
        rays = [2 * B] + ([B] if t.w_grad > 0 else [])
        caps = [r._capacities(n) for n in rays]
        if any(c is None for c in caps):
            return None
        sig = tuple(sorted((k, tuple(v.shape), str(v.dtype)) for k, v in batch.items() if isinstance(v, torch.Tensor)))
        shape = (B, float(self.lr_scale), self.grad_sampling_mode(), jitter_start is not None, jitter_grad is not None, sig,
                 t.train_contrast_threshold, t.train_refractory_period, float(t.w_grad))
        need = [tuple(int(n * s * 1.08) + 1024 for s in r._spr) for n in rays]
        for key in self._graphs:
            if key[1:] == shape and all(c >= m and c <= 3 * m + 16384 for kc, km in zip(key[0], need) for c, m in zip(kc, km)):
                return key
        return (tuple(caps),) + shape

    def _graph_step(self, batch, jitter_start, jitter_end, jitter_grad):
        """replay (or capture, the second time a step shape occurs in a row) -> (loss, aux), or None: run the step eagerly"""
        key = self._graph_key(batch, jitter_start, jitter_end, jitter_grad)
        if key is None:
            self._graph_last_key = None
            return None
        sg = self._graphs.get(key)
        if sg is not None and sg["ws_ptr"] != (self.r._bin_ws.data_ptr() if self.r._bin_ws is not None else 0):
            # the staging pool of the binned scatter was reallocated since this graph was captured (an eager step in between
            # needed more, or handed it back): its launches carry the old address
            self._graphs.pop(key)
            sg = None
        if sg is None:
            if self.use_graph is None:                   # auto: a shape has to keep repeating before it is worth a capture
                if self._graph_bad.get(key[1:], 0) >= 3:
                    return None                          # (captured three times, never replayed faster than the eager step: stays eager)
                # (the SHAPE has to repeat -- event count, flags, lr factor; the capacities in the key follow the counts)
                self._graph_streak = self._graph_streak + 1 if (self._graph_last_key or (None,))[1:] == key[1:] else 0
                self._graph_last_key = key               # (the reference's dynamic batch size changes the event count nearly
                if self._graph_streak < 2:               # every step: such a run never captures -- scripts/train.py --batch-size-quantum)
                    # the eager steps in front of a capture are timed: a graph has to beat them to be kept (_capture)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    out = self._step_passes(batch, jitter_start, jitter_end, jitter_grad, optimizer=self.world_size == 1)
                    e1.record()                          # (what the graph will contain: under data parallelism the passes only)
                    self._graph_eager_ev = (key[1:], e0, e1)
                    if self.world_size > 1:
                        self._optimizer_after_passes(out[1])
                    return out
            sg = self._capture(key, batch, jitter_start if jitter_end is None else torch.cat([jitter_start, jitter_end]),
                               jitter_grad)
            if sg is None:
                return None
        self._graph_last_key = key
        # inputs -> the graph's static buffers (skipped for a tensor that already IS the static buffer: graph_inputs())
        if "_pack" in batch and "_pack" in sg["batch"] and batch["_pack"].numel() == sg["batch"]["_pack"].numel():
            if batch["_pack"].data_ptr() != sg["batch"]["_pack"].data_ptr():
                sg["batch"]["_pack"].copy_(batch["_pack"], non_blocking=True)      # (engine.pack_batch: one launch for all fields)
        else:
            for k, dst in sg["batch"].items():
                if k != "_pack" and batch[k].data_ptr() != dst.data_ptr():
                    dst.copy_(batch[k], non_blocking=True)
        B = batch["position"].shape[0]
        parts = [(sg["j0"], jitter_start), (sg["j2"], jitter_grad)] if jitter_end is None else \
            [(sg["j0"][:B], jitter_start), (sg["j0"][B:], jitter_end), (sg["j2"], jitter_grad)]
        for dst, src in parts:
            if dst is not None and src.data_ptr() != dst.data_ptr():
                dst.copy_(src, non_blocking=True)
        for log, _ in sg["passes"]:
            log.arm()
        sg["graph"].replay()
        self.graph_replays += 1
        self._ep_stale = True
        # the counts of this step's renders: a wait for its SAMPLING kernels (pinned words), the rest is still in flight
        over = False
        for log, _ in sg["passes"]:
            v = log.wait()
            over = over or bool(v[1] or v[3])
        if not over:
            for i, (log, aux) in enumerate(sg["passes"]):
                v = log.values
                self.r._learn_counts(log.n_rays, v[0], v[2])
                for a in ((aux, sg["aux"]) if i == 0 else (aux,)):      # (the step's aux is a copy of the first pass's dict)
                    a["n"] = v[2]
                    if "n_marched" in a:
                        a["n_marched"] = v[0]
            if self.world_size > 1:
                # data parallelism: the graph ends where the gradient exchange begins -- all-reduce and optimiser launches
                # are enqueued from here while the graph's backward is still running (they count the step themselves)
                self._optimizer_after_passes(sg["aux"])
                return sg["loss"], sg["aux"]
            self.step_count += 1
            if self.t.train_refractory_period:
                self._tau_adam_steps += 1
            return sg["loss"], sg["aux"]
        # a count did not fit: the graph's optimiser launches saw the skip word and changed nothing.  Clear what its passes
        # left in the gradient buffers and run the step again with host-side counts (the capacities learn from them).
        # (data parallelism: THIS rank repeats its passes by itself, before the collective that its peers are waiting in)
        f = self.r.field
        f.grad_all.zero_()
        if getattr(f, "n_wn_g", 0):
            f.g_mlp.zero_()
        self._gs.zero_()
        if self.world_size == 1:
            self._sync_hyper()
        self.device_count_overflows += 1
        self._dc_sync, self._grad_begun, self._grad_pending = True, None, None
        try:
            return self._step_passes(batch, jitter_start, jitter_end, jitter_grad)
        finally:
            self._dc_sync = False

    def graph_inputs(self, batch, jitter_start=None, jitter_grad=None):
        """the static input buffers of the captured step this call would replay -> (batch dict, jitter_start, jitter_grad) to
        write the NEXT step's inputs into directly (then step() copies nothing), or None when there is no such graph yet"""
        key = self._graph_key(batch, jitter_start, None, jitter_grad)
        sg = self._graphs.get(key) if key is not None else None
        return None if sg is None else (sg["batch"], sg["j0"], sg["j2"])

    def _capture(self, key, batch, jitter_start, jitter_grad):
        r = self.r
        dev = r.field.flat.device
        if len(self._graphs) >= self.GRAPH_CACHE:            # oldest out (its memory stays in the shared pool for the others)
            self._graphs.pop(next(iter(self._graphs)))
        st_batch = pack_batch(batch) if "_pack" in batch else {k: v.clone() for k, v in batch.items() if isinstance(v, torch.Tensor)}
        j0 = jitter_start.to(torch.float32).clone() if jitter_start is not None else None
        j2 = jitter_grad.to(torch.float32).clone() if jitter_grad is not None else None
        caps = key[0]
        if r.cfg.binned_scatter:
            # sized before the capture (nothing (re)allocates inside) and with room to spare: a larger step shape later on
            # then finds it big enough, and the graphs captured so far stay valid (they carry its address)
            need = max(max(c) for c in caps)
            if r._bin_ws is None or r._bin_ws.numel() < ops.hashgrid_bwd_binned_workspace_bytes(need):
                r._binned_workspace(2 * need, dev)
        _ = self.side_stream
        pinned = [torch.empty(4, dtype=torch.int64).pin_memory() for _ in range(len(caps))]
        # (a pool lives as long as a graph that was captured into it: with the last one gone the handle is dead -- torch asserts on
        # a capture into it -- so an empty cache starts a new pool)
        if self._graph_pool is None or not self._graphs or os.environ.get("REN_STEP_GRAPH_POOL") == "own":
            self._graph_pool = torch.cuda.graph_pool_handle()
        g = torch.cuda.CUDAGraph()
        dump = os.environ.get("REN_STEP_GRAPH_DUMP")             # debugging: <prefix><capture number>.dot of every captured step
        if dump:
            g.enable_debug_mode()
        host_state = (self.step_count, self._tau_adam_steps, self._ep_stale)
        r._polled_logs, r._polled_pinned, self._cap_passes, self._capturing = [], pinned, [], True
        self._grad_begun, self._grad_pending = None, None
        ok = False
        try:
            with torch.cuda.graph(g, pool=self._graph_pool):
                loss, aux = self._step_passes(st_batch, j0, None, j2, optimizer=self.world_size == 1)
            ok = True
        except Exception as e:                               # a capture that cannot be made is not an error of the step
            import warnings
            warnings.warn(f"step capture failed ({type(e).__name__}: {e}); this trainer runs eagerly from here on")
            self.use_graph = False
        finally:
            passes = self._cap_passes
            r._polled_logs, r._polled_pinned, self._cap_passes, self._capturing = None, None, None, False
            self.step_count, self._tau_adam_steps, self._ep_stale = host_state       # (nothing ran)
            self._grad_begun, self._grad_pending = None, None
        if not ok:
            return None
        self.graph_captures += 1
        if dump:
            g.debug_dump(f"{dump}{self.graph_captures}.dot")
        sg = dict(graph=g, batch=st_batch, j0=j0, j2=j2, loss=loss, aux=aux, passes=passes,
                  ws_ptr=r._bin_ws.data_ptr() if r._bin_ws is not None else 0)
        # Does the replay beat the eager step?  A captured step with a forked branch (the third render's sampling) replays
        # through a second hardware queue, and on this runtime WHICH queue the graph's internal stream lands on decides
        # whether the replay is faster than the eager launches or 25 % slower (tools/recapture_probe.py: every other capture
        # of the same step; a single-stream graph is slower still at a few million samples).  So a capture is measured before
        # it is used: three replays with the optimiser's skip word raised (parameters and moments untouched, the gradient
        # buffers cleared afterwards) against the eager steps that ran just before it.
        ev = self._graph_eager_ev
        if self.use_graph is None and ev is not None and ev[0] == key[1:]:
            t_eager = ev[1].elapsed_time(ev[2])
            t_graph = self._time_skip_replays(sg)
            sg["ms"] = (t_graph, t_eager)
            if t_graph > 0.97 * t_eager:
                self._graph_bad[key[1:]] = self._graph_bad.get(key[1:], 0) + 1
                self._graph_streak = 1                       # (the next step tries again: up to three attempts per shape)
                return None
        self._graphs[key] = sg
        return sg

    def _time_skip_replays(self, sg, reps: int = 3) -> float:
        """ms per replay of a captured step that changes nothing: skip word raised, gradients cleared afterwards"""
        f = self.r.field
        self._hyper[ops.HY_SKIP: ops.HY_SKIP + 1].fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        sg["graph"].replay()                                 # (first replay of a fresh executable: not timed)
        e0.record()
        for _ in range(reps):
            sg["graph"].replay()
        e1.record()
        f.grad_all.zero_()
        if getattr(f, "n_wn_g", 0):
            f.g_mlp.zero_()
        self._gs.zero_()
        self._sync_hyper()
        if self.t.train_refractory_period or self.t.train_contrast_threshold:
            self._refresh_event_params()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
