"""`arch: mlp`: vanilla-NeRF radiance field (frequency positional encoding + 8 x 256 MLP with skip, sigma
layer, bottleneck, 283 -> 128 -> C colour head) on the HIP dense-layer kernels (csrc/ren_dense.hip).

Replaces ``VanillaNeRFRadianceField`` / ``NerfMLP`` / ``MLP`` / ``SinusoidalEncoder``
(robust_e_nerf/external/mlp.py:26-113,126-205,208-243,246-358) as instantiated at
robust_e_nerf/models/nerf.py:143-160 with configs/train/synthetic.yaml:85-96.  Parameters live in one flat
float32 buffer in the order (and torch nn.Linear layout) of the reference state dict, so
``VanillaField.load(reference_state_dict)`` is a straight copy and data parallelism / Adam work exactly as
for the NGP field (``engine.Trainer``).  ``VanillaRenderer`` plugs the field into the same sampling,
compositing and loss path as the hash-grid field.
"""
from __future__ import annotations

import ctypes
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib, ops
from ._lib import check
from .engine import RenderCfg, Renderer
from .ops import _ptr, _stream

POS_DIM, VIEW_DIM = 63, 27                      # SinusoidalEncoder(3, 0, 10) / (3, 0, 4), mlp.py:208-243
DEPTH, WIDTH, SKIP, WIDTH_COND = 8, 256, 4, 128
ACT_NONE, ACT_SOFTPLUS100, ACT_SOFTPLUS1, ACT_TRUNC_EXP_SEL = 0, 1, 2, 3
ACT_RELU, ACT_SIGMOID, ACT_SOFTPLUS1_SEL, ACT_SHIFTED_SOFTPLUS1_SEL = 4, 5, 6, 7      # the YAML's alternatives (nerf.py:8-29)


def layer_shapes(C: int = 1) -> List[Tuple[str, int, int]]:
    """(reference state-dict stem, out_features, in_features) in parameter-block order."""
    out, fin = [], POS_DIM
    for i in range(DEPTH):
        out.append((f"mlp.base.hidden_layers.{i}", WIDTH, fin))
        fin = WIDTH + POS_DIM if (i % SKIP == 0 and i > 0) else WIDTH          # mlp.py:60-71
    out += [("mlp.sigma_layer.output_layer", 1, WIDTH), ("mlp.bottleneck_layer.output_layer", WIDTH, WIDTH),
            ("mlp.rgb_layer.hidden_layers.0", WIDTH_COND, WIDTH + VIEW_DIM), ("mlp.rgb_layer.output_layer", C, WIDTH_COND)]
    return out


class VanillaField:
    """Flat parameter / gradient buffers with per-layer (weight, bias) views.

    weight_norm (external/mlp.py:303-319: torch.nn.utils.weight_norm on every Linear): the optimiser's buffers `flat` / `grad`
    then hold [v and biases | g of every row | pad], and `eff` / `g_eff` -- what every kernel reads and accumulates into, and
    what the w / b / gw / gb views show -- are separate buffers with W = g v / ||v|| and its gradient, kept in step by
    refresh() (after every parameter change) and fold_grads() (before the optimiser).  Without it they are the same memory."""

    def __init__(self, device, radiance_dim: int = 1, weight_norm: bool = False):
        if radiance_dim not in (1, 3):
            raise NotImplementedError("radiance_dim must be 1 or 3 (robust_e_nerf.py:230-233)")
        self.C = radiance_dim
        self.layers = layer_shapes(radiance_dim)
        self.weight_norm = bool(weight_norm)
        nb = sum(o * i + o for _, o, i in self.layers)
        self.n_block = nb
        self.n_wn_g = sum(o for _, o, _ in self.layers) if self.weight_norm else 0
        n = nb + self.n_wn_g
        self.n_params = n
        n_pad = (n + 3) // 4 * 4
        self.flat = torch.zeros(n_pad, device=device, dtype=torch.float32)
        from .parallel import AUX_FLOATS                         # [parameters | pad | aux]: one all-reduce (parallel.GradSync)
        self.grad_all = torch.zeros(n_pad + AUX_FLOATS, device=device, dtype=torch.float32)
        self.grad = self.grad_all[:n_pad]
        self.aux = self.grad_all[n_pad:]
        if self.weight_norm:
            nb_pad = (nb + 3) // 4 * 4
            self.eff = torch.zeros(nb_pad, device=device, dtype=torch.float32)
            self.g_eff = torch.zeros(nb_pad, device=device, dtype=torch.float32)
            self.raw, self.g_raw = self.flat[:nb], self.grad[:nb]
            self.wn_g, self.g_wn_g = self.flat[nb: n], self.grad[nb: n]
        else:
            self.eff, self.g_eff = self.flat, self.grad
        self.w, self.b, self.gw, self.gb = {}, {}, {}, {}
        self._off, self._g0 = {}, {}
        off, g0, table = 0, 0, []
        for name, o, i in self.layers:
            self.w[name], self.gw[name] = self.eff[off: off + o * i].view(o, i), self.g_eff[off: off + o * i].view(o, i)
            self._off[name], self._g0[name] = off, g0
            table.append((off, o, i, g0))
            off += o * i
            g0 += o
            self.b[name], self.gb[name] = self.eff[off: off + o], self.g_eff[off: off + o]
            off += o
        if self.weight_norm:
            self._wn_table = ops.weight_norm_layers(table)

    def refresh(self):
        if self.weight_norm:
            ops.weight_norm_fwd(self.raw, self.wn_g, self._wn_table, self.eff[: self.n_block])

    def fold_grads(self, zero: bool = True):
        if self.weight_norm:
            ops.weight_norm_bwd(self.raw, self.wn_g, self.g_eff[: self.n_block], self._wn_table, self.g_raw, self.g_wn_g,
                                zero_d_eff=zero)

    def load(self, sd: Dict[str, torch.Tensor]):
        """sd: reference state dict (keys ``mlp.base.hidden_layers.0.weight`` ...; extra keys ignored).  Under weight_norm a
        layer is given as ``.weight_g`` (rows, 1) + ``.weight_v``, or as a plain ``.weight`` (v = W, g = ||W||_row)."""
        dev = self.flat.device
        for name, o, i in self.layers:
            self.b[name].copy_(sd[name + ".bias"].to(dev, torch.float32))
            if not self.weight_norm:
                if name + ".weight_v" in sd:
                    raise ValueError(f"{name}: weight_g / weight_v given for a field built without weight_norm")
                self.w[name].copy_(sd[name + ".weight"].to(dev, torch.float32))
                continue
            v = (sd[name + ".weight_v"] if name + ".weight_v" in sd else sd[name + ".weight"]).to(dev, torch.float32)
            g = sd[name + ".weight_g"].to(dev, torch.float32).reshape(-1) if name + ".weight_g" in sd else v.norm(dim=1)
            off = self._off[name]
            self.raw[off: off + o * i].copy_(v.reshape(-1))
            self.raw[off + o * i: off + o * i + o].copy_(self.b[name])
            self.wn_g[self._g0[name]: self._g0[name] + o].copy_(g)
        self.refresh()

    def state_dict(self, grad: bool = False, trainable: bool = False) -> Dict[str, torch.Tensor]:
        """effective (weight, bias) per layer, or their gradients; trainable=True: what the optimiser holds, under the names the
        reference module's state_dict has (``.weight_g`` / ``.weight_v`` under weight_norm; gradients after fold_grads())"""
        w, b = (self.gw, self.gb) if grad else (self.w, self.b)
        out = {}
        for name, o, i in self.layers:
            if trainable and self.weight_norm:
                raw, gg = (self.g_raw, self.g_wn_g) if grad else (self.raw, self.wn_g)
                off, g0 = self._off[name], self._g0[name]
                out[name + ".weight_g"], out[name + ".weight_v"] = gg[g0: g0 + o].view(o, 1), raw[off: off + o * i].view(o, i)
                out[name + ".bias"] = raw[off + o * i: off + o * i + o]
            else:
                out[name + ".weight"], out[name + ".bias"] = w[name], b[name]
        return out


class FusedField:
    """The whole field as fused launches (csrc/ren_vfield.hip; include/ren_amd.h `ren_vanilla_*`): forward with the
    activations saved in the kernels' fragment layout, backward (data), weight gradients.  mode 1 = bf16 operands / bf16
    saved copies, 6 = three-piece split (fp32 round-off) / fp32 saved copies."""

    def __init__(self, fld: VanillaField, mode: int, n_splits: int = 256, act: int = 0):
        self.field, self.mode, self.n_splits = fld, mode, n_splits
        self.act = int(act)                         # the renderer's activation code: the fused kernels take 0 only and refuse the rest
        self.image = torch.empty(int(_lib.load().ren_vanilla_image_bytes(mode)), device=fld.flat.device, dtype=torch.uint8)
        self._ws = None

    def prep(self):
        """rebuild the weight image from the current parameters"""
        check(_lib.load().ren_vanilla_prep(_ptr(self.field.eff), self.field.C, self.mode, _ptr(self.image, torch.uint8), _stream()),
              "ren_vanilla_prep")

    def new_saved(self, n: int) -> torch.Tensor:
        return torch.empty(int(_lib.load().ren_vanilla_saved_bytes(self.mode, n)), device=self.field.flat.device, dtype=torch.uint8)

    def forward(self, B, full: bool):
        """B: _Buffers (enc / view / sel filled by the encoder) -> B.sigma, B.rgb4 (full), B.saved (when allocated)"""
        check(_lib.load().ren_vanilla_fwd(_ptr(B.enc), 64, _ptr(B.view) if full else None, 32, _ptr(B.sel, torch.uint8),
                                          _ptr(self.field.eff), self.field.C, self.act, _ptr(self.image, torch.uint8), self.mode, B.n,
                                          _ptr(B.saved, torch.uint8) if (full and B.saved is not None) else None, _ptr(B.sigma),
                                          _ptr(B.rgb4) if full else None, _stream()), "ren_vanilla_fwd")

    def _range(self, B, s0: int, m: Optional[int]):
        """(samples in the range, byte offset of its first block inside a slot of B.saved, slot stride of B.saved)"""
        m = B.n - s0 if m is None else m
        esz = 2 if self.mode == 1 else 4
        if s0 % 256:
            raise ValueError("a backward sample range starts at a multiple of 256 samples")
        return m, s0 * 256 * esz, B.saved.numel() // 10

    def backward(self, dz_rgb, dz_sig, B, dz, s0: int = 0, m: Optional[int] = None):
        """backward (data) of the samples s0 .. s0 + m of the pass saved in B (default: all); dz: new_saved(m)"""
        m, off, stride = self._range(B, s0, m)
        check(_lib.load().ren_vanilla_bwd(_ptr(dz_rgb[s0:]), _ptr(dz_sig[s0:]), _ptr(self.image, torch.uint8), self.mode, self.act, m,
                                          B.saved.data_ptr() + off, stride, _ptr(dz, torch.uint8), _stream()), "ren_vanilla_bwd")

    def backward_weight(self, dz_rgb, dz_sig, B, dz, s0: int = 0, m: Optional[int] = None):
        lib = _lib.load()
        m, off, stride = self._range(B, s0, m)
        splits = max(1, min(self.n_splits, (m + 31) // 32))
        need = int(lib.ren_vanilla_bwd_weight_workspace_floats(splits))
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, device=B.enc.device, dtype=torch.float32)
        check(lib.ren_vanilla_bwd_weight(_ptr(dz, torch.uint8), B.saved.data_ptr() + off, stride, _ptr(B.enc[s0:]), 64,
                                         _ptr(B.view[s0:]), 32, _ptr(dz_rgb[s0:]), _ptr(dz_sig[s0:]), self.field.C, self.mode, m,
                                         splits, _ptr(self.field.g_eff), _ptr(self._ws), _stream()), "ren_vanilla_bwd_weight")

    # ---- value + forward-mode tangent as one launch each way (bf16 mode; csrc/ren_vfield.hip: vfield_fwd_jvp / vfield_bwd_jvp)
    def forward_jvp(self, B, encd, viewd, savedd, zsd4, zod4):
        check(_lib.load().ren_vanilla_fwd_jvp(_ptr(B.enc), 64, _ptr(B.view), 32, _ptr(encd), _ptr(viewd), _ptr(B.sel, torch.uint8),
                                              _ptr(self.field.eff), self.field.C, self.act, _ptr(self.image, torch.uint8), self.mode, B.n,
                                              _ptr(B.saved, torch.uint8), _ptr(savedd, torch.uint8), _ptr(B.sigma), _ptr(B.rgb4),
                                              _ptr(zsd4), _ptr(zod4), _stream()), "ren_vanilla_fwd_jvp")

    def backward_jvp_data(self, dz_rgb, dzd_rgb, dz_sig, dzd_sig, B, savedd):
        """reverse pass of forward_jvp for all of B's samples, data gradients -> dz, dzd (fragment slots)"""
        lib = _lib.load()
        dz, dzd = self.new_saved(B.n), self.new_saved(B.n)
        cpl = self.new_saved(B.n) if self.mode != 1 else None        # fp32 family (modes 6 / 3): the tangent side's coupling term into dz
        check(lib.ren_vanilla_bwd_jvp(_ptr(dz_rgb), _ptr(dzd_rgb), _ptr(dz_sig), _ptr(dzd_sig), _ptr(self.image, torch.uint8), self.mode,
                                      self.act, B.n, _ptr(B.saved, torch.uint8), _ptr(savedd, torch.uint8), 0, _ptr(dz, torch.uint8),
                                      _ptr(dzd, torch.uint8), _ptr(cpl, torch.uint8), _stream()), "ren_vanilla_bwd_jvp")
        return dz, dzd

    def backward_jvp_weight(self, dz_rgb, dzd_rgb, dz_sig, dzd_sig, B, savedd, encd, viewd, dz, dzd):
        """both streams' weight gradients: dW += dz^T x (+ db) and dW += dzd^T xd"""
        lib = _lib.load()
        self.backward_weight(dz_rgb, dz_sig, B, dz)
        splits = max(1, min(self.n_splits, (B.n + 31) // 32))
        check(lib.ren_vanilla_bwd_weight_tangent(_ptr(dzd, torch.uint8), _ptr(savedd, torch.uint8), 0, _ptr(encd), 64, _ptr(viewd), 32,
                                                 _ptr(dzd_rgb), _ptr(dzd_sig), self.field.C, self.mode, B.n, splits,
                                                 _ptr(self.field.g_eff), _ptr(self._ws), _stream()), "ren_vanilla_bwd_weight_tangent")

    def decode(self, saved: torch.Tensor, n: int) -> List[torch.Tensor]:
        """fragment layout -> ten row-major (n, 256) float32 tensors: slots 0-7 hidden layers, 8 bottleneck, 9 colour hidden
        layer (first 128 columns) (tests / tools only)"""
        S = 10
        n_blk = saved.numel() // (S * 32 * 256 * (2 if self.mode == 1 else 4))       # whole workgroup passes
        dev = saved.device
        lane = torch.arange(64, device=dev)
        sl, hi = lane & 31, lane >> 5
        if self.mode == 1:
            v = saved.view(torch.bfloat16).view(S, n_blk, 16, 64, 8).float()
            c, j = torch.arange(16, device=dev), torch.arange(8, device=dev)
            feat = ((c >> 1) * 32 + 16 * (c & 1))[:, None, None] + (8 * (j >> 2) + (j & 3))[None, None, :] + 4 * hi[None, :, None]
            samp = sl[None, :, None].expand(16, 64, 8)
        else:
            v = saved.view(torch.float32).view(S, n_blk, 32, 64, 4)
            tq, j = torch.arange(32, device=dev), torch.arange(4, device=dev)
            feat = (8 * tq)[:, None, None] + 4 * hi[None, :, None] + j[None, None, :]
            samp = sl[None, :, None].expand(32, 64, 4)
        out = torch.empty(S, n_blk, 32, 256, device=dev)
        out[:, :, samp.reshape(-1), feat.reshape(-1)] = v.reshape(S, n_blk, -1)
        return [out[l].reshape(n_blk * 32, 256)[:n] for l in range(S)]


class _Buffers:
    """Activation buffers for n samples: row-major [n_pad][ld], zero padding columns.  `fused` (a FusedField): the field runs
    as one launch -- only the encodings, the outputs and (save=True) the fragment-layout activations exist; otherwise one
    row-major buffer per layer (the dense-layer launches: exact-f32 mode, tangent stream)."""

    def __init__(self, n: int, dev, C: int, full: bool, backward: bool, fused: Optional[FusedField] = None, save: bool = False):
        self.n, self.n_pad = n, (n + 31) // 32 * 32
        # no pre-zeroing: every kernel writes all rows < n_pad of its output columns and the encoder writes the
        # zero padding columns; rows >= n only ever feed their own (discarded) output rows
        z = lambda ld: torch.empty(self.n_pad, ld, device=dev, dtype=torch.float32)
        self.enc = z(64)
        self.fused = fused
        self.sel = torch.zeros(self.n_pad, device=dev, dtype=torch.uint8)
        if fused is not None:
            self.cat = None
            self.sigma = torch.empty(self.n_pad, device=dev, dtype=torch.float32)
            self.saved = fused.new_saved(n) if save else None
            if full:
                self.view, self.rgb4 = z(32), z(4)
            return
        self.cat = z(320)                                   # [h4 (256) | enc (63) | 0]
        self.h = {i: (None if i == SKIP else z(WIDTH)) for i in range(DEPTH)}
        self.s4 = z(4)
        if full:
            self.rin = z(288)                               # [bottleneck (256) | view enc (27) | 0]
            self.r = z(WIDTH_COND)
            self.rgb4 = z(4)
        if backward:
            self.dz_rgb, self.dz_sig = z(32), z(32)
            self.dr, self.db = z(WIDTH_COND), z(WIDTH)
            self.dh = [z(WIDTH), z(WIDTH)]

    def out_of(self, i):                                    # (buffer, ld) holding the output of trunk layer i
        return (self.cat, 320) if i == SKIP else (self.h[i], WIDTH)


class VanillaRenderer(Renderer):
    """Renderer over a VanillaField: same sampler / compositing / losses, dense-layer field kernels."""

    def __init__(self, fld: VanillaField, cfg: RenderCfg, n_splits: int = 256):
        super().__init__(fld, cfg)
        self._reuse_prepass_feat = False            # frequency encoding: nothing worth carrying over
        self.n_splits = n_splits
        self._dw_ws = None
        self._fused_fields = {}
        self.fused_field = True                     # csrc/ren_vfield.hip: the field as one launch per pass (matrix-core modes)
        self.fused_tangent = True                   # bf16 mode: value + d/dt as one launch each way (vfield_fwd_jvp / vfield_bwd_jvp)
        # activation set (models/nerf.py:8-29; arch mlp has ONE hidden activation, mlp.py:258): alternatives run on the per-layer
        # launches -- the fused field implements the shipped set only
        if cfg.base_hidden_activation != cfg.head_hidden_activation:
            raise NotImplementedError("arch mlp has one hidden_activation (external/mlp.py:258)")
        self.act_hidden = ACT_RELU if cfg.base_hidden_activation == "relu" else ACT_SOFTPLUS100
        self.act_beta = 0.0 if cfg.base_hidden_activation == "relu" else 100.0          # ren_act_jvp*: beta 0 = relu
        self.act_density = {"shifted_trunc_exp": ACT_TRUNC_EXP_SEL, "softplus": ACT_SOFTPLUS1_SEL,
                            "shifted_softplus": ACT_SHIFTED_SOFTPLUS1_SEL}[cfg.density_activation]
        self.act_radiance = ACT_SIGMOID if cfg.radiance_activation == "sigmoid" else ACT_SOFTPLUS1
        if self._act_code != 0:
            self.fused_field = False
        self.bwd_chunk = 1 << 21                    # samples per backward range of the fused field (multiple of 256)
        # HIP-event timing per kernel family when ops.profile_start() is active (bench.py)
        self._fwd = ops._wrap("dense_fwd", self._fwd)
        self._bwd_data = ops._wrap("dense_bwd_data", self._bwd_data)
        self._bwd_weight = ops._wrap("dense_bwd_weight", self._bwd_weight)
        self._encode = ops._wrap("freq_encode", self._encode)
        self._fused_fwd = ops._wrap("vfield_fwd", self._fused_fwd)
        self._fused_bwd = ops._wrap("vfield_bwd", self._fused_bwd)
        self._fused_dw = ops._wrap("vfield_bwd_weight", self._fused_dw)
        self._fused_fwd_jvp = ops._wrap("vfield_fwd_jvp", self._fused_fwd_jvp)          # value + tangent (the l_grad render)
        self._fused_bwd_jvp = ops._wrap("vfield_bwd_jvp", self._fused_bwd_jvp)
        self._fused_dw_jvp = ops._wrap("vfield_bwd_weight_jvp", self._fused_dw_jvp)
        self._lin = ops._wrap("dense_fwd_tangent", self._lin)
        self._act_fwd = ops._wrap("act_jvp_fwd", self._act_fwd)
        self._act_bwd = ops._wrap("act_jvp_bwd", self._act_bwd)
        self._bwd_weight_nobias = ops._wrap("dense_bwd_weight_tangent", self._bwd_weight_nobias)
        self._encode_tangent = ops._wrap("freq_encode_jvp", self._encode_tangent)

    def dp_early_slice(self):
        return None                                 # no hash table: the 2.4 MB of dense weights go in the one packed all-reduce

    # ---- kernels ------------------------------------------------------------------------------------
    def _dense_mode(self) -> int:
        """matrix-core path of ren_dense_fwd / ren_dense_bwd_data: 0 exact f32 MFMA, 6 split-bf16 at fp32 accuracy,
        1 plain bf16 operands (RenderCfg.mlp_kernels / mlp_bf16, as for arch ngp)"""
        if self.cfg.mlp_bf16:
            return 1
        return 6 if self.cfg.mlp_kernels == "x" else 0

    def _fused_mode(self) -> int:
        """mode of the fused field (ren_vanilla_*): as _dense_mode(), and 3 -- two bf16 pieces per value, three products -- for
        `float32_matmul_precision: high` (RenderCfg.mlp_precision; the per-layer launches run that setting at fp32 accuracy)"""
        mode = self._dense_mode()
        return 3 if mode == 6 and self.cfg.mlp_precision == "high" else mode

    def _fused(self) -> Optional[FusedField]:
        """the fused-field object of the current matrix-core mode with a weight image of the CURRENT parameters, or None
        (exact-f32 mode / fused_field off).  Called once per field evaluation: the image is 9 us to rebuild, which is cheaper
        than tracking every place the flat parameter buffer can change (Adam, load, tests)."""
        mode = self._fused_mode()
        if not self.fused_field or mode == 0:
            return None
        ff = self._fused_fields.get(mode)
        if ff is None:
            ff = self._fused_fields[mode] = FusedField(self.field, mode, self.n_splits, act=self._act_code)
        ff.prep()
        return ff

    def _fused_fwd(self, B, full):
        B.fused.forward(B, full)

    def _fused_fwd_jvp(self, ff, *a):
        ff.forward_jvp(*a)

    def _fused_bwd_jvp(self, ff, *a):
        return ff.backward_jvp_data(*a)

    def _fused_dw_jvp(self, ff, *a):
        ff.backward_jvp_weight(*a)

    def _fused_bwd(self, dz_rgb, dz_sig, B, dz, s0=0, m=None):
        B.fused.backward(dz_rgb, dz_sig, B, dz, s0, m)

    def _fused_dw(self, dz_rgb, dz_sig, B, dz, s0=0, m=None):
        B.fused.backward_weight(dz_rgb, dz_sig, B, dz, s0, m)

    def _fwd(self, X, ldx, name, act, Y, ldy, n, sel=None):
        f = self.field
        o, i = f.w[name].shape
        check(_lib.load().ren_dense_fwd(_ptr(X), ldx, _ptr(f.w[name]), _ptr(f.b[name]), o, i, act | (self._dense_mode() << 8),
                                        _ptr(sel, torch.uint8) if sel is not None else None, _ptr(Y), ldy, n, _stream()),
              "ren_dense_fwd")

    def _bwd_data(self, dZ, ldz, name, n_store, prev_act, Yprev, ldyp, accumulate, dX, ldx, n):
        f = self.field
        o, i = f.w[name].shape
        check(_lib.load().ren_dense_bwd_data(_ptr(dZ), ldz, _ptr(f.w[name]), o, i, n_store, prev_act | (self._dense_mode() << 8),
                                             _ptr(Yprev) if Yprev is not None else None, ldyp, int(accumulate),
                                             _ptr(dX), ldx, n, _stream()), "ren_dense_bwd_data")

    def _bwd_weight(self, dZ, ldz, X, ldx, name, n):
        f, lib = self.field, _lib.load()
        o, i = f.w[name].shape
        splits = max(1, min(self.n_splits, (n + 31) // 32))      # one workgroup (= one slab) per split
        need = int(lib.ren_dense_bwd_weight_workspace_floats(o, i, splits))
        if self._dw_ws is None or self._dw_ws.numel() < need:
            self._dw_ws = None
            self._dw_ws = torch.empty(need, device=dZ.device, dtype=torch.float32)
        check(lib.ren_dense_bwd_weight(_ptr(dZ), ldz, _ptr(X), ldx, o, i, n, splits | (self._dense_mode() << 16), _ptr(f.gw[name]), _ptr(f.gb[name]),
                                       _ptr(self._dw_ws), _stream()), "ren_dense_bwd_weight")

    def _encode(self, B: _Buffers, full: bool, *, rays=None, samples=None, x_world=None, dirs=None):
        n = B.n
        o, d = rays if rays is not None else (None, None)
        ri, ts, te = samples if samples is not None else (None, None, None)
        check(_lib.load().ren_freq_encode(ctypes.byref(self.scene), _ptr(x_world), _ptr(dirs), _ptr(o), _ptr(d),
                                          _ptr(ri, torch.int32), _ptr(ts), _ptr(te), n, _ptr(B.enc), 64,
                                          _ptr(B.cat) if B.cat is not None else None, 320, 256,
                                          (_ptr(B.view) if B.fused is not None else _ptr(B.rin)) if full else None,
                                          32 if B.fused is not None else 288, 0 if B.fused is not None else 256,
                                          _ptr(B.sel, torch.uint8), _stream()),
              "ren_freq_encode")

    def _trunk(self, B: _Buffers):
        n = B.n
        if B.fused is not None:
            self._fused_fwd(B, False)
            return B.sigma[:n]
        X, ldx = B.enc, 64
        for i in range(DEPTH):                                             # mlp.py:99-113
            Y, ldy = B.out_of(i)
            self._fwd(X, ldx, f"mlp.base.hidden_layers.{i}", self.act_hidden, Y, ldy, n)
            X, ldx = Y, ldy
        self._fwd(B.h[DEPTH - 1], WIDTH, "mlp.sigma_layer.output_layer", self.act_density, B.s4, 4, n, sel=B.sel)
        return B.s4[:n, 0].contiguous()

    def _field_eval(self, B: _Buffers, full: bool):
        if B.fused is not None and full:
            self._fused_fwd(B, True)
            return B.rgb4[:B.n, :self.field.C].contiguous(), B.sigma[:B.n]
        sigma = self._trunk(B)
        if not full:
            return None, sigma
        n, C = B.n, self.field.C
        self._fwd(B.h[DEPTH - 1], WIDTH, "mlp.bottleneck_layer.output_layer", ACT_NONE, B.rin, 288, n)
        self._fwd(B.rin, 288, "mlp.rgb_layer.hidden_layers.0", self.act_hidden, B.r, WIDTH_COND, n)
        self._fwd(B.r, WIDTH_COND, "mlp.rgb_layer.output_layer", self.act_radiance, B.rgb4, 4, n)
        return B.rgb4[:n, :C].contiguous(), sigma

    # ---- Renderer hooks ---------------------------------------------------------------------------------
    def _density_stream(self, o, d, samples, n, return_feat: bool = False, chunk: int = 1 << 21):
        """no-grad density of the marched samples (the sampler's sigma_fn pre-pass), in chunks of whole 32-sample blocks: the
        trunk keeps ~8.7 KB of activations per sample, and a first step through an empty occupancy grid can march 60 M of
        them (the rendered-sample budget of the dynamic batch size does not bound the marched count)"""
        tr = self._fused()
        if n <= chunk:
            B = _Buffers(n, o.device, self.field.C, full=False, backward=False, fused=tr)
            self._encode(B, False, rays=(o, d), samples=samples)
            return self._trunk(B)
        ri, ts, te = samples
        out = torch.empty(n, device=o.device, dtype=torch.float32)
        for s0 in range(0, n, chunk):
            e0 = min(s0 + chunk, n)
            B = _Buffers(e0 - s0, o.device, self.field.C, full=False, backward=False, fused=tr)
            self._encode(B, False, rays=(o, d), samples=(ri[s0:e0], ts[s0:e0], te[s0:e0]))
            out[s0:e0] = self._trunk(B)
            del B
        return out

    def _field_forward(self, o, d, pk, save):
        B = _Buffers(pk.n, o.device, self.field.C, full=True, backward=False, fused=self._fused(), save=save)
        self._encode(B, True, rays=(o, d), samples=(pk.ray_indices, pk.t_starts, pk.t_ends))
        rgb, sigma = self._field_eval(B, True)
        return rgb, sigma, dict(buffers=B if save else None)

    def _field_backward(self, ctx, d_rgb, d_sig, final: bool = False):
        B, n, C = ctx["buffers"], ctx["pk"].n, self.field.C
        dev = d_rgb.device
        z = lambda ld: torch.empty(B.n_pad, ld, device=dev, dtype=torch.float32)
        dz_rgb, dz_sig = z(32), z(32)
        check(_lib.load().ren_vanilla_heads_bwd(_ptr(d_rgb.contiguous()), _ptr(ctx["rgb"]), _ptr(d_sig.contiguous()),
                                                _ptr(ctx["sigma"]), n, C, self._act_code, _ptr(dz_rgb), _ptr(dz_sig), _stream()),
              "ren_vanilla_heads_bwd")
        if B.fused is not None:
            # in ranges of bwd_chunk samples: the pre-activation gradients (as large as the saved activations) only ever
            # exist for one range
            for s0 in range(0, n, self.bwd_chunk):
                m = min(self.bwd_chunk, n - s0)
                dz = B.fused.new_saved(m)
                self._fused_bwd(dz_rgb, dz_sig, B, dz, s0, m)
                self._fused_dw(dz_rgb, dz_sig, B, dz, s0, m)
                del dz
            return
        dr, db, dh = z(WIDTH_COND), z(WIDTH), [z(WIDTH), z(WIDTH)]
        h7 = B.h[DEPTH - 1]
        # colour head: 128 -> C, [bottleneck | view] -> 128, bottleneck 256 -> 256 (no activation)
        self._bwd_weight(dz_rgb, 32, B.r, WIDTH_COND, "mlp.rgb_layer.output_layer", n)
        self._bwd_data(dz_rgb, 32, "mlp.rgb_layer.output_layer", WIDTH_COND, self.act_hidden, B.r, WIDTH_COND, False, dr,
                       WIDTH_COND, n)
        self._bwd_weight(dr, WIDTH_COND, B.rin, 288, "mlp.rgb_layer.hidden_layers.0", n)
        self._bwd_data(dr, WIDTH_COND, "mlp.rgb_layer.hidden_layers.0", WIDTH, ACT_NONE, None, 0, False, db, WIDTH, n)
        self._bwd_weight(db, WIDTH, h7, WIDTH, "mlp.bottleneck_layer.output_layer", n)
        self._bwd_data(db, WIDTH, "mlp.bottleneck_layer.output_layer", WIDTH, ACT_NONE, None, 0, False, dh[0], WIDTH, n)
        # sigma layer joins at h7; its data gradient is accumulated, then the trunk activation derivative applied
        self._bwd_weight(dz_sig, 32, h7, WIDTH, "mlp.sigma_layer.output_layer", n)
        self._bwd_data(dz_sig, 32, "mlp.sigma_layer.output_layer", WIDTH, self.act_hidden, h7, WIDTH, True, dh[0], WIDTH, n)
        cur = 0
        for i in range(DEPTH - 1, -1, -1):
            name = f"mlp.base.hidden_layers.{i}"
            if i == 0:
                X, ldx = B.enc, 64
            else:
                X, ldx = B.out_of(i - 1)
            self._bwd_weight(dh[cur], WIDTH, X, ldx, name, n)
            if i > 0:
                self._bwd_data(dh[cur], WIDTH, name, WIDTH, self.act_hidden, X, ldx, False, dh[1 - cur], WIDTH, n)
                cur = 1 - cur

    # ---- forward-mode tangent (d/dt) through the field and its reverse pass: the log-intensity-gradient loss -----
    # Same construction as the NGP path (csrc/ren_mlp_jvp.hip): per layer z = W a + b, zd = W ad; y = sp(z),
    # yd = s zd with s = sp'(z) recovered from the output; backward of (dy, dyd): dz = dy s + dyd zd s', dzd = dyd s,
    # dW += dz^T a + dzd^T ad, da = dz W, dad = dzd W.  Every GEMM is a ren_dense_* launch (value and tangent share the
    # weights: the tangent stream is a second launch without bias); the encodings' time derivatives, the activation
    # algebra and the output heads are HIP kernels on the same row-major buffers (ren_freq_encode_jvp, ren_act_jvp*,
    # ren_vanilla_heads_jvp / _bwd_jvp).
    def _lin(self, X, ldx, name, Y, ldy, n):
        """Y = X W^T (no bias, no activation): the tangent stream of a layer"""
        f = self.field
        o, i = f.w[name].shape
        check(_lib.load().ren_dense_fwd(_ptr(X), ldx, _ptr(f.w[name]), None, o, i, ACT_NONE | (self._dense_mode() << 8), None,
                                        _ptr(Y), ldy, n, _stream()), "ren_dense_fwd")

    def _act_fwd(self, Y, ldy, Zd, beta, Yd, ldyd, width, rows):
        check(_lib.load().ren_act_jvp_fwd(_ptr(Y), ldy, _ptr(Zd), width, ctypes.c_float(beta), _ptr(Yd), ldyd, rows, width,
                                          _stream()), "ren_act_jvp_fwd")

    def _act_bwd(self, gy, gyd, Y, ldy, Zd, beta, width, rows):
        gz, gzd = torch.empty_like(gy), torch.empty_like(gyd)
        check(_lib.load().ren_act_jvp_bwd(_ptr(gy), _ptr(gyd), _ptr(Y), ldy, _ptr(Zd), ctypes.c_float(beta), _ptr(gz), _ptr(gzd),
                                          rows, width, _stream()), "ren_act_jvp_bwd")
        return gz, gzd

    def _bwd_weight_nobias(self, dZ, ldz, X, ldx, name, n):
        """dW += dZ^T X only (the tangent stream has no bias)"""
        f, lib = self.field, _lib.load()
        o, i = f.w[name].shape
        splits = max(1, min(self.n_splits, (n + 31) // 32))
        need = int(lib.ren_dense_bwd_weight_workspace_floats(o, i, splits))
        if self._dw_ws is None or self._dw_ws.numel() < need:
            self._dw_ws = None
            self._dw_ws = torch.empty(need, device=dZ.device, dtype=torch.float32)
        check(lib.ren_dense_bwd_weight(_ptr(dZ), ldz, _ptr(X), ldx, o, i, n, splits | (self._dense_mode() << 16), _ptr(f.gw[name]), None,
                                       _ptr(self._dw_ws), _stream()), "ren_dense_bwd_weight")

    def _encode_tangent(self, B, o, d, od, dd, ddd, pk, order, enc, cat, view):
        """`order`-th time derivative of the position / view encodings of the packed samples (csrc/ren_jvp2.hip):
        enc (n_pad, 64), copy at cat[:, 256:320], view features at view[:, 256:288]"""
        if ddd is None:
            ddd = torch.zeros_like(dd)
        check(_lib.load().ren_freq_encode_jvp(ctypes.byref(self.scene), _ptr(o), _ptr(d), _ptr(od), _ptr(dd), _ptr(ddd),
                                              _ptr(pk.ray_indices, torch.int32), _ptr(pk.t_starts), _ptr(pk.t_ends), B.n, order,
                                              _ptr(enc), 64, _ptr(cat), 320, 256, _ptr(view), 288, 256, _stream()),
              "ren_freq_encode_jvp")

    def _field_forward_jvp(self, o, d, od, dd, pk, ddd=None):
        """value + d/dt (+ d2/dt2 when the rays carry `ddd`: second-order tangent, forward only -- d loss / d tau of the
        log-intensity-gradient loss, engine.Trainer.grad_loss_forward_backward) of (rgb, sigma) for the packed samples.
        -> rgb, rgbd, sigma, sigmad, T           (first order; T = what _field_backward_jvp needs)
        -> rgb, rgbd, rgbdd, sigma, sigmad, sigmadd   (second order)"""
        n, C, dev = pk.n, self.field.C, o.device
        second = ddd is not None
        lib = _lib.load()
        ff = self._fused() if (self.fused_tangent and not second and self._act_code == 0 and self._fused_mode() in (1, 3, 6)) else None
        if ff is not None:
            # first order: value + tangent through all twelve layers in ONE launch each way (bf16 mode) / two (fp32 round-off mode)
            B = _Buffers(n, dev, C, full=True, backward=False, fused=ff, save=True)
            self._encode(B, True, rays=(o, d), samples=(pk.ray_indices, pk.t_starts, pk.t_ends))
            zf = lambda ld: torch.empty(B.n_pad, ld, device=dev, dtype=torch.float32)
            encd, viewd, zsd4, zod4 = zf(64), zf(32), zf(4), zf(4)
            check(lib.ren_freq_encode_jvp(ctypes.byref(self.scene), _ptr(o), _ptr(d), _ptr(od), _ptr(dd), _ptr(dd),
                                          _ptr(pk.ray_indices, torch.int32), _ptr(pk.t_starts), _ptr(pk.t_ends), n, 1, _ptr(encd), 64,
                                          None, 0, 0, _ptr(viewd), 32, 0, _stream()), "ren_freq_encode_jvp")
            savedd = ff.new_saved(n)
            self._fused_fwd_jvp(ff, B, encd, viewd, savedd, zsd4, zod4)
            rgb, sigma = B.rgb4[:n, :C].contiguous(), B.sigma[:n]
            rgbd, sigmad = torch.empty(n, C, device=dev), torch.empty(n, device=dev)
            check(lib.ren_vanilla_heads_jvp(_ptr(rgb), _ptr(sigma), _ptr(zod4), None, _ptr(zsd4), None, n, C, self._act_code, _ptr(rgbd),
                                            None, _ptr(sigmad), None, _stream()), "ren_vanilla_heads_jvp")
            return rgb, rgbd, sigma, sigmad, dict(fused=ff, buffers=B, savedd=savedd, encd=encd, viewd=viewd, zod=zod4, zsd=zsd4)
        B = _Buffers(n, dev, C, full=True, backward=False)
        self._encode(B, True, rays=(o, d), samples=(pk.ray_indices, pk.t_starts, pk.t_ends))
        rgb, sigma = self._field_eval(B, True)
        z = lambda ld: torch.empty(B.n_pad, ld, device=dev, dtype=torch.float32)
        encd, catd, rind = z(64), z(320), z(288)
        self._encode_tangent(B, o, d, od, dd, ddd, pk, 1, encd, catd, rind)
        if second:
            encdd, catdd, rindd = z(64), z(320), z(288)
            self._encode_tangent(B, o, d, od, dd, ddd, pk, 2, encdd, catdd, rindd)
        T = dict(encd=encd, catd=catd, zd={}, yd={})
        Xd, ldx = encd, 64
        if second:
            Xdd = encdd
        for i in range(DEPTH):
            name = f"mlp.base.hidden_layers.{i}"
            Y, ldy = B.out_of(i)
            zd = z(WIDTH)
            self._lin(Xd, ldx, name, zd, WIDTH, n)
            yd, ldyd = (catd, 320) if i == SKIP else (z(WIDTH), WIDTH)
            if second:
                zdd = z(WIDTH)
                self._lin(Xdd, ldx, name, zdd, WIDTH, n)
                ydd = catdd if i == SKIP else z(WIDTH)
                check(lib.ren_act_jvp2_fwd(_ptr(Y), ldy, _ptr(zd), _ptr(zdd), WIDTH, ctypes.c_float(self.act_beta), _ptr(yd), ldyd,
                                           _ptr(ydd), ldyd, B.n_pad, WIDTH, _stream()), "ren_act_jvp2_fwd")
                Xdd = ydd
            else:
                self._act_fwd(Y, ldy, zd, self.act_beta, yd, ldyd, WIDTH, B.n_pad)
            T["zd"][i], T["yd"][i] = zd, yd
            Xd, ldx = yd, ldyd
        h7d = T["yd"][DEPTH - 1]
        s4d = z(4)
        self._lin(h7d, WIDTH, "mlp.sigma_layer.output_layer", s4d, 4, n)
        self._lin(h7d, WIDTH, "mlp.bottleneck_layer.output_layer", rind, 288, n)
        zrd, rd, zod = z(WIDTH_COND), z(WIDTH_COND), z(4)
        self._lin(rind, 288, "mlp.rgb_layer.hidden_layers.0", zrd, WIDTH_COND, n)
        s4dd = zodd = rgbdd = sigmadd = None
        if second:
            h7dd = Xdd
            s4dd, zrdd, rdd, zodd = z(4), z(WIDTH_COND), z(WIDTH_COND), z(4)
            self._lin(h7dd, WIDTH, "mlp.sigma_layer.output_layer", s4dd, 4, n)
            self._lin(h7dd, WIDTH, "mlp.bottleneck_layer.output_layer", rindd, 288, n)
            self._lin(rindd, 288, "mlp.rgb_layer.hidden_layers.0", zrdd, WIDTH_COND, n)
            check(lib.ren_act_jvp2_fwd(_ptr(B.r), WIDTH_COND, _ptr(zrd), _ptr(zrdd), WIDTH_COND, ctypes.c_float(self.act_beta), _ptr(rd),
                                       WIDTH_COND, _ptr(rdd), WIDTH_COND, B.n_pad, WIDTH_COND, _stream()), "ren_act_jvp2_fwd")
            self._lin(rdd, WIDTH_COND, "mlp.rgb_layer.output_layer", zodd, 4, n)
            rgbdd, sigmadd = torch.empty(n, C, device=dev), torch.empty(n, device=dev)
        else:
            self._act_fwd(B.r, WIDTH_COND, zrd, self.act_beta, rd, WIDTH_COND, WIDTH_COND, B.n_pad)
        self._lin(rd, WIDTH_COND, "mlp.rgb_layer.output_layer", zod, 4, n)
        rgbd, sigmad = torch.empty(n, C, device=dev), torch.empty(n, device=dev)
        check(lib.ren_vanilla_heads_jvp(_ptr(rgb), _ptr(sigma), _ptr(zod), _ptr(zodd), _ptr(s4d), _ptr(s4dd), n, C, self._act_code, _ptr(rgbd),
                                        _ptr(rgbdd), _ptr(sigmad), _ptr(sigmadd), _stream()), "ren_vanilla_heads_jvp")
        if second:
            return rgb, rgbd, rgbdd, sigma, sigmad, sigmadd
        T.update(rind=rind, zrd=zrd, rd=rd, zod=zod, zsd=s4d, buffers=B)
        return rgb, rgbd, sigma, sigmad, T

    def _field_backward_jvp(self, T, pk, rgb, sigma, d_rgb, d_rgbd, d_sig, d_sigd):
        B, n, C = T["buffers"], pk.n, self.field.C
        dev = rgb.device
        z = lambda ld: torch.empty(B.n_pad, ld, device=dev, dtype=torch.float32)
        dz_rgb, dzd_rgb, dz_sig, dzd_sig = z(32), z(32), z(32), z(32)
        check(_lib.load().ren_vanilla_heads_bwd_jvp(_ptr(d_rgb.contiguous()), _ptr(d_rgbd.contiguous()), _ptr(d_sig.contiguous()),
                                                    _ptr(d_sigd.contiguous()), _ptr(rgb), _ptr(sigma), _ptr(T["zod"]), _ptr(T["zsd"]),
                                                    n, C, self._act_code, _ptr(dz_rgb), _ptr(dzd_rgb), _ptr(dz_sig), _ptr(dzd_sig), _stream()),
              "ren_vanilla_heads_bwd_jvp")
        if T.get("fused") is not None:
            dz, dzd = self._fused_bwd_jvp(T["fused"], dz_rgb, dzd_rgb, dz_sig, dzd_sig, B, T["savedd"])
            self._fused_dw_jvp(T["fused"], dz_rgb, dzd_rgb, dz_sig, dzd_sig, B, T["savedd"], T["encd"], T["viewd"], dz, dzd)
            return
        h7, h7d = B.h[DEPTH - 1], T["yd"][DEPTH - 1]

        def lin_bwd(gz, gzd, ldz, name, X, ldx, Xd, ldxd, n_store, into=None):
            """dW (+ db from the value stream) and -> (gX, gXd); `into`: accumulate onto an existing pair"""
            self._bwd_weight(gz, ldz, X, ldx, name, n)
            self._bwd_weight_nobias(gzd, ldz, Xd, ldxd, name, n)
            gx, gxd = into if into is not None else (z(n_store), z(n_store))
            self._bwd_data(gz, ldz, name, n_store, ACT_NONE, None, 0, into is not None, gx, n_store, n)
            self._bwd_data(gzd, ldz, name, n_store, ACT_NONE, None, 0, into is not None, gxd, n_store, n)
            return gx, gxd

        gr, grd = lin_bwd(dz_rgb, dzd_rgb, 32, "mlp.rgb_layer.output_layer", B.r, WIDTH_COND, T["rd"], WIDTH_COND, WIDTH_COND)
        gzr, gzrd = self._act_bwd(gr, grd, B.r, WIDTH_COND, T["zrd"], self.act_beta, WIDTH_COND, B.n_pad)
        gb, gbd = lin_bwd(gzr, gzrd, WIDTH_COND, "mlp.rgb_layer.hidden_layers.0", B.rin, 288, T["rind"], 288, WIDTH)
        gy, gyd = lin_bwd(gb, gbd, WIDTH, "mlp.bottleneck_layer.output_layer", h7, WIDTH, h7d, WIDTH, WIDTH)
        lin_bwd(dz_sig, dzd_sig, 32, "mlp.sigma_layer.output_layer", h7, WIDTH, h7d, WIDTH, WIDTH, into=(gy, gyd))
        for i in range(DEPTH - 1, -1, -1):
            name = f"mlp.base.hidden_layers.{i}"
            Y, ldy = B.out_of(i)
            gz, gzd = self._act_bwd(gy, gyd, Y, ldy, T["zd"][i], self.act_beta, WIDTH, B.n_pad)
            if i == 0:
                X, ldx, Xd, ldxd = B.enc, 64, T["encd"], 64
            else:
                (X, ldx) = B.out_of(i - 1)
                Xd, ldxd = (T["catd"], 320) if i - 1 == SKIP else (T["yd"][i - 1], WIDTH)
            if i > 0:
                gy, gyd = lin_bwd(gz, gzd, WIDTH, name, X, ldx, Xd, ldxd, WIDTH)
            else:
                self._bwd_weight(gz, WIDTH, X, ldx, name, n)
                self._bwd_weight_nobias(gzd, WIDTH, Xd, ldxd, name, n)

    def query_density(self, x_world: torch.Tensor) -> torch.Tensor:
        """VanillaNeRFRadianceField.query_density (mlp.py:343-347) for arbitrary world points."""
        n = x_world.shape[0]
        chunk = 1 << 21                                    # an occupancy refresh queries up to 256^3 cells: 8.7 KB of
        tr = self._fused()
        out = torch.empty(n, device=x_world.device, dtype=torch.float32)     # trunk activations per point, so in pieces
        for s0 in range(0, n, chunk):
            xs = x_world[s0: s0 + chunk].contiguous()
            B = _Buffers(xs.shape[0], x_world.device, self.field.C, full=False, backward=False, fused=tr)
            self._encode(B, False, x_world=xs)
            out[s0: s0 + chunk] = self._trunk(B)
            del B
        return out

    def query(self, x_world: torch.Tensor, dirs: torch.Tensor):
        """field(x, d) -> (rgb (n, C), sigma (n,), buffers) for free-standing points (mlp.py:349-358)."""
        n = x_world.shape[0]
        B = _Buffers(n, x_world.device, self.field.C, full=True, backward=False, fused=self._fused(), save=True)
        self._encode(B, True, x_world=x_world.contiguous(), dirs=dirs.contiguous())
        rgb, sigma = self._field_eval(B, True)
        return rgb, sigma, B
