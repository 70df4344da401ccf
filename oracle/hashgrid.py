"""Oracle: multi-resolution hash-grid encoding (tiny-cuda-nn ``HashGrid``, ``Linear``).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Reference call site: ``robust_e_nerf/external/ngp.py:166-170`` builds
``tcnn.Encoding(n_input_dims=3, encoding_config=pos_encoding_config, dtype=float32)``
with the config of ``configs/train/synthetic.yaml:62-69``.  tiny-cuda-nn is an
un-vendored dependency (``environment.yml:31``, unpinned git master), so this is a
restatement of its published algorithm (``include/tiny-cuda-nn/encodings/grid.h``):

  level l:  scale_l = exp2f(l * log2f(per_level_scale)) * base_resolution - 1      (float32)
            res_l   = ceilf(scale_l) + 1
            size_l  = next_multiple(res_l^3, 8), then by grid type (``otype``, configs/train/synthetic.yaml:63):
                      HashGrid  min(size_l, 2^log2_hashmap_size);  DenseGrid  as it is;  TiledGrid  min(size_l, base_resolution^3)
  sample :  pos = fmaf(scale_l, x, 0.5); cell = floorf(pos); w = pos - cell
            corner index (``grid_index``): index = 0, stride = 1; for each dimension WHILE stride <= size_l:
            index += cell_d * stride, stride *= res_l  (x fastest; a dimension whose stride exceeds the level size is
            dropped -- a tiled level with res_l^2 > size_l ignores z); HashGrid with size_l < stride:
            index = (cx*1) ^ (cy*2654435761) ^ (cz*805459861) in uint32; finally ``% size_l``
            feature = sum_c prod_d (bit_d ? w_d : 1 - w_d) * table[offset_l + idx_c]
  output :  (n, L*F) level-major; params flat level-major -> entry -> feature.

PARITY UNPINNED (no reference test / golden vector exists for this dependency).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List

import numpy as np
import torch

PRIME_Y = 2654435761
PRIME_Z = 805459861
MASK32 = 0xFFFFFFFF


@dataclass(frozen=True)
class HashGridSpec:
    n_levels: int = 16
    n_features_per_level: int = 2
    log2_hashmap_size: int = 19
    base_resolution: int = 16
    per_level_scale: float = 1.4472692012786865
    otype: str = "HashGrid"
    # derived
    scales: tuple = ()
    resolutions: tuple = ()
    sizes: tuple = ()
    offsets: tuple = ()
    hashed: tuple = ()

    @property
    def n_output_dims(self) -> int:
        return self.n_levels * self.n_features_per_level

    @property
    def n_entries(self) -> int:
        return self.offsets[-1] + self.sizes[-1]

    @property
    def n_params(self) -> int:
        return self.n_entries * self.n_features_per_level


def make_spec(
    n_levels: int = 16,
    n_features_per_level: int = 2,
    log2_hashmap_size: int = 19,
    base_resolution: int = 16,
    per_level_scale: float = 1.4472692012786865,
    otype: str = "HashGrid",
) -> HashGridSpec:
    """Level table in float32 arithmetic, as tcnn computes it.  otype: HashGrid | DenseGrid | TiledGrid."""
    assert otype in ("HashGrid", "DenseGrid", "TiledGrid"), otype
    log2_pls = np.log2(np.float32(per_level_scale)).astype(np.float32)
    scales, ress, sizes, offsets, hashed = [], [], [], [], []
    offset = 0
    for lvl in range(n_levels):
        scale = np.float32(
            np.exp2(np.float32(lvl) * log2_pls).astype(np.float32) * np.float32(base_resolution)
            - np.float32(1.0)
        )
        res = int(np.ceil(scale)) + 1
        dense = res ** 3
        size = (min(dense, 2 ** 31 - 1) + 7) // 8 * 8
        if otype == "HashGrid":
            size = min(size, 1 << log2_hashmap_size)
        elif otype == "TiledGrid":
            size = min(size, base_resolution ** 3)
        scales.append(float(scale))
        ress.append(res)
        sizes.append(size)
        offsets.append(offset)
        hashed.append(otype == "HashGrid" and dense > size)
        offset += size
    return HashGridSpec(
        n_levels, n_features_per_level, log2_hashmap_size, base_resolution, per_level_scale, otype,
        tuple(scales), tuple(ress), tuple(sizes), tuple(offsets), tuple(hashed),
    )


def _corner_index(cx, cy, cz, res: int, size: int, is_hashed: bool):
    """uint32 index arithmetic emulated in int64 (wrap = mask to 32 bits)."""
    cx = cx & MASK32
    cy = cy & MASK32
    cz = cz & MASK32
    if is_hashed:
        idx = cx ^ ((cy * PRIME_Y) & MASK32) ^ ((cz * PRIME_Z) & MASK32)
    else:
        idx, stride = cx, res                                    # stride 1 <= size always
        if stride <= size:
            idx = (idx + ((cy * stride) & MASK32)) & MASK32
            stride = stride * res
            if stride <= size:
                idx = (idx + ((cz * stride) & MASK32)) & MASK32
    return idx % size


def encode(x: torch.Tensor, table: torch.Tensor, spec: HashGridSpec, chunk: int = 1 << 19) -> torch.Tensor:
    """x: (n,3) in the unit cube (float32 or float64); table: (n_params,) -> (n, L*F).

    Differentiable w.r.t. ``table`` and ``x`` (and twice differentiable), which is what
    the reference relies on for the log-intensity-gradient loss
    (``robust_e_nerf/models/robust_e_nerf.py:395-398``).
    """
    assert x.dim() == 2 and x.shape[-1] == 3
    if x.shape[0] > chunk:
        return torch.cat([encode(x[i:i + chunk], table, spec, chunk) for i in range(0, x.shape[0], chunk)], 0)
    F = spec.n_features_per_level
    tab = table.view(-1, F)
    idx_all, w_all = [], []
    for lvl in range(spec.n_levels):
        scale = spec.scales[lvl]
        if x.dtype == torch.float32:
            # fmaf(scale, x, 0.5): exact product in float64, one rounding to float32.
            # (the straight-through trick keeps d pos / d x = scale for autograd)
            pos_val = (x.detach().double() * float(scale) + 0.5).float()
            pos = pos_val + (x * float(scale) - (x * float(scale)).detach())
        else:
            pos = x * float(scale) + 0.5
        cell_f = torch.floor(pos.detach())
        w = pos - cell_f                                   # (n,3)
        cell = cell_f.to(torch.int64)
        for corner in range(8):
            bx, by, bz = corner & 1, (corner >> 1) & 1, (corner >> 2) & 1
            wx = w[:, 0] if bx else 1 - w[:, 0]
            wy = w[:, 1] if by else 1 - w[:, 1]
            wz = w[:, 2] if bz else 1 - w[:, 2]
            idx_all.append(spec.offsets[lvl] + _corner_index(
                cell[:, 0] + bx, cell[:, 1] + by, cell[:, 2] + bz,
                spec.resolutions[lvl], spec.sizes[lvl], spec.hashed[lvl]))
            w_all.append((wx * wy * wz).to(table.dtype))
    # ONE gather for all levels x corners: autograd's backward is then a single dense
    # index_add instead of 128 (each of which would allocate a 50 MB gradient).
    idx = torch.stack(idx_all, dim=1)                      # (n, L*8)
    wts = torch.stack(w_all, dim=1)                        # (n, L*8)
    feats = tab[idx.reshape(-1)].view(x.shape[0], spec.n_levels, 8, F)
    return (wts.view(x.shape[0], spec.n_levels, 8, 1) * feats).sum(dim=2).reshape(x.shape[0], -1)


def mix32_uniform(n: int, seed: int) -> np.ndarray:
    """Portable counter-based uniforms in [0,1): murmur3 finaliser of (i + seed*golden).
    Used for fixtures so a 50 MB table is reproducible from a seed on any numpy."""
    h = (np.arange(n, dtype=np.uint64) + np.uint64((seed * 0x9E3779B9) & MASK32)) & np.uint64(MASK32)
    h = h.astype(np.uint32)
    h ^= h >> np.uint32(16)
    h = (h.astype(np.uint64) * np.uint64(0x85EBCA6B) & np.uint64(MASK32)).astype(np.uint32)
    h ^= h >> np.uint32(13)
    h = (h.astype(np.uint64) * np.uint64(0xC2B2AE35) & np.uint64(MASK32)).astype(np.uint32)
    h ^= h >> np.uint32(16)
    return (h >> np.uint32(8)).astype(np.float32) / np.float32(1 << 24)


def init_table(spec: HashGridSpec, seed: int = 0, scale: float = 1e-4, kind: str = "uniform") -> torch.Tensor:
    """tcnn initialises grid params U(-1e-4, 1e-4); 'normal' gives a trained-like table;
    'mix32' is U(-scale, scale) from the portable counter-based generator above."""
    if kind == "mix32":
        u = mix32_uniform(spec.n_params, seed)
        return torch.from_numpy((u * np.float32(2) - np.float32(1)) * np.float32(scale))
    g = torch.Generator().manual_seed(seed)
    if kind == "uniform":
        return (torch.rand(spec.n_params, generator=g, dtype=torch.float32) * 2 - 1) * scale
    return torch.randn(spec.n_params, generator=g, dtype=torch.float32) * scale
