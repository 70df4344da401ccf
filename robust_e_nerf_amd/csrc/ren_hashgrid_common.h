// Shared device/host helpers of the hash-grid kernels (tcnn HashGrid indexing).  Internal.
#pragma once
#include "ren_common.h"

namespace {

struct GridDev {
    int n_levels;
    float scale[REN_MAX_LEVELS];
    uint32_t res[REN_MAX_LEVELS], size[REN_MAX_LEVELS], offset[REN_MAX_LEVELS], hashed[REN_MAX_LEVELS];
};

// tcnn grid_index(): index += cell_d * stride for each dimension WHILE stride <= level size (stride *= res after each).  A
// HashGrid level that is not hashed has res^3 <= size, so both strides are there; a TiledGrid level (size = base_resolution^3
// < res^3) wraps, and drops z once res^2 > size (and y once res > size).  res <= 65 535: res * res fits 32 bits.
__device__ __forceinline__ uint32_t dense_stride_y(uint32_t res, uint32_t size) { return res <= size ? res : 0u; }
__device__ __forceinline__ uint32_t dense_stride_z(uint32_t res, uint32_t size) { return res * res <= size ? res * res : 0u; }

__device__ __forceinline__ uint32_t corner_index(uint32_t cx, uint32_t cy, uint32_t cz, uint32_t res,
                                                 uint32_t size, bool hashed) {
    if (hashed) {
        uint32_t h = cx ^ (cy * 2654435761u) ^ (cz * 805459861u);
        return h & (size - 1u);                      // hashed levels have power-of-two size
    }
    const uint32_t sy = dense_stride_y(res, size), sz = dense_stride_z(res, size);
    uint32_t idx = cx + cy * sy + cz * sz;
    if (idx >= size) { idx -= size; if (idx >= size) idx %= size; }
    return idx;
}

// all 8 corner indices of a cell, corner c = (c&1, (c>>1)&1, c>>2).  Same values as corner_index();
// the hash products / dense strides are formed once instead of per corner.
__device__ __forceinline__ void corner_indices8(uint32_t cx, uint32_t cy, uint32_t cz, uint32_t res, uint32_t size,
                                                bool hashed, uint32_t *idx) {
    if (hashed) {
        const uint32_t hy0 = cy * 2654435761u, hy1 = hy0 + 2654435761u;
        const uint32_t hz0 = cz * 805459861u, hz1 = hz0 + 805459861u;
        const uint32_t m = size - 1u, x1 = cx + 1u;
        idx[0] = (cx ^ hy0 ^ hz0) & m; idx[1] = (x1 ^ hy0 ^ hz0) & m;
        idx[2] = (cx ^ hy1 ^ hz0) & m; idx[3] = (x1 ^ hy1 ^ hz0) & m;
        idx[4] = (cx ^ hy0 ^ hz1) & m; idx[5] = (x1 ^ hy0 ^ hz1) & m;
        idx[6] = (cx ^ hy1 ^ hz1) & m; idx[7] = (x1 ^ hy1 ^ hz1) & m;
    } else {
        const uint32_t sy = dense_stride_y(res, size), sz = dense_stride_z(res, size);
        const uint32_t b = cx + cy * sy + cz * sz;
        idx[0] = b; idx[1] = b + 1u; idx[2] = b + sy; idx[3] = b + sy + 1u;
        idx[4] = b + sz; idx[5] = b + sz + 1u; idx[6] = b + sz + sy; idx[7] = b + sz + sy + 1u;
        // idx[] ascends unless it wraps 2^32 (negative cell), and then idx[0] is huge: two tests cover all 8
        if (idx[0] >= size || idx[7] >= size) {      // positions outside the unit cube, and tiled levels (size < res^3)
#pragma unroll
            for (int c = 0; c < 8; ++c) idx[c] %= size;
        }
    }
}

// Gather the 8 corner entries of a cell.  The texture addresser handles divergent lanes at about one
// lane per clock whatever the access size, and it is the unit that bounds the encoder (TA_BUSY 84 %), so
// corners (x, y, z) and (x+1, y, z) are fetched with ONE 16-byte load whenever their entries are
// neighbours in the table: always on dense levels, and on hashed levels whenever x is even (x+1 then
// only flips bit 0 of the hash).  Lanes without that luck issue the two 8-byte loads.
__device__ __forceinline__ void gather_corners8(const float2 *__restrict__ tab, const uint32_t *idx, float2 *v) {
    const bool up = idx[1] == idx[0] + 1u, down = idx[0] == idx[1] + 1u;   // same relation for all 4 x-pairs
    const bool pair = (up || down) && idx[3] - idx[2] == idx[1] - idx[0] && idx[5] - idx[4] == idx[1] - idx[0] &&
                      idx[7] - idx[6] == idx[1] - idx[0];
    if (pair) {
        float4 q[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t lo = up ? idx[2 * j] : idx[2 * j + 1];
            q[j] = *reinterpret_cast<const float4 *>(tab + lo);             // 8-byte aligned 16-byte load
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float2 a = make_float2(q[j].x, q[j].y), b = make_float2(q[j].z, q[j].w);
            v[2 * j] = up ? a : b;
            v[2 * j + 1] = up ? b : a;
        }
    } else {
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = tab[idx[c]];
    }
}

struct LevelPos {
    uint32_t c[3];
    float w[3];
};

__device__ __forceinline__ LevelPos level_pos(float x, float y, float z, float scale) {
    LevelPos p;
    const float in[3] = {x, y, z};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float pos = fmaf(scale, in[k], 0.5f);
        float fl = floorf(pos);
        p.w[k] = pos - fl;
        p.c[k] = (uint32_t)(int)fl;
    }
    return p;
}

// contraction with tangent: (x, xd) world -> (u, ud) unit cube   (ngp.py:68-106,230-237)
__device__ __forceinline__ void contract_jvp(const ren_scene_dev &sc, const float *x, const float *xd, float *u, float *ud) {
    float y[3], yd[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float ext = sc.hi[k] - sc.lo[k];
        y[k] = (x[k] - sc.lo[k]) / ext;
        yd[k] = xd[k] / ext;
    }
    if (sc.ct == REN_CT_SPHERE) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { y[k] = y[k] * 2.f - 1.f; yd[k] *= 2.f; }
        const float m = sqrtf(y[0] * y[0] + y[1] * y[1] + y[2] * y[2]);
        if (m > 1.f) {
            const float g = (2.f - 1.f / m) / m;                       // 2/m - 1/m^2
            const float gp = (-2.f + 2.f / m) / (m * m);               // -2/m^2 + 2/m^3
            const float md = (y[0] * yd[0] + y[1] * yd[1] + y[2] * yd[2]) / m;
#pragma unroll
            for (int k = 0; k < 3; ++k) { const float yk = y[k]; y[k] = yk * g; yd[k] = yd[k] * g + yk * gp * md; }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) { u[k] = y[k] * 0.25f + 0.5f; ud[k] = yd[k] * 0.25f; }
    } else if (sc.ct == REN_CT_TANH) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float th = tanhf(y[k] - 0.5f);
            u[k] = (th + 1.f) * 0.5f;
            ud[k] = (1.f - th * th) * yd[k] * 0.5f;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) { u[k] = y[k]; ud[k] = yd[k]; }
    }
}

// sample position and its time derivative: x = o + d tm, xd = od + dd tm (tm fixed)
__device__ __forceinline__ void sample_pos_jvp(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                               const float *__restrict__ rays_do, const float *__restrict__ rays_dd,
                                               const int32_t *__restrict__ ri, const float *__restrict__ ts,
                                               const float *__restrict__ te, int64_t i, float *x, float *xd) {
    const int64_t ray = ri[i];
    const float tm = (ts[i] + te[i]) * 0.5f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        x[k] = rays_o[3 * ray + k] + rays_d[3 * ray + k] * tm;
        xd[k] = rays_do[3 * ray + k] + rays_dd[3 * ray + k] * tm;
    }
}

inline int make_grid(const ren_grid_desc *grid, GridDev &g) {
    if (!grid || grid->n_levels < 1 || grid->n_levels > REN_MAX_LEVELS) return REN_ERR_BAD_ARG;
    g.n_levels = grid->n_levels;
    for (int l = 0; l < REN_MAX_LEVELS; ++l) {
        g.scale[l] = grid->scale[l]; g.res[l] = grid->res[l]; g.size[l] = grid->size[l];
        g.offset[l] = grid->offset[l]; g.hashed[l] = grid->hashed[l];
        if (l < grid->n_levels && grid->hashed[l] && (grid->size[l] & (grid->size[l] - 1)))
            return REN_ERR_UNSUPPORTED;              // hashed level sizes are 2^log2_hashmap_size
    }
    return REN_OK;
}

}  // namespace
