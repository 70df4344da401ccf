// Shared device/host helpers of the hash-grid kernels (tcnn HashGrid indexing).  Internal.
#pragma once
#include "ren_common.h"

namespace {

struct GridDev {
    int n_levels;
    float scale[REN_MAX_LEVELS];
    uint32_t res[REN_MAX_LEVELS], size[REN_MAX_LEVELS], offset[REN_MAX_LEVELS], hashed[REN_MAX_LEVELS];
};

__device__ __forceinline__ uint32_t corner_index(uint32_t cx, uint32_t cy, uint32_t cz, uint32_t res,
                                                 uint32_t size, bool hashed) {
    if (hashed) {
        uint32_t h = cx ^ (cy * 2654435761u) ^ (cz * 805459861u);
        return h & (size - 1u);                      // hashed levels have power-of-two size
    }
    uint32_t idx = cx + cy * res + cz * res * res;
    if (idx >= size) { idx -= size; if (idx >= size) idx %= size; }
    return idx;
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
