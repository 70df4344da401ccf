// Hash-grid parameter gradient WITHOUT global atomics: LDS-binned scatter.
//
// Why: on MI355X (8 XCDs, mutually incoherent L2s) a global atomic never executes in the L2; the
// TCC forwards every request to the memory side (measured: TCC_EA0_ATOMIC == TCC_ATOMIC, ~18 G
// atomic requests/s for the whole chip, independent of locality or XCD placement).  The tcnn-style
// scatter issues 128 (sample, level, corner) requests per sample = 2.1 G requests for the 16.8 M
// samples of one training step = 105 ms, 88 % of the step.  11 of the 16 levels are spatial hashes
// whose updates have no locality at all, so nothing short of a sort can aggregate them on chip.
//
// What: a counting sort of the updates by 8 192-entry table bin (2 features x int64 = 128 KiB = one
// LDS), then one workgroup per bin(-part) accumulates its updates with LDS atomics and adds the
// finished 128 KiB slice to the gradient table with plain coalesced read-modify-writes:
//   1. count    per (level, bin) number of updates                      (index math only)
//   2. offsets  exclusive scan of the ~770 bin counts + work partition  (one workgroup)
//   3. scatter  each workgroup sorts its 512 samples x 8 corners by bin in LDS and appends the
//               runs to the bins' regions of an HBM staging buffer {u16 local index, f32 v0, f32 v1}
//               with fully coalesced stores (288 GB of HBM is what makes a 10 B x 128 x n buffer
//               -- 21.5 GB at n = 16.8 M -- a reasonable thing to do)
//   4. accumulate  stream a bin part (coalesced), 64-bit fixed-point ds_add_u64 into LDS, flush.
// HBM traffic: 10 B written + 10 B read per update = 2.56 KB/sample (vs 2 KB of atomic RMW it
// replaces) but all of it streaming; global atomic requests drop from 128 to ~0.3 per sample.
#include "ren_hashgrid_common.h"

namespace {

constexpr int BIN_SHIFT = 13;
constexpr int BIN_ENTRIES = 1 << BIN_SHIFT;          // 8 192 table entries (x2 features x int64 = 128 KiB LDS)
constexpr int MAX_BINS_PER_LEVEL = 64;
constexpr int MAX_BINS = REN_MAX_LEVELS * MAX_BINS_PER_LEVEL;
constexpr int SCATTER_SAMPLES = 256;                 // samples per scatter workgroup (1 per thread)
constexpr int64_t PART_ENTRIES = 1 << 21;            // updates per accumulate workgroup

struct BinTab {
    int bin_base[REN_MAX_LEVELS + 1];                // first global bin of each level
};

struct Part {
    uint32_t gbin, single;
    uint64_t begin, end;
};

struct Workspace {
    uint32_t *counts, *level_max, *cursors, *n_parts;     // level_max: float bits of max |update| per level
    uint64_t *bin_start;
    Part *parts;
    uint16_t *out_idx;
    float *out_v0, *out_v1;
};

// optional tangent inputs (log-intensity-gradient loss): update = w * dfeat + wdot * dfeatd
struct TanSrc {
    const float *rays_do, *rays_dd, *dfeatd;
};

__device__ __forceinline__ bool load_sample(const GridDev &g, int lvl, int layout, const float *__restrict__ dfeat,
                                            const float *__restrict__ x_unit, const ren_scene_dev &sc,
                                            const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                            const int32_t *__restrict__ ray_indices, const float *__restrict__ t_starts,
                                            const float *__restrict__ t_ends, int64_t i, int64_t n, float &d0, float &d1,
                                            LevelPos &p, const TanSrc &tan, float &e0, float &e1, float *wd) {
    e0 = 0.f; e1 = 0.f; wd[0] = 0.f; wd[1] = 0.f; wd[2] = 0.f;
    if (i >= n) return false;
    if (tan.dfeatd) {                                               // fragment layout, packed rays only
        const int64_t b = ((i >> 5) * REN_MAX_LEVELS + lvl) * 64 + (i & 31);
        d0 = dfeat[b]; d1 = dfeat[b + 32];
        e0 = tan.dfeatd[b]; e1 = tan.dfeatd[b + 32];
        if (d0 == 0.f && d1 == 0.f && e0 == 0.f && e1 == 0.f) return false;
        float x[3], xd[3], u[3], ud[3];
        sample_pos_jvp(rays_o, rays_d, tan.rays_do, tan.rays_dd, ray_indices, t_starts, t_ends, i, x, xd);
        contract_jvp(sc, x, xd, u, ud);
        const float scale = g.scale[lvl];
        p = level_pos(u[0], u[1], u[2], scale);
        wd[0] = scale * ud[0]; wd[1] = scale * ud[1]; wd[2] = scale * ud[2];
        return true;
    }
    if (layout == 0) {
        const float2 d = reinterpret_cast<const float2 *>(dfeat)[i * g.n_levels + lvl];
        d0 = d.x; d1 = d.y;
    } else {
        const int64_t b = ((i >> 5) * REN_MAX_LEVELS + lvl) * 64 + (i & 31);
        d0 = dfeat[b];
        d1 = dfeat[b + 32];
    }
    if (d0 == 0.f && d1 == 0.f) return false;
    float ux, uy, uz;
    if (x_unit) {
        ux = x_unit[3 * i]; uy = x_unit[3 * i + 1]; uz = x_unit[3 * i + 2];
    } else {
        float x, y, z; int ray;
        ren_sample_pos(rays_o, rays_d, ray_indices, t_starts, t_ends, i, x, y, z, ray);
        ren_contract(sc, x, y, z, ux, uy, uz);
    }
    p = level_pos(ux, uy, uz, g.scale[lvl]);
    return true;
}

// ---- dense (non-hashed) levels: consecutive samples of a ray sit in the same cell, so all 8 corner
// updates of a run of lanes hit identical table entries.  Merging such runs in registers (segmented
// wave scan) before anything touches LDS removes the same-address serialisation of the LDS atomics
// and shrinks the staging traffic (x6.7 fewer updates at level 0 ... x1.5 at level 4).
__device__ __forceinline__ uint64_t cell_key(const LevelPos &p) {
    return ((uint64_t)p.c[2] << 42) ^ ((uint64_t)p.c[1] << 21) ^ (uint64_t)p.c[0];
}

// emit = this lane is the LAST lane of a run of valid lanes with equal cell (always true for hashed levels)
__device__ __forceinline__ bool run_tail(bool dense, bool have, uint64_t key, int lane, bool &head) {
    if (!dense) { head = true; return have; }
    const uint64_t kp = __shfl_up(key, 1, 64), kn = __shfl_down(key, 1, 64);
    const int hp = __shfl_up((int)have, 1, 64), hn = __shfl_down((int)have, 1, 64);
    head = !(lane > 0 && hp && have && kp == key);
    return have && (lane == 63 || !(hn && kn == key));
}

__device__ __forceinline__ void run_merge(bool head, int lane, float (&v0)[8], float (&v1)[8]) {
    int f = head ? 1 : 0;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int pf = __shfl_up(f, off, 64);
        const bool take = lane >= off && !f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float a = __shfl_up(v0[c], off, 64), b = __shfl_up(v1[c], off, 64);
            if (take) { v0[c] += a; v1[c] += b; }
        }
        if (lane >= off) f |= pf;
    }
}

// LDS counter bump with one atomic per distinct bin in the wave (dense levels: the lanes of a wave
// share 1-3 bins); returns the lane's rank inside its bin.
__device__ __forceinline__ uint32_t bin_rank(bool dense, bool emit, uint32_t bin, int lane, uint32_t *hist) {
    if (!dense) return emit ? atomicAdd(&hist[bin], 1u) : 0u;
    uint32_t rank = 0;
    uint64_t todo = __ballot(emit);
    while (todo) {
        const int leader = __ffsll((unsigned long long)todo) - 1;
        const uint32_t b = __shfl(bin, leader, 64);
        const uint64_t m = __ballot(emit && bin == b);
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(&hist[b], (uint32_t)__popcll(m));
        base = __shfl(base, leader, 64);
        if (emit && bin == b) rank = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        todo &= ~m;
    }
    return rank;
}

// ---- 1. count ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bin_count_kernel(
    GridDev g, BinTab bt, int layout, const float *__restrict__ dfeat, const float *__restrict__ x_unit,
    ren_scene_dev sc, const float *__restrict__ rays_o, const float *__restrict__ rays_d,
    const int32_t *__restrict__ ray_indices, const float *__restrict__ t_starts,
    const float *__restrict__ t_ends, int64_t n, uint32_t *__restrict__ counts, TanSrc tan) {
    __shared__ uint32_t hist[MAX_BINS_PER_LEVEL];
    const int lvl = blockIdx.x % g.n_levels;                      // level fastest: spreads the counter atomics
    const int64_t chunk = blockIdx.x / g.n_levels;
    if (threadIdx.x < MAX_BINS_PER_LEVEL) hist[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t res = g.res[lvl], size = g.size[lvl];
    const bool hashed = g.hashed[lvl] != 0;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < SCATTER_SAMPLES / 256; ++k) {
        const int64_t i = chunk * SCATTER_SAMPLES + k * 256 + threadIdx.x;
        float d0, d1, e0, e1, wdp[3]; LevelPos p = {};
        const bool have = load_sample(g, lvl, layout, dfeat, x_unit, sc, rays_o, rays_d, ray_indices, t_starts, t_ends, i, n, d0, d1, p, tan, e0, e1, wdp);
        bool head;
        const bool emit = run_tail(!hashed, have, cell_key(p), lane, head);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const uint32_t idx = corner_index(p.c[0] + (c & 1), p.c[1] + ((c >> 1) & 1), p.c[2] + (c >> 2), res, size, hashed);
            (void)bin_rank(!hashed, emit, idx >> BIN_SHIFT, lane, hist);
        }
    }
    __syncthreads();
    const int nb = bt.bin_base[lvl + 1] - bt.bin_base[lvl];
    if ((int)threadIdx.x < nb && hist[threadIdx.x]) atomicAdd(&counts[bt.bin_base[lvl] + threadIdx.x], hist[threadIdx.x]);
}

// ---- 2. offsets + work partition -----------------------------------------------------------------------
__global__ __launch_bounds__(MAX_BINS) void bin_offsets_kernel(int n_bins, const uint32_t *__restrict__ counts,
                                                               uint32_t *__restrict__ cursors,
                                                               uint64_t *__restrict__ bin_start,
                                                               Part *__restrict__ parts, uint32_t *__restrict__ n_parts) {
    __shared__ uint64_t s_cnt[MAX_BINS];
    __shared__ uint32_t s_np[MAX_BINS];
    const int t = threadIdx.x;
    const uint64_t c = t < n_bins ? counts[t] : 0;
    const uint32_t np = (uint32_t)((c + PART_ENTRIES - 1) / PART_ENTRIES);
    s_cnt[t] = c; s_np[t] = np;
    if (t < n_bins) cursors[t] = 0;
    __syncthreads();
    for (int off = 1; off < MAX_BINS; off <<= 1) {
        const uint64_t a = t >= off ? s_cnt[t - off] : 0;
        const uint32_t b = t >= off ? s_np[t - off] : 0;
        __syncthreads();
        s_cnt[t] += a; s_np[t] += b;
        __syncthreads();
    }
    const uint64_t start = s_cnt[t] - c;
    if (t < n_bins) bin_start[t] = start;
    if (t == n_bins - 1) { bin_start[n_bins] = s_cnt[t]; n_parts[0] = s_np[t]; }
    const uint32_t pbase = s_np[t] - np;
    for (uint32_t k = 0; k < np; ++k) {
        Part p;
        p.gbin = t; p.single = np == 1;
        p.begin = start + (uint64_t)k * PART_ENTRIES;
        p.end = k + 1 == np ? start + c : p.begin + PART_ENTRIES;
        parts[pbase + k] = p;
    }
}

// ---- 3. scatter (counting sort by bin inside the workgroup, coalesced append) ----------------------------
constexpr int SC_ENTRIES = SCATTER_SAMPLES * 8;      // 4096 staged updates = 48 KiB

__global__ __launch_bounds__(256) void bin_scatter_kernel(
    GridDev g, BinTab bt, int layout, const float *__restrict__ dfeat, const float *__restrict__ x_unit,
    ren_scene_dev sc, const float *__restrict__ rays_o, const float *__restrict__ rays_d,
    const int32_t *__restrict__ ray_indices, const float *__restrict__ t_starts,
    const float *__restrict__ t_ends, int64_t n, Workspace ws, TanSrc tan) {
    __shared__ uint32_t hist[MAX_BINS_PER_LEVEL], loc[MAX_BINS_PER_LEVEL + 1];
    __shared__ uint64_t gpos[MAX_BINS_PER_LEVEL];
    __shared__ uint32_t st_key[SC_ENTRIES];
    __shared__ float st_v0[SC_ENTRIES], st_v1[SC_ENTRIES];
    __shared__ float wave_max[4];
    const int lvl = blockIdx.x % g.n_levels;
    const int64_t chunk = blockIdx.x / g.n_levels;
    const int tid = threadIdx.x;
    if (tid < MAX_BINS_PER_LEVEL) hist[tid] = 0;
    __syncthreads();
    const uint32_t res = g.res[lvl], size = g.size[lvl];
    const bool hashed = g.hashed[lvl] != 0;
    constexpr int SPT = SCATTER_SAMPLES / 256;
    const int lane = tid & 63;
    uint32_t key[SPT][8];                                          // rank << 19 | table index in level
    float v0[SPT][8], v1[SPT][8];
    bool have[SPT];
    float vmax = 0.f;
#pragma unroll
    for (int k = 0; k < SPT; ++k) {
        const int64_t i = chunk * SCATTER_SAMPLES + k * 256 + tid;
        float d0 = 0.f, d1 = 0.f, e0, e1, wdp[3]; LevelPos p = {};
        const bool valid = load_sample(g, lvl, layout, dfeat, x_unit, sc, rays_o, rays_d, ray_indices, t_starts, t_ends, i, n, d0, d1, p, tan, e0, e1, wdp);
        uint32_t idx[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            idx[c] = corner_index(p.c[0] + (c & 1), p.c[1] + ((c >> 1) & 1), p.c[2] + (c >> 2), res, size, hashed);
            const float wx = (c & 1) ? p.w[0] : 1.f - p.w[0];
            const float wy = (c & 2) ? p.w[1] : 1.f - p.w[1];
            const float wz = (c & 4) ? p.w[2] : 1.f - p.w[2];
            const float w = valid ? wx * wy * wz : 0.f;
            const float bx = (c & 1) ? wdp[0] : -wdp[0], by = (c & 2) ? wdp[1] : -wdp[1], bz = (c & 4) ? wdp[2] : -wdp[2];
            const float wdc = bx * wy * wz + wx * by * wz + wx * wy * bz;       // d w / dt (0 without tangent)
            v0[k][c] = w * d0 + wdc * e0;
            v1[k][c] = w * d1 + wdc * e1;
        }
        bool head;
        have[k] = run_tail(!hashed, valid, cell_key(p), lane, head);
        if (!hashed) run_merge(head, lane, v0[k], v1[k]);
        if (have[k]) {
#pragma unroll
            for (int c = 0; c < 8; ++c) vmax = fmaxf(vmax, fmaxf(fabsf(v0[k][c]), fabsf(v1[k][c])));
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const uint32_t rank = bin_rank(!hashed, have[k], idx[c] >> BIN_SHIFT, lane, hist);
            key[k][c] = (rank << 19) | idx[c];                     // idx < 2^19: bin = idx >> 14, local = idx & 16383
        }
    }
    // max |update| of the level: scales the 64-bit fixed-point accumulation of the next kernel
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off, 64));
    if (lane == 0) wave_max[tid >> 6] = vmax;
    __syncthreads();
    if (tid == 0) {
        const float m = fmaxf(fmaxf(wave_max[0], wave_max[1]), fmaxf(wave_max[2], wave_max[3]));
        if (m > 0.f) atomicMax(&ws.level_max[lvl], __float_as_uint(m));      // non-negative floats order like uints
    }
    const int nb = bt.bin_base[lvl + 1] - bt.bin_base[lvl];
    if (tid < MAX_BINS_PER_LEVEL) {
        const uint32_t cnt = hist[tid];
        // inclusive wave scan over the 64 bins -> local offsets
        uint32_t inc = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(inc, off, 64);
            if (tid >= off) inc += t;
        }
        loc[tid] = inc - cnt;
        if (tid == MAX_BINS_PER_LEVEL - 1) loc[MAX_BINS_PER_LEVEL] = inc;
        uint64_t base = 0;
        if (tid < nb && cnt) {
            const int gb = bt.bin_base[lvl] + tid;
            base = ws.bin_start[gb] + atomicAdd(&ws.cursors[gb], cnt);   // reserve the run in the bin's region
        }
        gpos[tid] = base;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SPT; ++k) {
        if (!have[k]) continue;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const uint32_t kk = key[k][c];
            const uint32_t idx = kk & 0x7FFFFu;
            const uint32_t pos = loc[idx >> BIN_SHIFT] + (kk >> 19);
            st_key[pos] = idx;
            st_v0[pos] = v0[k][c];
            st_v1[pos] = v1[k][c];
        }
    }
    __syncthreads();
    const uint32_t total = loc[MAX_BINS_PER_LEVEL];
    for (uint32_t p = tid; p < total; p += 256) {
        const uint32_t idx = st_key[p];
        const uint32_t b = idx >> BIN_SHIFT;
        const uint64_t gp = gpos[b] + (p - loc[b]);
        ws.out_idx[gp] = (uint16_t)(idx & (BIN_ENTRIES - 1));
        ws.out_v0[gp] = st_v0[p];
        ws.out_v1[gp] = st_v1[p];
    }
}

// ---- 4. accumulate one bin part in LDS, flush to the gradient table -------------------------------------------
// LDS float atomics are lane-serialised on gfx950 (ds_add_f32: 0.37 lanes/clk/CU measured), integer
// ones are 3x faster, so the sums are formed in 64-bit fixed point: q = round(v * 2^k) with
// 2^k * max|v| ~ 2^38, i.e. a quantum of 4e-12 of the level's largest update; 2^21 updates per part
// cannot overflow, and the result is MORE accurate than an fp32 running sum.
__device__ __forceinline__ long long to_fixed(float v, double scale) {
    return __double2ll_rn((double)v * scale);
}

__global__ __launch_bounds__(1024) void bin_accumulate_kernel(GridDev g, BinTab bt, Workspace ws,
                                                              float *__restrict__ grad_table) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long acc[];   // acc0[8192] | acc1[8192]
    if (blockIdx.x >= ws.n_parts[0]) return;
    const Part part = ws.parts[blockIdx.x];
    unsigned long long *acc0 = acc, *acc1 = acc + BIN_ENTRIES;
    for (int e = threadIdx.x; e < 2 * BIN_ENTRIES; e += 1024) acc[e] = 0ull;
    int lvl = 0;
    while (lvl + 1 < g.n_levels && (int)part.gbin >= bt.bin_base[lvl + 1]) ++lvl;
    int ex;
    (void)frexpf(__uint_as_float(ws.level_max[lvl]), &ex);         // level_max < 2^ex
    const double scale = ldexp(1.0, 38 - ex), inv_scale = ldexp(1.0, ex - 38);
    __syncthreads();
    uint64_t e = part.begin + threadIdx.x;
    for (; e + 3 * 1024 < part.end; e += 4 * 1024) {               // 4 independent loads in flight per lane
        uint32_t ix[4]; float a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { ix[u] = ws.out_idx[e + u * 1024]; a[u] = ws.out_v0[e + u * 1024]; b[u] = ws.out_v1[e + u * 1024]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            atomicAdd(&acc0[ix[u]], (unsigned long long)to_fixed(a[u], scale));    // ds_add_u64
            atomicAdd(&acc1[ix[u]], (unsigned long long)to_fixed(b[u], scale));
        }
    }
    for (; e < part.end; e += 1024) {
        const uint32_t idx = ws.out_idx[e];
        atomicAdd(&acc0[idx], (unsigned long long)to_fixed(ws.out_v0[e], scale));
        atomicAdd(&acc1[idx], (unsigned long long)to_fixed(ws.out_v1[e], scale));
    }
    __syncthreads();
    const uint32_t first = ((uint32_t)part.gbin - bt.bin_base[lvl]) << BIN_SHIFT;   // first entry of the bin in its level
    const uint32_t lim = g.size[lvl] > first ? g.size[lvl] - first : 0;
    float2 *gt = reinterpret_cast<float2 *>(grad_table) + g.offset[lvl] + first;
    for (uint32_t k = threadIdx.x; k < BIN_ENTRIES && k < lim; k += 1024) {
        const long long qa = (long long)acc0[k], qb = (long long)acc1[k];
        if (qa == 0 && qb == 0) continue;
        const float a = (float)((double)qa * inv_scale), b = (float)((double)qb * inv_scale);
        if (part.single) {                                         // exclusive owner: plain coalesced RMW
            float2 v = gt[k];
            v.x += a; v.y += b;
            gt[k] = v;
        } else {
            atomicAdd(&gt[k].x, a);
            atomicAdd(&gt[k].y, b);
        }
    }
}

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

struct Layout { size_t counts, level_max, cursors, n_parts, bin_start, parts, out_idx, out_v0, out_v1, total; int64_t max_parts; };

Layout make_layout(int64_t n) {
    Layout L;
    const size_t E = (size_t)n * 128;
    L.max_parts = (int64_t)(E / PART_ENTRIES) + MAX_BINS + 1;
    size_t o = 0;
    L.counts = o; o += MAX_BINS * 4;                       // counts | level_max are cleared by one memset
    L.level_max = o; o = align256(o + REN_MAX_LEVELS * 4);
    L.cursors = o; o = align256(o + MAX_BINS * 4);
    L.n_parts = o; o = align256(o + 4);
    L.bin_start = o; o = align256(o + (MAX_BINS + 1) * 8);
    L.parts = o; o = align256(o + (size_t)L.max_parts * sizeof(Part));
    L.out_idx = o; o = align256(o + E * 2);
    L.out_v0 = o; o = align256(o + E * 4);
    L.out_v1 = o; o = align256(o + E * 4);
    L.total = o;
    return L;
}

}  // namespace

extern "C" int64_t ren_hashgrid_bwd_binned_workspace_bytes(int64_t n) {
    if (n < 0) return -1;
    return (int64_t)make_layout(n).total;
}

static int binned_impl(const ren_grid_desc *grid, float *grad_table, const float *x_unit,
                       const ren_scene_desc *scene, const float *rays_o, const float *rays_d,
                       const int32_t *ray_indices, const float *t_starts, const float *t_ends,
                       int64_t n, int32_t layout, const float *dfeat, void *workspace,
                       void *stream, TanSrc tan) {
    GridDev g;
    int rc = make_grid(grid, g);
    if (rc) return rc;
    if (!grad_table || !dfeat || !workspace || n < 0 || (layout != 0 && layout != 1)) return REN_ERR_BAD_ARG;
    const bool from_rays = x_unit == nullptr;
    if (from_rays && (!scene || !rays_o || !rays_d || !ray_indices || !t_starts || !t_ends)) return REN_ERR_BAD_ARG;
    if (layout == 1 && g.n_levels != REN_MAX_LEVELS) return REN_ERR_UNSUPPORTED;
    if (n >= ((int64_t)1 << 28)) return REN_ERR_UNSUPPORTED;      // 8 n updates per level must fit uint32
    BinTab bt;
    int nb = 0;
    for (int l = 0; l < REN_MAX_LEVELS; ++l) {
        bt.bin_base[l] = nb;
        if (l < g.n_levels) {
            if (g.size[l] > (uint32_t)(MAX_BINS_PER_LEVEL << BIN_SHIFT)) return REN_ERR_UNSUPPORTED;
            nb += (int)((g.size[l] + BIN_ENTRIES - 1) >> BIN_SHIFT);
        }
    }
    bt.bin_base[REN_MAX_LEVELS] = nb;
    for (int l = g.n_levels; l < REN_MAX_LEVELS; ++l) bt.bin_base[l] = nb;
    if (n == 0) return REN_OK;
    ren_scene_dev sc = {};
    if (scene) sc = ren_make_scene(scene);
    const Layout L = make_layout(n);
    char *w = (char *)workspace;
    Workspace ws;
    ws.counts = (uint32_t *)(w + L.counts); ws.level_max = (uint32_t *)(w + L.level_max);
    ws.cursors = (uint32_t *)(w + L.cursors);
    ws.n_parts = (uint32_t *)(w + L.n_parts); ws.bin_start = (uint64_t *)(w + L.bin_start);
    ws.parts = (Part *)(w + L.parts); ws.out_idx = (uint16_t *)(w + L.out_idx);
    ws.out_v0 = (float *)(w + L.out_v0); ws.out_v1 = (float *)(w + L.out_v1);
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(ws.counts, 0, (MAX_BINS + REN_MAX_LEVELS) * 4, st) != hipSuccess) return REN_ERR_LAUNCH;
    const int64_t chunks = (n + SCATTER_SAMPLES - 1) / SCATTER_SAMPLES;
    dim3 grd((unsigned)(chunks * g.n_levels)), blk(256);
    hipLaunchKernelGGL(bin_count_kernel, grd, blk, 0, st, g, bt, layout, dfeat, x_unit, sc, rays_o, rays_d,
                       ray_indices, t_starts, t_ends, n, ws.counts, tan);
    hipLaunchKernelGGL(bin_offsets_kernel, dim3(1), dim3(MAX_BINS), 0, st, nb, ws.counts, ws.cursors, ws.bin_start,
                       ws.parts, ws.n_parts);
    hipLaunchKernelGGL(bin_scatter_kernel, grd, blk, 0, st, g, bt, layout, dfeat, x_unit, sc, rays_o, rays_d,
                       ray_indices, t_starts, t_ends, n, ws, tan);
    const size_t acc_lds = 2 * BIN_ENTRIES * sizeof(unsigned long long);
    (void)hipFuncSetAttribute((const void *)bin_accumulate_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)acc_lds);
    hipLaunchKernelGGL(bin_accumulate_kernel, dim3((unsigned)L.max_parts), dim3(1024), acc_lds, st, g, bt, ws, grad_table);
    REN_CHECK_LAUNCH();
}

extern "C" int ren_hashgrid_bwd_binned(const ren_grid_desc *grid, float *grad_table, const float *x_unit,
                                       const ren_scene_desc *scene, const float *rays_o, const float *rays_d,
                                       const int32_t *ray_indices, const float *t_starts, const float *t_ends,
                                       int64_t n, int32_t layout, const float *dfeat, void *workspace,
                                       void *stream) {
    return binned_impl(grid, grad_table, x_unit, scene, rays_o, rays_d, ray_indices, t_starts, t_ends, n, layout,
                       dfeat, workspace, stream, TanSrc{nullptr, nullptr, nullptr});
}

extern "C" int ren_hashgrid_bwd_binned_jvp(const ren_grid_desc *grid, float *grad_table,
                                           const ren_scene_desc *scene, const float *rays_o, const float *rays_d,
                                           const float *rays_do, const float *rays_dd, const int32_t *ray_indices,
                                           const float *t_starts, const float *t_ends, int64_t n,
                                           const float *dfeat, const float *dfeatd, void *workspace, void *stream) {
    if (!rays_do || !rays_dd || !dfeatd) return REN_ERR_BAD_ARG;
    return binned_impl(grid, grad_table, nullptr, scene, rays_o, rays_d, ray_indices, t_starts, t_ends, n, 1, dfeat,
                       workspace, stream, TanSrc{rays_do, rays_dd, dfeatd});
}
