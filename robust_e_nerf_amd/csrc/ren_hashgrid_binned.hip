// Hash-grid parameter gradient WITHOUT global atomics: LDS-binned scatter.
//
// Why: on MI355X (8 XCDs, mutually incoherent L2s) a global atomic never executes in the L2; the
// TCC forwards every request to the memory side (measured: TCC_EA0_ATOMIC == TCC_ATOMIC, ~18 G
// atomic requests/s for the whole chip, independent of locality or XCD placement).  The tcnn-style
// scatter issues 128 (sample, level, corner) requests per sample = 2.1 G requests for the 16.8 M
// samples of one training step = 105 ms, 88 % of the step.  11 of the 16 levels are spatial hashes
// whose updates have no locality at all, so nothing short of a sort can aggregate them on chip.
//
// What: the updates are sorted by 8 192-entry table bin (2 features x int64 = 128 KiB = one LDS) in two steps:
//   1. scatter   one sample per thread, the workgroup (512 samples) walks the levels: per level rank its
//                records by bin (LDS counters), place them in an LDS staging area and write the sorted block
//                -- one "slab" per (level, workgroup) -- to HBM with one contiguous, fully coalesced 32-40 KiB
//                store, plus a 256-byte directory (start | count per bin)
//   2. accumulate  one workgroup per bin(-part) walks the directory column of its bin, gathers the bin's run
//                out of every slab (~0.5 KiB pieces), adds them with 64-bit fixed-point ds_add_u64 into LDS and
//                adds the finished 128 KiB slice to the gradient table with plain coalesced read-modify-writes.
// Round 1 appended every run to a per-bin region instead (reserved with a returning global atomic per run):
// 64 concurrent ~0.5 KiB append streams per workgroup cost 3.84 ms where the contiguous slab costs 2.70 ms
// (n = 16.8 M, hashed levels), the gather on the read side costs nothing (2.65 vs 2.58 ms), and the slab needs no
// count pass, no capacity estimate, no overflow path and no cursor atomics: memset + 2 scatter launches + 1
// accumulate launch.
//
// Pair records (hashed levels): corners (x, y, z) and (x+1, y, z) of a cell hash to indices that differ only in
// their low bits (x ^ (x+1) = 2^(k+1) - 1), so they fall into the same bin except with probability 2^-13 (those
// go to the table with global atomics), and their updates are (1 - fx) A and fx A with ONE shared A = wy wz dfeat.
// A hashed level therefore stages 4 records {u32 idx0 | idx1 << 13, float2 A, float fx} = 16 B per sample instead
// of 8 x 10 B: 8 B per update, half as many ranking atomics and LDS placements, one 16-byte store / load per
// record.  Dense levels keep single-update records {u16 idx, float2 v} (their run merge sums different fx).
// HBM traffic at config B: 64 B x 11 hashed levels + ~19 B x 5 dense levels per sample, written once and read once.
#include <cstdlib>
#include "ren_hashgrid_common.h"

namespace {

constexpr int BIN_SHIFT = 13;
constexpr int BIN_ENTRIES = 1 << BIN_SHIFT;          // 8 192 table entries (x2 features x int64 = 128 KiB LDS)
constexpr int MAX_BINS_PER_LEVEL = 64;
#ifndef REN_SC_THREADS
#define REN_SC_THREADS 512
#endif
#ifndef REN_SC_WAVES
#define REN_SC_WAVES 6
#endif
#ifndef REN_SC_WAVES_PAIR
#define REN_SC_WAVES_PAIR 8
#endif
constexpr int SC_THREADS = REN_SC_THREADS;           // scatter workgroup: one sample per thread
constexpr int SC_ENTRIES = SC_THREADS * 8;           // single-update records per level pass (48 KiB staged)
constexpr int SC_PAIRS = SC_THREADS * 4;             // pair records per level pass (32 KiB staged)
// one slab per (level, scatter workgroup): float4 rec[SC_PAIRS]  |  float2 v[SC_ENTRIES] + u16 idx[SC_ENTRIES]
constexpr size_t SLAB_BYTES = (size_t)SC_ENTRIES * 10;
static_assert(SLAB_BYTES >= (size_t)SC_PAIRS * 16 && SLAB_BYTES % 256 == 0, "slab layout");
// records per accumulate part of a dense bin (hashed bins are always ONE part: a bin cut in two flushes both
// halves with 16 384 float atomics -- memory-side, ~18 G/s chip-wide -- instead of one coalesced read-modify-write)
constexpr int64_t PART_RECORDS = 1 << 20;
constexpr int MAX_PARTS_PER_BIN = 256;
// max |update| per level is published with atomicMax: spread over LMAX_SLOTS cache lines per level and
// only raised when the value actually grows, so the workgroups do not queue on one memory channel.
constexpr int LMAX_SLOTS = 8, LMAX_STRIDE = 32;      // u32 words between slots (128 B)
constexpr int LMAX_WORDS = REN_MAX_LEVELS * LMAX_SLOTS * LMAX_STRIDE;

struct BinTab {
    int bin_base[REN_MAX_LEVELS + 1];                // first global bin of each level
    uint32_t pair[REN_MAX_LEVELS];                   // 1: the level is staged as 16-byte pair records
    int first_part[REN_MAX_LEVELS + 1];              // accumulate grid: parts of level l are [first_part[l], first_part[l+1])
    int parts_per_bin[REN_MAX_LEVELS];               // a bin's slabs are split into this many ranges of scatter workgroups
};

struct Workspace {
    uint32_t *level_max;                             // float bits of max |update|, [level][slot] one line each
    uint32_t *dir;                                   // [level][scatter workgroup][64 bins]: start | count << 16
    char *slabs;                                     // [level][scatter workgroup] SLAB_BYTES each
    int64_t n_wg;                                    // scatter workgroups = ceil(n / SC_THREADS)
};
constexpr int PAIR_BIN_SHIFT = 26;                   // pair record code: idx0 | idx1 << 13 | bin << 26 (bin: LDS staging only)

// optional tangent inputs (log-intensity-gradient loss): update = w * dfeat + wdot * dfeatd
struct TanSrc {
    const float *rays_do, *rays_dd, *dfeatd;
};

// workgroup barrier that orders LDS traffic only: global loads/stores/atomics stay in flight across it
// (__syncthreads() also drains vmcnt, which would put every HBM round trip on the per-level critical path)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct SampleArgs {
    int layout;                                      // 0: dfeat[i][level][2], 1: MFMA fragment order
    const float *dfeat, *x_unit;
    ren_scene_dev sc;
    const float *rays_o, *rays_d;
    const int32_t *ray_indices;
    const float *t_starts, *t_ends;
    int64_t n;
    TanSrc tan;
};

// unit-cube position (and its time derivative) of sample i: computed ONCE per thread, every level of
// the thread's level walk reuses it (the per-level recomputation was most of the old scatter's time)
template <bool TAN>
__device__ __forceinline__ void unit_pos(const SampleArgs &a, int64_t i, float *u, float *ud) {
    ud[0] = 0.f; ud[1] = 0.f; ud[2] = 0.f;
    if (TAN) {
        float x[3], xd[3];
        sample_pos_jvp(a.rays_o, a.rays_d, a.tan.rays_do, a.tan.rays_dd, a.ray_indices, a.t_starts, a.t_ends, i, x, xd);
        contract_jvp(a.sc, x, xd, u, ud);
    } else if (a.x_unit) {
        u[0] = a.x_unit[3 * i]; u[1] = a.x_unit[3 * i + 1]; u[2] = a.x_unit[3 * i + 2];
    } else {
        float x, y, z; int ray;
        ren_sample_pos(a.rays_o, a.rays_d, a.ray_indices, a.t_starts, a.t_ends, i, x, y, z, ray);
        ren_contract(a.sc, x, y, z, u[0], u[1], u[2]);
    }
}

// feature gradients of (sample i, level): false when there is nothing to scatter
template <bool TAN>
__device__ __forceinline__ bool load_dfeat(const SampleArgs &a, int n_levels, int lvl, int64_t i, float &d0, float &d1,
                                           float &e0, float &e1) {
    e0 = 0.f; e1 = 0.f;
    if (TAN || a.layout == 1) {
        const int64_t b = ((i >> 5) * REN_MAX_LEVELS + lvl) * 64 + (i & 31);
        d0 = a.dfeat[b]; d1 = a.dfeat[b + 32];
        if (TAN) { e0 = a.tan.dfeatd[b]; e1 = a.tan.dfeatd[b + 32]; }
    } else {
        const float2 d = reinterpret_cast<const float2 *>(a.dfeat)[i * n_levels + lvl];
        d0 = d.x; d1 = d.y;
    }
    return d0 != 0.f || d1 != 0.f || e0 != 0.f || e1 != 0.f;
}

// ---- dense (non-hashed) levels: consecutive samples of a ray sit in the same cell, so all 8 corner
// updates of a run of lanes hit identical table entries.  Merging such runs in registers (segmented
// wave scan) before anything touches LDS removes the same-address serialisation of the LDS atomics
// and shrinks the staging traffic (x6.7 fewer updates at level 0 ... x1.5 at level 4).
__device__ __forceinline__ uint64_t cell_key(const LevelPos &p) {
    return ((uint64_t)p.c[2] << 42) ^ ((uint64_t)p.c[1] << 21) ^ (uint64_t)p.c[0];
}

// Runs are confined to aligned groups of RUN_LANES = 8 lanes so the merge is three DPP row shifts (no LDS).
// emit = this lane is the LAST lane of a run of valid lanes with equal cell (always true for hashed levels)
constexpr int RUN_LANES = 8;

template <int OFF>
__device__ __forceinline__ float dpp_shr(float v) {               // value of lane - OFF in the 16-lane row, else 0
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x110 | OFF, 0xf, 0xf, true));
}
template <int OFF>
__device__ __forceinline__ int dpp_shr_i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x110 | OFF, 0xf, 0xf, true); }

__device__ __forceinline__ bool run_tail(bool dense, bool have, uint64_t key, int lane, bool &head) {
    if (!dense) { head = true; return have; }
    const uint64_t kp = __shfl_up(key, 1, 64), kn = __shfl_down(key, 1, 64);
    const int hp = __shfl_up((int)have, 1, 64), hn = __shfl_down((int)have, 1, 64);
    const int sub = lane & (RUN_LANES - 1);
    head = !(sub > 0 && hp && have && kp == key);
    return have && (sub == RUN_LANES - 1 || !(hn && kn == key));
}

// segmented inclusive scan: after it the tail lane of every run holds the run's sums
template <int OFF>
__device__ __forceinline__ void run_merge_step(int &f, float (&v0)[8], float (&v1)[8]) {
    const float take = f ? 0.f : 1.f;                             // heads (and lanes already joined to one) keep their value
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        v0[c] = fmaf(dpp_shr<OFF>(v0[c]), take, v0[c]);
        v1[c] = fmaf(dpp_shr<OFF>(v1[c]), take, v1[c]);
    }
    f |= dpp_shr_i<OFF>(f);
}

__device__ __forceinline__ void run_merge(bool head, int lane, float (&v0)[8], float (&v1)[8]) {
    int f = head ? 1 : 0;
    run_merge_step<1>(f, v0, v1);
    run_merge_step<2>(f, v0, v1);
    run_merge_step<4>(f, v0, v1);
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

// ---- 1. scatter (counting sort by bin inside the workgroup, one contiguous slab per level) --------------------
__device__ __forceinline__ void table_atomic(float *grad_table, uint32_t offset, uint32_t idx, float a, float b) {
    float *gt = grad_table + 2 * ((size_t)offset + idx);
    atomicAdd(gt, a);
    atomicAdd(gt + 1, b);
}

// KIND 0: every level with single-update records (the tangent variant), 1: the pair-record (hashed) levels only,
// 2: the single-update (dense) levels only.  Two launches instead of one let the hashed-level kernel run with a
// 32 KiB staging area and 64 registers: four workgroups per CU.
template <bool TAN, int KIND>
__global__ __launch_bounds__(SC_THREADS, KIND == 1 ? REN_SC_WAVES_PAIR : REN_SC_WAVES) void bin_scatter_kernel(
    GridDev g, BinTab bt, SampleArgs a, Workspace ws, float *__restrict__ grad_table) {
    __shared__ uint32_t hist[MAX_BINS_PER_LEVEL], loc[MAX_BINS_PER_LEVEL + 1];
    __shared__ __attribute__((aligned(16))) unsigned char stage[KIND == 1 ? SC_PAIRS * 16 : SC_ENTRIES * 12];   // float4[]  |  key u32[] + v float2[]
    __shared__ float wave_max[SC_THREADS / 64];
    uint32_t *st_key = reinterpret_cast<uint32_t *>(stage);
    float2 *st_v = reinterpret_cast<float2 *>(stage + SC_ENTRIES * 4);
    float4 *st_p = reinterpret_cast<float4 *>(stage);
    const int64_t chunk = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int64_t i = chunk * SC_THREADS + tid;
    const bool inb = i < a.n;
    float u[3] = {0.f, 0.f, 0.f}, ud[3] = {0.f, 0.f, 0.f};
    if (inb) unit_pos<TAN>(a, i, u, ud);
    constexpr bool pairs = KIND == 1;
#pragma unroll 1
    for (int lvl = 0; lvl < g.n_levels; ++lvl) {
        if (KIND == 1 && !bt.pair[lvl]) continue;
        if (KIND == 2 && bt.pair[lvl]) continue;
        if (tid < MAX_BINS_PER_LEVEL) hist[tid] = 0;
        // level maximum seen so far: loaded here, needed after the ranking (the read may be stale: that only costs an atomic)
        uint32_t *lmax_slot = ws.level_max + (lvl * LMAX_SLOTS + (int)(chunk % LMAX_SLOTS)) * LMAX_STRIDE;
        uint32_t lmax_seen = 0;
        if (tid == SC_THREADS - 1) lmax_seen = __builtin_nontemporal_load(lmax_slot);
        lds_barrier();                                           // also: the previous level's slab has left the staging area
        const uint32_t res = g.res[lvl], size = g.size[lvl];
        const bool hashed = g.hashed[lvl] != 0;
        const float scale = g.scale[lvl];
        float d0 = 0.f, d1 = 0.f, e0 = 0.f, e1 = 0.f;
        const bool valid = inb && load_dfeat<TAN>(a, g.n_levels, lvl, i, d0, d1, e0, e1);
        const LevelPos p = level_pos(u[0], u[1], u[2], scale);
        const float wx1 = p.w[0], wy1 = p.w[1], wz1 = p.w[2], wx0 = 1.f - wx1, wy0 = 1.f - wy1, wz0 = 1.f - wz1;
        if (!valid) { d0 = 0.f; d1 = 0.f; e0 = 0.f; e1 = 0.f; }
        uint32_t key[8];                                           // singles: rank << 19 | table index in level
        float v0[8], v1[8];                                        // pairs:   key[j] = record code, key[4 + j] = rank
        bool have;
        float vmax = 0.f;
        uint32_t idx[8];
        corner_indices8(p.c[0], p.c[1], p.c[2], res, size, hashed, idx);
        if (pairs) {
            // ---- pair records: (x, x+1) corners share A = wy wz dfeat, their weights are 1 - fx and fx
            const bool same = ((idx[0] ^ idx[1]) >> BIN_SHIFT) == 0;   // idx0 ^ idx1 is the same for the four pairs
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float wyz = ((j & 1) ? wy1 : wy0) * ((j & 2) ? wz1 : wz0);
                v0[j] = wyz * d0; v1[j] = wyz * d1;
            }
            if (valid && !same) {                                  // x + 1 carries past bit 12: once in 8 192 cells
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    table_atomic(grad_table, g.offset[lvl], idx[2 * j], wx0 * v0[j], wx0 * v1[j]);
                    table_atomic(grad_table, g.offset[lvl], idx[2 * j + 1], wx1 * v0[j], wx1 * v1[j]);
                }
            }
            have = valid && same;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t bin = idx[2 * j] >> BIN_SHIFT;
                key[4 + j] = have ? atomicAdd(&hist[bin], 1u) : 0u;
                key[j] = (idx[2 * j] & (BIN_ENTRIES - 1)) | ((idx[2 * j + 1] & (BIN_ENTRIES - 1)) << BIN_SHIFT) |
                         (bin << PAIR_BIN_SHIFT);
            }
            if (valid) vmax = fmaxf(fabsf(d0), fabsf(d1));
        } else {
            const float wxy[4] = {wx0 * wy0, wx1 * wy0, wx0 * wy1, wx1 * wy1};
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float wz = (c & 4) ? wz1 : wz0;
                const float w = wxy[c & 3] * wz;
                v0[c] = w * d0; v1[c] = w * d1;
                if (TAN) {                                             // + d w / dt * d(feature tangent)
                    const float wx = (c & 1) ? wx1 : wx0, wy = (c & 2) ? wy1 : wy0;
                    const float bx = (c & 1) ? ud[0] : -ud[0], by = (c & 2) ? ud[1] : -ud[1], bz = (c & 4) ? ud[2] : -ud[2];
                    const float wdc = scale * (bx * wy * wz + wx * by * wz + wxy[c & 3] * bz);
                    v0[c] += wdc * e0; v1[c] += wdc * e1;
                }
            }
            bool head;
            have = run_tail(!hashed, valid, cell_key(p), lane, head);
            if (!hashed) run_merge(head, lane, v0, v1);
            // upper bound of |update| in this level (scale of the fixed-point sums): interpolation weights are <= 1 and a
            // merged run adds at most 8 lanes, so 8 max|d feature| bounds every update (3 of the 38 bits); with tangents
            // the updates carry the scale * |ud| terms as well, so take them as they are
            if (TAN) {
                if (have) {
#pragma unroll
                    for (int c = 0; c < 8; ++c) vmax = fmaxf(vmax, fmaxf(fabsf(v0[c]), fabsf(v1[c])));
                }
            } else if (valid) {
                vmax = fmaxf(fabsf(d0), fabsf(d1)) * (hashed ? 1.f : 8.f);
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const uint32_t rank = bin_rank(!hashed, have, idx[c] >> BIN_SHIFT, lane, hist);
                key[c] = (rank << 19) | idx[c];                    // idx < 2^19, rank < 4096
            }
        }
        // max |update| of the level: scales the 64-bit fixed-point accumulation of the next kernel
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off, 64));
        if (lane == 0) wave_max[tid >> 6] = vmax;
        lds_barrier();
        if (tid == SC_THREADS - 1) {
            float m = 0.f;
#pragma unroll
            for (int w = 0; w < SC_THREADS / 64; ++w) m = fmaxf(m, wave_max[w]);
            // non-negative floats order like uints
            if (__float_as_uint(m) > lmax_seen) atomicMax(lmax_slot, __float_as_uint(m));
        }
        if (tid < MAX_BINS_PER_LEVEL) {
            const uint32_t cnt = hist[tid];
            uint32_t inc = cnt;                                    // inclusive wave scan over the 64 bins -> local offsets
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t t = __shfl_up(inc, off, 64);
                if (tid >= off) inc += t;
            }
            loc[tid] = inc - cnt;
            if (tid == MAX_BINS_PER_LEVEL - 1) loc[MAX_BINS_PER_LEVEL] = inc;
            ws.dir[((int64_t)lvl * ws.n_wg + chunk) * MAX_BINS_PER_LEVEL + tid] = (inc - cnt) | (cnt << 16);   // start < 2^16... see static_assert
        }
        lds_barrier();
        if (have) {
            if (pairs) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    st_p[loc[key[j] >> PAIR_BIN_SHIFT] + key[4 + j]] = make_float4(__uint_as_float(key[j]), v0[j], v1[j], wx1);
            } else {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const uint32_t ix = key[c] & 0x7FFFFu;
                    const uint32_t pos = loc[ix >> BIN_SHIFT] + (key[c] >> 19);
                    st_key[pos] = ix;
                    st_v[pos] = make_float2(v0[c], v1[c]);
                }
            }
        }
        lds_barrier();
        const uint32_t total = loc[MAX_BINS_PER_LEVEL];
        char *slab = ws.slabs + ((int64_t)lvl * ws.n_wg + chunk) * SLAB_BYTES;
        if (pairs) {
            float4 *out = reinterpret_cast<float4 *>(slab);
            for (uint32_t q = tid; q < total; q += SC_THREADS) out[q] = st_p[q];
        } else {
            float2 *out_v = reinterpret_cast<float2 *>(slab);
            uint16_t *out_i = reinterpret_cast<uint16_t *>(slab + (size_t)SC_ENTRIES * 8);
            for (uint32_t q = tid; q < total; q += SC_THREADS) {
                out_v[q] = st_v[q];
                out_i[q] = (uint16_t)(st_key[q] & (BIN_ENTRIES - 1));
            }
        }
    }
}
static_assert(SC_ENTRIES <= 65535, "directory entries pack start and count in 16 bits each");

// ---- 2. accumulate one bin part in LDS, flush to the gradient table -------------------------------------------
// LDS float atomics are lane-serialised on gfx950 (ds_add_f32: 0.37 lanes/clk/CU measured), integer
// ones are 3x faster, so the sums are formed in 64-bit fixed point: q = round(v * 2^k) with
// 2^k * max|v| ~ 2^38, i.e. a quantum of 4e-12 of the level's largest update; 2^25 updates per part
// cannot overflow, and the result is MORE accurate than an fp32 running sum.
__device__ __forceinline__ long long to_fixed(float v, double scale) {
    return __double2ll_rn((double)v * scale);
}

constexpr int ACC_THREADS = 1024, ACC_TILE = 1024;    // directory entries staged per round (one per thread)

__global__ __launch_bounds__(ACC_THREADS) void bin_accumulate_kernel(GridDev g, BinTab bt, Workspace ws,
                                                                     float *__restrict__ grad_table) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long acc[];   // acc0[8192] | acc1[8192] | dir tile
    unsigned long long *acc0 = acc, *acc1 = acc + BIN_ENTRIES;
    uint32_t *tile = reinterpret_cast<uint32_t *>(acc + 2 * BIN_ENTRIES);
    const int tid = threadIdx.x;
    int lvl = 0;
    while (lvl + 1 < g.n_levels && (int)blockIdx.x >= bt.first_part[lvl + 1]) ++lvl;
    const int ppb = bt.parts_per_bin[lvl];
    const int rel = (int)blockIdx.x - bt.first_part[lvl];
    const int bin = rel / ppb, part = rel - bin * ppb;
    const int64_t w0 = ws.n_wg * part / ppb, w1 = ws.n_wg * (part + 1) / ppb;
    for (int e = tid; e < 2 * BIN_ENTRIES; e += ACC_THREADS) acc[e] = 0ull;
    int ex;
    uint32_t lmax = 0;
    for (int k = 0; k < LMAX_SLOTS; ++k) lmax = max(lmax, ws.level_max[(lvl * LMAX_SLOTS + k) * LMAX_STRIDE]);
    (void)frexpf(__uint_as_float(lmax), &ex);                     // level max < 2^ex
    const double scale = ldexp(1.0, 38 - ex), inv_scale = ldexp(1.0, ex - 38);
    const uint32_t *dir = ws.dir + (int64_t)lvl * ws.n_wg * MAX_BINS_PER_LEVEL + bin;
    const char *slabs = ws.slabs + (int64_t)lvl * ws.n_wg * SLAB_BYTES;
    const bool pairs = bt.pair[lvl] != 0;
    auto add_pair = [&](const float4 &r) {
        const uint32_t code = __float_as_uint(r.x);
        const uint32_t i0 = code & (BIN_ENTRIES - 1), i1 = (code >> BIN_SHIFT) & (BIN_ENTRIES - 1);
        const float f0 = 1.f - r.w;
        atomicAdd(&acc0[i0], (unsigned long long)to_fixed(f0 * r.y, scale));      // ds_add_u64
        atomicAdd(&acc1[i0], (unsigned long long)to_fixed(f0 * r.z, scale));
        atomicAdd(&acc0[i1], (unsigned long long)to_fixed(r.w * r.y, scale));
        atomicAdd(&acc1[i1], (unsigned long long)to_fixed(r.w * r.z, scale));
    };
    uint32_t next = w0 + tid < w1 ? dir[(w0 + tid) * MAX_BINS_PER_LEVEL] : 0u;     // directory column, one tile ahead
    for (int64_t t0 = w0; t0 < w1; t0 += ACC_TILE) {
        __syncthreads();                                          // zeroing / previous tile done
        tile[tid] = next;
        const int64_t tn = t0 + ACC_TILE + tid;
        next = tn < w1 ? dir[tn * MAX_BINS_PER_LEVEL] : 0u;
        __syncthreads();
        const int n_tile = (int)(w1 - t0 < ACC_TILE ? w1 - t0 : ACC_TILE);
        if (pairs) {
            // half a wave per slab (a hashed bin holds ~32 +- 6 of a slab's 2 048 records), four slabs in flight
            const int hw = tid >> 5, hl = tid & 31;
            for (int j = hw; j < n_tile; j += 4 * (ACC_THREADS / 32)) {
                uint32_t st[4], cn[4];
                const float4 *rec[4];
                float4 r[4][2];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int ju = j + u * (ACC_THREADS / 32);
                    const uint32_t d = ju < n_tile ? tile[ju] : 0u;
                    st[u] = d & 0xFFFFu; cn[u] = d >> 16;
                    rec[u] = reinterpret_cast<const float4 *>(slabs + (t0 + ju) * SLAB_BYTES) + st[u];
#pragma unroll
                    for (int k = 0; k < 2; ++k)
                        if (hl + 32 * k < cn[u]) r[u][k] = rec[u][hl + 32 * k];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
#pragma unroll
                    for (int k = 0; k < 2; ++k)
                        if (hl + 32 * k < cn[u]) add_pair(r[u][k]);
                    for (uint32_t q = hl + 64; q < cn[u]; q += 32) add_pair(rec[u][q]);    // crowded bin (clustered samples)
                }
            }
        } else {
            // a wave per slab: dense bins hold anything from nothing to the whole slab
            const int wv = tid >> 6, ln = tid & 63;
            for (int j = wv; j < n_tile; j += ACC_THREADS / 64) {
                const uint32_t d = tile[j];
                const uint32_t st = d & 0xFFFFu, cn = d >> 16;
                if (cn == 0) continue;
                const char *slab = slabs + (t0 + j) * SLAB_BYTES;
                const float2 *pv = reinterpret_cast<const float2 *>(slab) + st;
                const uint16_t *pi = reinterpret_cast<const uint16_t *>(slab + (size_t)SC_ENTRIES * 8) + st;
                uint32_t q = ln;
                for (; q + 3 * 64 < cn; q += 4 * 64) {             // 4 independent loads in flight per lane
                    uint32_t ix[4]; float2 v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) { ix[u] = pi[q + u * 64]; v[u] = pv[q + u * 64]; }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        atomicAdd(&acc0[ix[u]], (unsigned long long)to_fixed(v[u].x, scale));
                        atomicAdd(&acc1[ix[u]], (unsigned long long)to_fixed(v[u].y, scale));
                    }
                }
                for (; q < cn; q += 64) {
                    const uint32_t ix = pi[q];
                    const float2 v = pv[q];
                    atomicAdd(&acc0[ix], (unsigned long long)to_fixed(v.x, scale));
                    atomicAdd(&acc1[ix], (unsigned long long)to_fixed(v.y, scale));
                }
            }
        }
    }
    __syncthreads();
    const uint32_t first = (uint32_t)bin << BIN_SHIFT;             // first entry of the bin in its level
    const uint32_t lim = g.size[lvl] > first ? g.size[lvl] - first : 0;
    float2 *gt = reinterpret_cast<float2 *>(grad_table) + g.offset[lvl] + first;
    for (uint32_t k = tid; k < BIN_ENTRIES && k < lim; k += ACC_THREADS) {
        const long long qa = (long long)acc0[k], qb = (long long)acc1[k];
        if (qa == 0 && qb == 0) continue;
        const float a = (float)((double)qa * inv_scale), b = (float)((double)qb * inv_scale);
        if (ppb == 1) {                                            // exclusive owner: plain coalesced RMW
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

struct Layout { size_t level_max, dir, slabs, total; int64_t n_wg; };

Layout make_layout(int64_t n) {
    Layout L;
    L.n_wg = (n + SC_THREADS - 1) / SC_THREADS;
    size_t o = 0;
    L.level_max = o; o = align256(o + LMAX_WORDS * 4);
    L.dir = o; o = align256(o + (size_t)REN_MAX_LEVELS * L.n_wg * MAX_BINS_PER_LEVEL * 4);
    L.slabs = o; o = align256(o + (size_t)REN_MAX_LEVELS * L.n_wg * SLAB_BYTES);
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
    if (n >= ((int64_t)1 << 28)) return REN_ERR_UNSUPPORTED;      // 2^25 updates per accumulate part keep the fixed-point sums exact
    BinTab bt;
    const char *no_pairs = getenv("REN_HGB_NO_PAIRS");             // verification knob: single-update records everywhere
    const bool use_pairs = !tan.dfeatd && !(no_pairs && no_pairs[0] == '1');
    int nb = 0, np = 0;
    bool any_pair = false, any_single = false;
    for (int l = 0; l < REN_MAX_LEVELS; ++l) {
        bt.bin_base[l] = nb;
        bt.first_part[l] = np;
        bt.pair[l] = 0;
        bt.parts_per_bin[l] = 1;
        if (l < g.n_levels) {
            if (g.size[l] > (uint32_t)(MAX_BINS_PER_LEVEL << BIN_SHIFT)) return REN_ERR_UNSUPPORTED;
            const int bins = (int)((g.size[l] + BIN_ENTRIES - 1) >> BIN_SHIFT);
            if (g.hashed[l]) {
                bt.pair[l] = use_pairs ? 1 : 0;                   // 4 pair records instead of 8 updates per sample
            } else {
                // dense level: split every bin's slabs into ranges of about PART_RECORDS updates (upper bound 8 n / bins)
                int64_t ppb = (8 * n / bins + PART_RECORDS - 1) / PART_RECORDS;
                const int64_t n_wg = (n + SC_THREADS - 1) / SC_THREADS;
                if (ppb > n_wg) ppb = n_wg;
                bt.parts_per_bin[l] = (int)(ppb < 1 ? 1 : (ppb > MAX_PARTS_PER_BIN ? MAX_PARTS_PER_BIN : ppb));
            }
            (bt.pair[l] ? any_pair : any_single) = true;
            nb += bins;
            np += bins * bt.parts_per_bin[l];
        }
    }
    bt.bin_base[REN_MAX_LEVELS] = nb;
    bt.first_part[REN_MAX_LEVELS] = np;
    if (n == 0) return REN_OK;
    ren_scene_dev sc = {};
    if (scene) sc = ren_make_scene(scene);
    const Layout L = make_layout(n);
    char *w = (char *)workspace;
    Workspace ws;
    ws.level_max = (uint32_t *)(w + L.level_max);
    ws.dir = (uint32_t *)(w + L.dir);
    ws.slabs = w + L.slabs;
    ws.n_wg = L.n_wg;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(ws.level_max, 0, LMAX_WORDS * 4, st) != hipSuccess) return REN_ERR_LAUNCH;
    SampleArgs a;
    a.layout = layout; a.dfeat = dfeat; a.x_unit = x_unit; a.sc = sc; a.rays_o = rays_o; a.rays_d = rays_d;
    a.ray_indices = ray_indices; a.t_starts = t_starts; a.t_ends = t_ends; a.n = n; a.tan = tan;
    const dim3 sgrd((unsigned)L.n_wg), sblk(SC_THREADS);
    if (tan.dfeatd) hipLaunchKernelGGL((bin_scatter_kernel<true, 0>), sgrd, sblk, 0, st, g, bt, a, ws, grad_table);
    else {
        if (any_pair)   hipLaunchKernelGGL((bin_scatter_kernel<false, 1>), sgrd, sblk, 0, st, g, bt, a, ws, grad_table);
        if (any_single) hipLaunchKernelGGL((bin_scatter_kernel<false, 2>), sgrd, sblk, 0, st, g, bt, a, ws, grad_table);
    }
    const size_t acc_lds = 2 * BIN_ENTRIES * sizeof(unsigned long long) + ACC_TILE * 4;
    (void)hipFuncSetAttribute((const void *)bin_accumulate_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)acc_lds);
    hipLaunchKernelGGL(bin_accumulate_kernel, dim3((unsigned)np), dim3(ACC_THREADS), acc_lds, st, g, bt, ws, grad_table);
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
