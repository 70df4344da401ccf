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
//   1. count    per (level, bin) number of updates; one thread walks all levels of its samples, the
//               histogram stays in LDS and reaches the global counters once per 1 024 samples
//   2. offsets  exclusive scan of the ~770 bin counts + work partition  (one workgroup)
//   3. scatter  one sample per thread, the workgroup walks the levels: per level rank the 512 x 8
//               corner updates by bin (LDS counters), place them in an LDS staging area, and append the
//               runs to the bins' regions of an HBM staging buffer {u16 local index | float2 value}
//               with fully coalesced stores (288 GB of HBM is what makes a 10 B x 128 x n buffer
//               -- 21.5 GB at n = 16.8 M -- a reasonable thing to do)
//   4. accumulate  stream a bin part (coalesced), 64-bit fixed-point ds_add_u64 into LDS, flush.
// HBM traffic: 10 B written + 10 B read per update = 2.56 KB/sample (vs 2 KB of atomic RMW it
// replaces) but all of it streaming; global atomic requests drop from 128 to ~0.1 per sample.
//
// Pair records (hashed levels, round 2): corners (x, y, z) and (x+1, y, z) of a cell hash to indices that differ
// only in their low bits (x ^ (x+1) = 2^(k+1) - 1), so they fall into the same 8 192-entry bin except with
// probability 2^-13, and their updates are (1 - fx) A and fx A with ONE shared A = wy wz dfeat.  A hashed level
// therefore stages 4 records {u32 idx0 | idx1 << 13, float2 A, float fx} = 16 B per sample instead of 8 x 10 B:
// 8 B per update, half as many ranking atomics and LDS placements in the scatter, and one 16-byte store / load
// per record.  Dense levels keep the single-update records (their run merge sums different fx).
#include <cstdlib>
#include "ren_hashgrid_common.h"

namespace {

constexpr int BIN_SHIFT = 13;
constexpr int BIN_ENTRIES = 1 << BIN_SHIFT;          // 8 192 table entries (x2 features x int64 = 128 KiB LDS)
constexpr int MAX_BINS_PER_LEVEL = 64;
constexpr int MAX_BINS = REN_MAX_LEVELS * MAX_BINS_PER_LEVEL;
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
constexpr int SC_ENTRIES = SC_THREADS * 8;           // staged updates per level pass = 48 KiB
constexpr int CNT_THREADS = 1024, CNT_SAMPLES = 1024; // count workgroup: one sample per thread, all levels (4 per thread: a 4x longer latency chain, 72 us however small the batch)
// Updates per accumulate workgroup ("part").  A part follows n (a fixed 2 M-entry part is a ~1 ms single-workgroup
// tail when the whole call is only a few million updates: occupancy-grid sampling, ~10 samples per ray), and it is
// the capacity of a hashed bin whenever there are hashed levels, so that every hashed bin is ONE part: a bin cut
// in two flushes both halves with 16 384 float atomics (memory-side, ~18 G/s chip-wide) instead of one coalesced
// read-modify-write.  <= 2^23 updates of magnitude < 2^38 cannot overflow the 64-bit fixed-point sums.
constexpr int64_t PART_ENTRIES_MIN = 1 << 16, PART_ENTRIES_MAX = 1 << 23;
// Pair-record (hashed) bins are split into N_SUB sub-regions, one per XCD: a workgroup appends to the sub-region of the XCD
// it runs on (its own cursor).  Runs are unaligned 512-byte pieces, so the 128-byte line at every run boundary is shared by
// two runs; with ONE cursor per bin the two halves of such a line usually come from workgroups on different XCDs, i.e. from
// two mutually incoherent L2s, each of which writes its part back as a masked partial line (tools/append_bench.hip: 4.1 TB/s
// against 7.2 TB/s for line-aligned runs).  With a sub-region per XCD both halves pass through the same L2, which merges them.
// Correctness does not depend on the id being an XCD: any value in [0, N_SUB) partitions the appends.
constexpr int N_SUB = 8;
__device__ __forceinline__ int xcd_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return (int)(x & (N_SUB - 1));
}
inline int64_t part_entries_for(int64_t n, int64_t hashed_cap) {
    int64_t p = hashed_cap;
    if (p == 0) {                                    // dense levels only: ~1 024 parts
        p = PART_ENTRIES_MIN;
        while (p < (1 << 21) && p * 1024 < n * 128) p <<= 1;
    }
    return p < PART_ENTRIES_MIN ? PART_ENTRIES_MIN : (p > PART_ENTRIES_MAX ? PART_ENTRIES_MAX : p);
}
// max |update| per level is published with atomicMax: spread over LMAX_SLOTS cache lines per level and
// only raised when the value actually grows, so the workgroups do not queue on one memory channel.
constexpr int LMAX_SLOTS = 8, LMAX_STRIDE = 32;      // u32 words between slots (128 B)
constexpr int LMAX_WORDS = REN_MAX_LEVELS * LMAX_SLOTS * LMAX_STRIDE;

struct BinTab {
    int bin_base[REN_MAX_LEVELS + 1];                // first global bin of each level
    // Hashed levels are NOT counted: the hash spreads the 8 n updates of a level evenly over its bins (binomial,
    // sigma/mean ~ 7e-4 at config B), so every bin of a hashed level gets a region of `cap` entries (mean + 2 %
    // + 4096); an update that would not fit falls back to a global atomic (never seen, kept for correctness).
    // cap = 0: dense level, regions sized by the count pass.  The count pass looks at one sample block in
    // `cnt_stride` (rays of a batch are i.i.d., so blocks are exchangeable) and the offsets kernel scales the
    // sampled count back up with a 25 % + 4096 margin; the same overflow fallback keeps it exact.
    uint32_t cap[REN_MAX_LEVELS];
    uint32_t pair[REN_MAX_LEVELS];                   // 1: the level's bins hold 16-byte pair records (cap counts records)
    uint32_t skip[REN_MAX_LEVELS];                   // 1: level not part of this call (level_mask of the *_levels entry points)
    int sub_by_xcd;                                  // pair bins: sub-region = XCD of the workgroup (REN_KNOB_HGB_SUBREGION)
    int cnt_stride;
    int halve;                                       // test hook (REN_HGB_HALVE_REGIONS=1): force the overflow path
};

struct Part {
    uint32_t gbin, single;
    uint64_t begin, end;                             // entries, relative to the bin's region
};

// Staging pool: bin b owns 16-byte slots [bin_start[b], bin_start[b+1]).  Single-update regions hold
// float2 v[bin_cap] followed by u16 idx[bin_cap]; pair regions hold float4 records[bin_cap].
struct Workspace {
    uint32_t *counts, *level_max, *cursors, *n_parts;     // level_max: float bits of max |update|, [level][slot] one line each
    uint32_t *bin_cap;                                    // entries the region can take
    uint64_t *bin_start;                                  // first 16-byte slot of the region
    Part *parts;
    char *pool;
};
constexpr int PAIR_BIN_SHIFT = 26;                        // pair record code: idx0 | idx1 << 13 | bin << 26 (bin: LDS staging only)

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
    const int64_t *n_dev;                            // device-side sample count (ren_eff_n), or NULL
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

// Runs are confined to aligned groups of RUN_LANES = 16 lanes (one DPP row) so the merge is four DPP row shifts (no LDS).
// emit = this lane is the LAST lane of a run of valid lanes with equal cell (always true for hashed levels)
constexpr int RUN_LANES = 16;

template <int OFF>
__device__ __forceinline__ float dpp_shr(float v) {               // value of lane - OFF in the 16-lane row, else 0
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x110 | OFF, 0xf, 0xf, true));
}
template <int OFF>
__device__ __forceinline__ int dpp_shr_i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x110 | OFF, 0xf, 0xf, true); }

// `key`: index of the cell's first corner.  Equal keys mean equal index sets for all 8 corners (the other seven are the
// first plus per-level constants, also after the wrap of positions outside the unit cube), which is all a merge needs; a
// lane without an update takes a key no other lane has, so ONE neighbour exchange each way decides head and tail
// (round 3 shuffled a 64-bit cell key and the `have` flag separately: six LDS permutes per level instead of two).
__device__ __forceinline__ bool run_tail(bool dense, bool have, uint32_t key, int lane, bool &head) {
    if (!dense) { head = true; return have; }
    const uint32_t k = have ? key : 0x80000000u | (uint32_t)lane;        // table indices are < 2^19
    const uint32_t kp = __shfl_up(k, 1, 64), kn = __shfl_down(k, 1, 64);
    const int sub = lane & (RUN_LANES - 1);
    head = !(sub > 0 && have && kp == k);
    return have && (sub == RUN_LANES - 1 || kn != k);
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
    run_merge_step<8>(f, v0, v1);
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

// The eight corners of a dense cell at once.  bin_rank costs one LDS round trip per corner (the leader's atomic has to
// come back before the next corner can start): ~1 000 cycles of latency per wave and level, which is what the dense
// levels' scatter was made of.  When every emitting lane of the wave has the same bin for corner c (all corners of
// levels 0-2, most waves of levels 3-4), the leader issues the eight counter bumps back to back and the wave waits once.
__device__ __forceinline__ void bin_rank8_dense(bool emit, const uint32_t (&bin)[8], int lane, uint32_t *hist, uint32_t (&rank)[8]) {
    const uint64_t m = __ballot(emit);
#pragma unroll
    for (int c = 0; c < 8; ++c) rank[c] = 0;
    if (!m) return;
    const int leader = __ffsll((unsigned long long)m) - 1;
    uint32_t lb[8];
    bool mixed = false;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        lb[c] = __shfl(bin[c], leader, 64);
        mixed |= emit && bin[c] != lb[c];
    }
    if (__ballot(mixed)) {                                         // wave-uniform branch
#pragma unroll
        for (int c = 0; c < 8; ++c) rank[c] = bin_rank(true, emit, bin[c], lane, hist);
        return;
    }
    const uint32_t cnt = (uint32_t)__popcll(m), pre = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    uint32_t base[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (lane == leader) {
#pragma unroll
        for (int c = 0; c < 8; ++c) base[c] = atomicAdd(&hist[lb[c]], cnt);   // same bin twice: ordered, distinct bases
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) rank[c] = __shfl(base[c], leader, 64) + pre;
}

// ---- 1. count ------------------------------------------------------------------------------------
// One thread walks all levels of its 4 samples; the per-(level, bin) histogram lives in LDS and reaches
// the global counters once per workgroup (the counters are memory-side atomics, ~18 G/s chip-wide).
template <bool TAN>
__global__ __launch_bounds__(CNT_THREADS) void bin_count_kernel(GridDev g, BinTab bt, SampleArgs a,
                                                                uint32_t *__restrict__ counts) {
    __shared__ uint32_t hist[MAX_BINS];
    for (int t = threadIdx.x; t < MAX_BINS; t += CNT_THREADS) hist[t] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
#pragma unroll 1
    for (int k = 0; k < CNT_SAMPLES / CNT_THREADS; ++k) {
        const int64_t i = (int64_t)blockIdx.x * bt.cnt_stride * CNT_SAMPLES + k * CNT_THREADS + threadIdx.x;
        const bool inb = i < ren_eff_n(a.n, a.n_dev);
        float u[3] = {0.f, 0.f, 0.f}, ud[3];
        if (inb) unit_pos<TAN>(a, i, u, ud);
#pragma unroll 1
        for (int lvl = 0; lvl < g.n_levels; ++lvl) {
            if (bt.cap[lvl] || bt.skip[lvl]) continue;               // hashed level: capacity-sized regions, no count
            float d0, d1, e0, e1;
            const bool have = inb && (a.dfeat == nullptr || load_dfeat<TAN>(a, g.n_levels, lvl, i, d0, d1, e0, e1));
            const LevelPos p = level_pos(u[0], u[1], u[2], g.scale[lvl]);
            const uint32_t res = g.res[lvl], size = g.size[lvl];
            const bool hashed = g.hashed[lvl] != 0;
            bool head;
            uint32_t idx[8];
            corner_indices8(p.c[0], p.c[1], p.c[2], res, size, hashed, idx);
            const bool emit = run_tail(!hashed, have, idx[0], lane, head);
#pragma unroll
            for (int c = 0; c < 8; ++c) (void)bin_rank(!hashed, emit, idx[c] >> BIN_SHIFT, lane, hist + lvl * MAX_BINS_PER_LEVEL);
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < MAX_BINS; t += CNT_THREADS) {
        const int lvl = t / MAX_BINS_PER_LEVEL, bin = t % MAX_BINS_PER_LEVEL;
        if (lvl < g.n_levels && bin < bt.bin_base[lvl + 1] - bt.bin_base[lvl] && hist[t])
            atomicAdd(&counts[bt.bin_base[lvl] + bin], hist[t]);
    }
}

// ---- 2. offsets: exclusive scan of the region sizes (count for dense bins, capacity for hashed bins) ------
__global__ __launch_bounds__(MAX_BINS) void bin_offsets_kernel(int n_bins, BinTab bt, uint64_t capacity_slots,
                                                               const uint32_t *__restrict__ counts,
                                                               uint32_t *__restrict__ cursors,
                                                               uint32_t *__restrict__ bin_cap,
                                                               uint64_t *__restrict__ bin_start) {
    __shared__ uint64_t s_cnt[MAX_BINS];
    const int t = threadIdx.x;
    uint64_t c = 0, slots = 0;
    bool pair = false;
    if (t < n_bins) {
        int lvl = 0;
        while (lvl + 1 < REN_MAX_LEVELS && t >= bt.bin_base[lvl + 1]) ++lvl;
        pair = bt.pair[lvl] != 0;
        c = bt.skip[lvl] ? 0 : bt.cap[lvl] ? bt.cap[lvl] : counts[t];
        if (!bt.skip[lvl] && !bt.cap[lvl] && bt.cnt_stride > 1) c = c * bt.cnt_stride + c * bt.cnt_stride / 4 + 4096;
        if (bt.halve) c = c / 2;
        c &= ~(uint64_t)7;                                         // idx[] of a single-update region stays 16-byte aligned
        if (pair) c = (c / N_SUB) & ~(uint64_t)7;                  // entries per sub-region (one per XCD)
        slots = pair ? c * N_SUB : (c * 10 + 15) / 16;
        for (int x = 0; x < N_SUB; ++x) cursors[t * N_SUB + x] = 0;
    }
    s_cnt[t] = slots;
    __syncthreads();
    for (int off = 1; off < MAX_BINS; off <<= 1) {
        const uint64_t a = t >= off ? s_cnt[t - off] : 0;
        __syncthreads();
        s_cnt[t] += a;
        __syncthreads();
    }
    // regions never leave the workspace: whatever does not fit takes the overflow path of the scatter
    if (t < n_bins) {
        const uint64_t start = s_cnt[t] - slots;
        uint64_t room = start < capacity_slots ? capacity_slots - start : 0;
        if (room > slots) room = slots;
        if (room < slots) c = (pair ? room / N_SUB : room * 16 / 10) & ~(uint64_t)7;
        bin_start[t] = start < capacity_slots ? start : capacity_slots;
        bin_cap[t] = (uint32_t)c;
    }
    if (t == n_bins - 1) bin_start[n_bins] = s_cnt[t] < capacity_slots ? s_cnt[t] : capacity_slots;
}

// ---- 3b. work partition of the accumulate pass, from the ACTUAL fill of every bin region ---------------------
__global__ __launch_bounds__(MAX_BINS) void bin_partition_kernel(int n_bins, uint64_t PART_ENTRIES, BinTab bt,
                                                                 const uint32_t *__restrict__ cursors,
                                                                 const uint32_t *__restrict__ bin_cap,
                                                                 Part *__restrict__ parts, uint32_t *__restrict__ n_parts) {
    __shared__ uint32_t s_np[MAX_BINS];
    const int t = threadIdx.x;
    uint64_t c = 0;
    bool pair = false;
    if (t < n_bins) {
        int lvl = 0;
        while (lvl + 1 < REN_MAX_LEVELS && t >= bt.bin_base[lvl + 1]) ++lvl;
        pair = bt.pair[lvl] != 0;
        // overflowing updates went to the table directly
        for (int x = 0; x < (pair ? N_SUB : 1); ++x) c += cursors[t * N_SUB + x] < bin_cap[t] ? cursors[t * N_SUB + x] : bin_cap[t];
    }
    // a pair bin is ONE part (the part size is the capacity of a hashed bin): it walks its N_SUB sub-regions itself
    const uint32_t np = pair ? (c ? 1u : 0u) : (uint32_t)((c + PART_ENTRIES - 1) / PART_ENTRIES);
    s_np[t] = np;
    __syncthreads();
    for (int off = 1; off < MAX_BINS; off <<= 1) {
        const uint32_t b = t >= off ? s_np[t - off] : 0;
        __syncthreads();
        s_np[t] += b;
        __syncthreads();
    }
    if (t == n_bins - 1) n_parts[0] = s_np[t];
    const uint32_t pbase = s_np[t] - np;
    for (uint32_t k = 0; k < np; ++k) {
        Part p;
        p.gbin = t; p.single = np == 1;                         // (pair bins: begin / end unused)
        p.begin = (uint64_t)k * PART_ENTRIES;
        p.end = k + 1 == np ? c : p.begin + PART_ENTRIES;
        parts[pbase + k] = p;
    }
}

// ---- 3. scatter (counting sort by bin inside the workgroup, coalesced append) ----------------------------
// One sample per thread; the workgroup walks the levels and runs one
// rank -> offsets -> LDS placement -> coalesced append pass per level over the same staging area.
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
    __shared__ uint32_t hist[MAX_BINS_PER_LEVEL], loc[MAX_BINS_PER_LEVEL + 1], fit[MAX_BINS_PER_LEVEL];
    // per bin: where staging position q of this pass goes in HBM (pointer - local offset of the bin's run):
    // gp0 = float2 v[] (single updates) or float4 records[] (pairs), gp1 = u16 idx[] (single updates)
    __shared__ char *gp0[MAX_BINS_PER_LEVEL], *gp1[MAX_BINS_PER_LEVEL];
    __shared__ __attribute__((aligned(16))) unsigned char stage[KIND == 1 ? SC_THREADS * 4 * 16 : SC_ENTRIES * 12];   // float4[]  |  key u32[] + v float2[]
    __shared__ float wave_max[SC_THREADS / 64];
    __shared__ int any_overflow;
    uint32_t *st_key = reinterpret_cast<uint32_t *>(stage);
    float2 *st_v = reinterpret_cast<float2 *>(stage + SC_ENTRIES * 4);
    float4 *st_p = reinterpret_cast<float4 *>(stage);
    const int64_t chunk = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int64_t i = chunk * SC_THREADS + tid;
    const bool inb = i < ren_eff_n(a.n, a.n_dev);
    float u[3] = {0.f, 0.f, 0.f}, ud[3] = {0.f, 0.f, 0.f};
    if (inb) unit_pos<TAN>(a, i, u, ud);
    constexpr bool pairs = KIND == 1;
    // which sub-region of a pair bin this workgroup appends to
    const int sub = !pairs ? 0 : bt.sub_by_xcd ? xcd_id() : (int)((chunk >> 3) & (N_SUB - 1));
#pragma unroll 1
    for (int lvl = 0; lvl < g.n_levels; ++lvl) {
        if (bt.skip[lvl]) continue;
        if (KIND == 1 && !bt.pair[lvl]) continue;
        if (KIND == 2 && bt.pair[lvl]) continue;
        if (tid < MAX_BINS_PER_LEVEL) hist[tid] = 0;
        if (tid == 0) any_overflow = 0;
        lds_barrier();                                           // also: previous level's append is done
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
            have = run_tail(!hashed, valid, idx[0], lane, head);
            if (!hashed) run_merge(head, lane, v0, v1);
            // upper bound of |update| in this level (scale of the fixed-point sums): interpolation weights are <= 1 and a
            // merged run adds at most RUN_LANES lanes, so RUN_LANES max|d feature| bounds every update (4 of the 38 bits); with tangents
            // the updates carry the scale * |ud| terms as well, so take them as they are
            if (TAN) {
                if (have) {
#pragma unroll
                    for (int c = 0; c < 8; ++c) vmax = fmaxf(vmax, fmaxf(fabsf(v0[c]), fabsf(v1[c])));
                }
            } else if (valid) {
                vmax = fmaxf(fabsf(d0), fabsf(d1)) * (hashed ? 1.f : (float)RUN_LANES);
            }
            if (!hashed) {
                uint32_t bins[8], rk[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) bins[c] = idx[c] >> BIN_SHIFT;
                bin_rank8_dense(have, bins, lane, hist, rk);
#pragma unroll
                for (int c = 0; c < 8; ++c) key[c] = (rk[c] << 19) | idx[c];
            } else {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const uint32_t rank = bin_rank(false, have, idx[c] >> BIN_SHIFT, lane, hist);
                    key[c] = (rank << 19) | idx[c];                // idx < 2^19, rank < 4096
                }
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
            uint32_t *slot = ws.level_max + (lvl * LMAX_SLOTS + (int)(chunk % LMAX_SLOTS)) * LMAX_STRIDE;
            // non-negative floats order like uints; the plain read may be stale (that only costs an atomic)
            if (__float_as_uint(m) > __builtin_nontemporal_load(slot)) atomicMax(slot, __float_as_uint(m));
        }
        if (tid < MAX_BINS_PER_LEVEL) {
            const int nb = bt.bin_base[lvl + 1] - bt.bin_base[lvl];
            const uint32_t cnt = hist[tid];
            uint32_t inc = cnt;                                    // inclusive wave scan over the 64 bins -> local offsets
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t t = __shfl_up(inc, off, 64);
                if (tid >= off) inc += t;
            }
            loc[tid] = inc - cnt;
            if (tid == MAX_BINS_PER_LEVEL - 1) loc[MAX_BINS_PER_LEVEL] = inc;
            uint32_t room = cnt;
            char *q0 = nullptr, *q1 = nullptr;
            if (tid < nb && cnt) {
                const int gb = bt.bin_base[lvl] + tid;
                const uint32_t at = atomicAdd(&ws.cursors[gb * N_SUB + sub], cnt);   // reserve the run in the bin's (sub-)region
                const uint32_t cap = ws.bin_cap[gb];
                room = at >= cap ? 0u : (cap - at < cnt ? cap - at : cnt);
                char *base = ws.pool + (ws.bin_start[gb] + (pairs ? (uint64_t)sub * cap : 0)) * 16;
                const int64_t first = (int64_t)at - (int64_t)(inc - cnt);    // region entry of staging position 0
                if (pairs) q0 = base + first * 16;
                else { q0 = base + first * 8; q1 = base + (int64_t)cap * 8 + first * 2; }
            }
            fit[tid] = room;                                           // entries of this run that fit the region
            if (room < cnt) any_overflow = 1;
            gp0[tid] = q0; gp1[tid] = q1;
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
        const bool ovf = any_overflow != 0;
        if (pairs) {
            for (uint32_t q = tid; q < total; q += SC_THREADS) {
                const float4 r = st_p[q];
                const uint32_t code = __float_as_uint(r.x), b = code >> PAIR_BIN_SHIFT;
                if (!ovf || q - loc[b] < fit[b]) {
                    reinterpret_cast<float4 *>(gp0[b])[q] = r;
                } else {                                               // region full (capacity-sized hashed bins only)
                    const uint32_t i0 = (b << BIN_SHIFT) | (code & (BIN_ENTRIES - 1));
                    const uint32_t i1 = (b << BIN_SHIFT) | ((code >> BIN_SHIFT) & (BIN_ENTRIES - 1));
                    table_atomic(grad_table, g.offset[lvl], i0, (1.f - r.w) * r.y, (1.f - r.w) * r.z);
                    table_atomic(grad_table, g.offset[lvl], i1, r.w * r.y, r.w * r.z);
                }
            }
            continue;
        }
        for (uint32_t q = tid; q < total; q += SC_THREADS) {
            const uint32_t ix = st_key[q], b = ix >> BIN_SHIFT;
            const float2 v = st_v[q];
            if (!ovf || q - loc[b] < fit[b]) {
                reinterpret_cast<uint16_t *>(gp1[b])[q] = (uint16_t)(ix & (BIN_ENTRIES - 1));
                reinterpret_cast<float2 *>(gp0[b])[q] = v;
            } else {
                table_atomic(grad_table, g.offset[lvl], ix, v.x, v.y);
            }
        }
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
    uint32_t lmax = 0;
    for (int k = 0; k < LMAX_SLOTS; ++k) lmax = max(lmax, ws.level_max[(lvl * LMAX_SLOTS + k) * LMAX_STRIDE]);
    (void)frexpf(__uint_as_float(lmax), &ex);                     // level max < 2^ex
    const double scale = ldexp(1.0, 38 - ex), inv_scale = ldexp(1.0, ex - 38);
    const char *base = ws.pool + ws.bin_start[part.gbin] * 16;
    // the bin's slice of the gradient table, fetched NOW for an exclusive owner: the flush at the end was eight dependent global
    // round trips per thread (load -> add -> store behind a data-dependent `continue`), most of the kernel at small n
    const uint32_t first = ((uint32_t)part.gbin - bt.bin_base[lvl]) << BIN_SHIFT;   // first entry of the bin in its level
    const uint32_t lim = g.size[lvl] > first ? g.size[lvl] - first : 0;
    float2 *gt = reinterpret_cast<float2 *>(grad_table) + g.offset[lvl] + first;
    constexpr int FL = BIN_ENTRIES / 1024;
    float2 gv[FL];
#pragma unroll
    for (int u = 0; u < FL; ++u) {
        const uint32_t k = threadIdx.x + u * 1024;
        gv[u] = (part.single && k < lim) ? gt[k] : make_float2(0.f, 0.f);
    }
    __syncthreads();
    uint64_t e = part.begin + threadIdx.x;
    if (bt.pair[lvl]) {
        const float4 *rec = reinterpret_cast<const float4 *>(base);
        auto add = [&](const float4 &r) {
            const uint32_t code = __float_as_uint(r.x);
            const uint32_t i0 = code & (BIN_ENTRIES - 1), i1 = (code >> BIN_SHIFT) & (BIN_ENTRIES - 1);
            const float f0 = 1.f - r.w;
            atomicAdd(&acc0[i0], (unsigned long long)to_fixed(f0 * r.y, scale));      // ds_add_u64
            atomicAdd(&acc1[i0], (unsigned long long)to_fixed(f0 * r.z, scale));
            atomicAdd(&acc0[i1], (unsigned long long)to_fixed(r.w * r.y, scale));
            atomicAdd(&acc1[i1], (unsigned long long)to_fixed(r.w * r.z, scale));
        };
        const uint32_t cap = ws.bin_cap[part.gbin];
        for (int x = 0; x < N_SUB; ++x) {                          // one sub-region per XCD of the scatter
            const float4 *rx = rec + (size_t)x * cap;
            const uint32_t fill = ws.cursors[part.gbin * N_SUB + x];
            const uint64_t end = fill < cap ? fill : cap;
            uint64_t q = threadIdx.x;
            for (; q + 3 * 1024 < end; q += 4 * 1024) {            // 4 independent 16-byte loads in flight per lane
                float4 r[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) r[u] = rx[q + u * 1024];
#pragma unroll
                for (int u = 0; u < 4; ++u) add(r[u]);
            }
            for (; q < end; q += 1024) add(rx[q]);
        }
    } else {
        const float2 *out_v = reinterpret_cast<const float2 *>(base);
        const uint16_t *out_idx = reinterpret_cast<const uint16_t *>(base + (size_t)ws.bin_cap[part.gbin] * 8);
        for (; e + 3 * 1024 < part.end; e += 4 * 1024) {           // 4 independent loads in flight per lane
            uint32_t ix[4]; float2 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { ix[u] = out_idx[e + u * 1024]; v[u] = out_v[e + u * 1024]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                atomicAdd(&acc0[ix[u]], (unsigned long long)to_fixed(v[u].x, scale));    // ds_add_u64
                atomicAdd(&acc1[ix[u]], (unsigned long long)to_fixed(v[u].y, scale));
            }
        }
        for (; e < part.end; e += 1024) {
            const uint32_t idx = out_idx[e];
            const float2 v = out_v[e];
            atomicAdd(&acc0[idx], (unsigned long long)to_fixed(v.x, scale));
            atomicAdd(&acc1[idx], (unsigned long long)to_fixed(v.y, scale));
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < FL; ++u) {
        const uint32_t k = threadIdx.x + u * 1024;
        if (k >= lim) continue;
        const long long qa = (long long)acc0[k], qb = (long long)acc1[k];
        if (qa == 0 && qb == 0) continue;
        const float a = (float)((double)qa * inv_scale), b = (float)((double)qb * inv_scale);
        if (part.single) {                                         // exclusive owner: plain coalesced RMW (value fetched above)
            float2 v = gv[u];
            v.x += a; v.y += b;
            gt[k] = v;
        } else {
            atomicAdd(&gt[k].x, a);
            atomicAdd(&gt[k].y, b);
        }
    }
}

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

struct Layout { size_t counts, level_max, cursors, n_parts, bin_cap, bin_start, parts, pool, total, slots; int64_t entries, max_parts; };

Layout make_layout(int64_t n) {
    Layout L;
    // at most 128 n single updates of 10 B (16 levels x 8 corners; pair records are 16 B for two) + capacity
    // slack: 2 % of the hashed and 25 % of the dense levels, 4096 per bin, one stride of count blocks; the
    // offsets kernel clamps the regions to this total
    const size_t E = (size_t)n * 140 + (size_t)MAX_BINS * 4096 + 16 * 8 * 16 * CNT_SAMPLES;
    L.entries = (int64_t)E;
    L.slots = (E * 10 + 15) / 16 + MAX_BINS;
    L.max_parts = (int64_t)(E / PART_ENTRIES_MIN) + MAX_BINS + 1;   // room for the smallest part size
    size_t o = 0;
    L.counts = o; o += MAX_BINS * 4;                       // counts | level_max are cleared by one memset
    L.level_max = o; o = align256(o + LMAX_WORDS * 4);
    L.cursors = o; o = align256(o + MAX_BINS * 4 * N_SUB);
    L.n_parts = o; o = align256(o + 4);
    L.bin_cap = o; o = align256(o + MAX_BINS * 4);
    L.bin_start = o; o = align256(o + (MAX_BINS + 1) * 8);
    L.parts = o; o = align256(o + (size_t)L.max_parts * sizeof(Part));
    L.pool = o; o = align256(o + L.slots * 16);
    L.total = o;
    return L;
}

}  // namespace

extern "C" int64_t ren_hashgrid_bwd_binned_workspace_bytes(int64_t n) {
    if (n < 0) return -1;
    return (int64_t)make_layout(n).total;
}

// The call in three phases (all of them entry points, see below): begin = clear + count + offsets (needs the sample stream
// only), scatter = any sample range [first, first + m) of the stream (needs that range's feature gradients), finish =
// partition + accumulate.  The one-shot entry points run begin, one scatter over everything, finish.
// phases: bit 0 begin, bit 1 scatter, bit 2 finish.  `n` is always the size of the WHOLE stream (it sizes the regions);
// the pointers of the per-sample arrays and of dfeat are those of the whole stream too.
static int binned_impl(const ren_grid_desc *grid, float *grad_table, const float *x_unit,
                       const ren_scene_desc *scene, const float *rays_o, const float *rays_d,
                       const int32_t *ray_indices, const float *t_starts, const float *t_ends,
                       int64_t n, int32_t layout, const float *dfeat, void *workspace,
                       void *stream, TanSrc tan, uint32_t level_mask = 0xFFFFFFFFu, int phases = 7,
                       int64_t first = 0, int64_t m = -1, const int64_t *n_dev = nullptr) {
    GridDev g;
    int rc = make_grid(grid, g);
    if (rc) return rc;
    const bool do_begin = phases & 1, do_scatter = phases & 2, do_finish = phases & 4;
    if (m < 0) m = n - first;
    if (!workspace || n < 0 || (layout != 0 && layout != 1) || first < 0 || m < 0 || first + m > n) return REN_ERR_BAD_ARG;
    if ((do_scatter && (!dfeat || !grad_table)) || (do_finish && !grad_table)) return REN_ERR_BAD_ARG;
    if (do_scatter && phases != 7 && ((first & 31) != 0 || tan.dfeatd)) return REN_ERR_BAD_ARG;   // ranges start on a 32-sample block
    const bool from_rays = x_unit == nullptr;
    if ((do_begin || do_scatter) && from_rays && (!scene || !rays_o || !rays_d || !ray_indices || !t_starts || !t_ends))
        return REN_ERR_BAD_ARG;
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
    int64_t hashed_cap = 0;
    const bool use_pairs = !tan.dfeatd && ren_knob(REN_KNOB_HGB_NO_PAIRS) != 1;   // verification knob: single-update records everywhere
    for (int l = 0; l < REN_MAX_LEVELS; ++l) {
        bt.cap[l] = 0;
        bt.pair[l] = 0;
        bt.skip[l] = l < g.n_levels && !((level_mask >> l) & 1u);
        if (l < g.n_levels && g.hashed[l]) {
            const int64_t bins = (g.size[l] + BIN_ENTRIES - 1) >> BIN_SHIFT;
            bt.pair[l] = use_pairs ? 1 : 0;                       // 4 pair records instead of 8 updates per sample
            const int64_t mean = ((use_pairs ? 4 : 8) * n + bins - 1) / bins;
            // pair bins: N_SUB sub-regions (one per XCD) of mean / N_SUB + 0.25 % + 1 024 records each (workgroups are dealt
            // to the XCDs round robin, so the sub-regions fill evenly; what does not fit goes to the table with atomics)
            bt.cap[l] = use_pairs ? (uint32_t)(N_SUB * (((mean / N_SUB + mean / 400 + 1024) + 7) & ~(int64_t)7))
                                  : (uint32_t)(mean + mean / 50 + 4096);
            if ((int64_t)bt.cap[l] > hashed_cap) hashed_cap = bt.cap[l];
        }
    }
    const int64_t part_entries = part_entries_for(n, hashed_cap);
    if (n == 0) return REN_OK;
    ren_scene_dev sc = {};
    if (scene) sc = ren_make_scene(scene);
    const Layout L = make_layout(n);
    char *w = (char *)workspace;
    Workspace ws;
    ws.counts = (uint32_t *)(w + L.counts); ws.level_max = (uint32_t *)(w + L.level_max);
    ws.cursors = (uint32_t *)(w + L.cursors);
    ws.n_parts = (uint32_t *)(w + L.n_parts); ws.bin_cap = (uint32_t *)(w + L.bin_cap);
    ws.bin_start = (uint64_t *)(w + L.bin_start);
    ws.parts = (Part *)(w + L.parts); ws.pool = w + L.pool;
    hipStream_t st = (hipStream_t)stream;
    SampleArgs a;
    a.layout = layout; a.dfeat = dfeat; a.x_unit = x_unit; a.sc = sc; a.rays_o = rays_o; a.rays_d = rays_d;
    a.ray_indices = ray_indices; a.t_starts = t_starts; a.t_ends = t_ends; a.n = n; a.tan = tan;
    a.n_dev = n_dev;                                  // (one-shot calls only: a phased call's ranges are host numbers)
    if (n_dev && phases != 7) return REN_ERR_BAD_ARG;
    bt.halve = ren_knob(REN_KNOB_HGB_HALVE_REGIONS) == 1;
    bt.sub_by_xcd = ren_knob(REN_KNOB_HGB_SUBREGION) != 0;
    const int64_t cnt_blocks = (n + CNT_SAMPLES - 1) / CNT_SAMPLES;
    bt.cnt_stride = cnt_blocks >= 4096 ? 16 : cnt_blocks >= 2048 ? 8 : cnt_blocks >= 1024 ? 4 : 1;   // >= 256 sampled blocks or exact
    bool any_pair = false, any_single = false;
    for (int l = 0; l < g.n_levels; ++l)
        if (!bt.skip[l]) { any_pair |= bt.pair[l] != 0; any_single |= bt.pair[l] == 0; }
    if (do_begin) {
        if (hipMemsetAsync(ws.counts, 0, (MAX_BINS + LMAX_WORDS) * 4, st) != hipSuccess) return REN_ERR_LAUNCH;
        const dim3 cgrd((unsigned)((cnt_blocks + bt.cnt_stride - 1) / bt.cnt_stride)), cblk(CNT_THREADS);
        SampleArgs ac = a;
        if (phases != 7) ac.dfeat = nullptr;          // phased call: the regions are sized before any gradient exists (all samples count)
        if (tan.dfeatd) hipLaunchKernelGGL(bin_count_kernel<true>, cgrd, cblk, 0, st, g, bt, ac, ws.counts);
        else            hipLaunchKernelGGL(bin_count_kernel<false>, cgrd, cblk, 0, st, g, bt, ac, ws.counts);
        hipLaunchKernelGGL(bin_offsets_kernel, dim3(1), dim3(MAX_BINS), 0, st, nb, bt, (uint64_t)L.slots, ws.counts,
                           ws.cursors, ws.bin_cap, ws.bin_start);
    }
    if (!any_pair && !any_single) return REN_OK;
    if (do_scatter && m > 0) {
        // the kernels index samples from 0: a range is the same launch over shifted per-sample pointers
        SampleArgs as = a;
        as.n = m;
        if (first) {
            if (as.x_unit) as.x_unit += 3 * first;
            if (as.ray_indices) { as.ray_indices += first; as.t_starts += first; as.t_ends += first; }
            as.dfeat += layout == 1 ? (first >> 5) * (int64_t)(REN_MAX_LEVELS * 64) : first * (int64_t)g.n_levels * 2;
        }
        const dim3 sgrd((unsigned)((m + SC_THREADS - 1) / SC_THREADS)), sblk(SC_THREADS);
        if (tan.dfeatd) hipLaunchKernelGGL((bin_scatter_kernel<true, 0>), sgrd, sblk, 0, st, g, bt, as, ws, grad_table);
        else {
            if (any_pair)   hipLaunchKernelGGL((bin_scatter_kernel<false, 1>), sgrd, sblk, 0, st, g, bt, as, ws, grad_table);
            if (any_single) hipLaunchKernelGGL((bin_scatter_kernel<false, 2>), sgrd, sblk, 0, st, g, bt, as, ws, grad_table);
        }
    }
    if (do_finish) {
        hipLaunchKernelGGL(bin_partition_kernel, dim3(1), dim3(MAX_BINS), 0, st, nb, (uint64_t)part_entries, bt, ws.cursors, ws.bin_cap, ws.parts,
                           ws.n_parts);
        const size_t acc_lds = 2 * BIN_ENTRIES * sizeof(unsigned long long);
        (void)hipFuncSetAttribute((const void *)bin_accumulate_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)acc_lds);
        const int64_t acc_grid = (int64_t)(L.entries / part_entries) + nb + 1;
        hipLaunchKernelGGL(bin_accumulate_kernel, dim3((unsigned)acc_grid), dim3(1024), acc_lds, st, g, bt, ws, grad_table);
    }
    REN_CHECK_LAUNCH();
}

extern "C" int ren_hashgrid_bwd_binned(const ren_grid_desc *grid, float *grad_table, const float *x_unit,
                                       const ren_scene_desc *scene, const float *rays_o, const float *rays_d,
                                       const int32_t *ray_indices, const float *t_starts, const float *t_ends,
                                       int64_t n, int32_t layout, const float *dfeat, void *workspace,
                                       const int64_t *n_dev, void *stream) {
    return binned_impl(grid, grad_table, x_unit, scene, rays_o, rays_d, ray_indices, t_starts, t_ends, n, layout,
                       dfeat, workspace, stream, TanSrc{nullptr, nullptr, nullptr}, 0xFFFFFFFFu, 7, 0, -1, n_dev);
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

// The same for a subset of the levels (bit l of level_mask): data-parallel training splits the call in two so that the
// all-reduce of the first group's slice of the table gradient runs beside the second group's scatter (engine.py).
extern "C" int ren_hashgrid_bwd_binned_levels(const ren_grid_desc *grid, float *grad_table, const float *x_unit,
                                              const ren_scene_desc *scene, const float *rays_o, const float *rays_d,
                                              const int32_t *ray_indices, const float *t_starts, const float *t_ends,
                                              int64_t n, int32_t layout, const float *dfeat, const float *rays_do,
                                              const float *rays_dd, const float *dfeatd, uint32_t level_mask,
                                              void *workspace, const int64_t *n_dev, void *stream) {
    if ((rays_do || rays_dd || dfeatd) && (!rays_do || !rays_dd || !dfeatd || layout != 1 || x_unit)) return REN_ERR_BAD_ARG;
    return binned_impl(grid, grad_table, x_unit, scene, rays_o, rays_d, ray_indices, t_starts, t_ends, n, layout, dfeat,
                       workspace, stream, TanSrc{rays_do, rays_dd, dfeatd}, level_mask, 7, 0, -1, n_dev);
}

// ---- the same call in phases, so that the scatter of one sample range can run beside whatever produces the next range's
// feature gradients (engine.py: MLP backward of chunk k + 1 on one stream, scatter of chunk k on another) and the bins are
// flushed ONCE at the end.  Every phase takes the WHOLE stream's n and pointers; a range starts on a 32-sample block.
//   begin   clear, count the dense levels' regions from the sample positions (all samples count: no gradient exists yet),
//           offsets.  Needs the sample stream only.
//   scatter samples [first, first + m): appends to the regions (cursors persist across calls).  Stream-ordered after begin.
//   finish  partition + accumulate + flush into grad_table.  Stream-ordered after every scatter.
extern "C" int ren_hashgrid_bwd_binned_begin(const ren_grid_desc *grid, const float *x_unit, const ren_scene_desc *scene,
                                             const float *rays_o, const float *rays_d, const int32_t *ray_indices,
                                             const float *t_starts, const float *t_ends, int64_t n, int32_t layout,
                                             void *workspace, void *stream) {
    return binned_impl(grid, nullptr, x_unit, scene, rays_o, rays_d, ray_indices, t_starts, t_ends, n, layout, nullptr,
                       workspace, stream, TanSrc{nullptr, nullptr, nullptr}, 0xFFFFFFFFu, 1);
}

extern "C" int ren_hashgrid_bwd_binned_scatter(const ren_grid_desc *grid, float *grad_table, const float *x_unit,
                                               const ren_scene_desc *scene, const float *rays_o, const float *rays_d,
                                               const int32_t *ray_indices, const float *t_starts, const float *t_ends,
                                               int64_t n, int32_t layout, const float *dfeat, int64_t first, int64_t m,
                                               void *workspace, void *stream) {
    return binned_impl(grid, grad_table, x_unit, scene, rays_o, rays_d, ray_indices, t_starts, t_ends, n, layout, dfeat,
                       workspace, stream, TanSrc{nullptr, nullptr, nullptr}, 0xFFFFFFFFu, 2, first, m);
}

extern "C" int ren_hashgrid_bwd_binned_finish(const ren_grid_desc *grid, float *grad_table, int64_t n, int32_t layout,
                                              void *workspace, void *stream) {
    return binned_impl(grid, grad_table, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, n, layout, nullptr,
                       workspace, stream, TanSrc{nullptr, nullptr, nullptr}, 0xFFFFFFFFu, 4);
}
