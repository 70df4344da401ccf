// `arch: mlp`: the 8 x 256 trunk of the vanilla-NeRF field (robust_e_nerf/external/mlp.py:26-113 as instantiated by
// NerfMLP, mlp.py:126-205: eight Linear + Softplus(beta = 100) layers, the input encoding concatenated again after
// layer 4) as THREE launches per pass instead of 8 + 7 + 8 dense-layer launches with a 2 KB-per-sample HBM round trip
// between each of them (ren_dense.hip):
//
//   trunk_fwd   a wave keeps its 32-sample blocks in registers through all eight layers: the 256 outputs of a layer are
//               its accumulator tiles (lane = sample), and the accumulator layout IS the next layer's B-operand layout
//               (a k-slot permutation the weight image is pre-permuted for, as in ren_mlp_x.hip), so activations never
//               leave the register file except as the copy saved for the backward pass;
//   trunk_bwd   the same chain backwards over W^T: dz_{l-1} = (W_l^T dz_l) * softplus'(h_{l-1});
//   trunk_dw    dW_l = dz_l^T h_{l-1}, db_l = sum dz_l from the saved copies (samples are the reduction: the
//               transposition goes through LDS as in dense_dw_x_kernel), slab-reduced: deterministic, no atomics.
//
// Weights: `trunk_prep` turns the fp32 parameter block into bf16 MFMA A-fragment images (1 KB = one 32 x 16 fragment,
// lane-linear), once per optimiser step; the kernels stream them through a double-buffered LDS stage with
// `global_load_lds` (1 KB per wave instruction, no staging registers) while the matrix cores work on the previous stage.
// All four waves of the workgroup (one per SIMD: the accumulators and operands of a wave fill the 512-register file)
// share a stage.  One weight fragment read from LDS feeds two MFMAs (bf16 mode: two sample blocks per wave; fp32 mode:
// the six product terms of the three-piece split use three fragments), so LDS bandwidth stays at half of its peak.
//
// Saved activations / pre-activation gradients ("fragment layout", per layer and 32-sample block, lane-linear 1 KB
// pieces = exactly the registers of a wave):
//   bf16 mode (MODE 1)  [blk][c 16][lane 64][8 bf16]   feature kmap(c, lane >> 5, j), sample blk * 32 + (lane & 31)
//   fp32 mode (MODE 6)  [blk][t 8][q 4][lane 64][4 f32] feature 32 t + 8 q + 4 (lane >> 5) + j
// bf16 mode stores what the next layer's matrix product sees anyway (bf16-rounded activations: the weight gradient is
// unchanged by the rounding; the activation derivative is taken from the rounded value).
#include "ren_mlp_common.h"

namespace {

constexpr int T_SKIP = 5;                                // layer 5 takes [h4 | encoding] (mlp.py:60-71,99-113)
__host__ __device__ constexpr int t_nch(int l) { return l == 0 ? 4 : (l == T_SKIP ? 20 : 16); }       // k-chunks of 16
__host__ __device__ constexpr int t_in(int l) { return l == 0 ? 63 : (l == T_SKIP ? 319 : 256); }
__host__ __device__ constexpr int t_woff(int l) {        // float offset of W_l in the trunk block [W0 b0 W1 b1 ...]
    int o = 0;
    for (int i = 0; i < l; ++i) o += 256 * t_in(i) + 256;
    return o;
}
__host__ __device__ constexpr int t_boff(int l) { return t_woff(l) + 256 * t_in(l); }
__host__ __device__ constexpr int t_coff(int l) {        // chunk offset of layer l in the forward image
    int o = 0;
    for (int i = 0; i < l; ++i) o += t_nch(i);
    return o;
}
constexpr int T_FWD_CHUNKS = t_coff(8);                  // 120
constexpr int T_BWD_CHUNKS = 7 * 16;                     // layers 1..7, W^T restricted to the 256 hidden inputs
// register identity of the chain: accumulator register g of output tile t' (neuron 32 t' + rowc(g) + 4 hi) is k-slot
// j = g & 7 of chunk c = 2 t' + (g >> 3)
__host__ __device__ constexpr int kmap(int c, int hi, int j) { return (c >> 1) * 32 + 16 * (c & 1) + 8 * (j >> 2) + 4 * hi + (j & 3); }

template <int MODE> struct TC;
template <> struct TC<1> { static constexpr int NP = 1, NB = 2, NTS = 2; typedef __bf16 ST; };
template <> struct TC<6> { static constexpr int NP = 3, NB = 1, NTS = 1; typedef float ST; };

// ---- weight images -------------------------------------------------------------------------------------------------
// forward:  [layer][t 8][c nch][p NP][lane 64][8]   A fragment of W_l rows 32 t.., k-slots of chunk c
// backward: [layer 1..7][t 8][c 16][p NP][lane 64][8]   A fragment of W_l^T rows (= inputs) 32 t.., k = outputs kmap(c,.)
template <int NP>
__global__ __launch_bounds__(256) void trunk_prep_kernel(const float *__restrict__ P, __bf16 *__restrict__ fimg,
                                                         __bf16 *__restrict__ bimg) {
    const int id = blockIdx.x * 256 + threadIdx.x;
    const int lane = id & 63, sl = lane & 31, hi = lane >> 5;
    int tc = id >> 6;
    float v[8];
    __bf16 *dst;
    if (tc < T_FWD_CHUNKS * 8) {
        int l = 0;
        while (l < 7 && tc >= t_coff(l + 1) * 8) ++l;
        const int nch = t_nch(l), r = tc - t_coff(l) * 8, t = r / nch, c = r % nch, kin = t_in(l);
        const float *W = P + t_woff(l) + (t * 32 + sl) * kin;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int col;
            if (l == 0) col = 16 * c + 8 * hi + j;
            else if (c < 16) col = kmap(c, hi, j);
            else col = 256 + 16 * (c - 16) + 8 * hi + j;
            v[j] = col < kin ? W[col] : 0.f;
        }
        dst = fimg + ((size_t)tc * NP * 64 + lane) * 8;
    } else {
        tc -= T_FWD_CHUNKS * 8;
        if (tc >= T_BWD_CHUNKS * 8) return;
        const int l = 1 + tc / 128, r = tc % 128, t = r >> 4, c = r & 15, kin = t_in(l);
        const float *W = P + t_woff(l) + t * 32 + sl;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = W[kmap(c, hi, j) * kin];
        dst = bimg + ((size_t)tc * NP * 64 + lane) * 8;
    }
    bf16x8 o[3];
    split8<NP>(v, o);
#pragma unroll
    for (int p = 0; p < NP; ++p) *reinterpret_cast<bf16x8 *>(dst + p * 512) = o[p];
}

__device__ __forceinline__ void glds16(const void *g, void *l) {
#ifdef TRUNK_NO_GLDS                                       // timing experiment: no weight stream (results are garbage)
    return;
#endif
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)l, 16, 0, 0);
}

struct TrunkArgs {
    const float *enc; int ld_enc;                        // [n_pad][>= 64] position encoding (63 features + zero)
    const float *P;                                      // trunk parameter block (the biases)
    const __bf16 *img;                                   // forward / backward weight image
    void *acts;                                          // saved activations, 8 layers (forward: may be null)
    float *h7; int ld_h7;                                // forward: row-major copy of the last layer for the heads
    const float *dz7; int ld_dz7;                        // backward: row-major d loss / d (pre-activation of layer 7)
    void *dz;                                            // backward: pre-activation gradients, 8 layers (fragment layout)
    int64_t n;
};

// values of one accumulator tile -> the two k-chunks it is in the next layer, and the saved copy
template <int MODE>
__device__ __forceinline__ void pack_tile(const float (&y)[16], bf16x8 (&lo)[3], bf16x8 (&hi8)[3]) {
    constexpr int NP = TC<MODE>::NP;
    split8<NP>(y, lo);
    split8<NP>(y + 8, hi8);
}
template <int MODE>
__device__ __forceinline__ void store_tile(void *base, int64_t blk, int t, int lane, const float (&y)[16], const bf16x8 &lo,
                                           const bf16x8 &hi8) {
    if (MODE == 1) {
        __bf16 *p = reinterpret_cast<__bf16 *>(base) + (((blk * 16 + 2 * t) * 64) + lane) * 8;
        *reinterpret_cast<bf16x8 *>(p) = lo;
        *reinterpret_cast<bf16x8 *>(p + 512) = hi8;
    } else {
        float *p = reinterpret_cast<float *>(base) + (((blk * 8 + t) * 4) * 64 + lane) * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<float4 *>(p + q * 256) = make_float4(y[4 * q], y[4 * q + 1], y[4 * q + 2], y[4 * q + 3]);
    }
}

// ---- forward ---------------------------------------------------------------------------------------------------------
// The layer body exists in three compile-time shapes (first layer: encoding chunks only; hidden; skip layer: hidden +
// encoding chunks) and with / without the saved copy: a run-time `if` around a group of MFMAs makes the compiler copy
// whole accumulator tuples between AGPRs at the join (a first version spent 30 % of its VALU issue on such copies).
template <bool B> struct BoolC { static constexpr bool value = B; };
// scheduling fence between the MFMA stages and between the epilogue tiles: without it the scheduler interleaves them
// across the whole unrolled layer and spills ~150 VGPRs
#define TRUNK_FENCE() __builtin_amdgcn_sched_barrier(0)

template <int MODE, bool SAVE>
__global__ __launch_bounds__(256, 1) void trunk_fwd_kernel(TrunkArgs a) {
    using C = TC<MODE>;
    using PR = Pairs<MODE>;
    constexpr int NP = C::NP, NB = C::NB, NTS = C::NTS, NTG = 8 / NTS;
    constexpr int STAGE = NTS * 20 * NP * 1024;          // bytes of the largest stage (layer 5)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    float *bias = reinterpret_cast<float *>(smem_all);   // [8][256], FIRST: its reads fold into 16-bit ds offsets of one base
    unsigned char *smem_tf = smem_all + 8192;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hi = lane >> 5, sl = lane & 31;
    const uint32_t lane16 = lane * 16;
    for (int i = threadIdx.x; i < 2048; i += 256) bias[i] = a.P[t_boff(i >> 8) + (i & 255)];
    const int64_t n_blk = (a.n + 31) >> 5, n_grp = (n_blk + 4 * NB - 1) / (4 * NB);
    const size_t lstride = (size_t)n_grp * 4 * NB * 32 * 256;                // elements per saved layer (whole groups)

    auto issue = [&](int l, int tg, int buf) {           // stage (layer l, tile group tg) -> LDS buffer buf
        const int nch = t_nch(l), pieces = NTS * nch * NP;
        // uniform (scalar) piece offset + one 32-bit lane offset; the empty asm keeps the compiler from precomputing a
        // 64-bit vector address per piece and stage in the kernel prologue (it spilled ~180 registers doing so)
        uint32_t off = (uint32_t)(t_coff(l) * 8 + tg * NTS * nch) * NP * 1024 + wave * 1024;
        unsigned char *dst = smem_tf + buf * STAGE + wave * 1024;
        for (int i = wave; i < pieces; i += 4) {
            asm volatile("" : "+s"(off));
            glds16(reinterpret_cast<const unsigned char *>(a.img) + off + lane16, dst);
            off += 4096; dst += 4096;
        }
    };
    int buf = 0;
    if ((int64_t)blockIdx.x < n_grp) issue(0, 0, 0);
    for (int64_t grp = blockIdx.x; grp < n_grp; grp += gridDim.x) {
        const int64_t blk0 = (grp * 4 + wave) * NB;
        const bool more_grp = grp + gridDim.x < n_grp;
        bf16x8 x[NB][NP][16];
        // encoding operands (layers 0 and 5); rows of blocks past the end are clamped (their results are never stored
        // row-major, and the fragment-layout buffers are whole groups)
        auto load_enc = [&](bf16x8 (&e)[NB][NP][4]) {
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int64_t blk = blk0 + u < n_blk ? blk0 + u : n_blk - 1;
                const float *xp = a.enc + (blk * 32 + sl) * a.ld_enc + 8 * hi;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float4 v0 = *reinterpret_cast<const float4 *>(xp + 16 * c), v1 = *reinterpret_cast<const float4 *>(xp + 16 * c + 4);
                    const float xs[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                    bf16x8 o[3];
                    split8<NP>(xs, o);
#pragma unroll
                    for (int p = 0; p < NP; ++p) e[u][p][c] = o[p];
                }
            }
        };
        auto layer = [&](auto has_h_c, auto has_e_c, auto last_c, const int l) {
            constexpr bool HAS_H = decltype(has_h_c)::value, HAS_E = decltype(has_e_c)::value, LAST = decltype(last_c)::value;
            constexpr int nch = (HAS_H ? 16 : 0) + (HAS_E ? 4 : 0), ce = HAS_H ? 16 : 0;
            bf16x8 e[NB][NP][4];
            if (HAS_E) load_enc(e);
            f32x16 acc[NB][8];
#pragma unroll
            for (int tg = 0; tg < NTG; ++tg) {
                __syncthreads();                         // stage (l, tg) has landed (the fence drains this wave's LDS-DMA)
                if (tg + 1 < NTG) issue(l, tg + 1, buf ^ 1);
                else if (!LAST) issue(l + 1, 0, buf ^ 1);
                else if (more_grp) issue(0, 0, buf ^ 1);
                const unsigned char *st = smem_tf + buf * STAGE + lane * 16;
#pragma unroll
                for (int tt = 0; tt < NTS; ++tt) {
                    const int t = tg * NTS + tt;
                    f32x16 part;                         // odd product terms when a stage has a single accumulator (fp32 mode)
#pragma unroll
                    for (int g = 0; g < 16; ++g) {
                        part[g] = 0.f;
#pragma unroll
                        for (int u = 0; u < NB; ++u) acc[u][t][g] = 0.f;
                    }
                    const unsigned char *wt = st + tt * nch * NP * 1024;
                    int m = 0;
#pragma unroll
                    for (int c = 0; c < nch; ++c) {
                        bf16x8 w[NP];
#pragma unroll
                        for (int p = 0; p < NP; ++p) w[p] = *reinterpret_cast<const bf16x8 *>(wt + (c * NP + p) * 1024);
#pragma unroll
                        for (int k = 0; k < PR::N; ++k)
#pragma unroll
                            for (int u = 0; u < NB; ++u, ++m) {
                                const bf16x8 &xv = (HAS_H && c < 16) ? x[u][PR::A[k]][c & 15] : e[u][PR::A[k]][(c - ce) & 3];
                                if (NB * NTS == 1 && (m & 1)) part = MFMAB(w[PR::W[k]], xv, part);
                                else acc[u][t] = MFMAB(w[PR::W[k]], xv, acc[u][t]);
                            }
                    }
                    if (NB * NTS == 1) acc[0][t] += part;
                }
                buf ^= 1;
                TRUNK_FENCE();
            }
            // bias + activation; the accumulators become the next layer's operands and the saved copy
            typename C::ST *sv = reinterpret_cast<typename C::ST *>(a.acts) + (size_t)l * lstride;
#pragma unroll
            for (int u = 0; u < NB; ++u) {
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    float y[16];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 b4 = *reinterpret_cast<const float4 *>(bias + l * 256 + t * 32 + 8 * q + 4 * hi);
                        y[4 * q] = softplus100(acc[u][t][4 * q] + b4.x);
                        y[4 * q + 1] = softplus100(acc[u][t][4 * q + 1] + b4.y);
                        y[4 * q + 2] = softplus100(acc[u][t][4 * q + 2] + b4.z);
                        y[4 * q + 3] = softplus100(acc[u][t][4 * q + 3] + b4.w);
                    }
                    bf16x8 lo[3], hi8[3];
                    pack_tile<MODE>(y, lo, hi8);
#pragma unroll
                    for (int p = 0; p < NP; ++p) { x[u][p][2 * t] = lo[p]; x[u][p][2 * t + 1] = hi8[p]; }
                    if (SAVE) store_tile<MODE>(sv, blk0 + u, t, lane, y, lo[0], hi8[0]);
                    if (LAST && a.h7 && blk0 + u < n_blk) {
                        float *hp = a.h7 + ((blk0 + u) * 32 + sl) * a.ld_h7 + t * 32 + 4 * hi;
#pragma unroll
                        for (int q = 0; q < 4; ++q) *reinterpret_cast<float4 *>(hp + 8 * q) = make_float4(y[4 * q], y[4 * q + 1], y[4 * q + 2], y[4 * q + 3]);
                    }
                    TRUNK_FENCE();
                }
            }
        };
        layer(BoolC<false>(), BoolC<true>(), BoolC<false>(), 0);
        for (int l = 1; l < T_SKIP; ++l) layer(BoolC<true>(), BoolC<false>(), BoolC<false>(), l);
        layer(BoolC<true>(), BoolC<true>(), BoolC<false>(), T_SKIP);
        layer(BoolC<true>(), BoolC<false>(), BoolC<false>(), 6);
        layer(BoolC<true>(), BoolC<false>(), BoolC<true>(), 7);
    }
}

// ---- backward (data): dz_{l-1} = (W_l^T dz_l)[:256] * softplus'(h_{l-1}), l = 7 .. 1 ---------------------------------------
template <int MODE>
__global__ __launch_bounds__(256, 1) void trunk_bwd_kernel(TrunkArgs a) {
    using C = TC<MODE>;
    using PR = Pairs<MODE>;
    typedef typename C::ST ST;
    constexpr int NP = C::NP, NB = C::NB, NTS = C::NTS, NTG = 8 / NTS;
    constexpr int STAGE = NTS * 16 * NP * 1024;
    constexpr int HV = MODE == 1 ? 2 : 4;                // 16-byte loads per lane and tile of a saved layer
    constexpr int PD = 4;                                // saved-activation tiles in flight
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_tb[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hi = lane >> 5, sl = lane & 31;
    const uint32_t lane16 = lane * 16;
    const int64_t n_blk = (a.n + 31) >> 5, n_grp = (n_blk + 4 * NB - 1) / (4 * NB);
    const size_t lstride = (size_t)n_grp * 4 * NB * 32 * 256;                // whole groups: stores need no bounds

    auto issue = [&](int l, int tg, int buf) {           // stage (layer l in 1..7, input-tile group tg)
        const int pieces = NTS * 16 * NP;
        uint32_t off = (uint32_t)((l - 1) * 128 + tg * NTS * 16) * NP * 1024 + wave * 1024;       // see trunk_fwd_kernel
        unsigned char *dst = smem_tb + buf * STAGE + wave * 1024;
        for (int i = wave; i < pieces; i += 4) {
            asm volatile("" : "+s"(off));
            glds16(reinterpret_cast<const unsigned char *>(a.img) + off + lane16, dst);
            off += 4096; dst += 4096;
        }
    };
    int buf = 0;
    if ((int64_t)blockIdx.x < n_grp) issue(7, 0, 0);
    for (int64_t grp = blockIdx.x; grp < n_grp; grp += gridDim.x) {
        const int64_t blk0 = (grp * 4 + wave) * NB;
        const bool more_grp = grp + gridDim.x < n_grp;
        bf16x8 x[NB][NP][16];
        // dz7 (row-major, written by the heads' backward) -> chain operands + fragment-layout copy for trunk_dw
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int64_t blkc = blk0 + u < n_blk ? blk0 + u : n_blk - 1;            // past the end: any valid rows (never used)
            const float *zp = a.dz7 + (blkc * 32 + sl) * a.ld_dz7 + 4 * hi;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                float y[16];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = *reinterpret_cast<const float4 *>(zp + t * 32 + 8 * q);
                    y[4 * q] = v.x; y[4 * q + 1] = v.y; y[4 * q + 2] = v.z; y[4 * q + 3] = v.w;
                }
                bf16x8 lo[3], hi8[3];
                pack_tile<MODE>(y, lo, hi8);
#pragma unroll
                for (int p = 0; p < NP; ++p) { x[u][p][2 * t] = lo[p]; x[u][p][2 * t + 1] = hi8[p]; }
                store_tile<MODE>(reinterpret_cast<ST *>(a.dz) + 7 * lstride, blk0 + u, t, lane, y, lo[0], hi8[0]);
            }
        }
        for (int l = 7; l >= 1; --l) {
            const ST *hs = reinterpret_cast<const ST *>(a.acts) + (size_t)(l - 1) * lstride;     // h_{l-1}
            uint4 hpre[PD][HV];
            auto load_h = [&](int i, uint4 (&dst)[HV]) {                                           // i = u * 8 + t
                const int u = i >> 3, t = i & 7;
                const int64_t blk = blk0 + u;
                const uint4 *p = MODE == 1 ? reinterpret_cast<const uint4 *>(hs + ((blk * 16 + 2 * t) * 64 + lane) * 8)
                                           : reinterpret_cast<const uint4 *>(hs + ((blk * 8 + t) * 4 * 64 + lane) * 4);
#pragma unroll
                for (int q = 0; q < HV; ++q) dst[q] = p[q * 64];
            };
            f32x16 acc[NB][8];
#pragma unroll
            for (int tg = 0; tg < NTG; ++tg) {
                __syncthreads();
                if (tg + 1 < NTG) issue(l, tg + 1, buf ^ 1);
                else if (l > 1) issue(l - 1, 0, buf ^ 1);
                else if (more_grp) issue(7, 0, buf ^ 1);
                if (tg == NTG - 1) {
#pragma unroll
                    for (int i = 0; i < PD; ++i) load_h(i, hpre[i]);
                }
                const unsigned char *st = smem_tb + buf * STAGE;
#pragma unroll
                for (int tt = 0; tt < NTS; ++tt) {
                    const int t = tg * NTS + tt;
                    f32x16 part;
#pragma unroll
                    for (int g = 0; g < 16; ++g) {
                        part[g] = 0.f;
#pragma unroll
                        for (int u = 0; u < NB; ++u) acc[u][t][g] = 0.f;
                    }
                    const unsigned char *wt = st + (size_t)tt * 16 * NP * 1024 + lane * 16;
#pragma unroll
                    for (int c = 0; c < 16; ++c) {
                        bf16x8 w[NP];
#pragma unroll
                        for (int p = 0; p < NP; ++p) w[p] = *reinterpret_cast<const bf16x8 *>(wt + (c * NP + p) * 1024);
#pragma unroll
                        for (int k = 0; k < PR::N; ++k)
#pragma unroll
                            for (int u = 0; u < NB; ++u) {
                                if (NB * NTS == 1 && (c & 1)) part = MFMAB(w[PR::W[k]], x[u][PR::A[k]][c], part);
                                else acc[u][t] = MFMAB(w[PR::W[k]], x[u][PR::A[k]][c], acc[u][t]);
                            }
                    }
                    if (NB * NTS == 1) acc[0][t] += part;
                }
                buf ^= 1;
            }
            ST *sv = reinterpret_cast<ST *>(a.dz) + (size_t)(l - 1) * lstride;
#pragma unroll
            for (int i = 0; i < NB * 8; ++i) {
                const int u = i >> 3, t = i & 7;
                float h[16];
                if (MODE == 1) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const bf16x8 hv = *reinterpret_cast<const bf16x8 *>(&hpre[i % PD][q]);
#pragma unroll
                        for (int j = 0; j < 8; ++j) h[8 * q + j] = (float)hv[j];
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 hv = *reinterpret_cast<const float4 *>(&hpre[i % PD][q]);
                        h[4 * q] = hv.x; h[4 * q + 1] = hv.y; h[4 * q + 2] = hv.z; h[4 * q + 3] = hv.w;
                    }
                }
                if (i + PD < NB * 8) load_h(i + PD, hpre[i % PD]);
                float y[16];
#pragma unroll
                for (int g = 0; g < 16; ++g) y[g] = acc[u][t][g] * dsoftplus_from_out(h[g], 100.f);
                bf16x8 lo[3], hi8[3];
                pack_tile<MODE>(y, lo, hi8);
#pragma unroll
                for (int p = 0; p < NP; ++p) { x[u][p][2 * t] = lo[p]; x[u][p][2 * t + 1] = hi8[p]; }
                store_tile<MODE>(sv, blk0 + u, t, lane, y, lo[0], hi8[0]);
            }
        }
    }
}

// ---- weight / bias gradient of one trunk layer from the fragment-layout copies ---------------------------------------------
// as dense_dw_x_kernel (ren_dense.hip): one workgroup of 8 waves per sample split, per 32-sample stage the block's dz (256
// features) and inputs (<= 256 per K group) go to LDS transposed, [feature][sample] bf16 pieces, wave w owns the 32 output
// rows 32 w.. against up to 8 input tiles.
constexpr int TDW_ST = 40;
struct TrunkDwArgs {
    const void *dz, *x;                                  // fragment layout (this layer's dz; the previous layer's activations)
    const float *enc; int ld_enc;                        // row-major encoding: layer 0's input, layer 5's second K group
    int layer;
    int64_t n;
    float *slab_w, *slab_b;                              // [n_splits][256][K], [n_splits][256]
};

template <int MODE>
__global__ __launch_bounds__(512, 1) void trunk_dw_kernel(TrunkDwArgs a) {
    typedef typename TC<MODE>::ST ST;
    constexpr int NP = TC<MODE>::NP;
    constexpr int NV = MODE == 1 ? 2 : 4;                // 16-byte pieces per thread and operand per stage
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_td[];
    __bf16 *ZT = reinterpret_cast<__bf16 *>(smem_td), *XT = ZT + NP * 256 * TDW_ST;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hi = lane >> 5, sl = lane & 31;
    const int n_splits = gridDim.x;
    const int64_t n_blk = (a.n + 31) >> 5;
    const int K = t_in(a.layer);
    const int k_groups = a.layer == T_SKIP ? 2 : 1;
    float *sw = a.slab_w + (int64_t)blockIdx.x * 256 * K, *sb = a.slab_b + (int64_t)blockIdx.x * 256;
    for (int kg = 0; kg < k_groups; ++kg) {
        const bool x_enc = a.layer == 0 || kg == 1;      // inputs of this K group: the row-major encoding (64 columns)
        const int kt = x_enc ? 2 : 8;
        f32x16 acc[8];
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[t][g] = 0.f;
        float bsum[NV * (MODE == 1 ? 8 : 4)];
#pragma unroll
        for (int q = 0; q < NV * (MODE == 1 ? 8 : 4); ++q) bsum[q] = 0.f;
        uint4 pz[NV], px[NV];
        float4 pe;
        auto fetch = [&](int64_t blk) {
            const uint4 *zb = reinterpret_cast<const uint4 *>(reinterpret_cast<const ST *>(a.dz) + blk * 32 * 256);
            const uint4 *xb = reinterpret_cast<const uint4 *>(reinterpret_cast<const ST *>(a.x) + blk * 32 * 256);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                pz[v] = zb[(wave * NV + v) * 64 + lane];
                if (!x_enc) px[v] = xb[(wave * NV + v) * 64 + lane];
            }
            if (x_enc) pe = *reinterpret_cast<const float4 *>(a.enc + (blk * 32 + (threadIdx.x & 31)) * a.ld_enc + 4 * (threadIdx.x >> 5));
        };
        auto put = [&](__bf16 *T, int f, float v) {
            __bf16 s[3];
            split<NP>(v, s);
#pragma unroll
            for (int p = 0; p < NP; ++p) T[(p * 256 + f) * TDW_ST + sl] = s[p];
        };
        auto stash = [&]() {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int piece = wave * NV + v;
                if (MODE == 1) {                         // piece = chunk c: features kmap(c, hi, j)
                    const bf16x8 z8 = *reinterpret_cast<const bf16x8 *>(&pz[v]), x8 = *reinterpret_cast<const bf16x8 *>(&px[v]);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int f = kmap(piece, hi, j);
                        ZT[f * TDW_ST + sl] = z8[j];
                        if (!x_enc) XT[f * TDW_ST + sl] = x8[j];
                        bsum[8 * v + j] += (float)z8[j];
                    }
                } else {                                 // piece = (t, q): features 32 t + 8 q + 4 hi + j
                    const float4 z4 = *reinterpret_cast<const float4 *>(&pz[v]), x4 = *reinterpret_cast<const float4 *>(&px[v]);
                    const float z[4] = {z4.x, z4.y, z4.z, z4.w}, xx[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int f = 8 * piece + 4 * hi + j;
                        put(ZT, f, z[j]);
                        if (!x_enc) put(XT, f, xx[j]);
                        bsum[4 * v + j] += z[j];
                    }
                }
            }
            if (x_enc) {                                 // thread -> (sample tid & 31, features 4 (tid >> 5) ..)
                const float ev[4] = {pe.x, pe.y, pe.z, pe.w};
                const int s = threadIdx.x & 31, f0 = 4 * (threadIdx.x >> 5);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    __bf16 sp[3];
                    split<NP>(ev[j], sp);
#pragma unroll
                    for (int p = 0; p < NP; ++p) XT[(p * 256 + f0 + j) * TDW_ST + s] = sp[p];
                }
            }
        };
        int64_t blk = blockIdx.x;
        if (blk < n_blk) fetch(blk);
        for (; blk < n_blk; blk += n_splits) {
            __syncthreads();
            stash();
            __syncthreads();
            const int64_t nxt = blk + n_splits;
            if (nxt < n_blk) fetch(nxt);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 az[NP];
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    az[p] = *reinterpret_cast<const bf16x8 *>(ZT + (p * 256 + wave * 32 + sl) * TDW_ST + 16 * ks + 8 * hi);
#pragma unroll
                for (int t = 0; t < 8; ++t)
                    if (t < kt) {
                        bf16x8 bx[NP];
#pragma unroll
                        for (int p = 0; p < NP; ++p)
                            bx[p] = *reinterpret_cast<const bf16x8 *>(XT + (p * 256 + t * 32 + sl) * TDW_ST + 16 * ks + 8 * hi);
                        if (NP == 3) {
                            acc[t] = MFMAB(az[2], bx[0], acc[t]);
                            acc[t] = MFMAB(az[0], bx[2], acc[t]);
                            acc[t] = MFMAB(az[1], bx[1], acc[t]);
                            acc[t] = MFMAB(az[1], bx[0], acc[t]);
                            acc[t] = MFMAB(az[0], bx[1], acc[t]);
                        }
                        acc[t] = MFMAB(az[0], bx[0], acc[t]);
                    }
            }
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int k = kg * 256 + t * 32 + sl;
            if (t >= kt || k >= K) continue;
#pragma unroll
            for (int g = 0; g < 16; ++g) sw[(int64_t)(wave * 32 + rowc(g) + 4 * hi) * K + k] = acc[t][g];
        }
        if (kg == 0) {                                   // bias gradient: sum over the 32 sample lanes
#pragma unroll
            for (int v = 0; v < NV; ++v)
#pragma unroll
                for (int j = 0; j < (MODE == 1 ? 8 : 4); ++j) {
                    float s = bsum[(MODE == 1 ? 8 : 4) * v + j];
#pragma unroll
                    for (int off = 1; off < 32; off <<= 1) s += __shfl_xor(s, off, 64);
                    const int piece = wave * NV + v;
                    const int f = MODE == 1 ? kmap(piece, hi, j) : 8 * piece + 4 * hi + j;
                    if (sl == 0) sb[f] = s;
                }
        }
        __syncthreads();
    }
}

template <int MODE> size_t fwd_lds() { return 2 * (size_t)TC<MODE>::NTS * 20 * TC<MODE>::NP * 1024 + 8 * 256 * sizeof(float); }
template <int MODE> size_t bwd_lds() { return 2 * (size_t)TC<MODE>::NTS * 16 * TC<MODE>::NP * 1024; }

}  // namespace

static inline int trunk_np(int mode) { return mode == 1 ? 1 : 3; }

extern "C" int64_t ren_trunk_image_bytes(int32_t mode) {
    if (mode != 1 && mode != 6) return -1;
    return (int64_t)(T_FWD_CHUNKS + T_BWD_CHUNKS) * 8 * trunk_np(mode) * 1024;
}

extern "C" int64_t ren_trunk_saved_bytes(int32_t mode, int64_t n) {
    if ((mode != 1 && mode != 6) || n < 0) return -1;
    const int64_t per_grp = 4 * (mode == 1 ? TC<1>::NB : TC<6>::NB), n_grp = ((n + 31) / 32 + per_grp - 1) / per_grp;
    return 8 * n_grp * per_grp * 32 * 256 * (mode == 1 ? 2 : 4);                // whole workgroup passes of 32-sample blocks
}

extern "C" int ren_trunk_prep(const float *trunk_params, int32_t mode, void *image, void *stream) {
    if (!trunk_params || !image || (mode != 1 && mode != 6)) return REN_ERR_BAD_ARG;
    const int np = trunk_np(mode);
    __bf16 *f = reinterpret_cast<__bf16 *>(image), *b = f + (size_t)T_FWD_CHUNKS * 8 * np * 512;
    const int threads = (T_FWD_CHUNKS + T_BWD_CHUNKS) * 8 * 64;
    if (mode == 1) hipLaunchKernelGGL(trunk_prep_kernel<1>, dim3(threads / 256), dim3(256), 0, (hipStream_t)stream, trunk_params, f, b);
    else hipLaunchKernelGGL(trunk_prep_kernel<3>, dim3(threads / 256), dim3(256), 0, (hipStream_t)stream, trunk_params, f, b);
    REN_CHECK_LAUNCH();
}

static int trunk_grid(int64_t n, int nb) {
    const int64_t n_grp = ((n + 31) / 32 + 4 * nb - 1) / (4 * nb);
    return (int)(n_grp < 256 ? n_grp : 256);
}

extern "C" int ren_trunk_fwd(const float *enc, int32_t ld_enc, const float *trunk_params, const void *image, int32_t mode,
                             int64_t n, void *saved, float *h7, int32_t ld_h7, void *stream) {
    if (!enc || !trunk_params || !image || (mode != 1 && mode != 6) || n < 0 || ld_enc < 64 || (ld_enc & 3) || (h7 && (ld_h7 < 256 || (ld_h7 & 3))))
        return REN_ERR_BAD_ARG;
    if (n == 0) return REN_OK;
    TrunkArgs a = {};
    a.enc = enc; a.ld_enc = ld_enc; a.P = trunk_params; a.img = reinterpret_cast<const __bf16 *>(image);
    a.acts = saved; a.h7 = h7; a.ld_h7 = ld_h7; a.n = n;
    hipStream_t st = (hipStream_t)stream;
#define REN_TRUNK_FWD(MODE, SAVE)                                                                                              \
    do {                                                                                                                        \
        (void)hipFuncSetAttribute((const void *)trunk_fwd_kernel<MODE, SAVE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fwd_lds<MODE>()); \
        hipLaunchKernelGGL((trunk_fwd_kernel<MODE, SAVE>), dim3(trunk_grid(n, TC<MODE>::NB)), dim3(256), fwd_lds<MODE>(), st, a); \
    } while (0)
    if (mode == 1) { if (saved) REN_TRUNK_FWD(1, true); else REN_TRUNK_FWD(1, false); }
    else { if (saved) REN_TRUNK_FWD(6, true); else REN_TRUNK_FWD(6, false); }
    REN_CHECK_LAUNCH();
}

extern "C" int ren_trunk_bwd(const float *dz7, int32_t ld_dz7, const void *image, int32_t mode, int64_t n, const void *saved,
                             void *dz, void *stream) {
    if (!dz7 || !image || !saved || !dz || (mode != 1 && mode != 6) || n < 0 || ld_dz7 < 256 || (ld_dz7 & 3)) return REN_ERR_BAD_ARG;
    if (n == 0) return REN_OK;
    TrunkArgs a = {};
    a.img = reinterpret_cast<const __bf16 *>(image) + (size_t)T_FWD_CHUNKS * 8 * trunk_np(mode) * 512;
    a.acts = const_cast<void *>(saved); a.dz7 = dz7; a.ld_dz7 = ld_dz7; a.dz = dz; a.n = n;
    hipStream_t st = (hipStream_t)stream;
    if (mode == 1) {
        (void)hipFuncSetAttribute((const void *)trunk_bwd_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bwd_lds<1>());
        hipLaunchKernelGGL(trunk_bwd_kernel<1>, dim3(trunk_grid(n, TC<1>::NB)), dim3(256), bwd_lds<1>(), st, a);
    } else {
        (void)hipFuncSetAttribute((const void *)trunk_bwd_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bwd_lds<6>());
        hipLaunchKernelGGL(trunk_bwd_kernel<6>, dim3(trunk_grid(n, TC<6>::NB)), dim3(256), bwd_lds<6>(), st, a);
    }
    REN_CHECK_LAUNCH();
}

extern "C" int64_t ren_trunk_bwd_weight_workspace_floats(int32_t n_splits) {
    if (n_splits < 1) return -1;
    return (int64_t)n_splits * (256 * 319 + 256);
}

extern "C" int ren_trunk_bwd_weight(const void *dz, const void *saved, const float *enc, int32_t ld_enc, int32_t mode, int64_t n,
                                    int32_t n_splits, float *trunk_grads, float *workspace, void *stream) {
    if (!dz || !saved || !enc || !trunk_grads || !workspace || (mode != 1 && mode != 6) || n < 0 || n_splits < 1 || ld_enc < 64 || (ld_enc & 3))
        return REN_ERR_BAD_ARG;
    if (n == 0) return REN_OK;
    hipStream_t st = (hipStream_t)stream;
    const size_t lbytes = (size_t)ren_trunk_saved_bytes(mode, n) / 8;
    for (int l = 7; l >= 0; --l) {
        TrunkDwArgs a;
        a.dz = reinterpret_cast<const unsigned char *>(dz) + l * lbytes;
        a.x = l > 0 ? reinterpret_cast<const unsigned char *>(saved) + (l - 1) * lbytes : nullptr;
        a.enc = enc; a.ld_enc = ld_enc; a.layer = l; a.n = n;
        const int K = t_in(l);
        a.slab_w = workspace; a.slab_b = workspace + (int64_t)n_splits * 256 * K;
        const size_t lds = 2 * (size_t)trunk_np(mode) * 256 * TDW_ST * 2;
        if (mode == 1) {
            (void)hipFuncSetAttribute((const void *)trunk_dw_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(trunk_dw_kernel<1>, dim3(n_splits), dim3(512), lds, st, a);
        } else {
            (void)hipFuncSetAttribute((const void *)trunk_dw_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(trunk_dw_kernel<6>, dim3(n_splits), dim3(512), lds, st, a);
        }
        launch_reduce_slabs(a.slab_w, n_splits, 256 * K, trunk_grads + t_woff(l), st);
        launch_reduce_slabs(a.slab_b, n_splits, 256, trunk_grads + t_boff(l), st);
    }
    REN_CHECK_LAUNCH();
}
