// `arch: mlp`: the whole vanilla-NeRF field (robust_e_nerf/external/mlp.py:26-205: MLP / NerfMLP -- eight Linear +
// Softplus(beta = 100) hidden layers with the input encoding concatenated again after layer 4, the sigma layer, the
// bottleneck layer, the 283 -> 128 -> C colour head) as ONE launch forward and ONE launch backward (data), plus the weight
// gradients, instead of 12 + 11 + 12 dense-layer launches with a 2 KB-per-sample HBM round trip between each of them
// (ren_dense.hip, which stays for the exact-f32 mode and the forward-mode tangent stream):
//
//   vfield_fwd  a wave keeps its 32-sample blocks in registers through all layers: the outputs of a layer are its
//               accumulator tiles (lane = sample), and the accumulator layout IS the next layer's B-operand layout (a
//               k-slot permutation the weight image is pre-permuted for, as in ren_mlp_x.hip), so activations never leave
//               the register file except as the copy saved for the backward pass;
//   vfield_bwd  the same chain backwards over W^T, from (d loss / d rgb pre-activation, d loss / d sigma pre-activation)
//               to the first layer: dz_{l-1} = (W_l^T dz_l) * act'(h_{l-1});
//   vfield_dw   dW_l = dz_l^T x_{l-1}, db_l = sum dz_l from the saved copies (samples are the reduction: the tiles are
//               transposed on the matrix cores), slab-reduced: deterministic, no atomics.
//
// Kernels in this file: vfield_fwd1 / vfield_bwd<1> (bf16 mode; vfield_fwd<1> is the forward without the software-pipelined
// epilogue, kept behind REN_KNOB_VFIELD_PLAIN as its bit-exact reference), vfield_fwd6 / vfield_bwd6 (fp32 mode,
// reduction-outer), vfield_dw<1 | 6> (both modes), vfield_prep.
//
// Weights: `vfield_prep` turns the fp32 parameter block into bf16 MFMA A-fragment images (1 KB = one 32 x 16 fragment,
// lane-linear), once per field evaluation (9 us); the kernels stream them through a double-buffered LDS stage with
// `global_load_lds` (1 KB per wave instruction, no staging registers) while the matrix cores work on the previous stage.
// All four waves of the workgroup (one per SIMD: the accumulators and operands of a wave fill the 512-register file)
// share a stage.  One weight fragment read from LDS feeds two MFMAs (bf16 mode: two sample blocks per wave; fp32 mode:
// the six product terms of the three-piece split use three fragments), so LDS bandwidth stays at half of its peak.
// bf16 mode: a stage is two output tiles of all k-chunks ([tile][chunk] image); fp32 mode: two k-chunks of all output
// tiles ([chunk][tile] image), see the reduction-outer kernels below.
//
// Saved activations / pre-activation gradients ("fragment layout": per slot and 32-sample block, lane-linear 1 KB pieces =
// exactly the registers of a wave; slots 0-7 = hidden layers, 8 = bottleneck output, 9 = colour-head hidden layer (128)):
//   bf16 mode (MODE 1)  [blk][c 16][lane 64][8 bf16]   feature kmap(c, lane >> 5, j), sample blk * 32 + (lane & 31)
//   fp32 mode (MODE 6)  [blk][t 8][q 4][lane 64][4 f32] feature 32 t + 8 q + 4 (lane >> 5) + j
// bf16 mode stores what the next layer's matrix product sees anyway (bf16-rounded activations: the weight gradient is
// unchanged by the rounding; the activation derivative is taken from the rounded value).
#include "ren_mlp_common.h"

namespace {

// ---- the network ---------------------------------------------------------------------------------------------------------
// forward layers: 0-7 hidden, 8 sigma, 9 bottleneck, 10 colour hidden, 11 colour output
constexpr int NL = 12, L_SKIP = 5, L_SIGMA = 8, L_BOTT = 9, L_RGBH = 10, L_RGBO = 11;
__host__ __device__ constexpr int l_in(int l) {          // torch in_features
    return l == 0 ? 63 : l == L_SKIP ? 319 : l == L_RGBH ? 283 : l == L_RGBO ? 128 : 256;
}
__host__ __device__ constexpr int l_out(int l, int C) { return l == L_SIGMA ? 1 : l == L_RGBH ? 128 : l == L_RGBO ? C : 256; }
__host__ __device__ constexpr int l_nh(int l) { return l == 0 ? 0 : l == L_RGBO ? 8 : 16; }       // k-chunks from the chain
__host__ __device__ constexpr int l_ne(int l) { return l == 0 || l == L_SKIP ? 4 : l == L_RGBH ? 2 : 0; }   // encoding chunks
__host__ __device__ constexpr int l_nch(int l) { return l_nh(l) + l_ne(l); }
__host__ __device__ constexpr int l_nt(int l) { return l == L_SIGMA || l == L_RGBO ? 1 : l == L_RGBH ? 4 : 8; }   // output tiles
// parameter block (floats), reference state-dict order: hidden 0-7, sigma, bottleneck, colour hidden, colour output
__host__ __device__ constexpr int l_woff(int l, int C) {
    int o = 0;
    for (int i = 0; i < l; ++i) o += l_out(i, C) * l_in(i) + l_out(i, C);
    return o;
}
__host__ __device__ constexpr int l_boff(int l, int C) { return l_woff(l, C) + l_out(l, C) * l_in(l); }
// forward image: fragment (= 1 KB x NP) offset of layer l: [tile][chunk] fragments per layer
__host__ __device__ constexpr int f_off(int l) {
    int o = 0;
    for (int i = 0; i < l; ++i) o += l_nt(i) * l_nch(i);
    return o;
}
constexpr int F_FRAGS = f_off(NL);
// backward (W^T) image, steps 11, 10, 9, 7 .. 1 (sigma is the extra chunk of step 9):
//   11: 4 input tiles x 1 chunk (k = the C outputs, natural order)      10: 8 tiles (the 256 bottleneck inputs) x 8 chunks
//    9: 8 tiles x (16 chunks + 1 chunk whose k-slot 0 is the sigma layer)   7..1: 8 tiles x 16 chunks
__host__ __device__ constexpr int b_nt(int l) { return l == L_RGBO ? 4 : 8; }
__host__ __device__ constexpr int b_nch(int l) { return l == L_RGBO ? 1 : l == L_RGBH ? 8 : l == L_BOTT ? 17 : 16; }
__host__ __device__ constexpr int b_off(int l) {         // order in the image: 1..7, 9, 10, 11
    int o = 0;
    for (int i = 1; i < l; ++i)
        if (i != L_SIGMA) o += b_nt(i) * b_nch(i);
    return o;
}
constexpr int B_FRAGS = b_off(NL);
// register identity of the chain: accumulator register g of output tile t' (neuron 32 t' + rowc(g) + 4 hi) is k-slot
// j = g & 7 of chunk c = 2 t' + (g >> 3)
__host__ __device__ constexpr int kmap(int c, int hi, int j) { return (c >> 1) * 32 + 16 * (c & 1) + 8 * (j >> 2) + 4 * hi + (j & 3); }
constexpr int N_SLOTS = 10;                              // saved / dz slots (see the header)

template <int MODE> struct TC;
template <> struct TC<1> { static constexpr int NP = 1, NB = 2, NTS = 2; typedef __bf16 ST; };
template <> struct TC<6> { static constexpr int NP = 3, NB = 1, NTS = 1; typedef float ST; };
template <> struct TC<3> { static constexpr int NP = 2, NB = 1, NTS = 1; typedef float ST; };   // two pieces, three products: the fp32 family's layouts

// ---- weight images -------------------------------------------------------------------------------------------------
template <int NP>
__global__ __launch_bounds__(256) void vfield_prep_kernel(const float *__restrict__ P, int C, __bf16 *__restrict__ fimg,
                                                          __bf16 *__restrict__ bimg) {
    const int id = blockIdx.x * 256 + threadIdx.x;
    const int lane = id & 63, sl = lane & 31, hi = lane >> 5;
    int fr = id >> 6;
    float v[8];
    __bf16 *dst;
    if (fr < F_FRAGS) {
        int l = 0;
        while (l < NL - 1 && fr >= f_off(l + 1)) ++l;
        const int nch = l_nch(l), r = fr - f_off(l), kin = l_in(l), nh = l_nh(l);
        // bf16 mode: [tile][chunk] (a stage = tiles); fp32 mode: [chunk][tile] (a stage = two chunks of all tiles)
        const int t = NP > 1 ? r % l_nt(l) : r / nch, c = NP > 1 ? r / l_nt(l) : r % nch;
        const int row = t * 32 + sl;
        const float *W = P + l_woff(l, C) + row * kin;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int col = c < nh ? kmap(c, hi, j) : 16 * nh + 16 * (c - nh) + 8 * hi + j;
            v[j] = (row < l_out(l, C) && col < kin) ? W[col] : 0.f;
        }
        dst = fimg + ((size_t)fr * NP * 64 + lane) * 8;
    } else {
        fr -= F_FRAGS;
        if (fr >= B_FRAGS) return;
        int l = 1;
        for (int i = 2; i < NL; ++i)
            if (i != L_SIGMA && fr >= b_off(i)) l = i;
        const int nch = b_nch(l), r = fr - b_off(l), kin = l_in(l);
        const int t = NP > 1 ? r % b_nt(l) : r / nch, c = NP > 1 ? r / b_nt(l) : r % nch;
        const int i = t * 32 + sl;                       // input index = row of W^T
        const float *W = P + l_woff(l, C) + i;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float w = 0.f;
            if (l == L_RGBO) {                           // k = output o, natural order
                const int o = 8 * hi + j;
                if (o < C) w = W[o * kin];
            } else if (l == L_BOTT && c == 16) {         // sigma layer: k-slot 0
                if (hi == 0 && j == 0) w = P[l_woff(L_SIGMA, C) + i];
            } else {
                w = W[kmap(c, hi, j) * kin];
            }
            v[j] = w;
        }
        dst = bimg + ((size_t)fr * NP * 64 + lane) * 8;
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

struct FieldArgs {
    const float *enc; int ld_enc;                        // [n_pad][>= 64] position encoding (63 features + zero)
    const float *view; int ld_view;                      // [n_pad][>= 32] direction encoding (27 features + zero)
    const uint8_t *sel;                                  // [n_pad] inside-the-box selector of the sigma activation
    const float *P; int C;                               // parameter block (biases)
    const __bf16 *img;                                   // forward / backward weight image
    void *acts;                                          // saved activations (N_SLOTS slots)
    float *sigma, *rgb4;                                 // forward outputs: [n_pad], [n_pad][4]
    const float *dz_rgb, *dz_sig;                        // backward inputs: [n_pad][32] (columns < C / column 0)
    void *dz;                                            // backward: pre-activation gradients (N_SLOTS slots)
    int64_t acts_sstride;                                // backward: elements per slot of `acts` (0: as for n samples) -- the
                                                         // backward of a sample RANGE of a larger forward pass
    int64_t n;
    // tangent stream (vfield_fwd_jvp_kernel / vfield_bwd_jvp_kernel): d/dt of the encodings, saved tangent activations, the
    // tangent pre-activations of the two heads ([n_pad][4]), and the tangent-side gradients
    const float *encd, *viewd;
    void *actsd;
    float *zsd4, *zod4;
    const float *dzd_rgb, *dzd_sig;
    void *dzd;
    void *cpl;                                           // fp32 mode: the tangent side's coupling term into dz (vfield_bwd6_kernel<1 | 2>)
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

// fp32 mode: the third (smallest) bf16 piece of the chain operands lives in AGPRs by construction and is copied to VGPRs
// next to the one MFMA per chunk that uses it.  Three pieces x 16 chunks x 4 registers = 192 operand registers do not fit
// beside the weight fragments and epilogue temporaries in the 256 architectural VGPRs; left to itself the compiler spilled
// them to scratch and reloaded them for every output tile.
__device__ __forceinline__ void agpr_put(uint32_t (&dst)[4], const bf16x8 &v) {
    const uint4 u = *reinterpret_cast<const uint4 *>(&v);
    asm("v_accvgpr_write_b32 %0, %1" : "=a"(dst[0]) : "v"(u.x));
    asm("v_accvgpr_write_b32 %0, %1" : "=a"(dst[1]) : "v"(u.y));
    asm("v_accvgpr_write_b32 %0, %1" : "=a"(dst[2]) : "v"(u.z));
    asm("v_accvgpr_write_b32 %0, %1" : "=a"(dst[3]) : "v"(u.w));
}
__device__ __forceinline__ bf16x8 agpr_get(const uint32_t (&src)[4]) {
    uint4 u;
    asm("v_accvgpr_read_b32 %0, %1" : "=v"(u.x) : "a"(src[0]));
    asm("v_accvgpr_read_b32 %0, %1" : "=v"(u.y) : "a"(src[1]));
    asm("v_accvgpr_read_b32 %0, %1" : "=v"(u.z) : "a"(src[2]));
    asm("v_accvgpr_read_b32 %0, %1" : "=v"(u.w) : "a"(src[3]));
    return *reinterpret_cast<const bf16x8 *>(&u);
}

// A layer shape is a compile-time type: a run-time `if` around a group of MFMAs makes the compiler copy whole accumulator
// tuples between AGPRs at the join (a first version spent 30 % of its VALU issue on such copies).
constexpr int A_SP100 = 0, A_NONE = 1, A_SIGMA = 2, A_RGB = 3;
template <int I> struct IntC { static constexpr int value = I; };
template <int NH_, int NE_, int NT_, int ACT_> struct Shape { static constexpr int NH = NH_, NE = NE_, NT = NT_, ACT = ACT_; };
// scheduling fence between the MFMA stages and between the epilogue tiles
#define TRUNK_FENCE() __builtin_amdgcn_sched_barrier(0)
// LDS: bias table first (its reads fold into 16-bit ds offsets of one base register), then the two stage buffers
constexpr int LB_HID = 0, LB_BOTT = 2048, LB_RGBH = 2304, LB_SIGMA = 2432, LB_RGBO = 2436, LB_FLOATS = 2560;
__host__ __device__ constexpr int lb_off(int l) {
    return l < 8 ? LB_HID + 256 * l : l == L_SIGMA ? LB_SIGMA : l == L_BOTT ? LB_BOTT : l == L_RGBH ? LB_RGBH : LB_RGBO;
}

// ---- forward (tile-outer; instantiated for bf16 mode only: the plain reference of vfield_fwd1_kernel) ---------------------
// FULL: all twelve layers (rgb and sigma); otherwise the hidden layers and sigma only (the sampler's density pre-pass,
// occupancy-grid queries).  SAVE: keep the activations for the backward pass.
template <int MODE, bool SAVE, bool FULL>
__global__ __launch_bounds__(256, 1) void vfield_fwd_kernel(FieldArgs a) {
    using C = TC<MODE>;
    using PR = Pairs<MODE>;
    typedef typename C::ST ST;
    constexpr int NP = C::NP, NB = C::NB, NTS = C::NTS, XP = NP == 3 ? 2 : NP;
    constexpr int STAGE = NTS * 20 * NP * 1024;          // bytes of the largest stage (layer 5)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    float *bias = reinterpret_cast<float *>(smem_all);
    unsigned char *smem_tf = smem_all + LB_FLOATS * 4;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hi = lane >> 5, sl = lane & 31;
    const uint32_t lane16_k = lane * 16;
    for (int i = threadIdx.x; i < LB_FLOATS; i += 256) {
        float b = 0.f;
        if (i < 2048) b = a.P[l_boff(i >> 8, a.C) + (i & 255)];
        else if (i < LB_RGBH) b = a.P[l_boff(L_BOTT, a.C) + i - LB_BOTT];
        else if (i < LB_SIGMA) b = a.P[l_boff(L_RGBH, a.C) + i - LB_RGBH];
        else if (i == LB_SIGMA) b = a.P[l_boff(L_SIGMA, a.C)];
        else if (i >= LB_RGBO && i < LB_RGBO + a.C) b = a.P[l_boff(L_RGBO, a.C) + i - LB_RGBO];
        bias[i] = b;
    }
    const int64_t n_blk = (a.n + 31) >> 5, n_grp = (n_blk + 4 * NB - 1) / (4 * NB);
    const size_t sstride = (size_t)n_grp * 4 * NB * 32 * 256;                // elements per slot (whole groups)

    auto issue = [&](int l, int tg, int buf) {           // stage (layer l, tile group tg) -> LDS buffer buf
        const int nch = l_nch(l), nts = min(NTS, l_nt(l) - tg * NTS), pieces = nts * nch * NP;
        // uniform (scalar) piece offset + one 32-bit lane offset; the empty asm keeps the compiler from precomputing a
        // 64-bit vector address per piece and stage in the kernel prologue (it spilled ~180 registers doing so)
        uint32_t off = (uint32_t)(f_off(l) + tg * NTS * nch) * NP * 1024 + wave * 1024;
        unsigned char *dst = smem_tf + buf * STAGE + wave * 1024;
        for (int i = wave; i < pieces; i += 4) {
            asm volatile("" : "+s"(off));
            glds16(reinterpret_cast<const unsigned char *>(a.img) + off + lane16_k, dst);
            off += 4096; dst += 4096;
        }
    };
    int buf = 0;
    if ((int64_t)blockIdx.x < n_grp) issue(0, 0, 0);
    for (int64_t grp = blockIdx.x; grp < n_grp; grp += gridDim.x) {
        const int64_t blk0_g = (grp * 4 + wave) * NB;
        const bool more_grp = grp + gridDim.x < n_grp;
        bf16x8 x[NB][XP][16];
        uint32_t x2a[NB][16][4];                         // fp32 mode: piece 2 (AGPRs)
        // encoding operands; rows of blocks past the end are clamped to the last block (their results are never stored
        // row-major, and the fragment-layout buffers are whole groups)
        auto load_rows = [&](const float *src, int ld, int nchunks, bf16x8 (&e)[NB][NP][4], int64_t blk0, int sl, int hi) {
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int64_t blk = blk0 + u < n_blk ? blk0 + u : n_blk - 1;
                const float *xp = src + (blk * 32 + sl) * ld + 8 * hi;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (c >= nchunks) continue;
                    const float4 v0 = *reinterpret_cast<const float4 *>(xp + 16 * c), v1 = *reinterpret_cast<const float4 *>(xp + 16 * c + 4);
                    const float xs[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                    bf16x8 o[3];
                    split8<NP>(xs, o);
#pragma unroll
                    for (int p = 0; p < NP; ++p) e[u][p][c] = o[p];
                }
            }
        };
        // l: this layer, ln: the layer whose first stage follows (-1: none)
        auto layer = [&](auto shape, const int l, const int ln, const int slot) {
            using S = decltype(shape);
            constexpr int NH = S::NH, NE = S::NE, NT = S::NT, ACT = S::ACT, nch = NH + NE;
            constexpr int NTG = (NT + NTS - 1) / NTS;
            // opaque per-layer copies of the lane offset and block index: everything addressed through them is computed
            // inside the layer instead of once per kernel / group for all six layer shapes (which spilled ~300 registers)
            uint32_t lane16 = lane16_k;
            int64_t blk0 = blk0_g;
            asm volatile("" : "+v"(lane16), "+s"(blk0));
            const int lane = lane16 >> 4, hi = lane >> 5, sl = lane & 31;
            bf16x8 e[NB][NP][4];
            if (NE == 4) load_rows(a.enc, a.ld_enc, 4, e, blk0, sl, hi);
            if (NE == 2) load_rows(a.view, a.ld_view, 2, e, blk0, sl, hi);
            f32x16 acc[NB][NT];
#pragma unroll
            for (int tg = 0; tg < NTG; ++tg) {
                __syncthreads();                         // stage (l, tg) has landed (the fence drains this wave's LDS-DMA)
                if (tg + 1 < NTG) issue(l, tg + 1, buf ^ 1);
                else if (ln >= 0) issue(ln, 0, buf ^ 1);
                const unsigned char *st = smem_tf + buf * STAGE + lane16;
#pragma unroll
                for (int tt = 0; tt < NTS; ++tt) {
                    const int t = tg * NTS + tt;
                    if (t >= NT) continue;
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
                                const bf16x8 xv = c < NH ? ((NP == 3 && PR::A[k] == 2) ? agpr_get(x2a[u][c & 15]) : x[u][PR::A[k] % XP][c & 15])
                                                         : e[u][PR::A[k]][(c - NH) & 3];
                                if (NB * NTS == 1 && (m & 1)) part = MFMAB(w[PR::W[k]], xv, part);
                                else acc[u][t] = MFMAB(w[PR::W[k]], xv, acc[u][t]);
                            }
                    }
                    if (NB * NTS == 1) acc[0][t] += part;
                }
                buf ^= 1;
                TRUNK_FENCE();
            }
            const float *bl = bias + lb_off(l);
            if (ACT == A_SIGMA) {                        // row 0 of the tile: lanes hi == 0, register 0 (ngp.py:45-65 trunc_exp)
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const int64_t row = (blk0 + u) * 32 + sl;
                    if (hi == 0 && row < a.n) a.sigma[row] = a.sel[row] ? __expf(acc[u][0][0] + bl[0] - 1.f) : 0.f;
                }
                return;
            }
            if (ACT == A_RGB) {                          // rows 0 .. C-1: lanes hi == 0, registers 0 .. 3
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const int64_t row = (blk0 + u) * 32 + sl;
                    if (hi == 0 && row < a.n) {
                        float r[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) r[c] = c < a.C ? softplus1(acc[u][0][c] + bl[c]) : 0.f;
                        *reinterpret_cast<float4 *>(a.rgb4 + row * 4) = make_float4(r[0], r[1], r[2], r[3]);
                    }
                }
                return;
            }
            // bias + activation; the accumulators become the next layer's operands and the saved copy
            ST *sv = reinterpret_cast<ST *>(a.acts) + (size_t)slot * sstride;
#pragma unroll
            for (int u = 0; u < NB; ++u) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    float y[16];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 b4 = *reinterpret_cast<const float4 *>(bl + t * 32 + 8 * q + 4 * hi);
                        const float z[4] = {acc[u][t][4 * q] + b4.x, acc[u][t][4 * q + 1] + b4.y, acc[u][t][4 * q + 2] + b4.z, acc[u][t][4 * q + 3] + b4.w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) y[4 * q + j] = ACT == A_SP100 ? softplus100(z[j]) : z[j];
                    }
                    bf16x8 lo[3], hi8[3];
                    pack_tile<MODE>(y, lo, hi8);
#pragma unroll
                    for (int p = 0; p < XP; ++p) { x[u][p][2 * t] = lo[p]; x[u][p][2 * t + 1] = hi8[p]; }
                    if (NP == 3) { agpr_put(x2a[u][2 * t], lo[2]); agpr_put(x2a[u][2 * t + 1], hi8[2]); }
                    if (SAVE) store_tile<MODE>(sv, blk0 + u, t, lane, y, lo[0], hi8[0]);
                    TRUNK_FENCE();
                }
            }
        };
        layer(Shape<0, 4, 8, A_SP100>(), 0, 1, 0);
        for (int l = 1; l < L_SKIP; ++l) layer(Shape<16, 0, 8, A_SP100>(), l, l + 1, l);
        layer(Shape<16, 4, 8, A_SP100>(), L_SKIP, 6, L_SKIP);
        layer(Shape<16, 0, 8, A_SP100>(), 6, 7, 6);
        layer(Shape<16, 0, 8, A_SP100>(), 7, L_SIGMA, 7);
        layer(Shape<16, 0, 1, A_SIGMA>(), L_SIGMA, FULL ? L_BOTT : (more_grp ? 0 : -1), -1);
        if (FULL) {
            layer(Shape<16, 0, 8, A_NONE>(), L_BOTT, L_RGBH, 8);
            layer(Shape<16, 2, 4, A_SP100>(), L_RGBH, L_RGBO, 9);
            layer(Shape<8, 0, 1, A_RGB>(), L_RGBO, more_grp ? 0 : -1, -1);
        }
    }
}

// ---- backward (data) ---------------------------------------------------------------------------------------------------
// steps: 11 (dz_rgb -> dz of the colour hidden layer), 10 (-> d bottleneck), 9 (+ sigma -> dz7), 7 .. 1
template <int NH_, int NE_, int NT_, bool DERIV_> struct BShape { static constexpr int NH = NH_, NE = NE_, NT = NT_; static constexpr bool DERIV = DERIV_; };

template <int MODE>
__global__ __launch_bounds__(256, 1) void vfield_bwd_kernel(FieldArgs a) {
    using C = TC<MODE>;
    using PR = Pairs<MODE>;
    typedef typename C::ST ST;
    constexpr int NP = C::NP, NB = C::NB, NTS = C::NTS, XP = NP == 3 ? 2 : NP;
    constexpr int STAGE = NTS * 17 * NP * 1024;
    constexpr int HV = MODE == 1 ? 2 : 4;                // 16-byte loads per lane and tile of a saved layer
    constexpr int PD = 4;                                // saved-activation tiles in flight
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_tb[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hi = lane >> 5, sl = lane & 31;
    const uint32_t lane16_k = lane * 16;
    const int64_t n_blk = (a.n + 31) >> 5, n_grp = (n_blk + 4 * NB - 1) / (4 * NB);
    const size_t sstride = (size_t)n_grp * 4 * NB * 32 * 256;                // whole groups: stores need no bounds

    auto issue = [&](int l, int tg, int buf) {           // stage (step l, input-tile group tg)
        const int nch = b_nch(l), nts = min(NTS, b_nt(l) - tg * NTS), pieces = nts * nch * NP;
        uint32_t off = (uint32_t)(b_off(l) + tg * NTS * nch) * NP * 1024 + wave * 1024;       // see vfield_fwd_kernel
        unsigned char *dst = smem_tb + buf * STAGE + wave * 1024;
        for (int i = wave; i < pieces; i += 4) {
            asm volatile("" : "+s"(off));
            glds16(reinterpret_cast<const unsigned char *>(a.img) + off + lane16_k, dst);
            off += 4096; dst += 4096;
        }
    };
    int buf = 0;
    if ((int64_t)blockIdx.x < n_grp) issue(L_RGBO, 0, 0);
    for (int64_t grp = blockIdx.x; grp < n_grp; grp += gridDim.x) {
        const int64_t blk0_g = (grp * 4 + wave) * NB;
        const bool more_grp = grp + gridDim.x < n_grp;
        bf16x8 x[NB][XP][16];
        uint32_t x2a[NB][16][4];                         // fp32 mode: piece 2 (AGPRs)
        // one extra operand chunk from a row-major [n_pad][32] gradient buffer: columns 8 hi .. 8 hi + 7 (step 11: dz_rgb),
        // or k-slot 0 only (step 9: dz_sigma, column 0)
        auto load_extra = [&](const float *src, bool slot0, bf16x8 (&e)[NB][NP], int64_t blk0, int sl, int hi) {
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int64_t blk = blk0 + u < n_blk ? blk0 + u : n_blk - 1;
                const float *xp = src + (blk * 32 + sl) * 32 + 8 * hi;
                float xs[8];
                if (slot0) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) xs[j] = 0.f;
                    xs[0] = hi == 0 ? xp[0] : 0.f;
                } else {
                    const float4 v0 = *reinterpret_cast<const float4 *>(xp), v1 = *reinterpret_cast<const float4 *>(xp + 4);
                    xs[0] = v0.x; xs[1] = v0.y; xs[2] = v0.z; xs[3] = v0.w; xs[4] = v1.x; xs[5] = v1.y; xs[6] = v1.z; xs[7] = v1.w;
                }
                bf16x8 o[3];
                split8<NP>(xs, o);
#pragma unroll
                for (int p = 0; p < NP; ++p) e[u][p] = o[p];
            }
        };
        // step l; ln: the step whose first stage follows (-1 none); hslot: saved activation whose derivative multiplies the
        // result (DERIV); dslot: where the result (a pre-activation gradient) is stored
        auto step = [&](auto shape, const int l, const int ln, const int hslot, const int dslot, const float *extra) {
            using S = decltype(shape);
            constexpr int NH = S::NH, NE = S::NE, NT = S::NT, nch = NH + NE;
            constexpr bool DERIV = S::DERIV;
            constexpr int NTG = (NT + NTS - 1) / NTS;
            uint32_t lane16 = lane16_k;                  // opaque per-step copies: see vfield_fwd_kernel
            int64_t blk0 = blk0_g;
            asm volatile("" : "+v"(lane16), "+s"(blk0));
            const int lane = lane16 >> 4, hi = lane >> 5, sl = lane & 31;
            const ST *hs = reinterpret_cast<const ST *>(a.acts) + (size_t)(DERIV ? hslot : 0) * (a.acts_sstride ? (size_t)a.acts_sstride : sstride);
            uint4 hpre[PD][HV];
            auto load_h = [&](int i, uint4 (&dst)[HV]) {                                           // i = u * NT + t
                const int u = i / NT, t = i % NT;
                const int64_t blk = blk0 + u;
                const uint4 *p = MODE == 1 ? reinterpret_cast<const uint4 *>(hs + ((blk * 16 + 2 * t) * 64 + lane) * 8)
                                           : reinterpret_cast<const uint4 *>(hs + ((blk * 8 + t) * 4 * 64 + lane) * 4);
#pragma unroll
                for (int q = 0; q < HV; ++q) dst[q] = p[q * 64];
            };
            bf16x8 e[NB][NP];
            if (NE) load_extra(extra, l == L_BOTT, e, blk0, sl, hi);
            f32x16 acc[NB][NT];
#pragma unroll
            for (int tg = 0; tg < NTG; ++tg) {
                __syncthreads();
                if (tg + 1 < NTG) issue(l, tg + 1, buf ^ 1);
                else if (ln >= 0) issue(ln, 0, buf ^ 1);
                if (DERIV && tg == NTG - 1) {
#pragma unroll
                    for (int i = 0; i < PD; ++i) load_h(i, hpre[i]);
                }
                const unsigned char *st = smem_tb + buf * STAGE + lane16;
#pragma unroll
                for (int tt = 0; tt < NTS; ++tt) {
                    const int t = tg * NTS + tt;
                    if (t >= NT) continue;
                    f32x16 part;
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
                                const bf16x8 xv = c < NH ? ((NP == 3 && PR::A[k] == 2) ? agpr_get(x2a[u][c & 15]) : x[u][PR::A[k] % XP][c & 15])
                                                         : e[u][PR::A[k]];
                                if (NB * NTS == 1 && (m & 1)) part = MFMAB(w[PR::W[k]], xv, part);
                                else acc[u][t] = MFMAB(w[PR::W[k]], xv, acc[u][t]);
                            }
                    }
                    if (NB * NTS == 1) acc[0][t] += part;
                }
                buf ^= 1;
                TRUNK_FENCE();
            }
            ST *sv = reinterpret_cast<ST *>(a.dz) + (size_t)dslot * sstride;
#pragma unroll
            for (int i = 0; i < NB * NT; ++i) {
                const int u = i / NT, t = i % NT;
                float y[16];
                if (DERIV) {
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
                    if (i + PD < NB * NT) load_h(i + PD, hpre[i % PD]);
#pragma unroll
                    for (int g = 0; g < 16; ++g) y[g] = acc[u][t][g] * dsoftplus_from_out(h[g], 100.f);
                } else {
#pragma unroll
                    for (int g = 0; g < 16; ++g) y[g] = acc[u][t][g];
                }
                bf16x8 lo[3], hi8[3];
                pack_tile<MODE>(y, lo, hi8);
#pragma unroll
                for (int p = 0; p < XP; ++p) { x[u][p][2 * t] = lo[p]; x[u][p][2 * t + 1] = hi8[p]; }
                if (NP == 3) { agpr_put(x2a[u][2 * t], lo[2]); agpr_put(x2a[u][2 * t + 1], hi8[2]); }
                store_tile<MODE>(sv, blk0 + u, t, lane, y, lo[0], hi8[0]);
                TRUNK_FENCE();
            }
        };
        step(BShape<0, 1, 4, true>(), L_RGBO, L_RGBH, 9, 9, a.dz_rgb);
        step(BShape<8, 0, 8, false>(), L_RGBH, L_BOTT, 0, 8, nullptr);
        step(BShape<16, 1, 8, true>(), L_BOTT, 7, 7, 7, a.dz_sig);
        for (int l = 7; l >= 1; --l) step(BShape<16, 0, 8, true>(), l, l > 1 ? l - 1 : (more_grp ? L_RGBO : -1), l - 1, l - 1, nullptr);
    }
}

// ---- bf16 mode forward, software-pipelined --------------------------------------------------------------------------------
// In vfield_fwd_kernel<1> the matrix cores idle while a wave runs the activation epilogue (per output: accumulator read, bias,
// two quarter-rate transcendentals, pack: ~1.5x the MFMA time of a layer, and with one wave per SIMD nothing else to issue).
// Here the epilogue of tile group g is issued between the MFMAs of group g + 1 -- in eight slices of eight values, one
// every second k-chunk -- and the last group of a layer between the first MFMAs of the NEXT layer, whose k-chunks 0 .. 11
// do not depend on it.  The outputs of a layer's finished groups wait in a second operand set `xn` until the layer ends.
// The saved copies of a slice group are stored at the START of the following stage (from the operand registers they sit in
// anyway): a store issued inside a stage is still in flight at the stage's closing barrier, whose fence waits for it
// (+0.3 ms per pass when the slices issued them; one burst per layer instead spilled 136 registers).
template <int NH_, int NE_, int NT_, int ACT_, int PACT_, int PNT_, int S0_ = -1> struct Shape1 {
    static constexpr int NH = NH_, NE = NE_, NT = NT_, ACT = ACT_, PACT = PACT_, PNT = PNT_;     // P*: the pending tile group's layer
    static constexpr int S0 = S0_;                       // >= 0: first of four operand chunks whose saved copy this layer's first stage stores as well
};

template <bool SAVE, bool FULL>
__global__ __launch_bounds__(256, 1) void vfield_fwd1_kernel(FieldArgs a) {
    constexpr int NB = 2, NTS = 2;
    constexpr int STAGE = NTS * 20 * 1024;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    float *bias = reinterpret_cast<float *>(smem_all);
    unsigned char *smem_tf = smem_all + LB_FLOATS * 4;
    const int lane_k = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane16_k = lane_k * 16;
    for (int i = threadIdx.x; i < LB_FLOATS; i += 256) {
        float b = 0.f;
        if (i < 2048) b = a.P[l_boff(i >> 8, a.C) + (i & 255)];
        else if (i < LB_RGBH) b = a.P[l_boff(L_BOTT, a.C) + i - LB_BOTT];
        else if (i < LB_SIGMA) b = a.P[l_boff(L_RGBH, a.C) + i - LB_RGBH];
        else if (i == LB_SIGMA) b = a.P[l_boff(L_SIGMA, a.C)];
        else if (i >= LB_RGBO && i < LB_RGBO + a.C) b = a.P[l_boff(L_RGBO, a.C) + i - LB_RGBO];
        bias[i] = b;
    }
    const int64_t n_blk = (a.n + 31) >> 5, n_grp = (n_blk + 4 * NB - 1) / (4 * NB);
    const size_t sstride = (size_t)n_grp * 4 * NB * 32 * 256;

    auto issue = [&](int l, int tg, int buf) {
        const int nch = l_nch(l), nts = min(NTS, l_nt(l) - tg * NTS), pieces = nts * nch;
        uint32_t off = (uint32_t)(f_off(l) + tg * NTS * nch) * 1024 + wave * 1024;
        unsigned char *dst = smem_tf + buf * STAGE + wave * 1024;
        for (int i = wave; i < pieces; i += 4) {
            asm volatile("" : "+s"(off));
            glds16(reinterpret_cast<const unsigned char *>(a.img) + off + lane16_k, dst);
            off += 4096; dst += 4096;
        }
    };
    int buf = 0;
    if ((int64_t)blockIdx.x < n_grp) issue(0, 0, 0);
    for (int64_t grp = blockIdx.x; grp < n_grp; grp += gridDim.x) {
        const int64_t blk0_g = (grp * 4 + wave) * NB;
        const bool more_grp = grp + gridDim.x < n_grp;
        bf16x8 x[NB][16], xn[NB][12];
        f32x16 pend[NB][2];                              // the previous layer's last tile group, epilogue not yet run
        // l: this layer, ln: the layer whose first stage follows (-1: none), slot: its saved slot; pl / pslot: the layer the
        // pending group belongs to
        auto layer = [&](auto shape, const int l, const int ln, const int slot, const int pl, const int pslot) {
            using S = decltype(shape);
            constexpr int NH = S::NH, NE = S::NE, NT = S::NT, ACT = S::ACT, PACT = S::PACT, PNT = S::PNT, nch = NH + NE;
            constexpr int NTG = (NT + NTS - 1) / NTS;
            constexpr int PC0 = PACT >= 0 ? 2 * PNT - 4 : 0;             // k-chunks the pending group will become
            uint32_t lane16 = lane16_k;
            int64_t blk0 = blk0_g;
            asm volatile("" : "+v"(lane16), "+s"(blk0));
            const int lane = lane16 >> 4, hi = lane >> 5, sl = lane & 31;
            if (PACT >= 0) {
#pragma unroll
                for (int u = 0; u < NB; ++u)
#pragma unroll
                    for (int c = 0; c < PC0; ++c) x[u][c] = xn[u][c];
            }
            bf16x8 e[NB][4];
            if (NE) {
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const int64_t blk = blk0 + u < n_blk ? blk0 + u : n_blk - 1;
                    const float *xp = (NE == 4 ? a.enc + (blk * 32 + sl) * a.ld_enc : a.view + (blk * 32 + sl) * a.ld_view) + 8 * hi;
#pragma unroll
                    for (int c = 0; c < NE; ++c) {
                        const float4 v0 = *reinterpret_cast<const float4 *>(xp + 16 * c), v1 = *reinterpret_cast<const float4 *>(xp + 16 * c + 4);
                        const float xs[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                        bf16x8 o[3];
                        split8<1>(xs, o);
                        e[u][c] = o[0];
                    }
                }
            }
            // eight values of one tile (half h of its 16 registers) -> one k-chunk of the next layer (+ the saved copy)
            auto slice = [&](auto act_c, const f32x16 &tile, const int t, const int h, const float *bl, const int u, bf16x8 &dst,
                             const int sslot) {
                constexpr int A = decltype(act_c)::value;
                const float4 b0 = *reinterpret_cast<const float4 *>(bl + t * 32 + 16 * h + 4 * hi);
                const float4 b1 = *reinterpret_cast<const float4 *>(bl + t * 32 + 16 * h + 8 + 4 * hi);
                const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
                bf16x8 p;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float v = tile[8 * h + j] + bb[j];
                    p[j] = (__bf16)(A == A_SP100 ? softplus100(v) : v);
                }
                dst = p;
                (void)sslot; (void)u;
            };
            // saved copy of four operand chunks c0 .. c0 + 3 (both blocks) of slot sslot, from x or xn
            auto late_store = [&](const bool from_x, const int c0, const int sslot) {
                if (!SAVE) return;
                __bf16 *sv = reinterpret_cast<__bf16 *>(a.acts) + (size_t)sslot * sstride;
#pragma unroll
                for (int u = 0; u < NB; ++u)
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        *reinterpret_cast<bf16x8 *>(sv + (((blk0 + u) * 16 + c0 + k) * 64 + lane) * 8) = from_x ? x[u][(c0 + k) & 15] : xn[u][(c0 + k) % 12];
            };
            f32x16 acc[NB][NT];
#pragma unroll
            for (int tg = 0; tg < NTG; ++tg) {
                __syncthreads();
                if (tg + 1 < NTG) issue(l, tg + 1, buf ^ 1);
                else if (ln >= 0) issue(ln, 0, buf ^ 1);
                const unsigned char *st = smem_tf + buf * STAGE + lane16;
                // saved copies of what the previous stage's slices produced
                if (tg == 0 && PACT >= 0) late_store(true, 2 * PNT - 8, pslot);
                if (tg == 0 && S::S0 >= 0) late_store(true, S::S0, pslot);
                if (tg == 1 && PACT >= 0) late_store(true, PC0, pslot);
                if (tg >= 2) late_store(false, 4 * (tg - 2), slot);
#pragma unroll
                for (int tt = 0; tt < NTS; ++tt)
                    if (tg * NTS + tt < NT) {
#pragma unroll
                        for (int g = 0; g < 16; ++g)
#pragma unroll
                            for (int u = 0; u < NB; ++u) acc[u][tg * NTS + tt][g] = 0.f;
                    }
                // the eight slices of the group whose epilogue runs inside this stage, spread over its first LIM k-chunks
                const bool HAS = tg == 0 ? PACT >= 0 : true;
                const int LIM = tg == 0 ? (PC0 < nch ? PC0 : nch) : nch;
#pragma unroll
                for (int c = 0; c < nch; ++c) {
#pragma unroll
                    for (int tt = 0; tt < NTS; ++tt) {
                        const int t = tg * NTS + tt;
                        if (t >= NT) continue;
                        const bf16x8 w = *reinterpret_cast<const bf16x8 *>(st + (tt * nch + c) * 1024);
#pragma unroll
                        for (int u = 0; u < NB; ++u) acc[u][t] = MFMAB(w, c < NH ? x[u][c & 15] : e[u][(c - NH) & 3], acc[u][t]);
                    }
                    if (HAS && c < LIM) {
#pragma unroll
                        for (int s = c * 8 / (LIM > 0 ? LIM : 1); s < (c + 1) * 8 / (LIM > 0 ? LIM : 1); ++s) {
                            const int u = s >> 2, tt = (s >> 1) & 1, h = s & 1;
                            if (tg == 0) {
                                const int t = PNT - 2 + tt;
                                slice(IntC<(PACT >= 0 ? PACT : 0)>(), pend[u][tt], t, h, bias + lb_off(pl), u, x[u][(2 * t + h) & 15], pslot);
                            } else {
                                const int t = 2 * (tg - 1) + tt;
                                slice(IntC<ACT>(), acc[u][t < NT ? t : 0], t, h, bias + lb_off(l), u, xn[u][(2 * t + h) % 12], slot);
                            }
                        }
                    }
                }
                buf ^= 1;
                TRUNK_FENCE();
            }
            const float *bl = bias + lb_off(l);
            if (ACT == A_RGB && PACT >= 0) late_store(true, PC0, pslot);              // last layer: nothing follows
            if (ACT == A_SIGMA) {
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const int64_t row = (blk0 + u) * 32 + sl;
                    if (hi == 0 && row < a.n) a.sigma[row] = a.sel[row] ? __expf(acc[u][0][0] + bl[0] - 1.f) : 0.f;
                }
            } else if (ACT == A_RGB) {
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const int64_t row = (blk0 + u) * 32 + sl;
                    if (hi == 0 && row < a.n) {
                        float r[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) r[c] = c < a.C ? softplus1(acc[u][0][c] + bl[c]) : 0.f;
                        *reinterpret_cast<float4 *>(a.rgb4 + row * 4) = make_float4(r[0], r[1], r[2], r[3]);
                    }
                }
            } else {                                     // the last tile group stays pending
#pragma unroll
                for (int u = 0; u < NB; ++u)
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) pend[u][tt] = acc[u][NT >= 2 ? NT - 2 + tt : 0];
            }
        };
        layer(Shape1<0, 4, 8, A_SP100, -1, 8>(), 0, 1, 0, 0, 0);
        for (int l = 1; l < L_SKIP; ++l) layer(Shape1<16, 0, 8, A_SP100, A_SP100, 8>(), l, l + 1, l, l - 1, l - 1);
        layer(Shape1<16, 4, 8, A_SP100, A_SP100, 8>(), L_SKIP, 6, L_SKIP, 4, 4);
        layer(Shape1<16, 0, 8, A_SP100, A_SP100, 8>(), 6, 7, 6, 5, 5);
        layer(Shape1<16, 0, 8, A_SP100, A_SP100, 8>(), 7, L_SIGMA, 7, 6, 6);
        layer(Shape1<16, 0, 1, A_SIGMA, A_SP100, 8>(), L_SIGMA, FULL ? L_BOTT : (more_grp ? 0 : -1), -1, 7, 7);
        if (FULL) {
            layer(Shape1<16, 0, 8, A_NONE, -1, 8, 12>(), L_BOTT, L_RGBH, 8, 7, 7);      // (+ the copy of layer 7's last group, finished inside the sigma stage)
            layer(Shape1<16, 2, 4, A_SP100, A_NONE, 8>(), L_RGBH, L_RGBO, 9, L_BOTT, 8);
            layer(Shape1<8, 0, 1, A_RGB, A_SP100, 4>(), L_RGBO, more_grp ? 0 : -1, -1, L_RGBH, 9);
        }
    }
}

// ---- bf16 mode, value + forward-mode tangent in one launch (the log-intensity-gradient loss's third render) ---------------
// d/dt of the field through all twelve layers beside the value (external/mlp.py:126-205 under utils/autograd.py:4-34): per
// layer z = W a + b, zd = W ad; y = sp(z), yd = sp'(z) zd.  The two-blocks-per-wave structure of the bf16 kernels carries it
// as is: "block" 0 of a wave is the VALUE of its 32 samples, "block" 1 their TANGENT -- the same weight fragment feeds both
// MFMAs, the tangent has no bias, and the activation epilogue sees z and zd of a neuron in the same lane.  Saved: the
// bf16-rounded y (slots of `acts`) and yd (`actsd`), same fragment layout and slot stride as vfield_fwd_kernel<1>'s, so the
// weight-gradient kernel reads either.  The reverse pass of (dy, dyd):
//     dz = dy s + dyd zd s' = dy s + dyd yd beta (1 - s),   dzd = dyd s          (s = sp'(z) = 1 - exp(-beta y))
// needs nothing else.  Until round 5 this render ran ~70 per-layer launches with fp32 activations through HBM
// (ren_dense_*, 31 ms of a 39 ms step at 524 k samples).
template <bool SAVE>
__global__ __launch_bounds__(256, 1) void vfield_fwd_jvp_kernel(FieldArgs a) {
    constexpr int NTS = 2;
    constexpr int STAGE = NTS * 20 * 1024;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    float *bias = reinterpret_cast<float *>(smem_all);
    unsigned char *smem_tf = smem_all + LB_FLOATS * 4;
    const int lane_k = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane16_k = lane_k * 16;
    for (int i = threadIdx.x; i < LB_FLOATS; i += 256) {
        float b = 0.f;
        if (i < 2048) b = a.P[l_boff(i >> 8, a.C) + (i & 255)];
        else if (i < LB_RGBH) b = a.P[l_boff(L_BOTT, a.C) + i - LB_BOTT];
        else if (i < LB_SIGMA) b = a.P[l_boff(L_RGBH, a.C) + i - LB_RGBH];
        else if (i == LB_SIGMA) b = a.P[l_boff(L_SIGMA, a.C)];
        else if (i >= LB_RGBO && i < LB_RGBO + a.C) b = a.P[l_boff(L_RGBO, a.C) + i - LB_RGBO];
        bias[i] = b;
    }
    const int64_t n_blk = (a.n + 31) >> 5, n_grp = (n_blk + 3) >> 2;          // one 32-sample block per wave
    const size_t sstride = (size_t)((n_blk + 7) >> 3) * 8 * 32 * 256;         // slot stride of the plain kernels (whole groups of 8)

    auto issue = [&](int l, int tg, int buf) {
        const int nch = l_nch(l), nts = min(NTS, l_nt(l) - tg * NTS), pieces = nts * nch;
        uint32_t off = (uint32_t)(f_off(l) + tg * NTS * nch) * 1024 + wave * 1024;
        unsigned char *dst = smem_tf + buf * STAGE + wave * 1024;
        for (int i = wave; i < pieces; i += 4) {
            asm volatile("" : "+s"(off));
            glds16(reinterpret_cast<const unsigned char *>(a.img) + off + lane16_k, dst);
            off += 4096; dst += 4096;
        }
    };
    int buf = 0;
    if ((int64_t)blockIdx.x < n_grp) issue(0, 0, 0);
    for (int64_t grp = blockIdx.x; grp < n_grp; grp += gridDim.x) {
        const int64_t blk0_g = grp * 4 + wave;
        const bool more_grp = grp + gridDim.x < n_grp;
        bf16x8 x[2][16];                                                       // [value | tangent] operand chunks
        auto load_rows = [&](const float *src, const float *srcd, int ld, int nchunks, bf16x8 (&e)[2][4], int64_t blk0, int sl, int hi) {
            const int64_t blk = blk0 < n_blk ? blk0 : n_blk - 1;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const float *xp = (u ? srcd : src) + (blk * 32 + sl) * ld + 8 * hi;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (c >= nchunks) continue;
                    const float4 v0 = *reinterpret_cast<const float4 *>(xp + 16 * c), v1 = *reinterpret_cast<const float4 *>(xp + 16 * c + 4);
                    const float xs[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                    bf16x8 o[3];
                    split8<1>(xs, o);
                    e[u][c] = o[0];
                }
            }
        };
        auto layer = [&](auto shape, const int l, const int ln, const int slot) {
            using S = decltype(shape);
            constexpr int NH = S::NH, NE = S::NE, NT = S::NT, ACT = S::ACT, nch = NH + NE;
            constexpr int NTG = (NT + NTS - 1) / NTS;
            uint32_t lane16 = lane16_k;
            int64_t blk0 = blk0_g;
            asm volatile("" : "+v"(lane16), "+s"(blk0));
            const int lane = lane16 >> 4, hi = lane >> 5, sl = lane & 31;
            bf16x8 e[2][4];
            if (NE == 4) load_rows(a.enc, a.encd, a.ld_enc, 4, e, blk0, sl, hi);
            if (NE == 2) load_rows(a.view, a.viewd, a.ld_view, 2, e, blk0, sl, hi);
            f32x16 acc[2][NT];
#pragma unroll
            for (int tg = 0; tg < NTG; ++tg) {
                __syncthreads();
                if (tg + 1 < NTG) issue(l, tg + 1, buf ^ 1);
                else if (ln >= 0) issue(ln, 0, buf ^ 1);
                const unsigned char *st = smem_tf + buf * STAGE + lane16;
#pragma unroll
                for (int tt = 0; tt < NTS; ++tt) {
                    const int t = tg * NTS + tt;
                    if (t >= NT) continue;
#pragma unroll
                    for (int g = 0; g < 16; ++g) { acc[0][t][g] = 0.f; acc[1][t][g] = 0.f; }
                    const unsigned char *wt = st + tt * nch * 1024;
#pragma unroll
                    for (int c = 0; c < nch; ++c) {
                        const bf16x8 w = *reinterpret_cast<const bf16x8 *>(wt + c * 1024);
#pragma unroll
                        for (int u = 0; u < 2; ++u) acc[u][t] = MFMAB(w, c < NH ? x[u][c & 15] : e[u][(c - NH) & 3], acc[u][t]);
                    }
                }
                buf ^= 1;
                TRUNK_FENCE();
            }
            const float *bl = bias + lb_off(l);
            const int64_t row = blk0 * 32 + sl;
            if (ACT == A_SIGMA) {
                if (hi == 0 && row < a.n) {
                    a.sigma[row] = a.sel[row] ? __expf(acc[0][0][0] + bl[0] - 1.f) : 0.f;
                    a.zsd4[row * 4] = acc[1][0][0];
                }
                return;
            }
            if (ACT == A_RGB) {
                if (hi == 0 && row < a.n) {
                    float r[4], rd[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) { r[c] = c < a.C ? softplus1(acc[0][0][c] + bl[c]) : 0.f; rd[c] = c < a.C ? acc[1][0][c] : 0.f; }
                    *reinterpret_cast<float4 *>(a.rgb4 + row * 4) = make_float4(r[0], r[1], r[2], r[3]);
                    *reinterpret_cast<float4 *>(a.zod4 + row * 4) = make_float4(rd[0], rd[1], rd[2], rd[3]);
                }
                return;
            }
            __bf16 *sv = reinterpret_cast<__bf16 *>(a.acts) + (size_t)slot * sstride;
            __bf16 *svd = reinterpret_cast<__bf16 *>(a.actsd) + (size_t)slot * sstride;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                float y[16], yd[16];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 b4 = *reinterpret_cast<const float4 *>(bl + t * 32 + 8 * q + 4 * hi);
                    const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float z = acc[0][t][4 * q + j] + bb[j], zd = acc[1][t][4 * q + j];
                        if (ACT == A_SP100) {
                            y[4 * q + j] = softplus100(z);
                            yd[4 * q + j] = dsoftplus_from_out(y[4 * q + j], 100.f) * zd;
                        } else {
                            y[4 * q + j] = z; yd[4 * q + j] = zd;
                        }
                    }
                }
                bf16x8 lo[3], hi8[3], lod[3], hid[3];
                pack_tile<1>(y, lo, hi8);
                pack_tile<1>(yd, lod, hid);
                x[0][2 * t] = lo[0]; x[0][2 * t + 1] = hi8[0];
                x[1][2 * t] = lod[0]; x[1][2 * t + 1] = hid[0];
                if (SAVE) {
                    store_tile<1>(sv, blk0, t, lane, y, lo[0], hi8[0]);
                    store_tile<1>(svd, blk0, t, lane, yd, lod[0], hid[0]);
                }
                TRUNK_FENCE();
            }
        };
        layer(Shape<0, 4, 8, A_SP100>(), 0, 1, 0);
        for (int l = 1; l < L_SKIP; ++l) layer(Shape<16, 0, 8, A_SP100>(), l, l + 1, l);
        layer(Shape<16, 4, 8, A_SP100>(), L_SKIP, 6, L_SKIP);
        layer(Shape<16, 0, 8, A_SP100>(), 6, 7, 6);
        layer(Shape<16, 0, 8, A_SP100>(), 7, L_SIGMA, 7);
        layer(Shape<16, 0, 1, A_SIGMA>(), L_SIGMA, L_BOTT, -1);
        layer(Shape<16, 0, 8, A_NONE>(), L_BOTT, L_RGBH, 8);
        layer(Shape<16, 2, 4, A_SP100>(), L_RGBH, L_RGBO, 9);
        layer(Shape<8, 0, 1, A_RGB>(), L_RGBO, more_grp ? 0 : -1, -1);
    }
}

// reverse pass of vfield_fwd_jvp_kernel: "block" 0 carries d loss / d y, "block" 1 d loss / d yd through W^T; the epilogue
// couples them through the saved value (s, s') and the saved tangent (see above).  Results: dz (slots of `dz`) and dzd (`dzd`).
__global__ __launch_bounds__(256, 1) void vfield_bwd_jvp_kernel(FieldArgs a) {
    constexpr int NTS = 2;
    constexpr int STAGE = NTS * 17 * 1024;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_tb[];
    const int lane_k = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane16_k = lane_k * 16;
    const int64_t n_blk = (a.n + 31) >> 5, n_grp = (n_blk + 3) >> 2;
    const size_t sstride = (size_t)((n_blk + 7) >> 3) * 8 * 32 * 256;
    const size_t hstride = a.acts_sstride ? (size_t)a.acts_sstride : sstride;

    auto issue = [&](int l, int tg, int buf) {
        const int nch = b_nch(l), nts = min(NTS, b_nt(l) - tg * NTS), pieces = nts * nch;
        uint32_t off = (uint32_t)(b_off(l) + tg * NTS * nch) * 1024 + wave * 1024;
        unsigned char *dst = smem_tb + buf * STAGE + wave * 1024;
        for (int i = wave; i < pieces; i += 4) {
            asm volatile("" : "+s"(off));
            glds16(reinterpret_cast<const unsigned char *>(a.img) + off + lane16_k, dst);
            off += 4096; dst += 4096;
        }
    };
    int buf = 0;
    if ((int64_t)blockIdx.x < n_grp) issue(L_RGBO, 0, 0);
    for (int64_t grp = blockIdx.x; grp < n_grp; grp += gridDim.x) {
        const int64_t blk0_g = grp * 4 + wave;
        const bool more_grp = grp + gridDim.x < n_grp;
        bf16x8 x[2][16];
        auto load_extra = [&](const float *src, const float *srcd, bool slot0, bf16x8 (&e)[2], int64_t blk0, int sl, int hi) {
            const int64_t blk = blk0 < n_blk ? blk0 : n_blk - 1;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const float *xp = (u ? srcd : src) + (blk * 32 + sl) * 32 + 8 * hi;
                float xs[8];
                if (slot0) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) xs[j] = 0.f;
                    xs[0] = hi == 0 ? xp[0] : 0.f;
                } else {
                    const float4 v0 = *reinterpret_cast<const float4 *>(xp), v1 = *reinterpret_cast<const float4 *>(xp + 4);
                    xs[0] = v0.x; xs[1] = v0.y; xs[2] = v0.z; xs[3] = v0.w; xs[4] = v1.x; xs[5] = v1.y; xs[6] = v1.z; xs[7] = v1.w;
                }
                bf16x8 o[3];
                split8<1>(xs, o);
                e[u] = o[0];
            }
        };
        auto step = [&](auto shape, const int l, const int ln, const int hslot, const int dslot, const float *extra, const float *extrad) {
            using S = decltype(shape);
            constexpr int NH = S::NH, NE = S::NE, NT = S::NT, nch = NH + NE;
            constexpr bool DERIV = S::DERIV;
            constexpr int NTG = (NT + NTS - 1) / NTS;
            uint32_t lane16 = lane16_k;
            int64_t blk0 = blk0_g;
            asm volatile("" : "+v"(lane16), "+s"(blk0));
            const int lane = lane16 >> 4, hi = lane >> 5, sl = lane & 31;
            const __bf16 *hs = reinterpret_cast<const __bf16 *>(a.acts) + (size_t)(DERIV ? hslot : 0) * hstride;
            const __bf16 *hds = reinterpret_cast<const __bf16 *>(a.actsd) + (size_t)(DERIV ? hslot : 0) * hstride;
            bf16x8 e[2];
            if (NE) load_extra(extra, extrad, l == L_BOTT, e, blk0, sl, hi);
            // saved value / tangent tiles of the epilogue, PD tiles ahead (issued under the last stage's MFMAs)
            constexpr int PD = 2;
            uint4 hpre[PD][2], hdpre[PD][2];
            auto load_h = [&](int t, uint4 (&dst)[2], uint4 (&dstd)[2]) {
                const uint4 *p = reinterpret_cast<const uint4 *>(hs + ((blk0 * 16 + 2 * t) * 64 + lane) * 8);
                const uint4 *pd = reinterpret_cast<const uint4 *>(hds + ((blk0 * 16 + 2 * t) * 64 + lane) * 8);
                dst[0] = p[0]; dst[1] = p[64]; dstd[0] = pd[0]; dstd[1] = pd[64];
            };
            f32x16 acc[2][NT];
#pragma unroll
            for (int tg = 0; tg < NTG; ++tg) {
                __syncthreads();
                if (tg + 1 < NTG) issue(l, tg + 1, buf ^ 1);
                else if (ln >= 0) issue(ln, 0, buf ^ 1);
                if (DERIV && tg == NTG - 1) {
#pragma unroll
                    for (int i = 0; i < PD; ++i) load_h(i, hpre[i], hdpre[i]);
                }
                const unsigned char *st = smem_tb + buf * STAGE + lane16;
#pragma unroll
                for (int tt = 0; tt < NTS; ++tt) {
                    const int t = tg * NTS + tt;
                    if (t >= NT) continue;
#pragma unroll
                    for (int g = 0; g < 16; ++g) { acc[0][t][g] = 0.f; acc[1][t][g] = 0.f; }
                    const unsigned char *wt = st + tt * nch * 1024;
#pragma unroll
                    for (int c = 0; c < nch; ++c) {
                        const bf16x8 w = *reinterpret_cast<const bf16x8 *>(wt + c * 1024);
#pragma unroll
                        for (int u = 0; u < 2; ++u) acc[u][t] = MFMAB(w, c < NH ? x[u][c & 15] : e[u], acc[u][t]);
                    }
                }
                buf ^= 1;
                TRUNK_FENCE();
            }
            __bf16 *sv = reinterpret_cast<__bf16 *>(a.dz) + (size_t)dslot * sstride;
            __bf16 *svd = reinterpret_cast<__bf16 *>(a.dzd) + (size_t)dslot * sstride;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                float y[16], yd[16];
                if (DERIV) {
                    uint4 hc[2] = {hpre[t % PD][0], hpre[t % PD][1]}, hdc[2] = {hdpre[t % PD][0], hdpre[t % PD][1]};
                    if (t + PD < NT) load_h(t + PD, hpre[t % PD], hdpre[t % PD]);
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const bf16x8 hv = *reinterpret_cast<const bf16x8 *>(&hc[q]), hdv = *reinterpret_cast<const bf16x8 *>(&hdc[q]);
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const int g = 8 * q + j;
                            const float sd = dsoftplus_from_out((float)hv[j], 100.f);
                            y[g] = acc[0][t][g] * sd + acc[1][t][g] * (float)hdv[j] * (100.f * (1.f - sd));
                            yd[g] = acc[1][t][g] * sd;
                        }
                    }
                } else {
#pragma unroll
                    for (int g = 0; g < 16; ++g) { y[g] = acc[0][t][g]; yd[g] = acc[1][t][g]; }
                }
                bf16x8 lo[3], hi8[3], lod[3], hid[3];
                pack_tile<1>(y, lo, hi8);
                pack_tile<1>(yd, lod, hid);
                x[0][2 * t] = lo[0]; x[0][2 * t + 1] = hi8[0];
                x[1][2 * t] = lod[0]; x[1][2 * t + 1] = hid[0];
                store_tile<1>(sv, blk0, t, lane, y, lo[0], hi8[0]);
                store_tile<1>(svd, blk0, t, lane, yd, lod[0], hid[0]);
                TRUNK_FENCE();
            }
        };
        step(BShape<0, 1, 4, true>(), L_RGBO, L_RGBH, 9, 9, a.dz_rgb, a.dzd_rgb);
        step(BShape<8, 0, 8, false>(), L_RGBH, L_BOTT, 0, 8, nullptr, nullptr);
        step(BShape<16, 1, 8, true>(), L_BOTT, 7, 7, 7, a.dz_sig, a.dzd_sig);
        for (int l = 7; l >= 1; --l) step(BShape<16, 0, 8, true>(), l, l > 1 ? l - 1 : (more_grp ? L_RGBO : -1), l - 1, l - 1, nullptr, nullptr);
    }
}

// ---- fp32 mode (three-piece split): reduction-outer variants -------------------------------------------------------------------
// Three bf16 pieces of 256 activations are 192 operand registers per 32-sample block: beside the weight fragments and the
// epilogue they do not fit in the 256 architectural VGPRs (the tile-outer kernels above, instantiated for this mode, spilled
// 250-350 registers to scratch and reloaded the operands for every output tile).  Here the loops are turned around: the
// previous layer's activated outputs stay as fp32 in 128 registers (`yprev`, in the accumulator layout), a k-chunk of them
// is split into its three pieces right before use -- each chunk is used exactly once per layer, so the split costs what it
// cost as part of the epilogue -- and feeds the MFMAs of ALL output tiles, whose 128 accumulators fill the AGPRs.  The
// weight image of this mode is chunk-major ([chunk][tile][piece]) and a stage is two chunks of all tiles (<= 48 KB).
// TAN: the forward-mode TANGENT stream of the same layers as a second launch (fp32 mode has no room for two streams in one
// wave: 128 fp32 activations + 128 accumulators fill the register file): inputs are d/dt of the encodings, there is no bias,
// and the epilogue multiplies by sp'(z) taken from the VALUE launch's saved copies (`acts`, read) -- yd = sp'(z) (W ad) --
// and saves yd to `actsd`; the heads leave their tangent pre-activations in zsd4 / zod4.
template <bool SAVE, bool FULL, bool TAN = false, int MODE = 6>     // MODE 6 | 3 (float32_matmul_precision high: two pieces, three products)
__global__ __launch_bounds__(256, 1) void vfield_fwd6_kernel(FieldArgs a) {
    using PR = Pairs<MODE>;
    constexpr int NP = PR::NT;
    constexpr int STAGE = 2 * 8 * NP * 1024;             // two chunks x eight tiles
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    float *bias = reinterpret_cast<float *>(smem_all);
    unsigned char *smem_tf = smem_all + LB_FLOATS * 4;
    const int lane_k = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane16_k = lane_k * 16;
    for (int i = threadIdx.x; i < LB_FLOATS; i += 256) {
        float b = 0.f;
        if (i < 2048) b = a.P[l_boff(i >> 8, a.C) + (i & 255)];
        else if (i < LB_RGBH) b = a.P[l_boff(L_BOTT, a.C) + i - LB_BOTT];
        else if (i < LB_SIGMA) b = a.P[l_boff(L_RGBH, a.C) + i - LB_RGBH];
        else if (i == LB_SIGMA) b = a.P[l_boff(L_SIGMA, a.C)];
        else if (i >= LB_RGBO && i < LB_RGBO + a.C) b = a.P[l_boff(L_RGBO, a.C) + i - LB_RGBO];
        bias[i] = b;
    }
    const int64_t n_blk = (a.n + 31) >> 5, n_grp = (n_blk + 3) / 4;
    const size_t sstride = (size_t)n_grp * 4 * 32 * 256;

    auto issue = [&](int l, int cg, int buf) {           // stage (layer l, chunks 2 cg, 2 cg + 1) -> LDS buffer buf
        const int nt = l_nt(l), pieces = min(2, l_nch(l) - 2 * cg) * nt * NP;
        uint32_t off = (uint32_t)(f_off(l) + 2 * cg * nt) * NP * 1024 + wave * 1024;
        unsigned char *dst = smem_tf + buf * STAGE + wave * 1024;
        for (int i = wave; i < pieces; i += 4) {
            asm volatile("" : "+s"(off));
            glds16(reinterpret_cast<const unsigned char *>(a.img) + off + lane16_k, dst);
            off += 4096; dst += 4096;
        }
    };
    int buf = 0;
    if ((int64_t)blockIdx.x < n_grp) issue(0, 0, 0);
    for (int64_t grp = blockIdx.x; grp < n_grp; grp += gridDim.x) {
        const int64_t blk_g = grp * 4 + wave;
        const bool more_grp = grp + gridDim.x < n_grp;
        f32x16 yprev[8];                                 // activated outputs of the previous layer (accumulator layout)
        auto layer = [&](auto shape, const int l, const int ln, const int slot) {
            using S = decltype(shape);
            constexpr int NH = S::NH, NE = S::NE, NT = S::NT, ACT = S::ACT, nch = NH + NE, NCG = (nch + 1) / 2;
            uint32_t lane16 = lane16_k;
            int64_t blk = blk_g;
            asm volatile("" : "+v"(lane16), "+s"(blk));
            const int lane = lane16 >> 4, hi = lane >> 5, sl = lane & 31;
            float ef[NE ? NE : 1][8];                    // encoding chunks of this block (fp32; split at use)
            if (NE) {
                const int64_t bc = blk < n_blk ? blk : n_blk - 1;
                const float *xp = (NE == 4 ? (TAN ? a.encd : a.enc) + (bc * 32 + sl) * a.ld_enc : (TAN ? a.viewd : a.view) + (bc * 32 + sl) * a.ld_view) + 8 * hi;
#pragma unroll
                for (int c = 0; c < NE; ++c) {
                    const float4 v0 = *reinterpret_cast<const float4 *>(xp + 16 * c), v1 = *reinterpret_cast<const float4 *>(xp + 16 * c + 4);
                    ef[c][0] = v0.x; ef[c][1] = v0.y; ef[c][2] = v0.z; ef[c][3] = v0.w; ef[c][4] = v1.x; ef[c][5] = v1.y; ef[c][6] = v1.z; ef[c][7] = v1.w;
                }
            }
            f32x16 acc[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int g = 0; g < 16; ++g) acc[t][g] = 0.f;
#pragma unroll
            for (int cg = 0; cg < NCG; ++cg) {
                __syncthreads();
                if (cg + 1 < NCG) issue(l, cg + 1, buf ^ 1);
                else if (ln >= 0) issue(ln, 0, buf ^ 1);
                const unsigned char *st = smem_tf + buf * STAGE + lane16;
#pragma unroll
                for (int ci = 0; ci < 2; ++ci) {
                    const int c = 2 * cg + ci;
                    if (c >= nch) continue;
                    float xs[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) xs[j] = c < NH ? yprev[(c >> 1) & 7][8 * (c & 1) + j] : ef[(c - NH) % (NE ? NE : 1)][j];
                    bf16x8 xp[3];
                    split8<NP>(xs, xp);
#pragma unroll
                    for (int t0 = 0; t0 < NT; t0 += 2) { // two tiles at a time: consecutive MFMAs alternate accumulators
                        bf16x8 w[2][NP];
#pragma unroll
                        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                            for (int p = 0; p < NP; ++p)
                                if (t0 + tt < NT) w[tt][p] = *reinterpret_cast<const bf16x8 *>(st + ((ci * NT + t0 + tt) * NP + p) * 1024);
#pragma unroll
                        for (int k = 0; k < PR::N; ++k) {
                            acc[t0] = MFMAB(w[0][PR::W[k]], xp[PR::A[k]], acc[t0]);
                            if (t0 + 1 < NT) acc[t0 + 1] = MFMAB(w[1][PR::W[k]], xp[PR::A[k]], acc[t0 + 1]);
                        }
                    }
                }
                buf ^= 1;
                TRUNK_FENCE();
            }
            const float *bl = bias + lb_off(l);
            if (ACT == A_SIGMA) {
                const int64_t row = blk * 32 + sl;
                if (TAN) { if (hi == 0 && row < a.n) a.zsd4[row * 4] = acc[0][0]; return; }
                if (hi == 0 && row < a.n) a.sigma[row] = a.sel[row] ? __expf(acc[0][0] + bl[0] - 1.f) : 0.f;
                return;
            }
            if (ACT == A_RGB) {
                const int64_t row = blk * 32 + sl;
                if (TAN) {
                    if (hi == 0 && row < a.n)
                        *reinterpret_cast<float4 *>(a.zod4 + row * 4) = make_float4(acc[0][0], a.C > 1 ? acc[0][1] : 0.f, a.C > 2 ? acc[0][2] : 0.f,
                                                                                   a.C > 3 ? acc[0][3] : 0.f);
                    return;
                }
                if (hi == 0 && row < a.n) {
                    float r[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) r[c] = c < a.C ? softplus1(acc[0][c] + bl[c]) : 0.f;
                    *reinterpret_cast<float4 *>(a.rgb4 + row * 4) = make_float4(r[0], r[1], r[2], r[3]);
                }
                return;
            }
            float *sv = reinterpret_cast<float *>(TAN ? a.actsd : a.acts) + (size_t)slot * sstride;
            const float *hv = reinterpret_cast<const float *>(a.acts) + (size_t)slot * sstride;   // TAN: the value launch's outputs
            constexpr int PDH = 2;
            float4 hq[PDH][4];
            auto load_hv = [&](int t, float4 (&dst)[4]) {
                const float4 *hp = reinterpret_cast<const float4 *>(hv + ((blk * 8 + t) * 4 * 64 + lane) * 4);
#pragma unroll
                for (int q = 0; q < 4; ++q) dst[q] = hp[q * 64];
            };
            if (TAN && ACT == A_SP100) {
#pragma unroll
                for (int i = 0; i < PDH && i < NT; ++i) load_hv(i, hq[i]);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                float y[16];
                if (TAN) {
                    if (ACT == A_SP100) {
                        float4 hc[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) hc[q] = hq[t % PDH][q];
                        if (t + PDH < NT) load_hv(t + PDH, hq[t % PDH]);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float4 h4 = hc[q];
                            y[4 * q] = acc[t][4 * q] * dsoftplus_from_out(h4.x, 100.f);
                            y[4 * q + 1] = acc[t][4 * q + 1] * dsoftplus_from_out(h4.y, 100.f);
                            y[4 * q + 2] = acc[t][4 * q + 2] * dsoftplus_from_out(h4.z, 100.f);
                            y[4 * q + 3] = acc[t][4 * q + 3] * dsoftplus_from_out(h4.w, 100.f);
                        }
                    } else {
#pragma unroll
                        for (int g = 0; g < 16; ++g) y[g] = acc[t][g];
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 b4 = *reinterpret_cast<const float4 *>(bl + t * 32 + 8 * q + 4 * hi);
                        const float z[4] = {acc[t][4 * q] + b4.x, acc[t][4 * q + 1] + b4.y, acc[t][4 * q + 2] + b4.z, acc[t][4 * q + 3] + b4.w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) y[4 * q + j] = ACT == A_SP100 ? softplus100(z[j]) : z[j];
                    }
                }
#pragma unroll
                for (int g = 0; g < 16; ++g) yprev[t][g] = y[g];
                if (SAVE) {
                    float *p = sv + (((blk * 8 + t) * 4) * 64 + lane) * 4;
#pragma unroll
                    for (int q = 0; q < 4; ++q) *reinterpret_cast<float4 *>(p + q * 256) = make_float4(y[4 * q], y[4 * q + 1], y[4 * q + 2], y[4 * q + 3]);
                }
                TRUNK_FENCE();
            }
        };
        layer(Shape<0, 4, 8, A_SP100>(), 0, 1, 0);
        for (int l = 1; l < L_SKIP; ++l) layer(Shape<16, 0, 8, A_SP100>(), l, l + 1, l);
        layer(Shape<16, 4, 8, A_SP100>(), L_SKIP, 6, L_SKIP);
        layer(Shape<16, 0, 8, A_SP100>(), 6, 7, 6);
        layer(Shape<16, 0, 8, A_SP100>(), 7, L_SIGMA, 7);
        layer(Shape<16, 0, 1, A_SIGMA>(), L_SIGMA, FULL ? L_BOTT : (more_grp ? 0 : -1), -1);
        if (FULL) {
            layer(Shape<16, 0, 8, A_NONE>(), L_BOTT, L_RGBH, 8);
            layer(Shape<16, 2, 4, A_SP100>(), L_RGBH, L_RGBO, 9);
            layer(Shape<8, 0, 1, A_RGB>(), L_RGBO, more_grp ? 0 : -1, -1);
        }
    }
}

// CPL: the reverse pass of value + tangent (fp32 mode) as two launches of this kernel.  The tangent-side gradients are a chain
// of their own -- d loss / d zd_(l-1) = (W^T dzd_l) sp'(z_(l-1)), i.e. this kernel as it is -- and couple INTO the value side:
// dz_(l-1) = (W^T dz_l) s + (W^T dzd_l) yd_(l-1) beta (1 - s).  CPL 1 (run first, on the tangent-side inputs): also writes
// that second term, from its own accumulators and the saved tangent `actsd`, to `cpl`; CPL 2 (the value side): adds it.
template <int CPL = 0, int MODE = 6>
__global__ __launch_bounds__(256, 1) void vfield_bwd6_kernel(FieldArgs a) {
    using PR = Pairs<MODE>;
    constexpr int NP = PR::NT;
    constexpr int STAGE = 2 * 8 * NP * 1024;
    constexpr int PD = 2;                                // saved-activation tiles in flight (16 registers each)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_tb[];
    const int lane_k = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane16_k = lane_k * 16;
    const int64_t n_blk = (a.n + 31) >> 5, n_grp = (n_blk + 3) / 4;
    const size_t sstride = (size_t)n_grp * 4 * 32 * 256;

    auto issue = [&](int l, int cg, int buf) {
        const int nt = b_nt(l), pieces = min(2, b_nch(l) - 2 * cg) * nt * NP;
        uint32_t off = (uint32_t)(b_off(l) + 2 * cg * nt) * NP * 1024 + wave * 1024;
        unsigned char *dst = smem_tb + buf * STAGE + wave * 1024;
        for (int i = wave; i < pieces; i += 4) {
            asm volatile("" : "+s"(off));
            glds16(reinterpret_cast<const unsigned char *>(a.img) + off + lane16_k, dst);
            off += 4096; dst += 4096;
        }
    };
    int buf = 0;
    if ((int64_t)blockIdx.x < n_grp) issue(L_RGBO, 0, 0);
    for (int64_t grp = blockIdx.x; grp < n_grp; grp += gridDim.x) {
        const int64_t blk_g = grp * 4 + wave;
        const bool more_grp = grp + gridDim.x < n_grp;
        f32x16 yprev[8];                                 // pre-activation gradient of the layer above (accumulator layout)
        auto step = [&](auto shape, const int l, const int ln, const int hslot, const int dslot, const float *extra) {
            using S = decltype(shape);
            constexpr int NH = S::NH, NE = S::NE, NT = S::NT, nch = NH + NE, NCG = (nch + 1) / 2;
            constexpr bool DERIV = S::DERIV;
            uint32_t lane16 = lane16_k;
            int64_t blk = blk_g;
            asm volatile("" : "+v"(lane16), "+s"(blk));
            const int lane = lane16 >> 4, hi = lane >> 5, sl = lane & 31;
            const float *hs = reinterpret_cast<const float *>(a.acts) + (size_t)(DERIV ? hslot : 0) * (a.acts_sstride ? (size_t)a.acts_sstride : sstride);
            float ef[8];                                 // the extra operand chunk (dz_rgb columns / dz_sigma in k-slot 0)
            if (NE) {
                const int64_t bc = blk < n_blk ? blk : n_blk - 1;
                const float *xp = extra + (bc * 32 + sl) * 32 + 8 * hi;
                if (l == L_BOTT) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) ef[j] = 0.f;
                    ef[0] = hi == 0 ? xp[0] : 0.f;
                } else {
                    const float4 v0 = *reinterpret_cast<const float4 *>(xp), v1 = *reinterpret_cast<const float4 *>(xp + 4);
                    ef[0] = v0.x; ef[1] = v0.y; ef[2] = v0.z; ef[3] = v0.w; ef[4] = v1.x; ef[5] = v1.y; ef[6] = v1.z; ef[7] = v1.w;
                }
            }
            float4 hpre[PD][4], xpre[PD][4];                 // xpre: the saved tangent (CPL 1) / the coupling term (CPL 2) of the same tile
            const float *xs_ = CPL == 1 ? reinterpret_cast<const float *>(a.actsd) + (size_t)(DERIV ? hslot : 0) * (a.acts_sstride ? (size_t)a.acts_sstride : sstride)
                                        : reinterpret_cast<const float *>(a.cpl) + (size_t)dslot * sstride;
            auto load_h = [&](int t, float4 (&dst)[4], float4 (&dstx)[4]) {
                const int64_t tile = ((blk * 8 + t) * 4 * 64 + lane) * 4;
                const float4 *p = reinterpret_cast<const float4 *>(hs + tile);
#pragma unroll
                for (int q = 0; q < 4; ++q) dst[q] = p[q * 64];
                if (CPL) {
                    const float4 *px = reinterpret_cast<const float4 *>(xs_ + tile);
#pragma unroll
                    for (int q = 0; q < 4; ++q) dstx[q] = px[q * 64];
                }
            };
            f32x16 acc[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int g = 0; g < 16; ++g) acc[t][g] = 0.f;
#pragma unroll
            for (int cg = 0; cg < NCG; ++cg) {
                __syncthreads();
                if (cg + 1 < NCG) issue(l, cg + 1, buf ^ 1);
                else if (ln >= 0) issue(ln, 0, buf ^ 1);
                if (DERIV && cg == NCG - 1) {
#pragma unroll
                    for (int i = 0; i < PD; ++i) load_h(i, hpre[i], xpre[i]);
                }
                const unsigned char *st = smem_tb + buf * STAGE + lane16;
#pragma unroll
                for (int ci = 0; ci < 2; ++ci) {
                    const int c = 2 * cg + ci;
                    if (c >= nch) continue;
                    float xs[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) xs[j] = c < NH ? yprev[(c >> 1) & 7][8 * (c & 1) + j] : ef[j];
                    bf16x8 xp[3];
                    split8<NP>(xs, xp);
#pragma unroll
                    for (int t0 = 0; t0 < NT; t0 += 2) {
                        bf16x8 w[2][NP];
#pragma unroll
                        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                            for (int p = 0; p < NP; ++p)
                                if (t0 + tt < NT) w[tt][p] = *reinterpret_cast<const bf16x8 *>(st + ((ci * NT + t0 + tt) * NP + p) * 1024);
#pragma unroll
                        for (int k = 0; k < PR::N; ++k) {
                            acc[t0] = MFMAB(w[0][PR::W[k]], xp[PR::A[k]], acc[t0]);
                            if (t0 + 1 < NT) acc[t0 + 1] = MFMAB(w[1][PR::W[k]], xp[PR::A[k]], acc[t0 + 1]);
                        }
                    }
                }
                buf ^= 1;
                TRUNK_FENCE();
            }
            float *sv = reinterpret_cast<float *>(a.dz) + (size_t)dslot * sstride;
            float *cv = reinterpret_cast<float *>(a.cpl) + (size_t)dslot * sstride;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                float y[16];
                if (DERIV) {
                    float h[16];
#pragma unroll
                    for (int q = 0; q < 4; ++q) { h[4 * q] = hpre[t % PD][q].x; h[4 * q + 1] = hpre[t % PD][q].y; h[4 * q + 2] = hpre[t % PD][q].z; h[4 * q + 3] = hpre[t % PD][q].w; }
                    float4 xc[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) xc[q] = xpre[t % PD][q];
                    if (t + PD < NT) load_h(t + PD, hpre[t % PD], xpre[t % PD]);
                    const int64_t tile = ((blk * 8 + t) * 4 * 64 + lane) * 4;
                    if (CPL == 1) {                              // coupling term from the raw accumulators and the saved tangent
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float4 yd = xc[q];
                            const float ydv[4] = {yd.x, yd.y, yd.z, yd.w};
                            float c4[4];
#pragma unroll
                            for (int j = 0; j < 4; ++j) c4[j] = acc[t][4 * q + j] * ydv[j] * (100.f * (1.f - dsoftplus_from_out(h[4 * q + j], 100.f)));
                            *reinterpret_cast<float4 *>(cv + tile + q * 256) = make_float4(c4[0], c4[1], c4[2], c4[3]);
                        }
                    }
#pragma unroll
                    for (int g = 0; g < 16; ++g) y[g] = acc[t][g] * dsoftplus_from_out(h[g], 100.f);
                    if (CPL == 2) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float4 c4 = xc[q];
                            y[4 * q] += c4.x; y[4 * q + 1] += c4.y; y[4 * q + 2] += c4.z; y[4 * q + 3] += c4.w;
                        }
                    }
                } else {
#pragma unroll
                    for (int g = 0; g < 16; ++g) y[g] = acc[t][g];
                }
#pragma unroll
                for (int g = 0; g < 16; ++g) yprev[t][g] = y[g];
                float *p = sv + (((blk * 8 + t) * 4) * 64 + lane) * 4;
#pragma unroll
                for (int q = 0; q < 4; ++q) *reinterpret_cast<float4 *>(p + q * 256) = make_float4(y[4 * q], y[4 * q + 1], y[4 * q + 2], y[4 * q + 3]);
                TRUNK_FENCE();
            }
        };
        step(BShape<0, 1, 4, true>(), L_RGBO, L_RGBH, 9, 9, a.dz_rgb);
        step(BShape<8, 0, 8, false>(), L_RGBH, L_BOTT, 0, 8, nullptr);
        step(BShape<16, 1, 8, true>(), L_BOTT, 7, 7, 7, a.dz_sig);
        for (int l = 7; l >= 1; --l) step(BShape<16, 0, 8, true>(), l, l > 1 ? l - 1 : (more_grp ? L_RGBO : -1), l - 1, l - 1, nullptr);
    }
}

// ---- weight / bias gradient of one layer from the fragment-layout copies -------------------------------------------------------
// dW = dz^T x with the SAMPLES as the reduction: both MFMA operands need, per lane = feature, eight consecutive samples --
// the transposes of the saved [sample][feature] data.  A wave transposes the 32 x 32 tile it loaded on the matrix cores:
// the saved fragment (lane = sample, k-slots = 16 features) is exactly an A operand, so T = X . Sel with a 0/1 selection
// matrix Sel as B operand is the tile with lane = feature and the 32 samples in the accumulator registers (two MFMAs per tile
// and piece, exact: every product is a bf16 value times 1), which packs into the two k-chunk operands of the gradient
// product.  The operands go to LDS as ready-made 1 KB fragments (one ds_write_b128 per lane; a first version scattered 2-byte
// LDS writes to transpose, which cost more than the gradient MFMAs of a stage).  One workgroup of 8 waves per sample split,
// wave w owns the output rows 32 w .. against up to 10 input tiles; partial sums go to per-split slabs.
// Narrow layers (sigma: 1 output, colour output: C) take dz from the row-major [n_pad][32] buffers; the first layer, the skip
// layer and the colour hidden layer take (part of) their input from the row-major encodings.
#ifndef DW_FLIP
#define DW_FLIP 4                                       // stages between sign changes of the dW accumulators (see vfield_dw_kernel)
#endif
struct FieldDwArgs {
    const void *dz;                                      // fragment slot (256-feature stride), or ..
    const float *dz_rows;                                // .. row-major [n_pad][32]
    const void *x; int nx;                               // fragment slot with nx valid features
    const float *x_rows; int ld_rows, n_rows;            // row-major encoding columns (n_rows 64 / 32)
    int N, K;                                            // torch out / in features
    int k_base, skip_bias;                               // first input column of this launch; 1: weights only (second launch of a layer)
    int64_t n;
    float *slab_w, *slab_b;                              // per split (stride slab_stride floats): [N][K] then [N], so that ONE
    int64_t slab_stride;                                 // reduction adds a layer's dW and db (contiguous in the parameter block)
};

// ZROWS / XFRAG / XROWS compile-time: the loads of a stage are unconditional straight-line code, so the compiler counts them
// (`s_waitcnt vmcnt(N)`) and a whole stage stays in flight behind the one being worked on -- with a run-time `if` around them
// every use waited for vmcnt(0), i.e. for the loads issued last (3.3 TB/s, latency-bound).
template <int MODE, bool ZROWS, bool XFRAG, bool XROWS>
__global__ __launch_bounds__(512, 1) void vfield_dw_kernel(FieldDwArgs a) {
    typedef typename TC<MODE>::ST ST;
    constexpr int NP = TC<MODE>::NP;
    constexpr int NV = MODE == 1 ? 2 : 4;                // 16-byte pieces per lane of one 32-feature tile
    constexpr int NXT = XFRAG && XROWS ? 10 : 8;         // input tiles held in LDS: the slot's, then the encoding's
    constexpr int XR0 = XFRAG ? 8 : 0;                   // first encoding tile
    constexpr int RD = ZROWS ? (MODE == 1 ? 3 : 2) : 1;           // stages of loads in flight: narrow layers do little per stage (wide ones: 2 measured slower)
    constexpr int BUFBYTES = NXT * 2 * NP * 1024;        // one LDS buffer: the input operand fragments (dz fragments stay in registers)
    using PRD = Pairs<MODE>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_td[];
    const int wave_k = threadIdx.x >> 6;
    const int n_splits = gridDim.x;
    const int64_t n_blk = (a.n + 31) >> 5;
    const bool has_tile = wave_k * 32 < a.N;
    float *sw = a.slab_w + (int64_t)blockIdx.x * a.slab_stride, *sb = a.slab_b + (int64_t)blockIdx.x * a.slab_stride;
    const int kt_frag = XFRAG ? a.nx / 32 : 0, kt_rows = XROWS ? a.n_rows / 32 : 0;
    const int nzt = a.N >= 256 ? 8 : a.N > 64 ? 4 : 8, nxt_ = kt_frag == 4 ? 4 : 8;     // valid 32-feature tiles of the dz / x slots

    f32x16 acc[NXT];
#pragma unroll
    for (int t = 0; t < NXT; ++t)
#pragma unroll
        for (int g = 0; g < 16; ++g) acc[t][g] = 0.f;
    float bsum = 0.f;                                    // lane = feature (transposed layout): one partial per lane
    float bs4[4] = {0.f, 0.f, 0.f, 0.f};                 // ZROWS: lane = sample, columns 0 .. 3
    struct Stage { uint4 pz[NV], px[NV]; float4 rz[4], rx[4]; };
    Stage sq[RD];
    bf16x8 azr[2][3];                                    // this wave's dz tile, transposed: [k-chunk][piece]
    // `tid`: an opaque copy of threadIdx.x (see `stage`): the addresses derived from it are computed per stage instead of once
    // per kernel (where they spilled)
    auto fetch = [&](int tid, int64_t blk, Stage &q) {
        const int lane = tid & 63, wave = tid >> 6, hi = lane >> 5, sl = lane & 31;
        const uint4 *zb = reinterpret_cast<const uint4 *>(reinterpret_cast<const ST *>(a.dz) + blk * 32 * 256);
        const uint4 *xb = reinterpret_cast<const uint4 *>(reinterpret_cast<const ST *>(a.x) + blk * 32 * 256);
#pragma unroll
        for (int v = 0; v < NV; ++v) {                   // wave w: tile w of dz and of the slot part of x
            // (128-feature slots: the upper waves re-read the lower tiles -- cache hits instead of HBM reads of unused pieces)
            if (!ZROWS) q.pz[v] = zb[((wave & (nzt - 1)) * NV + v) * 64 + lane];
            if (XFRAG) q.px[v] = xb[((wave & (nxt_ - 1)) * NV + v) * 64 + lane];
        }
        // row-major sources: lane = sample, 2 x 8 consecutive columns per 16-column chunk of the wave's 32-column tile
        if (XROWS) {
            const float *p = a.x_rows + (blk * 32 + sl) * a.ld_rows + ((wave * 32) & (a.n_rows - 1)) + 8 * hi;
            q.rx[0] = *reinterpret_cast<const float4 *>(p); q.rx[1] = *reinterpret_cast<const float4 *>(p + 4);
            q.rx[2] = *reinterpret_cast<const float4 *>(p + 16); q.rx[3] = *reinterpret_cast<const float4 *>(p + 20);
        }
        if (ZROWS) {
            const float *p = a.dz_rows + (blk * 32 + sl) * 32 + 8 * hi;
            q.rz[0] = *reinterpret_cast<const float4 *>(p); q.rz[1] = *reinterpret_cast<const float4 *>(p + 4);
            q.rz[2] = *reinterpret_cast<const float4 *>(p + 16); q.rz[3] = *reinterpret_cast<const float4 *>(p + 20);
        }
    };
    // the 32 x 32 tile whose two 16-feature chunks are `c0`, `c1` (lane = sample; kmap order: KM, natural order: !KM),
    // transposed, as the two k-chunk operands (k = samples) at dst[ks][piece]
    // `sgn` (+-1): the selection matrix carries the sign, so the transposed tile comes out negated for free (dz tiles of the
    // stages that accumulate into the negated accumulator, see the stage loop)
    auto transpose_store = [&](const float (&c0)[8], const float (&c1)[8], bool km, int lane, unsigned char *dst, float *colsum,
                               bf16x8 (*regs)[3], float sgn) {
        const int hi = lane >> 5, sl = lane & 31;
        bf16x8 sel[2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int f = km ? 16 * u + 8 * (j >> 2) + 4 * hi + (j & 3) : 16 * u + 8 * hi + j;
                sel[u][j] = (__bf16)(sl == f ? sgn : 0.f);
            }
        bf16x8 p0[3], p1[3];
        split8<NP>(c0, p0);
        split8<NP>(c1, p1);
        float cs = 0.f;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            f32x16 T;
#pragma unroll
            for (int g = 0; g < 16; ++g) T[g] = 0.f;
            T = MFMAB(p0[p], sel[0], T);
            T = MFMAB(p1[p], sel[1], T);
            bf16x8 o0, o1;
#pragma unroll
            for (int j = 0; j < 8; ++j) { o0[j] = (__bf16)T[j]; o1[j] = (__bf16)T[8 + j]; cs += T[j] + T[8 + j]; }
            if (regs) { regs[0][p] = o0; regs[1][p] = o1; }
            else {
                *reinterpret_cast<bf16x8 *>(dst + (0 * NP + p) * 1024 + lane * 16) = o0;
                *reinterpret_cast<bf16x8 *>(dst + (1 * NP + p) * 1024 + lane * 16) = o1;
            }
        }
        if (colsum) *colsum += sgn * cs;
    };
    auto stash = [&](int tid, int64_t blk, const Stage &q, unsigned char *buf, float zsgn) {
        const int lane = tid & 63, wave = tid >> 6, hi = lane >> 5, sl = lane & 31;
        const bool live = blk * 32 + sl < a.n;           // lane = sample here: samples past the end contribute nothing
        unsigned char *xf = buf;
        auto unpack = [&](const uint4 (&pp)[NV], float (&c0)[8], float (&c1)[8]) {
            if (MODE == 1) {                             // the two 1 KB pieces are the tile's chunks
                const bf16x8 a0 = *reinterpret_cast<const bf16x8 *>(&pp[0]), a1 = *reinterpret_cast<const bf16x8 *>(&pp[NV > 1 ? 1 : 0]);
#pragma unroll
                for (int j = 0; j < 8; ++j) { c0[j] = live ? (float)a0[j] : 0.f; c1[j] = live ? (float)a1[j] : 0.f; }
            } else {                                     // pieces q = 0 .. 3: chunk u = q >> 1, k-slots 4 (q & 1) ..
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const float4 v = *reinterpret_cast<const float4 *>(&pp[qq % NV]);
                    float *c = qq < 2 ? c0 : c1;
                    c[4 * (qq & 1)] = live ? v.x : 0.f; c[4 * (qq & 1) + 1] = live ? v.y : 0.f;
                    c[4 * (qq & 1) + 2] = live ? v.z : 0.f; c[4 * (qq & 1) + 3] = live ? v.w : 0.f;
                }
            }
        };
        auto unrows = [&](const float4 (&r)[4], float (&c0)[8], float (&c1)[8]) {
            c0[0] = r[0].x; c0[1] = r[0].y; c0[2] = r[0].z; c0[3] = r[0].w; c0[4] = r[1].x; c0[5] = r[1].y; c0[6] = r[1].z; c0[7] = r[1].w;
            c1[0] = r[2].x; c1[1] = r[2].y; c1[2] = r[2].z; c1[3] = r[2].w; c1[4] = r[3].x; c1[5] = r[3].y; c1[6] = r[3].z; c1[7] = r[3].w;
#pragma unroll
            for (int j = 0; j < 8; ++j) { c0[j] = live ? c0[j] : 0.f; c1[j] = live ? c1[j] : 0.f; }
        };
        float c0[8], c1[8];
        if (!ZROWS) {
            unpack(q.pz, c0, c1);
            transpose_store(c0, c1, true, lane, nullptr, &bsum, azr, zsgn);      // wave w's dz tile is used by wave w only
        } else if (wave == 0) {
            unrows(q.rz, c0, c1);
            transpose_store(c0, c1, false, lane, nullptr, nullptr, azr, zsgn);
            if (hi == 0) {                               // bias of a narrow layer: fp32 sums of the rows' own values (columns 0 .. 3)
#pragma unroll
                for (int j = 0; j < 4; ++j) bs4[j] += c0[j];
            }
        }
        if (XFRAG) {
            unpack(q.px, c0, c1);
            transpose_store(c0, c1, true, lane, xf + wave * 2 * NP * 1024, nullptr, nullptr, 1.f);
        }
        if (XROWS && wave < kt_rows) {
            unrows(q.rx, c0, c1);
            transpose_store(c0, c1, false, lane, xf + (XR0 + wave) * 2 * NP * 1024, nullptr, nullptr, 1.f);
        }
    };
    auto products = [&](int tid, const unsigned char *buf) {
        const int lane = tid & 63, wave = tid >> 6;
        const unsigned char *xf = buf + lane * 16;
        (void)wave;
        if (has_tile) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 az[NP];
#pragma unroll
                for (int p = 0; p < NP; ++p) az[p] = azr[ks][p];
#pragma unroll
                for (int t = 0; t < NXT; t += 2)         // two input tiles at a time: consecutive MFMAs alternate accumulators
                    if (t >= XR0 ? t - XR0 < kt_rows : t < kt_frag) {
                        bf16x8 bx[2][NP];
#pragma unroll
                        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                            for (int p = 0; p < NP; ++p)
                                bx[tt][p] = *reinterpret_cast<const bf16x8 *>(xf + (((t + tt) * 2 + ks) * NP + p) * 1024);
#pragma unroll
                        for (int k = 0; k < PRD::N; ++k) {
                            acc[t] = MFMAB(az[PRD::W[k]], bx[0][PRD::A[k]], acc[t]);
                            acc[t + 1] = MFMAB(az[PRD::W[k]], bx[1][PRD::A[k]], acc[t + 1]);       // (odd tile counts: never stored)
                        }
                    }
            }
        }
    };
    const int64_t b0 = blockIdx.x;
    float fsgn = 1.f;                                    // sign the accumulators carry at the end
    {
        // products of stage i (this wave's dz fragments in registers, the input fragments from LDS buffer i & 1), then stage
        // i + 1 is transposed into the other buffer: one barrier a stage.  RD stages of loads are in flight (narrow layers do
        // so little per stage that one stage of lead time is shorter than the HBM latency).  Everything in the loop is
        // unconditional -- blocks past the end are clamped loads whose samples count as not live -- so that the compiler's
        // load counters stay exact (counted vmcnt waits).
        // Rounding bias of the MFMA accumulate (toward -inf whatever the sign, tools/mlp_bias_probe.py; it grows like the
        // number of accumulations: 4e-5 of max |dW| at n = 1 M against 4e-6 at 10 k): every DW_FLIP stages the accumulators
        // AND the dz operand change sign, so the running sum is held alternately as +S and -S and the bias of one period
        // cancels the next one's -- the cure of ren_mlp_x.hip without a second accumulator set.
        const int64_t S = n_splits, n_it = b0 < n_blk ? (n_blk - b0 + S - 1) / S : 0, n_it_pad = (n_it + RD - 1) / RD * RD;
        auto clampb = [&](int64_t blk) { return blk < n_blk ? blk : n_blk - 1; };
        auto sgn_of = [](int64_t stage) { return ((stage / DW_FLIP) & 1) ? -1.f : 1.f; };
#pragma unroll
        for (int k = 0; k < RD; ++k) fetch(threadIdx.x, clampb(b0 + k * S), sq[k]);
        {
            int tid = threadIdx.x;
            asm volatile("" : "+v"(tid));
            stash(tid, b0, sq[0], smem_td, 1.f);
            fetch(tid, clampb(b0 + RD * S), sq[0]);
        }
        __syncthreads();
        for (int64_t i = 0; i < n_it_pad; i += RD) {
#pragma unroll
            for (int k = 0; k < RD; ++k) {
                int tid = threadIdx.x;
                asm volatile("" : "+v"(tid));
                const int64_t ii = i + k, nxt = b0 + (ii + 1) * S;               // (nominal: may be past the end)
                if (ii > 0 && ii % DW_FLIP == 0) {       // wave-uniform
#pragma unroll
                    for (int t = 0; t < NXT; ++t)
#pragma unroll
                        for (int g = 0; g < 16; ++g) acc[t][g] = -acc[t][g];
                }
                products(tid, smem_td + (ii & 1) * BUFBYTES);
                stash(tid, nxt, sq[(k + 1) % RD], smem_td + ((ii + 1) & 1) * BUFBYTES, sgn_of(ii + 1));
                fetch(tid, clampb(nxt + RD * S), sq[(k + 1) % RD]);
                __syncthreads();
            }
        }
        if (n_it_pad > 0) fsgn = sgn_of(n_it_pad - 1);
    }
    const int lane = threadIdx.x & 63, hi = lane >> 5, sl = lane & 31;
    if (has_tile) {
#pragma unroll
        for (int t = 0; t < NXT; ++t) {
            // tile t -> input column: the slot part first, the encoding part after it
            const int k = a.k_base + (t >= XR0 ? kt_frag * 32 + (t - XR0) * 32 + sl : t * 32 + sl);
            const bool valid = t >= XR0 ? t - XR0 < kt_rows : t < kt_frag;
            if (!valid || k >= a.K) continue;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int o = wave_k * 32 + rowc(g) + 4 * hi;
                if (o < a.N) sw[(int64_t)o * a.K + k] = fsgn * acc[t][g];
            }
        }
    }
    // bias gradient: lane (feature sl of the wave's dz tile, hi) summed 16 of the 32 samples of every stage
    if (ZROWS) {
        if (wave_k == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = bs4[j];
#pragma unroll
                for (int off = 1; off < 32; off <<= 1) v += __shfl_xor(v, off, 64);
                if (lane == 0 && j < a.N) sb[j] = v;
            }
        }
    } else {
        bsum += __shfl_xor(bsum, 32, 64);
        if (hi == 0 && !a.skip_bias) {
            const int f = wave_k * 32 + sl;
            if (f < a.N) sb[f] = bsum;
        }
    }
}

template <int MODE> size_t fwd_lds() { return LB_FLOATS * 4 + 2 * (size_t)TC<MODE>::NTS * 20 * TC<MODE>::NP * 1024; }
static size_t fwd6_lds(int np = 3) { return LB_FLOATS * 4 + 2 * (size_t)2 * 8 * np * 1024; }
static size_t bwd6_lds(int np = 3) { return 2 * (size_t)2 * 8 * np * 1024; }
template <int MODE> size_t bwd_lds() { return 2 * (size_t)TC<MODE>::NTS * 17 * TC<MODE>::NP * 1024; }

}  // namespace

static inline int vfield_np(int mode) { return mode == 1 ? 1 : mode == 3 ? 2 : 3; }
static inline bool vfield_mode_ok(int mode) { return mode == 1 || mode == 3 || mode == 6; }
static inline bool vfield_ok(int mode, int C) { return vfield_mode_ok(mode) && C >= 1 && C <= 4; }

extern "C" int64_t ren_vanilla_image_bytes(int32_t mode) {
    if (!vfield_mode_ok(mode)) return -1;
    return (int64_t)(F_FRAGS + B_FRAGS) * vfield_np(mode) * 1024;
}

extern "C" int64_t ren_vanilla_saved_bytes(int32_t mode, int64_t n) {
    if (!vfield_mode_ok(mode) || n < 0) return -1;
    const int64_t per_grp = 4 * (mode == 1 ? TC<1>::NB : TC<6>::NB), n_grp = ((n + 31) / 32 + per_grp - 1) / per_grp;
    return N_SLOTS * n_grp * per_grp * 32 * 256 * (mode == 1 ? 2 : 4);          // whole workgroup passes of 32-sample blocks
}

extern "C" int ren_vanilla_prep(const float *params, int32_t C, int32_t mode, void *image, void *stream) {
    if (!params || !image || !vfield_ok(mode, C)) return REN_ERR_BAD_ARG;
    const int np = vfield_np(mode);
    __bf16 *f = reinterpret_cast<__bf16 *>(image), *b = f + (size_t)F_FRAGS * np * 512;
    const int blocks = ((F_FRAGS + B_FRAGS) * 64 + 255) / 256;
    if (mode == 1) hipLaunchKernelGGL(vfield_prep_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, params, C, f, b);
    else if (mode == 3) hipLaunchKernelGGL(vfield_prep_kernel<2>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, params, C, f, b);
    else hipLaunchKernelGGL(vfield_prep_kernel<3>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, params, C, f, b);
    REN_CHECK_LAUNCH();
}

static int vfield_grid(int64_t n, int nb) {
    const int64_t n_grp = ((n + 31) / 32 + 4 * nb - 1) / (4 * nb);
    return (int)(n_grp < 256 ? n_grp : 256);
}

extern "C" int ren_vanilla_fwd(const float *enc, int32_t ld_enc, const float *view, int32_t ld_view, const uint8_t *selector,
                               const float *params, int32_t C, int32_t activations, const void *image, int32_t mode, int64_t n, void *saved,
                               float *sigma, float *rgb4, void *stream) {
    if (!enc || !selector || !params || !image || !sigma || !vfield_ok(mode, C) || n < 0 || ld_enc < 64 || (ld_enc & 3)) return REN_ERR_BAD_ARG;
    if (rgb4 && (!view || ld_view < 32 || (ld_view & 3))) return REN_ERR_BAD_ARG;
    if (saved && !rgb4) return REN_ERR_BAD_ARG;                                 // the backward pass needs the whole field
    if (activations != 0) return REN_ERR_UNSUPPORTED;         // activation alternatives: the per-layer launches (ren_dense_*)
    if (n == 0) return REN_OK;
    FieldArgs a = {};
    a.enc = enc; a.ld_enc = ld_enc; a.view = view; a.ld_view = ld_view; a.sel = selector; a.P = params; a.C = C;
    a.img = reinterpret_cast<const __bf16 *>(image); a.acts = saved; a.sigma = sigma; a.rgb4 = rgb4; a.n = n;
    hipStream_t st = (hipStream_t)stream;
#define REN_VFIELD_FWD(MODE, SAVE, FULL)                                                                                        \
    do {                                                                                                                        \
        (void)hipFuncSetAttribute((const void *)vfield_fwd_kernel<MODE, SAVE, FULL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fwd_lds<MODE>()); \
        hipLaunchKernelGGL((vfield_fwd_kernel<MODE, SAVE, FULL>), dim3(vfield_grid(n, TC<MODE>::NB)), dim3(256), fwd_lds<MODE>(), st, a); \
    } while (0)
    if (mode == 1) {
#define REN_VFIELD_FWD1(SAVE, FULL)                                                                                             \
    do {                                                                                                                        \
        (void)hipFuncSetAttribute((const void *)vfield_fwd1_kernel<SAVE, FULL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fwd_lds<1>()); \
        hipLaunchKernelGGL((vfield_fwd1_kernel<SAVE, FULL>), dim3(vfield_grid(n, 2)), dim3(256), fwd_lds<1>(), st, a);          \
    } while (0)
        if (ren_knob(REN_KNOB_VFIELD_PLAIN)) { if (saved) REN_VFIELD_FWD(1, true, true); else if (rgb4) REN_VFIELD_FWD(1, false, true); else REN_VFIELD_FWD(1, false, false); }
        else if (saved) REN_VFIELD_FWD1(true, true); else if (rgb4) REN_VFIELD_FWD1(false, true); else REN_VFIELD_FWD1(false, false);
    }
    else {
#define REN_VFIELD_FWD6(SAVE, FULL, MODE)                                                                                       \
    do {                                                                                                                        \
        (void)hipFuncSetAttribute((const void *)vfield_fwd6_kernel<SAVE, FULL, false, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fwd6_lds(vfield_np(MODE))); \
        hipLaunchKernelGGL((vfield_fwd6_kernel<SAVE, FULL, false, MODE>), dim3(vfield_grid(n, 1)), dim3(256), fwd6_lds(vfield_np(MODE)), st, a); \
    } while (0)
        if (mode == 3) { if (saved) REN_VFIELD_FWD6(true, true, 3); else if (rgb4) REN_VFIELD_FWD6(false, true, 3); else REN_VFIELD_FWD6(false, false, 3); }
        else if (saved) REN_VFIELD_FWD6(true, true, 6); else if (rgb4) REN_VFIELD_FWD6(false, true, 6); else REN_VFIELD_FWD6(false, false, 6);
    }
    REN_CHECK_LAUNCH();
}

extern "C" int ren_vanilla_bwd(const float *dz_rgb, const float *dz_sigma, const void *image, int32_t mode, int32_t activations, int64_t n,
                               const void *saved, int64_t saved_slot_bytes, void *dz, void *stream) {
    if (!dz_rgb || !dz_sigma || !image || !saved || !dz || !vfield_mode_ok(mode) || n < 0 || saved_slot_bytes < 0) return REN_ERR_BAD_ARG;
    if (activations != 0) return REN_ERR_UNSUPPORTED;
    if (n == 0) return REN_OK;
    FieldArgs a = {};
    a.img = reinterpret_cast<const __bf16 *>(image) + (size_t)F_FRAGS * vfield_np(mode) * 512;
    a.acts = const_cast<void *>(saved); a.dz_rgb = dz_rgb; a.dz_sig = dz_sigma; a.dz = dz; a.n = n;
    a.acts_sstride = saved_slot_bytes / (mode == 1 ? 2 : 4);
    hipStream_t st = (hipStream_t)stream;
    if (mode == 1) {
        (void)hipFuncSetAttribute((const void *)vfield_bwd_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bwd_lds<1>());
        hipLaunchKernelGGL(vfield_bwd_kernel<1>, dim3(vfield_grid(n, TC<1>::NB)), dim3(256), bwd_lds<1>(), st, a);
    } else if (mode == 3) {
        (void)hipFuncSetAttribute((const void *)vfield_bwd6_kernel<0, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bwd6_lds(2));
        hipLaunchKernelGGL((vfield_bwd6_kernel<0, 3>), dim3(vfield_grid(n, 1)), dim3(256), bwd6_lds(2), st, a);
    } else {
        (void)hipFuncSetAttribute((const void *)vfield_bwd6_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bwd6_lds());
        hipLaunchKernelGGL(vfield_bwd6_kernel<0>, dim3(vfield_grid(n, 1)), dim3(256), bwd6_lds(), st, a);
    }
    REN_CHECK_LAUNCH();
}

extern "C" int64_t ren_vanilla_bwd_weight_workspace_floats(int32_t n_splits) {
    if (n_splits < 1) return -1;
    return (int64_t)n_splits * (256 * 319 + 256);
}

static int vanilla_bwd_weight_impl(const void *dz, const void *saved, int64_t saved_slot_bytes, const float *enc, int32_t ld_enc,
                                   const float *view, int32_t ld_view, const float *dz_rgb, const float *dz_sigma, int32_t C,
                                   int32_t mode, int64_t n, int32_t n_splits, float *grads, float *workspace, void *stream, bool no_bias) {
    if (!dz || !saved || !enc || !view || !dz_rgb || !dz_sigma || !grads || !workspace || !vfield_ok(mode, C) || n < 0 || n_splits < 1 ||
        ld_enc < 64 || (ld_enc & 3) || ld_view < 32 || (ld_view & 3))
        return REN_ERR_BAD_ARG;
    if (n == 0) return REN_OK;
    hipStream_t st = (hipStream_t)stream;
    const size_t sbytes = (size_t)ren_vanilla_saved_bytes(mode, n) / N_SLOTS;       // dz: n samples; saved: possibly a range of more
    const size_t sbytes_saved = saved_slot_bytes > 0 ? (size_t)saved_slot_bytes : sbytes;
    auto slot = [&](const void *base, int s) {
        return reinterpret_cast<const unsigned char *>(base) + s * (base == saved ? sbytes_saved : sbytes);
    };
    for (int l = NL - 1; l >= 0; --l) {
        FieldDwArgs a = {};
        a.N = l_out(l, C); a.K = l_in(l); a.n = n;
        if (l == L_RGBO) { a.dz_rows = dz_rgb; a.x = slot(saved, 9); a.nx = 128; }
        else if (l == L_RGBH) { a.dz = slot(dz, 9); a.x = slot(saved, 8); a.nx = 256; a.x_rows = view; a.ld_rows = ld_view; a.n_rows = 32; }
        else if (l == L_BOTT) { a.dz = slot(dz, 8); a.x = slot(saved, 7); a.nx = 256; }
        else if (l == L_SIGMA) { a.dz_rows = dz_sigma; a.x = slot(saved, 7); a.nx = 256; }
        else {
            a.dz = slot(dz, l);
            if (l > 0) { a.x = slot(saved, l - 1); a.nx = 256; }
            if (l == 0 || l == L_SKIP) { a.x_rows = enc; a.ld_rows = ld_enc; a.n_rows = 64; }
        }
        a.slab_stride = (int64_t)a.N * a.K + a.N; a.slab_w = workspace; a.slab_b = workspace + (int64_t)a.N * a.K;
        a.skip_bias = no_bias ? 1 : 0;                                          // the tangent stream has no bias (z' = W a')
        const bool zrows = a.dz_rows != nullptr, xfrag = a.x != nullptr, xrows = a.x_rows != nullptr;
#define REN_VFIELD_DW(MODE, ZR, XF, XR)                                                                                         \
    do {                                                                                                                        \
        const size_t lds = (size_t)2 * ((XF) && (XR) ? 10 : 8) * 2 * vfield_np(MODE) * 1024;                                  \
        (void)hipFuncSetAttribute((const void *)vfield_dw_kernel<MODE, ZR, XF, XR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((vfield_dw_kernel<MODE, ZR, XF, XR>), dim3(n_splits), dim3(512), lds, st, a);                         \
    } while (0)
#define REN_VFIELD_DW_MODE(MODE)                                                                                                \
    do {                                                                                                                        \
        if (zrows) REN_VFIELD_DW(MODE, true, true, false);                                                                      \
        else if (xfrag && xrows && MODE == 1) REN_VFIELD_DW(MODE, false, true, true);                                           \
        else if (xfrag && xrows) {             /* fp32 mode: 10 input tiles of accumulators do not fit: two launches */            \
            REN_VFIELD_DW(MODE, false, true, false);                                                                            \
            a.k_base = a.nx; a.skip_bias = 1;                                                                                   \
            REN_VFIELD_DW(MODE, false, false, true);                                                                            \
        }                                                                                                                       \
        else if (xfrag) REN_VFIELD_DW(MODE, false, true, false);                                                                \
        else REN_VFIELD_DW(MODE, false, false, true);                                                                           \
    } while (0)
        if (mode == 1) REN_VFIELD_DW_MODE(1); else if (mode == 3) REN_VFIELD_DW_MODE(3); else REN_VFIELD_DW_MODE(6);
        launch_reduce_slabs(a.slab_w, n_splits, a.N * a.K + (no_bias ? 0 : a.N), grads + l_woff(l, C), st, a.slab_stride);
    }
    REN_CHECK_LAUNCH();
}

extern "C" int ren_vanilla_bwd_weight(const void *dz, const void *saved, int64_t saved_slot_bytes, const float *enc, int32_t ld_enc,
                                      const float *view, int32_t ld_view, const float *dz_rgb, const float *dz_sigma, int32_t C,
                                      int32_t mode, int64_t n, int32_t n_splits, float *grads, float *workspace, void *stream) {
    return vanilla_bwd_weight_impl(dz, saved, saved_slot_bytes, enc, ld_enc, view, ld_view, dz_rgb, dz_sigma, C, mode, n, n_splits, grads,
                                   workspace, stream, false);
}

// the tangent stream's share of the weight gradients: dW_l += dzd_l^T xd_{l-1} (no bias terms), from the buffers of
// ren_vanilla_fwd_jvp / ren_vanilla_bwd_jvp (same layouts as the value stream's)
extern "C" int ren_vanilla_bwd_weight_tangent(const void *dzd, const void *savedd, int64_t saved_slot_bytes, const float *encd,
                                              int32_t ld_enc, const float *viewd, int32_t ld_view, const float *dzd_rgb,
                                              const float *dzd_sigma, int32_t C, int32_t mode, int64_t n, int32_t n_splits, float *grads,
                                              float *workspace, void *stream) {
    return vanilla_bwd_weight_impl(dzd, savedd, saved_slot_bytes, encd, ld_enc, viewd, ld_view, dzd_rgb, dzd_sigma, C, mode, n, n_splits,
                                   grads, workspace, stream, true);
}

extern "C" int ren_vanilla_fwd_jvp(const float *enc, int32_t ld_enc, const float *view, int32_t ld_view, const float *encd,
                                   const float *viewd, const uint8_t *selector, const float *params, int32_t C, int32_t activations,
                                   const void *image, int32_t mode, int64_t n, void *saved, void *savedd, float *sigma, float *rgb4,
                                   float *zsd4, float *zod4, void *stream) {
    if (!enc || !view || !encd || !viewd || !selector || !params || !image || !sigma || !rgb4 || !zsd4 || !zod4 || n < 0 || ld_enc < 64 ||
        (ld_enc & 3) || ld_view < 32 || (ld_view & 3) || C < 1 || C > 4 || ((saved == nullptr) != (savedd == nullptr)))
        return REN_ERR_BAD_ARG;
    if (!vfield_mode_ok(mode) || activations != 0) return REN_ERR_UNSUPPORTED;   // shipped activations: else the per-layer launches
    if (mode != 1 && !saved) return REN_ERR_BAD_ARG;                            // fp32 mode: the tangent launch reads the value's saved copies
    if (n == 0) return REN_OK;
    FieldArgs a = {};
    a.enc = enc; a.ld_enc = ld_enc; a.view = view; a.ld_view = ld_view; a.encd = encd; a.viewd = viewd; a.sel = selector; a.P = params;
    a.C = C; a.img = reinterpret_cast<const __bf16 *>(image); a.acts = saved; a.actsd = savedd; a.sigma = sigma; a.rgb4 = rgb4;
    a.zsd4 = zsd4; a.zod4 = zod4; a.n = n;
    hipStream_t st = (hipStream_t)stream;
    if (mode != 1) {
        // fp32 family (modes 6 / 3): two launches of the reduction-outer forward -- the value (saves y), then the tangent, whose
        // epilogue takes sp'(z) from those saved copies
#define REN_VFIELD_FWD6_JVP(MODE)                                                                                               \
    do {                                                                                                                        \
        const size_t lds = fwd6_lds(vfield_np(MODE));                                                                           \
        (void)hipFuncSetAttribute((const void *)vfield_fwd6_kernel<true, true, false, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((vfield_fwd6_kernel<true, true, false, MODE>), dim3(vfield_grid(n, 1)), dim3(256), lds, st, a);      \
        (void)hipFuncSetAttribute((const void *)vfield_fwd6_kernel<true, true, true, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((vfield_fwd6_kernel<true, true, true, MODE>), dim3(vfield_grid(n, 1)), dim3(256), lds, st, a);       \
    } while (0)
        if (mode == 3) REN_VFIELD_FWD6_JVP(3); else REN_VFIELD_FWD6_JVP(6);
#undef REN_VFIELD_FWD6_JVP
        REN_CHECK_LAUNCH();
    }
    const int grid = vfield_grid(n, 1);
    if (saved) {
        (void)hipFuncSetAttribute((const void *)vfield_fwd_jvp_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fwd_lds<1>());
        hipLaunchKernelGGL(vfield_fwd_jvp_kernel<true>, dim3(grid), dim3(256), fwd_lds<1>(), st, a);
    } else {
        (void)hipFuncSetAttribute((const void *)vfield_fwd_jvp_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fwd_lds<1>());
        hipLaunchKernelGGL(vfield_fwd_jvp_kernel<false>, dim3(grid), dim3(256), fwd_lds<1>(), st, a);
    }
    REN_CHECK_LAUNCH();
}

extern "C" int ren_vanilla_bwd_jvp(const float *dz_rgb, const float *dzd_rgb, const float *dz_sigma, const float *dzd_sigma,
                                   const void *image, int32_t mode, int32_t activations, int64_t n, const void *saved, const void *savedd,
                                   int64_t saved_slot_bytes, void *dz, void *dzd, void *coupling, void *stream) {
    if (!dz_rgb || !dzd_rgb || !dz_sigma || !dzd_sigma || !image || !saved || !savedd || !dz || !dzd || n < 0 || saved_slot_bytes < 0)
        return REN_ERR_BAD_ARG;
    if (!vfield_mode_ok(mode) || activations != 0) return REN_ERR_UNSUPPORTED;
    if (mode != 1 && !coupling) return REN_ERR_BAD_ARG;
    if (n == 0) return REN_OK;
    FieldArgs a = {};
    a.img = reinterpret_cast<const __bf16 *>(image) + (size_t)F_FRAGS * vfield_np(mode) * 512;
    a.acts = const_cast<void *>(saved); a.actsd = const_cast<void *>(savedd); a.dz_rgb = dz_rgb; a.dzd_rgb = dzd_rgb; a.dz_sig = dz_sigma;
    a.dzd_sig = dzd_sigma; a.dz = dz; a.dzd = dzd; a.n = n;
    a.acts_sstride = saved_slot_bytes / (mode == 1 ? 2 : 4);
    if (mode != 1) {
        // the tangent side first (its own chain; leaves the coupling term per layer), then the value side, which adds it
        a.cpl = coupling;
        FieldArgs t = a;
        t.dz_rgb = dzd_rgb; t.dz_sig = dzd_sigma; t.dz = dzd;
#define REN_VFIELD_BWD6_JVP(MODE)                                                                                               \
    do {                                                                                                                        \
        const size_t lds = bwd6_lds(vfield_np(MODE));                                                                           \
        (void)hipFuncSetAttribute((const void *)vfield_bwd6_kernel<1, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((vfield_bwd6_kernel<1, MODE>), dim3(vfield_grid(n, 1)), dim3(256), lds, (hipStream_t)stream, t);     \
        (void)hipFuncSetAttribute((const void *)vfield_bwd6_kernel<2, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((vfield_bwd6_kernel<2, MODE>), dim3(vfield_grid(n, 1)), dim3(256), lds, (hipStream_t)stream, a);     \
    } while (0)
        if (mode == 3) REN_VFIELD_BWD6_JVP(3); else REN_VFIELD_BWD6_JVP(6);
#undef REN_VFIELD_BWD6_JVP
        REN_CHECK_LAUNCH();
    }
    (void)hipFuncSetAttribute((const void *)vfield_bwd_jvp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bwd_lds<1>());
    hipLaunchKernelGGL(vfield_bwd_jvp_kernel, dim3(vfield_grid(n, 1)), dim3(256), bwd_lds<1>(), (hipStream_t)stream, a);
    REN_CHECK_LAUNCH();
}
