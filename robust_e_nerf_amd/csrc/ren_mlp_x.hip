// Fused Instant-NGP MLPs on the bf16 matrix cores at fp32 accuracy ("split-bf16").
//
// Why: on gfx950 the f32 MFMA runs at the f32 vector rate and its cycles ADD to the VALU's (tools/
// mfma_valu_bench.hip), so the exact-fp32 kernels of ren_mlp.hip cost MFMA + VALU.  The bf16 MFMA
// (v_mfma_f32_32x32x16_bf16) is 16x faster per multiply-add; its cycles ADD to the VALU's as well -- measured in round 5,
// tools/phase_overlap_bench.hip: a SIMD's matrix pipe and VALU take turns at 1, 2 and 4 waves per SIMD, PMC MfmaUtil +
// VALUBusy ~ 100 % -- so what these kernels cost is their matrix-pipe cycles PLUS their VALU cycles.  An fp32 value
// splits exactly into three bf16 pieces v = v1 + v2 + v3 (8 + 8 + 8 significant bits, residual <= 2^-27 |v|);
// bf16 x bf16 products are exact in the fp32 accumulator, so
//     a b  =  a1 b1 + a1 b2 + a2 b1 + a1 b3 + a3 b1 + a2 b2  + O(2^-25 |a b|)
// i.e. six bf16 MFMAs per 16-wide k-chunk (192 matrix-pipe cycles) reproduce the fp32 product to fp32
// round-off, against 8 f32 MFMAs (512 cycles on the shared pipe).  MODE 1 keeps only a1 b1: the plain bf16
// numerics of BASELINE configs[2].
//
// Layout: as in ren_mlp.hip a wavefront owns 32 samples = MFMA columns and the accumulator registers of one
// layer feed the next: register g of lane (sample, hi) holds neuron (g&3) + 8 (g>>2) + 4 hi, so k-chunk
// (tile t', half h) of the next layer takes registers 8h..8h+7 of accumulator t' as its 8 k-slots and the
// weight fragments are stored with the SAME slot -> neuron map.  Fragments live in LDS pre-split and
// pre-permuted: [tile][chunk][term][lane] x 8 bf16 = one ds_read_b128 per MFMA operand.
#include "ren_mlp_xfrag.h"

namespace {

// LDS image (bytes): fragments, then f32 biases and the output layer
template <int NT> struct XL {
    static constexpr int F_W1 = 0;                              // 2 tiles x 2 chunks
    static constexpr int F_W2 = F_W1 + 2 * 2 * NT * 512;        // 1 x 4
    static constexpr int F_WH1 = F_W2 + 1 * 4 * NT * 512;       // 2 x 2
    static constexpr int F_WH2 = F_WH1 + 2 * 2 * NT * 512;      // 2 x 4
    static constexpr int F_END = F_WH2 + 2 * 4 * NT * 512;      // bf16 elements
    static constexpr int BYTES_F = F_END * 2;
    // f32 tail: b1[64] b2[32] bh1[64] bh2[64] wh3[3*64] bh3[4]
    static constexpr int T_B1 = 0, T_B2 = 64, T_BH1 = 96, T_BH2 = 160, T_WH3 = 224, T_BH3 = 416, T_END = 420;
    static constexpr size_t BYTES = (size_t)BYTES_F + T_END * 4;
};

struct FwdXArgs {
    const float *params, *feat;
    SampleSrc src;
    ren_scene_dev sc;
    int64_t n;
    float *rgb, *sigma, *base_out, *acts;
    const int64_t *n_dev;                                     // device-side sample count (ren_eff_n) or NULL
};

constexpr int ACT_SAVE_FLOATS_X = 3 * 2 * 16 * 64;            // same layout as ren_mlp.hip's ACT_SAVE_FLOATS

// 8 waves share one 63 KB fragment image, two workgroups per CU: four waves per SIMD hide the LDS latency of the weight
// fragments and the global loads (they do NOT run one wave's VALU work under another's MFMAs: the two pipes of a SIMD take
// turns)
constexpr int FWD_X_WAVES = 8;
#ifndef REN_FWD_WAVES
#define REN_FWD_WAVES 4                                   // waves per SIMD the register allocation must allow (2 workgroups per CU)
#endif

template <int C, int MODE, bool DENSITY_ONLY>
__global__ __launch_bounds__(64 * FWD_X_WAVES, REN_FWD_WAVES) void mlp_fwd_x_kernel(FwdXArgs a) {
    using PR = Pairs<MODE>;
    constexpr int NT = PR::NT;
    using L = XL<NT>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __bf16 *frag = reinterpret_cast<__bf16 *>(smem);
    float *tail = reinterpret_cast<float *>(smem + L::BYTES_F);
    fill_frags<NT, 0>(frag + L::F_W1, a.params, 2, 2);
    fill_frags<NT, 1, true>(frag + L::F_W2, a.params, 1, 4);           // odd chunks negated: o - o2 below
    if (!DENSITY_ONLY) {
        fill_frags<NT, 2>(frag + L::F_WH1, a.params, 2, 2);
        fill_frags<NT, 3>(frag + L::F_WH2, a.params, 2, 4);
    }
    for (int i = threadIdx.x; i < 64; i += blockDim.x) {
        tail[L::T_B1 + i] = a.params[P_BB0 + i];
        tail[L::T_BH1 + i] = a.params[P_HB0 + i];
        tail[L::T_BH2 + i] = a.params[P_HB1 + i];
        if (i < 32) tail[L::T_B2 + i] = i < 16 ? a.params[P_BBO + i] : 0.f;
    }
    for (int i = threadIdx.x; i < 64 * C; i += blockDim.x) tail[L::T_WH3 + i] = a.params[P_HWO + i];
    if (threadIdx.x < C) tail[L::T_BH3 + threadIdx.x] = a.params[P_HWO + 64 * C + threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hi = lane >> 5, sl = lane & 31;
    const int64_t n_smp = ren_eff_n(a.n, a.n_dev), n_blk = (n_smp + 31) >> 5;

    for (int64_t blk = (int64_t)blockIdx.x * FWD_X_WAVES + wave; blk < n_blk; blk += (int64_t)gridDim.x * FWD_X_WAVES) {
        int zo = 0;                                             // keep the (loop-invariant) LDS reads inside the loop
        asm volatile("" : "+v"(zo));
        const __bf16 *fr = frag + zo;
        const float *tl = tail + zo;
        const int64_t i = blk * 32 + sl;
        const bool live = i < n_smp;
        // ---- hash features -> two k-chunks
        bf16x8 bx[2][3];
        {
            const float *f = a.feat + blk * (16 * 64) + lane;
            float x[16];
#pragma unroll
            for (int s = 0; s < 16; ++s) x[s] = f[s * 64];
            split8<NT>(x, bx[0]);
            split8<NT>(x + 8, bx[1]);
        }
        // ---- base layer 0: 32 -> 64, softplus(beta = 100)
        f32x16 h[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 16; ++g) h[t][g] = tl[L::T_B1 + 32 * t + rowc(g) + 4 * hi];
#pragma unroll
        for (int c = 0; c < 2; ++c) mma2<MODE>(h[0], h[1], fr + L::F_W1, 2, c, bx[c], lane);
        bf16x8 bh[4][3];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float y[16];
#pragma unroll
            for (int g = 0; g < 16; ++g) y[g] = softplus100(h[t][g]);
            if (a.acts) {                                       // ONE branch, constant offsets
                float *ap = a.acts + blk * ACT_SAVE_FLOATS_X + t * 1024 + lane;
#pragma unroll
                for (int g = 0; g < 16; ++g) __builtin_nontemporal_store(y[g], ap + g * 64);
            }
            split8<NT>(y, bh[2 * t]);
            split8<NT>(y + 8, bh[2 * t + 1]);
        }
        // ---- base output: 64 -> 16 (rows 16..31 of the tile are zero)
        f32x16 o, o2;
#pragma unroll
        for (int g = 0; g < 16; ++g) { o[g] = tl[L::T_B2 + rowc(g) + 4 * hi]; o2[g] = 0.f; }
        mma1x2<MODE>(o, o2, fr + L::F_W2, 4, 0, 1, bh[0], bh[1], lane);
        mma1x2<MODE>(o, o2, fr + L::F_W2, 4, 2, 3, bh[2], bh[3], lane);
#pragma unroll
        for (int g = 0; g < 16; ++g) o[g] -= o2[g];                // o2 ran on the negated odd chunks (fill_frags NEG_ODD)
        bool sel = false;
        float dx = 0.f, dy = 0.f, dz = 1.f;
        if (live) sample_geom(a.src, a.sc, i, sel, dx, dy, dz);
        if (live && hi == 0) a.sigma[i] = sel ? __expf(o[0] - 1.f) : 0.f;          // ngp.py:247-250
        if (a.base_out) {
            float *bo = a.base_out + blk * (8 * 64) + lane;
#pragma unroll
            for (int g = 0; g < 8; ++g) bo[g * 64] = o[g];
        }
        if (DENSITY_ONLY) continue;
        // ---- head layer 0: [base_out(16) | SH(16)] -> 64
        bf16x8 bv[2][3];
        {
            float v0[8], shs[8];
#pragma unroll
            for (int g = 0; g < 8; ++g) v0[g] = o[g];
            sh4_select(dx, dy, dz, hi, shs);
            split8<NT>(v0, bv[0]);
            split8<NT>(shs, bv[1]);
        }
        f32x16 p[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 16; ++g) p[t][g] = tl[L::T_BH1 + 32 * t + rowc(g) + 4 * hi];
#pragma unroll
        for (int c = 0; c < 2; ++c) mma2<MODE>(p[0], p[1], fr + L::F_WH1, 2, c, bv[c], lane);
        bf16x8 bp[4][3];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float y[16];
#pragma unroll
            for (int g = 0; g < 16; ++g) y[g] = softplus100(p[t][g]);
            if (a.acts) {
                float *ap = a.acts + blk * ACT_SAVE_FLOATS_X + (2 + t) * 1024 + lane;
#pragma unroll
                for (int g = 0; g < 16; ++g) __builtin_nontemporal_store(y[g], ap + g * 64);
            }
            split8<NT>(y, bp[2 * t]);
            split8<NT>(y + 8, bp[2 * t + 1]);
        }
        // ---- head layer 1: 64 -> 64
        f32x16 q[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 16; ++g) q[t][g] = tl[L::T_BH2 + 32 * t + rowc(g) + 4 * hi];
#pragma unroll
        for (int c = 0; c < 4; ++c) mma2<MODE>(q[0], q[1], fr + L::F_WH2, 4, c, bp[c], lane);
        // ---- head output: 64 -> C on the VALU in fp32 (MODE 1: bf16-rounded operands, as every other layer)
        float acc[C];
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int g = 0; g < 16; ++g) q[t][g] = softplus100(q[t][g]);
            if (a.acts) {
                float *ap = a.acts + blk * ACT_SAVE_FLOATS_X + (4 + t) * 1024 + lane;
#pragma unroll
                for (int g = 0; g < 16; ++g) __builtin_nontemporal_store(q[t][g], ap + g * 64);
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const float qy = q[t][g];
                const float qa = MODE == 1 ? (float)(__bf16)qy : qy;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const float w3 = tl[L::T_WH3 + c * 64 + 32 * t + rowc(g) + 4 * hi];
                    acc[c] += qa * (MODE == 1 ? (float)(__bf16)w3 : w3);
                }
            }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float s = acc[c] + __shfl_xor(acc[c], 32, 64);
            if (hi == 0 && live) a.rgb[i * C + c] = softplus1(s + tl[L::T_BH3 + c]);
        }
    }
}

template <int MODE>
int launch_fwd_x(const FwdXArgs &a, int C, bool density_only, bool one, hipStream_t st) {
    using L = XL<Pairs<MODE>::NT>;
    const int64_t n_blk = (a.n + 31) / 32;
    int64_t blocks = (n_blk + FWD_X_WAVES - 1) / FWD_X_WAVES;
    // two workgroups per CU; REN_MLP_SHARE_CU: one (the LDS request keeps a second one out), so that a kernel on
    // another stream finds half of every CU's registers and wave slots free
    if (blocks > (one ? 256 : 512)) blocks = one ? 256 : 512;
    const size_t lds = one ? (L::BYTES > 84 * 1024 ? L::BYTES : 84 * 1024) : L::BYTES;
    const dim3 grd((int)blocks), blk(64 * FWD_X_WAVES);
#define REN_X(CC, DO)                                                                                       \
    do {                                                                                                    \
        (void)hipFuncSetAttribute((const void *)mlp_fwd_x_kernel<CC, MODE, DO>,                             \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);               \
        hipLaunchKernelGGL((mlp_fwd_x_kernel<CC, MODE, DO>), grd, blk, lds, st, a);                    \
    } while (0)
    if (density_only) REN_X(1, true);
    else if (C == 1)  REN_X(1, false);
    else              REN_X(3, false);
#undef REN_X
    REN_CHECK_LAUNCH();
}

}  // namespace

// mode: 6 = split-bf16 at fp32 accuracy (fp32 parameter block), 1 = plain bf16 operands (BASELINE configs[2])
extern "C" int ren_mlp_fwd_x(const float *mlp_params, int32_t C, int32_t activations, int32_t mode, const float *feat,
                             const ren_scene_desc *scene, const float *x_world, const float *dirs,
                             const float *rays_o, const float *rays_d, const int32_t *ray_indices,
                             const float *t_starts, const float *t_ends, int64_t n, int32_t flags,
                             float *rgb, float *sigma, float *base_out, float *act_save, const int64_t *n_dev, void *stream) {
    if (!mlp_params || !feat || !scene || !sigma || n < 0) return REN_ERR_BAD_ARG;
    if (C != 1 && C != 3) return REN_ERR_UNSUPPORTED;
    if (mode != 1 && mode != 3 && mode != 6) return REN_ERR_UNSUPPORTED;
    if (activations != 0) return REN_ERR_UNSUPPORTED;      // activation alternatives: exact-f32 kernels only
    const bool density_only = (flags & REN_MLP_DENSITY_ONLY) != 0, share = (flags & REN_MLP_SHARE_CU) != 0;
    if (flags & ~(REN_MLP_DENSITY_ONLY | REN_MLP_SHARE_CU)) return REN_ERR_BAD_ARG;
    if (!density_only && !rgb) return REN_ERR_BAD_ARG;
    if (act_save && (density_only || !base_out)) return REN_ERR_BAD_ARG;
    if (!x_world && (!rays_o || !rays_d || !ray_indices || !t_starts || !t_ends)) return REN_ERR_BAD_ARG;
    if (n == 0) return REN_OK;
    FwdXArgs a;
    a.params = mlp_params; a.feat = feat;
    a.src = SampleSrc{x_world, dirs, rays_o, rays_d, x_world ? nullptr : ray_indices, t_starts, t_ends};
    a.sc = ren_make_scene(scene);
    a.n = n; a.rgb = rgb; a.sigma = sigma; a.base_out = base_out; a.acts = act_save; a.n_dev = n_dev;
    if (mode == 3) return launch_fwd_x<3>(a, C, density_only, share, (hipStream_t)stream);
    return mode == 6 ? launch_fwd_x<6>(a, C, density_only, share, (hipStream_t)stream)
                     : launch_fwd_x<1>(a, C, density_only, share, (hipStream_t)stream);
}

// ================================================================================================ backward
// Two persistent kernels (head, base), one wave per SIMD with the whole register file.  Data gradients (W^T dZ
// chains) and weight gradients use the same term pairs as the forward: six bf16 products per fp32 product in MODE 6
// (every product of the default path is formed to fp32 round-off), one in MODE 1.  The weight-gradient operands
// (k = sample) come from the chain's own split operands by a transposition on the matrix cores (transpose_tile,
// ren_mlp_xfrag.h): nothing is staged through the LDS, which holds the weight fragments only.
namespace {

#ifndef REN_BASE_WAVES
#define REN_BASE_WAVES 2                                  // waves per SIMD of the base kernel
#endif
#ifdef REN_NO_PHASE_FENCE
#define REN_PHASE_FENCE() do { } while (0)
#else
#define REN_PHASE_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

#ifndef REN_GRID_CUS
#define REN_GRID_CUS 256                                  // CUs the persistent backward kernels occupy
#endif
constexpr int GRID_XH = REN_GRID_CUS, GRID_XB = REN_GRID_CUS * REN_BASE_WAVES;   // persistent workgroups of 4 waves: head one per CU, base REN_BASE_WAVES

struct BwdXHArgs {
    const float *params, *base_out, *acts;
    SampleSrc src;
    ren_scene_dev sc;
    int64_t n;
    const float *rgb, *d_rgb, *d_sigma;
    float *d_base, *slab;
    const int64_t *n_dev;
};

// RECOMP: the hidden activations p, q are recomputed from the saved base outputs with the forward's own MFMA
// sequence (bit-identical to what the forward would have stored) instead of being loaded: 512 B/sample less HBM
// traffic each way (the forward then writes 64 B/sample instead of 832).
template <int MODE, bool RECOMP> struct HeadLds {
    static constexpr int NT = Pairs<MODE>::NT;
    static constexpr int F_WH2T = 0, F_WH1T = 2 * 4 * NT * 512, F_WH1 = F_WH1T + 1 * 4 * NT * 512;    // bf16 elements
    static constexpr int F_WH2 = F_WH1 + (RECOMP ? 2 * 2 * NT * 512 : 0), F_END = F_WH2 + (RECOMP ? 2 * 4 * NT * 512 : 0);
    static constexpr int T_W3 = 0, T_BH1 = 256, T_BH2 = 320, T_END = 384;                          // f32 tail
    static constexpr int TAIL_BYTES = T_END * 4;
};

template <int C, int MODE, bool RECOMP>
__global__ __launch_bounds__(256, 1) void mlp_bwd_head_x_kernel(BwdXHArgs a) {
    using PR = Pairs<MODE>;
    using HL = HeadLds<MODE, RECOMP>;
    constexpr int NT = PR::NT;
    constexpr int F_WH2T = HL::F_WH2T, F_WH1T = HL::F_WH1T, F_END = HL::F_END;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __bf16 *frag = reinterpret_cast<__bf16 *>(smem);
    float *w3 = reinterpret_cast<float *>(smem + F_END * 2);                       // [C][64] (+ pad), then bh1[64] bh2[64]
    fill_frags_t<NT, 3, true>(frag + F_WH2T, a.params, 2, 4);                      // head.w1^T : rows = p index (odd chunks negated)
    fill_frags_t<NT, 2, true>(frag + F_WH1T, a.params, 1, 4);                      // head.w0^T : rows = v index (odd chunks negated)
    if (RECOMP) {
        fill_frags<NT, 2>(frag + HL::F_WH1, a.params, 2, 2);                       // forward fragments of head.w0, head.w1
        fill_frags<NT, 3>(frag + HL::F_WH2, a.params, 2, 4);
        for (int i = threadIdx.x; i < 64; i += blockDim.x) {
            w3[HL::T_BH1 + i] = a.params[P_HB0 + i];
            w3[HL::T_BH2 + i] = a.params[P_HB1 + i];
        }
    }
    for (int i = threadIdx.x; i < 64 * C; i += blockDim.x) {
        const float w = a.params[P_HWO + i];
        w3[i] = MODE == 1 ? (float)(__bf16)w : w;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hi = lane >> 5, sl = lane & 31;
    const bf16x8 sel0 = make_sel<0>(lane), sel1 = make_sel<1>(lane), sel_sh = make_sel<3>(lane);
    __syncthreads();
    const int64_t n_smp = ren_eff_n(a.n, a.n_dev), n_blk = (n_smp + 31) >> 5;

    // weight gradients: 32 x 32 tiles accumulated over the whole persistent loop.  head.b0's gradient rides in
    // column v = 0 of acc_wh1 (that input slot carries no weight: the V operand holds 1 there); the other bias /
    // output-layer sums are kept per lane (= per sample slot) and reduced over the 32 lanes of a half at the end
    // the per-lane sums of head.b1 (and of the output layer for C == 1) are kept in lane-private LDS slots ([value / 4][lane]
    // float4: one b128 read-modify-write per four values), which leaves their registers to the chain -- the C == 1
    // kernel does not spill; C == 3 (Bayer sensors) keeps its 96 output-layer sums in registers
    constexpr bool W3_LDS = C == 1;                            // slots 0..7: head.wo (C == 1), 8..15: head.b1
    float4 *accl = reinterpret_cast<float4 *>(smem + F_END * 2 + HL::TAIL_BYTES) + wave * (16 * 64) + lane;
    f32x16 acc_wh2[2][2], acc_wh1[2];
    float acc_w3[W3_LDS ? 1 : C][32], acc_bh3[C];
#pragma unroll
    for (int v = 0; v < 16; ++v) accl[v * 64] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        acc_wh2[0][0][g] = 0.f; acc_wh2[0][1][g] = 0.f; acc_wh2[1][0][g] = 0.f; acc_wh2[1][1][g] = 0.f;
        acc_wh1[0][g] = 0.f; acc_wh1[1][g] = 0.f;
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
        acc_bh3[c] = 0.f;
        if (!W3_LDS) {
#pragma unroll
            for (int k = 0; k < 32; ++k) acc_w3[c][k] = 0.f;
        }
    }

    // One wave per SIMD hides no latency by itself: the block's global inputs are fetched one iteration ahead (the ray
    // index of the packed stream two ahead, because the ray origin / direction loads depend on it).  Out-of-range
    // blocks read clamped addresses and are never used.
    const int64_t stride = (int64_t)gridDim.x * 4, blk0 = (int64_t)blockIdx.x * 4 + wave;
    const int64_t i_last = n_smp > 0 ? n_smp - 1 : 0;
    const bool packed = a.src.ray_indices != nullptr;
    // Branch-free: a run-time `if (packed)` around the loads made the compiler wait for them inside the prefetch (counters
    // are merged conservatively at the join, and `tm` was formed right there): a whole memory latency exposed per block
    // at one wave per SIMD.  Sources are selected by pointer instead, all arithmetic on the loaded values happens at use.
    struct Staged { float pos[3], dir[3], ts, te, o[8], rgb[C], d_rgb[C], d_sigma; };
    const float *p_pos = packed ? a.src.rays_o : a.src.x_world;
    const float *p_dir = packed ? a.src.rays_d : (a.src.dirs ? a.src.dirs : a.src.x_world);   // (no dirs: any three readable floats)
    const float *p_ts = packed ? a.src.t_starts : a.d_sigma, *p_te = packed ? a.src.t_ends : a.d_sigma;
    const bool has_dir = packed || a.src.dirs != nullptr, dir3 = has_dir;
    auto load_ray = [&](int64_t blk) -> int {
        const int64_t i = min(blk * 32 + sl, i_last);
        return a.src.ray_indices ? a.src.ray_indices[i] : 0;
    };
    auto load_inputs = [&](int64_t blk, int ray, Staged &st) {
        const int64_t i = min(blk * 32 + sl, i_last), b = min(blk, n_blk - 1);
        const int64_t ip = packed ? 3 * (int64_t)ray : 3 * i, id = dir3 ? ip : 0;
        st.ts = p_ts[i]; st.te = p_te[i];
#pragma unroll
        for (int k = 0; k < 3; ++k) { st.pos[k] = p_pos[ip + k]; st.dir[k] = p_dir[id + k]; }
        const float *bo = a.base_out + b * (8 * 64) + lane;
#pragma unroll
        for (int g = 0; g < 8; ++g) st.o[g] = bo[g * 64];
#pragma unroll
        for (int c = 0; c < C; ++c) { st.rgb[c] = a.rgb[i * C + c]; st.d_rgb[c] = a.d_rgb[i * C + c]; }
        st.d_sigma = a.d_sigma[i];
    };
    Staged nxt;
    int ray_nn = 0;
    if (blk0 < n_blk) {
        load_inputs(blk0, load_ray(blk0), nxt);
        ray_nn = load_ray(blk0 + stride);
    }

    for (int64_t blk = blk0; blk < n_blk; blk += stride) {
        int zo = 0;
        asm volatile("" : "+v"(zo));
        const __bf16 *fr = frag + zo;
        const float *W3 = w3 + zo;
        const int64_t i = blk * 32 + sl;
        const bool live = i < n_smp;
        const Staged cur = nxt;
        load_inputs(blk + stride, ray_nn, nxt);
        ray_nn = load_ray(blk + 2 * stride);
        bool sel = false;
        float dx = 0.f, dy = 0.f, dz = 1.f;
        if (live) {
            // position (contracted -> selector, ngp.py:238) and view direction, as sample_geom()
            float ux, uy, uz;
            const float tm = packed ? (cur.ts + cur.te) * 0.5f : 0.f;
            const float cdx = has_dir ? cur.dir[0] : 0.f, cdy = has_dir ? cur.dir[1] : 0.f, cdz = has_dir ? cur.dir[2] : 1.f;
            ren_contract(a.sc, cur.pos[0] + cdx * tm, cur.pos[1] + cdy * tm, cur.pos[2] + cdz * tm, ux, uy, uz);
            sel = ux > 0.f && ux < 1.f && uy > 0.f && uy < 1.f && uz > 0.f && uz < 1.f;
            dx = cdx; dy = cdy; dz = cdz;
        }
        float shs[8], o[8];
        sh4_select(dx, dy, dz, hi, shs);
#pragma unroll
        for (int g = 0; g < 8; ++g) o[g] = cur.o[g];
        const float o_sigma = o[0];                                // raw density (lanes hi = 0)
        if (hi == 0) o[0] = 1.f;                                   // v = 0: weight 0 in the chain, the ones row of V^T
        bf16x8 bv[2][3];                                           // V = [base_out | SH]: chain operand and source of V^T
        split8<NT>(o, bv[0]);
        split8<NT>(shs, bv[1]);
        float p[2][16], q[2][16];
        bf16x8 bp[4][3];                                           // split p: chain operand and source of p^T
        if (RECOMP) {
            f32x16 pa[2], qa[2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int g = 0; g < 16; ++g) pa[t][g] = W3[HL::T_BH1 + 32 * t + rowc(g) + 4 * hi];
#pragma unroll
            for (int c = 0; c < 2; ++c) mma2<MODE>(pa[0], pa[1], fr + HL::F_WH1, 2, c, bv[c], lane);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int g = 0; g < 16; ++g) p[t][g] = softplus100(pa[t][g]);
                split8<NT>(p[t], bp[2 * t]);
                split8<NT>(p[t] + 8, bp[2 * t + 1]);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int g = 0; g < 16; ++g) qa[t][g] = W3[HL::T_BH2 + 32 * t + rowc(g) + 4 * hi];
#pragma unroll
            for (int c = 0; c < 4; ++c) mma2<MODE>(qa[0], qa[1], fr + HL::F_WH2, 4, c, bp[c], lane);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int g = 0; g < 16; ++g) q[t][g] = softplus100(qa[t][g]);
        } else {
            const float *ac = a.acts + blk * ACT_SAVE_FLOATS_X + lane;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int g = 0; g < 16; ++g) { p[t][g] = ac[((2 + t) * 16 + g) * 64]; q[t][g] = ac[((4 + t) * 16 + g) * 64]; }
                split8<NT>(p[t], bp[2 * t]);
                split8<NT>(p[t] + 8, bp[2 * t + 1]);
            }
        }
        REN_PHASE_FENCE();
        // ---- p^T (k = sample operands of dW(head.w1))
        bf16x8 pT[2][2][3];
        transpose_tile<NT>(bp[0], bp[1], sel0, sel1, pT[0]);
        transpose_tile<NT>(bp[2], bp[3], sel0, sel1, pT[1]);
        REN_PHASE_FENCE();
        // ---- output layer: dz3, dW3, dq -> dz2 (fp32 VALU)
        float dz3[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            dz3[c] = live ? cur.d_rgb[c] * dsoftplus_from_out(cur.rgb[c], 1.f) : 0.f;
            if (hi == 0) acc_bh3[c] += dz3[c];
        }
        float dz2[2][16];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 ab = accl[(8 + t * 4 + g4) * 64];
                float4 aw = make_float4(0.f, 0.f, 0.f, 0.f);
                if (W3_LDS) aw = accl[(t * 4 + g4) * 64];
                float w[4] = {aw.x, aw.y, aw.z, aw.w}, b[4] = {ab.x, ab.y, ab.z, ab.w};
#pragma unroll
                for (int gi = 0; gi < 4; ++gi) {
                    const int g = 4 * g4 + gi;
                    const float qv = MODE == 1 ? (float)(__bf16)q[t][g] : q[t][g];
                    float dq = 0.f;
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        if (W3_LDS) w[gi] += dz3[c] * qv;
                        else acc_w3[c][t * 16 + g] += dz3[c] * qv;
                        dq += dz3[c] * W3[c * 64 + 32 * t + rowc(g) + 4 * hi];
                    }
                    dz2[t][g] = dq * dsoftplus_from_out(q[t][g], 100.f);
                    b[gi] += dz2[t][g];
                }
                if (W3_LDS) accl[(t * 4 + g4) * 64] = make_float4(w[0], w[1], w[2], w[3]);
                accl[(8 + t * 4 + g4) * 64] = make_float4(b[0], b[1], b[2], b[3]);
            }
        REN_PHASE_FENCE();
        // ---- dW(head.w1)[ot][it] += dz2(ot) . p(it)^T ;  d p = W1^T dz2   (dz2 is split once for both)
        f32x16 dp[2], dpn[2];                                      // even chunks | negated odd chunks
#pragma unroll
        for (int g = 0; g < 16; ++g) { dp[0][g] = 0.f; dp[1][g] = 0.f; dpn[0][g] = 0.f; dpn[1][g] = 0.f; }
        {
            bf16x8 bz[4][3];
#pragma unroll
            for (int t = 0; t < 2; ++t) { split8<NT>(dz2[t], bz[2 * t]); split8<NT>(dz2[t] + 8, bz[2 * t + 1]); }
            // the block's dW contributions are formed in FRESH accumulators and added to the persistent ones on the VALU
            // (round to nearest) -- see dw_block() -- after the d p chain has been issued, so the adds do not wait
            f32x16 tw[2][2];
#pragma unroll
            for (int ot = 0; ot < 2; ++ot) {
#pragma unroll
                for (int g = 0; g < 16; ++g) { tw[ot][0][g] = 0.f; tw[ot][1][g] = 0.f; }
                bf16x8 zT[2][3];
                transpose_tile<NT>(bz[2 * ot], bz[2 * ot + 1], sel0, sel1, zT);
                dw_acc2<MODE>(tw[ot][0], tw[ot][1], zT, pT[0], pT[1]);
            }
            mma2_pn<MODE>(dp[0], dp[1], dpn[0], dpn[1], fr + F_WH2T, 4, 0, 1, bz[0], bz[1], lane);
            mma2_pn<MODE>(dp[0], dp[1], dpn[0], dpn[1], fr + F_WH2T, 4, 2, 3, bz[2], bz[3], lane);
#pragma unroll
            for (int ot = 0; ot < 2; ++ot)
#pragma unroll
                for (int g = 0; g < 16; ++g) { acc_wh2[ot][0][g] += tw[ot][0][g]; acc_wh2[ot][1][g] += tw[ot][1][g]; }
        }
        REN_PHASE_FENCE();
        float dz1[2][16];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 16; ++g) dz1[t][g] = (dp[t][g] - dpn[t][g]) * dsoftplus_from_out(p[t][g], 100.f);
        // ---- dW(head.w0)[ot] += dz1(ot) . V^T (column v = 0: the bias gradient) ;  d V = W0^T dz1
        f32x16 dv, dv2;
#pragma unroll
        for (int g = 0; g < 16; ++g) { dv[g] = 0.f; dv2[g] = 0.f; }
        {
            bf16x8 vT[2][3];
            transpose_tile<NT>(bv[0], bv[1], sel0, sel_sh, vT);
            bf16x8 bz[4][3];
#pragma unroll
            for (int t = 0; t < 2; ++t) { split8<NT>(dz1[t], bz[2 * t]); split8<NT>(dz1[t] + 8, bz[2 * t + 1]); }
            f32x16 tw[2];
#pragma unroll
            for (int ot = 0; ot < 2; ++ot) {
#pragma unroll
                for (int g = 0; g < 16; ++g) tw[ot][g] = 0.f;
                bf16x8 zT[2][3];
                transpose_tile<NT>(bz[2 * ot], bz[2 * ot + 1], sel0, sel1, zT);
                dw_acc<MODE>(tw[ot], zT, vT);
            }
            mma1x2<MODE>(dv, dv2, fr + F_WH1T, 4, 0, 1, bz[0], bz[1], lane);
            mma1x2<MODE>(dv, dv2, fr + F_WH1T, 4, 2, 3, bz[2], bz[3], lane);
#pragma unroll
            for (int g = 0; g < 16; ++g) { acc_wh1[0][g] += tw[0][g]; acc_wh1[1][g] += tw[1][g]; }
#pragma unroll
            for (int g = 0; g < 16; ++g) dv[g] -= dv2[g];
        }
        if (hi == 0) {                                             // rows 0..15 = d base_out; row 0 takes the density gradient
            const float ds = live ? cur.d_sigma : 0.f;
            dv[0] = sel ? ds * __expf(fminf(o_sigma - 1.f, 15.f)) : 0.f;           // ngp.py:54-58,247-250
        }
        {
            float *db = a.d_base + blk * (8 * 64) + lane;
#pragma unroll
            for (int g = 0; g < 8; ++g) db[g * 64] = dv[g];
        }
    }
    // ---- slab (head part of the parameter block), same layout as ren_mlp.hip
    float *slab = a.slab + ((int64_t)blockIdx.x * 4 + wave) * (p_total(C) - P_BASE_N);
    constexpr int O = -P_BASE_N;
#pragma unroll
    for (int ob = 0; ob < 2; ++ob) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int out = 32 * ob + rowc(g) + 4 * hi;
            slab[O + P_HW1 + out * 64 + sl] = acc_wh2[ob][0][g];
            slab[O + P_HW1 + out * 64 + 32 + sl] = acc_wh2[ob][1][g];
            if (sl != 0) slab[O + P_HW0 + out * 31 + (sl < 16 ? 15 + sl : sl - 16)] = acc_wh1[ob][g];
            else slab[O + P_HB0 + out] = acc_wh1[ob][g];
        }
#pragma unroll
        for (int g = 0; g < 16; ++g) {                              // bias: sum over the 32 sample lanes of this half
            float b2 = reinterpret_cast<const float *>(accl + (8 + ob * 4 + (g >> 2)) * 64)[g & 3];
#pragma unroll
            for (int off = 1; off < 32; off <<= 1) b2 += __shfl_xor(b2, off, 64);
            if (sl == 0) slab[O + P_HB1 + 32 * ob + rowc(g) + 4 * hi] = b2;
        }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            float v = W3_LDS ? reinterpret_cast<const float *>(accl + (k >> 2) * 64)[k & 3] : acc_w3[c][k];
#pragma unroll
            for (int off = 1; off < 32; off <<= 1) v += __shfl_xor(v, off, 64);
            if (sl == 0) slab[O + P_HWO + c * 64 + 32 * (k >> 4) + rowc(k & 15) + 4 * hi] = v;
        }
        const float b3 = ren_wave_sum(acc_bh3[c]);
        if (lane == 0) slab[O + P_HWO + 64 * C + c] = b3;
    }
}

struct BwdXBArgs {
    const float *params, *feat, *d_base, *acts;
    int64_t n;
    float *dfeat, *slab;
    const int64_t *n_dev;
};

template <int MODE, bool RECOMP> struct BaseLds {
    static constexpr int NT = Pairs<MODE>::NT;
    // base.wo^T is stored twice, as it is and negated (F_W2TN): d h = Wo^T dO is a single 16-wide k-chunk, so there is no
    // second chunk to pair it with; instead every other block of a wave computes -d h from the negated fragments (the sign
    // is folded into the activation derivative), and the accumulate's rounding bias alternates in sign from block to
    // block: it cancels in the sums over samples that base.b0 / base.w0 are
    static constexpr int F_W2T = 0, F_W2TN = 2 * 1 * NT * 512, F_W1T = 2 * F_W2TN, F_W1 = F_W1T + 1 * 4 * NT * 512;
    static constexpr int F_END = F_W1 + (RECOMP ? 2 * 2 * NT * 512 : 0);
    static constexpr int TAIL_BYTES = RECOMP ? 64 * 4 : 0;                                           // b1[64]
};

template <int MODE, bool RECOMP>
__global__ __launch_bounds__(256, REN_BASE_WAVES) void mlp_bwd_base_x_kernel(BwdXBArgs a) {
    using PR = Pairs<MODE>;
    using BL = BaseLds<MODE, RECOMP>;
    constexpr int NT = PR::NT;
    constexpr int F_W2T = BL::F_W2T, F_W1T = BL::F_W1T, F_END = BL::F_END;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __bf16 *frag = reinterpret_cast<__bf16 *>(smem);
    float *b1 = reinterpret_cast<float *>(smem + F_END * 2);
    fill_frags_t<NT, 1>(frag + F_W2T, a.params, 2, 1);                             // base.wo^T : rows = h index, 1 chunk (16 outs)
    fill_frags_t<NT, 1>(frag + BL::F_W2TN, a.params, 2, 1);
    __syncthreads();
    for (int e = threadIdx.x; e < BL::F_W2TN; e += blockDim.x) frag[BL::F_W2TN + e] = -frag[BL::F_W2TN + e];
    fill_frags_t<NT, 0, true>(frag + F_W1T, a.params, 1, 4);                       // base.w0^T : rows = feature index (odd chunks negated)
    if (RECOMP) {
        fill_frags<NT, 0>(frag + BL::F_W1, a.params, 2, 2);                        // forward fragments of base.w0
        for (int i = threadIdx.x; i < 64; i += blockDim.x) b1[i] = a.params[P_BB0 + i];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hi = lane >> 5, sl = lane & 31;
    const bf16x8 sel0 = make_sel<0>(lane), sel1 = make_sel<1>(lane), sel_x0 = make_sel<2>(lane), sel_x1 = make_sel<3>(lane);
    __syncthreads();
    const int64_t n_blk = (ren_eff_n(a.n, a.n_dev) + 31) >> 5;
    // base.b0's 32 per-lane sums live in lane-private LDS slots ([value / 4][lane] float4), as in the head kernel
    // (slots 0..7: base.b0, 8..9: base.bo)
    float4 *accl = reinterpret_cast<float4 *>(smem + F_END * 2 + BL::TAIL_BYTES) + wave * (10 * 64) + lane;
    f32x16 acc_w2[2], acc_w1[2];
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        acc_w2[0][g] = 0.f; acc_w2[1][g] = 0.f; acc_w1[0][g] = 0.f; acc_w1[1][g] = 0.f;
        if (g < 10) accl[g * 64] = make_float4(0.f, 0.f, 0.f, 0.f);
    }

    bool flip = false;
    for (int64_t blk = (int64_t)blockIdx.x * 4 + wave; blk < n_blk; blk += (int64_t)gridDim.x * 4, flip = !flip) {
        int zo = 0;
        asm volatile("" : "+v"(zo));
        const __bf16 *fr = frag + zo;
        float x[16], dob[8], h[2][16];
        {
            const float *f = a.feat + blk * (16 * 64) + lane;
#pragma unroll
            for (int s = 0; s < 16; ++s) x[s] = f[s * 64];
            const float *db = a.d_base + blk * (8 * 64) + lane;
#pragma unroll
            for (int g = 0; g < 8; ++g) dob[g] = db[g * 64];
            if (!RECOMP) {
                const float *ac = a.acts + blk * ACT_SAVE_FLOATS_X + lane;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int g = 0; g < 16; ++g) h[t][g] = ac[(t * 16 + g) * 64];
            }
        }
        bf16x8 bx[2][3];                                           // split x: chain operand (RECOMP) and source of x^T
        split8<NT>(x, bx[0]);
        split8<NT>(x + 8, bx[1]);
        if (RECOMP) {
            const float *B1 = b1 + zo;
            f32x16 ha[2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int g = 0; g < 16; ++g) ha[t][g] = B1[32 * t + rowc(g) + 4 * hi];
#pragma unroll
            for (int c = 0; c < 2; ++c) mma2<MODE>(ha[0], ha[1], fr + BL::F_W1, 2, c, bx[c], lane);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int g = 0; g < 16; ++g) h[t][g] = softplus100(ha[t][g]);
        }
        // ---- dW(base.wo)[it] += dO . h(it)^T   (dO: 16 real rows = ONE chain operand; rows 16..31 of the tile stay zero)
        bf16x8 bo[3];
        split8<NT>(dob, bo);
#pragma unroll
        for (int g4 = 0; g4 < 2; ++g4) {
            const float4 ab = accl[(8 + g4) * 64];
            accl[(8 + g4) * 64] = make_float4(ab.x + dob[4 * g4], ab.y + dob[4 * g4 + 1], ab.z + dob[4 * g4 + 2], ab.w + dob[4 * g4 + 3]);
        }
        {
            bf16x8 oT[2][3];
            transpose_half<NT>(bo, sel0, oT);
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                bf16x8 bh[2][3], hT[2][3];
                split8<NT>(h[it], bh[0]);
                split8<NT>(h[it] + 8, bh[1]);
                transpose_tile<NT>(bh[0], bh[1], sel0, sel1, hT);
                dw_block<MODE>(acc_w2[it], oT, hT);
            }
        }
        // ---- d h = Wo^T dO (one k-chunk: slot j -> base_out neuron rowc(j) + 4 hi) ; dz0 = d h * softplus'(h)
        f32x16 dh[2];
#pragma unroll
        for (int g = 0; g < 16; ++g) { dh[0][g] = 0.f; dh[1][g] = 0.f; }
        mma2<MODE>(dh[0], dh[1], fr + (flip ? BL::F_W2TN : F_W2T), 1, 0, bo, lane);      // flip: -d h
        const float sgn = flip ? -1.f : 1.f;
        float dz0[2][16];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 ab = accl[(t * 4 + g4) * 64];
                float b[4] = {ab.x, ab.y, ab.z, ab.w};
#pragma unroll
                for (int gi = 0; gi < 4; ++gi) {
                    const int g = 4 * g4 + gi;
                    // softplus'(beta = 100) through the output, times the block's sign: sgn (1 - exp(-100 h))
                    dz0[t][g] = dh[t][g] * fmaf(-sgn, __builtin_amdgcn_exp2f(h[t][g] * -144.26950408889634f), sgn);
                    b[gi] += dz0[t][g];
                }
                accl[(t * 4 + g4) * 64] = make_float4(b[0], b[1], b[2], b[3]);
            }
        // ---- dW(base.w0)[ot] += dz0(ot) . x^T ;  d x = W0^T dz0 -> hash-feature gradient, fragment layout
        f32x16 dxv, dxv2;
#pragma unroll
        for (int g = 0; g < 16; ++g) { dxv[g] = 0.f; dxv2[g] = 0.f; }
        {
            bf16x8 xT[2][3];
            transpose_tile<NT>(bx[0], bx[1], sel_x0, sel_x1, xT);
            bf16x8 bz[4][3];
#pragma unroll
            for (int t = 0; t < 2; ++t) { split8<NT>(dz0[t], bz[2 * t]); split8<NT>(dz0[t] + 8, bz[2 * t + 1]); }
            f32x16 tw[2];
#pragma unroll
            for (int ot = 0; ot < 2; ++ot) {
#pragma unroll
                for (int g = 0; g < 16; ++g) tw[ot][g] = 0.f;
                bf16x8 zT[2][3];
                transpose_tile<NT>(bz[2 * ot], bz[2 * ot + 1], sel0, sel1, zT);
                dw_acc<MODE>(tw[ot], zT, xT);
            }
            mma1x2<MODE>(dxv, dxv2, fr + F_W1T, 4, 0, 1, bz[0], bz[1], lane);
            mma1x2<MODE>(dxv, dxv2, fr + F_W1T, 4, 2, 3, bz[2], bz[3], lane);
#pragma unroll
            for (int g = 0; g < 16; ++g) { acc_w1[0][g] += tw[0][g]; acc_w1[1][g] += tw[1][g]; }
#pragma unroll
            for (int g = 0; g < 16; ++g) dxv[g] -= dxv2[g];
        }
        {
            float *df = a.dfeat + blk * (16 * 64) + sl;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int f = rowc(g) + 4 * hi;                    // feature index = 2*level + parity
                df[(f >> 1) * 64 + (f & 1) * 32] = dxv[g];
            }
        }
    }
    float *slab = a.slab + ((int64_t)blockIdx.x * 4 + wave) * P_BASE_N;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        const int out = rowc(g) + 4 * hi;
        if (out < 16) {
            slab[P_BWO + out * 64 + sl] = acc_w2[0][g];
            slab[P_BWO + out * 64 + 32 + sl] = acc_w2[1][g];
        }
#pragma unroll
        for (int ob = 0; ob < 2; ++ob) slab[P_BW0 + (32 * ob + out) * 32 + sl] = acc_w1[ob][g];
    }
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        float b2 = g < 8 ? reinterpret_cast<const float *>(accl + (8 + (g >> 2)) * 64)[g & 3] : 0.f, b10 = reinterpret_cast<const float *>(accl + (g >> 2) * 64)[g & 3],
              b11 = reinterpret_cast<const float *>(accl + (4 + (g >> 2)) * 64)[g & 3];
#pragma unroll
        for (int off = 1; off < 32; off <<= 1) {
            b2 += __shfl_xor(b2, off, 64); b10 += __shfl_xor(b10, off, 64); b11 += __shfl_xor(b11, off, 64);
        }
        if (sl == 0) {
            if (g < 8) slab[P_BBO + rowc(g) + 4 * hi] = b2;
            slab[P_BB0 + rowc(g) + 4 * hi] = b10;
            slab[P_BB0 + 32 + rowc(g) + 4 * hi] = b11;
        }
    }
}

// persistent workgroups of this launch: the call's `grid_cus` (the chunked backward leaves CUs to the scatter on the other
// stream); 0 or anything >= REN_GRID_CUS = all of them
static inline int bwd_grid_cus(int k) { return k >= 1 && k < REN_GRID_CUS ? k : REN_GRID_CUS; }

template <int MODE, bool RECOMP>
int launch_bwd_x(const BwdXHArgs &h, const BwdXBArgs &b, int C, float *grad, int grid_cus, hipStream_t st) {
    const int GRID_XH = bwd_grid_cus(grid_cus), GRID_XB = GRID_XH * REN_BASE_WAVES;      // (the slab layout stays that of the full grids)
    using HL = HeadLds<MODE, RECOMP>;
    using BL = BaseLds<MODE, RECOMP>;
    const size_t lds_h = (size_t)HL::F_END * 2 + HL::TAIL_BYTES + 4 * 16 * 64 * sizeof(float4);
    const size_t lds_b = (size_t)BL::F_END * 2 + BL::TAIL_BYTES + 4 * 10 * 64 * sizeof(float4);
    if (C == 1) {
        (void)hipFuncSetAttribute((const void *)mlp_bwd_head_x_kernel<1, MODE, RECOMP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_h);
        hipLaunchKernelGGL((mlp_bwd_head_x_kernel<1, MODE, RECOMP>), dim3(GRID_XH), dim3(256), lds_h, st, h);
    } else {
        (void)hipFuncSetAttribute((const void *)mlp_bwd_head_x_kernel<3, MODE, RECOMP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_h);
        hipLaunchKernelGGL((mlp_bwd_head_x_kernel<3, MODE, RECOMP>), dim3(GRID_XH), dim3(256), lds_h, st, h);
    }
    (void)hipFuncSetAttribute((const void *)mlp_bwd_base_x_kernel<MODE, RECOMP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b);
    hipLaunchKernelGGL((mlp_bwd_base_x_kernel<MODE, RECOMP>), dim3(GRID_XB), dim3(256), lds_b, st, b);
    const int head_len = p_total(C) - P_BASE_N;
    SlabSets sets;
    sets.n = 2;
    sets.s[0] = SlabSet{h.slab, grad + P_BASE_N, GRID_XH * 4, head_len, 0};
    sets.s[1] = SlabSet{b.slab, grad, GRID_XB * 4, P_BASE_N, 0};
    sets.s[2] = sets.s[1];
    launch_reduce_slab_sets(sets, st);
    REN_CHECK_LAUNCH();
}

}  // namespace

extern "C" int64_t ren_mlp_bwd_x_workspace_floats(int32_t C) {
    if (C != 1 && C != 3) return -1;
    return (int64_t)GRID_XH * 4 * (p_total(C) - P_BASE_N) + (int64_t)GRID_XB * 4 * P_BASE_N;
}

extern "C" int ren_mlp_bwd_x(const float *mlp_params, int32_t C, int32_t activations, int32_t mode, const float *feat, const float *base_out,
                             const float *act_save, const ren_scene_desc *scene, const float *x_world,
                             const float *dirs, const float *rays_o, const float *rays_d, const int32_t *ray_indices,
                             const float *t_starts, const float *t_ends, int64_t n, const float *rgb,
                             const float *d_rgb, const float *d_sigma, float *d_base, float *dfeat,
                             float *grad_mlp_params, float *workspace, int32_t grid_cus, const int64_t *n_dev, void *stream) {
    // act_save == nullptr: the hidden activations are recomputed from feat / base_out (the forward need not save them)
    if (!mlp_params || !feat || !base_out || !scene || !rgb || !d_rgb || !d_sigma || !d_base || !dfeat ||
        !grad_mlp_params || !workspace || n < 0)
        return REN_ERR_BAD_ARG;
    if (C != 1 && C != 3) return REN_ERR_UNSUPPORTED;
    if (mode != 1 && mode != 3 && mode != 6) return REN_ERR_UNSUPPORTED;
    if (activations != 0) return REN_ERR_UNSUPPORTED;      // activation alternatives: exact-f32 kernels only
    if (grid_cus < 0) return REN_ERR_BAD_ARG;
    if (!x_world && (!rays_o || !rays_d || !ray_indices || !t_starts || !t_ends)) return REN_ERR_BAD_ARG;
    if (n == 0) return REN_OK;
    const int head_len = p_total(C) - P_BASE_N;
    BwdXHArgs h;
    h.params = mlp_params; h.base_out = base_out; h.acts = act_save;
    h.src = SampleSrc{x_world, dirs, rays_o, rays_d, x_world ? nullptr : ray_indices, t_starts, t_ends};
    h.sc = ren_make_scene(scene);
    h.n = n; h.rgb = rgb; h.d_rgb = d_rgb; h.d_sigma = d_sigma; h.d_base = d_base; h.slab = workspace; h.n_dev = n_dev;
    BwdXBArgs b;
    b.params = mlp_params; b.feat = feat; b.d_base = d_base; b.acts = act_save; b.n = n; b.dfeat = dfeat; b.n_dev = n_dev;
    b.slab = workspace + (int64_t)GRID_XH * 4 * head_len;
    if (act_save)
        return mode == 3 ? launch_bwd_x<3, false>(h, b, C, grad_mlp_params, grid_cus, (hipStream_t)stream) :
               mode == 6 ? launch_bwd_x<6, false>(h, b, C, grad_mlp_params, grid_cus, (hipStream_t)stream)
                         : launch_bwd_x<1, false>(h, b, C, grad_mlp_params, grid_cus, (hipStream_t)stream);
    if (mode == 3) return launch_bwd_x<3, true>(h, b, C, grad_mlp_params, grid_cus, (hipStream_t)stream);
    return mode == 6 ? launch_bwd_x<6, true>(h, b, C, grad_mlp_params, grid_cus, (hipStream_t)stream)
                     : launch_bwd_x<1, true>(h, b, C, grad_mlp_params, grid_cus, (hipStream_t)stream);
}
