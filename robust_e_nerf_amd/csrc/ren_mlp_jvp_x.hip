// Fused NGP MLPs with a forward-mode tangent (value + d/dt) and the reverse pass over the (value, tangent) pair
// on the bf16 matrix cores: the tangent kernels of ren_mlp_jvp.hip with the operand scheme of ren_mlp_x.hip.
//
//   mode 6  split-bf16 products at fp32 accuracy (six v_mfma_f32_32x32x16_bf16 per 16-wide k-chunk)
//   mode 1  plain bf16 operands, fp32 accumulate: BASELINE configs[2] ("bf16 MLP with fp32 composite", l_grad on).
//           Value AND tangent operands of every nn.Linear are rounded to bf16, as are the operands of the backward
//           products (what a bf16 autocast of the reference's double-backward would feed the matrix cores);
//           biases, activations, their derivatives and all accumulations stay fp32.
//
// Per layer with input (a, ad):  z = W a + b, zd = W ad;  y = sp(z), yd = s zd with s = sp'(z) = 1 - exp(-beta y)
// recovered from the OUTPUT, s' = sp''(z) = beta (1 - s) s.  Backward given (dy, dyd):  dz = dy s + dyd zd s',
// dzd = dyd s;  dW += dz a^T + dzd ad^T, db += dz, da = W^T dz, dad = W^T dzd.  Value and tangent chains share
// every weight fragment read: one ds_read_b128 feeds two MFMAs with independent accumulators.
//
// Backward = three persistent kernels, as in ren_mlp_jvp.hip (register-resident dW tiles):
//   head2: output layer + head layer 1 (64->64) -> (dz1, dz1d)       [recomputes head layers 0 and 1]
//   head1: head layer 0 ([base_out|SH] -> 64) + density -> (d base_out, d base_outd)
//   base : base MLP -> (d feat, d featd)                              [recomputes the hidden layer]
// Weight gradients: two bf16 pieces per operand, three terms (mode 6) / one piece (mode 1), operands staged per wave
// as [piece][neuron][sample] and read back along the sample axis (k = sample), the value and the tangent product of
// a tile accumulated into the same registers.
#include "ren_mlp_xfrag.h"
#include "ren_mlp_jvp_common.h"

namespace {

// two output tiles, value and tangent B operands: four independent accumulators per fragment pair
template <int MODE>
__device__ __forceinline__ void mma2j(f32x16 &a0, f32x16 &a1, f32x16 &d0, f32x16 &d1, const __bf16 *frag, int chunks, int c,
                                      const bf16x8 (&b)[3], const bf16x8 (&bd)[3], int lane) {
    using PR = Pairs<MODE>;
#pragma unroll
    for (int k = 0; k < PR::N; ++k) {
        const bf16x8 w0 = ldfrag<PR::NT>(frag, 0, chunks, c, PR::W[k], lane);
        const bf16x8 w1 = ldfrag<PR::NT>(frag, 1, chunks, c, PR::W[k], lane);
        a0 = MFMAB(w0, b[PR::A[k]], a0);
        a1 = MFMAB(w1, b[PR::A[k]], a1);
        d0 = MFMAB(w0, bd[PR::A[k]], d0);
        d1 = MFMAB(w1, bd[PR::A[k]], d1);
    }
}
// one output tile, value and tangent
template <int MODE>
__device__ __forceinline__ void mma1j(f32x16 &a0, f32x16 &d0, const __bf16 *frag, int chunks, int c, const bf16x8 (&b)[3],
                                      const bf16x8 (&bd)[3], int lane) {
    using PR = Pairs<MODE>;
#pragma unroll
    for (int k = 0; k < PR::N; ++k) {
        const bf16x8 w0 = ldfrag<PR::NT>(frag, 0, chunks, c, PR::W[k], lane);
        a0 = MFMAB(w0, b[PR::A[k]], a0);
        d0 = MFMAB(w0, bd[PR::A[k]], d0);
    }
}

// softplus(beta = 100) with tangent, keeping the pre-activation tangent: z -> y, zd stays, yd = s zd
__device__ __forceinline__ void act100_jvp(const f32x16 &z, const f32x16 &zd, float *y, float *yd) {
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        y[g] = softplus100(z[g]);
        yd[g] = dsoftplus_from_out(y[g], 100.f) * zd[g];
    }
}

// through softplus(beta = 100) backwards: (dy, dyd) -> (dz, dzd), given the output y and the pre-activation tangent zd
__device__ __forceinline__ void act100_bwd(float y, float zd, float dy, float dyd, float &dz, float &dzd) {
    const float s = dsoftplus_from_out(y, 100.f);
    dz = dy * s + dyd * zd * d2softplus_from_s(s, 100.f);
    dzd = dyd * s;
}

// ================================================================================================ forward
struct FwdJXArgs {
    const float *params, *feat, *featd;
    RaySrc src;
    ren_scene_dev sc;
    int64_t n;
    float *rgb, *rgbd, *sigma, *sigmad, *base_out, *base_outd;
    const int64_t *n_dev;                                     // device-side sample count (ren_eff_n) or NULL
};

// LDS image: forward fragments of the four hidden/base layers + f32 tail (same as XL in ren_mlp_x.hip)
template <int NT> struct JL {
    static constexpr int F_W1 = 0;                              // 2 tiles x 2 chunks
    static constexpr int F_W2 = F_W1 + 2 * 2 * NT * 512;        // 1 x 4
    static constexpr int F_WH1 = F_W2 + 1 * 4 * NT * 512;       // 2 x 2
    static constexpr int F_WH2 = F_WH1 + 2 * 2 * NT * 512;      // 2 x 4
    static constexpr int F_END = F_WH2 + 2 * 4 * NT * 512;      // bf16 elements
    static constexpr int BYTES_F = F_END * 2;
    static constexpr int T_B1 = 0, T_B2 = 64, T_BH1 = 96, T_BH2 = 160, T_WH3 = 224, T_BH3 = 416, T_END = 420;
    static constexpr size_t BYTES = (size_t)BYTES_F + T_END * 4;
};

template <int C, int MODE>
__global__ __launch_bounds__(256, 2) void mlp_fwd_jvp_x_kernel(FwdJXArgs a) {
    using PR = Pairs<MODE>;
    constexpr int NT = PR::NT;
    using L = JL<NT>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __bf16 *frag = reinterpret_cast<__bf16 *>(smem);
    float *tail = reinterpret_cast<float *>(smem + L::BYTES_F);
    fill_frags<NT, 0>(frag + L::F_W1, a.params, 2, 2);
    fill_frags<NT, 1>(frag + L::F_W2, a.params, 1, 4);
    fill_frags<NT, 2>(frag + L::F_WH1, a.params, 2, 2);
    fill_frags<NT, 3>(frag + L::F_WH2, a.params, 2, 4);
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
    for (int64_t blk = (int64_t)blockIdx.x * 4 + wave; blk < n_blk; blk += (int64_t)gridDim.x * 4) {
        int zo = 0;                                             // keep the (loop-invariant) LDS reads inside the loop
        asm volatile("" : "+v"(zo));
        const __bf16 *fr = frag + zo;
        const float *tl = tail + zo;
        const int64_t i = blk * 32 + sl;
        const bool live = i < n_smp;
        // ---- hash features and their tangents -> two k-chunks each
        bf16x8 bx[2][3], bxd[2][3];
        {
            const float *f = a.feat + blk * (16 * 64) + lane, *fd = a.featd + blk * (16 * 64) + lane;
            float x[16], xd[16];
#pragma unroll
            for (int s = 0; s < 16; ++s) { x[s] = f[s * 64]; xd[s] = fd[s * 64]; }
            split8<NT>(x, bx[0]); split8<NT>(x + 8, bx[1]);
            split8<NT>(xd, bxd[0]); split8<NT>(xd + 8, bxd[1]);
        }
        // ---- base layer 0
        f32x16 h[2], hd[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 16; ++g) { h[t][g] = tl[L::T_B1 + 32 * t + rowc(g) + 4 * hi]; hd[t][g] = 0.f; }
#pragma unroll
        for (int c = 0; c < 2; ++c) mma2j<MODE>(h[0], h[1], hd[0], hd[1], fr + L::F_W1, 2, c, bx[c], bxd[c], lane);
        bf16x8 bh[4][3], bhd[4][3];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float y[16], yd[16];
            act100_jvp(h[t], hd[t], y, yd);
            split8<NT>(y, bh[2 * t]); split8<NT>(y + 8, bh[2 * t + 1]);
            split8<NT>(yd, bhd[2 * t]); split8<NT>(yd + 8, bhd[2 * t + 1]);
        }
        // ---- base output: 64 -> 16 (rows 16..31 of the tile are zero)
        f32x16 o, od;
#pragma unroll
        for (int g = 0; g < 16; ++g) { o[g] = tl[L::T_B2 + rowc(g) + 4 * hi]; od[g] = 0.f; }
#pragma unroll
        for (int c = 0; c < 4; ++c) mma1j<MODE>(o, od, fr + L::F_W2, 4, c, bh[c], bhd[c], lane);
        bool sel = false;
        float dir[3] = {0.f, 0.f, 1.f}, dird[3] = {0.f, 0.f, 0.f};
        if (live) geom_jvp(a.src, a.sc, i, sel, dir, dird);
        if (live && hi == 0) {
            a.sigma[i] = sel ? __expf(o[0] - 1.f) : 0.f;
            a.sigmad[i] = sel ? __expf(fminf(o[0] - 1.f, 15.f)) * od[0] : 0.f;
        }
        {
            float *bo = a.base_out + blk * (8 * 64) + lane, *bod = a.base_outd + blk * (8 * 64) + lane;
#pragma unroll
            for (int g = 0; g < 8; ++g) { bo[g * 64] = o[g]; bod[g * 64] = od[g]; }
        }
        // ---- head layer 0: [base_out(16) | SH(16)] -> 64
        bf16x8 bv[2][3], bvd[2][3];
        {
            float v0[8], v0d[8], shs[8], shd[8];
#pragma unroll
            for (int g = 0; g < 8; ++g) { v0[g] = o[g]; v0d[g] = od[g]; }
            sh4_jvp_select(dir[0], dir[1], dir[2], dird[0], dird[1], dird[2], hi, shs, shd);
            split8<NT>(v0, bv[0]); split8<NT>(shs, bv[1]);
            split8<NT>(v0d, bvd[0]); split8<NT>(shd, bvd[1]);
        }
        f32x16 p[2], pd[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 16; ++g) { p[t][g] = tl[L::T_BH1 + 32 * t + rowc(g) + 4 * hi]; pd[t][g] = 0.f; }
#pragma unroll
        for (int c = 0; c < 2; ++c) mma2j<MODE>(p[0], p[1], pd[0], pd[1], fr + L::F_WH1, 2, c, bv[c], bvd[c], lane);
        bf16x8 bp[4][3], bpd[4][3];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float y[16], yd[16];
            act100_jvp(p[t], pd[t], y, yd);
            split8<NT>(y, bp[2 * t]); split8<NT>(y + 8, bp[2 * t + 1]);
            split8<NT>(yd, bpd[2 * t]); split8<NT>(yd + 8, bpd[2 * t + 1]);
        }
        // ---- head layer 1: 64 -> 64
        f32x16 q[2], qd[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 16; ++g) { q[t][g] = tl[L::T_BH2 + 32 * t + rowc(g) + 4 * hi]; qd[t][g] = 0.f; }
#pragma unroll
        for (int c = 0; c < 4; ++c) mma2j<MODE>(q[0], q[1], qd[0], qd[1], fr + L::F_WH2, 4, c, bp[c], bpd[c], lane);
        // ---- head output: 64 -> C on the VALU in fp32 (MODE 1: bf16-rounded operands, as every other layer)
        float acc[C], accd[C];
#pragma unroll
        for (int c = 0; c < C; ++c) { acc[c] = 0.f; accd[c] = 0.f; }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float y[16], yd[16];
            act100_jvp(q[t], qd[t], y, yd);
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const float qa = MODE == 1 ? (float)(__bf16)y[g] : y[g];
                const float qb = MODE == 1 ? (float)(__bf16)yd[g] : yd[g];
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    float w3 = tl[L::T_WH3 + c * 64 + 32 * t + rowc(g) + 4 * hi];
                    if (MODE == 1) w3 = (float)(__bf16)w3;
                    acc[c] += qa * w3;
                    accd[c] += qb * w3;
                }
            }
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float z3 = acc[c] + __shfl_xor(acc[c], 32, 64) + tl[L::T_BH3 + c];
            const float z3d = accd[c] + __shfl_xor(accd[c], 32, 64);
            if (hi == 0 && live) {
                const float y = softplus1(z3);
                a.rgb[i * C + c] = y;
                a.rgbd[i * C + c] = dsoftplus_from_out(y, 1.f) * z3d;
            }
        }
    }
}

// ================================================================================================ backward
constexpr int GRID_JX2 = 256, GRID_JX1 = 512, GRID_JXB = 256;  // persistent workgroups of 4 waves
constexpr int LEN_XH2 = 64 * 64 + 64;                          // head.w1 | head.b1  (+ 65 C for head.wo | head.bo)
constexpr int LEN_XH1 = 64 * 31 + 64;                          // head.w0 | head.b0
__host__ __device__ constexpr int len_xh2(int C) { return LEN_XH2 + 65 * C; }

// acc += dz(ot) . a(it)^T for the value pair staged in (Tz, Ta): one 32 x 32 tile, k = 32 samples
// (dw_tile of ren_mlp_xfrag.h); the tangent pair is restaged into the same tiles and accumulated on top.

// ---------------------------------------------------------------------------------- output + head layer 1
struct BwdJX2Args {
    const float *params, *base_out, *base_outd;
    RaySrc src;
    ren_scene_dev sc;
    int64_t n;
    const float *rgb, *d_rgb, *d_rgbd;
    float *dz1, *dz1d, *slab;                      // dz1/dz1d: [blk][2][16][64] fragment order
    const int64_t *n_dev;
};

template <int MODE> struct H2Lds {
    static constexpr int NT = Pairs<MODE>::NT, NP = Pairs<MODE>::NT;
    static constexpr int F_WH1 = 0, F_WH2 = F_WH1 + 2 * 2 * NT * 512, F_WH2T = F_WH2 + 2 * 4 * NT * 512;
    static constexpr int F_END = F_WH2T + 2 * 4 * NT * 512;
    static constexpr int T_W3 = 0, T_BH1 = 256, T_BH2 = 320, T_END = 384;
    static constexpr size_t TILE = (size_t)NP * 32 * ST * 2;
    static constexpr size_t BYTES = (size_t)F_END * 2 + T_END * 4 + 4 * 3 * TILE;
};

template <int C, int MODE>
__global__ __launch_bounds__(256, 1) void mlp_bwd_jvp_head2_x_kernel(BwdJX2Args a) {
    using PR = Pairs<MODE>;
    using HL = H2Lds<MODE>;
    constexpr int NT = PR::NT, NP = HL::NP;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __bf16 *frag = reinterpret_cast<__bf16 *>(smem);
    float *tail = reinterpret_cast<float *>(smem + HL::F_END * 2);
    fill_frags<NT, 2>(frag + HL::F_WH1, a.params, 2, 2);
    fill_frags<NT, 3>(frag + HL::F_WH2, a.params, 2, 4);
    fill_frags_t<NT, 3>(frag + HL::F_WH2T, a.params, 2, 4);                        // head.w1^T : rows = p index
    for (int i = threadIdx.x; i < 64; i += blockDim.x) {
        tail[HL::T_BH1 + i] = a.params[P_HB0 + i];
        tail[HL::T_BH2 + i] = a.params[P_HB1 + i];
    }
    for (int i = threadIdx.x; i < 64 * C; i += blockDim.x) {
        const float w = a.params[P_HWO + i];
        tail[HL::T_W3 + i] = MODE == 1 ? (float)(__bf16)w : w;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hi = lane >> 5, sl = lane & 31;
    __bf16 *Tz = reinterpret_cast<__bf16 *>(smem + HL::F_END * 2 + HL::T_END * 4) + wave * (3 * NP * 32 * ST);
    __bf16 *Ta = Tz + NP * 32 * ST, *Ta2 = Ta + NP * 32 * ST;
    __syncthreads();
    const int64_t n_smp = ren_eff_n(a.n, a.n_dev), n_blk = (n_smp + 31) >> 5;
    f32x16 acc_w[2][2];
    float acc_w3[C][32], acc_b2[2][16], acc_b3[C];
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        acc_w[0][0][g] = 0.f; acc_w[0][1][g] = 0.f; acc_w[1][0][g] = 0.f; acc_w[1][1][g] = 0.f;
        acc_b2[0][g] = 0.f; acc_b2[1][g] = 0.f;
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
        acc_b3[c] = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) acc_w3[c][k] = 0.f;
    }
    for (int64_t blk = (int64_t)blockIdx.x * 4 + wave; blk < n_blk; blk += (int64_t)gridDim.x * 4) {
        int zo = 0;
        asm volatile("" : "+v"(zo));
        const __bf16 *fr = frag + zo;
        const float *tl = tail + zo;
        const int64_t i = blk * 32 + sl;
        const bool live = i < n_smp;
        bool sel = false;
        float dir[3] = {0.f, 0.f, 1.f}, dird[3] = {0.f, 0.f, 0.f};
        if (live) geom_jvp(a.src, a.sc, i, sel, dir, dird);
        // ---- recompute head layer 0: p = sp(z1), keep z1d (pre-activation tangent), pd = s1 z1d
        bf16x8 bv[2][3], bvd[2][3];
        {
            float shs[8], shd[8], o[8], od[8];
            sh4_jvp_select(dir[0], dir[1], dir[2], dird[0], dird[1], dird[2], hi, shs, shd);
            const float *bo = a.base_out + blk * (8 * 64) + lane, *bod = a.base_outd + blk * (8 * 64) + lane;
#pragma unroll
            for (int g = 0; g < 8; ++g) { o[g] = bo[g * 64]; od[g] = bod[g * 64]; }
            split8<NT>(o, bv[0]); split8<NT>(shs, bv[1]);
            split8<NT>(od, bvd[0]); split8<NT>(shd, bvd[1]);
        }
        f32x16 z1[2], z1d[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 16; ++g) { z1[t][g] = tl[HL::T_BH1 + 32 * t + rowc(g) + 4 * hi]; z1d[t][g] = 0.f; }
#pragma unroll
        for (int c = 0; c < 2; ++c) mma2j<MODE>(z1[0], z1[1], z1d[0], z1d[1], fr + HL::F_WH1, 2, c, bv[c], bvd[c], lane);
        float p[2][16];
        bf16x8 bp[4][3], bpd[4][3];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float yd[16];
            act100_jvp(z1[t], z1d[t], p[t], yd);
            split8<NT>(p[t], bp[2 * t]); split8<NT>(p[t] + 8, bp[2 * t + 1]);
            split8<NT>(yd, bpd[2 * t]); split8<NT>(yd + 8, bpd[2 * t + 1]);
        }
        // ---- recompute head layer 1: q = sp(z2), keep z2d
        f32x16 z2[2], z2d[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 16; ++g) { z2[t][g] = tl[HL::T_BH2 + 32 * t + rowc(g) + 4 * hi]; z2d[t][g] = 0.f; }
#pragma unroll
        for (int c = 0; c < 4; ++c) mma2j<MODE>(z2[0], z2[1], z2d[0], z2d[1], fr + HL::F_WH2, 4, c, bp[c], bpd[c], lane);
        float q[2][16], qd[2][16];
#pragma unroll
        for (int t = 0; t < 2; ++t) act100_jvp(z2[t], z2d[t], q[t], qd[t]);
        // ---- output layer: z3d = sum_n qd[n] w3[n]
        float z3d[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float s = 0.f;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int g = 0; g < 16; ++g)
                    s += (MODE == 1 ? (float)(__bf16)qd[t][g] : qd[t][g]) * tl[HL::T_W3 + c * 64 + 32 * t + rowc(g) + 4 * hi];
            z3d[c] = s + __shfl_xor(s, 32, 64);
        }
        float dz3[C], dz3d[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float y = live ? a.rgb[i * C + c] : 0.f;
            const float s3 = dsoftplus_from_out(y, 1.f);
            const float gy = live ? a.d_rgb[i * C + c] : 0.f, gyd = live ? a.d_rgbd[i * C + c] : 0.f;
            dz3[c] = gy * s3 + gyd * z3d[c] * d2softplus_from_s(s3, 1.f);
            dz3d[c] = gyd * s3;
            if (hi == 0) acc_b3[c] += dz3[c];
        }
        // ---- d q, d qd -> dz2, dz2d ; dW3 (needs q and qd)
        float dz2[2][16], dz2d[2][16];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const float qv = MODE == 1 ? (float)(__bf16)q[t][g] : q[t][g];
                const float qdv = MODE == 1 ? (float)(__bf16)qd[t][g] : qd[t][g];
                float dq = 0.f, dqd = 0.f;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const float w3 = tl[HL::T_W3 + c * 64 + 32 * t + rowc(g) + 4 * hi];
                    dq += dz3[c] * w3; dqd += dz3d[c] * w3;
                    acc_w3[c][t * 16 + g] += dz3[c] * qv + dz3d[c] * qdv;
                }
                act100_bwd(q[t][g], z2d[t][g], dq, dqd, dz2[t][g], dz2d[t][g]);
                acc_b2[t][g] += dz2[t][g];
            }
        // ---- dW(head.w1)[ot][it] += dz2(ot) p(it)^T + dz2d(ot) pd(it)^T ;  d p = W1^T dz2, d pd = W1^T dz2d
        f32x16 dp[2], dpd[2];
#pragma unroll
        for (int g = 0; g < 16; ++g) { dp[0][g] = 0.f; dp[1][g] = 0.f; dpd[0][g] = 0.f; dpd[1][g] = 0.f; }
        // value pair first, then the tangent pair: each operand set is split, staged and consumed before the next one is
        // formed (all four at once do not fit the register file next to the 130 accumulator registers)
        stage_pieces<NP>(Ta, bp[0], bp[1], hi, sl);
        stage_pieces<NP>(Ta2, bp[2], bp[3], hi, sl);
        {
            bf16x8 bz[4][3];
#pragma unroll
            for (int t = 0; t < 2; ++t) { split8<NT>(dz2[t], bz[2 * t]); split8<NT>(dz2[t] + 8, bz[2 * t + 1]); }
#pragma unroll
            for (int ot = 0; ot < 2; ++ot) {
                stage_pieces<NP>(Tz, bz[2 * ot], bz[2 * ot + 1], hi, sl);
                dw_tile<NP>(acc_w[ot][0], Tz, Ta, hi, sl);
                dw_tile<NP>(acc_w[ot][1], Tz, Ta2, hi, sl);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) mma2<MODE>(dp[0], dp[1], fr + HL::F_WH2T, 4, c, bz[c], lane);
        }
        stage_pieces<NP>(Ta, bpd[0], bpd[1], hi, sl);
        stage_pieces<NP>(Ta2, bpd[2], bpd[3], hi, sl);
        {
            bf16x8 bzd[4][3];
#pragma unroll
            for (int t = 0; t < 2; ++t) { split8<NT>(dz2d[t], bzd[2 * t]); split8<NT>(dz2d[t] + 8, bzd[2 * t + 1]); }
#pragma unroll
            for (int ot = 0; ot < 2; ++ot) {
                stage_pieces<NP>(Tz, bzd[2 * ot], bzd[2 * ot + 1], hi, sl);
                dw_tile<NP>(acc_w[ot][0], Tz, Ta, hi, sl);
                dw_tile<NP>(acc_w[ot][1], Tz, Ta2, hi, sl);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) mma2<MODE>(dpd[0], dpd[1], fr + HL::F_WH2T, 4, c, bzd[c], lane);
        }
        // ---- through the activation of head layer 0 -> dz1, dz1d (to HBM, fragment order)
        {
            float *oz = a.dz1 + blk * (32 * 64) + lane, *ozd = a.dz1d + blk * (32 * 64) + lane;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    float dz, dzd;
                    act100_bwd(p[t][g], z1d[t][g], dp[t][g], dpd[t][g], dz, dzd);
                    oz[(t * 16 + g) * 64] = dz;
                    ozd[(t * 16 + g) * 64] = dzd;
                }
        }
    }
    // ---- slab: [head.w1 | head.b1 | head.wo | head.bo]
    float *slab = a.slab + ((int64_t)blockIdx.x * 4 + wave) * len_xh2(C);
#pragma unroll
    for (int ob = 0; ob < 2; ++ob) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int out = 32 * ob + rowc(g) + 4 * hi;
            slab[out * 64 + sl] = acc_w[ob][0][g];
            slab[out * 64 + 32 + sl] = acc_w[ob][1][g];
            float b2 = acc_b2[ob][g];
#pragma unroll
            for (int off = 1; off < 32; off <<= 1) b2 += __shfl_xor(b2, off, 64);
            if (sl == 0) slab[64 * 64 + out] = b2;
        }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            float v = acc_w3[c][k];
#pragma unroll
            for (int off = 1; off < 32; off <<= 1) v += __shfl_xor(v, off, 64);
            if (sl == 0) slab[LEN_XH2 + c * 64 + 32 * (k >> 4) + rowc(k & 15) + 4 * hi] = v;
        }
        const float b3 = ren_wave_sum(acc_b3[c]);
        if (lane == 0) slab[LEN_XH2 + 64 * C + c] = b3;
    }
}

// ---------------------------------------------------------------------------------- head layer 0 + density
struct BwdJX1Args {
    const float *params, *base_out, *base_outd, *dz1, *dz1d;
    RaySrc src;
    ren_scene_dev sc;
    int64_t n;
    const float *d_sigma, *d_sigmad;
    float *d_base, *d_based, *slab;
    const int64_t *n_dev;
};

template <int MODE> struct H1Lds {
    static constexpr int NT = Pairs<MODE>::NT, NP = Pairs<MODE>::NT;
    static constexpr int F_WH1T = 0, F_END = 1 * 4 * NT * 512;
    static constexpr size_t TILE = (size_t)NP * 32 * ST * 2;
    static constexpr size_t BYTES = (size_t)F_END * 2 + 4 * 2 * TILE;
};

template <int MODE>
__global__ __launch_bounds__(256, 2) void mlp_bwd_jvp_head1_x_kernel(BwdJX1Args a) {
    using PR = Pairs<MODE>;
    using HL = H1Lds<MODE>;
    constexpr int NT = PR::NT, NP = HL::NP;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __bf16 *frag = reinterpret_cast<__bf16 *>(smem);
    fill_frags_t<NT, 2>(frag + HL::F_WH1T, a.params, 1, 4);                        // head.w0^T : rows = v index
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hi = lane >> 5, sl = lane & 31;
    __bf16 *Tz = reinterpret_cast<__bf16 *>(smem + HL::F_END * 2) + wave * (2 * NP * 32 * ST);
    __bf16 *Ta = Tz + NP * 32 * ST;
    __syncthreads();
    const int64_t n_smp = ren_eff_n(a.n, a.n_dev), n_blk = (n_smp + 31) >> 5;
    f32x16 acc_w[2];
    float acc_b[2][16];
#pragma unroll
    for (int g = 0; g < 16; ++g) { acc_w[0][g] = 0.f; acc_w[1][g] = 0.f; acc_b[0][g] = 0.f; acc_b[1][g] = 0.f; }
    for (int64_t blk = (int64_t)blockIdx.x * 4 + wave; blk < n_blk; blk += (int64_t)gridDim.x * 4) {
        int zo = 0;
        asm volatile("" : "+v"(zo));
        const __bf16 *fr = frag + zo;
        const int64_t i = blk * 32 + sl;
        const bool live = i < n_smp;
        bool sel = false;
        float dir[3] = {0.f, 0.f, 1.f}, dird[3] = {0.f, 0.f, 0.f};
        if (live) geom_jvp(a.src, a.sc, i, sel, dir, dird);
        float shs[8], shd[8], o[8], od[8], dz[2][16], dzd[2][16];
        sh4_jvp_select(dir[0], dir[1], dir[2], dird[0], dird[1], dird[2], hi, shs, shd);
        {
            const float *bo = a.base_out + blk * (8 * 64) + lane, *bod = a.base_outd + blk * (8 * 64) + lane;
#pragma unroll
            for (int g = 0; g < 8; ++g) { o[g] = bo[g * 64]; od[g] = bod[g * 64]; }
            const float *iz = a.dz1 + blk * (32 * 64) + lane, *izd = a.dz1d + blk * (32 * 64) + lane;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    dz[t][g] = iz[(t * 16 + g) * 64]; dzd[t][g] = izd[(t * 16 + g) * 64];
                    acc_b[t][g] += dz[t][g];
                }
        }
        f32x16 dv, dvd;
#pragma unroll
        for (int g = 0; g < 16; ++g) { dv[g] = 0.f; dvd[g] = 0.f; }
        // ---- dW(head.w0)[ot] += dz1(ot) v^T + dz1d(ot) vd^T,  v = [base_out(16) | SH(16)] in v order;
        //      d v = W0^T dz1 (rows 0..15 = d base_out), d vd = W0^T dz1d.  Value set first, then the tangent set.
        {
            bf16x8 bz[4][3];
#pragma unroll
            for (int t = 0; t < 2; ++t) { split8<NT>(dz[t], bz[2 * t]); split8<NT>(dz[t] + 8, bz[2 * t + 1]); }
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                stage_one<NP>(Ta, rowc(g) + 4 * hi, o[g], sl);
                stage_one<NP>(Ta, 16 + 2 * g + hi, shs[g], sl);
            }
#pragma unroll
            for (int ot = 0; ot < 2; ++ot) {
                stage_pieces<NP>(Tz, bz[2 * ot], bz[2 * ot + 1], hi, sl);
                dw_tile<NP>(acc_w[ot], Tz, Ta, hi, sl);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) mma<MODE>(dv, fr + HL::F_WH1T, 0, 4, c, bz[c], lane);
        }
        {
            bf16x8 bzd[4][3];
#pragma unroll
            for (int t = 0; t < 2; ++t) { split8<NT>(dzd[t], bzd[2 * t]); split8<NT>(dzd[t] + 8, bzd[2 * t + 1]); }
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                stage_one<NP>(Ta, rowc(g) + 4 * hi, od[g], sl);
                stage_one<NP>(Ta, 16 + 2 * g + hi, shd[g], sl);
            }
#pragma unroll
            for (int ot = 0; ot < 2; ++ot) {
                stage_pieces<NP>(Tz, bzd[2 * ot], bzd[2 * ot + 1], hi, sl);
                dw_tile<NP>(acc_w[ot], Tz, Ta, hi, sl);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) mma<MODE>(dvd, fr + HL::F_WH1T, 0, 4, c, bzd[c], lane);
        }
        if (hi == 0) {
            // sigma = e sel, sigmad = e sel od0 with e = exp(min(o0 - 1, 15)):
            // d o0 = d sigma e + d sigmad sigmad (unclamped branch), d od0 = d sigmad e
            const float ds = live ? a.d_sigma[i] : 0.f, dsd = live ? a.d_sigmad[i] : 0.f;
            const float e = sel ? __expf(fminf(o[0] - 1.f, 15.f)) : 0.f;
            dv[0] = ds * e + ((o[0] - 1.f) < 15.f ? dsd * e * od[0] : 0.f);
            dvd[0] = dsd * e;
        }
        {
            float *db = a.d_base + blk * (8 * 64) + lane, *dbd = a.d_based + blk * (8 * 64) + lane;
#pragma unroll
            for (int g = 0; g < 8; ++g) { db[g * 64] = dv[g]; dbd[g * 64] = dvd[g]; }
        }
    }
    float *slab = a.slab + ((int64_t)blockIdx.x * 4 + wave) * LEN_XH1;          // [head.w0 | head.b0]
#pragma unroll
    for (int ob = 0; ob < 2; ++ob) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int out = 32 * ob + rowc(g) + 4 * hi;
            if (sl != 0) slab[out * 31 + (sl < 16 ? 15 + sl : sl - 16)] = acc_w[ob][g];
            float b1 = acc_b[ob][g];
#pragma unroll
            for (int off = 1; off < 32; off <<= 1) b1 += __shfl_xor(b1, off, 64);
            if (sl == 0) slab[64 * 31 + out] = b1;
        }
    }
}

// ---------------------------------------------------------------------------------- base MLP
struct BwdJXBArgs {
    const float *params, *feat, *featd, *d_base, *d_based;
    int64_t n;
    float *dfeat, *dfeatd, *slab;
    const int64_t *n_dev;
};

template <int MODE> struct BJLds {
    static constexpr int NT = Pairs<MODE>::NT, NP = Pairs<MODE>::NT;
    static constexpr int F_W1 = 0, F_W2T = F_W1 + 2 * 2 * NT * 512, F_W1T = F_W2T + 2 * 1 * NT * 512;
    static constexpr int F_END = F_W1T + 1 * 4 * NT * 512;
    static constexpr size_t TILE = (size_t)NP * 32 * ST * 2;
    static constexpr size_t BYTES = (size_t)F_END * 2 + 64 * 4 + 4 * 2 * TILE;
};

template <int MODE>
__global__ __launch_bounds__(256, 1) void mlp_bwd_jvp_base_x_kernel(BwdJXBArgs a) {
    using PR = Pairs<MODE>;
    using BL = BJLds<MODE>;
    constexpr int NT = PR::NT, NP = BL::NP;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __bf16 *frag = reinterpret_cast<__bf16 *>(smem);
    float *b1 = reinterpret_cast<float *>(smem + BL::F_END * 2);
    fill_frags<NT, 0>(frag + BL::F_W1, a.params, 2, 2);                             // forward fragments of base.w0
    fill_frags_t<NT, 1>(frag + BL::F_W2T, a.params, 2, 1);                          // base.wo^T : rows = h index, 1 chunk
    fill_frags_t<NT, 0>(frag + BL::F_W1T, a.params, 1, 4);                          // base.w0^T : rows = feature index
    for (int i = threadIdx.x; i < 64; i += blockDim.x) b1[i] = a.params[P_BB0 + i];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hi = lane >> 5, sl = lane & 31;
    __bf16 *Tz = reinterpret_cast<__bf16 *>(smem + BL::F_END * 2 + 64 * 4) + wave * (2 * NP * 32 * ST);
    __bf16 *Ta = Tz + NP * 32 * ST;
    for (int k = lane; k < 2 * NP * 32 * ST; k += 64) Tz[k] = (__bf16)0.f;          // dO rows 16..31 stay zero
    __syncthreads();
    const int64_t n_smp = ren_eff_n(a.n, a.n_dev), n_blk = (n_smp + 31) >> 5;
    f32x16 acc_w2[2], acc_w1[2];
    float acc_b2[8], acc_b1[2][16];
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        acc_w2[0][g] = 0.f; acc_w2[1][g] = 0.f; acc_w1[0][g] = 0.f; acc_w1[1][g] = 0.f;
        acc_b1[0][g] = 0.f; acc_b1[1][g] = 0.f;
        if (g < 8) acc_b2[g] = 0.f;
    }
    for (int64_t blk = (int64_t)blockIdx.x * 4 + wave; blk < n_blk; blk += (int64_t)gridDim.x * 4) {
        int zo = 0;
        asm volatile("" : "+v"(zo));
        const __bf16 *fr = frag + zo;
        const float *B1 = b1 + zo;
        float dob[8], dobd[8];
        bf16x8 bx[2][3], bxd[2][3];
        {
            float x[16], xd[16];
            const float *f = a.feat + blk * (16 * 64) + lane, *fd = a.featd + blk * (16 * 64) + lane;
#pragma unroll
            for (int s = 0; s < 16; ++s) { x[s] = f[s * 64]; xd[s] = fd[s * 64]; }
            const float *db = a.d_base + blk * (8 * 64) + lane, *dbd = a.d_based + blk * (8 * 64) + lane;
#pragma unroll
            for (int g = 0; g < 8; ++g) { dob[g] = db[g * 64]; dobd[g] = dbd[g * 64]; acc_b2[g] += dob[g]; }
            split8<NT>(x, bx[0]); split8<NT>(x + 8, bx[1]);
            split8<NT>(xd, bxd[0]); split8<NT>(xd + 8, bxd[1]);
        }
        // ---- recompute the hidden layer: h = sp(z0), keep z0d; hd = s z0d
        f32x16 z0[2], z0d[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 16; ++g) { z0[t][g] = B1[32 * t + rowc(g) + 4 * hi]; z0d[t][g] = 0.f; }
#pragma unroll
        for (int c = 0; c < 2; ++c) mma2j<MODE>(z0[0], z0[1], z0d[0], z0d[1], fr + BL::F_W1, 2, c, bx[c], bxd[c], lane);
        float h[2][16], hd[2][16];
#pragma unroll
        for (int t = 0; t < 2; ++t) act100_jvp(z0[t], z0d[t], h[t], hd[t]);
        // ---- dW(base.wo)[it] += dO h(it)^T + dOd hd(it)^T   (dO: 16 real rows of a 32-row tile)
#pragma unroll
        for (int g = 0; g < 8; ++g) stage_one<NP>(Tz, rowc(g) + 4 * hi, dob[g], sl);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            stage_tile<NP>(Ta, h[it], hi, sl);
            dw_tile<NP>(acc_w2[it], Tz, Ta, hi, sl);
        }
#pragma unroll
        for (int g = 0; g < 8; ++g) stage_one<NP>(Tz, rowc(g) + 4 * hi, dobd[g], sl);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            stage_tile<NP>(Ta, hd[it], hi, sl);
            dw_tile<NP>(acc_w2[it], Tz, Ta, hi, sl);
        }
        // ---- d h = Wo^T dO, d hd = Wo^T dOd (one k-chunk: slot j -> base_out neuron rowc(j) + 4 hi)
        f32x16 dh[2], dhd[2];
#pragma unroll
        for (int g = 0; g < 16; ++g) { dh[0][g] = 0.f; dh[1][g] = 0.f; dhd[0][g] = 0.f; dhd[1][g] = 0.f; }
        {
            bf16x8 bo[3], bod[3];
            split8<NT>(dob, bo);
            split8<NT>(dobd, bod);
            mma2j<MODE>(dh[0], dh[1], dhd[0], dhd[1], fr + BL::F_W2T, 1, 0, bo, bod, lane);
        }
        float dz0[2][16], dz0d[2][16];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                act100_bwd(h[t][g], z0d[t][g], dh[t][g], dhd[t][g], dz0[t][g], dz0d[t][g]);
                acc_b1[t][g] += dz0[t][g];
            }
        // ---- dW(base.w0)[ot] += dz0(ot) x^T + dz0d(ot) xd^T ;  d x = W0^T dz0, d xd = W0^T dz0d
        f32x16 dxv, dxdv;
#pragma unroll
        for (int g = 0; g < 16; ++g) { dxv[g] = 0.f; dxdv[g] = 0.f; }
        {
            bf16x8 bz[4][3], bzd[4][3];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                split8<NT>(dz0[t], bz[2 * t]); split8<NT>(dz0[t] + 8, bz[2 * t + 1]);
                split8<NT>(dz0d[t], bzd[2 * t]); split8<NT>(dz0d[t] + 8, bzd[2 * t + 1]);
            }
#pragma unroll
            for (int k = 0; k < NP; ++k)
#pragma unroll
                for (int s = 0; s < 16; ++s) Ta[(k * 32 + 2 * s + hi) * ST + sl] = bx[s >> 3][k][s & 7];
#pragma unroll
            for (int ot = 0; ot < 2; ++ot) {
                // Tz rows 16..31 are overwritten here; the dO staging of the next block rewrites rows 0..15 only,
                // so the zero rows are restored below
                stage_pieces<NP>(Tz, bz[2 * ot], bz[2 * ot + 1], hi, sl);
                dw_tile<NP>(acc_w1[ot], Tz, Ta, hi, sl);
            }
#pragma unroll
            for (int k = 0; k < NP; ++k)
#pragma unroll
                for (int s = 0; s < 16; ++s) Ta[(k * 32 + 2 * s + hi) * ST + sl] = bxd[s >> 3][k][s & 7];
#pragma unroll
            for (int ot = 0; ot < 2; ++ot) {
                stage_pieces<NP>(Tz, bzd[2 * ot], bzd[2 * ot + 1], hi, sl);
                dw_tile<NP>(acc_w1[ot], Tz, Ta, hi, sl);
            }
#pragma unroll
            for (int g = 8; g < 16; ++g) stage_one<NP>(Tz, rowc(g) + 4 * hi, 0.f, sl);
#pragma unroll
            for (int c = 0; c < 4; ++c) mma1j<MODE>(dxv, dxdv, fr + BL::F_W1T, 4, c, bz[c], bzd[c], lane);
        }
        {
            float *df = a.dfeat + blk * (16 * 64) + sl, *dfd = a.dfeatd + blk * (16 * 64) + sl;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int f = rowc(g) + 4 * hi;                    // feature index = 2*level + parity
                df[(f >> 1) * 64 + (f & 1) * 32] = dxv[g];
                dfd[(f >> 1) * 64 + (f & 1) * 32] = dxdv[g];
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
        float b2 = g < 8 ? acc_b2[g] : 0.f, b10 = acc_b1[0][g], b11 = acc_b1[1][g];
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

template <int MODE>
int launch_fwd_jvp_x(const FwdJXArgs &a, int C, hipStream_t st) {
    using L = JL<Pairs<MODE>::NT>;
    int64_t blocks = ((a.n + 31) / 32 + 3) / 4;
    if (blocks > 512) blocks = 512;
    const size_t lds = L::BYTES;
    if (C == 1) {
        (void)hipFuncSetAttribute((const void *)mlp_fwd_jvp_x_kernel<1, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((mlp_fwd_jvp_x_kernel<1, MODE>), dim3((int)blocks), dim3(256), lds, st, a);
    } else {
        (void)hipFuncSetAttribute((const void *)mlp_fwd_jvp_x_kernel<3, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((mlp_fwd_jvp_x_kernel<3, MODE>), dim3((int)blocks), dim3(256), lds, st, a);
    }
    REN_CHECK_LAUNCH();
}

template <int MODE>
int launch_bwd_jvp_x(const BwdJX2Args &a2, const BwdJX1Args &a1, const BwdJXBArgs &ab, int C, float *grad, hipStream_t st) {
    const size_t l2 = H2Lds<MODE>::BYTES, l1 = H1Lds<MODE>::BYTES, lb = BJLds<MODE>::BYTES;
    if (C == 1) {
        (void)hipFuncSetAttribute((const void *)mlp_bwd_jvp_head2_x_kernel<1, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2);
        hipLaunchKernelGGL((mlp_bwd_jvp_head2_x_kernel<1, MODE>), dim3(GRID_JX2), dim3(256), l2, st, a2);
    } else {
        (void)hipFuncSetAttribute((const void *)mlp_bwd_jvp_head2_x_kernel<3, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2);
        hipLaunchKernelGGL((mlp_bwd_jvp_head2_x_kernel<3, MODE>), dim3(GRID_JX2), dim3(256), l2, st, a2);
    }
    (void)hipFuncSetAttribute((const void *)mlp_bwd_jvp_head1_x_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l1);
    hipLaunchKernelGGL((mlp_bwd_jvp_head1_x_kernel<MODE>), dim3(GRID_JX1), dim3(256), l1, st, a1);
    (void)hipFuncSetAttribute((const void *)mlp_bwd_jvp_base_x_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb);
    hipLaunchKernelGGL((mlp_bwd_jvp_base_x_kernel<MODE>), dim3(GRID_JXB), dim3(256), lb, st, ab);
    SlabSets sets;
    sets.n = 3;
    sets.s[0] = SlabSet{a2.slab, grad + P_HW1, GRID_JX2 * 4, len_xh2(C), 0};
    sets.s[1] = SlabSet{a1.slab, grad + P_HW0, GRID_JX1 * 4, LEN_XH1, 0};
    sets.s[2] = SlabSet{ab.slab, grad, GRID_JXB * 4, P_BASE_N, 0};
    launch_reduce_slab_sets(sets, st);
    REN_CHECK_LAUNCH();
}

}  // namespace

// mode: 6 = split-bf16 at fp32 accuracy, 1 = plain bf16 operands (BASELINE configs[2]); otherwise as ren_mlp_fwd_jvp
extern "C" int ren_mlp_fwd_jvp_x(const float *mlp_params, int32_t C, int32_t activations, int32_t mode, const float *feat, const float *featd,
                                 const ren_scene_desc *scene, const float *rays_o, const float *rays_d,
                                 const float *rays_dd, const int32_t *ray_indices, const float *t_starts,
                                 const float *t_ends, int64_t n, float *rgb, float *rgbd, float *sigma,
                                 float *sigmad, float *base_out, float *base_outd, const int64_t *n_dev, void *stream) {
    if (!mlp_params || !feat || !featd || !scene || !rays_o || !rays_d || !rays_dd || !ray_indices || !t_starts ||
        !t_ends || !rgb || !rgbd || !sigma || !sigmad || !base_out || !base_outd || n < 0)
        return REN_ERR_BAD_ARG;
    if (C != 1 && C != 3) return REN_ERR_UNSUPPORTED;
    if (mode != 1 && mode != 3 && mode != 6) return REN_ERR_UNSUPPORTED;
    if (activations != 0) return REN_ERR_UNSUPPORTED;      // activation alternatives: exact-f32 kernels only
    if (n == 0) return REN_OK;
    FwdJXArgs a;
    a.n_dev = n_dev;
    a.params = mlp_params; a.feat = feat; a.featd = featd;
    a.src = RaySrc{rays_o, rays_d, rays_dd, ray_indices, t_starts, t_ends};
    a.sc = ren_make_scene(scene);
    a.n = n; a.rgb = rgb; a.rgbd = rgbd; a.sigma = sigma; a.sigmad = sigmad; a.base_out = base_out; a.base_outd = base_outd;
    if (mode == 3) return launch_fwd_jvp_x<3>(a, C, (hipStream_t)stream);
    return mode == 6 ? launch_fwd_jvp_x<6>(a, C, (hipStream_t)stream) : launch_fwd_jvp_x<1>(a, C, (hipStream_t)stream);
}

extern "C" int64_t ren_mlp_bwd_jvp_x_workspace_floats(int32_t C) {
    if (C != 1 && C != 3) return -1;
    return (int64_t)GRID_JX2 * 4 * len_xh2(C) + (int64_t)GRID_JX1 * 4 * LEN_XH1 + (int64_t)GRID_JXB * 4 * P_BASE_N;
}

// scratch: 5 120 floats per 32-sample block, as for ren_mlp_bwd_jvp
extern "C" int ren_mlp_bwd_jvp_x(const float *mlp_params, int32_t C, int32_t activations, int32_t mode, const float *feat, const float *featd,
                                 const float *base_out, const float *base_outd, const ren_scene_desc *scene,
                                 const float *rays_o, const float *rays_d, const float *rays_dd,
                                 const int32_t *ray_indices, const float *t_starts, const float *t_ends, int64_t n,
                                 const float *rgb, const float *d_rgb, const float *d_rgbd, const float *d_sigma,
                                 const float *d_sigmad, float *scratch, float *dfeat, float *dfeatd,
                                 float *grad_mlp_params, float *workspace, const int64_t *n_dev, void *stream) {
    if (!mlp_params || !feat || !featd || !base_out || !base_outd || !scene || !rays_o || !rays_d || !rays_dd ||
        !ray_indices || !t_starts || !t_ends || !rgb || !d_rgb || !d_rgbd || !d_sigma || !d_sigmad || !scratch ||
        !dfeat || !dfeatd || !grad_mlp_params || !workspace || n < 0)
        return REN_ERR_BAD_ARG;
    if (C != 1 && C != 3) return REN_ERR_UNSUPPORTED;
    if (mode != 1 && mode != 3 && mode != 6) return REN_ERR_UNSUPPORTED;
    if (activations != 0) return REN_ERR_UNSUPPORTED;      // activation alternatives: exact-f32 kernels only
    if (n == 0) return REN_OK;
    const int64_t n_blk = (n + 31) / 32;
    // scratch (floats): dz1 | dz1d (2048 per block each) | d_base | d_based (512 per block each)
    float *dz1 = scratch, *dz1d = dz1 + n_blk * 2048, *d_base = dz1d + n_blk * 2048, *d_based = d_base + n_blk * 512;
    float *slab2 = workspace, *slab1 = slab2 + (int64_t)GRID_JX2 * 4 * len_xh2(C), *slabb = slab1 + (int64_t)GRID_JX1 * 4 * LEN_XH1;
    const RaySrc src{rays_o, rays_d, rays_dd, ray_indices, t_starts, t_ends};
    const ren_scene_dev sc = ren_make_scene(scene);
    BwdJX2Args a2;
    a2.params = mlp_params; a2.base_out = base_out; a2.base_outd = base_outd; a2.src = src; a2.sc = sc; a2.n = n; a2.n_dev = n_dev;
    a2.rgb = rgb; a2.d_rgb = d_rgb; a2.d_rgbd = d_rgbd; a2.dz1 = dz1; a2.dz1d = dz1d; a2.slab = slab2;
    BwdJX1Args a1;
    a1.params = mlp_params; a1.base_out = base_out; a1.base_outd = base_outd; a1.dz1 = dz1; a1.dz1d = dz1d;
    a1.src = src; a1.sc = sc; a1.n = n; a1.n_dev = n_dev; a1.d_sigma = d_sigma; a1.d_sigmad = d_sigmad; a1.d_base = d_base;
    a1.d_based = d_based; a1.slab = slab1;
    BwdJXBArgs ab;
    ab.params = mlp_params; ab.feat = feat; ab.featd = featd; ab.d_base = d_base; ab.d_based = d_based; ab.n = n; ab.n_dev = n_dev;
    ab.dfeat = dfeat; ab.dfeatd = dfeatd; ab.slab = slabb;
    if (mode == 3) return launch_bwd_jvp_x<3>(a2, a1, ab, C, grad_mlp_params, (hipStream_t)stream);
    return mode == 6 ? launch_bwd_jvp_x<6>(a2, a1, ab, C, grad_mlp_params, (hipStream_t)stream)
                     : launch_bwd_jvp_x<1>(a2, a1, ab, C, grad_mlp_params, (hipStream_t)stream);
}
