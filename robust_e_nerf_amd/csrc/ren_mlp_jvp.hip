// Fused NGP MLPs with a forward-mode tangent (value + d/dt) and the reverse-mode pass over the
// (value, tangent) pair, for the log-intensity-gradient loss (see ren_jvp.hip for the rationale).
//
// Per layer with input (a, ad):  z = W a + b, zd = W ad;  y = sp(z), yd = s zd with s = sp'(z) =
// 1 - exp(-beta y) recovered from the OUTPUT, and s' = sp''(z) = beta (1 - s) s.
// Backward given (dy, dyd):  dz = dy s + dyd zd s',  dzd = dyd s;  dW += dz a^T + dzd ad^T, db += dz,
// da = W^T dz, dad = W^T dzd.  The tangent chain reuses every weight fragment of the value chain, so
// the forward costs 2x the MFMAs of ren_mlp_fwd and no extra LDS traffic.
//
// Backward is three persistent kernels so the register-resident dW tiles stay under 512 VGPRs:
//   head2: output layer + head layer 1 (64->64) -> (dz1, dz1d)      [recomputes head layers 0 and 1]
//   head1: head layer 0 ([base_out|SH] -> 64) + density -> (d base_out, d base_outd)
//   base : base MLP -> (d feat, d featd)                             [recomputes the hidden layer]
#include "ren_mlp_jvp_common.h"

namespace {

// activation with tangent, in place: z -> y = sp100(z), zd -> yd = s zd; optionally keeps zd (pre-activation tangent)
// (k: hidden-activation kind of the layer's MLP, ren_mlp_common.h act_kinds)
__device__ __forceinline__ void act_jvp(f32x16 (&y)[2], f32x16 (&yd)[2], int k) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const float v = act_hidden(y[r][g], k);
            yd[r][g] *= dact_hidden(v, k);
            y[r][g] = v;
        }
}

// ============================================================================ forward with tangent
struct FwdJArgs {
    const float *params, *feat, *featd;
    RaySrc src;
    ren_scene_dev sc;
    int64_t n;
    float *rgb, *rgbd, *sigma, *sigmad, *base_out, *base_outd;
    int act_code;                                  // activation alternatives (ren_mlp_common.h); 0 = shipped configs
};

template <int C>
__global__ __launch_bounds__(256, 2) void mlp_fwd_jvp_kernel(FwdJArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds_base[];
    fill_base(lds_base, a.params, L_W1, L_W2, L_B1, L_B2);
    fill_head(lds_base, a.params, C, L_WH1, L_WH2, L_WH3, L_BH1, L_BH2, L_BH3);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hi = lane >> 5, sl = lane & 31;
    const int64_t n_blk = (a.n + 31) >> 5;
    const ActKinds ak = act_kinds(a.act_code);
    for (int64_t blk = (int64_t)blockIdx.x * 4 + wave; blk < n_blk; blk += (int64_t)gridDim.x * 4) {
        int zo = 0;
        asm volatile("" : "+v"(zo));
        const float *lds = lds_base + zo;
        const float *W1 = lds + L_W1, *W2 = lds + L_W2, *WH1 = lds + L_WH1, *WH2 = lds + L_WH2;
        const int64_t i = blk * 32 + sl;
        const bool live = i < a.n;
        float x[16], xd[16];
        {
            const float *f = a.feat + blk * (16 * 64) + lane, *fd = a.featd + blk * (16 * 64) + lane;
#pragma unroll
            for (int s = 0; s < 16; ++s) { x[s] = f[s * 64]; xd[s] = fd[s * 64]; }
        }
        f32x16 h[2], hd[2];
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            h[0][g] = lds[L_B1 + rowc(g) + 4 * hi]; h[1][g] = lds[L_B1 + 32 + rowc(g) + 4 * hi];
            hd[0][g] = 0.f; hd[1][g] = 0.f;
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float a0 = W1[sl * 33 + 2 * s + hi], a1 = W1[(32 + sl) * 33 + 2 * s + hi];
            h[0] = MFMA(a0, x[s], h[0]); h[1] = MFMA(a1, x[s], h[1]);
            hd[0] = MFMA(a0, xd[s], hd[0]); hd[1] = MFMA(a1, xd[s], hd[1]);
        }
        act_jvp(h, hd, ak.bh);
        f32x16 o, od;
#pragma unroll
        for (int g = 0; g < 16; ++g) { o[g] = lds[L_B2 + rowc(g) + 4 * hi]; od[g] = 0.f; }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const float aw = W2[sl * 65 + 32 * r + rowc(g) + 4 * hi];
                o = MFMA(aw, h[r][g], o);
                od = MFMA(aw, hd[r][g], od);
            }
        bool sel = false;
        float dir[3] = {0.f, 0.f, 1.f}, dird[3] = {0.f, 0.f, 0.f};
        if (live) geom_jvp(a.src, a.sc, i, sel, dir, dird);
        if (live && hi == 0) {
            a.sigma[i] = sel ? act_density(o[0], ak.dn) : 0.f;
            a.sigmad[i] = sel ? dact_density(o[0], ak.dn) * od[0] : 0.f;
        }
        {
            float *bo = a.base_out + blk * (8 * 64) + lane, *bod = a.base_outd + blk * (8 * 64) + lane;
#pragma unroll
            for (int g = 0; g < 8; ++g) { bo[g * 64] = o[g]; bod[g * 64] = od[g]; }
        }
        float shs[8], shd[8];
        sh4_jvp_select(dir[0], dir[1], dir[2], dird[0], dird[1], dird[2], hi, shs, shd);
        f32x16 p[2], pd[2];
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            p[0][g] = lds[L_BH1 + rowc(g) + 4 * hi]; p[1][g] = lds[L_BH1 + 32 + rowc(g) + 4 * hi];
            pd[0][g] = 0.f; pd[1][g] = 0.f;
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int col = s < 8 ? rowc(s) + 4 * hi : 16 + 2 * (s - 8) + hi;
            const float bv = s < 8 ? o[s] : shs[s < 8 ? 0 : s - 8];
            const float bd = s < 8 ? od[s] : shd[s < 8 ? 0 : s - 8];
            const float a0 = WH1[sl * 33 + col], a1 = WH1[(32 + sl) * 33 + col];
            p[0] = MFMA(a0, bv, p[0]); p[1] = MFMA(a1, bv, p[1]);
            pd[0] = MFMA(a0, bd, pd[0]); pd[1] = MFMA(a1, bd, pd[1]);
        }
        act_jvp(p, pd, ak.hh);
        f32x16 q[2], qd[2];
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            q[0][g] = lds[L_BH2 + rowc(g) + 4 * hi]; q[1][g] = lds[L_BH2 + 32 + rowc(g) + 4 * hi];
            qd[0][g] = 0.f; qd[1][g] = 0.f;
        }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int col = 32 * r + rowc(g) + 4 * hi;
                const float a0 = WH2[sl * 65 + col], a1 = WH2[(32 + sl) * 65 + col];
                q[0] = MFMA(a0, p[r][g], q[0]); q[1] = MFMA(a1, p[r][g], q[1]);
                qd[0] = MFMA(a0, pd[r][g], qd[0]); qd[1] = MFMA(a1, pd[r][g], qd[1]);
            }
        act_jvp(q, qd, ak.hh);
        float acc[C], accd[C];
#pragma unroll
        for (int c = 0; c < C; ++c) { acc[c] = 0.f; accd[c] = 0.f; }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int g = 0; g < 16; ++g)
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const float w3 = lds[L_WH3 + c * 64 + 32 * r + rowc(g) + 4 * hi];
                    acc[c] += q[r][g] * w3;
                    accd[c] += qd[r][g] * w3;
                }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float z3 = acc[c] + __shfl_xor(acc[c], 32, 64) + lds[L_BH3 + c];
            const float z3d = accd[c] + __shfl_xor(accd[c], 32, 64);
            if (hi == 0 && live) {
                const float y = act_radiance(z3, ak.rd);
                a.rgb[i * C + c] = y;
                a.rgbd[i * C + c] = dact_radiance(y, ak.rd) * z3d;
            }
        }
    }
}

// ============================================================================ backward helpers
// stage a 64-neuron D-layout pair of accumulators as T[neuron][33] (per-wave LDS tile)
__device__ __forceinline__ void stage64(float *T, const f32x16 (&v)[2], int hi, int sl) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int g = 0; g < 16; ++g) T[(32 * r + rowc(g) + 4 * hi) * 33 + sl] = v[r][g];
}

// acc[ob][ib] += Tz(64 x 32 samples) . Ta(64 x 32 samples)^T ; also row sums of Tz into bsum[ob]
__device__ __forceinline__ void dw_64x64(f32x16 (&acc)[2][2], const float *Tz, const float *Ta, int hi, int sl,
                                         float *bsum) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const float z0 = Tz[sl * 33 + 2 * s + hi], z1 = Tz[(32 + sl) * 33 + 2 * s + hi];
        const float a0 = Ta[sl * 33 + 2 * s + hi], a1 = Ta[(32 + sl) * 33 + 2 * s + hi];
        if (bsum) { bsum[0] += z0; bsum[1] += z1; }
        acc[0][0] = MFMA(z0, a0, acc[0][0]); acc[0][1] = MFMA(z0, a1, acc[0][1]);
        acc[1][0] = MFMA(z1, a0, acc[1][0]); acc[1][1] = MFMA(z1, a1, acc[1][1]);
    }
}

// acc[ob] += Tz(64 x 32) . Ta(32 x 32)^T
__device__ __forceinline__ void dw_64x32(f32x16 (&acc)[2], const float *Tz, const float *Ta, int hi, int sl, float *bsum) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const float z0 = Tz[sl * 33 + 2 * s + hi], z1 = Tz[(32 + sl) * 33 + 2 * s + hi];
        const float av = Ta[sl * 33 + 2 * s + hi];
        if (bsum) { bsum[0] += z0; bsum[1] += z1; }
        acc[0] = MFMA(z0, av, acc[0]);
        acc[1] = MFMA(z1, av, acc[1]);
    }
}

constexpr int GRID_J = 256;                       // persistent workgroups of the jvp backward kernels
constexpr int GRID_J1 = 512;                      // head1 fits 256 registers: two workgroups per CU hide its waits
constexpr int LEN_H2 = 64 * 64 + 64;              // head.w1 | head.b1  (+ 65 C for head.wo | head.bo)
constexpr int LEN_H1 = 64 * 31 + 64;              // head.w0 | head.b0
__host__ __device__ constexpr int len_h2(int C) { return LEN_H2 + 65 * C; }

// ============================================================================ backward: output + head layer 1
struct BwdJ2Args {
    const float *params, *base_out, *base_outd;
    RaySrc src;
    ren_scene_dev sc;
    int64_t n;
    const float *rgb, *d_rgb, *d_rgbd;
    float *dz1, *dz1d, *slab;                      // dz1/dz1d: [blk][2][16][64] fragment order
    int act_code;                                  // activation alternatives (ren_mlp_common.h); 0 = shipped configs
};

template <int C>
__global__ __launch_bounds__(256, 1) void mlp_bwd_jvp_head2_kernel(BwdJ2Args a) {
    extern __shared__ __attribute__((aligned(16))) float lds_base[];
    fill_head(lds_base, a.params, C, LH_WH1, LH_WH2, LH_WH3, LH_BH1, LH_BH2, LH_BH3);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hi = lane >> 5, sl = lane & 31;
    float *T_z = lds_base + LH_END + wave * (2 * 64 * 33);
    float *T_a = T_z + 64 * 33;
    __syncthreads();
    const int64_t n_blk = (a.n + 31) >> 5;
    const ActKinds ak = act_kinds(a.act_code);
    f32x16 acc_w[2][2];
    float acc_w3[C][32], acc_b2[2] = {0.f, 0.f}, acc_b3[C];
#pragma unroll
    for (int g = 0; g < 16; ++g) { acc_w[0][0][g] = 0.f; acc_w[0][1][g] = 0.f; acc_w[1][0][g] = 0.f; acc_w[1][1][g] = 0.f; }
#pragma unroll
    for (int c = 0; c < C; ++c) {
        acc_b3[c] = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) acc_w3[c][k] = 0.f;
    }
    for (int64_t blk = (int64_t)blockIdx.x * 4 + wave; blk < n_blk; blk += (int64_t)gridDim.x * 4) {
        int zo = 0;
        asm volatile("" : "+v"(zo));
        const float *lds = lds_base + zo;
        const float *WH1 = lds + LH_WH1, *WH2 = lds + LH_WH2;
        const int64_t i = blk * 32 + sl;
        const bool live = i < a.n;
        bool sel = false;
        float dir[3] = {0.f, 0.f, 1.f}, dird[3] = {0.f, 0.f, 0.f};
        if (live) geom_jvp(a.src, a.sc, i, sel, dir, dird);
        float shs[8], shd[8], o[8], od[8];
        sh4_jvp_select(dir[0], dir[1], dir[2], dird[0], dird[1], dird[2], hi, shs, shd);
        {
            const float *bo = a.base_out + blk * (8 * 64) + lane, *bod = a.base_outd + blk * (8 * 64) + lane;
#pragma unroll
            for (int g = 0; g < 8; ++g) { o[g] = bo[g * 64]; od[g] = bod[g * 64]; }
        }
        // ---- recompute head layer 0: p = sp(z1), keep z1d (pre-activation tangent)
        f32x16 p[2], z1d[2];
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            p[0][g] = lds[LH_BH1 + rowc(g) + 4 * hi]; p[1][g] = lds[LH_BH1 + 32 + rowc(g) + 4 * hi];
            z1d[0][g] = 0.f; z1d[1][g] = 0.f;
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int col = s < 8 ? rowc(s) + 4 * hi : 16 + 2 * (s - 8) + hi;
            const float bv = s < 8 ? o[s] : shs[s < 8 ? 0 : s - 8];
            const float bd = s < 8 ? od[s] : shd[s < 8 ? 0 : s - 8];
            const float a0 = WH1[sl * 33 + col], a1 = WH1[(32 + sl) * 33 + col];
            p[0] = MFMA(a0, bv, p[0]); p[1] = MFMA(a1, bv, p[1]);
            z1d[0] = MFMA(a0, bd, z1d[0]); z1d[1] = MFMA(a1, bd, z1d[1]);
        }
        f32x16 pd[2];                                       // post-activation tangent s1 * z1d
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                p[r][g] = act_hidden(p[r][g], ak.hh);
                pd[r][g] = dact_hidden(p[r][g], ak.hh) * z1d[r][g];
            }
        // ---- recompute head layer 1: q = sp(z2), keep z2d
        f32x16 q[2], z2d[2];
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            q[0][g] = lds[LH_BH2 + rowc(g) + 4 * hi]; q[1][g] = lds[LH_BH2 + 32 + rowc(g) + 4 * hi];
            z2d[0][g] = 0.f; z2d[1][g] = 0.f;
        }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int col = 32 * r + rowc(g) + 4 * hi;
                const float a0 = WH2[sl * 65 + col], a1 = WH2[(32 + sl) * 65 + col];
                q[0] = MFMA(a0, p[r][g], q[0]); q[1] = MFMA(a1, p[r][g], q[1]);
                z2d[0] = MFMA(a0, pd[r][g], z2d[0]); z2d[1] = MFMA(a1, pd[r][g], z2d[1]);
            }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int g = 0; g < 16; ++g) q[r][g] = act_hidden(q[r][g], ak.hh);
        // ---- output layer: z3d = sum_n (s2 z2d)[n] w3[n]
        float z3d[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int g = 0; g < 16; ++g)
                    t += dact_hidden(q[r][g], ak.hh) * z2d[r][g] * lds[LH_WH3 + c * 64 + 32 * r + rowc(g) + 4 * hi];
            z3d[c] = t + __shfl_xor(t, 32, 64);
        }
        float dz3[C], dz3d[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float y = live ? a.rgb[i * C + c] : 0.f;
            const float s3 = dact_radiance(y, ak.rd);
            const float gy = live ? a.d_rgb[i * C + c] : 0.f, gyd = live ? a.d_rgbd[i * C + c] : 0.f;
            dz3[c] = gy * s3 + gyd * z3d[c] * d2act_radiance(y, s3, ak.rd);
            dz3d[c] = gyd * s3;
            if (hi == 0) acc_b3[c] += dz3[c];
        }
        // d q, d qd -> dz2, dz2d ; dW3 accumulation (needs q and qd = s2 z2d)
        f32x16 dz2[2], dz2d[2];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const float s2 = dact_hidden(q[r][g], ak.hh);
                const float qd = s2 * z2d[r][g];
                float dq = 0.f, dqd = 0.f;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const float w3 = lds[LH_WH3 + c * 64 + 32 * r + rowc(g) + 4 * hi];
                    dq += dz3[c] * w3; dqd += dz3d[c] * w3;
                    acc_w3[c][r * 16 + g] += dz3[c] * q[r][g] + dz3d[c] * qd;
                }
                dz2[r][g] = dq * s2 + dqd * z2d[r][g] * d2act_hidden(s2, ak.hh);
                dz2d[r][g] = dqd * s2;
            }
        // ---- dW(head.w1) += dz2 p^T + dz2d pd^T
        stage64(T_z, dz2, hi, sl); stage64(T_a, p, hi, sl);
        dw_64x64(acc_w, T_z, T_a, hi, sl, acc_b2);
        stage64(T_z, dz2d, hi, sl); stage64(T_a, pd, hi, sl);
        dw_64x64(acc_w, T_z, T_a, hi, sl, nullptr);
        // ---- d p = W^T dz2, d pd = W^T dz2d
        f32x16 dp[2], dpd[2];
#pragma unroll
        for (int g = 0; g < 16; ++g) { dp[0][g] = 0.f; dp[1][g] = 0.f; dpd[0][g] = 0.f; dpd[1][g] = 0.f; }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int orow = 32 * r + rowc(g) + 4 * hi;
                const float a0 = WH2[orow * 65 + sl], a1 = WH2[orow * 65 + 32 + sl];
                dp[0] = MFMA(a0, dz2[r][g], dp[0]); dp[1] = MFMA(a1, dz2[r][g], dp[1]);
                dpd[0] = MFMA(a0, dz2d[r][g], dpd[0]); dpd[1] = MFMA(a1, dz2d[r][g], dpd[1]);
            }
        // ---- through the activation of head layer 0 -> dz1, dz1d (to HBM, fragment order)
        {
            float *oz = a.dz1 + blk * (32 * 64) + lane, *ozd = a.dz1d + blk * (32 * 64) + lane;
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const float s1 = dact_hidden(p[r][g], ak.hh);
                    oz[(r * 16 + g) * 64] = dp[r][g] * s1 + dpd[r][g] * z1d[r][g] * d2act_hidden(s1, ak.hh);
                    ozd[(r * 16 + g) * 64] = dpd[r][g] * s1;
                }
        }
    }
    // ---- slab: [head.w1 | head.b1 | head.wo | head.bo]
    float *slab = a.slab + ((int64_t)blockIdx.x * 4 + wave) * len_h2(C);
#pragma unroll
    for (int ob = 0; ob < 2; ++ob) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int out = 32 * ob + rowc(g) + 4 * hi;
            slab[out * 64 + sl] = acc_w[ob][0][g];
            slab[out * 64 + 32 + sl] = acc_w[ob][1][g];
        }
        const float b2 = acc_b2[ob] + __shfl_xor(acc_b2[ob], 32, 64);
        if (hi == 0) slab[64 * 64 + 32 * ob + sl] = b2;
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            float v = acc_w3[c][k];
#pragma unroll
            for (int off = 1; off < 32; off <<= 1) v += __shfl_xor(v, off, 64);
            if (sl == 0) slab[LEN_H2 + c * 64 + 32 * (k >> 4) + rowc(k & 15) + 4 * hi] = v;
        }
        const float b3 = ren_wave_sum(acc_b3[c]);
        if (lane == 0) slab[LEN_H2 + 64 * C + c] = b3;
    }
}

// ============================================================================ backward: head layer 0 + density
struct BwdJ1Args {
    const float *params, *base_out, *base_outd, *dz1, *dz1d;
    RaySrc src;
    ren_scene_dev sc;
    int64_t n;
    const float *d_sigma, *d_sigmad;
    float *d_base, *d_based, *slab;
    int act_code;                                  // activation alternatives (ren_mlp_common.h); 0 = shipped configs
};

__global__ __launch_bounds__(256, 2) void mlp_bwd_jvp_head1_kernel(BwdJ1Args a) {
    extern __shared__ __attribute__((aligned(16))) float lds_base[];
    fill_head(lds_base, a.params, 1, LH_WH1, LH_WH2, LH_WH3, LH_BH1, LH_BH2, LH_BH3);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hi = lane >> 5, sl = lane & 31;
    float *T_z = lds_base + LH_END + wave * (96 * 33);
    float *T_a = T_z + 64 * 33;                              // [32][33]
    __syncthreads();
    const int64_t n_blk = (a.n + 31) >> 5;
    const ActKinds ak = act_kinds(a.act_code);
    f32x16 acc_w[2];
    float acc_b[2] = {0.f, 0.f};
#pragma unroll
    for (int g = 0; g < 16; ++g) { acc_w[0][g] = 0.f; acc_w[1][g] = 0.f; }
    for (int64_t blk = (int64_t)blockIdx.x * 4 + wave; blk < n_blk; blk += (int64_t)gridDim.x * 4) {
        int zo = 0;
        asm volatile("" : "+v"(zo));
        const float *WH1 = lds_base + zo + LH_WH1;
        const int64_t i = blk * 32 + sl;
        const bool live = i < a.n;
        bool sel = false;
        float dir[3] = {0.f, 0.f, 1.f}, dird[3] = {0.f, 0.f, 0.f};
        if (live) geom_jvp(a.src, a.sc, i, sel, dir, dird);
        float shs[8], shd[8], o[8], od[8];
        sh4_jvp_select(dir[0], dir[1], dir[2], dird[0], dird[1], dird[2], hi, shs, shd);
        f32x16 dz[2], dzd[2];
        {
            const float *bo = a.base_out + blk * (8 * 64) + lane, *bod = a.base_outd + blk * (8 * 64) + lane;
#pragma unroll
            for (int g = 0; g < 8; ++g) { o[g] = bo[g * 64]; od[g] = bod[g * 64]; }
            const float *iz = a.dz1 + blk * (32 * 64) + lane, *izd = a.dz1d + blk * (32 * 64) + lane;
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int g = 0; g < 16; ++g) { dz[r][g] = iz[(r * 16 + g) * 64]; dzd[r][g] = izd[(r * 16 + g) * 64]; }
        }
        // ---- dW(head.w0) += dz1 v^T + dz1d vd^T,  v = [base_out(16) | SH(16)]
        stage64(T_z, dz, hi, sl);
#pragma unroll
        for (int g = 0; g < 8; ++g) { T_a[(rowc(g) + 4 * hi) * 33 + sl] = o[g]; T_a[(16 + 2 * g + hi) * 33 + sl] = shs[g]; }
        dw_64x32(acc_w, T_z, T_a, hi, sl, acc_b);
        stage64(T_z, dzd, hi, sl);
#pragma unroll
        for (int g = 0; g < 8; ++g) { T_a[(rowc(g) + 4 * hi) * 33 + sl] = od[g]; T_a[(16 + 2 * g + hi) * 33 + sl] = shd[g]; }
        dw_64x32(acc_w, T_z, T_a, hi, sl, nullptr);
        // ---- d v = WH1^T dz1 (rows 0..15 = d base_out), d vd = WH1^T dz1d
        f32x16 dv, dvd;
#pragma unroll
        for (int g = 0; g < 16; ++g) { dv[g] = 0.f; dvd[g] = 0.f; }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const float aw = WH1[(32 * r + rowc(g) + 4 * hi) * 33 + sl];
                dv = MFMA(aw, dz[r][g], dv);
                dvd = MFMA(aw, dzd[r][g], dvd);
            }
        if (hi == 0) {
            // sigma = e sel, sigmad = e sel od0 with e = exp(min(o0 - 1, 15)):
            // d o0 = d sigma e + d sigmad sigmad (unclamped branch), d od0 = d sigmad e
            const float ds = live ? a.d_sigma[i] : 0.f, dsd = live ? a.d_sigmad[i] : 0.f;
            const float e = sel ? dact_density(o[0], ak.dn) : 0.f;                 // (default: exp(min(o0 - 1, 15)))
            dv[0] = ds * e + dsd * d2act_density(o[0], e, ak.dn) * od[0];
            dvd[0] = dsd * e;
        }
        {
            float *db = a.d_base + blk * (8 * 64) + lane, *dbd = a.d_based + blk * (8 * 64) + lane;
#pragma unroll
            for (int g = 0; g < 8; ++g) { db[g * 64] = dv[g]; dbd[g * 64] = dvd[g]; }
        }
    }
    float *slab = a.slab + ((int64_t)blockIdx.x * 4 + wave) * LEN_H1;          // [head.w0 | head.b0]
#pragma unroll
    for (int ob = 0; ob < 2; ++ob) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int out = 32 * ob + rowc(g) + 4 * hi;
            if (sl != 0) slab[out * 31 + (sl < 16 ? 15 + sl : sl - 16)] = acc_w[ob][g];
        }
        const float b1 = acc_b[ob] + __shfl_xor(acc_b[ob], 32, 64);
        if (hi == 0) slab[64 * 31 + 32 * ob + sl] = b1;
    }
}

// ============================================================================ backward: base MLP
struct BwdJBArgs {
    const float *params, *feat, *featd, *d_base, *d_based;
    int64_t n;
    float *dfeat, *dfeatd, *slab;
    int act_code;                                  // activation alternatives (ren_mlp_common.h); 0 = shipped configs
};

__global__ __launch_bounds__(256, 1) void mlp_bwd_jvp_base_kernel(BwdJBArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds_base[];
    fill_base(lds_base, a.params, LB_W1, LB_W2, LB_B1, LB_B2);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hi = lane >> 5, sl = lane & 31;
    float *T_a = lds_base + LB_END + wave * (96 * 33);       // [64][33]
    float *T_b = T_a + 64 * 33;                              // [32][33]
    for (int k = lane; k < 32 * 33; k += 64) T_b[k] = 0.f;
    __syncthreads();
    const int64_t n_blk = (a.n + 31) >> 5;
    const ActKinds ak = act_kinds(a.act_code);
    f32x16 acc_w2[2], acc_w1[2];
    float acc_b2 = 0.f, acc_b1[2] = {0.f, 0.f};
#pragma unroll
    for (int g = 0; g < 16; ++g) { acc_w2[0][g] = 0.f; acc_w2[1][g] = 0.f; acc_w1[0][g] = 0.f; acc_w1[1][g] = 0.f; }
    for (int64_t blk = (int64_t)blockIdx.x * 4 + wave; blk < n_blk; blk += (int64_t)gridDim.x * 4) {
        int zo = 0;
        asm volatile("" : "+v"(zo));
        const float *lds = lds_base + zo;
        const float *W1 = lds + LB_W1, *W2 = lds + LB_W2;
        float x[16], xd[16], dob[8], dobd[8];
        {
            const float *f = a.feat + blk * (16 * 64) + lane, *fd = a.featd + blk * (16 * 64) + lane;
#pragma unroll
            for (int s = 0; s < 16; ++s) { x[s] = f[s * 64]; xd[s] = fd[s * 64]; }
            const float *db = a.d_base + blk * (8 * 64) + lane, *dbd = a.d_based + blk * (8 * 64) + lane;
#pragma unroll
            for (int g = 0; g < 8; ++g) { dob[g] = db[g * 64]; dobd[g] = dbd[g * 64]; }
        }
        // ---- recompute hidden layer: h = sp(z0), keep z0d; hd = s z0d
        f32x16 h[2], z0d[2], hd[2];
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            h[0][g] = lds[LB_B1 + rowc(g) + 4 * hi]; h[1][g] = lds[LB_B1 + 32 + rowc(g) + 4 * hi];
            z0d[0][g] = 0.f; z0d[1][g] = 0.f;
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float a0 = W1[sl * 33 + 2 * s + hi], a1 = W1[(32 + sl) * 33 + 2 * s + hi];
            h[0] = MFMA(a0, x[s], h[0]); h[1] = MFMA(a1, x[s], h[1]);
            z0d[0] = MFMA(a0, xd[s], z0d[0]); z0d[1] = MFMA(a1, xd[s], z0d[1]);
        }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                h[r][g] = act_hidden(h[r][g], ak.bh);
                hd[r][g] = dact_hidden(h[r][g], ak.bh) * z0d[r][g];
            }
        // ---- dW(base.wo) += dO h^T + dOd hd^T   (rows 16..31 of T_b stay zero)
        stage64(T_a, h, hi, sl);
#pragma unroll
        for (int g = 0; g < 8; ++g) T_b[(rowc(g) + 4 * hi) * 33 + sl] = dob[g];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float az = T_b[sl * 33 + 2 * s + hi];
            acc_b2 += az;
            acc_w2[0] = MFMA(az, T_a[sl * 33 + 2 * s + hi], acc_w2[0]);
            acc_w2[1] = MFMA(az, T_a[(32 + sl) * 33 + 2 * s + hi], acc_w2[1]);
        }
        stage64(T_a, hd, hi, sl);
#pragma unroll
        for (int g = 0; g < 8; ++g) T_b[(rowc(g) + 4 * hi) * 33 + sl] = dobd[g];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float az = T_b[sl * 33 + 2 * s + hi];
            acc_w2[0] = MFMA(az, T_a[sl * 33 + 2 * s + hi], acc_w2[0]);
            acc_w2[1] = MFMA(az, T_a[(32 + sl) * 33 + 2 * s + hi], acc_w2[1]);
        }
        // ---- d h = Wo^T dO, d hd = Wo^T dOd ; through the activation -> dz0, dz0d
        f32x16 dz[2], dzd[2];
#pragma unroll
        for (int g = 0; g < 16; ++g) { dz[0][g] = 0.f; dz[1][g] = 0.f; dzd[0][g] = 0.f; dzd[1][g] = 0.f; }
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const int orow = rowc(g) + 4 * hi;
            const float a0 = W2[orow * 65 + sl], a1 = W2[orow * 65 + 32 + sl];
            dz[0] = MFMA(a0, dob[g], dz[0]); dz[1] = MFMA(a1, dob[g], dz[1]);
            dzd[0] = MFMA(a0, dobd[g], dzd[0]); dzd[1] = MFMA(a1, dobd[g], dzd[1]);
        }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const float s0 = dact_hidden(h[r][g], ak.bh);
                const float dh = dz[r][g], dhd = dzd[r][g];
                dz[r][g] = dh * s0 + dhd * z0d[r][g] * d2act_hidden(s0, ak.bh);
                dzd[r][g] = dhd * s0;
            }
        // ---- dW(base.w0) += dz0 x^T + dz0d xd^T
        stage64(T_a, dz, hi, sl);
#pragma unroll
        for (int s = 0; s < 16; ++s) T_b[(2 * s + hi) * 33 + sl] = x[s];
        dw_64x32(acc_w1, T_a, T_b, hi, sl, acc_b1);
        stage64(T_a, dzd, hi, sl);
#pragma unroll
        for (int s = 0; s < 16; ++s) T_b[(2 * s + hi) * 33 + sl] = xd[s];
        dw_64x32(acc_w1, T_a, T_b, hi, sl, nullptr);
#pragma unroll
        for (int s = 8; s < 16; ++s) T_b[(2 * s + hi) * 33 + sl] = 0.f;       // restore the zero padding rows
        // ---- d x = W0^T dz0, d xd = W0^T dz0d  -> fragment layout
        f32x16 dxv, dxdv;
#pragma unroll
        for (int g = 0; g < 16; ++g) { dxv[g] = 0.f; dxdv[g] = 0.f; }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const float aw = W1[(32 * r + rowc(g) + 4 * hi) * 33 + sl];
                dxv = MFMA(aw, dz[r][g], dxv);
                dxdv = MFMA(aw, dzd[r][g], dxdv);
            }
        {
            float *df = a.dfeat + blk * (16 * 64) + sl, *dfd = a.dfeatd + blk * (16 * 64) + sl;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int f = rowc(g) + 4 * hi;
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
    const float b2 = acc_b2 + __shfl_xor(acc_b2, 32, 64);
    if (hi == 0 && sl < 16) slab[P_BBO + sl] = b2;
#pragma unroll
    for (int ob = 0; ob < 2; ++ob) {
        const float b1 = acc_b1[ob] + __shfl_xor(acc_b1[ob], 32, 64);
        if (hi == 0) slab[P_BB0 + 32 * ob + sl] = b1;
    }
}

constexpr size_t FWDJ_LDS = (size_t)L_WEIGHTS_END * 4;
constexpr size_t J2_LDS = (size_t)(LH_END + 4 * 2 * 64 * 33) * 4;
constexpr size_t J1_LDS = (size_t)(LH_END + 4 * 96 * 33) * 4;
constexpr size_t JB_LDS = (size_t)(LB_END + 4 * 96 * 33) * 4;

}  // namespace

extern "C" int ren_mlp_fwd_jvp(const float *mlp_params, int32_t C, int32_t activations, const float *feat, const float *featd,
                               const ren_scene_desc *scene, const float *rays_o, const float *rays_d,
                               const float *rays_dd, const int32_t *ray_indices, const float *t_starts,
                               const float *t_ends, int64_t n, float *rgb, float *rgbd, float *sigma,
                               float *sigmad, float *base_out, float *base_outd, void *stream) {
    if (!mlp_params || !feat || !featd || !scene || !rays_o || !rays_d || !rays_dd || !ray_indices || !t_starts ||
        !t_ends || !rgb || !rgbd || !sigma || !sigmad || !base_out || !base_outd || n < 0)
        return REN_ERR_BAD_ARG;
    if (C != 1 && C != 3) return REN_ERR_UNSUPPORTED;
    if (n == 0) return REN_OK;
    FwdJArgs a;
    a.params = mlp_params; a.feat = feat; a.featd = featd;
    a.src = RaySrc{rays_o, rays_d, rays_dd, ray_indices, t_starts, t_ends};
    a.sc = ren_make_scene(scene);
    a.n = n; a.rgb = rgb; a.rgbd = rgbd; a.sigma = sigma; a.sigmad = sigmad; a.base_out = base_out; a.base_outd = base_outd;
    a.act_code = activations;
    int64_t blocks = ((n + 31) / 32 + 3) / 4;
    if (blocks > 768) blocks = 768;
    hipStream_t st = (hipStream_t)stream;
    if (C == 1) hipLaunchKernelGGL((mlp_fwd_jvp_kernel<1>), dim3((int)blocks), dim3(256), FWDJ_LDS, st, a);
    else        hipLaunchKernelGGL((mlp_fwd_jvp_kernel<3>), dim3((int)blocks), dim3(256), FWDJ_LDS, st, a);
    REN_CHECK_LAUNCH();
}

extern "C" int64_t ren_mlp_bwd_jvp_workspace_floats(int32_t C) {
    if (C != 1 && C != 3) return -1;
    return (int64_t)GRID_J * 4 * (len_h2(C) + P_BASE_N) + (int64_t)GRID_J1 * 4 * LEN_H1;
}

extern "C" int ren_mlp_bwd_jvp(const float *mlp_params, int32_t C, int32_t activations, const float *feat, const float *featd,
                               const float *base_out, const float *base_outd, const ren_scene_desc *scene,
                               const float *rays_o, const float *rays_d, const float *rays_dd,
                               const int32_t *ray_indices, const float *t_starts, const float *t_ends, int64_t n,
                               const float *rgb, const float *d_rgb, const float *d_rgbd, const float *d_sigma,
                               const float *d_sigmad, float *scratch, float *dfeat, float *dfeatd,
                               float *grad_mlp_params, float *workspace, void *stream) {
    if (!mlp_params || !feat || !featd || !base_out || !base_outd || !scene || !rays_o || !rays_d || !rays_dd ||
        !ray_indices || !t_starts || !t_ends || !rgb || !d_rgb || !d_rgbd || !d_sigma || !d_sigmad || !scratch ||
        !dfeat || !dfeatd || !grad_mlp_params || !workspace || n < 0)
        return REN_ERR_BAD_ARG;
    if (C != 1 && C != 3) return REN_ERR_UNSUPPORTED;
    if (n == 0) return REN_OK;
    hipStream_t st = (hipStream_t)stream;
    (void)hipFuncSetAttribute((const void *)mlp_bwd_jvp_head2_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)J2_LDS);
    (void)hipFuncSetAttribute((const void *)mlp_bwd_jvp_head2_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)J2_LDS);
    (void)hipFuncSetAttribute((const void *)mlp_bwd_jvp_head1_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)J1_LDS);
    (void)hipFuncSetAttribute((const void *)mlp_bwd_jvp_base_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)JB_LDS);
    const int64_t n_blk = (n + 31) / 32;
    // scratch (floats): dz1 | dz1d (2048 per block each) | d_base | d_based (512 per block each)
    float *dz1 = scratch, *dz1d = dz1 + n_blk * 2048, *d_base = dz1d + n_blk * 2048, *d_based = d_base + n_blk * 512;
    float *slab2 = workspace, *slab1 = slab2 + (int64_t)GRID_J * 4 * len_h2(C), *slabb = slab1 + (int64_t)GRID_J1 * 4 * LEN_H1;
    const RaySrc src{rays_o, rays_d, rays_dd, ray_indices, t_starts, t_ends};
    const ren_scene_dev sc = ren_make_scene(scene);
    BwdJ2Args a2;
    a2.params = mlp_params; a2.base_out = base_out; a2.base_outd = base_outd; a2.src = src; a2.sc = sc; a2.n = n;
    a2.rgb = rgb; a2.d_rgb = d_rgb; a2.d_rgbd = d_rgbd; a2.dz1 = dz1; a2.dz1d = dz1d; a2.slab = slab2;
    a2.act_code = activations;
    if (C == 1) hipLaunchKernelGGL((mlp_bwd_jvp_head2_kernel<1>), dim3(GRID_J), dim3(256), J2_LDS, st, a2);
    else        hipLaunchKernelGGL((mlp_bwd_jvp_head2_kernel<3>), dim3(GRID_J), dim3(256), J2_LDS, st, a2);
    BwdJ1Args a1;
    a1.params = mlp_params; a1.base_out = base_out; a1.base_outd = base_outd; a1.dz1 = dz1; a1.dz1d = dz1d;
    a1.src = src; a1.sc = sc; a1.n = n; a1.d_sigma = d_sigma; a1.d_sigmad = d_sigmad; a1.d_base = d_base;
    a1.d_based = d_based; a1.slab = slab1;
    a1.act_code = activations;
    hipLaunchKernelGGL(mlp_bwd_jvp_head1_kernel, dim3(GRID_J1), dim3(256), J1_LDS, st, a1);
    BwdJBArgs ab;
    ab.params = mlp_params; ab.feat = feat; ab.featd = featd; ab.d_base = d_base; ab.d_based = d_based; ab.n = n;
    ab.dfeat = dfeat; ab.dfeatd = dfeatd; ab.slab = slabb;
    ab.act_code = activations;
    hipLaunchKernelGGL(mlp_bwd_jvp_base_kernel, dim3(GRID_J), dim3(256), JB_LDS, st, ab);
    launch_reduce_slabs(slab2, GRID_J * 4, len_h2(C), grad_mlp_params + P_HW1, st);
    launch_reduce_slabs(slab1, GRID_J1 * 4, LEN_H1, grad_mlp_params + P_HW0, st);
    launch_reduce_slabs(slabb, GRID_J * 4, P_BASE_N, grad_mlp_params, st);
    REN_CHECK_LAUNCH();
}
