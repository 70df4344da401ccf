// Fused Instant-NGP MLPs (base 32->64->16, head [SH16|geo15]->64->64->C) on the gfx950 matrix
// cores in exact fp32 (v_mfma_f32_32x32x2_f32), forward and backward.
// Replaces NGPradianceField.query_density/_query_rgb/forward
// (robust_e_nerf/external/ngp.py:230-280) = 5 torch sgemms + SH + ~20 elementwise launches and
// ~1.2 KB/sample of HBM intermediates with one forward launch and two backward launches whose
// only HBM traffic is the 128 B/sample hash features, 64 B/sample of saved base outputs and the
// per-sample (rgb, sigma) results.
//
// Data layout ("lane = sample").  A wavefront works on blocks of 32 samples.  For
// D = A.B with the 32x32x2 f32 MFMA, lane l supplies A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31]
// and receives D[row(g, l>>5)][col = l&31], row(g,hi) = (g&3) + 8*(g>>2) + 4*hi, g = 0..15.
// With A = weights (i = output neuron) and B = activations (j = sample) the accumulator
// registers of one layer ARE the B operands of the next layer: register g of lane (sample, hi)
// holds neuron row(g,hi), so k-step g pairs the neurons {row(g,0), row(g,1)} and the weight
// fragment is read with the same pairing.  No cross-lane movement, no LDS round trip for
// activations in the forward chain.  Weights live in LDS once per workgroup, row-major with the
// row stride padded to K+1 words so both the "lanes = output rows" (forward) and "lanes = input
// columns" (backward, W^T) fragment reads are bank-conflict free.
//
// Weight gradients need the transposed layout (lane = neuron, k = sample): activations and
// their gradients are staged once per layer through a per-wave LDS tile [neuron][33] and the
// 32x32 dW tiles accumulate in registers over the whole kernel; every wave writes its partial
// dW to a slab in HBM and a small kernel reduces the slabs (deterministic, no atomics).
#include "ren_mlp_common.h"

namespace {

// post-activation values of the three hidden layers (h, p, q) saved by the forward for ren_mlp_bwd_saved:
// 3 x 2 x 16 registers x 64 lanes per 32-sample block = 768 B/sample.  Recomputing them in the backward
// costs 128 f32 MFMAs + 192 softplus per block (~40 % of the backward); on a chip whose f32 MFMA and VALU
// share a pipe while HBM idles, the 13 GB round trip per render is the cheaper side.
constexpr int ACT_SAVE_FLOATS = 3 * 2 * 16 * 64;

// ============================================================================ forward
struct FwdArgs {
    const float *params, *feat;
    SampleSrc src;
    ren_scene_dev sc;
    int64_t n;
    float *rgb, *sigma, *base_out;
    float *acts;                                    // optional [blk][3 = h,p,q][2][16][64]: post-activation values for ren_mlp_bwd_saved
    int act_code;                                   // activation alternatives (ren_mlp_common.h act_kinds); 0 = shipped configs
};

template <int C, bool DENSITY_ONLY, bool RB>
__global__ __launch_bounds__(256, 2) void mlp_fwd_kernel(FwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds_base[];
    fill_base(lds_base, a.params, L_W1, L_W2, L_B1, L_B2);
    if (!DENSITY_ONLY) fill_head(lds_base, a.params, C, L_WH1, L_WH2, L_WH3, L_BH1, L_BH2, L_BH3);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hi = lane >> 5, sl = lane & 31;
    const int64_t n_blk = (a.n + 31) >> 5;
    const ActKinds ak = act_kinds(a.act_code);

    for (int64_t blk = (int64_t)blockIdx.x * 4 + wave; blk < n_blk; blk += (int64_t)gridDim.x * 4) {
        // The weight image is loop invariant; without this opaque offset LICM hoists every LDS
        // read (~400 values) out of the persistent loop and spills them.
        int zo = 0;
        asm volatile("" : "+v"(zo));
        const float *lds = lds_base + zo;
        const float *W1 = lds + L_W1, *W2 = lds + L_W2, *WH1 = lds + L_WH1, *WH2 = lds + L_WH2;
        const int64_t i = blk * 32 + sl;
        const bool live = i < a.n;
        // ---- hash features as B operands: k-step s = level s, lane hi = feature parity
        float x[16];
        {
            const float *f = a.feat + blk * (16 * 64) + lane;
#pragma unroll
            for (int s = 0; s < 16; ++s) x[s] = lin_in<RB>(f[s * 64]);
        }
        // ---- base layer 0: 32 -> 64, softplus(beta=100)
        f32x16 h[2];
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            h[0][g] = lds[L_B1 + rowc(g) + 4 * hi];
            h[1][g] = lds[L_B1 + 32 + rowc(g) + 4 * hi];
        }
        // tile-major: tile 1's MFMAs cover tile 0's activation on the VALU, and the next layer's
        // k-steps over tile 0 cover tile 1's (a 32x32x2 f32 MFMA leaves 60 issue cycles free)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
#pragma unroll
            for (int s = 0; s < 16; ++s) h[r] = MFMA(W1[(32 * r + sl) * 33 + 2 * s + hi], x[s], h[r]);
#pragma unroll
            for (int g = 0; g < 16; ++g) h[r][g] = act_hidden(h[r][g], ak.bh);
            if (a.acts) {                                       // ONE branch, constant offsets
                float *ap = a.acts + blk * ACT_SAVE_FLOATS + r * 1024 + lane;
#pragma unroll
                for (int g = 0; g < 16; ++g) __builtin_nontemporal_store(h[r][g], ap + g * 64);
            }
#pragma unroll
            for (int g = 0; g < 16; ++g) h[r][g] = lin_in<RB>(h[r][g]);
        }
        // ---- base output: 64 -> 16 (rows 16..31 of the tile are zero padding)
        f32x16 o;
#pragma unroll
        for (int g = 0; g < 16; ++g) o[g] = lds[L_B2 + rowc(g) + 4 * hi];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int g = 0; g < 16; ++g)
                o = MFMA(W2[sl * 65 + 32 * r + rowc(g) + 4 * hi], h[r][g], o);
        // ---- density + per-sample geometry
        bool sel = false;
        float dx = 0.f, dy = 0.f, dz = 1.f;
        if (live) sample_geom(a.src, a.sc, i, sel, dx, dy, dz);
        if (live && hi == 0) a.sigma[i] = sel ? act_density(o[0], ak.dn) : 0.f;     // ngp.py:247-250
        if (a.base_out) {
            float *bo = a.base_out + blk * (8 * 64) + lane;
#pragma unroll
            for (int g = 0; g < 8; ++g) bo[g * 64] = o[g];
        }
        if (DENSITY_ONLY) continue;
        float shs[8];
        sh4_select(dx, dy, dz, hi, shs);
        // ---- head layer 0: v = [base_out(16) | SH(16)] -> 64
        f32x16 p[2];
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            p[0][g] = lds[L_BH1 + rowc(g) + 4 * hi];
            p[1][g] = lds[L_BH1 + 32 + rowc(g) + 4 * hi];
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int col = s < 8 ? rowc(s) + 4 * hi : 16 + 2 * (s - 8) + hi;
                const float bv = lin_in<RB>(s < 8 ? o[s] : shs[s < 8 ? 0 : s - 8]);
                p[r] = MFMA(WH1[(32 * r + sl) * 33 + col], bv, p[r]);
            }
#pragma unroll
            for (int g = 0; g < 16; ++g) p[r][g] = act_hidden(p[r][g], ak.hh);
            if (a.acts) {
                float *ap = a.acts + blk * ACT_SAVE_FLOATS + (2 + r) * 1024 + lane;
#pragma unroll
                for (int g = 0; g < 16; ++g) __builtin_nontemporal_store(p[r][g], ap + g * 64);
            }
#pragma unroll
            for (int g = 0; g < 16; ++g) p[r][g] = lin_in<RB>(p[r][g]);
        }
        // ---- head layer 1: 64 -> 64
        f32x16 q[2];
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            q[0][g] = lds[L_BH2 + rowc(g) + 4 * hi];
            q[1][g] = lds[L_BH2 + 32 + rowc(g) + 4 * hi];
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int g = 0; g < 16; ++g)
                    q[t] = MFMA(WH2[(32 * t + sl) * 65 + 32 * r + rowc(g) + 4 * hi], p[r][g], q[t]);
        // ---- head output: 64 -> C on the VALU (each lane holds 32 of its sample's 64 neurons)
        float acc[C];
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = 0.f;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
#pragma unroll
            for (int g = 0; g < 16; ++g) q[r][g] = act_hidden(q[r][g], ak.hh);
            if (a.acts) {
                float *ap = a.acts + blk * ACT_SAVE_FLOATS + (4 + r) * 1024 + lane;
#pragma unroll
                for (int g = 0; g < 16; ++g) __builtin_nontemporal_store(q[r][g], ap + g * 64);
            }
        }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const float qa = lin_in<RB>(q[r][g]);
#pragma unroll
                for (int c = 0; c < C; ++c) acc[c] += qa * lds[L_WH3 + c * 64 + 32 * r + rowc(g) + 4 * hi];
            }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float t = acc[c] + __shfl_xor(acc[c], 32, 64);
            if (hi == 0 && live) a.rgb[i * C + c] = act_radiance(t + lds[L_BH3 + c], ak.rd);
        }
    }
}

// ============================================================================ backward, head
// f32 MFMA and VALU share one issue pipe on gfx950 (tools/mfma_valu_bench.hip: their times add, also
// across two waves of one SIMD), so a second wave per SIMD buys nothing here and would need spills.
#ifndef REN_HEAD_OCC
#define REN_HEAD_OCC 1
#endif
constexpr int GRID_H = 256 * REN_HEAD_OCC, GRID_B = 512;   // persistent workgroups (4 waves each)

struct BwdHArgs {
    const float *params, *base_out;
    SampleSrc src;
    ren_scene_dev sc;
    int64_t n;
    const float *rgb, *d_rgb, *d_sigma;
    float *d_base, *slab;                           // d_base: fragment layout [blk][8][64]
    const float *acts;                              // SAVED: forward's post-activation values (see ACT_SAVE_FLOATS)
    int act_code;
};

template <int C, bool RB, bool SAVED>
__global__ __launch_bounds__(256, REN_HEAD_OCC) void mlp_bwd_head_kernel(BwdHArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds_base[];
    fill_head(lds_base, a.params, C, LH_WH1, LH_WH2, LH_WH3, LH_BH1, LH_BH2, LH_BH3);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hi = lane >> 5, sl = lane & 31;
    float *T_dz = lds_base + LH_END + wave * (96 * 33);          // [64][33]
    float *T_act = T_dz + 64 * 33;                                 // [32][33]
    __syncthreads();
    const int64_t n_blk = (a.n + 31) >> 5;
    const ActKinds ak = act_kinds(a.act_code);

    f32x16 acc_wh2[2][2], acc_wh1[2];
    float acc_w3[C][32], acc_bh2[2] = {0.f, 0.f}, acc_bh1[2] = {0.f, 0.f}, acc_bh3[C];
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        acc_wh2[0][0][g] = 0.f; acc_wh2[0][1][g] = 0.f; acc_wh2[1][0][g] = 0.f; acc_wh2[1][1][g] = 0.f;
        acc_wh1[0][g] = 0.f; acc_wh1[1][g] = 0.f;
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
        acc_bh3[c] = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) acc_w3[c][k] = 0.f;
    }

    // SAVED: one wave per SIMD means nothing hides a load, so the 64 saved activations of the NEXT block
    // are requested before this block's arithmetic starts
    f32x16 pn[2], qn[2];
    auto fetch_acts = [&](int64_t b) {
        const float *ac = a.acts + b * ACT_SAVE_FLOATS + lane;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int g = 0; g < 16; ++g) { pn[r][g] = ac[(32 + r * 16 + g) * 64]; qn[r][g] = ac[(64 + r * 16 + g) * 64]; }
    };
    if (SAVED && (int64_t)blockIdx.x * 4 + wave < n_blk) fetch_acts((int64_t)blockIdx.x * 4 + wave);

    for (int64_t blk = (int64_t)blockIdx.x * 4 + wave; blk < n_blk; blk += (int64_t)gridDim.x * 4) {
        int zo = 0;                                     // defeat LICM of LDS weight reads (see forward)
        asm volatile("" : "+v"(zo));
        const float *lds = lds_base + zo;
        const float *WH1 = lds + LH_WH1, *WH2 = lds + LH_WH2;
        const int64_t i = blk * 32 + sl;
        const bool live = i < a.n;
        bool sel = false;
        float dx = 0.f, dy = 0.f, dz = 1.f;
        if (live) sample_geom(a.src, a.sc, i, sel, dx, dy, dz);
        float shs[8], o[8];
        sh4_select(dx, dy, dz, hi, shs);
        {
            const float *bo = a.base_out + blk * (8 * 64) + lane;
#pragma unroll
            for (int g = 0; g < 8; ++g) o[g] = bo[g * 64];
        }
        // ---- head activations: saved by the forward (SAVED) or recomputed
        f32x16 p[2], q[2];
        // bf16 mode: the activation VALUE that feeds the next linear layer (and its weight gradient) is
        // rounded, the softplus derivative is taken at the unrounded output: s1/s2 are formed here
        f32x16 s1[2], s2[2];
        if (SAVED) {
#pragma unroll
            for (int r = 0; r < 2; ++r) { p[r] = pn[r]; q[r] = qn[r]; }
            const int64_t nxt = blk + (int64_t)gridDim.x * 4;
            if (nxt < n_blk) fetch_acts(nxt);
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    if (RB) { s1[r][g] = dact_hidden(p[r][g], ak.hh); s2[r][g] = dact_hidden(q[r][g], ak.hh); }
                    p[r][g] = lin_in<RB>(p[r][g]);
                    q[r][g] = lin_in<RB>(q[r][g]);
                }
        } else {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            p[0][g] = lds[LH_BH1 + rowc(g) + 4 * hi];
            p[1][g] = lds[LH_BH1 + 32 + rowc(g) + 4 * hi];
            q[0][g] = lds[LH_BH2 + rowc(g) + 4 * hi];
            q[1][g] = lds[LH_BH2 + 32 + rowc(g) + 4 * hi];
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int col = s < 8 ? rowc(s) + 4 * hi : 16 + 2 * (s - 8) + hi;
            const float b = lin_in<RB>(s < 8 ? o[s] : shs[s - 8]);
            p[0] = MFMA(WH1[sl * 33 + col], b, p[0]);
            p[1] = MFMA(WH1[(32 + sl) * 33 + col], b, p[1]);
        }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const float y = act_hidden(p[r][g], ak.hh);
                if (RB) s1[r][g] = dact_hidden(y, ak.hh);
                p[r][g] = lin_in<RB>(y);
            }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int col = 32 * r + rowc(g) + 4 * hi;
                q[0] = MFMA(WH2[sl * 65 + col], p[r][g], q[0]);
                q[1] = MFMA(WH2[(32 + sl) * 65 + col], p[r][g], q[1]);
            }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const float y = act_hidden(q[r][g], ak.hh);
                if (RB) s2[r][g] = dact_hidden(y, ak.hh);
                q[r][g] = lin_in<RB>(y);
            }
        }
        // ---- output layer backward: d z3 = d rgb * softplus1'(z3) = d rgb * (1 - exp(-rgb))
        float dz3[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            dz3[c] = live ? a.d_rgb[i * C + c] * dact_radiance(a.rgb[i * C + c], ak.rd) : 0.f;
            if (hi == 0) acc_bh3[c] += dz3[c];
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int g = 0; g < 16; ++g) acc_w3[c][r * 16 + g] += dz3[c] * q[r][g];
        }
        // d q -> d z2 (in place in q)
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                float dq = 0.f;
#pragma unroll
                for (int c = 0; c < C; ++c) dq += dz3[c] * lds[LH_WH3 + c * 64 + 32 * r + rowc(g) + 4 * hi];
                q[r][g] = dq * (RB ? s2[r][g] : dact_hidden(q[r][g], ak.hh));
            }
        // ---- dW(head.w1) += dZ2 . P^T  (stage both as [neuron][sample]; P in two 32-row halves so
        //      the per-wave staging area stays at 96 rows and two workgroups fit one CU)
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int g = 0; g < 16; ++g) T_dz[(32 * r + rowc(g) + 4 * hi) * 33 + sl] = q[r][g];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
#pragma unroll
            for (int g = 0; g < 16; ++g) T_act[(rowc(g) + 4 * hi) * 33 + sl] = p[r][g];
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const float az0 = T_dz[sl * 33 + 2 * s + hi], az1 = T_dz[(32 + sl) * 33 + 2 * s + hi];
                const float bp = T_act[sl * 33 + 2 * s + hi];
                if (r == 0) { acc_bh2[0] += az0; acc_bh2[1] += az1; }
                acc_wh2[0][r] = MFMA(az0, bp, acc_wh2[0][r]);
                acc_wh2[1][r] = MFMA(az1, bp, acc_wh2[1][r]);
            }
        }
        // ---- d p = W2^T dZ2 ; dZ1 = d p * softplus'(p)
        f32x16 dp[2];
#pragma unroll
        for (int g = 0; g < 16; ++g) { dp[0][g] = 0.f; dp[1][g] = 0.f; }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int orow = 32 * r + rowc(g) + 4 * hi;
                dp[0] = MFMA(WH2[orow * 65 + sl], q[r][g], dp[0]);
                dp[1] = MFMA(WH2[orow * 65 + 32 + sl], q[r][g], dp[1]);
            }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int g = 0; g < 16; ++g) dp[r][g] *= RB ? s1[r][g] : dact_hidden(p[r][g], ak.hh);
        // ---- dW(head.w0) += dZ1 . V^T,  V = [base_out(16) | SH(16)]
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int g = 0; g < 16; ++g) T_dz[(32 * r + rowc(g) + 4 * hi) * 33 + sl] = dp[r][g];
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            T_act[(rowc(g) + 4 * hi) * 33 + sl] = lin_in<RB>(o[g]);
            T_act[(16 + 2 * g + hi) * 33 + sl] = lin_in<RB>(shs[g]);
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float az0 = T_dz[sl * 33 + 2 * s + hi], az1 = T_dz[(32 + sl) * 33 + 2 * s + hi];
            const float bv = T_act[sl * 33 + 2 * s + hi];
            acc_bh1[0] += az0; acc_bh1[1] += az1;
            acc_wh1[0] = MFMA(az0, bv, acc_wh1[0]);
            acc_wh1[1] = MFMA(az1, bv, acc_wh1[1]);
        }
        // ---- d V (rows 0..15 = d base_out) = WH1^T dZ1 ; row 0 takes the density gradient
        f32x16 dv;
#pragma unroll
        for (int g = 0; g < 16; ++g) dv[g] = 0.f;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int g = 0; g < 16; ++g)
                dv = MFMA(WH1[(32 * r + rowc(g) + 4 * hi) * 33 + sl], dp[r][g], dv);
        if (hi == 0) {
            // d sigma / d raw = exp(clamp(raw - 1, max=15)) * selector   (ngp.py:54-58,247-250)
            const float ds = live ? a.d_sigma[i] : 0.f;
            dv[0] = sel ? ds * dact_density(o[0], ak.dn) : 0.f;
        }
        {
            float *db = a.d_base + blk * (8 * 64) + lane;
#pragma unroll
            for (int g = 0; g < 8; ++g) db[g * 64] = dv[g];
        }
    }
    // ---- write this wave's partial weight gradients (head part of the parameter block)
    float *slab = a.slab + ((int64_t)blockIdx.x * 4 + wave) * (p_total(C) - P_BASE_N);
    constexpr int O = -P_BASE_N;                         // slab holds the head part only
#pragma unroll
    for (int ob = 0; ob < 2; ++ob) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int out = 32 * ob + rowc(g) + 4 * hi;
            slab[O + P_HW1 + out * 64 + sl] = acc_wh2[ob][0][g];
            slab[O + P_HW1 + out * 64 + 32 + sl] = acc_wh2[ob][1][g];
            // head.w0: v = sl -> column (v==0: none, v<16: 15+v, else v-16)
            if (sl != 0) slab[O + P_HW0 + out * 31 + (sl < 16 ? 15 + sl : sl - 16)] = acc_wh1[ob][g];
        }
        const float b2 = acc_bh2[ob] + __shfl_xor(acc_bh2[ob], 32, 64);
        const float b1 = acc_bh1[ob] + __shfl_xor(acc_bh1[ob], 32, 64);
        if (hi == 0) { slab[O + P_HB1 + 32 * ob + sl] = b2; slab[O + P_HB0 + 32 * ob + sl] = b1; }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            float v = acc_w3[c][k];
#pragma unroll
            for (int off = 1; off < 32; off <<= 1) v += __shfl_xor(v, off, 64);
            if (sl == 0) slab[O + P_HWO + c * 64 + 32 * (k >> 4) + rowc(k & 15) + 4 * hi] = v;
        }
        const float b3 = ren_wave_sum(acc_bh3[c]);
        if (lane == 0) slab[O + P_HWO + 64 * C + c] = b3;
    }
}

// ============================================================================ backward, base
struct BwdBArgs {
    const float *params, *feat, *d_base;
    int64_t n;
    float *dfeat, *slab;
    const float *acts;
    int act_code;
};

template <bool RB, bool SAVED>
__global__ __launch_bounds__(256, 1) void mlp_bwd_base_kernel(BwdBArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds_base[];
    fill_base(lds_base, a.params, LB_W1, LB_W2, LB_B1, LB_B2);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hi = lane >> 5, sl = lane & 31;
    float *T_a = lds_base + LB_END + wave * (96 * 33);           // [64][33]: h, later dZ0
    float *T_b = T_a + 64 * 33;                                    // [32][33]: d base_out, later x
    for (int k = lane; k < 32 * 33; k += 64) T_b[k] = 0.f;         // rows 16..31 stay zero
    __syncthreads();
    const int64_t n_blk = (a.n + 31) >> 5;
    const ActKinds ak = act_kinds(a.act_code);

    f32x16 acc_w2[2], acc_w1[2];
    float acc_b2 = 0.f, acc_b1[2] = {0.f, 0.f};
#pragma unroll
    for (int g = 0; g < 16; ++g) { acc_w2[0][g] = 0.f; acc_w2[1][g] = 0.f; acc_w1[0][g] = 0.f; acc_w1[1][g] = 0.f; }

    for (int64_t blk = (int64_t)blockIdx.x * 4 + wave; blk < n_blk; blk += (int64_t)gridDim.x * 4) {
        int zo = 0;                                     // defeat LICM of LDS weight reads (see forward)
        asm volatile("" : "+v"(zo));
        const float *lds = lds_base + zo;
        const float *W1 = lds + LB_W1, *W2 = lds + LB_W2;
        float x[16], dob[8];
        {
            const float *f = a.feat + blk * (16 * 64) + lane;
#pragma unroll
            for (int s = 0; s < 16; ++s) x[s] = lin_in<RB>(f[s * 64]);
            const float *db = a.d_base + blk * (8 * 64) + lane;
#pragma unroll
            for (int g = 0; g < 8; ++g) dob[g] = db[g * 64];
        }
        // ---- hidden layer: saved by the forward (SAVED) or recomputed
        f32x16 h[2];
        f32x16 s0[2];
        if (SAVED) {
            const float *ac = a.acts + blk * ACT_SAVE_FLOATS + lane;
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const float y = ac[(r * 16 + g) * 64];
                    if (RB) s0[r][g] = dact_hidden(y, ak.bh);
                    h[r][g] = lin_in<RB>(y);
                }
        } else {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            h[0][g] = lds[LB_B1 + rowc(g) + 4 * hi];
            h[1][g] = lds[LB_B1 + 32 + rowc(g) + 4 * hi];
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            h[0] = MFMA(W1[sl * 33 + 2 * s + hi], x[s], h[0]);
            h[1] = MFMA(W1[(32 + sl) * 33 + 2 * s + hi], x[s], h[1]);
        }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const float y = act_hidden(h[r][g], ak.bh);
                if (RB) s0[r][g] = dact_hidden(y, ak.bh);
                h[r][g] = lin_in<RB>(y);
            }
        }
        // ---- dW(base.wo) += dO . H^T
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int g = 0; g < 16; ++g) T_a[(32 * r + rowc(g) + 4 * hi) * 33 + sl] = h[r][g];
#pragma unroll
        for (int g = 0; g < 8; ++g) T_b[(rowc(g) + 4 * hi) * 33 + sl] = dob[g];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float az = T_b[sl * 33 + 2 * s + hi];
            acc_b2 += az;
            acc_w2[0] = MFMA(az, T_a[sl * 33 + 2 * s + hi], acc_w2[0]);
            acc_w2[1] = MFMA(az, T_a[(32 + sl) * 33 + 2 * s + hi], acc_w2[1]);
        }
        // ---- d h = Wo^T dO (only 16 real output rows = 8 k-steps); dZ0 = d h * softplus'(h)
        f32x16 dh[2];
#pragma unroll
        for (int g = 0; g < 16; ++g) { dh[0][g] = 0.f; dh[1][g] = 0.f; }
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const int orow = rowc(g) + 4 * hi;
            dh[0] = MFMA(W2[orow * 65 + sl], dob[g], dh[0]);
            dh[1] = MFMA(W2[orow * 65 + 32 + sl], dob[g], dh[1]);
        }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int g = 0; g < 16; ++g) dh[r][g] *= RB ? s0[r][g] : dact_hidden(h[r][g], ak.bh);
        // ---- dW(base.w0) += dZ0 . X^T
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int g = 0; g < 16; ++g) T_a[(32 * r + rowc(g) + 4 * hi) * 33 + sl] = dh[r][g];
#pragma unroll
        for (int s = 0; s < 16; ++s) T_b[(2 * s + hi) * 33 + sl] = x[s];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float az0 = T_a[sl * 33 + 2 * s + hi], az1 = T_a[(32 + sl) * 33 + 2 * s + hi];
            const float bx = T_b[sl * 33 + 2 * s + hi];
            acc_b1[0] += az0; acc_b1[1] += az1;
            acc_w1[0] = MFMA(az0, bx, acc_w1[0]);
            acc_w1[1] = MFMA(az1, bx, acc_w1[1]);
        }
        // T_b rows 16..31 now hold x; restore the zero padding the dO staging relies on
#pragma unroll
        for (int s = 8; s < 16; ++s) T_b[(2 * s + hi) * 33 + sl] = 0.f;
        // ---- d x = W0^T dZ0 -> hash-feature gradient, fragment layout
        f32x16 dxv;
#pragma unroll
        for (int g = 0; g < 16; ++g) dxv[g] = 0.f;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int g = 0; g < 16; ++g)
                dxv = MFMA(W1[(32 * r + rowc(g) + 4 * hi) * 33 + sl], dh[r][g], dxv);
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
    const float b2 = acc_b2 + __shfl_xor(acc_b2, 32, 64);
    if (hi == 0 && sl < 16) slab[P_BBO + sl] = b2;
#pragma unroll
    for (int ob = 0; ob < 2; ++ob) {
        const float b1 = acc_b1[ob] + __shfl_xor(acc_b1[ob], 32, 64);
        if (hi == 0) slab[P_BB0 + 32 * ob + sl] = b1;
    }
}

constexpr size_t FWD_LDS = (size_t)L_WEIGHTS_END * 4;
constexpr size_t BWD_H_LDS = (size_t)(LH_END + 4 * 96 * 33) * 4;          //  77 072 B -> 2 WG/CU
constexpr size_t BWD_B_LDS = (size_t)(LB_END + 4 * 96 * 33) * 4;          //  67 840 B -> 2 WG/CU

}  // namespace

extern "C" int64_t ren_mlp_bwd_workspace_floats(int32_t C) {
    if (C != 1 && C != 3) return -1;
    // d_base (fragment layout) is sized by the caller; this is the slab area only
    return (int64_t)GRID_H * 4 * (p_total(C) - P_BASE_N) + (int64_t)GRID_B * 4 * P_BASE_N;
}

static int mlp_fwd_impl(bool rb, const float *mlp_params, int32_t C, int32_t activations, const float *feat,
                        const ren_scene_desc *scene, const float *x_world, const float *dirs,
                        const float *rays_o, const float *rays_d, const int32_t *ray_indices,
                        const float *t_starts, const float *t_ends, int64_t n, int32_t density_only,
                        float *rgb, float *sigma, float *base_out, float *acts, void *stream) {
    if (!mlp_params || !feat || !scene || !sigma || n < 0 || activations < 0 || activations > 255) return REN_ERR_BAD_ARG;
    if (acts && (density_only || !base_out)) return REN_ERR_BAD_ARG;
    if (C != 1 && C != 3) return REN_ERR_UNSUPPORTED;          // robust_e_nerf.py:230-233
    if (!density_only && !rgb) return REN_ERR_BAD_ARG;
    if (!x_world && (!rays_o || !rays_d || !ray_indices || !t_starts || !t_ends)) return REN_ERR_BAD_ARG;
    if (n == 0) return REN_OK;
    FwdArgs a;
    a.params = mlp_params; a.feat = feat;
    a.src = SampleSrc{x_world, dirs, rays_o, rays_d, x_world ? nullptr : ray_indices, t_starts, t_ends};
    a.sc = ren_make_scene(scene);
    a.n = n; a.rgb = rgb; a.sigma = sigma; a.base_out = base_out; a.acts = acts;
    a.act_code = activations;
    const int64_t n_blk = (n + 31) / 32;
    int64_t blocks = (n_blk + 3) / 4;
    if (blocks > 768) blocks = 768;                            // 3 workgroups / CU (43.5 KB LDS each)
    dim3 grd((int)blocks), blk(256);
    hipStream_t st = (hipStream_t)stream;
#define REN_FWD(CC, DO)                                                                               \
    do {                                                                                              \
        if (rb) hipLaunchKernelGGL((mlp_fwd_kernel<CC, DO, true>), grd, blk, FWD_LDS, st, a);         \
        else    hipLaunchKernelGGL((mlp_fwd_kernel<CC, DO, false>), grd, blk, FWD_LDS, st, a);        \
    } while (0)
    if (density_only) REN_FWD(1, true);
    else if (C == 1)  REN_FWD(1, false);
    else              REN_FWD(3, false);
#undef REN_FWD
    REN_CHECK_LAUNCH();
}

extern "C" int ren_mlp_fwd(const float *mlp_params, int32_t C, int32_t activations, const float *feat,
                           const ren_scene_desc *scene, const float *x_world, const float *dirs,
                           const float *rays_o, const float *rays_d, const int32_t *ray_indices,
                           const float *t_starts, const float *t_ends, int64_t n, int32_t density_only,
                           float *rgb, float *sigma, float *base_out, void *stream) {
    return mlp_fwd_impl(false, mlp_params, C, activations, feat, scene, x_world, dirs, rays_o, rays_d, ray_indices, t_starts, t_ends,
                        n, density_only, rgb, sigma, base_out, nullptr, stream);
}

extern "C" int64_t ren_mlp_act_save_floats(int64_t n) { return n < 0 ? -1 : (n + 31) / 32 * ACT_SAVE_FLOATS; }

extern "C" int ren_mlp_fwd_save(const float *mlp_params, int32_t C, int32_t activations, int32_t bf16, const float *feat,
                                const ren_scene_desc *scene, const float *x_world, const float *dirs,
                                const float *rays_o, const float *rays_d, const int32_t *ray_indices,
                                const float *t_starts, const float *t_ends, int64_t n, float *rgb, float *sigma,
                                float *base_out, float *act_save, void *stream) {
    if (!act_save) return REN_ERR_BAD_ARG;
    return mlp_fwd_impl(bf16 != 0, mlp_params, C, activations, feat, scene, x_world, dirs, rays_o, rays_d, ray_indices, t_starts,
                        t_ends, n, 0, rgb, sigma, base_out, act_save, stream);
}

extern "C" int ren_mlp_fwd_bf16(const float *mlp_params_bf16, int32_t C, int32_t activations, const float *feat,
                                const ren_scene_desc *scene, const float *x_world, const float *dirs,
                                const float *rays_o, const float *rays_d, const int32_t *ray_indices,
                                const float *t_starts, const float *t_ends, int64_t n, int32_t density_only,
                                float *rgb, float *sigma, float *base_out, void *stream) {
    return mlp_fwd_impl(true, mlp_params_bf16, C, activations, feat, scene, x_world, dirs, rays_o, rays_d, ray_indices, t_starts,
                        t_ends, n, density_only, rgb, sigma, base_out, nullptr, stream);
}

static int mlp_bwd_impl(bool rb, const float *mlp_params, int32_t C, int32_t activations, const float *feat, const float *base_out,
                        const ren_scene_desc *scene, const float *x_world, const float *dirs,
                        const float *rays_o, const float *rays_d, const int32_t *ray_indices,
                        const float *t_starts, const float *t_ends, int64_t n, const float *rgb,
                        const float *d_rgb, const float *d_sigma, float *d_base, float *dfeat,
                        float *grad_mlp_params, float *workspace, const float *acts, void *stream) {
    if (!mlp_params || !feat || !base_out || !scene || !rgb || !d_rgb || !d_sigma || !d_base || !dfeat ||
        !grad_mlp_params || !workspace || n < 0 || activations < 0 || activations > 255)
        return REN_ERR_BAD_ARG;
    if (C != 1 && C != 3) return REN_ERR_UNSUPPORTED;
    if (!x_world && (!rays_o || !rays_d || !ray_indices || !t_starts || !t_ends)) return REN_ERR_BAD_ARG;
    if (n == 0) return REN_OK;
    hipStream_t st = (hipStream_t)stream;
    // > 64 KiB of dynamic LDS needs the attribute; setting it is idempotent (no library state)
#define REN_BWD_ATTR(K) (void)hipFuncSetAttribute((const void *)K, hipFuncAttributeMaxDynamicSharedMemorySize, (int)BWD_H_LDS)
    REN_BWD_ATTR((mlp_bwd_head_kernel<1, false, false>)); REN_BWD_ATTR((mlp_bwd_head_kernel<3, false, false>));
    REN_BWD_ATTR((mlp_bwd_head_kernel<1, true, false>));  REN_BWD_ATTR((mlp_bwd_head_kernel<3, true, false>));
    REN_BWD_ATTR((mlp_bwd_head_kernel<1, false, true>));  REN_BWD_ATTR((mlp_bwd_head_kernel<3, false, true>));
    REN_BWD_ATTR((mlp_bwd_head_kernel<1, true, true>));   REN_BWD_ATTR((mlp_bwd_head_kernel<3, true, true>));
#undef REN_BWD_ATTR
#define REN_BWD_ATTR(K) (void)hipFuncSetAttribute((const void *)K, hipFuncAttributeMaxDynamicSharedMemorySize, (int)BWD_B_LDS)
    REN_BWD_ATTR((mlp_bwd_base_kernel<false, false>)); REN_BWD_ATTR((mlp_bwd_base_kernel<true, false>));
    REN_BWD_ATTR((mlp_bwd_base_kernel<false, true>));  REN_BWD_ATTR((mlp_bwd_base_kernel<true, true>));
#undef REN_BWD_ATTR
    const int head_len = p_total(C) - P_BASE_N;
    float *slab_h = workspace, *slab_b = workspace + (int64_t)GRID_H * 4 * head_len;
    BwdHArgs h;
    h.params = mlp_params; h.base_out = base_out;
    h.src = SampleSrc{x_world, dirs, rays_o, rays_d, x_world ? nullptr : ray_indices, t_starts, t_ends};
    h.sc = ren_make_scene(scene);
    h.n = n; h.rgb = rgb; h.d_rgb = d_rgb; h.d_sigma = d_sigma; h.d_base = d_base; h.slab = slab_h; h.acts = acts;
    h.act_code = activations;
    const bool sv = acts != nullptr;
#define REN_HEAD(CC, R, S) hipLaunchKernelGGL((mlp_bwd_head_kernel<CC, R, S>), dim3(GRID_H), dim3(256), BWD_H_LDS, st, h)
    if (C == 1) {
        if (rb) { if (sv) REN_HEAD(1, true, true); else REN_HEAD(1, true, false); }
        else    { if (sv) REN_HEAD(1, false, true); else REN_HEAD(1, false, false); }
    } else {
        if (rb) { if (sv) REN_HEAD(3, true, true); else REN_HEAD(3, true, false); }
        else    { if (sv) REN_HEAD(3, false, true); else REN_HEAD(3, false, false); }
    }
#undef REN_HEAD
    BwdBArgs b;
    b.params = mlp_params; b.feat = feat; b.d_base = d_base; b.n = n; b.dfeat = dfeat; b.slab = slab_b; b.acts = acts;
    b.act_code = activations;
#define REN_BASE(R, S) hipLaunchKernelGGL((mlp_bwd_base_kernel<R, S>), dim3(GRID_B), dim3(256), BWD_B_LDS, st, b)
    if (rb) { if (sv) REN_BASE(true, true); else REN_BASE(true, false); }
    else    { if (sv) REN_BASE(false, true); else REN_BASE(false, false); }
#undef REN_BASE
    launch_reduce_slabs(slab_h, GRID_H * 4, head_len, grad_mlp_params + P_BASE_N, st);
    launch_reduce_slabs(slab_b, GRID_B * 4, P_BASE_N, grad_mlp_params, st);
    REN_CHECK_LAUNCH();
}

extern "C" int ren_mlp_bwd(const float *mlp_params, int32_t C, int32_t activations, const float *feat, const float *base_out,
                           const ren_scene_desc *scene, const float *x_world, const float *dirs,
                           const float *rays_o, const float *rays_d, const int32_t *ray_indices,
                           const float *t_starts, const float *t_ends, int64_t n, const float *rgb,
                           const float *d_rgb, const float *d_sigma, float *d_base, float *dfeat,
                           float *grad_mlp_params, float *workspace, void *stream) {
    return mlp_bwd_impl(false, mlp_params, C, activations, feat, base_out, scene, x_world, dirs, rays_o, rays_d, ray_indices, t_starts,
                        t_ends, n, rgb, d_rgb, d_sigma, d_base, dfeat, grad_mlp_params, workspace, nullptr, stream);
}

extern "C" int ren_mlp_bwd_bf16(const float *mlp_params_bf16, int32_t C, int32_t activations, const float *feat, const float *base_out,
                                const ren_scene_desc *scene, const float *x_world, const float *dirs,
                                const float *rays_o, const float *rays_d, const int32_t *ray_indices,
                                const float *t_starts, const float *t_ends, int64_t n, const float *rgb,
                                const float *d_rgb, const float *d_sigma, float *d_base, float *dfeat,
                                float *grad_mlp_params, float *workspace, void *stream) {
    return mlp_bwd_impl(true, mlp_params_bf16, C, activations, feat, base_out, scene, x_world, dirs, rays_o, rays_d, ray_indices,
                        t_starts, t_ends, n, rgb, d_rgb, d_sigma, d_base, dfeat, grad_mlp_params, workspace, nullptr, stream);
}

extern "C" int ren_mlp_bwd_saved(const float *mlp_params, int32_t C, int32_t activations, int32_t bf16, const float *feat,
                                 const float *base_out, const float *act_save, const ren_scene_desc *scene,
                                 const float *x_world, const float *dirs, const float *rays_o, const float *rays_d,
                                 const int32_t *ray_indices, const float *t_starts, const float *t_ends, int64_t n,
                                 const float *rgb, const float *d_rgb, const float *d_sigma, float *d_base,
                                 float *dfeat, float *grad_mlp_params, float *workspace, void *stream) {
    if (!act_save) return REN_ERR_BAD_ARG;
    return mlp_bwd_impl(bf16 != 0, mlp_params, C, activations, feat, base_out, scene, x_world, dirs, rays_o, rays_d, ray_indices,
                        t_starts, t_ends, n, rgb, d_rgb, d_sigma, d_base, dfeat, grad_mlp_params, workspace, act_save,
                        stream);
}
