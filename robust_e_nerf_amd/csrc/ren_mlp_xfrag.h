// Split-bf16 operand fragments, MFMA chains and weight-gradient staging shared by the fused NGP MLP kernels on the
// bf16 matrix cores (ren_mlp_x.hip: value path; ren_mlp_jvp_x.hip: value + tangent).  Internal; the design notes
// are at the top of ren_mlp_x.hip.
#pragma once
#include "ren_mlp_common.h"

namespace {

// ---- logical weights (torch layouts inside the parameter block) ---------------------------------------------
__device__ __forceinline__ float w_base0(const float *P, int out, int k) { return P[P_BW0 + out * 32 + k]; }
__device__ __forceinline__ float w_base1(const float *P, int out, int k) { return out < 16 ? P[P_BWO + out * 64 + k] : 0.f; }
// head layer 0 input order v: 0 = sigma slot (weight 0), 1..15 geo features, 16..31 SH components (ngp.py:244-259)
__device__ __forceinline__ float w_head0(const float *P, int out, int v) {
    return v == 0 ? 0.f : v < 16 ? P[P_HW0 + out * 31 + 15 + v] : P[P_HW0 + out * 31 + (v - 16)];
}
__device__ __forceinline__ float w_head1(const float *P, int out, int k) { return P[P_HW1 + out * 64 + k]; }

// k-slot (chunk c, lane half hi, element j) -> input index of the layer
__device__ __forceinline__ int ksrc_x(int c, int hi, int j) { return 2 * (8 * c + j) + hi; }                   // hash features
__device__ __forceinline__ int ksrc_h(int c, int hi, int j) { return 32 * (c >> 1) + rowc(8 * (c & 1) + j) + 4 * hi; }  // D layout
__device__ __forceinline__ int ksrc_v(int c, int hi, int j) { return c == 0 ? rowc(j) + 4 * hi : 16 + 2 * j + hi; }      // [base | SH]

// layer ids: 0 base.w0 (x input), 1 base.wo (h input), 2 head.w0 (v input), 3 head.w1 (p input)
template <int LAYER>
__device__ __forceinline__ float w_of(const float *P, int out, int k) {
    return LAYER == 0 ? w_base0(P, out, k) : LAYER == 1 ? w_base1(P, out, k) : LAYER == 2 ? w_head0(P, out, k) : w_head1(P, out, k);
}
template <int LAYER>
__device__ __forceinline__ int k_of(int c, int hi, int j) {
    return LAYER == 0 ? ksrc_x(c, hi, j) : LAYER == 2 ? ksrc_v(c, hi, j) : ksrc_h(c, hi, j);
}

// fragment store: frag[((t * NC + c) * NT + term) * 64 + lane] (8 bf16 each)
// NEG_ODD: the fragments of odd k-chunks are stored negated.  The bf16 MFMA's fp32 accumulation rounds toward -inf below
// ~2^-27 of its largest addend (tools/mlp_bias_probe.py: element errors of 2e-7 with a mean of -0.03 rms, which grows like
// N in a sum over N samples); a layer that accumulates its even chunks in one accumulator and the negated odd chunks in a
// second one and SUBTRACTS them (mma1x2 call sites) carries the two biases with opposite signs.
template <int NT, int LAYER, bool NEG_ODD = false>
__device__ void fill_frags(__bf16 *frag, const float *__restrict__ P, int tiles, int chunks) {
    const int total = tiles * chunks * 64 * 8;
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
        const int j = e & 7, lane = (e >> 3) & 63, tc = e >> 9;
        const int c = tc % chunks, t = tc / chunks;
        float w = w_of<LAYER>(P, t * 32 + (lane & 31), k_of<LAYER>(c, lane >> 5, j));
        if (NEG_ODD && (c & 1)) w = -w;
        __bf16 s[3];
        split<NT>(w, s);
#pragma unroll
        for (int k = 0; k < NT; ++k) frag[(((t * chunks + c) * NT + k) * 64 + lane) * 8 + j] = s[k];
    }
}

template <int NT>
__device__ __forceinline__ bf16x8 ldfrag(const __bf16 *frag, int t, int chunks, int c, int term, int lane) {
    return *reinterpret_cast<const bf16x8 *>(frag + (((t * chunks + c) * NT + term) * 64 + lane) * 8);
}

// acc += W(tile t, chunk c) . B   over the MODE's term pairs
template <int MODE>
__device__ __forceinline__ void mma(f32x16 &acc, const __bf16 *frag, int t, int chunks, int c, const bf16x8 (&b)[3], int lane) {
    using PR = Pairs<MODE>;
#pragma unroll
    for (int k = 0; k < PR::N; ++k)
        acc = MFMAB(ldfrag<PR::NT>(frag, t, chunks, c, PR::W[k], lane), b[PR::A[k]], acc);
}

// two output tiles that share the B operand, interleaved: consecutive MFMAs never hit the same accumulator,
// so the matrix pipe does not wait on the dependent-accumulate latency
template <int MODE>
__device__ __forceinline__ void mma2(f32x16 &acc0, f32x16 &acc1, const __bf16 *frag, int chunks, int c,
                                     const bf16x8 (&b)[3], int lane) {
    using PR = Pairs<MODE>;
#pragma unroll
    for (int k = 0; k < PR::N; ++k) {
        acc0 = MFMAB(ldfrag<PR::NT>(frag, 0, chunks, c, PR::W[k], lane), b[PR::A[k]], acc0);
        acc1 = MFMAB(ldfrag<PR::NT>(frag, 1, chunks, c, PR::W[k], lane), b[PR::A[k]], acc1);
    }
}
// mma2 over a chunk PAIR (ca even, cb odd: cb's fragments are stored negated, fill_frags NEG_ODD) into separate
// accumulators per chunk parity: the caller subtracts n from p, which cancels the accumulate's rounding bias
template <int MODE>
__device__ __forceinline__ void mma2_pn(f32x16 &p0, f32x16 &p1, f32x16 &n0, f32x16 &n1, const __bf16 *frag, int chunks,
                                        int ca, int cb, const bf16x8 (&ba)[3], const bf16x8 (&bb)[3], int lane) {
    using PR = Pairs<MODE>;
#pragma unroll
    for (int k = 0; k < PR::N; ++k) {
        p0 = MFMAB(ldfrag<PR::NT>(frag, 0, chunks, ca, PR::W[k], lane), ba[PR::A[k]], p0);
        p1 = MFMAB(ldfrag<PR::NT>(frag, 1, chunks, ca, PR::W[k], lane), ba[PR::A[k]], p1);
        n0 = MFMAB(ldfrag<PR::NT>(frag, 0, chunks, cb, PR::W[k], lane), bb[PR::A[k]], n0);
        n1 = MFMAB(ldfrag<PR::NT>(frag, 1, chunks, cb, PR::W[k], lane), bb[PR::A[k]], n1);
    }
}
// one output tile, two k-chunks into two independent partial accumulators
template <int MODE>
__device__ __forceinline__ void mma1x2(f32x16 &acca, f32x16 &accb, const __bf16 *frag, int chunks, int ca, int cb,
                                       const bf16x8 (&ba)[3], const bf16x8 (&bb)[3], int lane) {
    using PR = Pairs<MODE>;
#pragma unroll
    for (int k = 0; k < PR::N; ++k) {
        acca = MFMAB(ldfrag<PR::NT>(frag, 0, chunks, ca, PR::W[k], lane), ba[PR::A[k]], acca);
        accb = MFMAB(ldfrag<PR::NT>(frag, 0, chunks, cb, PR::W[k], lane), bb[PR::A[k]], accb);
    }
}

// ---- weight-gradient staging through LDS (the tangent kernels of ren_mlp_jvp_x.hip; the value kernels of ren_mlp_x.hip
// transpose on the matrix cores instead, see below) ---------------------------------------------------------------
constexpr int ST = 40;                                   // staging row stride in bf16 (32 samples + pad, 80 B)
template <int NP> struct StageBytes { static constexpr int TILE = NP * 32 * ST * 2; };   // one 32-neuron tile

// write 16 D-layout register values (neurons rowc(g) + 4 hi of one tile) as NP bf16 pieces
template <int NP>
__device__ __forceinline__ void stage_tile(__bf16 *T, const float *v, int hi, int sl) {
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        __bf16 s[3];
        split<NP>(v[g], s);
#pragma unroll
        for (int k = 0; k < NP; ++k) T[(k * 32 + rowc(g) + 4 * hi) * ST + sl] = s[k];
    }
}
// the same from already split operands (b0 = registers 0..7, b1 = registers 8..15 of the tile): the first NP
// pieces of the 3-way split ARE the NP-way split, so values that also feed a data-gradient MFMA are split once
template <int NP>
__device__ __forceinline__ void stage_pieces(__bf16 *T, const bf16x8 (&b0)[3], const bf16x8 (&b1)[3], int hi, int sl) {
#pragma unroll
    for (int k = 0; k < NP; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            T[(k * 32 + rowc(j) + 4 * hi) * ST + sl] = b0[k][j];
            T[(k * 32 + rowc(8 + j) + 4 * hi) * ST + sl] = b1[k][j];
        }
}
// write one value for an explicit neuron row
template <int NP>
__device__ __forceinline__ void stage_one(__bf16 *T, int row, float v, int sl) {
    __bf16 s[3];
    split<NP>(v, s);
#pragma unroll
    for (int k = 0; k < NP; ++k) T[(k * 32 + row) * ST + sl] = s[k];
}

// acc[32 x 32] += Tz(32 neurons x 32 samples) . Ta(32 neurons x 32 samples)^T
template <int NP>
__device__ __forceinline__ void dw_tile(f32x16 &acc, const __bf16 *Tz, const __bf16 *Ta, int hi, int sl) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        bf16x8 az[NP], ba[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            az[k] = *reinterpret_cast<const bf16x8 *>(Tz + (k * 32 + sl) * ST + 16 * c + 8 * hi);
            ba[k] = *reinterpret_cast<const bf16x8 *>(Ta + (k * 32 + sl) * ST + 16 * c + 8 * hi);
        }
        if (NP == 3) {                                    // six terms, smallest first: fp32 round-off per product
            acc = MFMAB(az[2], ba[0], acc); acc = MFMAB(az[0], ba[2], acc); acc = MFMAB(az[1], ba[1], acc);
        }
        if (NP >= 2) { acc = MFMAB(az[1], ba[0], acc); acc = MFMAB(az[0], ba[1], acc); }
        acc = MFMAB(az[0], ba[0], acc);
    }
}

// ---- weight-gradient operands without LDS: transposition on the matrix cores ----------------------------------
// A weight gradient sums dz[out][s] * act[in][s] over SAMPLES, so both operands are needed with samples in the
// k-slots, while the chain keeps lane = sample.  T[s][n] = sum_k X[s][k] Sel[k][n] with X = a chain operand (lane =
// sample, slots = 16 neurons of a tile: it has the A-operand register format as it stands) and Sel a 0/1
// selection matrix moves the tile into the D layout with lane = neuron n and registers = samples rowc(g) + 4 hi:
// exactly a k = sample operand (chunk c = registers 8c..8c+7; dz and act go through the same map, so the sum
// runs over every sample once).  Pieces are bf16 values and Sel is 0/1: the products and sums are exact, the
// repack to bf16 is exact.  2 MFMAs per piece and tile replace 16 two-byte LDS stores per piece and the b128
// reads, and the staging tiles (60 KB per workgroup at three pieces) leave the LDS.
// KIND 0: n == rowc(j) + 4 hi (registers 0..7 of a D tile)   1: n == rowc(8 + j) + 4 hi (registers 8..15)
//      2: n == 2 j + hi (hash features 0..15)                 3: n == 16 + 2 j + hi (hash features 16..31, SH slots)
template <int KIND>
__device__ __forceinline__ bf16x8 make_sel(int lane) {
    const int n = lane & 31, hi = lane >> 5;
    bf16x8 s;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int col = KIND == 0 ? rowc(j) + 4 * hi : KIND == 1 ? rowc(8 + j) + 4 * hi : KIND == 2 ? 2 * j + hi : 16 + 2 * j + hi;
        s[j] = (__bf16)(col == n ? 1.f : 0.f);
    }
    return s;
}

// out[c][k]: piece k of the transposed tile, sample chunk c.  b0 / b1: the tile's two chain operands (k-chunks).
template <int NT>
__device__ __forceinline__ void transpose_tile(const bf16x8 (&b0)[3], const bf16x8 (&b1)[3], const bf16x8 &s0,
                                               const bf16x8 &s1, bf16x8 (&out)[2][3]) {
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        f32x16 t;
#pragma unroll
        for (int g = 0; g < 16; ++g) t[g] = 0.f;
        t = MFMAB(b0[k], s0, t);
        t = MFMAB(b1[k], s1, t);
#pragma unroll
        for (int j = 0; j < 8; ++j) { out[0][k][j] = (__bf16)t[j]; out[1][k][j] = (__bf16)t[8 + j]; }
    }
}
// one chain operand only (a tile with 16 real rows: the base MLP's output gradient)
template <int NT>
__device__ __forceinline__ void transpose_half(const bf16x8 (&b0)[3], const bf16x8 &s0, bf16x8 (&out)[2][3]) {
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        f32x16 t;
#pragma unroll
        for (int g = 0; g < 16; ++g) t[g] = 0.f;
        t = MFMAB(b0[k], s0, t);
#pragma unroll
        for (int j = 0; j < 8; ++j) { out[0][k][j] = (__bf16)t[j]; out[1][k][j] = (__bf16)t[8 + j]; }
    }
}

// acc[out][in] += sum_s z[out][s] a[in][s] over the tile's 32 samples, all of the MODE's term pairs (fp32 round-off
// in MODE 6, one bf16 product in MODE 1)
template <int MODE>
__device__ __forceinline__ void dw_acc(f32x16 &acc, const bf16x8 (&z)[2][3], const bf16x8 (&a)[2][3]) {
    using PR = Pairs<MODE>;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int k = 0; k < PR::N; ++k) acc = MFMAB(z[c][PR::W[k]], a[c][PR::A[k]], acc);
}
// two input tiles that share the dz operand, interleaved (no back-to-back MFMAs on one accumulator)
template <int MODE>
__device__ __forceinline__ void dw_acc2(f32x16 &acc0, f32x16 &acc1, const bf16x8 (&z)[2][3], const bf16x8 (&a0)[2][3],
                                        const bf16x8 (&a1)[2][3]) {
    using PR = Pairs<MODE>;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int k = 0; k < PR::N; ++k) {
            acc0 = MFMAB(z[c][PR::W[k]], a0[c][PR::A[k]], acc0);
            acc1 = MFMAB(z[c][PR::W[k]], a1[c][PR::A[k]], acc1);
        }
}

// The block's contribution is formed in a FRESH accumulator and added to the persistent one on the VALU (round to
// nearest): chained straight into the persistent accumulator every MFMA would round toward -inf relative to the running
// sum's magnitude, 12 times per block -- a bias that grows with the number of blocks (measured 1e-5 of the gradient at
// 8.4 M samples); relative to one block's contribution it is 16 x smaller and does not accumulate coherently.
template <int MODE>
__device__ __forceinline__ void dw_block(f32x16 &acc, const bf16x8 (&z)[2][3], const bf16x8 (&a)[2][3]) {
    f32x16 t;
#pragma unroll
    for (int g = 0; g < 16; ++g) t[g] = 0.f;
    dw_acc<MODE>(t, z, a);
#pragma unroll
    for (int g = 0; g < 16; ++g) acc[g] += t[g];
}
template <int MODE>
__device__ __forceinline__ void dw_block2(f32x16 &acc0, f32x16 &acc1, const bf16x8 (&z)[2][3], const bf16x8 (&a0)[2][3],
                                          const bf16x8 (&a1)[2][3]) {
    f32x16 t0, t1;
#pragma unroll
    for (int g = 0; g < 16; ++g) { t0[g] = 0.f; t1[g] = 0.f; }
    dw_acc2<MODE>(t0, t1, z, a0, a1);
#pragma unroll
    for (int g = 0; g < 16; ++g) { acc0[g] += t0[g]; acc1[g] += t1[g]; }
}

// transposed fragments: rows = INPUT index of the layer (tile t_in), k-slots = output neurons in D-layout order
// layer ids as above; value = W(out = ksrc_h(c, hi, j), in = row)
template <int NT, int LAYER, bool NEG_ODD = false>
__device__ void fill_frags_t(__bf16 *frag, const float *__restrict__ P, int tiles, int chunks) {
    const int total = tiles * chunks * 64 * 8;
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
        const int j = e & 7, lane = (e >> 3) & 63, tc = e >> 9;
        const int c = tc % chunks, t = tc / chunks;
        const int row = t * 32 + (lane & 31), hi = lane >> 5;
        // base.wo has 16 real outputs: its single chunk maps slot j -> neuron rowc(j) + 4 hi
        const int out = LAYER == 1 ? rowc(j) + 4 * hi : ksrc_h(c, hi, j);
        float w = w_of<LAYER>(P, out, row);
        if (NEG_ODD && (c & 1)) w = -w;
        __bf16 s[3];
        split<NT>(w, s);
#pragma unroll
        for (int k = 0; k < NT; ++k) frag[(((t * chunks + c) * NT + k) * 64 + lane) * 8 + j] = s[k];
    }
}

}  // namespace
