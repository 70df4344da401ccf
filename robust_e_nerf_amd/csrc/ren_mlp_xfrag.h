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
template <int NT, int LAYER>
__device__ void fill_frags(__bf16 *frag, const float *__restrict__ P, int tiles, int chunks) {
    const int total = tiles * chunks * 64 * 8;
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
        const int j = e & 7, lane = (e >> 3) & 63, tc = e >> 9;
        const int c = tc % chunks, t = tc / chunks;
        const float w = w_of<LAYER>(P, t * 32 + (lane & 31), k_of<LAYER>(c, lane >> 5, j));
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

// ---- weight-gradient staging (backward kernels) --------------------------------------------------------------
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
        if (NP == 2) { acc = MFMAB(az[1], ba[0], acc); acc = MFMAB(az[0], ba[1], acc); }
        acc = MFMAB(az[0], ba[0], acc);
    }
}

// transposed fragments: rows = INPUT index of the layer (tile t_in), k-slots = output neurons in D-layout order
// layer ids as above; value = W(out = ksrc_h(c, hi, j), in = row)
template <int NT, int LAYER>
__device__ void fill_frags_t(__bf16 *frag, const float *__restrict__ P, int tiles, int chunks) {
    const int total = tiles * chunks * 64 * 8;
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
        const int j = e & 7, lane = (e >> 3) & 63, tc = e >> 9;
        const int c = tc % chunks, t = tc / chunks;
        const int row = t * 32 + (lane & 31), hi = lane >> 5;
        // base.wo has 16 real outputs: its single chunk maps slot j -> neuron rowc(j) + 4 hi
        const int out = LAYER == 1 ? rowc(j) + 4 * hi : ksrc_h(c, hi, j);
        const float w = w_of<LAYER>(P, out, row);
        __bf16 s[3];
        split<NT>(w, s);
#pragma unroll
        for (int k = 0; k < NT; ++k) frag[(((t * chunks + c) * NT + k) * 64 + lane) * 8 + j] = s[k];
    }
}

}  // namespace
