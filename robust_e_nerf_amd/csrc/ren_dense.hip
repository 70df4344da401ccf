// Vanilla-NeRF radiance field (`arch: mlp`): frequency positional encoding and dense layers on the f32
// matrix cores.  Replaces SinusoidalEncoder / MLP / NerfMLP / VanillaNeRFRadianceField
// (robust_e_nerf/external/mlp.py:26-113,126-205,208-243,246-358): every nn.Linear (+ activation) of the
// 8 x 256 trunk, the sigma / bottleneck layers and the 283 -> 128 -> C colour head is one launch of
// `dense_kernel` (forward, or backward-data with the previous layer's activation derivative fused into the
// epilogue) and one of `dense_dw_kernel` (weight/bias gradient, slab-reduced: deterministic, no atomics).
//
// Layout: activations are row-major [n_pad][ld] f32 with zero padding columns up to a multiple of 32
// (concatenations are column ranges of one buffer: [h4 | enc] and [bottleneck | view enc] are never
// copied).  As in ren_mlp.hip a wavefront owns 32 samples = the MFMA columns; the weight chunk
// W[:, k0:k0+32] is staged in LDS (double buffered, row stride 33 words: conflict-free for the A-operand
// read) and shared by the four waves of the workgroup; the 32 x N output tile of a wave lives in
// N/32 x 16 accumulator registers.  This is 593 k MACs per sample -- the one MFMA-bound variant of the
// model; exact fp32 (v_mfma_f32_32x32x2_f32), bf16 mode is a later round.
#include "ren_mlp_common.h"

namespace {

constexpr int ACT_NONE = 0, ACT_SOFTPLUS100 = 1, ACT_SOFTPLUS1 = 2, ACT_TRUNC_EXP_SEL = 3;
// the YAML's alternatives (models/nerf.py:8-29): relu hidden layers, sigmoid radiance, softplus / shifted_softplus densities
constexpr int ACT_RELU = 4, ACT_SIGMOID = 5, ACT_SOFTPLUS1_SEL = 6, ACT_SHIFTED_SOFTPLUS1_SEL = 7, ACT_LAST = 7;
__device__ __forceinline__ bool act_uses_sel(int act) { return act == ACT_TRUNC_EXP_SEL || act >= ACT_SOFTPLUS1_SEL; }
__device__ __forceinline__ float dense_act(float z, int act, bool selv) {
    if (act == ACT_SOFTPLUS100) return softplus100(z);
    if (act == ACT_SOFTPLUS1) return softplus1(z);
    if (act == ACT_TRUNC_EXP_SEL) return selv ? __expf(z - 1.f) : 0.f;                            // ngp.py:45-65
    if (act == ACT_RELU) return fmaxf(z, 0.f);
    if (act == ACT_SIGMOID) return 1.f / (1.f + __expf(-z));
    if (act == ACT_SOFTPLUS1_SEL) return selv ? softplus1(z) : 0.f;
    if (act == ACT_SHIFTED_SOFTPLUS1_SEL) return selv ? softplus1(z - 1.f) : 0.f;                  // nerf.py:8-13
    return z;
}
// derivative of a hidden activation through its OUTPUT (backward-data epilogue, tangent algebra)
__device__ __forceinline__ bool act_has_dout(int act) { return act == ACT_SOFTPLUS100 || act == ACT_RELU; }
__device__ __forceinline__ float dense_dact_from_out(float y, int act) {
    return act == ACT_RELU ? (y > 0.f ? 1.f : 0.f) : dsoftplus_from_out(y, 100.f);
}
// beta of the tangent-algebra launches: 0 selects relu (s = [y > 0], s' = 0 falls out of beta s (1 - s))
__device__ __forceinline__ float dact_beta(float y, float beta) { return beta == 0.f ? (y > 0.f ? 1.f : 0.f) : dsoftplus_from_out(y, beta); }
constexpr int KC = 32;                                 // reduction chunk staged in LDS

struct DenseArgs {
    const float *X; int ldx;                           // [n_pad][ldx]: B operand rows (inputs, or dZ for backward-data)
    const float *W; int w_rows, w_cols;                // torch nn.Linear weight [w_rows][w_cols]
    int red;                                           // reduction length: w_cols (forward) / w_rows (backward)
    int groups;                                        // output groups of 128 interleaved over blockIdx.x (1 or 2)
    int out0;                                          // first output feature of this launch (wide layers run as two
                                                       // 128-output launches: 64 accumulator registers, two waves/SIMD)
    int n_out;                                         // outputs stored (absolute bound)
    const float *bias;                                 // forward only (may be null)
    int act;                                           // forward: this layer's activation; backward: the PREVIOUS layer's
    const float *Yprev; int ldyp;                      // backward: saved outputs of the previous layer
    const uint8_t *sel;                                // ACT_TRUNC_EXP_SEL: per-sample selector
    int accumulate;                                    // backward: add to what is already in Y before the derivative
    float *Y; int ldy;
    int64_t n;
};

template <int NT, bool BWD>
__device__ __forceinline__ void load_chunk(const DenseArgs &a, int c, float (&pre)[NT * 4]) {
    const int k0 = c * KC;
#pragma unroll
    for (int j = 0; j < NT * 4; ++j) {
        const int e = threadIdx.x + 256 * j;
        float v = 0.f;
        if (!BWD) {                                    // out row = W row, reduction col = W col
            const int row = a.out0 + (e >> 5), col = e & 31;
            if (row < a.w_rows && k0 + col < a.w_cols) v = a.W[(int64_t)row * a.w_cols + k0 + col];
        } else {                                       // out row = W col, reduction col = W row (W^T), coalesced along W cols
            const int col = e / (NT * 32), row = a.out0 + e % (NT * 32);
            if (k0 + col < a.w_rows && row < a.w_cols) v = a.W[(int64_t)(k0 + col) * a.w_cols + row];
        }
        pre[j] = v;
    }
}

template <int NT, bool BWD>
__device__ __forceinline__ void store_chunk(float *buf, const float (&pre)[NT * 4]) {
#pragma unroll
    for (int j = 0; j < NT * 4; ++j) {
        const int e = threadIdx.x + 256 * j;
        const int row = BWD ? e % (NT * 32) : e >> 5, col = BWD ? e / (NT * 32) : e & 31;
        buf[row * 33 + col] = pre[j];
    }
}

// bias / activation (forward) or accumulate + previous layer's activation derivative (backward-data), row-major
// float4 stores; shared by the f32 and the split-bf16 kernels (same accumulator layout)
template <int NT, bool BWD>
__device__ __forceinline__ void dense_epilogue(const DenseArgs &a, f32x16 (&acc)[NT], bool active, int64_t row, int hi) {
    if (!active || row >= ((a.n + 31) & ~(int64_t)31)) return;
    const bool live = row < a.n;
    const bool selv = (!BWD && act_uses_sel(a.act) && live) ? a.sel[row] != 0 : false;
    // epilogue in two halves of NT/2 tiles: all loads of a half (saved activations / accumulate target) are
    // issued as float4 before any arithmetic, so their latencies overlap instead of queueing per element
    constexpr int HT = NT > 1 ? NT / 2 : 1;
#pragma unroll
    for (int half = 0; half < NT / HT; ++half) {
        float4 yp4[HT][4], yo4[HT][4];
        if (BWD) {
#pragma unroll
            for (int u = 0; u < HT; ++u)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int o0 = a.out0 + (half * HT + u) * 32 + 8 * q + 4 * hi;
                    const bool in = o0 + 3 < a.n_out;                 // n_out is a multiple of 4 on this path (host check)
                    yp4[u][q] = (in && act_has_dout(a.act)) ? *reinterpret_cast<const float4 *>(a.Yprev + row * a.ldyp + o0)
                                                                : make_float4(0.f, 0.f, 0.f, 0.f);
                    yo4[u][q] = (in && a.accumulate) ? *reinterpret_cast<const float4 *>(a.Y + row * a.ldy + o0)
                                                     : make_float4(0.f, 0.f, 0.f, 0.f);
                }
        }
#pragma unroll
        for (int u = 0; u < HT; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int t = half * HT + u;
                const int o0 = a.out0 + t * 32 + 8 * q + 4 * hi;
                if (o0 >= a.n_out) continue;
                float v[4];
                const float yp[4] = {yp4[u][q].x, yp4[u][q].y, yp4[u][q].z, yp4[u][q].w};
                const float yo[4] = {yo4[u][q].x, yo4[u][q].y, yo4[u][q].z, yo4[u][q].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float z = acc[t][4 * q + j];
                    if (!BWD) {
                        z = dense_act(z, a.act, selv);
                    } else {
                        z += yo[j];
                        if (act_has_dout(a.act)) z *= dense_dact_from_out(yp[j], a.act);
                    }
                    v[j] = live ? z : 0.f;
                }
                float *yptr = a.Y + row * a.ldy + o0;
                if (o0 + 3 < a.n_out) *reinterpret_cast<float4 *>(yptr) = make_float4(v[0], v[1], v[2], v[3]);
                else
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (o0 + j < a.n_out) yptr[j] = v[j];
            }
    }
}

template <int NT, bool BWD>
__global__ __launch_bounds__(256, 2) void dense_kernel(DenseArgs a) {
    // wide layers: the two 128-output halves of one 128-sample block are neighbouring workgroups, so the second
    // one finds the activation rows in L2 instead of re-reading them from HBM a whole pass later
    const int grp = a.groups > 1 ? (int)(blockIdx.x % a.groups) : 0;
    a.out0 += grp * 128;
    extern __shared__ __attribute__((aligned(16))) float lds[];        // 2 x [NT*32][33]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hi = lane >> 5, sl = lane & 31;
    const int64_t n_blk = (a.n + 31) >> 5;
    const int64_t blk = (int64_t)(blockIdx.x / (a.groups > 1 ? a.groups : 1)) * 4 + wave;
    const bool active = blk < n_blk;
    const int64_t row = blk * 32 + sl;
    const int n_chunks = (a.red + KC - 1) / KC;
    constexpr int BUF = NT * 32 * 33;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int o = a.out0 + t * 32 + rowc(g) + 4 * hi;
            float bv = 0.f;
            if (!BWD && a.bias) { bv = a.bias[min(o, a.w_rows - 1)]; bv = o < a.w_rows ? bv : 0.f; }   // branch-free: loads batch
            acc[t][g] = bv;
        }
    {
        float pre[NT * 4];
        load_chunk<NT, BWD>(a, 0, pre);
        store_chunk<NT, BWD>(lds, pre);
    }
    __syncthreads();
    // both operands of chunk c+1 are fetched while chunk c is on the matrix cores (one wave per SIMD at
    // NT = 8: nothing else would hide the latency)
    float4 xv[8];
    auto load_x = [&](int c, float4 (&dst)[8]) {
        if (active) {
            const float4 *xp = reinterpret_cast<const float4 *>(a.X + row * a.ldx + c * KC);
#pragma unroll
            for (int j = 0; j < 8; ++j) dst[j] = xp[j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) dst[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    load_x(0, xv);
    for (int c = 0; c < n_chunks; ++c) {
        float pre[NT * 4];
        float4 xn[8];
        const bool more = c + 1 < n_chunks;
        if (more) { load_chunk<NT, BWD>(a, c + 1, pre); load_x(c + 1, xn); }
        const float *buf = lds + (c & 1) * BUF;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float4 q = xv[s >> 1];
            const float b = (s & 1) ? (hi ? q.w : q.z) : (hi ? q.y : q.x);       // X[row][k0 + 2 s + hi]
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = MFMA(buf[(t * 32 + sl) * 33 + 2 * s + hi], b, acc[t]);
        }
        if (more) {
            store_chunk<NT, BWD>(lds + ((c + 1) & 1) * BUF, pre);
#pragma unroll
            for (int j = 0; j < 8; ++j) xv[j] = xn[j];
        }
        __syncthreads();
    }
    dense_epilogue<NT, BWD>(a, acc, active, row, hi);
}

// ---- the same layer on the bf16 matrix cores ------------------------------------------------------------------
// MODE 6: every fp32 operand split into three bf16 pieces, six v_mfma_f32_32x32x16_bf16 per 16-wide k-step
// reproduce the fp32 product to fp32 round-off (see ren_mlp_x.hip) -- 192 matrix-pipe cycles per k-step against
// 512 for eight f32 MFMAs (both add to the VALU's cycles: DESIGN 3.2).  MODE 1: plain bf16 operands.
// The weight chunk W[:, k0:k0+32] is split ONCE per workgroup while it is staged in LDS ([piece][row][40] bf16:
// one ds_read_b128 per A operand, conflict-free for 16-byte reads at an 80-byte row stride); the 8 k-values a lane
// feeds per k-step are contiguous in the row-major activations (two float4 loads) and are split in registers.
constexpr int XST = 40;                                  // LDS row stride in bf16 (32 k + 8 pad)

template <int NT, bool BWD, int NP, int NTH = 256>
__device__ __forceinline__ void load_chunk_x(const DenseArgs &a, int c, float (&pre)[NT * 1024 / NTH], int j0 = 0,
                                             int j1 = NT * 512 / NTH) {
    const int k0 = c * KC;
    // 32-bit element offsets from the (wave-uniform) weight pointer: one address register per load instead of a
    // 64-bit pair -- sixteen hoisted pairs were what spilled inside the backward-data loop
#pragma unroll
    for (int j = 0; j < NT * 512 / NTH; ++j) {           // thread -> (row, column pair) of the [NT*32][32] chunk
        if (j < j0 || j >= j1) continue;
        const int e = threadIdx.x + NTH * j;
        float v0 = 0.f, v1 = 0.f;
        if (!BWD) {
            const int row = a.out0 + (e >> 4), col = k0 + 2 * (e & 15);
            if (row < a.w_rows) {
                const uint32_t o = (uint32_t)(row * a.w_cols + col);
                if (col < a.w_cols) v0 = a.W[o];
                if (col + 1 < a.w_cols) v1 = a.W[o + 1];
            }
        } else {                                         // W^T: out row = W column, k = W row; coalesced along W columns
            const int row = a.out0 + e % (NT * 32), col = k0 + 2 * (e / (NT * 32));
            if (row < a.w_cols) {
                const uint32_t o = (uint32_t)(col * a.w_cols + row);
                if (col < a.w_rows) v0 = a.W[o];
                if (col + 1 < a.w_rows) v1 = a.W[o + (uint32_t)a.w_cols];
            }
        }
        pre[2 * j] = v0; pre[2 * j + 1] = v1;
    }
}

template <int NT, bool BWD, int NP, int NTH = 256>
__device__ __forceinline__ void store_chunk_x(__bf16 *buf, const float (&pre)[NT * 1024 / NTH], int j0 = 0,
                                              int j1 = NT * 512 / NTH) {
#pragma unroll
    for (int j = 0; j < NT * 512 / NTH; ++j) {
        if (j < j0 || j >= j1) continue;
        const int e = threadIdx.x + NTH * j;
        const int row = BWD ? e % (NT * 32) : e >> 4, cp = BWD ? e / (NT * 32) : e & 15;
        __bf16 s0[3], s1[3];
        split<NP>(pre[2 * j], s0);
        split<NP>(pre[2 * j + 1], s1);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
            bf16x2 v; v[0] = s0[p]; v[1] = s1[p];
            *reinterpret_cast<bf16x2 *>(buf + ((p * NT * 32 + row) * XST + 2 * cp)) = v;
        }
    }
}

// NB = 32-sample blocks per wave: every weight fragment read from LDS feeds NB MFMAs (one per block).  With NB = 1 the
// kernel is LDS-bound (one 1 KB ds_read_b128 per 32-cycle MFMA and SIMD = the whole 128 B/clk of the CU in bf16 mode)
// and the weight staging (global load + split + LDS store by every workgroup) is amortised over 128 samples only.
template <int NT, bool BWD, int MODE, int NB>
__global__ __launch_bounds__(256, 2) void dense_x_kernel(DenseArgs a) {
    // wide layers: the two 128-output halves of one sample block are neighbouring workgroups, so the second
    // one finds the activation rows in L2 instead of re-reading them from HBM a whole pass later
    const int grp = a.groups > 1 ? (int)(blockIdx.x % a.groups) : 0;
    a.out0 += grp * 128;
    using PR = Pairs<MODE>;
    constexpr int NP = PR::NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_x[];   // 2 x [NP][NT*32][XST] bf16
    __bf16 *lds = reinterpret_cast<__bf16 *>(smem_x);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hi = lane >> 5, sl = lane & 31;
    const int64_t n_blk = (a.n + 31) >> 5;
    const int64_t blk0 = ((int64_t)(blockIdx.x / (a.groups > 1 ? a.groups : 1)) * 4 + wave) * NB;
    bool active[NB];
    int64_t row[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) { active[u] = blk0 + u < n_blk; row[u] = (blk0 + u) * 32 + sl; }
    const int n_chunks = (a.red + KC - 1) / KC;
    constexpr int BUF = NP * NT * 32 * XST;

    f32x16 acc[NB][NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int o = a.out0 + t * 32 + rowc(g) + 4 * hi;
            float bv = 0.f;
            if (!BWD && a.bias) { bv = a.bias[min(o, a.w_rows - 1)]; bv = o < a.w_rows ? bv : 0.f; }
#pragma unroll
            for (int u = 0; u < NB; ++u) acc[u][t][g] = bv;
        }
    {
        float pre[NT * 4];
        load_chunk_x<NT, BWD, NP>(a, 0, pre);
        store_chunk_x<NT, BWD, NP>(lds, pre);
    }
    __syncthreads();
    // k-steps of 16 are software-pipelined across chunk boundaries: the 8 k-values a lane feeds in step q = 2 c + s are
    // X[row][16 q + 8 hi ..+8] (two float4), fetched one step ahead -- 16 registers in flight per block instead of a
    // whole chunk for each of this and the next one
    const int n_steps = 2 * n_chunks;
    float4 xc[NB][2];
    auto load_x = [&](int q, float4 (&dst)[NB][2]) {
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            if (active[u]) {
                const float4 *xp = reinterpret_cast<const float4 *>(a.X + row[u] * a.ldx + 16 * q + 8 * hi);
                dst[u][0] = xp[0]; dst[u][1] = xp[1];
            } else {
                dst[u][0] = make_float4(0.f, 0.f, 0.f, 0.f); dst[u][1] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    load_x(0, xc);
    for (int c = 0; c < n_chunks; ++c) {
        float pre[NT * 4];
        const bool more = c + 1 < n_chunks;
        const __bf16 *buf = lds + (c & 1) * BUF;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (more && s == 0) load_chunk_x<NT, BWD, NP>(a, c + 1, pre);
            float4 xn[NB][2];
            const int q = 2 * c + s;
            if (q + 1 < n_steps) load_x(q + 1, xn);
            bf16x8 b[NB][3];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const float xs[8] = {xc[u][0].x, xc[u][0].y, xc[u][0].z, xc[u][0].w, xc[u][1].x, xc[u][1].y, xc[u][1].z, xc[u][1].w};
                split8<NP>(xs, b[u]);
            }
#pragma unroll
            for (int k = 0; k < PR::N; ++k)
#pragma unroll
                for (int t = 0; t < NT; ++t) {           // consecutive MFMAs hit different accumulators
                    const bf16x8 w = *reinterpret_cast<const bf16x8 *>(buf + ((PR::W[k] * NT * 32 + t * 32 + sl) * XST + 16 * s + 8 * hi));
#pragma unroll
                    for (int u = 0; u < NB; ++u) acc[u][t] = MFMAB(w, b[u][PR::A[k]], acc[u][t]);
                }
            if (q + 1 < n_steps) {
#pragma unroll
                for (int u = 0; u < NB; ++u) { xc[u][0] = xn[u][0]; xc[u][1] = xn[u][1]; }
            }
            if (more && s == 1) store_chunk_x<NT, BWD, NP>(lds + ((c + 1) & 1) * BUF, pre);
        }
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < NB; ++u) dense_epilogue<NT, BWD>(a, acc[u], active[u], row[u], hi);
}

// ---- wide layers (128 / 256 outputs): activation tiles staged through LDS ---------------------------------------------
// One workgroup of 8 waves per CU; a wave owns NB blocks of 32 samples and NT x NB = 8 accumulator tiles (NT = 8: all
// 256 outputs of a block, so the activation rows are read ONCE and every split of an activation operand feeds 48 MFMAs;
// NT = 4: two blocks share each weight fragment).  What the row-major "lane = sample" loads of dense_x_kernel cost
// (PMC at 1 M x 256 x 256, mode 6: MFMA 32 %, VALU 34 %, TA 54 % busy, waves waiting 35-40 % of their cycles): a wave
// load touched 32 rows x 2 x 16 B, half of every 128-B line, one k-step ahead of its use.  Here a load instruction
// covers 8 rows x 128 B (whole lines), a chunk ahead, and the tile is turned into the MFMA B layout through a
// wave-private LDS tile (row stride 36 words: conflict-free for 16-byte accesses both ways); the epilogue goes back
// through the same tile so that the stores (and the backward's saved-activation loads) are whole lines as well.
constexpr int TST = 36;

template <int NT, int NB, bool BWD, int MODE, int NW>
__global__ __launch_bounds__(64 * NW, 1) void dense_t_kernel(DenseArgs a) {
    using PR = Pairs<MODE>;
    constexpr int NP = PR::NT;
    constexpr int WBUF = NP * NT * 32 * XST;             // bf16 elements of one weight chunk buffer
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_t[];
    __bf16 *wl = reinterpret_cast<__bf16 *>(smem_t);     // 2 x [NP][NT*32][XST]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *xt = reinterpret_cast<float *>(smem_t + 2 * WBUF * 2) + wave * NB * 32 * TST;   // this wave's NB tiles
    const int hi = lane >> 5, sl = lane & 31;
    const int rl = lane >> 3, cl = 4 * (lane & 7);       // whole-line access: lane -> (row within 8, column group)
    const int64_t n_blk = (a.n + 31) >> 5;
    const int64_t blk0 = ((int64_t)blockIdx.x * NW + wave) * NB;
    bool active[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) active[u] = blk0 + u < n_blk;
    const int n_chunks = (a.red + KC - 1) / KC;

    f32x16 acc[NB][NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int o = a.out0 + t * 32 + rowc(g) + 4 * hi;
            float bv = 0.f;
            if (!BWD && a.bias) { bv = a.bias[min(o, a.w_rows - 1)]; bv = o < a.w_rows ? bv : 0.f; }
#pragma unroll
            for (int u = 0; u < NB; ++u) acc[u][t][g] = bv;
        }
    float4 xn[NB][4];
    auto load_x = [&](int c) {
#pragma unroll
        for (int u = 0; u < NB; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                xn[u][i] = active[u] ? *reinterpret_cast<const float4 *>(a.X + ((blk0 + u) * 32 + 8 * i + rl) * a.ldx + c * KC + cl)
                                     : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto store_x = [&]() {
#pragma unroll
        for (int u = 0; u < NB; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<float4 *>(xt + (u * 32 + 8 * i + rl) * TST + cl) = xn[u][i];
    };
    {
        float pre[NT * 16 / NW];
        load_chunk_x<NT, BWD, NP, 64 * NW>(a, 0, pre);
        load_x(0);
        store_chunk_x<NT, BWD, NP, 64 * NW>(wl, pre);
        store_x();
    }
    __syncthreads();
    for (int c = 0; c < n_chunks; ++c) {
        float pre[NT * 16 / NW];
        const bool more = c + 1 < n_chunks;
        if (more) load_x(c + 1);
        const __bf16 *buf = wl + (c & 1) * WBUF;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (more && s == 0) load_chunk_x<NT, BWD, NP, 64 * NW>(a, c + 1, pre);
            bf16x8 b[NB][3];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const float4 x0 = *reinterpret_cast<const float4 *>(xt + (u * 32 + sl) * TST + 16 * s + 8 * hi);
                const float4 x1 = *reinterpret_cast<const float4 *>(xt + (u * 32 + sl) * TST + 16 * s + 8 * hi + 4);
                const float xs[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
                split8<NP>(xs, b[u]);
            }
#pragma unroll
            for (int k = 0; k < PR::N; ++k)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const bf16x8 w = *reinterpret_cast<const bf16x8 *>(buf + ((PR::W[k] * NT * 32 + t * 32 + sl) * XST + 16 * s + 8 * hi));
#pragma unroll
                    for (int u = 0; u < NB; ++u) acc[u][t] = MFMAB(w, b[u][PR::A[k]], acc[u][t]);
                }
            if (more && s == 1) store_chunk_x<NT, BWD, NP, 64 * NW>(wl + ((c + 1) & 1) * WBUF, pre);
        }
        if (more) store_x();                             // wave-private tile: LDS operations of one wave stay in order
        __syncthreads();
    }
    // epilogue through the wave's tile: accumulators -> [sample][neuron] -> whole-line rows
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        if (!active[u]) continue;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float *T = xt + u * 32 * TST;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4 *>(T + sl * TST + 8 * q + 4 * hi) =
                    make_float4(acc[u][t][4 * q], acc[u][t][4 * q + 1], acc[u][t][4 * q + 2], acc[u][t][4 * q + 3]);
            const int o0 = a.out0 + t * 32 + cl;
            if (o0 >= a.n_out) continue;                 // (per lane; the LDS write above is unconditional)
            const bool whole = o0 + 3 < a.n_out;
            float4 yp4[4], yo4[4];
            if (BWD) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int64_t row = (blk0 + u) * 32 + 8 * i + rl;
                    yp4[i] = (whole && act_has_dout(a.act)) ? *reinterpret_cast<const float4 *>(a.Yprev + row * a.ldyp + o0)
                                                                 : make_float4(0.f, 0.f, 0.f, 0.f);
                    yo4[i] = (whole && a.accumulate) ? *reinterpret_cast<const float4 *>(a.Y + row * a.ldy + o0)
                                                     : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int64_t row = (blk0 + u) * 32 + 8 * i + rl;
                const bool live = row < a.n;
                const float4 z4 = *reinterpret_cast<const float4 *>(T + (8 * i + rl) * TST + cl);
                const float zz[4] = {z4.x, z4.y, z4.z, z4.w};
                const float yp[4] = {yp4[i].x, yp4[i].y, yp4[i].z, yp4[i].w};
                const float yo[4] = {yo4[i].x, yo4[i].y, yo4[i].z, yo4[i].w};
                const bool selv = (!BWD && act_uses_sel(a.act) && live) ? a.sel[row] != 0 : false;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float z = zz[j];
                    if (!BWD) {
                        z = dense_act(z, a.act, selv);
                    } else {
                        z += yo[j];
                        if (act_has_dout(a.act)) z *= dense_dact_from_out(yp[j], a.act);
                    }
                    v[j] = live ? z : 0.f;
                }
                float *yptr = a.Y + row * a.ldy + o0;
                if (whole) *reinterpret_cast<float4 *>(yptr) = make_float4(v[0], v[1], v[2], v[3]);
                else
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (o0 + j < a.n_out) yptr[j] = v[j];
            }
        }
    }
}

// ---- weight / bias gradient:  dW[N][K] += dZ^T X,  db[N] += sum dZ ------------------------------------
// One workgroup of 8 waves per sample split.  Per 32-sample stage the rows of dZ (<= 256 columns) and X
// (<= 320 columns) are loaded ONCE, coalesced, into LDS and every wave takes one 32-row tile of dW
// (wave = N tile; a second N-tile round covers N = 256 with 8 waves each owning one tile, smaller N leaves
// waves idle) against up to 8 K tiles: A = dZ^T (lane = neuron, k = sample), B = X (k = sample, lane =
// input feature), both read conflict-free from the row-major LDS tiles.  HBM traffic is therefore one
// pass over dZ and X per layer (per K group of 256 columns).  Partial sums go to per-split slabs.
struct DwArgs {
    const float *dZ; int ldz; int N;
    const float *X; int ldx; int K;
    int64_t n;
    float *slab_w, *slab_b;                              // [n_splits][N][K], [n_splits][N]
};

constexpr int DW_LDZ = 256, DW_LDX = 256;                // LDS tile widths (one K group = 256 columns)

__global__ __launch_bounds__(512, 1) void dense_dw_kernel(DwArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];       // 2 x ([32][256] dZ + [32][256] X)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hi = lane >> 5, sl = lane & 31;
    const int n_splits = gridDim.x;
    const int64_t n_blk = (a.n + 31) >> 5;
    const int k_tiles = (a.K + 31) >> 5, n_tiles = (a.N + 31) >> 5;
    const bool has_tile = wave < n_tiles;
    float *sw = a.slab_w + (int64_t)blockIdx.x * a.N * a.K, *sb = a.slab_b + (int64_t)blockIdx.x * a.N;
    constexpr int TILE = 32 * 256;                                    // floats per operand tile
    for (int kg = 0; kg * 8 < k_tiles; ++kg) {
        const int kt = min(8, k_tiles - kg * 8);                      // K tiles of this group
        f32x16 acc[8];
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[t][g] = 0.f;
        float accb = 0.f;
        // cooperative stage loader: 32 rows x 256 columns of each operand = 2 x 2048 float4, 512 threads
        float4 pz[4], px[4];
        auto fetch = [&](int64_t blk) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int e = threadIdx.x + 512 * j;                  // float4 index in the [32][64] float4 tile
                const int r = e >> 6, c4 = (e & 63) * 4;
                const int64_t row = blk * 32 + r;
                pz[j] = (row < a.n && c4 < a.ldz && c4 < ((a.N + 3) & ~3))
                            ? *reinterpret_cast<const float4 *>(a.dZ + row * a.ldz + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
                const int cx = kg * 256 + c4;
                px[j] = (row < a.n && cx < a.ldx && c4 < kt * 32)
                            ? *reinterpret_cast<const float4 *>(a.X + row * a.ldx + cx) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        auto stash = [&](float *buf) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int e = threadIdx.x + 512 * j;
                reinterpret_cast<float4 *>(buf)[e] = pz[j];
                reinterpret_cast<float4 *>(buf + TILE)[e] = px[j];
            }
        };
        int64_t blk = blockIdx.x;
        int cur = 0;
        if (blk < n_blk) { fetch(blk); stash(lds); }
        __syncthreads();
        for (; blk < n_blk; blk += n_splits) {
            const int64_t nxt = blk + n_splits;
            if (nxt < n_blk) fetch(nxt);
            const float *tz = lds + cur * 2 * TILE, *tx = tz + TILE;
            if (has_tile) {
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const float az = tz[(2 * s + hi) * DW_LDZ + wave * 32 + sl];
                    accb += az;
#pragma unroll
                    for (int t = 0; t < 8; ++t)
                        if (t < kt) acc[t] = MFMA(az, tx[(2 * s + hi) * DW_LDX + t * 32 + sl], acc[t]);   // wave-uniform
                }
            }
            if (nxt < n_blk) stash(lds + (1 - cur) * 2 * TILE);
            cur = 1 - cur;
            __syncthreads();
        }
        if (has_tile) {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int k = (kg * 8 + t) * 32 + sl;
                if (t >= kt || k >= a.K) continue;
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const int o = wave * 32 + rowc(g) + 4 * hi;
                    if (o < a.N) sw[(int64_t)o * a.K + k] = acc[t][g];
                }
            }
            if (kg == 0) {
                const float bsum = accb + __shfl_xor(accb, 32, 64);
                const int neuron = wave * 32 + sl;
                if (hi == 0 && neuron < a.N) sb[neuron] = bsum;
            }
        }
        __syncthreads();
    }
}

// ---- weight gradient on the bf16 matrix cores -------------------------------------------------------------------
// dW = dZ^T X with the SAMPLES as the reduction: both MFMA operands need 8 consecutive samples per lane, i.e. the
// transposes of the row-major [sample][feature] tiles.  The stage loader does the transposition while it splits:
// thread (sample s = tid & 31, feature group tid >> 5) loads 16 consecutive features of its sample row and writes
// them as bf16 pieces to [piece][feature][sample] (lanes of a wave vary s: consecutive 2-byte addresses, conflict
// free), so both operands are read back as one ds_read_b128.  fp32 modes: three pieces / six MFMAs per product pair
// (fp32 round-off per product, as everywhere else), i.e. 96 bf16 MFMAs (3 072 matrix-pipe cycles) per wave and 32-sample
// stage against 128 f32 MFMAs (8 192); bf16 mode: one piece, one MFMA.  The bias gradient is an
// fp32 side sum of the loader's own values.
constexpr int DWX_ST = 40;                               // [feature][32 samples + 8 pad] bf16: 80-byte rows
// DWX_NP pieces per operand: 3 (six product terms: fp32 round-off, the fp32 modes) or 1 (plain bf16 operands, `mlp_bf16`)
template <int DWX_NP>
__global__ __launch_bounds__(512, 1) void dense_dw_x_kernel(DwArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_dw[];    // ZT | XT: [NP][256][DWX_ST] bf16 each
    __bf16 *ZT = reinterpret_cast<__bf16 *>(smem_dw), *XT = ZT + DWX_NP * 256 * DWX_ST;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hi = lane >> 5, sl = lane & 31;
    const int ls = threadIdx.x & 31, fg = threadIdx.x >> 5;                   // loader: sample row, 16-feature group
    const int n_splits = gridDim.x;
    const int64_t n_blk = (a.n + 31) >> 5;
    const int k_tiles = (a.K + 31) >> 5, n_tiles = (a.N + 31) >> 5;
    const bool has_tile = wave < n_tiles;
    float *sw = a.slab_w + (int64_t)blockIdx.x * a.N * a.K, *sb = a.slab_b + (int64_t)blockIdx.x * a.N;
    for (int kg = 0; kg * 8 < k_tiles; ++kg) {
        const int kt = min(8, k_tiles - kg * 8);
        f32x16 acc[8];
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[t][g] = 0.f;
        float bsum[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) bsum[q] = 0.f;
        float4 pz[4], px[4];
        auto fetch = [&](int64_t blk) {
            const int64_t row = blk * 32 + ls;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = fg * 16 + 4 * j;
                pz[j] = (row < a.n && c < a.ldz && c < ((a.N + 3) & ~3))
                            ? *reinterpret_cast<const float4 *>(a.dZ + row * a.ldz + c) : make_float4(0.f, 0.f, 0.f, 0.f);
                const int cx = kg * 256 + c;
                px[j] = (row < a.n && cx < a.ldx && c < kt * 32)
                            ? *reinterpret_cast<const float4 *>(a.X + row * a.ldx + cx) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        auto stash = [&]() {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float z[4] = {pz[j].x, pz[j].y, pz[j].z, pz[j].w}, x[4] = {px[j].x, px[j].y, px[j].z, px[j].w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int f = fg * 16 + 4 * j + q;
                    __bf16 tz[3], tx[3];
                    split<DWX_NP>(z[q], tz);
                    split<DWX_NP>(x[q], tx);
#pragma unroll
                    for (int p = 0; p < DWX_NP; ++p) {
                        ZT[(p * 256 + f) * DWX_ST + ls] = tz[p];
                        XT[(p * 256 + f) * DWX_ST + ls] = tx[p];
                    }
                    bsum[4 * j + q] += z[q];
                }
            }
        };
        int64_t blk = blockIdx.x;
        if (blk < n_blk) fetch(blk);
        for (; blk < n_blk; blk += n_splits) {
            __syncthreads();                                       // previous stage's readers are done
            stash();
            __syncthreads();
            const int64_t nxt = blk + n_splits;
            if (nxt < n_blk) fetch(nxt);
            if (has_tile) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    bf16x8 az[DWX_NP];
#pragma unroll
                    for (int p = 0; p < DWX_NP; ++p)
                        az[p] = *reinterpret_cast<const bf16x8 *>(ZT + (p * 256 + wave * 32 + sl) * DWX_ST + 16 * ks + 8 * hi);
#pragma unroll
                    for (int t = 0; t < 8; ++t)
                        if (t < kt) {                              // wave-uniform
                            bf16x8 bx[DWX_NP];
#pragma unroll
                            for (int p = 0; p < DWX_NP; ++p)
                                bx[p] = *reinterpret_cast<const bf16x8 *>(XT + (p * 256 + t * 32 + sl) * DWX_ST + 16 * ks + 8 * hi);
                            if (DWX_NP == 3) {
                                acc[t] = MFMAB(az[2], bx[0], acc[t]);
                                acc[t] = MFMAB(az[0], bx[2], acc[t]);
                                acc[t] = MFMAB(az[1], bx[1], acc[t]);
                            }
                            if (DWX_NP >= 2) {
                                acc[t] = MFMAB(az[1], bx[0], acc[t]);
                                acc[t] = MFMAB(az[0], bx[1], acc[t]);
                            }
                            acc[t] = MFMAB(az[0], bx[0], acc[t]);
                        }
                }
            }
        }
        if (has_tile) {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int k = (kg * 8 + t) * 32 + sl;
                if (t >= kt || k >= a.K) continue;
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const int o = wave * 32 + rowc(g) + 4 * hi;
                    if (o < a.N) sw[(int64_t)o * a.K + k] = acc[t][g];
                }
            }
        }
        if (kg == 0) {                                             // bias: sum over the 32 sample lanes of the loader
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                float v = bsum[q];
#pragma unroll
                for (int off = 1; off < 32; off <<= 1) v += __shfl_xor(v, off, 64);
                const int neuron = fg * 16 + q;
                if (ls == 0 && neuron < a.N) sb[neuron] = v;
            }
        }
        __syncthreads();
    }
}

// ---- frequency encodings (SinusoidalEncoder, mlp.py:208-243) ---------------------------------------------------
struct EncArgs {
    SampleSrc src;
    ren_scene_dev sc;
    int64_t n;
    float *enc; int ld_enc;                              // [n_pad][>= 64]: 63 position features, rest zero
    float *cat; int ld_cat, cat_col;                     // optional second copy at columns cat_col.. of the skip buffer
    float *view; int ld_view, view_col;                  // optional 27 direction features (padded to 32)
    uint8_t *sel;
};

template <int deg>
__device__ __forceinline__ void sin_enc(const float *x, float *out) {            // [x, sin(2^k x), sin(2^k x + pi/2)]
    out[0] = x[0]; out[1] = x[1]; out[2] = x[2];
#pragma unroll
    for (int k = 0; k < deg; ++k) {
        const float sc = (float)(1 << k);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float xb = x[j] * sc;
            out[3 + 3 * k + j] = sinf(xb);
            out[3 + 3 * deg + 3 * k + j] = sinf(xb + 1.5707963267948966f);
        }
    }
}

__global__ __launch_bounds__(256) void freq_encode_kernel(EncArgs a) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) {                                      // rows up to the next multiple of 32: zeros (finite inputs for the
        if (i < ((a.n + 31) & ~(int64_t)31)) {           // padded rows of the last 32-sample block, csrc/ren_vfield.hip)
            const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int j = 0; j < 16; ++j) reinterpret_cast<float4 *>(a.enc + i * a.ld_enc)[j] = z4;
            if (a.cat) for (int j = 0; j < 16; ++j) reinterpret_cast<float4 *>(a.cat + i * a.ld_cat + a.cat_col)[j] = z4;
            if (a.view) for (int j = 0; j < 8; ++j) reinterpret_cast<float4 *>(a.view + i * a.ld_view + a.view_col)[j] = z4;
        }
        return;
    }
    float x, y, z, dx = 0.f, dy = 0.f, dz = 1.f;
    if (a.src.ray_indices) {
        int ray;
        ren_sample_pos(a.src.rays_o, a.src.rays_d, a.src.ray_indices, a.src.t_starts, a.src.t_ends, i, x, y, z, ray);
        const float *d = a.src.rays_d + 3 * (int64_t)ray;
        dx = d[0]; dy = d[1]; dz = d[2];
    } else {
        x = a.src.x_world[3 * i]; y = a.src.x_world[3 * i + 1]; z = a.src.x_world[3 * i + 2];
        if (a.src.dirs) { dx = a.src.dirs[3 * i]; dy = a.src.dirs[3 * i + 1]; dz = a.src.dirs[3 * i + 2]; }
    }
    float u[3];
    ren_contract(a.sc, x, y, z, u[0], u[1], u[2]);
    a.sel[i] = u[0] > 0.f && u[0] < 1.f && u[1] > 0.f && u[1] < 1.f && u[2] > 0.f && u[2] < 1.f;    // mlp.py:333
    const float TWO_PI = 6.283185307179586f;
    const float p[3] = {TWO_PI * (u[0] - 0.5f), TWO_PI * (u[1] - 0.5f), TWO_PI * (u[2] - 0.5f)};  // mlp.py:335
    float e[64];
    sin_enc<10>(p, e);
    e[63] = 0.f;
    float4 *o1 = reinterpret_cast<float4 *>(a.enc + i * a.ld_enc);
#pragma unroll
    for (int j = 0; j < 16; ++j) o1[j] = make_float4(e[4 * j], e[4 * j + 1], e[4 * j + 2], e[4 * j + 3]);
    if (a.cat) {
        float4 *o2 = reinterpret_cast<float4 *>(a.cat + i * a.ld_cat + a.cat_col);
#pragma unroll
        for (int j = 0; j < 16; ++j) o2[j] = make_float4(e[4 * j], e[4 * j + 1], e[4 * j + 2], e[4 * j + 3]);
    }
    if (a.view) {
        const float PI = 3.141592653589793f;
        const float c[3] = {dx * PI, dy * PI, dz * PI};                                              // mlp.py:352
        float v[32];
        sin_enc<4>(c, v);
#pragma unroll
        for (int j = 27; j < 32; ++j) v[j] = 0.f;
        float4 *o3 = reinterpret_cast<float4 *>(a.view + i * a.ld_view + a.view_col);
#pragma unroll
        for (int j = 0; j < 8; ++j) o3[j] = make_float4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
    }
}

// first / second derivative of the output activations through their OUTPUTS.  dn / rd: density / radiance kind of the
// activation code (ren_mlp_common.h act_kinds; 0 = shifted_trunc_exp / softplus, the shipped configs).
// density: sigma = sel phi(z).  trunc_exp: phi' = min(sigma, e^15) (ngp.py:45-65), phi'' = sigma below the clamp, 0 above;
// softplus kinds: phi' = 1 - exp(-sigma) (0 where sel = 0, as it must be), phi'' = phi' (1 - phi').
__device__ __forceinline__ void head_density_d(float sg, int dn, float &p1, float &p2) {
    if (dn == 0) {
        const float E15 = 3269017.3724721107f;
        p1 = fminf(sg, E15); p2 = sg < E15 ? sg : 0.f;
    } else {
        p1 = dsoftplus_from_out(sg, 1.f); p2 = p1 * (1.f - p1);
    }
}
__device__ __forceinline__ void head_radiance_d(float y, int rd, float &r1, float &r2) {
    if (rd == 1) { r1 = y * (1.f - y); r2 = r1 * (1.f - 2.f * y); }          // sigmoid
    else { r1 = dsoftplus_from_out(y, 1.f); r2 = r1 * (1.f - r1); }
}

// ---- output activations, backward:  dz_rgb = g_rgb act'(rgb),  dz_sigma = g_sigma phi'(sigma) ------------------------
__global__ __launch_bounds__(256) void heads_bwd_kernel(const float *__restrict__ g_rgb, const float *__restrict__ rgb,
                                                        const float *__restrict__ g_sigma,
                                                        const float *__restrict__ sigma, int64_t n, int64_t n_pad, int C,
                                                        int dn, int rd, float *__restrict__ dz_rgb, float *__restrict__ dz_sigma) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pad) return;
    float4 *zr = reinterpret_cast<float4 *>(dz_rgb + i * 32), *zs = reinterpret_cast<float4 *>(dz_sigma + i * 32);
    float r[4] = {0.f, 0.f, 0.f, 0.f}, s0 = 0.f;
    if (i < n) {
        for (int c = 0; c < C; ++c) {
            float r1, r2;
            head_radiance_d(rgb[i * C + c], rd, r1, r2);
            r[c] = g_rgb[i * C + c] * r1;
        }
        float p1, p2;
        head_density_d(sigma[i], dn, p1, p2);
        s0 = g_sigma[i] * p1;
    }
    zr[0] = make_float4(r[0], r[1], r[2], r[3]);
    zs[0] = make_float4(s0, 0.f, 0.f, 0.f);
    for (int j = 1; j < 8; ++j) { zr[j] = make_float4(0.f, 0.f, 0.f, 0.f); zs[j] = make_float4(0.f, 0.f, 0.f, 0.f); }
}

// ---- activation algebra of the tangent stream (log-intensity-gradient loss, arch mlp) ---------------------------
// forward:  yd = s zd,  s = softplus_beta'(z) = 1 - exp(-beta y) recovered from the output y
// backward: gz = gy s + gyd zd s',  gzd = gyd s,  s' = beta s (1 - s)
// Row-major [rows][ld] buffers (the value outputs may be column ranges of a wider buffer), float4 per thread.
__global__ __launch_bounds__(256) void act_jvp_fwd_kernel(const float *__restrict__ Y, int ldy, const float *__restrict__ Zd,
                                                          int ldz, float beta, float *__restrict__ Yd, int ldyd,
                                                          int64_t rows, int width4) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * width4) return;
    const int64_t r = e / width4;
    const int c = (int)(e - r * width4) * 4;
    const float4 y = *reinterpret_cast<const float4 *>(Y + r * ldy + c), zd = *reinterpret_cast<const float4 *>(Zd + r * ldz + c);
    float4 o;
    o.x = zd.x * dact_beta(y.x, beta); o.y = zd.y * dact_beta(y.y, beta);
    o.z = zd.z * dact_beta(y.z, beta); o.w = zd.w * dact_beta(y.w, beta);
    *reinterpret_cast<float4 *>(Yd + r * ldyd + c) = o;
}

__global__ __launch_bounds__(256) void act_jvp_bwd_kernel(const float *__restrict__ Gy, const float *__restrict__ Gyd,
                                                          const float *__restrict__ Y, int ldy, const float *__restrict__ Zd,
                                                          float beta, float *__restrict__ Gz, float *__restrict__ Gzd,
                                                          int64_t rows, int width4) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * width4) return;
    const int64_t r = e / width4;
    const int c = (int)(e - r * width4) * 4, w = width4 * 4;
    const float4 y4 = *reinterpret_cast<const float4 *>(Y + r * ldy + c);
    const float4 gy4 = *reinterpret_cast<const float4 *>(Gy + r * w + c), gd4 = *reinterpret_cast<const float4 *>(Gyd + r * w + c);
    const float4 zd4 = *reinterpret_cast<const float4 *>(Zd + r * w + c);
    const float y[4] = {y4.x, y4.y, y4.z, y4.w}, gy[4] = {gy4.x, gy4.y, gy4.z, gy4.w}, gd[4] = {gd4.x, gd4.y, gd4.z, gd4.w},
                zd[4] = {zd4.x, zd4.y, zd4.z, zd4.w};
    float gz[4], gzd[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float sj = dact_beta(y[j], beta);
        gz[j] = gy[j] * sj + gd[j] * zd[j] * (beta * sj * (1.f - sj));
        gzd[j] = gd[j] * sj;
    }
    *reinterpret_cast<float4 *>(Gz + r * w + c) = make_float4(gz[0], gz[1], gz[2], gz[3]);
    *reinterpret_cast<float4 *>(Gzd + r * w + c) = make_float4(gzd[0], gzd[1], gzd[2], gzd[3]);
}

// second-order forward:  yd = s zd,  ydd = s' zd^2 + s zdd  (y = softplus_beta(z); trainable tau under the
// log-intensity-gradient loss, arch mlp: value, d/dt and d2/dt2 streams through the same dense layers)
__global__ __launch_bounds__(256) void act_jvp2_fwd_kernel(const float *__restrict__ Y, int ldy, const float *__restrict__ Zd,
                                                           const float *__restrict__ Zdd, int ldz, float beta,
                                                           float *__restrict__ Yd, int ldyd, float *__restrict__ Ydd, int ldydd,
                                                           int64_t rows, int width4) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * width4) return;
    const int64_t r = e / width4;
    const int c = (int)(e - r * width4) * 4;
    const float4 y4 = *reinterpret_cast<const float4 *>(Y + r * ldy + c);
    const float4 zd4 = *reinterpret_cast<const float4 *>(Zd + r * ldz + c), ze4 = *reinterpret_cast<const float4 *>(Zdd + r * ldz + c);
    const float y[4] = {y4.x, y4.y, y4.z, y4.w}, zd[4] = {zd4.x, zd4.y, zd4.z, zd4.w}, ze[4] = {ze4.x, ze4.y, ze4.z, ze4.w};
    float od[4], oe[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float sj = dact_beta(y[j], beta);
        od[j] = sj * zd[j];
        oe[j] = beta * sj * (1.f - sj) * zd[j] * zd[j] + sj * ze[j];
    }
    *reinterpret_cast<float4 *>(Yd + r * ldyd + c) = make_float4(od[0], od[1], od[2], od[3]);
    *reinterpret_cast<float4 *>(Ydd + r * ldydd + c) = make_float4(oe[0], oe[1], oe[2], oe[3]);
}

// ---- output heads of the tangent streams ------------------------------------------------------------------------------
// rgb = softplus_1(zo):  rgbd = s zod,  rgbdd = s (1 - s) zod^2 + s zodd,  s = 1 - exp(-rgb)
// sigma = sel exp(zs - 1) with the reference's clamped derivative (ngp.py:45-65): phi' = min(sigma, e^15),
// phi'' = sigma below the clamp, 0 above:  sigmad = phi' zsd,  sigmadd = phi'' zsd^2 + phi' zsdd
__global__ __launch_bounds__(256) void heads_jvp_kernel(const float *__restrict__ rgb, const float *__restrict__ sigma,
                                                        const float *__restrict__ zod, const float *__restrict__ zodd,
                                                        const float *__restrict__ zsd, const float *__restrict__ zsdd,
                                                        int64_t n, int C, int dn, int rd, float *__restrict__ rgbd,
                                                        float *__restrict__ rgbdd, float *__restrict__ sigmad,
                                                        float *__restrict__ sigmadd) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float p1, p2;
    head_density_d(sigma[i], dn, p1, p2);
    const float d = zsd[4 * i];
    sigmad[i] = p1 * d;
    if (sigmadd) sigmadd[i] = p2 * d * d + p1 * zsdd[4 * i];
    for (int c = 0; c < C; ++c) {
        float s, s2;
        head_radiance_d(rgb[i * C + c], rd, s, s2);
        const float zd = zod[4 * i + c];
        rgbd[i * C + c] = s * zd;
        if (rgbdd) rgbdd[i * C + c] = s2 * zd * zd + s * zodd[4 * i + c];
    }
}

// reverse pass of (rgb, rgbd, sigma, sigmad): pre-activation gradients of the value and the tangent stream, written as
// zero-padded [n_pad][32] rows (the B operands of the output layers' weight / data gradient launches)
__global__ __launch_bounds__(256) void heads_bwd_jvp_kernel(const float *__restrict__ g_rgb, const float *__restrict__ g_rgbd,
                                                            const float *__restrict__ g_sigma, const float *__restrict__ g_sigmad,
                                                            const float *__restrict__ rgb, const float *__restrict__ sigma,
                                                            const float *__restrict__ zod, const float *__restrict__ zsd,
                                                            int64_t n, int64_t n_pad, int C, int dn, int rd,
                                                            float *__restrict__ dz_rgb, float *__restrict__ dzd_rgb,
                                                            float *__restrict__ dz_sigma, float *__restrict__ dzd_sigma) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pad) return;
    float r[4] = {0.f, 0.f, 0.f, 0.f}, rdv[4] = {0.f, 0.f, 0.f, 0.f}, s0 = 0.f, sd0 = 0.f;
    if (i < n) {
        for (int c = 0; c < C; ++c) {
            float s, s2;
            head_radiance_d(rgb[i * C + c], rd, s, s2);
            r[c] = g_rgb[i * C + c] * s + g_rgbd[i * C + c] * zod[4 * i + c] * s2;
            rdv[c] = g_rgbd[i * C + c] * s;
        }
        float p1, p2;
        head_density_d(sigma[i], dn, p1, p2);
        s0 = g_sigma[i] * p1 + g_sigmad[i] * zsd[4 * i] * p2;
        sd0 = g_sigmad[i] * p1;
    }
    float4 *o[4] = {reinterpret_cast<float4 *>(dz_rgb + i * 32), reinterpret_cast<float4 *>(dzd_rgb + i * 32),
                    reinterpret_cast<float4 *>(dz_sigma + i * 32), reinterpret_cast<float4 *>(dzd_sigma + i * 32)};
    o[0][0] = make_float4(r[0], r[1], r[2], r[3]);
    o[1][0] = make_float4(rdv[0], rdv[1], rdv[2], rdv[3]);
    o[2][0] = make_float4(s0, 0.f, 0.f, 0.f);
    o[3][0] = make_float4(sd0, 0.f, 0.f, 0.f);
    for (int q = 0; q < 4; ++q)
        for (int j = 1; j < 8; ++j) o[q][j] = make_float4(0.f, 0.f, 0.f, 0.f);
}

template <bool BWD>
int launch_dense(DenseArgs a, int tiles, int mode, hipStream_t st) {
    // kernel variant per (matrix-core mode, direction): 1 = dense_x_kernel, one block per wave; 2 = two blocks per wave;
    // 3 = dense_t_kernel (wide layers only).  Defaults from tools/dense_bench.py / bench.py --arch mlp on MI355X.
    // (round 4 re-measurement, tools/dense_bench.py at n = 131 k .. 2 M, 256 x 256: forward variant 1 is 2 x faster than 2 / 3
    // from n = 524 k on in both modes -- 1.11 vs 2.23 ms (mode 1), 1.64 vs 3.07 ms (mode 6) at 1 M -- and equal below; the
    // backward keeps 1 (mode 6) / 3 (mode 1: 0.83 vs 1.06 ms))
    static const int sel6 = [] { const char *e = getenv(BWD ? "REN_DENSE_BWD6" : "REN_DENSE_FWD6"); return e ? atoi(e) : 1; }();
    static const int sel1 = [] { const char *e = getenv(BWD ? "REN_DENSE_BWD1" : "REN_DENSE_FWD1"); return e ? atoi(e) : (BWD ? 3 : 1); }();
    const int sel = mode == 6 ? sel6 : sel1;
    const int nb_env = sel == 1 ? 1 : 2;
    const int use_t = sel == 3;
    a.groups = 1;
    if (use_t && mode != 0 && tiles >= 4 && (a.n + 31) / 32 > 2048) {      // LDS-staged activation tiles, 8 waves per workgroup
        a.out0 = 0;
        const int64_t nblk = (a.n + 31) / 32;
        constexpr int nw = 8;
        const int nbw = 8 / tiles;                        // NT x NB = 8 accumulator tiles per wave
        const dim3 grd((unsigned)((nblk + nw * nbw - 1) / (nw * nbw))), blk(64 * nw);
#define REN_DENSE_T(NT, NB, MODE, NP, NW)                                                                    \
        do {                                                                                                 \
            const size_t lds = 2 * (size_t)NP * NT * 32 * XST * 2 + (size_t)NW * NB * 32 * TST * 4;              \
            (void)hipFuncSetAttribute((const void *)dense_t_kernel<NT, NB, BWD, MODE, NW>,                    \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                 \
            hipLaunchKernelGGL((dense_t_kernel<NT, NB, BWD, MODE, NW>), grd, blk, lds, st, a);                \
        } while (0)
        if (tiles == 8) { if (mode == 6) REN_DENSE_T(8, 1, 6, 3, 8); else REN_DENSE_T(8, 1, 1, 1, 8); }
        else { if (mode == 6) REN_DENSE_T(4, 2, 6, 3, 8); else REN_DENSE_T(4, 2, 1, 1, 8); }
#undef REN_DENSE_T
        REN_CHECK_LAUNCH();
    }
    if (tiles == 8) {                                   // 256 outputs = two interleaved groups of 4 tiles (DenseArgs::out0)
        a.groups = 2;
        a.out0 = 0;
        tiles = 4;
    }
    const int64_t n_blk = (a.n + 31) / 32;
    const int nb = (mode != 0 && nb_env == 2 && n_blk > 4 * 2048) ? 2 : 1;     // small launches: more workgroups instead
    const dim3 grd((unsigned)((n_blk + 4 * nb - 1) / (4 * nb) * a.groups)), blk(256);
#define REN_DENSE_LAUNCH(KERNEL, LDS)                                                                        \
    do {                                                                                                     \
        const size_t lds = (LDS);                                                                            \
        (void)hipFuncSetAttribute((const void *)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(KERNEL, grd, blk, lds, st, a);                                                    \
    } while (0)
#define REN_DENSE_X(NT, MODE, NP)                                                                            \
    do {                                                                                                     \
        if (nb == 2) REN_DENSE_LAUNCH((dense_x_kernel<NT, BWD, MODE, 2>), 2 * (size_t)NP * NT * 32 * XST * 2); \
        else REN_DENSE_LAUNCH((dense_x_kernel<NT, BWD, MODE, 1>), 2 * (size_t)NP * NT * 32 * XST * 2);        \
    } while (0)
#define REN_DENSE_CASE(NT)                                                                                   \
    case NT:                                                                                                 \
        if (mode == 6) REN_DENSE_X(NT, 6, 3);                                                                \
        else if (mode == 1) REN_DENSE_X(NT, 1, 1);                                                           \
        else REN_DENSE_LAUNCH((dense_kernel<NT, BWD>), 2 * (size_t)NT * 32 * 33 * 4);                          \
        break;
    switch (tiles) {
        REN_DENSE_CASE(1) REN_DENSE_CASE(2) REN_DENSE_CASE(4)
        default: return REN_ERR_UNSUPPORTED;
    }
#undef REN_DENSE_CASE
#undef REN_DENSE_X
#undef REN_DENSE_LAUNCH
    REN_CHECK_LAUNCH();
}

// density / radiance kinds of the call's activation code (include/ren_amd.h "activations"): what the output-head kernels of
// arch mlp differentiate.  0 / 0 = the shipped configs.
struct ActKindsHost { int dn, rd; };
inline ActKindsHost act_kinds_host(int code) { return ActKindsHost{(code >> 2) & 3, (code >> 6) & 3}; }

inline int tiles_for(int n_out) { return n_out <= 32 ? 1 : n_out <= 64 ? 2 : n_out <= 128 ? 4 : n_out <= 256 ? 8 : -1; }

}  // namespace

extern "C" int ren_freq_encode(const ren_scene_desc *scene, const float *x_world, const float *dirs,
                               const float *rays_o, const float *rays_d, const int32_t *ray_indices,
                               const float *t_starts, const float *t_ends, int64_t n, float *enc, int32_t ld_enc,
                               float *cat, int32_t ld_cat, int32_t cat_col, float *view, int32_t ld_view,
                               int32_t view_col, uint8_t *selector, void *stream) {
    if (!scene || !enc || !selector || n < 0 || ld_enc < 64 || (ld_enc & 3)) return REN_ERR_BAD_ARG;
    if (!x_world && (!rays_o || !rays_d || !ray_indices || !t_starts || !t_ends)) return REN_ERR_BAD_ARG;
    if (cat && (ld_cat < cat_col + 64 || (ld_cat & 3) || (cat_col & 3))) return REN_ERR_BAD_ARG;
    if (view && (ld_view < view_col + 32 || (ld_view & 3) || (view_col & 3))) return REN_ERR_BAD_ARG;
    if (n == 0) return REN_OK;
    EncArgs a;
    a.src = SampleSrc{x_world, dirs, rays_o, rays_d, x_world ? nullptr : ray_indices, t_starts, t_ends};
    a.sc = ren_make_scene(scene);
    a.n = n; a.enc = enc; a.ld_enc = ld_enc; a.cat = cat; a.ld_cat = ld_cat; a.cat_col = cat_col;
    a.view = view; a.ld_view = ld_view; a.view_col = view_col; a.sel = selector;
    hipLaunchKernelGGL(freq_encode_kernel, dim3(ren_blocks((n + 31) & ~(int64_t)31, 256)), dim3(256), 0, (hipStream_t)stream, a);
    REN_CHECK_LAUNCH();
}

extern "C" int ren_dense_fwd(const float *X, int32_t ldx, const float *W, const float *bias, int32_t n_out,
                             int32_t n_in, int32_t act, const uint8_t *selector, float *Y, int32_t ldy, int64_t n,
                             void *stream) {
    const int mode = (act >> 8) & 0xff;                                      // REN_DENSE_F32 / _BF16X6 / _BF16
    act &= 0xff;
    if (mode != 0 && mode != 1 && mode != 6) return REN_ERR_BAD_ARG;
    if (!X || !W || !Y || n < 0 || n_out < 1 || n_in < 1 || (ldx & 3) || (ldy & 3)) return REN_ERR_BAD_ARG;
    if (ldx < ((n_in + KC - 1) / KC) * KC) return REN_ERR_BAD_ARG;           // X rows are read in whole 32-wide chunks
    if (act < ACT_NONE || act > ACT_LAST || ((act == ACT_TRUNC_EXP_SEL || act >= ACT_SOFTPLUS1_SEL) && !selector)) return REN_ERR_BAD_ARG;
    const int tiles = tiles_for(n_out);
    if (tiles < 0) return REN_ERR_UNSUPPORTED;
    if (n == 0) return REN_OK;
    DenseArgs a = {};
    a.X = X; a.ldx = ldx; a.W = W; a.w_rows = n_out; a.w_cols = n_in; a.red = n_in; a.n_out = n_out; a.bias = bias;
    a.act = act; a.sel = selector; a.Y = Y; a.ldy = ldy; a.n = n;
    return launch_dense<false>(a, tiles, mode, (hipStream_t)stream);
}

extern "C" int ren_dense_bwd_data(const float *dZ, int32_t ldz, const float *W, int32_t n_out, int32_t n_in,
                                  int32_t n_store, int32_t prev_act, const float *Yprev, int32_t ldyp,
                                  int32_t accumulate, float *dX, int32_t ldx, int64_t n, void *stream) {
    const int mode = (prev_act >> 8) & 0xff;
    prev_act &= 0xff;
    if (mode != 0 && mode != 1 && mode != 6) return REN_ERR_BAD_ARG;
    if (!dZ || !W || !dX || n < 0 || n_out < 1 || n_in < 1 || n_store < 1 || n_store > n_in || (ldz & 3) || (ldx & 3) ||
        (n_store & 3) || (Yprev && (ldyp & 3)))
        return REN_ERR_BAD_ARG;
    if (ldz < ((n_out + KC - 1) / KC) * KC) return REN_ERR_BAD_ARG;
    if (prev_act != ACT_NONE && prev_act != ACT_SOFTPLUS100 && prev_act != ACT_RELU) return REN_ERR_UNSUPPORTED;
    if (prev_act != ACT_NONE && !Yprev) return REN_ERR_BAD_ARG;
    const int tiles = tiles_for(n_store);
    if (tiles < 0) return REN_ERR_UNSUPPORTED;
    if (n == 0) return REN_OK;
    DenseArgs a = {};
    a.X = dZ; a.ldx = ldz; a.W = W; a.w_rows = n_out; a.w_cols = n_in; a.red = n_out; a.n_out = n_store;
    a.act = prev_act; a.Yprev = Yprev; a.ldyp = ldyp; a.accumulate = accumulate; a.Y = dX; a.ldy = ldx; a.n = n;
    return launch_dense<true>(a, tiles, mode, (hipStream_t)stream);
}

extern "C" int ren_act_jvp_fwd(const float *Y, int32_t ldy, const float *Zd, int32_t ldz, float beta, float *Yd, int32_t ldyd,
                               int64_t rows, int32_t width, void *stream) {
    if (!Y || !Zd || !Yd || rows < 0 || width < 4 || (width & 3) || (ldy & 3) || (ldz & 3) || (ldyd & 3) || ldy < width ||
        ldz < width || ldyd < width)
        return REN_ERR_BAD_ARG;
    if (rows == 0) return REN_OK;
    hipLaunchKernelGGL(act_jvp_fwd_kernel, dim3(ren_blocks(rows * (width / 4), 256)), dim3(256), 0, (hipStream_t)stream, Y, ldy,
                       Zd, ldz, beta, Yd, ldyd, rows, width / 4);
    REN_CHECK_LAUNCH();
}

extern "C" int ren_act_jvp_bwd(const float *gy, const float *gyd, const float *Y, int32_t ldy, const float *Zd, float beta,
                               float *gz, float *gzd, int64_t rows, int32_t width, void *stream) {
    if (!gy || !gyd || !Y || !Zd || !gz || !gzd || rows < 0 || width < 4 || (width & 3) || (ldy & 3) || ldy < width)
        return REN_ERR_BAD_ARG;
    if (rows == 0) return REN_OK;
    hipLaunchKernelGGL(act_jvp_bwd_kernel, dim3(ren_blocks(rows * (width / 4), 256)), dim3(256), 0, (hipStream_t)stream, gy, gyd,
                       Y, ldy, Zd, beta, gz, gzd, rows, width / 4);
    REN_CHECK_LAUNCH();
}

extern "C" int64_t ren_dense_bwd_weight_workspace_floats(int32_t n_out, int32_t n_in, int32_t n_splits) {
    if (n_out < 1 || n_in < 1 || n_splits < 1) return -1;
    return (int64_t)n_splits * ((int64_t)n_out * n_in + n_out);
}

extern "C" int ren_dense_bwd_weight(const float *dZ, int32_t ldz, const float *X, int32_t ldx, int32_t n_out,
                                    int32_t n_in, int64_t n, int32_t n_splits, float *grad_w, float *grad_b,
                                    float *workspace, void *stream) {
    const int mode = (n_splits >> 16) & 0xff;                                  // REN_DENSE_* >> 8 in bits 16..23
    n_splits &= 0xffff;
    if (!dZ || !X || !grad_w || !workspace || n < 0 || n_out < 1 || n_in < 1 || n_splits < 1)
        return REN_ERR_BAD_ARG;
    if (ldx < ((n_in + 31) / 32) * 32 || ldz < n_out || (ldx & 3) || (ldz & 3) || n_out > 256) return REN_ERR_BAD_ARG;
    if (n == 0) return REN_OK;
    DwArgs a;
    a.dZ = dZ; a.ldz = ldz; a.N = n_out; a.X = X; a.ldx = ldx; a.K = n_in; a.n = n;
    a.slab_w = workspace; a.slab_b = workspace + (int64_t)n_splits * n_out * n_in;
    hipStream_t st = (hipStream_t)stream;
    if (mode == 1) {                                                           // bf16 matrix cores, plain bf16 operands
        const size_t lds = 2 * (size_t)1 * 256 * DWX_ST * 2;                   // 40 KiB
        (void)hipFuncSetAttribute((const void *)dense_dw_x_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(dense_dw_x_kernel<1>, dim3(n_splits), dim3(512), lds, st, a);
    } else if (mode != 0) {                                                    // bf16 matrix cores, three pieces / six terms
        const size_t lds = 2 * (size_t)3 * 256 * DWX_ST * 2;                   // 120 KiB
        (void)hipFuncSetAttribute((const void *)dense_dw_x_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(dense_dw_x_kernel<3>, dim3(n_splits), dim3(512), lds, st, a);
    } else {
        const size_t lds = 2 * 2 * 32 * 256 * sizeof(float);                   // 128 KiB: one workgroup per CU
        (void)hipFuncSetAttribute((const void *)dense_dw_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(dense_dw_kernel, dim3(n_splits), dim3(512), lds, st, a);
    }
    const int len_w = n_out * n_in;
    launch_reduce_slabs(a.slab_w, n_splits, len_w, grad_w, st);
    if (grad_b) launch_reduce_slabs(a.slab_b, n_splits, n_out, grad_b, st);    // NULL: a tangent stream (no bias)
    REN_CHECK_LAUNCH();
}

extern "C" int ren_vanilla_heads_bwd(const float *g_rgb, const float *rgb, const float *g_sigma, const float *sigma,
                                     int64_t n, int32_t C, int32_t activations, float *dz_rgb, float *dz_sigma, void *stream) {
    if (!g_rgb || !rgb || !g_sigma || !sigma || !dz_rgb || !dz_sigma || n < 0) return REN_ERR_BAD_ARG;
    if (C != 1 && C != 3) return REN_ERR_UNSUPPORTED;
    if (n == 0) return REN_OK;
    const int64_t n_pad = (n + 31) / 32 * 32;
    const ActKindsHost ak = act_kinds_host(activations);
    hipLaunchKernelGGL(heads_bwd_kernel, dim3(ren_blocks(n_pad, 256)), dim3(256), 0, (hipStream_t)stream, g_rgb, rgb,
                       g_sigma, sigma, n, n_pad, C, ak.dn, ak.rd, dz_rgb, dz_sigma);
    REN_CHECK_LAUNCH();
}

extern "C" int ren_act_jvp2_fwd(const float *Y, int32_t ldy, const float *Zd, const float *Zdd, int32_t ldz, float beta,
                                float *Yd, int32_t ldyd, float *Ydd, int32_t ldydd, int64_t rows, int32_t width, void *stream) {
    if (!Y || !Zd || !Zdd || !Yd || !Ydd || rows < 0 || width < 4 || (width & 3) || (ldy & 3) || (ldz & 3) || (ldyd & 3) ||
        (ldydd & 3) || ldy < width || ldz < width || ldyd < width || ldydd < width)
        return REN_ERR_BAD_ARG;
    if (rows == 0) return REN_OK;
    hipLaunchKernelGGL(act_jvp2_fwd_kernel, dim3(ren_blocks(rows * (width / 4), 256)), dim3(256), 0, (hipStream_t)stream, Y, ldy,
                       Zd, Zdd, ldz, beta, Yd, ldyd, Ydd, ldydd, rows, width / 4);
    REN_CHECK_LAUNCH();
}

extern "C" int ren_vanilla_heads_jvp(const float *rgb, const float *sigma, const float *zod, const float *zodd,
                                     const float *zsd, const float *zsdd, int64_t n, int32_t C, int32_t activations, float *rgbd, float *rgbdd,
                                     float *sigmad, float *sigmadd, void *stream) {
    if (!rgb || !sigma || !zod || !zsd || !rgbd || !sigmad || n < 0) return REN_ERR_BAD_ARG;
    if ((rgbdd || sigmadd) && (!rgbdd || !sigmadd || !zodd || !zsdd)) return REN_ERR_BAD_ARG;
    if (C != 1 && C != 3) return REN_ERR_UNSUPPORTED;
    if (n == 0) return REN_OK;
    const ActKindsHost ak = act_kinds_host(activations);
    hipLaunchKernelGGL(heads_jvp_kernel, dim3(ren_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, rgb, sigma, zod, zodd,
                       zsd, zsdd, n, C, ak.dn, ak.rd, rgbd, rgbdd, sigmad, sigmadd);
    REN_CHECK_LAUNCH();
}

extern "C" int ren_vanilla_heads_bwd_jvp(const float *g_rgb, const float *g_rgbd, const float *g_sigma, const float *g_sigmad,
                                         const float *rgb, const float *sigma, const float *zod, const float *zsd, int64_t n,
                                         int32_t C, int32_t activations, float *dz_rgb, float *dzd_rgb, float *dz_sigma, float *dzd_sigma,
                                         void *stream) {
    if (!g_rgb || !g_rgbd || !g_sigma || !g_sigmad || !rgb || !sigma || !zod || !zsd || !dz_rgb || !dzd_rgb || !dz_sigma ||
        !dzd_sigma || n < 0)
        return REN_ERR_BAD_ARG;
    if (C != 1 && C != 3) return REN_ERR_UNSUPPORTED;
    if (n == 0) return REN_OK;
    const int64_t n_pad = (n + 31) / 32 * 32;
    const ActKindsHost ak = act_kinds_host(activations);
    hipLaunchKernelGGL(heads_bwd_jvp_kernel, dim3(ren_blocks(n_pad, 256)), dim3(256), 0, (hipStream_t)stream, g_rgb, g_rgbd,
                       g_sigma, g_sigmad, rgb, sigma, zod, zsd, n, n_pad, C, ak.dn, ak.rd, dz_rgb, dzd_rgb, dz_sigma, dzd_sigma);
    REN_CHECK_LAUNCH();
}
