// softmax.cuh — the reduction order of the policy softmax, shared by k_softmax (net_kernels.hip: one 256-thread block per
// position) and the tree backup (search_kernels.hip: one wave per game computes the same statistics from the logits, so
// the probabilities never go through HBM).  Both must return the same BITS: a host-side MCTS (the narrow seam, the parity
// tests) builds its trees from tg_policy_eval's probabilities, the engine from the in-kernel ones.
#pragma once
#include <hip/hip_runtime.h>

namespace tg {

__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}
// (the maximum does not depend on the order: six DPP steps — an inclusive max-scan, lane 63 holds the total — and a v_readlane
// instead of six ds_bpermute round trips; wave_sum keeps its butterfly: its association order is part of the softmax's bits)
__device__ inline float wave_max(float v) {
#define TG_WMAX_STEP(CTRL, ROWS) \
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(-INFINITY), __float_as_int(v), CTRL, ROWS, 0xf, false)));
    TG_WMAX_STEP(0x111, 0xf) TG_WMAX_STEP(0x112, 0xf) TG_WMAX_STEP(0x114, 0xf) TG_WMAX_STEP(0x118, 0xf)
    TG_WMAX_STEP(0x142, 0xa) TG_WMAX_STEP(0x143, 0xc)
#undef TG_WMAX_STEP
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

constexpr int SOFTMAX_KEEP = 8;  // elements per thread of the 256-thread block: rows up to 2048 outputs stay in registers

// max and 1/Σexp of a row of P ≤ 2048 logits exactly as k_softmax's register path computes them, by ONE wave: thread t of
// the block sums exp(x[t + 256k] − max) over k ascending, the four waves butterfly-reduce, the block adds
// (w0 + w1) + (w2 + w3).  Lane L plays threads L, 64 + L, 128 + L, 192 + L in turn.  (max is exact in any order.)
__device__ inline void softmax_stats_wave(const float* __restrict__ x, int P, float& mx_out, float& inv_out) {
    const int lane = threadIdx.x & 63;
    // thread (w, lane) of the block owns x[64·w + lane + 256·k] = x[lane + 64·(w + 4k)]: over the four w a lane touches
    // x[lane + 64·j], j = 0..31 — one register-resident pass serves the max and all four partial sums
    constexpr int J = 4 * SOFTMAX_KEEP;
    float v[J];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < J; j++) {
        const int p = lane + 64 * j;
        v[j] = p < P ? x[p] : -INFINITY;
        mx = fmaxf(mx, v[j]);
    }
    mx = wave_max(mx);
#pragma unroll
    for (int j = 0; j < J; j++) v[j] = (lane + 64 * j) < P ? expf(v[j] - mx) : 0.0f;
    float red[4];
#pragma unroll
    for (int w = 0; w < 4; w++) {
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < SOFTMAX_KEEP; k++) s += v[w + 4 * k];
        red[w] = wave_sum(s);
    }
    const float s = (red[0] + red[1]) + (red[2] + red[3]);
    mx_out = mx;
    inv_out = 1.0f / s;
}

// ------------------------------------------------------------------------------------------------------------------
// Block-wise softmax statistics of the FC policy head (exact-f32 path).  The lane that holds a logit in its accumulators is the
// cheapest place to take max and Σexp: per (row, block) the FC's epilogue emits  m_b = max of the block's policy columns,
// s_b = Σ exp(x − m_b)  — 96 B per row instead of the tree backup re-reading all 1575 logits (6.3 KB per game and iteration).
// Block geometry = what one wave of k_fc_ring holds of a row (round 4): the 1576 useful columns are 99 MFMA tiles of 16;
//   blocks 0 … 7   tiles 12 b … 12 b + 11  (192 columns: a workgroup column's 12 main tiles),
//   blocks 8 … 10  tile 96 + (b − 8)       (16 columns: the three leftover tiles, each computed by another workgroup),
// and pair FC_STAT_BLOCKS of a row's record holds {value pre-activation (column P), 0}: FC_STAT_STRIDE = 12 pairs per row.
// The canonical association order, followed by the FC epilogue (fc_block_stats), by k_fc_stats (any other producer of the
// logits) and therefore by everything that consumes the statistics (k_softmax_stats for tg_policy_eval, the tree backup):
//   lane (r16, q) of the row's wave holds columns  col0 + 16 j + 4 q + t  (j = 0 … tiles − 1, t = 0..3);
//   lane partial = ((…(e(0,0) + e(0,1)) + e(0,2)) + …) + e(tiles − 1, 3),  e(j,t) = exp(x − m_b) (stat_exp) or 0 beyond the policy columns;
//   s_b = (s_q + s_{q^1}) + (s_{q^2} + s_{q^3})   (butterfly over lanes 16 and 32 apart);
//   M = max_b m_b,  S = ((s_0·exp(m_0 − M) + s_1·exp(m_1 − M)) + …) + s_10·exp(m_10 − M),  p(x) = exp(x − M) · (1 / S).
// (Round 3 used 8 blocks of 208 columns: another association of the same sums — other low bits of the probabilities, inside the
// 1e-4 gate by the same margin; every producer and consumer here follows the one geometry, so they still agree bit for bit.)
// ------------------------------------------------------------------------------------------------------------------
// exp of this path: v_exp_f32(x · log2 e) — two instructions where expf() is a dozen (52 of them per lane made the FC's epilogue
// 2.2 µs longer than the 4 µs the backup saved).  Arguments are ≤ 0; relative error ≈ 1e-6 at x = −20, far inside the 1e-4
// gate against PyTorch.  EVERY consumer of the statistics uses this same function, so they agree bit for bit.
__device__ __forceinline__ float stat_exp(float x) { return __expf(x); }

constexpr int FC_MAIN_TILES = 12;                                        // tiles per main block
constexpr int FC_MAIN_BLOCKS = 8;
constexpr int FC_X_TILES = 3;                                            // leftover tiles = single-tile blocks
constexpr int FC_TILES = FC_MAIN_TILES * FC_MAIN_BLOCKS + FC_X_TILES;    // 99 tiles = 1584 columns ≥ 1575 + 1
constexpr int FC_STAT_BLOCKS = FC_MAIN_BLOCKS + FC_X_TILES;              // 11
constexpr int FC_STAT_STRIDE = FC_STAT_BLOCKS + 1;                       // pairs per row: the blocks, then {value pre-activation, 0}
__host__ __device__ inline int fc_stat_col0(int b) { return (b < FC_MAIN_BLOCKS ? b * FC_MAIN_TILES : FC_MAIN_TILES * FC_MAIN_BLOCKS + (b - FC_MAIN_BLOCKS)) * 16; }
__host__ __device__ inline int fc_stat_tiles(int b) { return b < FC_MAIN_BLOCKS ? FC_MAIN_TILES : 1; }

using softmax_f32x4 = __attribute__((ext_vector_type(4))) float;

// v[j] = the lane's four logits of tile j (bias added); col0 = the lane's first column of tile 0 (block start + 4 q); columns
// ≥ n_lim (= min(P, end of the block)) do not take part.  All 64 lanes call.  TILES = 12 / 1 on the exact-f32 FC's main and
// leftover blocks (a 1-tile block evaluated with TILES = 12 and n_lim at its end gives the same bits: the masked terms add 0),
// 7 on the split-bf16 FC (k_fc_s3b: 15 blocks of 112 columns — the block width belongs to the path, and a path's consumers only
// ever see that path's statistics).
template <int TILES>
__device__ __forceinline__ void fc_block_stats(const softmax_f32x4 (&v)[TILES], int col0, int n_lim, float& m_out, float& s_out) {
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < TILES; j++)
#pragma unroll
        for (int t = 0; t < 4; t++)
            if (col0 + 16 * j + t < n_lim) m = fmaxf(m, v[j][t]);
    m = fmaxf(m, __shfl_xor(m, 16));
    m = fmaxf(m, __shfl_xor(m, 32));
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < TILES; j++)
#pragma unroll
        for (int t = 0; t < 4; t++) s += (col0 + 16 * j + t < n_lim) ? stat_exp(v[j][t] - m) : 0.0f;
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    m_out = m;
    s_out = s;
}

// (M, 1/S) of a row from its per-block statistics stats[2b] = m_b, stats[2b + 1] = s_b; nblocks ≤ 64; called by a whole wave with
// a wave-uniform `stats`, every lane gets the result.  Lane b fetches block b's pair — ONE memory round trip; read one after
// the other through a wave-uniform pointer the pairs became scalar loads in two loops, 12 round trips in a row on the tree
// backup's critical path — and the sum runs over the lanes in block order, so the bits are those of the sequential loop.
__device__ __forceinline__ void fc_combine_stats(const float* __restrict__ stats, int nblocks, float& mx_out, float& inv_out,
                                                 float* tail = nullptr) {
    // tail (optional): first float of pair `nblocks` of the same record (the exact-f32 FC keeps the value pre-activation there),
    // fetched by lane `nblocks` with the same instruction
    const int lane = threadIdx.x & 63;
    const float2 ms = ((const float2*)stats)[lane < nblocks + (tail ? 1 : 0) ? lane : 0];
    const float M = wave_max(lane < nblocks ? ms.x : -INFINITY);
    const float term = ms.y * stat_exp(ms.x - M);
    float S = 0.0f;
    for (int b = 0; b < nblocks; b++) S += __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(term), b));
    mx_out = M;
    inv_out = 1.0f / S;
    if (tail) *tail = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(ms.x), nblocks));
}

}  // namespace tg
