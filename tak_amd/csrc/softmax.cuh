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
__device__ inline float wave_max(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d));
    return v;
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

}  // namespace tg
