// tower_cb.cuh — the constant input planes of game_repr as a per-position bias of layer 0 (TowerParams.cb / TowerS3Params.cb),
// shared by the exact-f32 towers (net_kernels.hip) and the split-bf16 towers (net_s3_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "board.cuh"
#include "conv_mainloop.cuh"

namespace tg {

// PB[position p][border class][F] = bias + Σ_{constant planes that are set} S[plane][class] + fcd · S[fcd plane][class], summed in
// exactly this order by every kernel that uses it (a position's result must not depend on the kernel that evaluates it).
// class = 3·(y = 0 ? 0 : y = n − 1 ? 2 : 1) + (x = 0 ? 0 : x = n − 1 ? 2 : 1): which of the 9 taps stay on the board.
// Called by the whole wave that holds position p's state; F4 = F / 4; S4 = S as float4; B4 = the layer's bias as float4.
__device__ __forceinline__ void tower_cb_table(const WState& ws, float fcd, int n, int p, int F4, const f32x4* __restrict__ S4,
                                               const f32x4* __restrict__ B4, f32x4* pb4) {
    const int lane = threadIdx.x & 63;
    int st0, cp0;
    starting_stones(n, st0, cp0);
    // the constant planes that are set (ws_row_mask's rules), as indices into S
    const bool w = ws.to_move == 0;
    const int my_st = w ? ws.ws : ws.bs, en_st = w ? ws.bs : ws.ws, my_cp = w ? ws.wc : ws.bc, en_cp = w ? ws.bc : ws.wc;
    int pl[5];
    pl[0] = (my_st > 0 && my_st <= st0) ? my_st - 1 : -1;
    pl[1] = (en_st > 0 && en_st <= st0) ? st0 + en_st - 1 : -1;
    pl[2] = (my_cp > 0 && my_cp <= cp0) ? 2 * st0 + my_cp - 1 : -1;
    pl[3] = (en_cp > 0 && en_cp <= cp0) ? 2 * st0 + cp0 + en_cp - 1 : -1;
    pl[4] = w ? 2 * st0 + 2 * cp0 : -1;
    const int fplane = 2 * st0 + 2 * cp0 + 1;
    for (int idx0 = 0; idx0 < 9 * F4; idx0 += 64) {  // idx = class·F/4 + channel quad
        const int idx = min(idx0 + lane, 9 * F4 - 1);
        const int cls = idx / F4;
        // all seven rows are requested before the first add (a row behind `if (plane is set)` was a load the next one waited
        // for: six L2 round trips in a row, per round and position); an unset plane reads row 0 and is not added — the sum
        // and its order are those of the conditional form, bit for bit
        f32x4 v = B4[idx - cls * F4];
        f32x4 t[5];
#pragma unroll
        for (int k = 0; k < 5; k++) t[k] = S4[(size_t)(pl[k] >= 0 ? pl[k] : 0) * 9 * F4 + idx];
        const f32x4 sf = S4[(size_t)fplane * 9 * F4 + idx];
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const f32x4 u = v + t[k];
            v = pl[k] >= 0 ? u : v;
        }
        v += f32x4{fcd * sf[0], fcd * sf[1], fcd * sf[2], fcd * sf[3]};
        if (idx0 + lane < 9 * F4) pb4[(size_t)p * 9 * F4 + idx] = v;
    }
}
// the 32 board-plane values of this lane's square as eight float quads (planes ≥ board_channels(n) are zero)
__device__ __forceinline__ void tower_cb_board_quads(const RowMask& m, int n, f32x4 (&qd)[8]) {
    const uint32_t bits = m.w[0] & ((1u << board_channels(n)) - 1u);  // board_channels ≤ 28
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint32_t nib = (bits >> (4 * k)) & 15u;
        qd[k] = f32x4{(nib & 1u) ? 1.0f : 0.0f, (nib & 2u) ? 1.0f : 0.0f, (nib & 4u) ? 1.0f : 0.0f, (nib & 8u) ? 1.0f : 0.0f};
    }
}
// index (in f32x4) of a row's entry of PB for channel quad chq
__device__ __forceinline__ int tower_cb_index(int rho, int rows, int n, int nsq, int F4, int chq) {
    if (rho >= rows) return chq;
    const int p = rho / nsq, sq = rho - p * nsq, y = sq / n, x = sq - y * n;
    const int cls = (y == 0 ? 0 : y == n - 1 ? 2 : 1) * 3 + (x == 0 ? 0 : x == n - 1 ? 2 : 1);
    return (p * 9 + cls) * F4 + chq;
}

}  // namespace tg
