// board_kernels.hip — batch board operators: one wavefront per game (see board.cuh).
// Kernels behind tg_movegen / tg_play / tg_result / tg_encode / tg_move_index / tg_perft.
#include "board.cuh"
#include "kernels.h"

namespace tg {

constexpr int WAVES_PER_BLOCK = 4;

__device__ inline int wave_global_id() { return (int)(blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6)); }

__global__ __launch_bounds__(256) void k_movegen(const uint8_t* __restrict__ states, int count, int n,
                                                 uint16_t* __restrict__ moves, int32_t* __restrict__ counts) {
    int gi = wave_global_id();
    if (gi >= count) return;
    Geom g = make_geom(n);
    WState s;
    ws_load(s, states + (size_t)gi * g.bytes, g);
    uint16_t* out = moves + (size_t)gi * TG_MAX_MOVES;
    int c = ws_movegen(s, g, TG_MAX_MOVES, [&](int idx, uint32_t code) { out[idx] = (uint16_t)code; });
    if (lane_id() == 0) counts[gi] = c;
}

__global__ __launch_bounds__(256) void k_play(uint8_t* __restrict__ states, int count, int n,
                                              const uint16_t* __restrict__ moves, uint8_t* __restrict__ status) {
    int gi = wave_global_id();
    if (gi >= count) return;
    Geom g = make_geom(n);
    WState s;
    uint8_t* st = states + (size_t)gi * g.bytes;
    ws_load(s, st, g);
    uint32_t err = ws_play(s, uni(moves[gi]), g);
    if (!err) ws_store(s, st, g);
    if (lane_id() == 0) status[gi] = (uint8_t)err;
}

__global__ __launch_bounds__(256) void k_result(const uint8_t* __restrict__ states, int count, int n,
                                                uint8_t* __restrict__ results) {
    int gi = wave_global_id();
    if (gi >= count) return;
    Geom g = make_geom(n);
    WState s;
    ws_load(s, states + (size_t)gi * g.bytes, g);
    uint32_t r = ws_result(s, g);
    if (lane_id() == 0) results[gi] = (uint8_t)r;
}

template <bool NHWC>
__global__ __launch_bounds__(256) void k_encode(const uint8_t* __restrict__ states, int count, int n,
                                                float* __restrict__ planes, int cstride) {
    int gi = wave_global_id();
    if (gi >= count) return;
    Geom g = make_geom(n);
    WState s;
    ws_load(s, states + (size_t)gi * g.bytes, g);
    ws_encode<NHWC>(s, g, planes + (size_t)gi * cstride * g.nsq, cstride);
}

__global__ void k_move_index(const uint16_t* __restrict__ moves, int count, int n, int legacy5,
                             const int16_t* __restrict__ lut5, int32_t* __restrict__ index) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint32_t mv = moves[i];
    int idx = -1;
    if ((int)(mv & 63u) < n * n) idx = move_index_dev(mv, n, legacy5 != 0, lut5);
    index[i] = idx;
}

// ---- perft (tak/tests/perft.rs:3-18), level by level ------------------------------------------
// pass 1: per frontier state, its number of children (0 if terminal) and its terminal flag
__global__ __launch_bounds__(256) void k_perft_count(const uint8_t* __restrict__ states, int count, int n,
                                                     int32_t* __restrict__ nchild, uint8_t* __restrict__ terminal) {
    int gi = wave_global_id();
    if (gi >= count) return;
    Geom g = make_geom(n);
    WState s;
    ws_load(s, states + (size_t)gi * g.bytes, g);
    uint32_t r = ws_result(s, g);
    int c = 0;
    if (r == TG_ONGOING) c = ws_movegen(s, g, 0, [](int, uint32_t) {});
    if (lane_id() == 0) { nchild[gi] = c; terminal[gi] = r != TG_ONGOING; }
}

// pass 2: write every child state of every non-terminal frontier state at offsets[gi] + k
__global__ __launch_bounds__(256) void k_perft_expand(const uint8_t* __restrict__ states, int count, int n,
                                                      const int64_t* __restrict__ offsets, const int32_t* __restrict__ root_of,
                                                      uint8_t* __restrict__ next_states, int32_t* __restrict__ next_root) {
    __shared__ uint16_t mv_lds[WAVES_PER_BLOCK][TG_MAX_MOVES];
    int gi = wave_global_id();
    if (gi >= count) return;
    Geom g = make_geom(n);
    WState s;
    ws_load(s, states + (size_t)gi * g.bytes, g);
    if (ws_result(s, g) != TG_ONGOING) return;
    uint16_t* mv = mv_lds[threadIdx.x >> 6];
    int c = ws_movegen(s, g, TG_MAX_MOVES, [&](int idx, uint32_t code) { mv[idx] = (uint16_t)code; });
    __builtin_amdgcn_wave_barrier();
    int64_t off = offsets[gi];
    int root = root_of[gi];
    for (int k = 0; k < c; k++) {
        WState t = s;
        ws_play(t, (uint32_t)mv[k], g);
        ws_store(t, next_states + (size_t)(off + k) * g.bytes, g);
        if (lane_id() == 0) next_root[off + k] = root;
    }
}

// One fused pass of the board path over a batch (the micro-benchmark of SURVEY.md §8d): play a move,
// evaluate the terminal test, count the legal moves and encode the NHWC planes.  Algorithmic bytes per
// position = state in + state out + f32 planes (256 + 256 + 7200 on 5×5).
__global__ __launch_bounds__(256) void k_board_pass(const uint8_t* __restrict__ states, const uint16_t* __restrict__ moves, int count,
                                                    int n, uint8_t* __restrict__ out_states, uint8_t* __restrict__ results,
                                                    int32_t* __restrict__ counts, float* __restrict__ planes, int cstride) {
    int gi = wave_global_id();
    if (gi >= count) return;
    Geom g = make_geom(n);
    WState s;
    ws_load(s, states + (size_t)gi * g.bytes, g);
    ws_play(s, uni((uint32_t)moves[gi]), g);
    uint32_t r = ws_result(s, g);
    int c = r == TG_ONGOING ? ws_movegen(s, g, 0, [](int, uint32_t) {}) : 0;
    ws_store(s, out_states + (size_t)gi * g.bytes, g);
    ws_encode<true>(s, g, planes + (size_t)gi * cstride * g.nsq, cstride);
    if (lane_id() == 0) { results[gi] = (uint8_t)r; counts[gi] = c; }
}

// ---- launchers --------------------------------------------------------------------------------
static inline dim3 wave_grid(int count) { return dim3((count + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK); }

void launch_movegen(hipStream_t st, const uint8_t* states, int count, int n, uint16_t* moves, int32_t* counts) {
    if (count > 0) hipLaunchKernelGGL(k_movegen, wave_grid(count), dim3(256), 0, st, states, count, n, moves, counts);
}
void launch_play(hipStream_t st, uint8_t* states, int count, int n, const uint16_t* moves, uint8_t* status) {
    if (count > 0) hipLaunchKernelGGL(k_play, wave_grid(count), dim3(256), 0, st, states, count, n, moves, status);
}
void launch_result(hipStream_t st, const uint8_t* states, int count, int n, uint8_t* results) {
    if (count > 0) hipLaunchKernelGGL(k_result, wave_grid(count), dim3(256), 0, st, states, count, n, results);
}
void launch_encode(hipStream_t st, const uint8_t* states, int count, int n, float* planes, bool nhwc) {
    if (count <= 0) return;
    int c = input_channels(n);
    if (nhwc) hipLaunchKernelGGL(k_encode<true>, wave_grid(count), dim3(256), 0, st, states, count, n, planes, c);
    else hipLaunchKernelGGL(k_encode<false>, wave_grid(count), dim3(256), 0, st, states, count, n, planes, c);
}
void launch_encode_nhwc(hipStream_t st, const uint8_t* states, int count, int n, float* planes, int cstride) {
    if (count > 0) hipLaunchKernelGGL(k_encode<true>, wave_grid(count), dim3(256), 0, st, states, count, n, planes, cstride);
}
void launch_board_pass(hipStream_t st, const uint8_t* states, const uint16_t* moves, int count, int n, uint8_t* out_states,
                       uint8_t* results, int32_t* counts, float* planes, int cstride) {
    if (count > 0) hipLaunchKernelGGL(k_board_pass, wave_grid(count), dim3(256), 0, st, states, moves, count, n, out_states, results, counts, planes, cstride);
}
void launch_move_index(hipStream_t st, const uint16_t* moves, int count, int n, bool legacy5, const int16_t* lut5, int32_t* index) {
    if (count > 0) hipLaunchKernelGGL(k_move_index, dim3((count + 255) / 256), dim3(256), 0, st, moves, count, n, legacy5 ? 1 : 0, lut5, index);
}
void launch_perft_count(hipStream_t st, const uint8_t* states, int count, int n, int32_t* nchild, uint8_t* terminal) {
    if (count > 0) hipLaunchKernelGGL(k_perft_count, wave_grid(count), dim3(256), 0, st, states, count, n, nchild, terminal);
}
void launch_perft_expand(hipStream_t st, const uint8_t* states, int count, int n, const int64_t* offsets, const int32_t* root_of,
                         uint8_t* next_states, int32_t* next_root) {
    if (count > 0) hipLaunchKernelGGL(k_perft_expand, wave_grid(count), dim3(256), 0, st, states, count, n, offsets, root_of, next_states, next_root);
}

}  // namespace tg
