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
// NB: the board size as a compile-time constant (0 = `n`): geometry masks fold, divisions by n become multiply-shifts.
template <int NB, int CS = 0>
__global__ __launch_bounds__(256) void k_board_pass(const uint8_t* __restrict__ states, const uint16_t* __restrict__ moves, int count,
                                                    int n, uint8_t* __restrict__ out_states, uint8_t* __restrict__ results,
                                                    int32_t* __restrict__ counts, float* __restrict__ planes, int cstride) {
    int gi = wave_global_id();
    if (gi >= count) return;
    Geom g = make_geom(NB ? NB : n);
    WState s;
    ws_load(s, states + (size_t)gi * g.bytes, g);
    ws_play(s, uni((uint32_t)moves[gi]), g);
    uint32_t r = ws_result(s, g);
    int c = r == TG_ONGOING ? ws_movegen(s, g, 0, [](int, uint32_t) {}) : 0;
    ws_store(s, out_states + (size_t)gi * g.bytes, g);
    // CS: the planes' row stride as a constant (the launcher instantiates the unpadded row of the micro-benchmark, 72 / 92
    // channels, and the 16-channel-padded one the per-layer conv kernels read)
    if (CS) ws_encode<true, CS>(s, g, planes + (size_t)gi * CS * g.nsq, CS);
    else ws_encode<true>(s, g, planes + (size_t)gi * cstride * g.nsq, cstride);
    if (lane_id() == 0) { results[gi] = (uint8_t)r; counts[gi] = c; }
}

// Symmetry (tak/src/symm.rs) + Example::to_tensors (alpha-tak/src/example.rs:62-78): one wave per
// (example, symmetry).  Squares: rotate (col,row) → (row, n-1-col), mirror col → n-1-col; symmetry i < 4 is
// rotate^i, i ≥ 4 is mirror then rotate^(i-4).  Directions follow the squares (Up→Right→Down→Left, Left↔Right).
__device__ inline void sym_apply(int n, int i, int& col, int& row) {
    if (i >= 4) col = n - 1 - col;
    for (int k = 0; k < (i & 3); k++) { int c = row, r = n - 1 - col; col = c; row = r; }
}
__device__ inline void sym_apply_inverse(int n, int i, int& col, int& row) {
    for (int k = 0; k < (i & 3); k++) { int c = n - 1 - row, r = col; col = c; row = r; }
    if (i >= 4) col = n - 1 - col;
}
__device__ inline uint32_t sym_dir(int i, uint32_t d) {
    if (i >= 4) d = d == LEFT ? RIGHT : d == RIGHT ? LEFT : d;
    for (int k = 0; k < (i & 3); k++) d = d == UP ? RIGHT : d == RIGHT ? DOWN : d == DOWN ? LEFT : UP;
    return d;
}

__global__ __launch_bounds__(256) void k_augment(const uint8_t* __restrict__ states, const int32_t* __restrict__ n_moves,
                                                 const uint16_t* __restrict__ moves, const uint32_t* __restrict__ visits, int count,
                                                 int n, int P, int legacy5, const int16_t* __restrict__ lut5,
                                                 uint8_t* __restrict__ out_states, float* __restrict__ pi) {
    int wi = wave_global_id();
    if (wi >= count * 8) return;
    const int ex = wi >> 3, sym = wi & 7;
    const int lane = lane_id();
    Geom g = make_geom(n);
    WState s;
    ws_load(s, states + (size_t)ex * g.bytes, g);
    // board: the square this lane ends up holding comes from its pre-image under the symmetry
    int col = lane % n, row = lane / n;
    sym_apply_inverse(n, sym, col, row);
    int src = lane < g.nsq ? row * n + col : lane;
    WState t = s;
    t.stack = shfl64(s.stack, src);
    t.height = (uint32_t)__shfl((int)s.height, src);
    t.top = (uint32_t)__shfl((int)s.top, src);
    ws_store(t, out_states + (size_t)wi * g.bytes, g);
    // policy target: visits / total at the index of the transformed move (pi is zeroed by the host)
    const int nm = n_moves[ex];
    const uint32_t* vs = visits + (size_t)ex * TG_MAX_MOVES;
    const uint16_t* mv = moves + (size_t)ex * TG_MAX_MOVES;
    uint32_t part = 0;
    for (int k = lane; k < nm; k += 64) part += vs[k];
    for (int d = 32; d >= 1; d >>= 1) part += (uint32_t)__shfl_xor((int)part, d);
    const float total = (float)part;
    float* row_pi = pi + (size_t)wi * P;
    for (int k = lane; k < nm; k += 64) {
        uint32_t m = mv[k];
        int c = (int)(m & 63u) % n, r = (int)(m & 63u) / n;
        sym_apply(n, sym, c, r);
        uint32_t pat = m >> 8, f = (m >> 6) & 3u;
        if (pat) f = sym_dir(sym, f);
        uint32_t tm = (uint32_t)(r * n + c) | (f << 6) | (pat << 8);
        int idx = move_index_dev(tm, n, legacy5 != 0, lut5);
        if (idx >= 0 && idx < P) row_pi[idx] = (float)vs[k] / total;
    }
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
    if (count <= 0) return;
#define TG_BP(NB, CS) hipLaunchKernelGGL((k_board_pass<NB, CS>), wave_grid(count), dim3(256), 0, st, states, moves, count, n, out_states, results, counts, planes, cstride)
    if (n == 5 && cstride == 72) TG_BP(5, 72);
    else if (n == 5 && cstride == 80) TG_BP(5, 80);
    else if (n == 6 && cstride == 92) TG_BP(6, 92);
    else if (n == 6 && cstride == 96) TG_BP(6, 96);
    else TG_BP(0, 0);
#undef TG_BP
}
void launch_augment(hipStream_t st, const uint8_t* states, const int32_t* n_moves, const uint16_t* moves, const uint32_t* visits, int count,
                    int n, int P, bool legacy5, const int16_t* lut5, uint8_t* out_states, float* pi) {
    if (count > 0) hipLaunchKernelGGL(k_augment, wave_grid(count * 8), dim3(256), 0, st, states, n_moves, moves, visits, count, n, P,
                                      legacy5 ? 1 : 0, lut5, out_states, pi);
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
