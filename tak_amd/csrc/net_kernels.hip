// net_kernels.hip — policy/value resnet forward on CDNA4 matrix cores, exact f32.
//
// Replaces the libtorch ops behind reference alpha-tak/src/model/{net5,net6,res_block}.rs
// (conv2d 3×3 pad 1 + batch_norm(eval) + relu + residual add, linear, softmax, tanh).
//
// Layout: activations are NHWC — row m = (position b, square sq), F contiguous channels — so a 3×3 convolution is an
// implicit GEMM  out[m][o] = Σ_{tap,c} X[nbr(m,tap)][c] · W[tap,c][o]  with M = B·N² rows, K = 9·C.  Halos never cross
// positions, so a workgroup keeps whole positions in LDS and a tap is an LDS offset: im2col lives in address arithmetic.
// Arithmetic: v_mfma_f32_16x16x4_f32 in the fused towers and the policy FC, v_mfma_f32_32x32x2_f32 in the generic per-layer
// kernels — f32 in, f32 accumulate, bit-exact fmaf chains (the parity path; 157.3 TFLOP/s peak).  BatchNorm (eval) is
// folded into weights / bias at load time; bias, residual and ReLU are fused into the accumulator epilogue.
//
// Kernels, in the order of the file:
//   k_conv3x3, k_conv_pos   one 3×3 layer (shapes the fused towers do not cover; the conv policy head of Net6)
//   k_tower                 conv0 + the whole residual tower in ONE launch on the plain LDS image (one zero row for
//                           off-board taps, per-tap masks): batches below 2048 / 1024 / 512 positions
//   k_tower_halo            the same tower on the HALO image (zero cells between board rows and positions, taps as
//                           ds_read immediates, conflict-free slot table): full batches, 89 – 94 % of the MFMA peak
//   k_gemm                  generic GEMM
//   k_fc_ring               policy FC for full batches: LDS-DMA ring of three K-steps, flag counters instead of barriers
//   k_fc_small              policy FC for ≤ 2048 rows (no LDS, no barrier)
//   k_softmax(_conv), k_value_head
// Every variant of a layer type performs the same products in the same order: a position's outputs are the same bits
// whatever batch (and therefore kernel) evaluates it (tests/test_gpu_net.py, tests/test_gpu_variants.py).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <mutex>
#include <vector>

#include "board.cuh"
#include "conv_mainloop.cuh"
#include "fc_ring.cuh"
#include "softmax.cuh"
#include "tower_cb.cuh"
#include "kernels.h"

namespace tg {

// Diagnostic build only (scripts/probes/tower_stamps.hip, -DTG_TOWER_STAMPS): s_memtime stamps of workgroup 0's waves at
// the phase boundaries of every layer, written to a buffer nothing else reads.  The product build compiles none of it.
#ifdef TG_TOWER_STAMPS
__device__ unsigned long long* g_tower_stamps = nullptr;  // [layer][wave][8]
#define TG_STAMP(layer, slot)                                                                                       \
    do {                                                                                                            \
        if (blockIdx.x == 0 && g_tower_stamps && (threadIdx.x & 63) == 0)                                           \
            g_tower_stamps[((size_t)(layer) * 16 + (threadIdx.x >> 6)) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define TG_STAMP(layer, slot) do { } while (0)
#endif


// Workgroup = 4 waves as 2 (rows) × 2 (cols); each wave owns RT×CT tiles of 32×32.
template <int RT, int CT>
__global__ __launch_bounds__(256) void k_conv3x3(const float* __restrict__ in, const float* __restrict__ Wp,
                                                 const float* __restrict__ bias, const float* __restrict__ res,
                                                 float* __restrict__ out, int M, int n, int Cpad, int CoutP,
                                                 int out_stride, int cout_valid, int relu) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int TM = 64 * RT;
    const int tid = threadIdx.x;
    const int nsq = n * n;
    const int LS = Cpad + LDS_PAD;
    const int m0 = blockIdx.x * TM;
    const int mlast = min(m0 + TM, M) - 1;
    const int pos0 = m0 / nsq;
    const int npos = mlast / nsq - pos0 + 1;
    const int rows = npos * nsq;

    // ---- stage the touched positions (contiguous in global) into padded LDS rows ----
    {
        const int vpr = Cpad >> 2;  // float4 per row
        const float4* src = (const float4*)(in + (size_t)pos0 * nsq * Cpad);
        const int total = rows * vpr;
        for (int idx = tid; idx < total; idx += 256) {
            int r = idx / vpr, v = idx - r * vpr;
            float4 x = src[idx];
            *(float4*)&lds[r * LS + v * 4] = x;
        }
        for (int idx = tid; idx < LS; idx += 256) lds[rows * LS + idx] = 0.0f;  // the zero row
    }
    __syncthreads();

    const int wave = tid >> 6, lane = tid & 63;
    const int wr = wave & 1, wc = wave >> 1;
    const int i = lane & 31, h = lane >> 5;
    const int zero_off = rows * LS + 4 * h;

    int base_off[RT], py[RT], px[RT];
    bool valid[RT];
#pragma unroll
    for (int rt = 0; rt < RT; rt++) {
        int m = m0 + (wr * RT + rt) * 32 + i;
        valid[rt] = m < M;
        int mm = valid[rt] ? m : m0;
        int p = mm / nsq;
        int sq = mm - p * nsq;
        py[rt] = sq / n;
        px[rt] = sq - py[rt] * n;
        base_off[rt] = ((p - pos0) * nsq + sq) * LS + 4 * h;
    }

    f32x16 acc[RT][CT];
#pragma unroll
    for (int rt = 0; rt < RT; rt++)
#pragma unroll
        for (int ct = 0; ct < CT; ct++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[rt][ct][r] = 0.0f;

    const int col0 = blockIdx.y * (64 * CT) + wc * (32 * CT);
    const int chunks = Cpad >> 3;
    // B fragment base: Wp[k/16][col][16]; 8-wide sub-chunk c8 lives at [c8>>1][col][8*(c8&1) ..]; this lane:
    // column col0 + ct*32 + i, k sub-offset 4h
    const float* wlane = Wp + ((size_t)(col0 + i) * 16 + 4 * h);
    const size_t wchunk_stride = (size_t)CoutP * 16;

    for (int tap = 0; tap < 9; tap++) {
        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
        int aoff[RT];
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
            int yy = py[rt] + dy, xx = px[rt] + dx;
            bool ok = valid[rt] && yy >= 0 && yy < n && xx >= 0 && xx < n;
            aoff[rt] = ok ? base_off[rt] + (dy * n + dx) * LS : zero_off;
        }
        const float* wtap = wlane + (size_t)tap * (chunks >> 1) * wchunk_stride;
        for (int c8 = 0; c8 < chunks; c8++) {
            f32x4 a[RT], b[CT];
#pragma unroll
            for (int ct = 0; ct < CT; ct++)
                b[ct] = *(const f32x4*)(wtap + (size_t)(c8 >> 1) * wchunk_stride + (size_t)ct * 32 * 16 + 8 * (c8 & 1));
#pragma unroll
            for (int rt = 0; rt < RT; rt++) a[rt] = *(const f32x4*)&lds[aoff[rt] + c8 * 8];
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int rt = 0; rt < RT; rt++)
#pragma unroll
                    for (int ct = 0; ct < CT; ct++)
                        acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[rt][t], b[ct][t], acc[rt][ct], 0, 0, 0);
        }
    }

    // ---- epilogue: bias (+ residual) (+ ReLU); C/D map: col = lane&31, row = (r&3) + 8(r>>2) + 4(lane>>5) ----
#pragma unroll
    for (int rt = 0; rt < RT; rt++) {
#pragma unroll
        for (int ct = 0; ct < CT; ct++) {
            const int col = col0 + ct * 32 + i;
            const float bv = bias[col];
#pragma unroll
            for (int r = 0; r < 16; r++) {
                int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                int m = m0 + (wr * RT + rt) * 32 + row;
                if (m < M && col < cout_valid) {
                    float v = acc[rt][ct][r] + bv;
                    size_t o = (size_t)m * out_stride + col;
                    if (res) v += res[o];
                    if (relu) v = fmaxf(v, 0.0f);
                    out[o] = v;
                }
            }
        }
    }
}



// ------------------------------------------------------------------------------------------------
// Whole-positions variant: a workgroup owns PW complete positions (PW·N² rows, e.g. 16 positions = 400
// rows = 25 row tiles on 5×5; 4 positions = 144 rows = 9 tiles on 6×6) and CTW 16-wide channel tiles, so
// the grid is an exact multiple of the 256 CUs (no tail wave) and no position is staged twice.
// v_mfma_f32_16x16x4_f32 with the WEIGHTS as the A operand and the activations as B: the accumulator then
// holds, per lane, 4 consecutive output channels of one row → 16-byte epilogue loads/stores.
// Waves: wave = (row group rg, channel tile ct); each wave owns RTW row tiles × 1 channel tile.
// ------------------------------------------------------------------------------------------------
template <int RTW, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void k_conv_pos(const float* __restrict__ in, const float* __restrict__ Wp,
                                                          const float* __restrict__ bias, const float* __restrict__ res,
                                                          float* __restrict__ out, int B, int n, int Cpad, int CoutP,
                                                          int out_stride, int cout_valid, int relu, int PW, int CTW) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    f32x4* lds4 = (f32x4*)lds;  // every access below is a whole 16-byte slot → ds_read_b128 / ds_write_b128
    const int tid = threadIdx.x;
    const int nsq = n * n;
    const int LS4 = (Cpad + LDS_PAD16) >> 2;
    const int pos0 = blockIdx.x * PW;
    const int npos = min(PW, B - pos0);
    const int rows = npos * nsq;
    {
        const int vpr = Cpad >> 2;
        const f32x4* src = (const f32x4*)(in + (size_t)pos0 * nsq * Cpad);
        const int total = rows * vpr;
        // all of a thread's loads are issued before the first LDS write (8 in flight per lane)
        constexpr int UNR = 8;
        for (int base = 0; base < total; base += NWAVES * 64 * UNR) {
            f32x4 tmp[UNR];
#pragma unroll
            for (int u = 0; u < UNR; u++) {
                int idx = base + u * NWAVES * 64 + tid;
                tmp[u] = idx < total ? src[idx] : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            }
#pragma unroll
            for (int u = 0; u < UNR; u++) {
                int idx = base + u * NWAVES * 64 + tid;
                if (idx < total) {
                    int r = idx / vpr, v = idx - r * vpr;
                    lds4[r * LS4 + v] = tmp[u];
                }
            }
        }
        for (int idx = tid; idx < LS4; idx += NWAVES * 64) lds4[rows * LS4 + idx] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
    __syncthreads();

    const int wave = tid >> 6, lane = tid & 63;
    const int ct = wave % CTW, rg = wave / CTW;
    const int r16 = lane & 15, q = lane >> 4;
    const int zero4 = rows * LS4 + q;

    int base4[RTW], pyx[RTW];
#pragma unroll
    for (int j = 0; j < RTW; j++) {
        int rho = (rg * RTW + j) * 16 + r16;
        bool valid = rho < rows;
        int rr = valid ? rho : 0;
        int p = rr / nsq;
        int sq = rr - p * nsq;
        int y = sq / n, x = sq - y * n;
        pyx[j] = valid ? (y | (x << 8)) : 0x7f7f;  // invalid rows: every tap falls off the board
        base4[j] = rr * LS4 + q;
    }

    f32x4 acc[RTW];
#pragma unroll
    for (int j = 0; j < RTW; j++) acc[j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    const int ch0 = (blockIdx.y * CTW + ct) * 16;
    const int chunks = Cpad >> 4;
    const int total_chunks = 9 * chunks;
    // weights: Wp[k/16][ch][16] = 4 slots of 16 B per (chunk, channel); this lane reads slot q of channel ch0 + r16
    const f32x4* wp = (const f32x4*)Wp + ((size_t)(ch0 + r16) * 4 + q);
    const size_t wstride4 = (size_t)CoutP * 4;

    f32x4 w_cur = wp[0];
    int kk = 0;
    for (int tap = 0; tap < 9; tap++) {
        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
        int aoff[RTW];
#pragma unroll
        for (int j = 0; j < RTW; j++) {
            int yy = (pyx[j] & 0xff) + dy, xx = (pyx[j] >> 8) + dx;
            bool ok = yy >= 0 && yy < n && xx >= 0 && xx < n;
            aoff[j] = ok ? base4[j] + (dy * n + dx) * LS4 : zero4;
        }
        f32x4 a_cur[RTW];
#pragma unroll
        for (int j = 0; j < RTW; j++) a_cur[j] = lds4[aoff[j]];
        // every tap's products in a chain of their own, added to acc when the tap is complete (conv_mainloop_halo's SPLIT: the same
        // chains in the same order → the same bits as k_conv_halo)
        f32x4 part[RTW];
#pragma unroll
        for (int j = 0; j < RTW; j++) part[j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        for (int kc = 0; kc < chunks; kc++) {
            // software pipeline: the next chunk's weights (L2) and activations (LDS) are in flight while
            // this chunk's 4·RTW MFMAs issue
            const int kkn = kk + 1 < total_chunks ? kk + 1 : kk;
            const f32x4 w_nxt = wp[(size_t)kkn * wstride4];
            const int kn = kc + 1 < chunks ? kc + 1 : kc;
            f32x4 a_nxt[RTW];
#pragma unroll
            for (int j = 0; j < RTW; j++) a_nxt[j] = lds4[aoff[j] + kn * 4];
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int j = 0; j < RTW; j++) part[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w_cur[t], a_cur[j][t], part[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < RTW; j++) a_cur[j] = a_nxt[j];
            w_cur = w_nxt;
            kk++;
        }
#pragma unroll
        for (int j = 0; j < RTW; j++) acc[j] += part[j];
    }

    // epilogue: lane holds out[row = tile*16 + (lane&15)][ch0 + 4q .. 4q+3]
    const int ch = ch0 + 4 * q;
    const f32x4 bv = *(const f32x4*)&bias[ch];
#pragma unroll
    for (int j = 0; j < RTW; j++) {
        int rho = (rg * RTW + j) * 16 + r16;
        if (rho < rows && ch < cout_valid) {
            size_t o = ((size_t)pos0 * nsq + rho) * out_stride + ch;
            f32x4 v = acc[j] + bv;
            if (res) v += *(const f32x4*)&res[o];
            if (relu) { v[0] = fmaxf(v[0], 0.0f); v[1] = fmaxf(v[1], 0.0f); v[2] = fmaxf(v[2], 0.0f); v[3] = fmaxf(v[3], 0.0f); }
            if (ch + 3 < cout_valid) *(f32x4*)&out[o] = v;
            else for (int t = 0; t < 4; t++) if (ch + t < cout_valid) out[o + t] = v[t];
        }
    }
}


// ------------------------------------------------------------------------------------------------
// Fused residual tower: conv0 + R × (conv1, conv2 + skip) in ONE launch.  A workgroup keeps its PW
// positions in LDS for the whole tower: each layer's MFMA loop reads the padded NHWC image of the
// previous layer from LDS, the epilogue (bias, ReLU, skip) runs on the accumulators, and — behind a
// barrier — the wave writes its 16-channel slice straight back into the same LDS image for the next
// layer.  The skip connection never leaves registers (each wave keeps the block input of exactly the
// tiles it produces).  Only the input planes are read from HBM and only the final activations are
// written (for the policy / value heads); per layer the only global traffic is the L2-resident weights.
// ------------------------------------------------------------------------------------------------
// ---- constant input planes as a per-position bias (TowerParams.cb; states entry of the fused towers; tower_cb.cuh) ----
// Stages what layer 0 needs for the positions of one workgroup: the 26 / 28 BOARD planes of every square as a plain image of
// 32 channels per row (last chunk permuted for cb_last_t = 3), and the table PB[position][border class][F].
template <int NWAVES>
__device__ __forceinline__ void tower_stage_states_cb(f32x4* lds4, f32x4* pb4, const uint8_t* states, int pos0, int npos, int n,
                                                      int LS4, const TowerParams& T) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const Geom geo = make_geom(n);
    const int nsq = n * n;
    // one wave per position at a time, lane = square; a wave's positions are requested two at a time (C2: 16 positions, 8 waves —
    // one memory round trip instead of two)
    auto stage_one = [&](int p, const WRaw& raw) {
        WState ws;
        ws_unpack(ws, raw, geo);
        const float fcd = fcd_value(ws, geo);
        const RowMask m = ws_row_mask(ws, geo);
        if (lane < nsq) {
            f32x4* row = lds4 + (size_t)(p * nsq + lane) * LS4;
            f32x4 qd[8];
            tower_cb_board_quads(m, n, qd);
#pragma unroll
            for (int k = 0; k < 4; k++) row[k] = qd[k];
            const f32x4 lc[4] = {qd[4], qd[5], qd[6], qd[7]};
            conv_last_chunk_store(row + 4, lc, T.cb_last_t);
        }
        tower_cb_table(ws, fcd, n, p, T.F >> 2, (const f32x4*)T.cplane_sums, (const f32x4*)T.b[0], pb4);
    };
    for (int p = wave; p < npos; p += 2 * NWAVES) {
        const int p1 = p + NWAVES;
        const WRaw r0 = ws_load_raw(states + (size_t)(pos0 + p) * geo.bytes, geo);
        const WRaw r1 = ws_load_raw(states + (size_t)(pos0 + (p1 < npos ? p1 : p)) * geo.bytes, geo);
        stage_one(p, r0);
        if (p1 < npos) stage_one(p1, r1);
    }
}

// FROM_STATES: `in` points at packed game states and the planes are encoded straight into the LDS image
// (game_repr fused into the tower: the f32 planes never touch HBM).
// CB (with FROM_STATES): layer 0 over the board planes only, the constant planes as the per-position bias PB
// (tower_stage_states_cb) — CH0 = 2 then.
template <int RTW, int NWAVES, int CH0, int CH, bool FROM_STATES, bool CB = false>
__global__ __launch_bounds__(NWAVES * 64) void k_tower(const float* __restrict__ in, TowerParams T, float* __restrict__ out,
                                                       int B, int n, int PW, int CTW) {
    static_assert(!CB || FROM_STATES, "the constant-plane bias needs the packed states");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    f32x4* lds4 = (f32x4*)lds;
    const int tid = threadIdx.x;
    const int nsq = n * n;
    const int pos0 = blockIdx.x * PW;
    const int npos = min(PW, B - pos0);
    const int rows = npos * nsq;
    const int F = T.F;
    const int wave = tid >> 6, lane = tid & 63;
    const int ct = wave % CTW, rg = wave / CTW;
    const int r16 = lane & 15, q = lane >> 4;
    const int ch0 = ct * 16;

    // ---- stage the input planes (row pitch cin_pad + 4) ----
    int Cpad = CB ? T.cb_cin_pad : T.cin_pad;
    int LS4 = (Cpad + LDS_PAD16) >> 2;
    f32x4* pb4 = lds4 + (size_t)(PW * nsq + 1) * LS4;  // CB: PB[position][class][F] behind the image and its zero row
    if (CB) {
        tower_stage_states_cb<NWAVES>(lds4, pb4, (const uint8_t*)in, pos0, npos, n, LS4, T);
        for (int idx = tid; idx < LS4; idx += NWAVES * 64) lds4[rows * LS4 + idx] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    } else if (FROM_STATES) {
        const Geom geo = make_geom(n);
        const uint8_t* states = (const uint8_t*)in;
        const int C = input_channels(n);
        for (int p = wave; p < npos; p += NWAVES) {  // one wave encodes one position at a time, lane = square
            WState ws;
            ws_load(ws, states + (size_t)(pos0 + p) * geo.bytes, geo);
            const float fcd = fcd_value(ws, geo);
            const RowMask m = ws_row_mask(ws, geo);
            if (lane < nsq) {
                f32x4* row = lds4 + (size_t)(p * nsq + lane) * LS4;
                const int kl = (Cpad >> 2) - 4;  // first quad of the last 16-channel chunk
                for (int k = 0; k < kl; k++) {
                    float4 v = row_mask_value(m, k, C, fcd);
                    row[k] = f32x4{v.x, v.y, v.z, v.w};
                }
                f32x4 lc[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    float4 v = row_mask_value(m, kl + k, C, fcd);
                    lc[k] = f32x4{v.x, v.y, v.z, v.w};
                }
                conv_last_chunk_store(row + kl, lc, T.cin_last_t);
            }
        }
        for (int idx = tid; idx < LS4; idx += NWAVES * 64) lds4[rows * LS4 + idx] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    } else {
        const int vpr = Cpad >> 2;
        const f32x4* src = (const f32x4*)(in + (size_t)pos0 * nsq * Cpad);
        const int total = rows * vpr;
        for (int idx = tid; idx < total; idx += NWAVES * 64) {
            int r = idx / vpr, v = idx - r * vpr;
            if (v < vpr - 4) lds4[r * LS4 + v] = src[idx];
        }
        for (int r = tid; r < rows; r += NWAVES * 64) {  // the last chunk of every row, permuted like the weights
            f32x4 lc[4];
#pragma unroll
            for (int k = 0; k < 4; k++) lc[k] = src[(size_t)r * vpr + vpr - 4 + k];
            conv_last_chunk_store(lds4 + (size_t)r * LS4 + vpr - 4, lc, T.cin_last_t);
        }
        for (int idx = tid; idx < LS4; idx += NWAVES * 64) lds4[rows * LS4 + idx] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
    __syncthreads();

    // Row tiles of a full workgroup are dealt to the NWAVES / CTW row groups as evenly as they go (25 = 13 + 12 with two
    // groups, 7 + 6 + 6 + 6 with four): the first `rem` groups own RTW tiles, the others RTW - 1 and run the loop
    // specialised for that count instead of issuing a whole tile of zero MFMAs (wave-uniform branch).  The waves of one
    // channel tile (wave % CTW) land on one SIMD, so every SIMD carries all the row tiles whatever the split.
    const int NRG = NWAVES / CTW;
    const int ntiles = (PW * nsq + 15) >> 4;
    const int tbase = ntiles / NRG, trem = ntiles - tbase * NRG;
    const int my_tiles = tbase + (rg < trem ? 1 : 0);
    const int rho0 = (rg * tbase + min(rg, trem)) * 16 + r16;
    const bool short_group = my_tiles < RTW;

    // Skip connection without any storage of its own: after conv1 of a block every wave reads the block input X
    // of exactly the tiles it owns back from the LDS image (just before it overwrites them with conv1's output)
    // and uses it as the initial value of conv2's accumulators.
    f32x4 acc[RTW];
#pragma unroll
    for (int j = 0; j < RTW; j++) acc[j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    int vmask[RTW];  // board geometry is the same for every layer
    conv_tap_masks<RTW>(rows, n, nsq, rho0, vmask);
    if (short_group) vmask[RTW - 1] = 0;  // that tile belongs to the next row group

    for (int layer = 0; layer < T.nlayers; layer++) {
        const f32x4* wp = (const f32x4*)(CB && layer == 0 ? T.w0_board : T.w[layer]) + ((size_t)(ch0 + r16) * 4 + q);
        const int last_t0 = CB ? T.cb_last_t : T.cin_last_t;
        TG_STAMP(layer, 0);
        if (RTW > 1 && short_group) {
            f32x4 (&acs)[RTW - 1] = *reinterpret_cast<f32x4 (*)[RTW - 1]>(&acc[0]);
            if (layer == 0) conv_mainloop<RTW - 1, CH0>(lds4, wp, (size_t)F * 4, LS4, rows, n, rho0, q, vmask, acs, last_t0);
            else conv_mainloop<RTW - 1, CH>(lds4, wp, (size_t)F * 4, LS4, rows, n, rho0, q, vmask, acs);
        } else {
            if (layer == 0) conv_mainloop<RTW, CH0>(lds4, wp, (size_t)F * 4, LS4, rows, n, rho0, q, vmask, acc, last_t0);
            else conv_mainloop<RTW, CH>(lds4, wp, (size_t)F * 4, LS4, rows, n, rho0, q, vmask, acc);
        }
        TG_STAMP(layer, 1);
        // ---- epilogue on the accumulators: lane holds out[row][ch0 + 4q .. 4q+3] ----
        const f32x4 bv = *(const f32x4*)&T.b[layer][ch0 + 4 * q];
#pragma unroll
        for (int j = 0; j < RTW; j++) {
            f32x4 v = acc[j] + ((CB && layer == 0) ? pb4[tower_cb_index(rho0 + j * 16, rows, n, nsq, F >> 2, (ch0 >> 2) + q)] : bv);
            v[0] = fmaxf(v[0], 0.0f); v[1] = fmaxf(v[1], 0.0f); v[2] = fmaxf(v[2], 0.0f); v[3] = fmaxf(v[3], 0.0f);
            acc[j] = v;
        }
        if (layer + 1 == T.nlayers) {
#pragma unroll
            for (int j = 0; j < RTW; j++)
                if (j < my_tiles && rho0 + j * 16 < rows) {
                    const int rho = rho0 + j * 16;
                    if (T.frag_out) {  // (tile of 16 positions, chunk = square·F/16 + channel tile) → one KB, lane (position, q)
                        const int p = pos0 + rho / nsq, sq = rho % nsq;
                        ((f32x4*)out)[((size_t)(p >> 4) * (nsq * (F >> 4)) + sq * (F >> 4) + ct) * 64 + (p & 15) * 4 + q] = acc[j];
                    } else *(f32x4*)&out[((size_t)pos0 * nsq + rho) * F + ch0 + 4 * q] = acc[j];
                }
            break;
        }
        TG_STAMP(layer, 2);
        __syncthreads();  // every wave has finished reading the previous image
        TG_STAMP(layer, 3);
        const int LS4n = (F + LDS_PAD16) >> 2;
        const bool conv1 = (layer & 1) == 1;  // next layer is conv2 of the same block: it starts from the block input
        Cpad = F;
        LS4 = LS4n;
        // tile by tile: read the own tile of X (the initial value of conv2's accumulator), overwrite it with this layer's
        // output — no second register set (at layer 0 the pitch changes: conv1 is false there, nothing is read back)
#pragma unroll
        for (int j = 0; j < RTW; j++) {
            f32x4 x0 = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if (j < my_tiles && rho0 + j * 16 < rows) {
                const int at = (rho0 + j * 16) * LS4 + (ch0 >> 2) + q;
                if (conv1) x0 = lds4[at];
                lds4[at] = acc[j];
            }
            acc[j] = x0;
        }
        if (layer == 0)
            for (int idx = tid; idx < LS4; idx += NWAVES * 64) lds4[rows * LS4 + idx] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        TG_STAMP(layer, 4);
        __syncthreads();
        TG_STAMP(layer, 5);
    }
}

// ------------------------------------------------------------------------------------------------
// The fused tower for SMALL batches of wide networks (round 6): k_tower gives every position to one workgroup — at the reference's
// own constants (32 lock-step games → 32 leaves per forward, Net6 = 16 blocks × 128 filters; train/src/self_play.rs:94,
// alpha-tak/src/model/net6.rs:16-17) that is 32 busy CUs of 256, each issuing 8 channel tiles × 3 row tiles × 288 MFMAs per layer
// (23 µs per layer, 882 µs per forward).  Here a position is SPLIT over G = F / 16 workgroups by output channel tile: workgroup
// (position p, group g) holds the whole input image of p in LDS and computes CTW channel tiles × NRT row tiles, one (row tile, channel
// tile) pair per wave = ONE chain of 9·16·CH/4 MFMAs — the shortest critical path the arithmetic allows (a chain cannot be cut: every
// output element is accumulated over k in k_tower's order, so the results are bit-identical).  Between two layers the G workgroups of a
// position exchange their 16·CTW-channel slices through global memory (two buffers used in turn; L2-resident: n²·F floats per
// position) and meet at a counter per position: slice stored → release → flag += 1; wait for flag = G·(layer + 1) → acquire → stage the
// next image.  Workgroup ids come in blocks of 8 positions × G groups with the position's low bits in the id's low bits, so the siblings
// of a position sit on one XCD (id mod 8) and the exchange stays in that XCD's L2 (SAME_L2, below; without that guarantee agent-scope
// fences make it correct wherever they sit).  Weights: a wave with one tile issues 4 MFMAs per 16-k chunk — far less than an L2 round
// trip — so they are fetched a whole TAP ahead (CH quads per lane, two sets) instead of two chunks ahead.  What bounds a layer is the
// chain itself: 288 DEPENDENT MFMAs at 46 cycles each (6.0 µs of a layer's 8.2; scripts/probes/split_stamps.hip).
// A workgroup never waits for more than its own G − 1 siblings, all of one launch whose grid (≤ 512 three-wave workgroups on 6×6, ≤ 1024
// two-wave ones on 5×5) is co-resident, and whose ids put a position's siblings within 8·G of each other in the dispatch order; the wait
// is bounded all the same: after SPLIT_SPIN_LIMIT polls it raises T.split_err (→ TG_ERR_HIP on the host) and the waits stop.
// ------------------------------------------------------------------------------------------------
constexpr unsigned SPLIT_SPIN_LIMIT = 1u << 21;
constexpr int SPLIT_FLAG_STRIDE = 32;  // u32 words between two positions' counters (kernels.h: TOWER_SPLIT_CTL_WORDS)

// one (row tile, channel tile) pair: the products of conv_mainloop<1, CHL> in the same order (tap, chunk, t); weights one tap ahead,
// the tap's activation quads one tap ahead as well.  w0 = the first tap's CHL weight quads, requested by the caller (conv_tile_first_weights)
// as early as it likes — they depend on nothing but the layer.  TAPCHAIN: every tap's products in a chain of their own that is added
// to acc when the tap is complete (k_conv_pos / conv_mainloop_halo<SPLIT>: the stand-alone layers' order).
template <int CHL>
__device__ __forceinline__ void conv_tile_first_weights(const f32x4* __restrict__ wp, size_t wstride4, f32x4 (&w0)[CHL]) {
#pragma unroll
    for (int kc = 0; kc < CHL; kc++) w0[kc] = wp[(size_t)kc * wstride4];
}
template <int CHL, int LAST_T = 4, bool TAPCHAIN = false>
__device__ __forceinline__ void conv_mainloop_tile(const f32x4* __restrict__ lds4, const f32x4* __restrict__ wp, size_t wstride4, int LS4,
                                                   int rows, int n, int rho, int q, int vmask, f32x4& acc, const f32x4 (&w0)[CHL]) {
    // LAST_T < 4 (layer 0): the last chunk's real channels fill MFMAs t < LAST_T, the others would multiply padding (conv_mainloop)
    const int zero4 = rows * LS4 + q, base = rho * LS4 + q;
    auto tap_addr = [&](int tap) { return ((vmask >> tap) & 1) ? base + ((tap / 3 - 1) * n + (tap % 3 - 1)) * LS4 : zero4; };
    f32x4 w[2][CHL], a[2][CHL];
#pragma unroll
    for (int kc = 0; kc < CHL; kc++) w[0][kc] = w0[kc];
    {
        const int a0 = tap_addr(0);
#pragma unroll
        for (int kc = 0; kc < CHL; kc++) a[0][kc] = lds4[a0 + kc * 4];
    }
#pragma unroll
    for (int tap = 0; tap < 9; tap++) {
        const int cur = tap & 1, nxt = cur ^ 1;
        if (tap + 1 < 9) {
#pragma unroll
            for (int kc = 0; kc < CHL; kc++) w[nxt][kc] = wp[(size_t)((tap + 1) * CHL + kc) * wstride4];
            const int a1 = tap_addr(tap + 1);
#pragma unroll
            for (int kc = 0; kc < CHL; kc++) a[nxt][kc] = lds4[a1 + kc * 4];
        }
        // (left alone hipcc sinks every load next to its use: the chain would then wait out an L2 round trip per chunk)
        __builtin_amdgcn_sched_barrier(0);
        f32x4 part = TAPCHAIN ? f32x4{0.0f, 0.0f, 0.0f, 0.0f} : acc;
#pragma unroll
        for (int kc = 0; kc < CHL; kc++)
#pragma unroll
            for (int t = 0; t < 4; t++)
                if (kc + 1 < CHL || t < LAST_T) part = __builtin_amdgcn_mfma_f32_16x16x4f32(w[cur][kc][t], a[cur][kc][t], part, 0, 0, 0);
        if (TAPCHAIN) acc += part;
        else acc = part;
        __builtin_amdgcn_sched_barrier(0);
    }
}

// SAME_L2: the launcher has verified on this device that workgroup id i runs on XCD i mod 8 (k_xcc_probe), so a position's siblings
// share one L2 and the exchange needs no agent-scope fences — those write back and INVALIDATE the XCD's whole L2 (buffer_wbl2 sc1 /
// buffer_inv sc1: 256 workgroups × every layer), after which every weight load of the next layer misses it (tower 359 against 252 µs).
// What it needs instead: the slice's stores complete (the vector L1 writes through: s_waitcnt vmcnt(0) = in L2), the counter as an L2
// atomic polled at device scope, and the image staged with device-scope loads (sc1: past this CU's L1).  Without the guarantee: the
// agent-scope fences, correct wherever the siblings sit.  (Measured and not kept, profiles/r06_k_split_exchange_variants.txt: the data as
// its own flag — three sentinel-filled buffers polled directly, no counter: 242 µs, i.e. the exchange is the siblings' skew, not the
// protocol's round trips.)
template <int NRT, int CTW, int CH, bool SAME_L2>
__global__ __launch_bounds__(NRT * CTW * 64) void k_tower_split(const uint8_t* __restrict__ states, TowerParams T, float* __restrict__ out,
                                                                float* __restrict__ scratch, int B, int n) {
    constexpr int NW = NRT * CTW, F = 16 * CH, F4 = F / 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    f32x4* lds4 = (f32x4*)lds;
    const int tid = threadIdx.x, nsq = n * n, rows = nsq;
    // workgroup id → (position, channel group): ids come in blocks of 8·G — 8 consecutive positions × their G groups — with the
    // position's low three bits in the id's low three bits: the G siblings of a position are dispatched within 8·G consecutive ids
    // (they never wait for a workgroup far behind them in the dispatch order) and land on one XCD (id mod 8), so their exchange
    // stays in that XCD's L2
    constexpr int G = (CH / CTW);
    const int blk = blockIdx.x / (8 * G), rem = blockIdx.x - blk * (8 * G);
    const int p = blk * 8 + (rem & 7), g = rem >> 3;
    if (p >= B) return;
    // the guarantee the fast exchange rests on — a position's siblings on ONE XCD — is checked by every workgroup of every launch:
    // group 0 posts its XCD in the position's counter line, the others compare at the first meeting (the dispatcher's round robin may
    // start anywhere, so the id alone does not name the XCD; ids that agree mod 8 share one — k_xcc_probe)
    unsigned my_xcc = 0;
    if (SAME_L2 && tid == 0) {
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(my_xcc));
        my_xcc = (my_xcc & 15u) + 1u;
    }
    const int wave = tid >> 6, lane = tid & 63;
    const int ct = wave % CTW, rt = wave / CTW;
    const int r16 = lane & 15, q = lane >> 4;
    const int ch0 = (g * CTW + ct) * 16;
    const int rho = rt * 16 + r16;  // this lane's row (square) of the position; ≥ rows: padding of the last row tile
    unsigned* flag = T.split_flags + (size_t)p * SPLIT_FLAG_STRIDE;  // a 128-byte line per position: its 8 pollers contend with nobody else

    // ---- layer 0's image: the board planes of position p, the per-class bias table behind it (as k_tower, CB) ----
    int LS4 = (T.cb_cin_pad + LDS_PAD16) >> 2;
    f32x4* pb4 = lds4 + (size_t)(nsq + 1) * LS4;
    tower_stage_states_cb<NW>(lds4, pb4, states, p, 1, n, LS4, T);
    for (int idx = tid; idx < LS4; idx += NW * 64) lds4[rows * LS4 + idx] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    __syncthreads();

    int vmask[1];
    conv_tap_masks<1>(rows, n, nsq, rho, vmask);
    f32x4 acc = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    bool failed = false;  // (thread 0) a wait ran into its bound: no further waits
    const size_t wlane = (size_t)(ch0 + r16) * 4 + q;
    f32x4 wf[CH];  // the first tap's weights of the NEXT layer: requested before the wait for the siblings, there when it ends
    for (int layer = 0; layer < T.nlayers; layer++) {
        const f32x4* wp = (const f32x4*)(layer == 0 ? T.w0_board : T.w[layer]) + wlane;
        const f32x4 bv = *(const f32x4*)&T.b[layer][ch0 + 4 * q];
        TG_STAMP(layer, 0);
        if (layer == 0) {  // (cb_last_t = 3, one 32-channel chunk pair: the launcher checks)
            f32x4 wf0[2];
            conv_tile_first_weights<2>(wp, (size_t)F * 4, wf0);
            conv_mainloop_tile<2, 3>(lds4, wp, (size_t)F * 4, LS4, rows, n, rho, q, vmask[0], acc, wf0);
        } else conv_mainloop_tile<CH>(lds4, wp, (size_t)F * 4, LS4, rows, n, rho, q, vmask[0], acc, wf);
        TG_STAMP(layer, 1);
        f32x4 v = acc + (layer == 0 ? pb4[tower_cb_index(rho, rows, n, nsq, F4, (ch0 >> 2) + q)] : bv);
        v[0] = fmaxf(v[0], 0.0f); v[1] = fmaxf(v[1], 0.0f); v[2] = fmaxf(v[2], 0.0f); v[3] = fmaxf(v[3], 0.0f);
        if (layer + 1 == T.nlayers) {
            if (rho < rows) {
                if (T.frag_out) ((f32x4*)out)[((size_t)(p >> 4) * (nsq * CH) + rho * CH + (ch0 >> 4)) * 64 + (p & 15) * 4 + q] = v;
                else *(f32x4*)&out[((size_t)p * nsq + rho) * F + ch0 + 4 * q] = v;
            }
            break;
        }
        // the block input of conv2's accumulator: this wave's own slice of the image conv1 has just read (k_tower's skip path)
        f32x4 x0 = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        if ((layer & 1) == 1 && rho < rows) x0 = lds4[rho * LS4 + (ch0 >> 2) + q];
        acc = x0;
        // two exchange buffers used in turn (`out` is written by the last layer only: in the FC's fragment order a position's output
        // lies across the rows of 15 others)
        float* xbuf = scratch + (size_t)(layer & 1) * TOWER_SPLIT_MAX_BATCH * nsq * F;
        if (rho < rows) *(f32x4*)&xbuf[((size_t)p * nsq + rho) * F + ch0 + 4 * q] = v;
        if (SAME_L2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();  // every wave's slice is in L2 (SAME_L2) / on its way and every wave has finished reading the image
        TG_STAMP(layer, 2);
        conv_tile_first_weights<CH>((const f32x4*)T.w[layer + 1] + wlane, (size_t)F * 4, wf);
        if (tid == 0) {
            if (!SAME_L2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            if (SAME_L2 && layer == 0 && g == 0) __hip_atomic_store(flag + 1, my_xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)G * (unsigned)(layer + 1);
            unsigned spins = 0;
            // (device scope: a group-scope load — sc0 — may hit this CU's L1 and then never sees the siblings' atomics: measured, the
            // bounded wait fired)
            auto poll = [&]() -> unsigned { return __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
            while (!failed && poll() < target) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > SPLIT_SPIN_LIMIT) { __hip_atomic_store(T.split_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); failed = true; }
                else if ((spins & 4095u) == 0 && __hip_atomic_load(T.split_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) failed = true;
            }
            if (!SAME_L2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            if (SAME_L2 && layer == 0 && !failed && __hip_atomic_load(flag + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != my_xcc)
                __hip_atomic_store(T.split_err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        TG_STAMP(layer, 3);
        // ---- the next layer's image: all F channels of position p, pitch F + 8 floats ----
        // SAME_L2: DEVICE-scope loads (sc1) — past this CU's L1, which may hold the buffer's lines of two layers ago, to the L2 the siblings'
        // stores went to.  (`buffer_inv sc0` + plain loads was 5 % faster and passed every test, but a counter polled that way saw the
        // siblings' atomics only after a long delay: the L1 was being emptied by the weight stream, not by the invalidate.)
        LS4 = (F + LDS_PAD16) >> 2;
        const f32x4* src = (const f32x4*)(xbuf + (size_t)p * nsq * F);
        const int total = nsq * F4;
        constexpr int UNR = 8;
        for (int base = 0; base < total; base += NW * 64 * UNR) {
            f32x4 tmp[UNR];
#pragma unroll
            for (int u = 0; u < UNR; u++) {
                const int idx = base + u * NW * 64 + tid;
                const f32x4* a = src + (idx < total ? idx : total - 1);
                if (SAME_L2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(tmp[u]) : "v"(a) : "memory");
                else tmp[u] = *a;
            }
            if (SAME_L2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int u = 0; u < UNR; u++) {
                const int idx = base + u * NW * 64 + tid;
                if (idx < total) lds4[(idx / F4) * LS4 + idx % F4] = tmp[u];
            }
        }
        if (layer == 0)
            for (int idx = tid; idx < LS4; idx += NW * 64) lds4[rows * LS4 + idx] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        __syncthreads();
        TG_STAMP(layer, 4);
    }
    // the counter returns to zero with the launch: the last of the position's G workgroups to finish resets it
    __syncthreads();
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1u == (unsigned)G * (unsigned)T.nlayers) {
            __hip_atomic_store(flag + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(flag, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The fused tower for full batches: as k_tower, but from layer 1 on the resident image is the HALO image of
// conv_mainloop.cuh (zero cells between board rows and between positions), so the 3×3 taps are immediates and the main
// loop is nothing but ds_read_b128 / MFMA / one weight load per chunk.  Layer 0 (other pitch, once per launch) runs
// the masked loop on the plain image and writes its output straight into halo cells.  From layer 1 on the squares
// are dealt to (row tile, lane) slots by T.slotmap (tower_halo_slotmap): a permutation of the GEMM's M dimension,
// invisible in the results, that keeps every ds_read_b128 of the loop off its neighbours' banks.
// Per-element arithmetic (taps, chunks, k-steps, bias, ReLU, skip) in k_tower's order → identical bits.
// ------------------------------------------------------------------------------------------------
template <int RTW, int NWAVES, int CH0, int CH, int NB, bool FROM_STATES, bool CB = false>
__global__ __launch_bounds__(NWAVES * 64) void k_tower_halo(const float* __restrict__ in, TowerParams T, float* __restrict__ out,
                                                            int B, int PW, int CTW) {
    static_assert(!CB || FROM_STATES, "the constant-plane bias needs the packed states");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    f32x4* lds4 = (f32x4*)lds;
    TG_STAMP(0, 6);  // kernel start (diagnostic build only)
    constexpr int n = NB, nsq = NB * NB, RS = NB + 1, LEAD = NB + 2, F = 16 * CH, P4 = 4 * CH + 1;
    const int PS = T.halo_ps;
    const int tid = threadIdx.x;
    const int pos0 = blockIdx.x * PW;
    const int npos = min(PW, B - pos0);
    const int rows = npos * nsq;
    const int wave = tid >> 6, lane = tid & 63;
    const int ct = wave % CTW, rg = wave / CTW;
    const int r16 = lane & 15, q = lane >> 4;
    const int ch0 = ct * 16;

    // ---- stage the input planes: plain image, row pitch cin_pad + 8 floats, one zero row behind it ----
    const int Cpad = CB ? T.cb_cin_pad : T.cin_pad;
    const int LS4 = (Cpad + LDS_PAD16) >> 2;
    f32x4* pb4 = lds4 + (size_t)(PW * nsq + 1) * LS4;  // CB: PB[position][class][F] behind the image and its zero row
    if (CB) {
        tower_stage_states_cb<NWAVES>(lds4, pb4, (const uint8_t*)in, pos0, npos, n, LS4, T);
    } else if (FROM_STATES) {
        const Geom geo = make_geom(n);
        const uint8_t* states = (const uint8_t*)in;
        const int C = input_channels(n);
        for (int p = wave; p < npos; p += NWAVES) {  // one wave encodes one position at a time, lane = square
            WState ws;
            ws_load(ws, states + (size_t)(pos0 + p) * geo.bytes, geo);
            const float fcd = fcd_value(ws, geo);
            const RowMask m = ws_row_mask(ws, geo);
            if (lane < nsq) {
                f32x4* row = lds4 + (size_t)(p * nsq + lane) * LS4;
                const int kl = (Cpad >> 2) - 4;  // first quad of the last 16-channel chunk
                for (int k = 0; k < kl; k++) {
                    float4 v = row_mask_value(m, k, C, fcd);
                    row[k] = f32x4{v.x, v.y, v.z, v.w};
                }
                f32x4 lc[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    float4 v = row_mask_value(m, kl + k, C, fcd);
                    lc[k] = f32x4{v.x, v.y, v.z, v.w};
                }
                conv_last_chunk_store(row + kl, lc, T.cin_last_t);
            }
        }
    } else {
        const int vpr = Cpad >> 2;
        const f32x4* src = (const f32x4*)(in + (size_t)pos0 * nsq * Cpad);
        const int total = rows * vpr;
        for (int idx = tid; idx < total; idx += NWAVES * 64) {
            int r = idx / vpr, v = idx - r * vpr;
            if (v < vpr - 4) lds4[r * LS4 + v] = src[idx];
        }
        for (int r = tid; r < rows; r += NWAVES * 64) {  // the last chunk of every row, permuted like the weights
            f32x4 lc[4];
#pragma unroll
            for (int k = 0; k < 4; k++) lc[k] = src[(size_t)r * vpr + vpr - 4 + k];
            conv_last_chunk_store(lds4 + (size_t)r * LS4 + vpr - 4, lc, T.cin_last_t);
        }
    }
    for (int idx = tid; idx < LS4; idx += NWAVES * 64) lds4[rows * LS4 + idx] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    __syncthreads();

    // row tiles dealt to the row groups as in k_tower
    const int NRG = NWAVES / CTW;
    const int ntiles = (PW * nsq + 15) >> 4;
    const int tbase = ntiles / NRG, trem = ntiles - tbase * NRG;
    const int my_tiles = tbase + (rg < trem ? 1 : 0);
    const int tile0 = rg * tbase + min(rg, trem);
    const bool short_group = my_tiles < RTW;

    f32x4 acc[RTW];
#pragma unroll
    for (int j = 0; j < RTW; j++) acc[j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    const uint32_t wlane = (uint32_t)(((ch0 + r16) * 4 + q) * 16);  // this lane's 16 B inside a chunk of weights
    f32x4 w0 = f32x4{0.0f, 0.0f, 0.0f, 0.0f}, w1 = w0;                // the weight stream's two chunks in flight between layers
    // ---- layer 0 on the plain image: tile t = rows 16t … 16t + 15 ----
    {
        const int rho0 = tile0 * 16 + r16;
        int vmask[RTW];
        conv_tap_masks<RTW>(rows, n, nsq, rho0, vmask);
        if (short_group) vmask[RTW - 1] = 0;
        const f32x4* wp = (const f32x4*)(CB ? T.w0_board : T.w[0]) + ((size_t)(ch0 + r16) * 4 + q);
        const int last_t0 = CB ? T.cb_last_t : T.cin_last_t;
        TG_STAMP(0, 0);
        if (RTW > 1 && short_group) {
            f32x4 (&acs)[RTW - 1] = *reinterpret_cast<f32x4 (*)[RTW - 1]>(&acc[0]);
            conv_mainloop<RTW - 1, CH0>(lds4, wp, (size_t)F * 4, LS4, rows, n, rho0, q, vmask, acs, last_t0);
        } else {
            conv_mainloop<RTW, CH0>(lds4, wp, (size_t)F * 4, LS4, rows, n, rho0, q, vmask, acc, last_t0);
        }
        TG_STAMP(0, 1);
        if (T.nlayers > 1) conv_halo_first_weights<CH>(T.w[1], wlane, w0, w1);  // in flight during the change of images
        const f32x4 bv = *(const f32x4*)&T.b[0][ch0 + 4 * q];
#pragma unroll
        for (int j = 0; j < RTW; j++) {
            f32x4 v = acc[j] + (CB ? pb4[tower_cb_index(rho0 + j * 16, rows, n, nsq, F >> 2, (ch0 >> 2) + q)] : bv);
            v[0] = fmaxf(v[0], 0.0f); v[1] = fmaxf(v[1], 0.0f); v[2] = fmaxf(v[2], 0.0f); v[3] = fmaxf(v[3], 0.0f);
            acc[j] = v;
        }
        TG_STAMP(0, 2);
        __syncthreads();  // every wave has finished reading the input planes
        TG_STAMP(0, 3);
        // the halo image replaces them: zero cells first (they are never written again), then this layer's output
        const int cells = LEAD + PW * PS + 1;  // + the spare cell of the idle slots
        for (int idx = tid; idx < cells * P4; idx += NWAVES * 64) {
            const int c = idx / P4 - LEAD;
            const int o = c < 0 || c >= PW * PS ? n * RS : c % PS;  // offset inside the position block; rows of RS cells, then the zero row
            if (o >= n * RS || o % RS == n) lds4[idx] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
#pragma unroll
        for (int j = 0; j < RTW; j++) {
            const int rho = rho0 + j * 16;
            if (j < my_tiles && rho < PW * nsq) {
                const int p = rho / nsq, sq = rho - p * nsq, y = sq / n, x = sq - y * n;
                lds4[(LEAD + p * PS + y * RS + x) * P4 + (ch0 >> 2) + q] = acc[j];
            }
            acc[j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
        TG_STAMP(0, 4);
        __syncthreads();
        TG_STAMP(0, 5);
    }

    // ---- layers 1 … : slot (tile, lane) → square through the slot table ----
    int cell4[RTW], rowid[RTW], addr4[RTW];
#pragma unroll
    for (int j = 0; j < RTW; j++) {
        const uint32_t e = j < my_tiles ? T.slotmap[(tile0 + j) * 16 + r16] : 0xFFFF0000u;
        rowid[j] = (int)(e >> 16);                                      // 0xFFFF: slot without a square
        const bool idle = rowid[j] == 0xFFFF;  // such a slot reads around a zero cell and writes the spare cell behind the image
        cell4[j] = (idle ? LEAD + PW * PS : (int)(e & 0xFFFFu)) * P4 + (ch0 >> 2) + q;  // this lane's 4 output channels of the square
        addr4[j] = ((idle ? LEAD + n * RS : (int)(e & 0xFFFFu)) - LEAD) * P4 + q;       // B-operand base: tap (-1,-1), chunk 0
    }
    const int turn = (wave >> 2) & 1;  // waves w and w + 4 share a SIMD
    for (int layer = 1; layer < T.nlayers; layer++) {
        // the addresses are the same in every layer, but the compiler must not know: it would hoist all 9·RTW
        // (address + tap offset) sums out of the layer loop and spill them instead of using ds_read immediates
#pragma unroll
        for (int j = 0; j < RTW; j++) asm volatile("" : "+v"(addr4[j]));
        TG_STAMP(layer, 0);
        const float* wnext = T.w[layer + 1 < T.nlayers ? layer + 1 : layer];
        const f32x4 bv = *(const f32x4*)&T.b[layer][ch0 + 4 * q];  // requested here: its latency passes under the main loop
        if (RTW > 1 && short_group) {
            f32x4 (&acs)[RTW - 1] = *reinterpret_cast<f32x4 (*)[RTW - 1]>(&acc[0]);
            conv_mainloop_halo<RTW - 1, CH, NB>(lds4, T.w[layer], wnext, wlane, addr4, acs, turn, w0, w1);
        } else {
            conv_mainloop_halo<RTW, CH, NB>(lds4, T.w[layer], wnext, wlane, addr4, acc, turn, w0, w1);
        }
        TG_STAMP(layer, 1);
#pragma unroll
        for (int j = 0; j < RTW; j++) {
            f32x4 v = acc[j] + bv;
            v[0] = fmaxf(v[0], 0.0f); v[1] = fmaxf(v[1], 0.0f); v[2] = fmaxf(v[2], 0.0f); v[3] = fmaxf(v[3], 0.0f);
            acc[j] = v;
        }
        if (layer + 1 == T.nlayers) {
#pragma unroll
            for (int j = 0; j < RTW; j++)
                if (rowid[j] < rows) {
                    if (T.frag_out) {
                        const int p = pos0 + rowid[j] / nsq, sq = rowid[j] % nsq;
                        ((f32x4*)out)[((size_t)(p >> 4) * (nsq * CH) + sq * CH + ct) * 64 + (p & 15) * 4 + q] = acc[j];
                    } else *(f32x4*)&out[((size_t)pos0 * nsq + rowid[j]) * F + ch0 + 4 * q] = acc[j];
                }
            break;
        }
        TG_STAMP(layer, 2);
        __syncthreads();  // every wave has finished reading the previous image
        TG_STAMP(layer, 3);
        const bool conv1 = (layer & 1) == 1;  // next layer is conv2 of the same block: it starts from the block input
#pragma unroll
        for (int j = 0; j < RTW; j++) {  // no per-tile branches: idle slots have their own cell
            f32x4 x0 = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if (conv1) x0 = lds4[cell4[j]];
            lds4[cell4[j]] = acc[j];
            acc[j] = x0;
        }
        TG_STAMP(layer, 4);
        __syncthreads();
        TG_STAMP(layer, 5);
    }
}

// ------------------------------------------------------------------------------------------------
// ONE 3×3 layer on the halo image (the main loop of k_tower_halo for a layer that stands alone): the convolutions of the
// training step — forward in training mode and the data gradient, F → F, activations in HBM between them because
// BatchNorm's batch statistics sit between two layers — and the conv policy head of Net6 (F → 251 in 256).  The input rows
// of the workgroup's PW positions are staged from global straight into halo cells; taps are ds_read immediates, the
// weights stream through a buffer descriptor two chunks ahead, slots come from the same slot table as the tower's.
// Epilogue: + bias, + res (optional), ReLU (optional).  COT = CoutP / 16; blockIdx.y picks a group of CTW channel tiles.
// ------------------------------------------------------------------------------------------------
// PSC: the position stride of the halo image (tower_halo_geometry) as a constant — the zero-cell fill divides by it 19 000 times
template <int RTW, int NWAVES, int CH, int NB, int COT, int PSC>
__global__ __launch_bounds__(NWAVES * 64) void k_conv_halo(const float* __restrict__ in, const float* __restrict__ Wp,
                                                           const float* __restrict__ bias, const float* __restrict__ res,
                                                           float* __restrict__ out, const uint32_t* __restrict__ slotmap, int B, int PW,
                                                           int PS, int CTW, int out_stride, int cout_valid, int relu,
                                                           double* __restrict__ stats_part, const ConvBnBwdIn bnb) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    f32x4* lds4 = (f32x4*)lds;
    constexpr int n = NB, nsq = NB * NB, RS = NB + 1, LEAD = NB + 2, F4 = 4 * CH, P4 = 4 * CH + 1;
    PS = PSC;
    const int tid = threadIdx.x;
    const int pos0 = blockIdx.x * PW;
    const int npos = min(PW, B - pos0);
    const int rows = npos * nsq;
    const int wave = tid >> 6, lane = tid & 63;
    const int ct = wave % CTW, rg = wave / CTW;
    const int r16 = lane & 15, q = lane >> 4;
    const int ch0 = (blockIdx.y * CTW + ct) * 16;
    const uint32_t wlane = (uint32_t)(((ch0 + r16) * 4 + q) * 16);
    f32x4 w0, w1;
    TG_STAMP(0, 0);
#ifdef TG_TOWER_STAMPS  // wall-clock (100 MHz) start and end of every workgroup: dispatch skew and tail of a launch
    if (g_tower_stamps && tid == 0) g_tower_stamps[128 + 2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
#endif
    conv_halo_first_weights<CH, COT>(Wp, wlane, w0, w1);  // in flight while the image is staged
    // zero cells (behind every board row, the zero row behind every position, lead and tail) …
    const int cells = LEAD + PW * PS + 1;
    for (int idx = tid; idx < cells * P4; idx += NWAVES * 64) {
        const int c = idx / P4 - LEAD;
        const int o = c < 0 || c >= PW * PS ? n * RS : c % PS;
        if (o >= n * RS || o % RS == n) lds4[idx] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
    // … and the squares' rows from global (row-major [row][16·CH]), 8 loads in flight per lane
    auto halo_cell = [&](int idx) {
        const int r = idx / F4, v = idx - r * F4;
        const int p = r / nsq, sq = r - p * nsq, y = sq / n, x = sq - y * n;
        return (LEAD + p * PS + y * RS + x) * P4 + v;
    };
    const f32x4* src = (const f32x4*)(in + (size_t)pos0 * nsq * (16 * CH));
    const int total = rows * F4;
    {
        constexpr int UNR = 8;
        for (int base = 0; base < total; base += NWAVES * 64 * UNR) {
            f32x4 tmp[UNR];
#pragma unroll
            for (int u = 0; u < UNR; u++) {
                const int idx = base + u * NWAVES * 64 + tid;
                tmp[u] = src[idx < total ? idx : total - 1];
            }
#pragma unroll
            for (int u = 0; u < UNR; u++) {
                const int idx = base + u * NWAVES * 64 + tid;
                if (idx < total) lds4[halo_cell(idx)] = tmp[u];
            }
        }
    }
    __syncthreads();
    TG_STAMP(0, 1);

    const int NRG = NWAVES / CTW;
    const int ntiles = (PW * nsq + 15) >> 4;
    const int tbase = ntiles / NRG, trem = ntiles - tbase * NRG;
    const int my_tiles = tbase + (rg < trem ? 1 : 0);
    const int tile0 = rg * tbase + min(rg, trem);
    const bool short_group = my_tiles < RTW;
    f32x4 acc[RTW];
    int addr4[RTW];
#pragma unroll
    for (int j = 0; j < RTW; j++) {
        acc[j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        const uint32_t e = j < my_tiles ? slotmap[(tile0 + j) * 16 + r16] : 0xFFFF0000u;
        const bool idle = (e >> 16) == 0xFFFFu;  // a slot without a square reads around a zero cell and stores nothing
        addr4[j] = ((idle ? LEAD + n * RS : (int)(e & 0xFFFFu)) - LEAD) * P4 + q;
    }
    const int turn = (wave >> 2) & 1;
    TG_STAMP(0, 2);
    if (RTW > 1 && short_group) {
        f32x4 (&acs)[RTW - 1] = *reinterpret_cast<f32x4 (*)[RTW - 1]>(&acc[0]);
        conv_mainloop_halo<RTW - 1, CH, NB, RTW, COT, true>(lds4, Wp, Wp, wlane, addr4, acs, turn, w0, w1);
    } else {
        conv_mainloop_halo<RTW, CH, NB, RTW, COT, true>(lds4, Wp, Wp, wlane, addr4, acc, turn, w0, w1);
    }
    TG_STAMP(0, 3);
    const int ch = ch0 + 4 * q;
    const f32x4 bv = *(const f32x4*)&bias[ch];
    // (the tiles' rows are looked up again here rather than kept in 13 registers across the main loop, whose two accumulator sets
    // leave none to spare)
    int rowid[RTW];
#pragma unroll
    for (int j = 0; j < RTW; j++) rowid[j] = j < my_tiles ? (int)(slotmap[(tile0 + j) * 16 + r16] >> 16) : 0xFFFF;
    // stats_part: Σ and Σ² of this lane's outputs, per channel — in double from the first add on: var = E[z²] − E[z]² loses
    // (mean/σ)² of the sums' relative accuracy, and f32 partials over up to 208 rows left 1e-4 of the variance at |mean| = 10σ.
    // With bnb.y set (round 4; the data-gradient convolution of the training step): the output IS dy of the layer below, and the
    // same two slots collect that layer's BatchNorm-backward sums Σg and Σg·x̂ (g = dy·[y > 0], x̂ = (z − mean)·invstd) while dy is
    // in registers — k_col_reduce<RED_BNBWD>'s pass over dy, y and z (27 µs per layer) is gone
    double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
    f32x4 bn_mu = f32x4{0.0f, 0.0f, 0.0f, 0.0f}, bn_is = bn_mu;
    if (bnb.y && ch0 + 4 * q < cout_valid) { bn_mu = *(const f32x4*)&bnb.mean[ch0 + 4 * q]; bn_is = *(const f32x4*)&bnb.invstd[ch0 + 4 * q]; }
#pragma unroll
    for (int j = 0; j < RTW; j++) {
        if (rowid[j] < rows && ch < cout_valid) {
            const size_t o = ((size_t)pos0 * nsq + rowid[j]) * out_stride + ch;
            f32x4 v = acc[j] + bv;
            if (res) v += *(const f32x4*)&res[o];
            if (relu) { v[0] = fmaxf(v[0], 0.0f); v[1] = fmaxf(v[1], 0.0f); v[2] = fmaxf(v[2], 0.0f); v[3] = fmaxf(v[3], 0.0f); }
            if (ch + 3 < cout_valid) *(f32x4*)&out[o] = v;
            else for (int t = 0; t < 4; t++) if (ch + t < cout_valid) out[o + t] = v[t];
            if (stats_part && !bnb.y) {
#pragma unroll
                for (int t = 0; t < 4; t++) { const double d = (double)v[t]; s1[t] += d; s2[t] = fma(d, d, s2[t]); }
            } else if (stats_part) {
                const f32x4 yy = *(const f32x4*)&bnb.y[o], zz = *(const f32x4*)&bnb.z[o];
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const float g = yy[t] > 0.0f ? v[t] : 0.0f;
                    s1[t] += (double)g;
                    s2[t] += (double)(g * ((zz[t] - bn_mu[t]) * bn_is[t]));
                }
            }
        }
    }
    if (stats_part) {
        // BatchNorm's batch statistics from the accumulators (training forward): the column sums of this wave's rows — 16
        // lanes of a q-group hold 16 rows of the same 4 channels — leave the kernel as doubles, one partial row per (workgroup,
        // row group): part[((blockIdx.x·NRG + rg)·2 + {Σ, Σ²})·CoutP + channel], k_col_reduce's layout, summed in fixed order later
#pragma unroll
        for (int d = 1; d < 16; d <<= 1)
#pragma unroll
            for (int t = 0; t < 4; t++) { s1[t] += __shfl_xor(s1[t], d); s2[t] += __shfl_xor(s2[t], d); }
        if (r16 == 0) {
            const int CoutP = 16 * COT;
            double* dst = stats_part + ((size_t)(blockIdx.x * NRG + rg) * 2) * CoutP + ch;
#pragma unroll
            for (int t = 0; t < 4; t++) { dst[t] = s1[t]; dst[CoutP + t] = s2[t]; }
        }
    }
    TG_STAMP(0, 4);
#ifdef TG_TOWER_STAMPS
    if (g_tower_stamps && tid == NWAVES * 64 - 64) g_tower_stamps[128 + 2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
#endif
}

// ONE 3×3 layer for SMALL batches (round 6; the conv policy head of Net6 at the reference's 32 leaves: k_conv_pos gave a position to
// one workgroup — 32 busy CUs — with 3 row tiles per wave and the weights one chunk ahead): workgroup = (position, 16-channel tile),
// wave = row tile, ONE chain per wave with the weights a tap ahead (conv_mainloop_tile).  k_conv_pos's sums — a chain per tap, added in
// tap order — so the same bits.  No residual, no statistics: the head and plain layers.
template <int NRT, int CH>
__global__ __launch_bounds__(NRT * 64) void k_conv_split(const float* __restrict__ in, const float* __restrict__ Wp, const float* __restrict__ bias,
                                                         float* __restrict__ out, int n, int CoutP, int out_stride, int cout_valid, int relu) {
    constexpr int F = 16 * CH, F4 = F / 4, LS4 = (F + LDS_PAD16) >> 2;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    f32x4* lds4 = (f32x4*)lds;
    const int tid = threadIdx.x, nsq = n * n, rows = nsq;
    const int p = blockIdx.x, ch0 = blockIdx.y * 16;
    const int rt = tid >> 6, lane = tid & 63, r16 = lane & 15, q = lane >> 4;
    const int rho = rt * 16 + r16;
    const f32x4* wp = (const f32x4*)Wp + ((size_t)(ch0 + r16) * 4 + q);
    const size_t wstride4 = (size_t)CoutP * 4;
    f32x4 wf[CH];
    conv_tile_first_weights<CH>(wp, wstride4, wf);  // in flight while the image is staged
    {
        const f32x4* src = (const f32x4*)(in + (size_t)p * nsq * F);
        const int total = nsq * F4;
        constexpr int UNR = 8;
        for (int base = 0; base < total; base += NRT * 64 * UNR) {
            f32x4 tmp[UNR];
#pragma unroll
            for (int u = 0; u < UNR; u++) {
                const int idx = base + u * NRT * 64 + tid;
                tmp[u] = src[idx < total ? idx : total - 1];
            }
#pragma unroll
            for (int u = 0; u < UNR; u++) {
                const int idx = base + u * NRT * 64 + tid;
                if (idx < total) lds4[(idx / F4) * LS4 + idx % F4] = tmp[u];
            }
        }
        for (int idx = tid; idx < LS4; idx += NRT * 64) lds4[rows * LS4 + idx] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
    __syncthreads();
    int vmask[1];
    conv_tap_masks<1>(rows, n, nsq, rho, vmask);
    f32x4 acc = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    conv_mainloop_tile<CH, 4, true>(lds4, wp, wstride4, LS4, rows, n, rho, q, vmask[0], acc, wf);
    const int ch = ch0 + 4 * q;
    if (rho < rows && ch < cout_valid) {
        f32x4 v = acc + *(const f32x4*)&bias[ch];
        if (relu) { v[0] = fmaxf(v[0], 0.0f); v[1] = fmaxf(v[1], 0.0f); v[2] = fmaxf(v[2], 0.0f); v[3] = fmaxf(v[3], 0.0f); }
        const size_t o = ((size_t)p * nsq + rho) * out_stride + ch;
        if (ch + 3 < cout_valid) *(f32x4*)&out[o] = v;
        else for (int t = 0; t < 4; t++) if (ch + t < cout_valid) out[o + t] = v[t];
    }
}

// Plain GEMM out[M][N] = A[M][K]·W[K][N] + bias for the 5×5 policy FC (net5.rs:56-61,108): the same
// fragments, A staged through LDS in K-chunks of 32.
template <int RT, int CT>
__global__ __launch_bounds__(256) void k_gemm(const float* __restrict__ A, int lda, const float* __restrict__ Wp,
                                              const float* __restrict__ bias, float* __restrict__ out, int M, int K,
                                              int NP, int out_stride, int n_valid) {
    constexpr int TM = 64 * RT;
    constexpr int KC = 32;
    constexpr int LS = KC + LDS_PAD;
    __shared__ __attribute__((aligned(16))) float lds[2][TM * LS];
    const int tid = threadIdx.x;
    const int m0 = blockIdx.x * TM;
    const int wave = tid >> 6, lane = tid & 63;
    const int wr = wave & 1, wc = wave >> 1;
    const int i = lane & 31, h = lane >> 5;
    const int col0 = blockIdx.y * (64 * CT) + wc * (32 * CT);
    const float* wlane = Wp + ((size_t)(col0 + i) * 16 + 4 * h);
    const size_t wchunk_stride = (size_t)NP * 16;

    f32x16 acc[RT][CT];
#pragma unroll
    for (int rt = 0; rt < RT; rt++)
#pragma unroll
        for (int ct = 0; ct < CT; ct++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[rt][ct][r] = 0.0f;

    auto stage = [&](int buf, int k0) {
        // TM rows × 8 float4
        for (int idx = tid; idx < TM * (KC / 4); idx += 256) {
            int r = idx >> 3, v = idx & 7;
            int m = m0 + r;
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < M) x = *(const float4*)(A + (size_t)m * lda + k0 + v * 4);
            *(float4*)&lds[buf][r * LS + v * 4] = x;
        }
    };

    const int nchunk = K / KC;
    stage(0, 0);
    __syncthreads();
    for (int kc = 0; kc < nchunk; kc++) {
        const int buf = kc & 1;
        if (kc + 1 < nchunk) stage(buf ^ 1, (kc + 1) * KC);
#pragma unroll
        for (int c8 = 0; c8 < KC / 8; c8++) {
            f32x4 a[RT], b[CT];
            const size_t kchunk = (size_t)kc * (KC / 8) + c8;
#pragma unroll
            for (int ct = 0; ct < CT; ct++)
                b[ct] = *(const f32x4*)(wlane + (kchunk >> 1) * wchunk_stride + (size_t)ct * 32 * 16 + 8 * (kchunk & 1));
#pragma unroll
            for (int rt = 0; rt < RT; rt++) a[rt] = *(const f32x4*)&lds[buf][((wr * RT + rt) * 32 + i) * LS + c8 * 8 + 4 * h];
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int rt = 0; rt < RT; rt++)
#pragma unroll
                    for (int ct = 0; ct < CT; ct++)
                        acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[rt][t], b[ct][t], acc[rt][ct], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int rt = 0; rt < RT; rt++)
#pragma unroll
        for (int ct = 0; ct < CT; ct++) {
            const int col = col0 + ct * 32 + i;
            const float bv = bias[col];
#pragma unroll
            for (int r = 0; r < 16; r++) {
                int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                int m = m0 + (wr * RT + rt) * 32 + row;
                if (m < M && col < n_valid) out[(size_t)m * out_stride + col] = acc[rt][ct][r] + bv;
            }
        }
}


// Policy FC (net5.rs:56-61,108) for the BASELINE shape M = 4096, K = 1600, N = 1575 (+ the value head in column 1575): 1576
// useful columns are 98.5 MFMA tiles of 16 → FC_TILES = 99.  A workgroup covers 128 rows (8 row tiles, one per wave) × 12 MAIN
// tiles (column block cb: tiles 12 cb … 12 cb + 11 = 192 columns) → 32 × 8 = 256 workgroups, one per CU — and ONE of the three
// leftover tiles (96, 97, 98) for some of its row tiles: the 8 row tiles × 3 leftover tiles of a row block are 24 (row tile,
// tile) pairs, dealt 3 / 3 / 2 / 3 / 3 / 2 / 4 / 4 to the 8 workgroups of the row block (fc_extra): workgroup cb computes
// leftover tile 96 + l for the row tiles s … s + ne − 1, in its waves 0 … ne − 1 — different SIMDs (waves w and w + 4 share
// one), so a SIMD carries 12 + 12 or 12 + 13 tile chains where round 3's 8 × 13-tile blocks (104 tiles, 5 of them padding)
// made it 26.  Wave w owns row tile (w + s) mod 8: the rotation puts the rows that need the leftover tile into waves 0 … ne − 1.
// The weights (shared by the 8 waves) go global → LDS in K-steps of 64, in four 16-byte-slot planes (one per k-quarter) so that
// a wave's 16 lanes of one plane hit 16 different bank groups: conflict-free ds_read_b128; slot 12 of a chunk's 13 tile slots
// holds the workgroup's leftover tile.  The wave's own 16 activation rows are the MFMA B operand, read straight from global.
constexpr int FC_CT = FC_MAIN_TILES + 1;  // tile slots per workgroup: 12 main + its leftover tile
constexpr int FC_KSTEP = 64;              // 4 chunks of 16
constexpr int FC_PLANE = (FC_KSTEP / 16) * FC_CT * 16;  // 832 slots per k-quarter plane (≡ 0 mod 16)

// Barrier-free ring: three weight buffers of one K-step filled by LDS-DMA (global_load_lds_dwordx4: no staging registers, no
// ds_write phase, 16 cache lines per instruction); the eight waves synchronise through two sets of monotonic counters in LDS
// instead of s_barrier:
//   ready[b] += 1 by every wave once its share of the K-step now in buffer b has landed (s_waitcnt vmcnt),
//   done[b]  += 1 by every wave once it has read the last fragment of the K-step in buffer b.
// A wave reads step s after ready[s % 3] = 8·(s/3 + 1) and refills buffer (s + 2) % 3 — in the MIDDLE of step s, half a
// step after it finished reading it itself — after done[(s + 2) % 3] = 8·⌊(s + 2)/3⌋.  Both flags are read half a chunk
// before they are needed and normally hold by then, so no wave waits out a round trip and the waves may drift half a step
// apart instead of draining the MFMA pipe at a barrier every 17 k cycles.  Every output element is accumulated over k in the
// same order by the same MFMA as in k_fc_small → identical logits bits (tests/test_gpu_net.py, batch independence).
// What bounds it (round 4's probe builds, profiles/r04_b_fc_candidates.txt): with neither refills nor flags the loop is 17 µs shorter —
// the LDS-DMA pieces' issue slots beside the fragment reads and waves held back for a slower one; the MFMAs of the 88 padded
// columns were 3.2 µs, the logits burst 3.7 µs.  Measured and discarded: a ninth wave that only fills the ring (8 – 10 µs
// slower), non-temporal logits stores, the barrier version k_fc_lds (rounds 1 – 3: + 6 µs), a register-tiled FC without LDS
// (k_fc_reg, scripts/probes/fc_reg.cuh: 227 µs — 2.7 × the operand bytes through the vector-memory path); round 4 also: a static
// s_setprio 1 for waves 4-7 (−1 µs, inside the noise) or for waves 0-3 (0), and waves 4-7 issuing their share of a refill half a
// step after waves 0-3 so that the two waves of a SIMD never issue LDS-DMA pieces at the same time (+ 11 µs: the older wave of
// a SIMD runs ahead of the younger one anyway, and the later refill makes the younger one the workgroup's laggard).
constexpr int FC_RING = 3;
constexpr int FC_RING_SLOTS = 4 * FC_PLANE;                                        // f32x4 slots per buffer (3328)
constexpr size_t FC_RING_LDS = (size_t)FC_RING * FC_RING_SLOTS * 16 + 2 * FC_RING * sizeof(uint32_t);
// (diagnostic build only — scripts/probes/fc_ring_stamps.hip: stamps of workgroup (0, 0); s_memtime has another base on every XCD)
#define TG_FC_STAMP(step, slot) do { if (blockIdx.y == 0) { TG_STAMP(step, slot); } } while (0)
template <int GEOM>  // 0: the policy head's 99 tiles (8 blocks + 3 leftover tiles, fc_extra); 1: 25x tiles as 2x blocks + x leftover tiles
__global__ __launch_bounds__(512) void k_fc_ring(const float* __restrict__ A, int lda, const float* __restrict__ Wp,
                                                 const float* __restrict__ bias, float* __restrict__ out, int M, int K, int NP,
                                                 int out_stride, int n_valid, int a_frag, float* __restrict__ stats, int n_soft,
                                                 const FcGather gather, const float* __restrict__ Wlin) {
    extern __shared__ __attribute__((aligned(16))) float fc_ring_lds[];
    f32x4* wl = (f32x4*)fc_ring_lds;                                // [FC_RING][chunk][tile slot][q][r16]
    uint32_t* flags = (uint32_t*)(wl + FC_RING * FC_RING_SLOTS);    // ready[FC_RING], done[FC_RING]
    const uint32_t ready0 = (uint32_t)(size_t)(__attribute__((address_space(3))) uint32_t*)flags;  // LDS byte addresses
    const uint32_t done0 = ready0 + FC_RING * 4;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int r16 = lane & 15, q = lane >> 4;
    // Which (row block, column block) this workgroup computes.  Workgroups go to the 8 XCDs round robin by their linear index, so with
    // (row block, column block) = blockIdx every XCD's L2 fetches ALL the weights (8 × 10.1 MB) and its quarter of the rows once: 107 MB.
    // An XCD that computes a column blocks × 32 / a row blocks fetches 10.1 a + 210 / a MB: least at a = 4 — XCD c gets column blocks
    // 4 (c & 1) … + 3 and every fourth row block from c >> 1 on.  Measured (scripts/probes/fc_xcd_map.sh): memory-side traffic 134.0 →
    // 120.5 MB with logits rows (3.64 → 3.27 × algorithmic), the launch time unchanged (166.3 – 166.6 µs either way: L2 misses that hit
    // the Infinity Cache were never what it waited for).  The same results by other workgroups: nothing changes in the output.
    int rbx = (int)blockIdx.x, cbx = (int)blockIdx.y;
    if (GEOM == 0 && (gridDim.x & 7) == 0) {
        const int xcd = rbx & 7, j = (rbx >> 3) + (int)(gridDim.x >> 3) * cbx;
        cbx = 4 * (xcd & 1) + (j & 3);
        rbx = (xcd >> 1) + 4 * (j >> 2);
    }
    const int cb = cbx;
    // GEOM 1 (round 4: the training step's FC data gradient, 200 tiles = 16 × 12 + 8): gridDim.y = 2x blocks; leftover tile cb / 2 for
    // row tiles 0 … 3 (even cb) or 4 … 7 (odd cb), in waves 0 … 3 — every SIMD carries 12 + 13 tile chains, no tile is padding.  (A template
    // parameter: the block count as a kernel argument cost the policy head 2.7 µs, 167.5 against 164.8 µs.)
    const int mb = GEOM == 0 ? FC_MAIN_BLOCKS : (int)gridDim.y;
    const FcExtra X = GEOM == 0 ? fc_extra(cb) : FcExtra{cb >> 1, 4 * (cb & 1), 4};
    const bool has13 = wave < X.ne;                    // this wave also computes the leftover tile for its rows (wave-uniform)
    const int rt = (wave + X.s) & 7;                   // row tile of the row block owned by this wave
    const int row = rbx * 128 + rt * 16 + r16;
    const bool row_ok = row < M;
    const int n0 = cb * (FC_MAIN_TILES * 16);          // first column of the main tiles
    const int nx = (FC_MAIN_TILES * mb + X.l) * 16;               // first column of the leftover tile
    // loads are unconditional (rows past the end read a valid row and are never stored): hipcc puts s_waitcnt vmcnt(0)
    // right behind an exec-masked global load.  Activations: row-major (one 16-B slot of its row per lane and chunk), or
    // fragment-major (TowerParams.frag_out: the wave's 16 rows × 16 k of a chunk are one contiguous KB)
    const int last_tile = (M - 1) >> 4;
    const int my_tile = min(rbx * 8 + rt, last_tile);
    const f32x4* ap = a_frag ? (const f32x4*)A + (size_t)my_tile * (K >> 4) * 64 + r16 * 4 + q
                             : (const f32x4*)(A + (size_t)(row_ok ? row : M - 1) * lda) + q;
    const size_t achunk = a_frag ? 64 : 4;
    const f32x4* wg = (const f32x4*)Wp;  // slot (chunk, col, q) at (chunk*NP + col)*4 + q
    const int nsteps = K / FC_KSTEP;
    const int nchunks = nsteps * 4;

    // LDS-DMA: one wave-instruction fills 64 consecutive slots of a buffer (1 KB) = one (chunk, tile slot) block, slot
    // q·16 + r16 inside it — the lane number of its reader, so the fragment reads are contiguous and conflict free — from
    // the block's 1 KB of the weight matrix (slot r16·4 + q: the permutation is on the source side, 16 cache lines per
    // instruction).  52 blocks per K-step, issued by the filler waves (below).
    // Wlin (optional): the same weights with every (chunk, tile) block stored in the READER's lane order — slot (chunk·NP/16 + tile)·64 +
    // q·16 + r16 — so that an LDS-DMA instruction's 64 lanes read 64 consecutive 16-byte slots (the permuted source makes each
    // quarter-wave touch 16 different cache lines of the block)
    if (Wlin) wg = (const f32x4*)Wlin;
    // Who issues the refills: waves 4-7 — the YOUNGER wave of every SIMD — issue all 52 pieces of a K-step (13 each), waves 0-3 none,
    // and ready[] counts 4 per use.  The older wave of a SIMD wins the matrix pipe and runs ahead; the younger one lags anyway, and
    // while it spends 2 – 3.5 k cycles per step handing pieces to the memory pipe its partner issues MFMAs undisturbed (measured,
    // profiles/r04_b_fc_candidates.txt §7: every wave issuing its share 168.9 µs, waves 0-3 all of them 168.5 µs, waves 4-7 all of
    // them 166.2 µs; TG_FC_ALL_FILL restores the first)
#ifdef TG_FC_ALL_FILL
    constexpr int FILLERS = 8, PER = 7;
    const int fwave = wave;
#else
    constexpr int FILLERS = 4, PER = 13;
    const int fwave = wave - 4;
#endif
    constexpr bool HALF_FILL = FILLERS < 8;
    uint32_t src0[PER];
#pragma unroll
    for (int u = 0; u < PER; u++) {
        int blk = (fwave < 0 ? 0 : fwave) + FILLERS * u;
        blk = blk < FC_RING_SLOTS / 64 ? blk : FC_RING_SLOTS / 64 - 1;
        const int c = blk / FC_CT, j = blk - c * FC_CT;
        const int col0 = j < FC_MAIN_TILES ? n0 + j * 16 : nx;
        src0[u] = Wlin ? (uint32_t)(((size_t)c * (NP >> 4) + (col0 >> 4)) * 64 + lane) : (uint32_t)(((size_t)c * NP + col0 + r16) * 4 + q);
    }
    const uint32_t step_slots = (uint32_t)(4 * NP * 4);  // f32x4 slots of the weights per K-step (either layout)
    auto fill = [&](int step, int buf) {
        if (HALF_FILL && (fwave < 0 || fwave >= FILLERS)) return;
#pragma unroll
        for (int u = 0; u < PER; u++)
            if (fwave + FILLERS * u < FC_RING_SLOTS / 64)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wg + (size_t)step * step_slots + src0[u]),
                                                 (__attribute__((address_space(3))) void*)(wl + buf * FC_RING_SLOTS + (fwave + FILLERS * u) * 64), 16, 0, 0);
    };
    auto aload = [&](int kc) { return ap[(size_t)(kc < nchunks ? kc : nchunks - 1) * achunk]; };
    if (tid < 2 * FC_RING) flags[tid] = 0u;
    __syncthreads();
    fill(0, 0);
    if (nsteps > 1) fill(1, 1);
    f32x4 acc[FC_CT];
#pragma unroll
    for (int j = 0; j < FC_CT; j++) acc[j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    // Activations: hipcc waits vmcnt(0) wherever the result of an ordinary load is consumed while an LDS-DMA load may be in
    // flight, and loads return in order.  So the four chunks up to the middle of the next step are requested at the top of
    // a step and forced to complete right before the refill is issued (two chunks later): no activation load is ever queued
    // behind a young refill, whose data comes from the MALL or HBM and takes its time.
    f32x4 a0 = aload(0), a1 = aload(1), a2, a3, b0, b1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const bool filler = !HALF_FILL || (fwave >= 0 && fwave < FILLERS);
    if (filler) {
        fc_ring_signal(ready0);
        if (nsteps > 1) fc_ring_signal(ready0 + 4);
    }
    // One chunk: the 13 weight fragments in two halves (7 + 6 tile slots; the 13th only feeds MFMAs in the waves that own a
    // leftover tile); each half is requested while the other half's MFMAs run, across chunk boundaries inside a step (the
    // tower's half-tile pipeline, conv_mainloop.cuh).
    constexpr int FC_H1 = 7;
    f32x4 w[FC_CT];
#define TG_FC_LOAD(C, J0, J1) _Pragma("unroll") for (int j = J0; j < J1; j++) w[j] = wb[((C) * FC_CT + j) * 64 + lane];
#define TG_FC_MFMA(AV, J0, J1)                                                                                       \
    _Pragma("unroll") for (int t = 0; t < 4; t++)                                                                    \
        _Pragma("unroll") for (int j = J0; j < J1; j++)                                                              \
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j][t], (AV)[t], acc[j], 0, 0, 0);
#define TG_FC_CHUNK(C, AV, NEXT, EARLY)                                                                              \
    TG_FC_LOAD(C, FC_H1, FC_CT)                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    TG_FC_MFMA(AV, 0, FC_H1)                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    if (NEXT) { TG_FC_LOAD((C) + 1, 0, FC_H1) }                                                                      \
    EARLY;                                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    TG_FC_MFMA(AV, FC_H1, FC_MAIN_TILES)                                                                             \
    if (has13) { TG_FC_MFMA(AV, FC_MAIN_TILES, FC_CT) }                                                              \
    __builtin_amdgcn_sched_barrier(0);
    const volatile __attribute__((address_space(3))) uint32_t* flag_lds = (const volatile __attribute__((address_space(3))) uint32_t*)flags;
    uint32_t early_ready = 0u, early_done = 0u;
    // gather mode: what the epilogue needs of this wave's rows is requested under the last K-steps' MFMAs — row r16's child
    // count in lane r16, and per row the first 128 child indices, two 16-bit indices per lane
    uint32_t g_cnt = 0u, g_pidx[16];
#pragma unroll
    for (int r = 0; r < 16; r++) g_pidx[r] = 0u;
    for (int step = 0; step < nsteps; step++) {
        const int buf = step % FC_RING;
        const f32x4* wb = wl + buf * FC_RING_SLOTS;
        // (the flags were read half a chunk ago, under the MFMAs: they normally hold already and nobody waits out a round trip)
        TG_FC_STAMP(step, 0);  // (diagnostic build only: scripts/probes/fc_ring_stamps.hip)
        if ((int)__builtin_amdgcn_readfirstlane((int)early_ready) < FILLERS * (step / FC_RING + 1))
            fc_ring_wait(ready0 + 4 * buf, (uint32_t)FILLERS * (uint32_t)(step / FC_RING + 1));
        TG_FC_STAMP(step, 1);
        __builtin_amdgcn_sched_barrier(0);
        TG_FC_LOAD(0, 0, FC_H1)
        a2 = aload(step * 4 + 2);
        a3 = aload(step * 4 + 3);
        b0 = aload(step * 4 + 4);
        b1 = aload(step * 4 + 5);
        if (gather.child_logit && step == nsteps - 1) {  // (no refill follows in the last step: these loads wait for nobody)
            const int tile_row0 = rbx * 128 + rt * 16;
            g_cnt = gather.leaf_rec[2 * (size_t)min(tile_row0 + r16, M - 1) + 1];
#pragma unroll
            for (int r = 0; r < 16; r++)
                g_pidx[r] = ((const uint32_t*)(gather.child_pidx + (size_t)min(tile_row0 + r, M - 1) * gather.stride))[lane];
        }
        __builtin_amdgcn_sched_barrier(0);
        TG_FC_CHUNK(0, a0, true, (void)0)
        TG_FC_CHUNK(1, a1, true, early_done = flag_lds[FC_RING + (step + 2) % FC_RING])
        // the middle of the step: signal the step after this one, refill the buffer of the step before it
        TG_FC_STAMP(step, 2);
        asm volatile("" : "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1));  // the compiler's own wait for the four loads above …
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // … which, loads returning in order, covers last step's refill too
        // (leaving this wait to the compiler in the waves that issue no LDS-DMA: no change, 165.6 – 167.3 against 165.7 – 165.9 µs)
        TG_FC_STAMP(step, 3);
        if (step >= 1 && step + 1 < nsteps && filler) fc_ring_signal(ready0 + 4 * ((step + 1) % FC_RING));
        if (step + 2 < nsteps && filler) {
            if ((int)__builtin_amdgcn_readfirstlane((int)early_done) < 8 * ((step + 2) / FC_RING))
                fc_ring_wait(done0 + 4 * ((step + 2) % FC_RING), 8u * (uint32_t)((step + 2) / FC_RING));
            TG_FC_STAMP(step, 4);
            fill(step + 2, (step + 2) % FC_RING);
        }
        TG_FC_STAMP(step, 5);
        __builtin_amdgcn_sched_barrier(0);
        TG_FC_CHUNK(2, a2, true, (void)0)
        TG_FC_CHUNK(3, a3, false, early_ready = flag_lds[(step + 1) % FC_RING])
        fc_ring_signal(done0 + 4 * buf);
        TG_FC_STAMP(step, 6);
        a0 = b0;
        a1 = b1;
    }
    TG_FC_STAMP(nsteps, 0);
#undef TG_FC_LOAD
#undef TG_FC_MFMA
#undef TG_FC_CHUNK
    // ---- epilogue: bias; logits or the children's logits; the statistics of this wave's blocks of its rows ----
    f32x4 v[FC_CT];
#pragma unroll
    for (int j = 0; j < FC_MAIN_TILES; j++) v[j] = acc[j] + *(const f32x4*)&bias[n0 + j * 16 + 4 * q];
    v[FC_MAIN_TILES] = acc[FC_MAIN_TILES] + *(const f32x4*)&bias[nx + 4 * q];
    if (stats) {
        float m, sm;
        float* srow = stats + (size_t)(row_ok ? row : 0) * (FC_STAT_STRIDE * 2);
        fc_block_stats<FC_MAIN_TILES>(*reinterpret_cast<const f32x4(*)[FC_MAIN_TILES]>(&v[0]), n0 + 4 * q, min(n_soft, n0 + FC_MAIN_TILES * 16), m, sm);
        if (row_ok && q == 0) *(float2*)&srow[cb * 2] = make_float2(m, sm);
        if (has13) {
            fc_block_stats<1>(*reinterpret_cast<const f32x4(*)[1]>(&v[FC_MAIN_TILES]), nx + 4 * q, min(n_soft, nx + 16), m, sm);
            if (row_ok && q == 0) *(float2*)&srow[(FC_MAIN_BLOCKS + X.l) * 2] = make_float2(m, sm);
            // column n_soft (= P) is the value head's pre-activation: pair FC_STAT_BLOCKS of the record
            const int dv = n_soft - (nx + 4 * q);
            if (row_ok && dv >= 0 && dv < 4) *(float2*)&srow[FC_STAT_BLOCKS * 2] = make_float2(dv == 0 ? v[FC_MAIN_TILES][0] : dv == 1 ? v[FC_MAIN_TILES][1] : dv == 2 ? v[FC_MAIN_TILES][2] : v[FC_MAIN_TILES][3], 0.0f);
        }
    }
    if (out && row_ok) {
#pragma unroll
        for (int j = 0; j < FC_CT; j++) {
            const int nn = (j < FC_MAIN_TILES ? n0 + j * 16 : nx) + 4 * q;
            if (nn < n_valid && (j < FC_MAIN_TILES || has13)) {
                float* o = out + (size_t)row * out_stride + nn;
                if (nn + 3 < n_valid) *(f32x4*)o = v[j];
                else for (int t = 0; t < 4; t++) if (nn + t < n_valid) o[t] = v[j][t];
            }
        }
    }
    if (gather.child_logit) {
        // This wave's 16 rows × 13 tile slots (13 312 B) go into its eighth of the two ring buffers that hold no data of the last
        // K-step, once every wave has read the steps that lived there (done[] — normally long true: a wave is at most half a
        // step behind); every LDS-DMA into them landed steps ago.  The laggard of the workgroup never waits here.
        const int bA = nsteps % FC_RING, bB = (nsteps + 1) % FC_RING;
        auto uses = [&](int b) { return b < nsteps ? (nsteps - b + FC_RING - 1) / FC_RING : 0; };
        fc_ring_wait(done0 + 4 * bA, 8u * (uint32_t)uses(bA));
        fc_ring_wait(done0 + 4 * bB, 8u * (uint32_t)uses(bB));
        constexpr int RP = FC_CT * 16;  // floats per parked row (208)
        float* park[2] = {(float*)(wl + bA * FC_RING_SLOTS) + wave * (8 * RP), (float*)(wl + bB * FC_RING_SLOTS) + wave * (8 * RP)};
        {
            float* dst = park[r16 >> 3] + (r16 & 7) * RP + 4 * q;
#pragma unroll
            for (int j = 0; j < FC_CT; j++) *(f32x4*)&dst[j * 16] = v[j];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the wave's own parked rows, now read by other lanes of the same wave
        const int tile_row0 = rbx * 128 + rt * 16;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int grow = min(tile_row0 + r, M - 1);
            const uint32_t cnt = tile_row0 + r < M ? min((uint32_t)__builtin_amdgcn_readlane((int)g_cnt, r), (uint32_t)gather.stride) : 0u;
            const float* prow = park[r >> 3] + (r & 7) * RP;
            float* crow = gather.child_logit + (size_t)grow * gather.stride;
            const uint16_t* irow = gather.child_pidx + (size_t)grow * gather.stride;
            for (uint32_t c0 = 0; c0 < cnt; c0 += 128) {
                const uint32_t pair = c0 == 0 ? g_pidx[r] : ((const uint32_t*)(irow + c0))[lane];
#pragma unroll
                for (int hlf = 0; hlf < 2; hlf++) {
                    const uint32_t c = c0 + 2 * lane + hlf;
                    const uint32_t p = hlf ? pair >> 16 : pair & 0xFFFFu;
                    const uint32_t dm = p - (uint32_t)n0, dx = p - (uint32_t)nx;
                    const bool in_main = dm < (uint32_t)(FC_MAIN_TILES * 16), in_x = has13 && dx < 16u;
                    if (c < cnt && (in_main || in_x)) crow[c] = prow[in_main ? dm : FC_MAIN_TILES * 16 + dx];
                }
            }
        }
    }
}

// The same FC for SMALL batches (host-driven MCTS evaluates 16–32 leaves per call; Player, pit): k_fc_ring gives a row block
// of 128 positions to one workgroup and needs ≥ 4096 rows to fill the chip, so a 32-row call took as long as a 4096-row
// one.  Here a wave owns one 16-row tile × 2 output tiles and streams both operands straight from global (no LDS, no
// barrier): M/16 × NP/32 waves.  Every output element is accumulated over k in the same order by the same MFMA as in
// k_fc_ring, so the two kernels return identical bits and the choice between them is invisible.
constexpr int FCS_CT = 2;
// up to here the small-batch kernel is the faster one: k_fc_ring's launch takes ≈ 160 µs whatever the rows (M / 128 row blocks × 8 column
// blocks of workgroups, each through the whole K loop: 64 of 256 CUs at 1024 rows), k_fc_small 39 µs per 512 rows.  Round 6 (the games sweep's
// plateau between 512 and 1024 games was THIS, not the tower): 512 → 2048; 700 games 423 → 320 µs per iteration, 1024: 422 → 342, 1500:
// 589 → 527, 2048: 591 → 566 (profiles/r06_e_tower_pw_sweep.txt)
constexpr int FC_SMALL_ROWS = 2048;
__global__ __launch_bounds__(256) void k_fc_small(const float* __restrict__ A, int lda, const float* __restrict__ Wp,
                                                  const float* __restrict__ bias, float* __restrict__ out, int M, int K, int NP,
                                                  int out_stride, int n_valid, int a_frag) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r16 = lane & 15, q = lane >> 4;
    const int row = blockIdx.x * 16 + r16;
    const bool row_ok = row < M;
    const int n0 = (blockIdx.y * 4 + wave) * (FCS_CT * 16);
    if (n0 >= NP) return;
    const f32x4* ap = a_frag ? (const f32x4*)A + (size_t)blockIdx.x * (K >> 4) * 64 + r16 * 4 + q
                             : (const f32x4*)(A + (size_t)(row_ok ? row : M - 1) * lda) + q;
    const size_t achunk = a_frag ? 64 : 4;
    const f32x4* wg = (const f32x4*)Wp + ((size_t)(n0 + r16) * 4 + q);  // slot (chunk, col, q) at (chunk*NP + col)*4 + q
    const size_t wchunk = (size_t)NP * 4;
    const int nchunks = K >> 4;
    constexpr int D = 4;  // chunks in flight
    f32x4 a[D], w[D][FCS_CT];
#pragma unroll
    for (int d = 0; d < D; d++) {
        const int kc = d < nchunks ? d : nchunks - 1;
        a[d] = ap[(size_t)kc * achunk];
#pragma unroll
        for (int j = 0; j < FCS_CT; j++) w[d][j] = wg[(size_t)kc * wchunk + j * 64];
    }
    f32x4 acc[FCS_CT];
#pragma unroll
    for (int j = 0; j < FCS_CT; j++) acc[j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    for (int kc0 = 0; kc0 < nchunks; kc0 += D) {
#pragma unroll
        for (int d = 0; d < D; d++) {
            if (kc0 + d < nchunks) {
#pragma unroll
                for (int t = 0; t < 4; t++)
#pragma unroll
                    for (int j = 0; j < FCS_CT; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[d][j][t], a[d][t], acc[j], 0, 0, 0);
            }
            const int kn = kc0 + d + D < nchunks ? kc0 + d + D : nchunks - 1;
            a[d] = ap[(size_t)kn * achunk];
#pragma unroll
            for (int j = 0; j < FCS_CT; j++) w[d][j] = wg[(size_t)kn * wchunk + j * 64];
        }
    }
    if (row_ok) {
#pragma unroll
        for (int j = 0; j < FCS_CT; j++) {
            const int nn = n0 + j * 16 + 4 * q;
            if (nn < n_valid) {
                f32x4 v = acc[j] + *(const f32x4*)&bias[nn];
                float* o = out + (size_t)row * out_stride + nn;
                if (nn + 3 < n_valid) *(f32x4*)o = v;
                else for (int t = 0; t < 4; t++) if (nn + t < n_valid) o[t] = v[t];
            }
        }
    }
}

// The softmax statistics of softmax.cuh from logits already in memory, for the producers that cannot emit them from their
// accumulators (k_fc_small: a wave there owns 2 output tiles, not a block's 12).  A wave covers 16 (row, block) pairs with the
// FC's own lane layout — lane = pair + 16·q holds columns col0(block) + 16 j + 4 q + t — so fc_block_stats runs unchanged (a
// single-tile block through the 12-tile template with its limit at the block's end: the same bits); the wave that handles a
// row's block 0 also copies the value pre-activation (column n_soft) into pair FC_STAT_BLOCKS of the record.
__global__ __launch_bounds__(256) void k_fc_stats(const float* __restrict__ logits, int ld, int M, int n_soft, float* __restrict__ stats) {
    const int lane = threadIdx.x & 63, r16 = lane & 15, q = lane >> 4;
    const long pair0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16;
    const long total = (long)M * FC_STAT_BLOCKS;
    if (pair0 >= total) return;
    const long pair = pair0 + r16 < total ? pair0 + r16 : total - 1;
    const int row = (int)(pair / FC_STAT_BLOCKS), b = (int)(pair - (long)row * FC_STAT_BLOCKS);
    const int col0 = fc_stat_col0(b), tiles = fc_stat_tiles(b);
    const float* x = logits + (size_t)row * ld + col0 + 4 * q;
    f32x4 v[FC_MAIN_TILES];
#pragma unroll
    for (int j = 0; j < FC_MAIN_TILES; j++) v[j] = j < tiles ? *(const f32x4*)&x[16 * j] : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    float m, sm;
    fc_block_stats<FC_MAIN_TILES>(v, col0 + 4 * q, min(n_soft, col0 + 16 * tiles), m, sm);
    if (q == 0 && pair0 + r16 < total) {
        float* srow = stats + (size_t)row * (FC_STAT_STRIDE * 2);
        *(float2*)&srow[b * 2] = make_float2(m, sm);
        if (b == 0) *(float2*)&srow[FC_STAT_BLOCKS * 2] = make_float2(logits[(size_t)row * ld + n_soft], 0.0f);
    }
}

// softmax of the FC head from the block statistics (tg_policy_eval; the search never materialises probabilities): the same
// exp(x − M) · (1 / S) the tree backup evaluates for a leaf's children, so host-side trees built from these probabilities
// and the engine's own agree bit for bit.  One block per position.
__global__ __launch_bounds__(256) void k_softmax_stats(const float* __restrict__ logits, int row_stride, const float* __restrict__ stats,
                                                       int blocks, int stat_stride, int P, float* __restrict__ policy, float* __restrict__ eval) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* x = logits + (size_t)b * row_stride;
    float mx, inv;
    fc_combine_stats(stats + (size_t)b * stat_stride * 2, blocks, mx, inv);
    if (eval && tid == 0) eval[b] = tanhf(x[P]);
    float* o = policy + (size_t)b * P;
    for (int p = tid; p < P; p += 256) o[p] = stat_exp(x[p] - mx) * inv;
}

// value head: Linear(F·N² → 1) + tanh (net5.rs:62,109 / net6.rs:57,104-107).  One wave per position;
// wv is permuted to the NHWC order of the activations.
__global__ __launch_bounds__(256) void k_value_head(const float* __restrict__ act, const float* __restrict__ wv, float bv,
                                                    int B, int len, float* __restrict__ eval) {
    int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    int lane = threadIdx.x & 63;
    const float4* a = (const float4*)(act + (size_t)b * len);
    const float4* w = (const float4*)wv;
    float s = 0.0f;
    for (int k = lane; k < (len >> 2); k += 64) {
        float4 x = a[k], y = w[k];
        s = fmaf(x.x, y.x, s);
        s = fmaf(x.y, y.y, s);
        s = fmaf(x.z, y.z, s);
        s = fmaf(x.w, y.w, s);
    }
    s = wave_sum(s);
    if (lane == 0) eval[b] = tanhf(s + bv);
}

// softmax over ALL P outputs (no legal-move mask; net5.rs:108, net6.rs:100-103).  One 256-thread block
// per position.  logits are stored [b][row_stride] with element (sq, ch) at sq*ch_stride + ch when
// conv_head (NHWC conv output) or simply [b][p] for the FC head; the probabilities are written in the
// reference's order p = ch·N² + sq.
__global__ __launch_bounds__(256) void k_softmax(const float* __restrict__ logits, int row_stride, int conv_head, int nsq,
                                                 int ch_stride, int P, float* __restrict__ policy, float* __restrict__ eval) {
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* x = logits + (size_t)b * row_stride;
    if (eval && tid == 0) eval[b] = tanhf(x[P]);  // FC head: column P of the policy FC is the value head's pre-activation
    auto at = [&](int p) -> float {
        if (!conv_head) return x[p];
        int ch = p / nsq, sq = p - ch * nsq;
        return x[sq * ch_stride + ch];
    };
    // the row is read once and kept in registers when it fits (P ≤ 8·256: the FC head's 1575 outputs)
    constexpr int KEEP = SOFTMAX_KEEP;
    const bool cached = P <= KEEP * 256;
    float v[KEEP];
    float mx = -INFINITY;
    if (cached) {
#pragma unroll
        for (int k = 0; k < KEEP; k++) {
            int p = tid + k * 256;
            v[k] = p < P ? at(p) : -INFINITY;
            mx = fmaxf(mx, v[k]);
        }
    } else {
        for (int p = tid; p < P; p += 256) mx = fmaxf(mx, at(p));
    }
    mx = wave_max(mx);
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.0f;
    if (cached) {
#pragma unroll
        for (int k = 0; k < KEEP; k++) {
            int p = tid + k * 256;
            v[k] = p < P ? expf(v[k] - mx) : 0.0f;
            s += v[k];
        }
    } else {
        for (int p = tid; p < P; p += 256) s += expf(at(p) - mx);
    }
    s = wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    s = (red[0] + red[1]) + (red[2] + red[3]);
    float inv = 1.0f / s;
    float* o = policy + (size_t)b * P;
    if (cached) {
#pragma unroll
        for (int k = 0; k < KEEP; k++) {
            int p = tid + k * 256;
            if (p < P) o[p] = v[k] * inv;
        }
    } else {
        for (int p = tid; p < P; p += 256) o[p] = expf(at(p) - mx) * inv;
    }
}

// Conv policy head (net6.rs:98-103): the logits sit in NHWC ([sq][ch_stride]) and the probabilities leave in the
// reference's order p = ch·N² + sq.  One block per position: the row is read once, coalesced, into LDS (pitch
// ch_stride + 1 so that the transposed read-out is bank-conflict free), exp is evaluated once per output.
// act != nullptr (round 6): the block's first wave also computes the position's value head — k_value_head's sum, lane for lane — so the
// conv-head forward is one launch shorter (7 µs of the 291 µs iteration at the reference's 32 leaves)
__global__ __launch_bounds__(256) void k_softmax_conv(const float* __restrict__ logits, int nsq, int ch_stride, int C,
                                                      float* __restrict__ policy, const float* __restrict__ act,
                                                      const float* __restrict__ wv, float bv, int len, float* __restrict__ eval) {
    extern __shared__ float row[];  // nsq × (ch_stride + 1)
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int pitch = ch_stride + 1;
    const int vpr = ch_stride >> 2;
    const f32x4* x4 = (const f32x4*)(logits + (size_t)b * nsq * ch_stride);
    float mx = -INFINITY;
    for (int idx = tid; idx < nsq * vpr; idx += 256) {
        int sq = idx / vpr, v = idx - sq * vpr;
        f32x4 x = x4[idx];
#pragma unroll
        for (int t = 0; t < 4; t++) {
            int ch = 4 * v + t;
            row[sq * pitch + ch] = x[t];
            if (ch < C) mx = fmaxf(mx, x[t]);
        }
    }
    mx = wave_max(mx);
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    const int P = C * nsq;
    float s = 0.0f;
    for (int p = tid; p < P; p += 256) {
        int ch = p / nsq, sq = p - ch * nsq;
        float e = expf(row[sq * pitch + ch] - mx);
        row[sq * pitch + ch] = e;
        s += e;
    }
    s = wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    s = (red[0] + red[1]) + (red[2] + red[3]);
    const float inv = 1.0f / s;
    float* o = policy + (size_t)b * P;
    for (int p = tid; p < P; p += 256) {
        int ch = p / nsq, sq = p - ch * nsq;
        o[p] = row[sq * pitch + ch] * inv;
    }
    if (act && tid < 64) {  // value head: Linear(F·N² → 1) + tanh, exactly as k_value_head (one wave per position, lane = tid)
        const float4* a = (const float4*)(act + (size_t)b * len);
        const float4* w = (const float4*)wv;
        float v = 0.0f;
        for (int k = tid; k < (len >> 2); k += 64) {
            float4 x = a[k], y = w[k];
            v = fmaf(x.x, y.x, v);
            v = fmaf(x.y, y.y, v);
            v = fmaf(x.z, y.z, v);
            v = fmaf(x.w, y.w, v);
        }
        v = wave_sum(v);
        if (tid == 0) eval[b] = tanhf(v + bv);
    }
}

// NCHW planes (the reference tensor layout) → NHWC rows padded to Cpad channels
__global__ void k_nchw_to_nhwc(const float* __restrict__ src, int B, int C, int nsq, int Cpad, float* __restrict__ dst) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)B * nsq * Cpad;
    if (idx >= total) return;
    int c = (int)(idx % Cpad);
    size_t r = idx / Cpad;
    int sq = (int)(r % nsq);
    size_t b = r / nsq;
    dst[idx] = c < C ? src[(b * C + c) * nsq + sq] : 0.0f;
}

// ---- launchers --------------------------------------------------------------------------------
size_t conv_lds_bytes(int rt, int n, int Cpad) {
    int TM = 64 * rt, nsq = n * n;
    int max_pos = (TM - 1) / nsq + 2;
    return (size_t)(max_pos * nsq + 1) * (Cpad + LDS_PAD) * sizeof(float);
}

template <int RT, int CT>
static hipError_t launch_conv_t(hipStream_t st, const float* in, const float* Wp, const float* bias, const float* res,
                                float* out, int M, int n, int Cpad, int CoutP, int out_stride, int cout_valid, bool relu) {
    size_t lds = conv_lds_bytes(RT, n, Cpad);
    static LdsAttr lds_attr;
    if (hipError_t e = lds_attr.ensure((const void*)k_conv3x3<RT, CT>, lds); e != hipSuccess) return e;
    dim3 grid((M + 64 * RT - 1) / (64 * RT), CoutP / (64 * CT));
    hipLaunchKernelGGL((k_conv3x3<RT, CT>), grid, dim3(256), lds, st, in, Wp, bias, res, out, M, n, Cpad, CoutP, out_stride,
                       cout_valid, relu ? 1 : 0);
    return hipGetLastError();
}

template <int RTW, int NWAVES>
static hipError_t launch_conv_pos_t(hipStream_t st, const float* in, const float* Wp, const float* bias, const float* res,
                                    float* out, int B, int n, int Cpad, int CoutP, int out_stride, int cout_valid, bool relu,
                                    int PW, int CTW) {
    size_t lds = (size_t)(PW * n * n + 1) * (Cpad + LDS_PAD16) * sizeof(float);
    static LdsAttr lds_attr;
    if (hipError_t e = lds_attr.ensure((const void*)k_conv_pos<RTW, NWAVES>, lds); e != hipSuccess) return e;
    dim3 grid((B + PW - 1) / PW, CoutP / (CTW * 16));
    hipLaunchKernelGGL((k_conv_pos<RTW, NWAVES>), grid, dim3(NWAVES * 64), lds, st, in, Wp, bias, res, out, B, n, Cpad, CoutP,
                       out_stride, cout_valid, relu ? 1 : 0, PW, CTW);
    return hipGetLastError();
}

// slot table of the halo image for (n, F) on the current device, built on first use (launch_conv3x3's halo path; the fused
// tower carries its own copy in TowerParams)
static const uint32_t* conv_halo_slotmap(int n, int F, int pw, int ps) {
    struct Entry { int dev, n, F, pw, ps; uint32_t* d; };
    static std::vector<Entry> cache;
    static std::mutex guard;  // trainers of several engines may run on several host threads (data-parallel tests)
    std::lock_guard<std::mutex> lock(guard);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    for (const Entry& e : cache) if (e.dev == dev && e.n == n && e.F == F && e.pw == pw && e.ps == ps) return e.d;
    std::vector<uint32_t> map((size_t)((pw * n * n + 15) / 16) * 16);
    tower_halo_slotmap(n, pw, ps, map.data());
    uint32_t* d = nullptr;
    if (hipMalloc((void**)&d, map.size() * 4) != hipSuccess) return nullptr;
    if (hipMemcpy(d, map.data(), map.size() * 4, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return nullptr; }
    cache.push_back({dev, n, F, pw, ps, d});
    return d;
}

template <int RTW, int NWAVES, int CH, int NB, int COT, int PSC>
static hipError_t launch_conv_halo_t(hipStream_t st, const float* in, const float* Wp, const float* bias, const float* res, float* out,
                                     const uint32_t* slotmap, int B, int PW, int PS, int CTW, int out_stride, int cout_valid, bool relu,
                                     double* stats_part, int* stats_blocks, const ConvBnBwdIn* bnb) {
    if (PS != PSC) return hipErrorInvalidValue;
    const size_t lds = (size_t)(NB + 2 + PW * PS + 1) * (16 * CH + 4) * sizeof(float);
    static LdsAttr lds_attr;
    if (hipError_t e = lds_attr.ensure((const void*)k_conv_halo<RTW, NWAVES, CH, NB, COT, PSC>, lds); e != hipSuccess) return e;
    dim3 grid((B + PW - 1) / PW, COT / CTW);
    if (grid.y != 1) stats_part = nullptr;  // (statistics only for layers whose channels one workgroup column covers)
    ConvBnBwdIn bn{nullptr, nullptr, nullptr, nullptr};
    if (bnb && stats_part && out_stride == 16 * COT) bn = *bnb;  // (y and z share the output's row layout)
    else if (bnb) stats_part = nullptr;
    hipLaunchKernelGGL((k_conv_halo<RTW, NWAVES, CH, NB, COT, PSC>), grid, dim3(NWAVES * 64), lds, st, in, Wp, bias, res, out, slotmap, B, PW, PS,
                       CTW, out_stride, cout_valid, relu ? 1 : 0, stats_part, bn);
    if (stats_blocks) *stats_blocks = stats_part ? (int)grid.x * (NWAVES / CTW) : 0;
    return hipGetLastError();
}

hipError_t launch_conv3x3(hipStream_t st, const float* in, const float* Wp, const float* bias, const float* res, float* out,
                          int M, int n, int Cpad, int CoutP, int out_stride, int cout_valid, bool relu, double* stats_part,
                          int* stats_blocks, const ConvBnBwdIn* bnb) {
    const int B = M / (n * n);
    if (stats_blocks) *stats_blocks = 0;
    {   // F → F (and F → 2F) layers of the BASELINE topologies at full batches: the halo image (k_conv_halo), same bits as k_conv_pos
        static const bool off = env_on("TG_NO_HALO_CONV");
        int pw, ps;
        if (!off && B >= 1024 && tower_halo_geometry(n, Cpad, &pw, &ps)) {
            const uint32_t* map = conv_halo_slotmap(n, Cpad, pw, ps);
            if (map) {
                if (n == 5 && Cpad == 64 && CoutP == 64) return launch_conv_halo_t<13, 8, 4, 5, 4, 36>(st, in, Wp, bias, res, out, map, B, pw, ps, 4, out_stride, cout_valid, relu, stats_part, stats_blocks, bnb);
                if (n == 5 && Cpad == 128 && CoutP == 128) return launch_conv_halo_t<13, 8, 8, 5, 8, 37>(st, in, Wp, bias, res, out, map, B, pw, ps, 8, out_stride, cout_valid, relu, stats_part, stats_blocks, bnb);
                if (n == 6 && Cpad == 128 && CoutP == 128) return launch_conv_halo_t<9, 8, 8, 6, 8, 51>(st, in, Wp, bias, res, out, map, B, pw, ps, 8, out_stride, cout_valid, relu, stats_part, stats_blocks, bnb);
                if (n == 6 && Cpad == 128 && CoutP == 256) return launch_conv_halo_t<9, 8, 8, 6, 16, 51>(st, in, Wp, bias, res, out, map, B, pw, ps, 8, out_stride, cout_valid, relu, stats_part, stats_blocks, bnb);
            }
        }
    }
    {   // small batches of a layer without residual and statistics (the conv policy head): (position, channel tile) workgroups
        static const bool off = env_on("TG_NO_SPLIT_TOWER");
        if (!off && !res && !stats_part && !bnb && B >= 1 && B <= TOWER_SPLIT_MAX_BATCH && Cpad == 128 && CoutP % 16 == 0 && (n == 5 || n == 6)) {
            const size_t lds = (size_t)(n * n + 1) * (128 + LDS_PAD16) * sizeof(float);
            const dim3 grid(B, CoutP / 16);
            if (n == 6) hipLaunchKernelGGL((k_conv_split<3, 8>), grid, dim3(192), lds, st, in, Wp, bias, out, n, CoutP, out_stride, cout_valid, relu ? 1 : 0);
            else hipLaunchKernelGGL((k_conv_split<2, 8>), grid, dim3(128), lds, st, in, Wp, bias, out, n, CoutP, out_stride, cout_valid, relu ? 1 : 0);
            return hipGetLastError();
        }
    }
    // whole-positions kernel where the shape divides evenly (the BASELINE configs); generic tiles otherwise
    // small batches take fewer positions per workgroup (shorter critical path, same bits — see launch_tower)
#define TG_CONV_POS(RTW, NW, PW, CTW) \
    return launch_conv_pos_t<RTW, NW>(st, in, Wp, bias, res, out, B, n, Cpad, CoutP, out_stride, cout_valid, relu, PW, CTW)
    if (n == 5 && CoutP == 64 && Cpad <= 80) {  // 16 positions = 25 row tiles, 4 channel tiles × 2 row groups of 13
        if (B <= 256) TG_CONV_POS(2, 4, 1, 4);
        if (B <= 512) TG_CONV_POS(4, 4, 2, 4);
        if (B <= 1024) TG_CONV_POS(7, 4, 4, 4);
        if (B <= 2048) TG_CONV_POS(13, 4, 8, 4);
        TG_CONV_POS(13, 8, 16, 4);
    }
    if (n == 6 && CoutP % 128 == 0 && Cpad <= 128) {  // 4 positions = 9 row tiles, 8 channel tiles
        if (B <= 256) TG_CONV_POS(3, 8, 1, 8);
        if (B <= 512) TG_CONV_POS(5, 8, 2, 8);
        TG_CONV_POS(9, 8, 4, 8);
    }
    if (n == 5 && CoutP % 128 == 0 && Cpad <= 128) {  // 8 positions = 200 rows in 13 row tiles, 8 channel tiles
        if (B <= 256) TG_CONV_POS(2, 8, 1, 8);
        if (B <= 512) TG_CONV_POS(4, 8, 2, 8);
        if (B <= 1024) TG_CONV_POS(7, 8, 4, 8);
        TG_CONV_POS(13, 8, 8, 8);
    }
#undef TG_CONV_POS
    if (conv_lds_bytes(2, n, Cpad) > 160 * 1024) {  // wide inputs (data gradient of the 6×6 policy head): 64-row tiles
        if (CoutP % 128 == 0) return launch_conv_t<1, 2>(st, in, Wp, bias, res, out, M, n, Cpad, CoutP, out_stride, cout_valid, relu);
        return launch_conv_t<1, 1>(st, in, Wp, bias, res, out, M, n, Cpad, CoutP, out_stride, cout_valid, relu);
    }
    if (CoutP % 128 == 0) return launch_conv_t<2, 2>(st, in, Wp, bias, res, out, M, n, Cpad, CoutP, out_stride, cout_valid, relu);
    return launch_conv_t<2, 1>(st, in, Wp, bias, res, out, M, n, Cpad, CoutP, out_stride, cout_valid, relu);
}


// bytes of the per-position bias table PB behind the layer-0 image (TowerParams.cb)
static size_t tower_cb_table_bytes(int PW, int F) { return (size_t)PW * 9 * F * sizeof(float); }

template <int RTW, int NWAVES, int CH0, int CH, bool FROM_STATES, bool CB = false>
static hipError_t launch_tower_t(hipStream_t st, const float* in, const TowerParams& T, float* out, int B, int n, int PW, int CTW) {
    const size_t rows1 = (size_t)(PW * n * n + 1);
    const size_t first = rows1 * ((CB ? T.cb_cin_pad : T.cin_pad) + LDS_PAD16) * sizeof(float) + (CB ? tower_cb_table_bytes(PW, T.F) : 0);
    const size_t later = rows1 * (T.F + LDS_PAD16) * sizeof(float);
    const size_t lds = first > later ? first : later;
    static LdsAttr lds_attr;
    if (hipError_t e = lds_attr.ensure((const void*)k_tower<RTW, NWAVES, CH0, CH, FROM_STATES, CB>, lds); e != hipSuccess) return e;
    hipLaunchKernelGGL((k_tower<RTW, NWAVES, CH0, CH, FROM_STATES, CB>), dim3((B + PW - 1) / PW), dim3(NWAVES * 64), lds, st, in, T, out, B, n, PW, CTW);
    return hipGetLastError();
}


// ---- halo image (k_tower_halo): geometry and the square → tile-slot table -------------------------
bool tower_halo_geometry(int n, int F, int* pw, int* ps) {
    // position strides found by simulating the ds_read_b128 bank groups over all tiles / taps (conflict factor ≤ 1.11)
    if (n == 5 && F == 64) { *pw = 16; *ps = 36; return true; }   // 158 848 B of LDS (with the spare cell)
    if (n == 5 && F == 128) { *pw = 8; *ps = 37; return true; }   // 160 512 B
    if (n == 6 && F == 128) { *pw = 4; *ps = 51; return true; }   // 112 464 B
    return false;
}

void tower_halo_slotmap(int n, int pw, int ps, uint32_t* out) {
    // ds_read_b128 serves lanes {0-3, 12-15} of one 16-byte slot q together with lanes {4-11} of slot q + 1 (and vice
    // versa).  With a pitch of F + 4 floats the bank quad of a read is (cell + q + 4·chunk) mod 16, so a tile whose 8
    // "outer" lanes and 8 "inner" lanes each hold 8 cells with distinct residues of ONE parity is conflict free for every
    // tap and chunk (a tap shifts all cells alike).  Greedy: per tile take one square per residue of the richer parity.
    const int nsq = n * n, RS = n + 1, LEAD = n + 2, rows = pw * nsq, tiles = (rows + 15) / 16;
    std::vector<std::vector<int>> bucket(16);
    std::vector<int> cell(rows);
    for (int r = rows - 1; r >= 0; r--) {
        const int p = r / nsq, sq = r % nsq;
        cell[r] = LEAD + p * ps + (sq / n) * RS + sq % n;
        bucket[cell[r] % 16].push_back(r);
    }
    static const int outer[8] = {0, 1, 2, 3, 12, 13, 14, 15}, inner[8] = {4, 5, 6, 7, 8, 9, 10, 11};
    for (int t = 0; t < tiles; t++) {
        size_t left[2] = {0, 0};
        for (int r = 0; r < 16; r++) left[r & 1] += bucket[r].size();
        const int par = left[0] >= left[1] ? 0 : 1;
        for (const int* slots : {outer, inner}) {
            int missing[8], nm = 0;
            for (int k = 0; k < 8; k++) {
                std::vector<int>& b = bucket[2 * k + par];
                if (!b.empty()) { out[t * 16 + slots[k]] = (uint32_t)cell[b.back()] | ((uint32_t)b.back() << 16); b.pop_back(); }
                else missing[nm++] = slots[k];
            }
            for (int m = 0; m < nm; m++) {
                int big = 0;
                for (int r = 1; r < 16; r++) if (bucket[r].size() > bucket[big].size()) big = r;
                if (!bucket[big].empty()) {
                    out[t * 16 + missing[m]] = (uint32_t)cell[bucket[big].back()] | ((uint32_t)bucket[big].back() << 16);
                    bucket[big].pop_back();
                } else out[t * 16 + missing[m]] = 0xFFFF0000u;  // no square left: the slot idles
            }
        }
    }
}

template <int RTW, int NWAVES, int CH0, int CH, int NB, bool FROM_STATES, bool CB = false>
static hipError_t launch_tower_halo_t(hipStream_t st, const float* in, const TowerParams& T, float* out, int B, int CTW) {
    const int PW = T.halo_pw;
    const size_t plain = (size_t)(PW * NB * NB + 1) * ((CB ? T.cb_cin_pad : T.cin_pad) + LDS_PAD16) * sizeof(float) +
                         (CB ? tower_cb_table_bytes(PW, 16 * CH) : 0);
    const size_t halo = (size_t)(NB + 2 + PW * T.halo_ps + 1) * (16 * CH + 4) * sizeof(float);  // + the spare cell
    const size_t lds = plain > halo ? plain : halo;
    static LdsAttr lds_attr;
    if (hipError_t e = lds_attr.ensure((const void*)k_tower_halo<RTW, NWAVES, CH0, CH, NB, FROM_STATES, CB>, lds); e != hipSuccess) return e;
    hipLaunchKernelGGL((k_tower_halo<RTW, NWAVES, CH0, CH, NB, FROM_STATES, CB>), dim3((B + PW - 1) / PW), dim3(NWAVES * 64), lds, st, in, T, out, B, PW, CTW);
    return hipGetLastError();
}

// full batches of the three BASELINE topologies run on the halo image (identical bits, see k_tower_halo)
template <bool FROM_STATES>
static bool launch_tower_halo(hipStream_t st, const float* in, const TowerParams& T, float* out, int B, int n, hipError_t* err) {
    static const bool off = env_on("TG_NO_HALO_TOWER");
    if (off || !T.slotmap) return false;
    if (FROM_STATES && T.cb) {  // layer 0 over the board planes, constant planes as a bias (CH0 = 2)
        if (n == 5 && T.F == 64 && B > 2048) { *err = launch_tower_halo_t<13, 8, 2, 4, 5, FROM_STATES, FROM_STATES>(st, in, T, out, B, 4); return true; }
        if (n == 6 && T.F == 128 && B > 512) { *err = launch_tower_halo_t<9, 8, 2, 8, 6, FROM_STATES, FROM_STATES>(st, in, T, out, B, 8); return true; }
        if (n == 5 && T.F == 128 && B > 1024) { *err = launch_tower_halo_t<13, 8, 2, 8, 5, FROM_STATES, FROM_STATES>(st, in, T, out, B, 8); return true; }
        return false;
    }
    if (n == 5 && T.F == 64 && T.cin_pad == 80 && B > 2048) { *err = launch_tower_halo_t<13, 8, 5, 4, 5, FROM_STATES>(st, in, T, out, B, 4); return true; }
    if (n == 6 && T.F == 128 && T.cin_pad == 96 && B > 512) { *err = launch_tower_halo_t<9, 8, 6, 8, 6, FROM_STATES>(st, in, T, out, B, 8); return true; }
    if (n == 5 && T.F == 128 && T.cin_pad == 80 && B > 1024) { *err = launch_tower_halo_t<13, 8, 5, 8, 5, FROM_STATES>(st, in, T, out, B, 8); return true; }
    return false;
}

bool tower_supported(int n, int F, int cin_pad) {
    if (n == 5 && F == 64 && cin_pad == 80) return true;   // config C2
    if (n == 6 && F == 128 && cin_pad == 96) return true;  // config C3
    if (n == 5 && F == 128 && cin_pad == 80) return true;  // config C5 network
    return false;
}

hipError_t launch_tower(hipStream_t st, const float* in, const TowerParams& T, float* out, int B, int n) {
    {
        hipError_t herr;
        if (launch_tower_halo<false>(st, in, T, out, B, n, &herr)) return herr;
    }
    // A workgroup's run time is that of its positions' row tiles, whatever the batch: with 16 positions per workgroup a
    // 32-position call (the reference's BATCH_SIZE) ran 2 workgroups for as long as 4096 positions take.  Small batches
    // therefore use instantiations with fewer positions (row tiles) per workgroup; the per-element arithmetic — taps,
    // chunks, MFMA k-steps in the same order — does not depend on the tiling, so results are bit-identical.
    if (n == 5 && T.F == 64 && T.cin_pad == 80) {
        if (B <= 256) return launch_tower_t<2, 4, 5, 4, false>(st, in, T, out, B, n, 1, 4);
        if (B <= 512) return launch_tower_t<4, 4, 5, 4, false>(st, in, T, out, B, n, 2, 4);
        if (B <= 1024) return launch_tower_t<7, 4, 5, 4, false>(st, in, T, out, B, n, 4, 4);
        if (B <= 2048) return launch_tower_t<13, 4, 5, 4, false>(st, in, T, out, B, n, 8, 4);
        return launch_tower_t<13, 8, 5, 4, false>(st, in, T, out, B, n, 16, 4);
    }
    if (n == 6 && T.F == 128 && T.cin_pad == 96) {
        if (B <= 256) return launch_tower_t<3, 8, 6, 8, false>(st, in, T, out, B, n, 1, 8);
        if (B <= 512) return launch_tower_t<5, 8, 6, 8, false>(st, in, T, out, B, n, 2, 8);
        return launch_tower_t<9, 8, 6, 8, false>(st, in, T, out, B, n, 4, 8);
    }
    if (n == 5 && T.F == 128 && T.cin_pad == 80) {
        if (B <= 256) return launch_tower_t<2, 8, 5, 8, false>(st, in, T, out, B, n, 1, 8);
        if (B <= 512) return launch_tower_t<4, 8, 5, 8, false>(st, in, T, out, B, n, 2, 8);
        if (B <= 1024) return launch_tower_t<7, 8, 5, 8, false>(st, in, T, out, B, n, 4, 8);
        return launch_tower_t<13, 8, 5, 8, false>(st, in, T, out, B, n, 8, 8);
    }
    return hipErrorInvalidValue;
}

// Does workgroup id i of a 1-D grid run on XCD i mod 8 on this device (every XCD its own L2)?  64 workgroups report HW_REG_XCC_ID.
__global__ void k_xcc_probe(unsigned* __restrict__ out) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (threadIdx.x == 0) out[blockIdx.x] = x & 15u;
}
static bool workgroups_round_robin_over_xcds() {
    static std::mutex guard;
    static int verdict[16] = {};  // per device: 0 unknown, 1 yes, 2 no
    std::lock_guard<std::mutex> lock(guard);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return false;
    if (verdict[dev]) return verdict[dev] == 1;
    verdict[dev] = 2;
    unsigned* d = nullptr;
    unsigned h[64];
    if (hipMalloc((void**)&d, sizeof(h)) != hipSuccess) return false;
    bool ok = true;
    for (int rep = 0; rep < 4 && ok; rep++) {  // (a fresh launch every time: the mapping must not depend on what ran before)
        hipLaunchKernelGGL(k_xcc_probe, dim3(64), dim3(64), 0, nullptr, d);
        ok = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess;
        for (int i = 8; i < 64 && ok; i++) ok = h[i] == h[i & 7];
    }
    (void)hipFree(d);
    if (ok) verdict[dev] = 1;
    return ok;
}

template <int NRT, int CTW, int CH>
static hipError_t launch_tower_split_t(hipStream_t st, const uint8_t* states, const TowerParams& T, float* out, float* scratch, int B, int n) {
    const int nsq = n * n, F = 16 * CH;
    const size_t first = (size_t)(nsq + 1) * (T.cb_cin_pad + LDS_PAD16) * sizeof(float) + tower_cb_table_bytes(1, F);
    const size_t later = (size_t)(nsq + 1) * (F + LDS_PAD16) * sizeof(float);
    const size_t lds = first > later ? first : later;
    constexpr int G = CH / CTW;
    const dim3 grid((B + 7) / 8 * 8 * G), block(NRT * CTW * 64);
    static const bool agent_fences = env_on("TG_SPLIT_AGENT_FENCES");  // A/B: the exchange with agent-scope fences wherever the siblings sit (same bits)
    if (!agent_fences && workgroups_round_robin_over_xcds()) {
        static LdsAttr lds_attr;
        if (hipError_t e = lds_attr.ensure((const void*)k_tower_split<NRT, CTW, CH, true>, lds); e != hipSuccess) return e;
        hipLaunchKernelGGL((k_tower_split<NRT, CTW, CH, true>), grid, block, lds, st, states, T, out, scratch, B, n);
    } else {
        static LdsAttr lds_attr;
        if (hipError_t e = lds_attr.ensure((const void*)k_tower_split<NRT, CTW, CH, false>, lds); e != hipSuccess) return e;
        hipLaunchKernelGGL((k_tower_split<NRT, CTW, CH, false>), grid, block, lds, st, states, T, out, scratch, B, n);
    }
    return hipGetLastError();
}

// small batches of the 128-filter networks: a position split over 8 / 4 / 2 workgroups by channel tile (k_tower_split; identical bits);
// scratch = TowerParams.split_buf's two exchange buffers
static bool launch_tower_split(hipStream_t st, const uint8_t* states, const TowerParams& T, float* out, float* scratch, int B, int n,
                               hipError_t* err) {
    static const bool off = env_on("TG_NO_SPLIT_TOWER");
    if (off || !scratch || !T.cb || T.cb_last_t != 3 || T.cb_cin_pad != 32 || !T.split_flags || B > TOWER_SPLIT_MAX_BATCH || (T.nlayers & 1) == 0)
        return false;
    if (n == 5 && T.F == 64) { *err = launch_tower_split_t<2, 1, 4>(st, states, T, out, scratch, B, n); return true; }  // G = 4
    if (T.F != 128) return false;
    // one (row tile, channel tile) pair per wave, G = 8 workgroups per position: 3-wave workgroups (6×6) fit twice on a CU at their
    // 200 registers, 2-wave ones (5×5) four times — 512 / 1024 resident workgroups.  (2 and 4 channel tiles per workgroup — G = 4, 2 —
    // were measured for the batches in between: a 12-wave workgroup per position pair gains nothing over k_tower.)
    if (n == 6 && B <= TOWER_SPLIT_MAX_BATCH / 2) { *err = launch_tower_split_t<3, 1, 8>(st, states, T, out, scratch, B, n); return true; }
    if (n == 5) { *err = launch_tower_split_t<2, 1, 8>(st, states, T, out, scratch, B, n); return true; }
    return false;
}

// same, with the input planes encoded in-kernel from packed game states.  scratch (optional): a second activation buffer of the
// batch's size — with it small batches of wide networks run split by channel tile (k_tower_split)
hipError_t launch_tower_states(hipStream_t st, const uint8_t* states, const TowerParams& T, float* out, int B, int n, float* scratch) {
    const float* in = (const float*)states;
    {
        hipError_t herr;
        if (launch_tower_halo<true>(st, in, T, out, B, n, &herr)) return herr;
        if (launch_tower_split(st, states, T, out, scratch, B, n, &herr)) return herr;
    }
    if (T.cb) {  // the same tilings with layer 0 over the board planes (identical bits for every batch size)
        if (n == 5 && T.F == 64) {
            // (round 6, measured at 300 … 2048 positions with 1 / 2 / 4 / 8 positions per workgroup: these brackets are within 11 % of the
            // best choice everywhere — two co-resident workgroups of half the size take as long as one; eight waves instead of four: 5 %;
            // every layer streaming the same L2-hot weights: no difference — profiles/r06_e_tower_pw_sweep.txt)
            if (B <= 256) return launch_tower_t<2, 4, 2, 4, true, true>(st, in, T, out, B, n, 1, 4);
            if (B <= 512) return launch_tower_t<4, 4, 2, 4, true, true>(st, in, T, out, B, n, 2, 4);
            if (B <= 1024) return launch_tower_t<7, 4, 2, 4, true, true>(st, in, T, out, B, n, 4, 4);
            if (B <= 2048) return launch_tower_t<13, 4, 2, 4, true, true>(st, in, T, out, B, n, 8, 4);
            return launch_tower_t<13, 8, 2, 4, true, true>(st, in, T, out, B, n, 16, 4);
        }
        if (n == 6 && T.F == 128) {
            if (B <= 256) return launch_tower_t<3, 8, 2, 8, true, true>(st, in, T, out, B, n, 1, 8);
            if (B <= 512) return launch_tower_t<5, 8, 2, 8, true, true>(st, in, T, out, B, n, 2, 8);
            return launch_tower_t<9, 8, 2, 8, true, true>(st, in, T, out, B, n, 4, 8);
        }
        if (n == 5 && T.F == 128) {
            if (B <= 256) return launch_tower_t<2, 8, 2, 8, true, true>(st, in, T, out, B, n, 1, 8);
            if (B <= 512) return launch_tower_t<4, 8, 2, 8, true, true>(st, in, T, out, B, n, 2, 8);
            if (B <= 1024) return launch_tower_t<7, 8, 2, 8, true, true>(st, in, T, out, B, n, 4, 8);
            return launch_tower_t<13, 8, 2, 8, true, true>(st, in, T, out, B, n, 8, 8);
        }
        return hipErrorInvalidValue;
    }
    if (n == 5 && T.F == 64 && T.cin_pad == 80) {
        // fewer positions per workgroup for small batches (see tower_small_batch below): identical bits, shorter critical path
        if (B <= 256) return launch_tower_t<2, 4, 5, 4, true>(st, in, T, out, B, n, 1, 4);
        if (B <= 512) return launch_tower_t<4, 4, 5, 4, true>(st, in, T, out, B, n, 2, 4);
        if (B <= 1024) return launch_tower_t<7, 4, 5, 4, true>(st, in, T, out, B, n, 4, 4);
        if (B <= 2048) return launch_tower_t<13, 4, 5, 4, true>(st, in, T, out, B, n, 8, 4);
        static const int variant = env_int("TG_TOWER_VARIANT");
        if (variant == 16) return launch_tower_t<7, 16, 5, 4, true>(st, in, T, out, B, n, 16, 4);
        if (variant == 4) return launch_tower_t<25, 4, 5, 4, true>(st, in, T, out, B, n, 16, 4);
        return launch_tower_t<13, 8, 5, 4, true>(st, in, T, out, B, n, 16, 4);
    }
    if (n == 6 && T.F == 128 && T.cin_pad == 96) {
        if (B <= 256) return launch_tower_t<3, 8, 6, 8, true>(st, in, T, out, B, n, 1, 8);
        if (B <= 512) return launch_tower_t<5, 8, 6, 8, true>(st, in, T, out, B, n, 2, 8);
        return launch_tower_t<9, 8, 6, 8, true>(st, in, T, out, B, n, 4, 8);
    }
    if (n == 5 && T.F == 128 && T.cin_pad == 80) {
        if (B <= 256) return launch_tower_t<2, 8, 5, 8, true>(st, in, T, out, B, n, 1, 8);
        if (B <= 512) return launch_tower_t<4, 8, 5, 8, true>(st, in, T, out, B, n, 2, 8);
        if (B <= 1024) return launch_tower_t<7, 8, 5, 8, true>(st, in, T, out, B, n, 4, 8);
        return launch_tower_t<13, 8, 5, 8, true>(st, in, T, out, B, n, 8, 8);
    }
    return hipErrorInvalidValue;
}

// the FC kernels of the 5×5 policy head: K-steps of 64, the 99 tiles of softmax.cuh's geometry inside NP columns
static bool fc_shape_ok(int K, int NP) { return K % FC_KSTEP == 0 && NP >= FC_TILES * 16 && NP % (FCS_CT * 16) == 0; }
bool fc_frag_supported(int K, int NP) { return fc_shape_ok(K, NP); }
bool fc_stats_supported(int K, int NP, int out_stride) { return fc_shape_ok(K, NP) && out_stride == NP; }
bool fc_gather_supported(int M, int K, int NP) { return fc_shape_ok(K, NP) && M > FC_SMALL_ROWS; }

hipError_t launch_gemm(hipStream_t st, const float* A, int lda, const float* Wp, const float* bias, float* out, int M, int K,
                       int NP, int out_stride, int n_valid, bool a_frag, float* stats, int n_soft, const FcGatherArgs* gather,
                       const float* Wlin) {
    if (a_frag && !fc_frag_supported(K, NP)) return hipErrorInvalidValue;
    if (stats && (!fc_stats_supported(K, NP, out_stride) || n_valid > FC_TILES * 16)) return hipErrorInvalidValue;
    if (gather && (!stats || !fc_gather_supported(M, K, NP))) return hipErrorInvalidValue;
    const bool fc = fc_shape_ok(K, NP) && n_valid <= FC_TILES * 16;
    if (fc && M <= FC_SMALL_ROWS) {
        dim3 grid((M + 15) / 16, (NP / (FCS_CT * 16) + 3) / 4);
        hipLaunchKernelGGL(k_fc_small, grid, dim3(256), 0, st, A, lda, Wp, bias, out, M, K, NP, out_stride, n_valid, a_frag ? 1 : 0);
        if (stats) {  // (columns ≥ n_valid of `out` are never written by any FC kernel and never enter the statistics: n_soft < n_valid)
            const long pairs = (long)M * FC_STAT_BLOCKS;
            hipLaunchKernelGGL(k_fc_stats, dim3((unsigned)((pairs + 63) / 64)), dim3(256), 0, st, out, out_stride, M, n_soft, stats);
        }
        return hipGetLastError();
    }
    if (fc) {
        static LdsAttr lds_attr;
        if (hipError_t e = lds_attr.ensure((const void*)k_fc_ring<0>, FC_RING_LDS); e != hipSuccess) return e;
        FcGather g{nullptr, nullptr, nullptr, 0};
        if (gather) g = FcGather{gather->child_pidx, gather->leaf_rec, gather->child_logit, gather->stride};
        hipLaunchKernelGGL(k_fc_ring<0>, dim3((M + 127) / 128, FC_MAIN_BLOCKS), dim3(512), FC_RING_LDS, st, A, lda, Wp, bias, gather ? nullptr : out, M, K, NP,
                           out_stride, n_valid, a_frag ? 1 : 0, stats, n_soft, g, Wlin);
        return hipGetLastError();
    }
    // Round 4: plain row-major GEMMs whose 25x output tiles split into 2x blocks of 12 + x leftover tiles take the ring too — the FC
    // head's data gradient in the training step (dlogits[4000 × 1600] · Wᵀ → 3200 columns = 200 tiles): 465 µs in k_gemm below
    static const bool ring_gemm = !env_on("TG_NO_RING_GEMM");
    if (ring_gemm && K % FC_KSTEP == 0 && NP % 400 == 0 && M > FC_SMALL_ROWS && !a_frag && !stats && !gather) {
        static LdsAttr lds_attr;
        if (hipError_t e = lds_attr.ensure((const void*)k_fc_ring<1>, FC_RING_LDS); e != hipSuccess) return e;
        hipLaunchKernelGGL(k_fc_ring<1>, dim3((M + 127) / 128, NP / 200), dim3(512), FC_RING_LDS, st, A, lda, Wp, bias, out, M, K, NP, out_stride, n_valid, 0,
                           nullptr, 0, FcGather{nullptr, nullptr, nullptr, 0}, nullptr);
        return hipGetLastError();
    }
    dim3 grid((M + 127) / 128, NP / 64);
    hipLaunchKernelGGL((k_gemm<2, 1>), grid, dim3(256), 0, st, A, lda, Wp, bias, out, M, K, NP, out_stride, n_valid);
    return hipGetLastError();
}

hipError_t launch_fc_stats(hipStream_t st, const float* logits, int ld, int M, int n_soft, float* stats) {
    const long pairs = (long)M * FC_STAT_BLOCKS;
    hipLaunchKernelGGL(k_fc_stats, dim3((unsigned)((pairs + 63) / 64)), dim3(256), 0, st, logits, ld, M, n_soft, stats);
    return hipGetLastError();
}

hipError_t launch_value_head(hipStream_t st, const float* act, const float* wv, float bv, int B, int len, float* eval) {
    hipLaunchKernelGGL(k_value_head, dim3((B + 3) / 4), dim3(256), 0, st, act, wv, bv, B, len, eval);
    return hipGetLastError();
}

// value (optional, conv head only): the value head's inputs — when the transposing kernel takes the batch it computes the eval too and
// *value_done is set; otherwise the caller launches k_value_head as before
hipError_t launch_softmax(hipStream_t st, const float* logits, int row_stride, bool conv_head, int nsq, int ch_stride, int P,
                          int B, float* policy, float* eval, const ValueHeadArgs* value, bool* value_done) {
    if (value_done) *value_done = false;
    if (conv_head && (ch_stride & 3) == 0 && (size_t)nsq * (ch_stride + 1) * 4 <= 64 * 1024) {
        const bool fuse = value && value->act && value->eval && (value->len & 3) == 0;
        hipLaunchKernelGGL(k_softmax_conv, dim3(B), dim3(256), (size_t)nsq * (ch_stride + 1) * 4, st, logits, nsq, ch_stride, P / nsq, policy,
                           fuse ? value->act : nullptr, fuse ? value->wv : nullptr, fuse ? value->bv : 0.0f, fuse ? value->len : 0,
                           fuse ? value->eval : nullptr);
        if (value_done) *value_done = fuse;
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_softmax, dim3(B), dim3(256), 0, st, logits, row_stride, conv_head ? 1 : 0, nsq, ch_stride, P, policy, conv_head ? nullptr : eval);
    return hipGetLastError();
}

hipError_t launch_softmax_stats(hipStream_t st, const float* logits, int row_stride, const float* stats, int blocks, int stat_stride, int P,
                                int B, float* policy, float* eval) {
    hipLaunchKernelGGL(k_softmax_stats, dim3(B), dim3(256), 0, st, logits, row_stride, stats, blocks, stat_stride, P, policy, eval);
    return hipGetLastError();
}

hipError_t launch_nchw_to_nhwc(hipStream_t st, const float* src, int B, int C, int nsq, int Cpad, float* dst) {
    size_t total = (size_t)B * nsq * Cpad;
    hipLaunchKernelGGL(k_nchw_to_nhwc, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, src, B, C, nsq, Cpad, dst);
    return hipGetLastError();
}

}  // namespace tg
