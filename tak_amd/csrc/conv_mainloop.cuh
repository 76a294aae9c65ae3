// conv_mainloop.cuh — the MFMA main loop of the fused tower (shared with scripts/probes/mainloop_probe.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tg {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int LDS_PAD = 4;  // floats appended to every LDS row of the 32x32x2 kernels: consecutive rows shift by 4 banks
// 16x16 operands (lane = row r16 + 16·q, q-th 16-byte slot of the chunk): ds_read_b128 is served in the lane groups
// {0-3,12-15,20-27}, {4-11,16-19,28-31}, … (MI355X_MICROARCH.md §LDS), i.e. 8 rows of one q with 8 rows of the next.  A
// pitch of C+4 floats makes rows 12-15 of one q collide with rows 10-11 of the other (every read 2-way: half of the LDS
// cycles were conflicts); with C+8 floats the rows of a group land on the even 16-byte bank groups and the other q on the
// odd ones: conflict free.
constexpr int LDS_PAD16 = 8;

// What the MFMA pipe is sensitive to (scripts/probes/, MI355X): a per-chunk global weight load consumed in the
// next iteration costs 11 % (hipcc sinks it next to its use), bursts of address arithmetic at every tap that the
// two waves of a SIMD execute in lock-step cost 12 %.  So: the 9-bit tap-validity mask of every row is computed
// once per kernel, a tap switch costs 3 VALU per tile, chunk offsets are ds_read immediates (CH is a compile-time
// constant), the weights run two chunks ahead of the MFMAs, and the row tiles are split in two halves that are
// reloaded in place while the other half's MFMAs issue.

// 9-bit tap-validity mask of the row tiles of one lane: bit t set ⇔ tap t (dy = t/3-1, dx = t%3-1) of that row
// stays on the board; rows ≥ `rows` get 0 (every tap reads the zero row).
template <int RTW>
__device__ __forceinline__ void conv_tap_masks(int rows, int n, int nsq, int rho0, int (&vmask)[RTW]) {
#pragma unroll
    for (int j = 0; j < RTW; j++) {
        int rho = rho0 + j * 16;
        int p = rho / nsq;
        int sq = rho - p * nsq;
        int y = sq / n, x = sq - y * n;
        int m = 0;
#pragma unroll
        for (int t = 0; t < 9; t++) {
            int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
            if (yy >= 0 && yy < n && xx >= 0 && xx < n) m |= 1 << t;
        }
        vmask[j] = rho < rows ? m : 0;
    }
}

// Layer 0 of the fused towers: C_in = 72 / 92 real channels in 80 / 96 padded ones.  Position p = 4q + t of a 16-channel
// chunk is element t of lane group q's 16 bytes, i.e. MFMA t of the chunk — so the r = 8 / 12 real channels of the LAST
// chunk are stored at q = i / LT, t = i % LT (LT = ⌈r/4⌉ = 2 / 3) and MFMAs t ≥ LT see nothing but padding.  The same
// permutation is applied to the layer's weights (net.hip upload_conv).  src = the chunk's four quads in channel order.
template <int LT>
__device__ __forceinline__ void conv_last_chunk_perm(const f32x4 (&src)[4], f32x4 (&dst)[4]) {
#pragma unroll
    for (int qq = 0; qq < 4; qq++)
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int i = qq * LT + t;  // compile-time after unrolling
            dst[qq][t] = t < LT ? src[i >> 2][i & 3] : 0.0f;
        }
}
__device__ __forceinline__ void conv_last_chunk_store(f32x4* row_last, const f32x4 (&src)[4], int last_t) {
    f32x4 d[4];
    if (last_t == 2) conv_last_chunk_perm<2>(src, d);
    else if (last_t == 3) conv_last_chunk_perm<3>(src, d);
    else { d[0] = src[0]; d[1] = src[1]; d[2] = src[2]; d[3] = src[3]; }
#pragma unroll
    for (int qq = 0; qq < 4; qq++) row_last[qq] = d[qq];
}

// Implicit-GEMM main loop of k_tower: RTW row tiles × one 16-channel tile, K = 9·16·CH.  Explicit half-tile
// software pipeline that also runs across tap boundaries:
//     load H2(s) | MFMA H1(s) | load H1(s+1), w(s+2) | MFMA H2(s)
// The two halves of the row tiles are reloaded in place (no second register set); the tap offsets of a half are
// refreshed (3 VALU per tile) right before that half's first load of the new tap.
template <int RTW, int CH, int NM>
__device__ __forceinline__ void conv_mainloop(const f32x4* __restrict__ lds4, const f32x4* __restrict__ wp, size_t wstride4,
                                                 int LS4, int rows, int n, int rho0, int q, const int (&vmask)[NM],
                                                 f32x4 (&acc)[RTW], const int last_t = 4) {
    // last_t < 4 (layer 0 of the fused towers): the real input channels of a tap's LAST 16-channel chunk were permuted so
    // that they fill MFMAs t = 0 … last_t − 1 of the chunk completely (conv_last_chunk_perm); the others would multiply
    // the zero padding and are skipped.
    static_assert(NM >= RTW, "tap masks for every row tile");
    constexpr int total = 9 * CH;
    constexpr int H1 = (RTW + 1) / 2;
    const int zero4 = rows * LS4 + q;
    const int base0 = rho0 * LS4 + q;
    int aoff[RTW];
    auto tap_shift = [&](int tap) { return ((tap / 3 - 1) * n + (tap % 3 - 1)) * LS4; };
    {
        const int sh = tap_shift(0);
#pragma unroll
        for (int j = 0; j < RTW; j++) aoff[j] = (vmask[j] & 1) ? base0 + j * 16 * LS4 + sh : zero4;
    }
    f32x4 a[RTW];
#pragma unroll
    for (int j = 0; j < H1; j++) a[j] = lds4[aoff[j]];
    f32x4 w0 = wp[0];
    f32x4 w1 = wp[wstride4];
    int kk = 0;
#pragma unroll 1
    for (int tap = 0; tap < 9; tap++) {
#pragma unroll
        for (int kc = 0; kc < CH; kc++) {
            if (kc == 0 && tap > 0) {
                const int sh = tap_shift(tap);
#pragma unroll
                for (int j = H1; j < RTW; j++) aoff[j] = ((vmask[j] >> tap) & 1) ? base0 + j * 16 * LS4 + sh : zero4;
            }
#pragma unroll
            for (int j = H1; j < RTW; j++) a[j] = lds4[aoff[j] + kc * 4];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 4; t++)
                if (kc + 1 < CH || t < last_t) {
#pragma unroll
                    for (int j = 0; j < H1; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[t], a[j][t], acc[j], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
            const int k2 = kk + 2 < total ? kk + 2 : total - 1;
            const f32x4 w2 = wp[(size_t)k2 * wstride4];
            if (kc + 1 < CH) {
#pragma unroll
                for (int j = 0; j < H1; j++) a[j] = lds4[aoff[j] + (kc + 1) * 4];
            } else if (tap + 1 < 9) {
                const int sh = tap_shift(tap + 1);
#pragma unroll
                for (int j = 0; j < H1; j++)
                    aoff[j] = ((vmask[j] >> (tap + 1)) & 1) ? base0 + j * 16 * LS4 + sh : zero4;
#pragma unroll
                for (int j = 0; j < H1; j++) a[j] = lds4[aoff[j]];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 4; t++)
                if (kc + 1 < CH || t < last_t) {
#pragma unroll
                    for (int j = H1; j < RTW; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[t], a[j][t], acc[j], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
            w0 = w1;
            w1 = w2;
            kk++;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The same loop over the HALO image (k_tower_halo, layers ≥ 1).  Every board row is stored with one zero cell behind
// it and every position with a zero row (+ 1 cell) behind it, so a tap is the same cell offset for every square of
// every position — on or off the board — and an off-board neighbour is simply a zero cell:
//     cell(p, y, x) = (NB + 2) + p·PS + y·(NB + 1) + x,   tap (dy, dx) → cell + dy·(NB + 1) + dx.
// A lane therefore keeps ONE address per row tile for the whole layer and every (tap, chunk) is an immediate of
// ds_read_b128: no per-tap address arithmetic, no validity masks, no select against a zero row (what cost the
// masked loop 8 % of a layer: scripts/probes/tower_stamps.hip), and — with the squares dealt to tile slots by the
// residue of their cell index (tower_halo_slotmap) — no bank conflicts either.  Row pitch = F + 4 floats
// (4·CH + 1 slots of 16 B).  addr4[j] = (cell − (NB + 2))·pitch4 + q.
// Same products in the same order as conv_mainloop → identical bits.
// ------------------------------------------------------------------------------------------------
// `turn` (0 / 1, wave-uniform): the two waves that share a SIMD take turns at s_setprio 1, one chunk each.  Left alone the
// arbiter prefers the older wave throughout: it finishes its tiles at ≈ 3/4 of the layer and the younger one runs the
// last quarter alone, where nothing fills the gaps between its own MFMA groups (37 instead of 33 cycles per MFMA).
// COT = output channel tiles of the layer's weight matrix (CoutP / 16): CH in the towers (F → F), more for a wider head.
// SPLIT (round 5; the stand-alone layers of the training step, k_conv_halo): every tap's 16·CH products are summed in a chain of their
// own that starts from zero (the first MFMA of a tap takes the constant 0 as its C operand) and is added to `acc` when the tap is
// complete — nine chains of 16·CH terms and nine additions instead of ONE chain of 9·16·CH.  A chain of K terms rounds K times on
// partial sums that grow like √k, so its error grows like √(K/2) ulp: for 128 → 128 layers the one-chain sum is 2.5 × less accurate
// than ATen's convolution (measured per layer against fp64, profiles/r05_a_train_error_budget.txt: 4.5e-7 σ against 1.76e-7 σ),
// which over 21 layers put 3 – 4 × as many pre-activations on the wrong side of a ReLU in the training step's forward pass.  The
// tower of the inference path keeps the single chain (its outputs sit 30 × inside the 1e-4 gate and its registers are full).
// The flush of one half of the tiles is issued behind the MFMA group of the other half, whose 4·H MFMAs ago its own last MFMA
// was issued — the adds do not wait for the matrix pipe, except behind the last tap of a row of taps (no chain lives across the
// trips of the row loop: the compiler otherwise rotates the loop around the carried chains and runs out of registers).
template <int RTW, int CH, int NB, int NM, int COT = CH, bool SPLIT = false>
__device__ __forceinline__ void conv_mainloop_halo(const f32x4* __restrict__ lds4, const float* __restrict__ wlayer,
                                                   const float* __restrict__ wnext, uint32_t wlane, const int (&addr4)[NM],
                                                   f32x4 (&acc)[RTW], const int turn, f32x4& w0, f32x4& w1) {
    static_assert(NM >= RTW, "an address for every row tile");
    constexpr int P4 = 4 * CH + 1, RS = NB + 1;
    constexpr int H1 = (RTW + 1) / 2;
    constexpr size_t WCHUNK = (size_t)16 * COT * 4 * 16;  // bytes of one 16-k chunk of the layer's weights: [CoutP][4 slots][16 B]
#ifndef TG_PRIO_PERIOD
#define TG_PRIO_PERIOD 2
#endif
#define TG_HALO_OFF(step) ((((step) / CH) / 3 * RS + ((step) / CH) % 3) * P4 + ((step) % CH) * 4)
    // weights through a buffer descriptor of the layer: the chunk is the scalar offset (one s_movk per load), this lane's
    // constant 16 B inside a chunk the vector offset — no vector address arithmetic in the loop
    // The stream runs on into the next layer: w0 / w1 arrive holding this layer's chunks 0 and 1 (conv_halo_first_weights, or
    // the previous call) and leave holding the next layer's, requested by the last two steps — the epilogue and the barriers
    // between two layers hide that latency.
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wlayer, 0, (int)(9 * CH * WCHUNK), 0x00020000);
#define TG_HALO_W(chunk) __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane, (int)((chunk) * WCHUNK), 0))
    f32x4 a[RTW];
    // One board row of taps (dy) per trip of a 3-trip loop, the 3·CH steps of a row unrolled: a third of the code of the
    // full unroll (the two loop instances of a workgroup fit the instruction cache with room to spare) at 13 vector adds
    // per row switch.  The software pipeline runs across the row switch.
    constexpr int ROW = 3 * CH;
#define TG_HALO_OFF2(step) (((step) / CH) * P4 + ((step) % CH) * 4)
    int ad[RTW];
#pragma unroll
    for (int j = 0; j < RTW; j++) ad[j] = addr4[j];
    f32x4 part[SPLIT ? RTW : 1];  // SPLIT: the current tap's chain per tile (dead between two rows of taps)
    const f32x4 zero = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    // MFMA t of step s for tile j: SPLIT → into the tap's chain, restarted from 0 at the tap's first product
#define TG_HALO_MFMA(j, t, s)                                                                                                  \
    if (SPLIT) part[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[t], a[j][t], ((s) % CH == 0 && (t) == 0) ? zero : part[j], 0, 0, 0); \
    else acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[t], a[j][t], acc[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < H1; j++) a[j] = lds4[ad[j] + TG_HALO_OFF2(0)];
    int wchunk = 2;  // next chunk of weights to request
#pragma unroll 1
    for (int dy = 0; dy < 3; dy++) {
#pragma unroll
        for (int s = 0; s < ROW; s++) {
#ifndef TG_NO_PRIO_TURNS
            if (s % TG_PRIO_PERIOD == 0) {
                if ((((s / TG_PRIO_PERIOD) + dy) & 1) == turn) __builtin_amdgcn_s_setprio(1);
                else __builtin_amdgcn_s_setprio(0);
            }
#endif
#pragma unroll
            for (int j = H1; j < RTW; j++) a[j] = lds4[ad[j] + TG_HALO_OFF2(s)];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int j = 0; j < H1; j++) { TG_HALO_MFMA(j, t, s) }
            __builtin_amdgcn_sched_barrier(0);
            // (the empty asm pins every flush where it stands: the adds depend on nothing but their operands, and instruction
            // selection otherwise collects them at the end of the trip — with every tap's chains of the row alive until then)
            if (SPLIT && s % CH == 0 && s > 0) {  // the previous tap's second half is complete
#pragma unroll
                for (int j = H1; j < RTW; j++) { acc[j] += part[j]; asm volatile("" : "+v"(acc[j])); }
            }
            f32x4 w2;
            if (s < ROW - 2) {
                w2 = TG_HALO_W(wchunk);
            } else {  // the last two steps of a row of taps — of the layer when dy = 2: on to the next layer's first chunks
                const bool on = dy == 2;
                const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(on ? wnext : wlayer), 0, (int)(9 * CH * WCHUNK), 0x00020000);
                w2 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, wlane, (int)((on ? s - (ROW - 2) : wchunk) * WCHUNK), 0));
            }
            wchunk++;
            if (s + 1 < ROW) {
#pragma unroll
                for (int j = 0; j < H1; j++) a[j] = lds4[ad[j] + TG_HALO_OFF2(s + 1)];
            } else {
                // row switch: the second half's fragments of this step are already in registers (or in flight with the
                // old addresses); from here on every read is of the next row of taps
#pragma unroll
                for (int j = H1; j < RTW; j++) ad[j] += RS * P4;
                if (dy < 2) {
#pragma unroll
                    for (int j = 0; j < H1; j++) { ad[j] += RS * P4; a[j] = lds4[ad[j] + TG_HALO_OFF2(0)]; }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int j = H1; j < RTW; j++) { TG_HALO_MFMA(j, t, s) }
            __builtin_amdgcn_sched_barrier(0);
            if (SPLIT && s % CH == CH - 1) {  // this tap's first half is complete; behind the row's last tap the second half too
#pragma unroll                             // (its last tiles' MFMAs are still in flight: a few dozen cycles per row of taps)
                for (int j = 0; j < (s == ROW - 1 ? RTW : H1); j++) { acc[j] += part[j]; asm volatile("" : "+v"(acc[j])); }
            }
            w0 = w1;
            w1 = w2;
        }
    }
#undef TG_HALO_MFMA
#undef TG_HALO_OFF2
#ifndef TG_NO_PRIO_TURNS
    __builtin_amdgcn_s_setprio(0);
#endif
#undef TG_HALO_OFF
#undef TG_HALO_W
}

// chunks 0 and 1 of a layer's weights for the first conv_mainloop_halo call of a kernel
template <int CH, int COT = CH>
__device__ __forceinline__ void conv_halo_first_weights(const float* __restrict__ wlayer, uint32_t wlane, f32x4& w0, f32x4& w1) {
    constexpr size_t WCHUNK = (size_t)16 * COT * 4 * 16;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wlayer, 0, (int)(9 * CH * WCHUNK), 0x00020000);
    w0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane, 0, 0));
    w1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane, (int)WCHUNK, 0));
}

}  // namespace tg
