// fc_reg.cuh — the register-tiled main loop of the policy-FC variant "k_fc_reg" (round 4, measured and NOT adopted: see
// scripts/probes/README.md and profiles/r04_b_fc_candidates.txt); used by scripts/probes/fc_reg_probe.hip
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../tak_amd/csrc/conv_mainloop.cuh"  // f32x4

namespace tg {

constexpr int FR_D = 4;          // chunks in flight
constexpr int FR_MAIN = 12;      // output tiles per column block
constexpr int FR_BLOCKS = 8;     // column blocks
constexpr int FR_XT0 = FR_MAIN * FR_BLOCKS;  // first leftover tile (96)
constexpr int FR_NX = 3;         // leftover tiles

#ifdef TG_FR_PROBE
#define TG_FR_PROBE_ALL 1
#else
#define TG_FR_PROBE_ALL 0
#endif
template <bool EXTRA>
__device__ __forceinline__ void fc_reg_loop(const f32x4* const (&ap)[4], size_t achunk, const f32x4* __restrict__ wb, size_t wchunk,
                                            const f32x4* __restrict__ wx, int nchunks, f32x4 (&acc)[4][3], f32x4& accx) {
    // Register sets a[s], w[s], x[s], s = chunk % FR_D.  While chunk c computes out of set c % FR_D, chunk c + FR_D − 1 is
    // requested into set (c − 1) % FR_D — the one chunk c − 1 has just finished with — in three groups between the four k-slices'
    // MFMAs.  The schedule is pinned with sched_barrier: left alone, hipcc sinks every load to just before its use (108 registers,
    // one chunk in flight, s_waitcnt vmcnt(0) in front of every chunk).
    f32x4 a[FR_D][4], w[FR_D][3], x[FR_D];
#pragma unroll
    for (int d = 0; d < (TG_FR_PROBE_ALL ? FR_D : FR_D - 1); d++) {
#pragma unroll
        for (int i = 0; i < 4; i++) a[d][i] = ap[i][(size_t)d * achunk];
#pragma unroll
        for (int j = 0; j < 3; j++) w[d][j] = wb[(size_t)d * wchunk + j * 64];
        if (EXTRA) x[d] = wx[(size_t)d * wchunk];
        // (the order matters to hipcc's wait counts: requested in another order than the loop's, the loop head waits vmcnt(0))
        __builtin_amdgcn_sched_barrier(0);
    }
#define TG_FR_MFMA(D_, T_)                                                                                                  \
    _Pragma("unroll") for (int i = 0; i < 4; i++)                                                                           \
        _Pragma("unroll") for (int j = 0; j < 3; j++)                                                                       \
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[D_][j][T_], a[D_][i][T_], acc[i][j], 0, 0, 0);              \
    if (EXTRA) accx = __builtin_amdgcn_mfma_f32_16x16x4f32(x[D_][T_], a[D_][0][T_], accx, 0, 0, 0);
    for (int kc0 = 0; kc0 < nchunks; kc0 += FR_D) {  // nchunks is a multiple of FR_D (K % 64 == 0)
#pragma unroll
        for (int d = 0; d < FR_D; d++) {
            constexpr int NS = FR_D - 1;
            const int dp = (d + NS) % FR_D;                                                       // the set to refill
            const int kn = kc0 + d + NS < nchunks ? kc0 + d + NS : nchunks - 1;                   // (the tail re-reads the last chunk, unused)
            __builtin_amdgcn_sched_barrier(0);
            TG_FR_MFMA(d, 0)
            __builtin_amdgcn_sched_barrier(0);
#ifndef TG_FR_PROBE
#define TG_FR_PROBE 0  // timing probes (wrong results): 1 = no activation stream, 2 = no weight stream
#endif
            if (!(TG_FR_PROBE & 1)) {
                a[dp][0] = ap[0][(size_t)kn * achunk];
                a[dp][1] = ap[1][(size_t)kn * achunk];
            }
            __builtin_amdgcn_sched_barrier(0);
            TG_FR_MFMA(d, 1)
            __builtin_amdgcn_sched_barrier(0);
            if (!(TG_FR_PROBE & 1)) {
                a[dp][2] = ap[2][(size_t)kn * achunk];
                a[dp][3] = ap[3][(size_t)kn * achunk];
            }
            __builtin_amdgcn_sched_barrier(0);
            TG_FR_MFMA(d, 2)
            __builtin_amdgcn_sched_barrier(0);
            if (!(TG_FR_PROBE & 2)) {
#pragma unroll
                for (int j = 0; j < 3; j++) w[dp][j] = wb[(size_t)kn * wchunk + j * 64];
                if (EXTRA) x[dp] = wx[(size_t)kn * wchunk];
            }
            __builtin_amdgcn_sched_barrier(0);
            TG_FR_MFMA(d, 3)
        }
    }
#undef TG_FR_MFMA
}

}  // namespace tg
