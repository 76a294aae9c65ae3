// fc_ring.cuh — what the two ring FCs (k_fc_ring in net_kernels.hip: exact f32; k_fc_s3_ring in net_s3_kernels.hip: split bf16) share:
// the dealing of the leftover tiles, the gather target, the LDS flag primitives.  Tile and statistics geometry: softmax.cuh.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "softmax.cuh"

namespace tg {

// which leftover tile (l), from which row tile on (s), for how many row tiles (ne) workgroup column cb computes it
struct FcExtra { int l, s, ne; };
__host__ __device__ inline FcExtra fc_extra(int cb) {
    if (cb < 6) { const int g = cb % 3; return FcExtra{cb / 3, 3 * g, g < 2 ? 3 : 2}; }
    return FcExtra{2, 4 * (cb - 6), 4};
}

// What a search iteration needs of the FC's output is not the 1576 logits of a leaf but the ≈ 45 of its children: with
// `gather` set the epilogue writes NO logits row; every wave parks its 16 rows × 13 tiles in LDS and copies, for each of its
// rows, the logits of that leaf's children (child_pidx: the policy index of child c, recorded by the select kernel) that fall
// into its columns to child_logit[row][c] — 0.7 MB per iteration instead of a 27 MB logits burst that the tree backup then
// gathers 45 of 1664 floats from.  The statistics record carries the value pre-activation (pair FC_STAT_BLOCKS).
struct FcGather {
    const uint16_t* child_pidx;  // [M][stride] policy index per child of row's leaf, 0xFFFF = unmapped
    const uint32_t* leaf_rec;    // [M][2]: {children block, child count}
    float* child_logit;          // [M][stride]
    int stride;                  // EX_MOVES
};

// flags are read and bumped with explicit LDS instructions: a volatile access through a generic pointer becomes a flat load
// with s_waitcnt vmcnt(0), which would drain the LDS-DMA loads in flight at every poll
__device__ __forceinline__ void fc_ring_wait(uint32_t flag_addr, uint32_t target) {
    for (;;) {
        uint32_t v;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(flag_addr) : "memory");
        if (__builtin_amdgcn_readfirstlane((int)v) >= (int)target) break;
        __builtin_amdgcn_s_sleep(1);
    }
}
__device__ __forceinline__ void fc_ring_signal(uint32_t flag_addr) {
    if ((threadIdx.x & 63) == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(flag_addr), "v"(1u) : "memory");
}
}  // namespace tg
