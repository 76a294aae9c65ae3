// kernels.h — launchers of the HIP kernels (definitions in *_kernels.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tg {

// board_kernels.hip
void launch_movegen(hipStream_t st, const uint8_t* states, int count, int n, uint16_t* moves, int32_t* counts);
void launch_play(hipStream_t st, uint8_t* states, int count, int n, const uint16_t* moves, uint8_t* status);
void launch_result(hipStream_t st, const uint8_t* states, int count, int n, uint8_t* results);
void launch_encode(hipStream_t st, const uint8_t* states, int count, int n, float* planes, bool nhwc);
void launch_move_index(hipStream_t st, const uint16_t* moves, int count, int n, bool legacy5, const int16_t* lut5, int32_t* index);
void launch_perft_count(hipStream_t st, const uint8_t* states, int count, int n, int32_t* nchild, uint8_t* terminal);
void launch_perft_expand(hipStream_t st, const uint8_t* states, int count, int n, const int64_t* offsets, const int32_t* root_of,
                         uint8_t* next_states, int32_t* next_root);

}  // namespace tg
