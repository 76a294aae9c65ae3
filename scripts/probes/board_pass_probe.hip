// board_pass_probe.hip — times k_board_pass (play → result → movegen count → encode) on 2^20 positions reached by
// pseudo-random play on the device, with components removed (-DBOARD_PROBE bit mask: 1 no result, 2 no movegen, 4 no encode).
// (round 5: the BOARD_PROBE masks were removed from the product kernel — commit 6da8f61 has them; this program now times the shipped kernel)
#ifndef BOARD_PROBE
#define BOARD_PROBE 0
#endif
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../tak_amd/csrc/board_kernels.hip"
#include "probe_env.h"

__global__ void k_pick(const uint16_t* moves, const int32_t* counts, int count, uint32_t salt, uint16_t* chosen) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint32_t h = (uint32_t)i * 2654435761u ^ salt * 40503u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    int c = counts[i];
    chosen[i] = c > 0 ? moves[(size_t)i * TG_MAX_MOVES + h % (uint32_t)c] : 0;
}
__global__ void k_init(uint8_t* states, int count, int n) {
    int gi = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gi >= count) return;
    tg::Geom g = tg::make_geom(n);
    tg::WState s;
    tg::ws_start(s, g, 4);
    tg::ws_store(s, states + (size_t)gi * g.bytes, g);
}

int main() {
    using namespace tg;
    const int n = 5, N = 1 << 20, plies = 30;
    const size_t sb = 256;
    uint8_t *st, *out, *status, *res; uint16_t *mv, *ch; int32_t* cnt; float* planes;
    hipMalloc((void**)&st, N * sb); hipMalloc((void**)&out, N * sb); hipMalloc((void**)&status, N); hipMalloc((void**)&res, N);
    hipMalloc((void**)&mv, (size_t)N * TG_MAX_MOVES * 2); hipMalloc((void**)&ch, N * 2); hipMalloc((void**)&cnt, N * 4);
    hipMalloc((void**)&planes, (size_t)N * 25 * 80 * 4);
    hipStream_t s0; hipStreamCreate(&s0);
    hipLaunchKernelGGL(k_init, dim3(N / 4), dim3(256), 0, s0, st, N, n);
    for (int p = 0; p <= plies; p++) {
        launch_movegen(s0, st, N, n, mv, cnt);
        hipLaunchKernelGGL(k_pick, dim3(N / 256), dim3(256), 0, s0, mv, cnt, N, (uint32_t)p, ch);
        if (p < plies) launch_play(s0, st, N, n, ch, status);
    }
    hipStreamSynchronize(s0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; i++) launch_board_pass(s0, st, ch, N, n, out, res, cnt, planes, 80);
    hipEventRecord(e0, s0);
    const int reps = 10;
    for (int i = 0; i < reps; i++) launch_board_pass(s0, st, ch, N, n, out, res, cnt, planes, 80);
    hipEventRecord(e1, s0);
    hipStreamSynchronize(s0);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("BOARD_PROBE=%d: %.3f ms per pass of %d positions (%s)\n", BOARD_PROBE, ms / reps, N, hipGetErrorString(hipGetLastError()));
    return 0;
}
