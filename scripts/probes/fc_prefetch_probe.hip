// Round 5, VERDICT item 5 (i): is the policy FC's FIRST FILL worth issuing early?  k_fc_ring starts with two K-steps of weights
// (2 × 52 KB per workgroup; per XCD the 4 column blocks it computes + the leftover tiles: 408 KB) that it cannot overlap with
// anything.  If those lines were already in the XCD's L2 when the kernel starts — touched from the tail of the tower kernel —
// the first fill would pay an L2 hit instead of an Infinity-Cache / HBM round trip.  This program measures the upper bound of
// that: per iteration a stand-in for the tower's output traffic (26 MB written), then EITHER nothing OR a prefetch kernel that
// reads exactly those lines from workgroups of the XCD that will use them, then the FC (C2 shape, logits rows), timed alone with
// HIP events.  (The prefetch kernel's own time is outside the timed region: in the product it would ride in the tower's tail.)
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I tak_amd/csrc scripts/probes/fc_prefetch_probe.hip -o scripts/probes/_bin/fc_prefetch_probe
#include <cstdio>
#include <vector>
#include "../../tak_amd/csrc/net_kernels.hip"
#include "probe_env.h"
using namespace tg;

// workgroup b runs on XCD b % 8 (round robin by linear index); k_fc_ring<0> gives XCD x the column blocks 4 (x & 1) … + 3
__global__ __launch_bounds__(128) void k_fc_prefetch(const f32x4* __restrict__ Wlin, int NP, int steps, float* __restrict__ sink) {
    const int x = blockIdx.x & 7, part = blockIdx.x >> 3, parts = gridDim.x >> 3;
    const int tiles = FC_MAIN_TILES * 4 + 3;            // 48 main tiles of the XCD's column blocks + the 3 leftover tiles
    const int lines = steps * 4 * tiles * 8;            // 128-byte lines: 8 per (chunk, tile) block of 1 KB
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = part * 128 + threadIdx.x; i < lines; i += parts * 128) {
        const int line = i & 7, blk = i >> 3, c = blk / tiles, t = blk - c * tiles;
        const int tile = t < FC_MAIN_TILES * 4 ? FC_MAIN_TILES * 4 * (x & 1) + t : FC_MAIN_TILES * FC_MAIN_BLOCKS + (t - FC_MAIN_TILES * 4);
        acc += Wlin[((size_t)c * (NP >> 4) + tile) * 64 + line * 8];
    }
    if (acc[0] == 12345.678f) sink[0] = acc[1];
}
__global__ void k_fill(float* __restrict__ p, size_t n, float v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v + (float)(i & 7);
}

int main() {
    const int M = 4096, K = 1600, NP = 1664;
    std::vector<float> hA((size_t)M * K), hW((size_t)K * NP), hb(NP, 0.0f);
    for (auto& v : hA) v = std::max(0.0f, (float)rand() / (float)RAND_MAX - 0.4f);
    for (auto& v : hW) v = ((float)rand() / (float)RAND_MAX - 0.5f) * 0.05f;
    float *A, *W, *Wl, *b, *out, *stats, *sink;
    hipMalloc(&A, hA.size() * 4); hipMalloc(&W, hW.size() * 4); hipMalloc(&Wl, hW.size() * 4); hipMalloc(&b, NP * 4);
    hipMalloc(&out, (size_t)M * NP * 4); hipMalloc(&stats, (size_t)M * FC_STAT_STRIDE * 2 * 4); hipMalloc(&sink, 64);
    hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice); hipMemcpy(Wl, hW.data(), hW.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(b, hb.data(), NP * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto fc = [&]() { return launch_gemm(nullptr, A, K, W, b, out, M, K, NP, NP, 1576, false, stats, 1575, nullptr, Wl); };
    for (int steps : {0, 2, 4, 25}) {
        for (int rep = 0; rep < 3; rep++) {
            double us = 0.0;
            const int iters = 40;
            for (int i = 0; i < iters + 3; i++) {
                hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, nullptr, A, hA.size(), 0.01f * (float)(i & 3));  // the tower's output: 26 MB written
                if (steps) hipLaunchKernelGGL(k_fc_prefetch, dim3(256), dim3(128), 0, nullptr, (const f32x4*)Wl, NP, steps, sink);
                hipEventRecord(e0);
                if (fc() != hipSuccess) { printf("launch failed\n"); return 1; }
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (i >= 3) us += 1000.0 * ms;
            }
            printf("first %2d K-steps of the weights touched from the consuming XCD before the launch: k_fc_ring %.2f us (events, %d launches)\n", steps, us / iters, iters);
        }
    }
    return 0;
}
