// Where a K-step of k_fc_ring goes, wave by wave: builds net_kernels.hip with -DTG_TOWER_STAMPS (TG_STAMP in the kernel: top of a step,
// after the wait for ready[], before / after the mid-step s_waitcnt vmcnt(0), after the wait for done[], after the refill's LDS-DMA issue,
// end of the step) and prints, for workgroup 0 of a C2-shaped launch (4096 rows, K = 1600, logits rows), the s_memtime deltas summed
// over the 25 steps of every wave.
// hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DTG_TOWER_STAMPS scripts/probes/fc_ring_stamps.hip -o scripts/probes/_bin/fc_ring_stamps
#include <cstdio>
#include <vector>
#include "../../tak_amd/csrc/net_kernels.hip"
#include "probe_env.h"
using namespace tg;
int main() {
    const int M = 4096, K = 1600, NP = 1664, nsteps = K / 64;
    std::vector<float> hA((size_t)M * K), hW((size_t)K * NP), hb(NP, 0.0f);
    for (auto& v : hA) v = std::max(0.0f, (float)rand() / (float)RAND_MAX - 0.4f);
    for (auto& v : hW) v = ((float)rand() / (float)RAND_MAX - 0.5f) * 0.05f;
    float *A, *W, *b, *out, *stats;
    hipMalloc(&A, hA.size() * 4); hipMalloc(&W, hW.size() * 4); hipMalloc(&b, NP * 4); hipMalloc(&out, (size_t)M * NP * 4);
    hipMalloc(&stats, (size_t)M * FC_STAT_STRIDE * 2 * 4);
    hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice); hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(b, hb.data(), NP * 4, hipMemcpyHostToDevice);
    unsigned long long* stamps; const size_t ns = (size_t)(nsteps + 1) * 16 * 8;
    hipMalloc(&stamps, ns * 8); hipMemset(stamps, 0, ns * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(g_tower_stamps), &stamps, sizeof(stamps));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&]() { return launch_gemm(nullptr, A, K, W, b, out, M, K, NP, NP, 1576, true, stats, 1575, nullptr); };
    for (int i = 0; i < 3; i++) launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 10; i++) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("k_fc_ring with stamps: %.1f us per launch\n", ms * 100);
    std::vector<unsigned long long> st(ns);
    hipMemcpy(st.data(), stamps, ns * 8, hipMemcpyDeviceToHost);
    auto S = [&](int step, int wave, int slot) { return (double)st[((size_t)step * 16 + wave) * 8 + slot]; };
    printf("cycles per wave over %d steps (workgroup 0): wait ready | chunks 0-1 | wait vmcnt | wait done | issue refill | chunks 2-3 | whole loop\n", nsteps);
    for (int w = 0; w < 8; w++) {
        double ready = 0, c01 = 0, vm = 0, done = 0, fill = 0, c23 = 0;
        for (int s = 0; s < nsteps; s++) {
            ready += S(s, w, 1) - S(s, w, 0);
            c01 += S(s, w, 2) - S(s, w, 1);
            vm += S(s, w, 3) - S(s, w, 2);
            const bool refill = s + 2 < nsteps;
            done += refill ? S(s, w, 4) - S(s, w, 3) : 0;
            fill += refill ? S(s, w, 5) - S(s, w, 4) : S(s, w, 5) - S(s, w, 3);
            c23 += S(s, w, 6) - S(s, w, 5);
        }
        printf("  wave %d: %8.0f | %8.0f | %8.0f | %8.0f | %8.0f | %8.0f | %8.0f   (start +%.0f)\n", w, ready, c01, vm, done, fill, c23, S(nsteps, w, 0) - S(0, w, 0),
               S(0, w, 0) - S(0, 0, 0));
    }
    printf("MFMA issue per wave: %d chains x 400 k-slices x 32 cycles = %d (two waves share a SIMD)\n", 13, 13 * 400 * 32);
    return 0;
}
