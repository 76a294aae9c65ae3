// One stand-alone 3×3 layer of the training step (k_conv_halo, 128 → 128 on 5×5, 4000 positions) with parts removed:
// -DTG_CONV_PROBE=mask (1 = no main loop, 2 = no input rows, 4 = no output) shows what the single workgroup per CU cannot hide.
// hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DTG_CONV_PROBE=0 -I../../tak_amd/csrc conv_halo_probe.hip -o _bin/conv_halo_probe
// (round 5: the TG_CONV_PROBE masks were removed from the product kernel — commit 6da8f61 has them; this program now times the shipped kernel)
#ifndef TG_CONV_PROBE
#define TG_CONV_PROBE 0
#endif
#include <cstdio>
#include <vector>
#include <algorithm>
#include "../../tak_amd/csrc/net_kernels.hip"
#include "probe_env.h"
using namespace tg;
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 4000, n = 5, F = 128, M = B * n * n;
    float *in, *out, *w, *b; double* part;
    hipMalloc(&in, (size_t)M * F * 4); hipMalloc(&out, (size_t)M * F * 4);
    hipMalloc(&w, (size_t)9 * F * F * 4); hipMalloc(&b, F * 4); hipMalloc(&part, (size_t)1024 * 2 * F * 8);
    std::vector<float> h((size_t)M * F);
    for (size_t i = 0; i < h.size(); i++) h[i] = 0.001f * (float)((i * 2654435761u) % 1999) - 1.0f;
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    h.resize((size_t)9 * F * F);
    for (size_t i = 0; i < h.size(); i++) h[i] = 0.0001f * (float)((i * 2654435761u) % 197) - 0.01f;
    hipMemcpy(w, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemset(b, 0, F * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int blocks = 0;
    auto launch = [&]() { return launch_conv3x3(nullptr, in, w, b, nullptr, out, M, n, F, F, F, F, false, part, &blocks); };
    for (int i = 0; i < 3; i++) if (launch() != hipSuccess) { printf("launch failed\n"); return 1; }
    hipDeviceSynchronize();
    const int reps = 20;
    hipEventRecord(e0);
    for (int i = 0; i < reps; i++) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1000 / reps, fl = 2.0 * M * F * F * 9;
#ifdef TG_TOWER_STAMPS
    {   // s_memtime stamps of workgroup 0 in one more launch: staging | slot setup | main loop | epilogue, per wave
        const int nwg = (B + 7) / 8;
        unsigned long long* stamps; hipMalloc(&stamps, (128 + 2 * 1024) * 8); hipMemset(stamps, 0, (128 + 2 * 1024) * 8);
        hipMemcpyToSymbol(HIP_SYMBOL(g_tower_stamps), &stamps, sizeof(stamps));
        launch(); hipDeviceSynchronize();
        unsigned long long hs[16 * 8];
        hipMemcpy(hs, stamps, sizeof(hs), hipMemcpyDeviceToHost);
        for (int w = 0; w < 8; w++) {
            const unsigned long long* t = &hs[w * 8];
            printf("wave %d: staging %llu  setup %llu  mainloop %llu  epilogue %llu cycles\n", w, t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3]);
        }
        std::vector<unsigned long long> wg(2 * 1024);
        hipMemcpy(wg.data(), stamps + 128, wg.size() * 8, hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull, t1 = 0;
        for (int i = 0; i < nwg; i++) { t0 = std::min(t0, wg[2 * i]); t1 = std::max(t1, wg[2 * i + 1]); }
        printf("launch span (first workgroup start -> last end): %.1f us over %d workgroups\n", (t1 - t0) / 100.0, nwg);
        // histogram of starts and of durations in 5 us bins
        int hs0[64] = {0}, hd[64] = {0}, he[64] = {0};
        for (int i = 0; i < nwg; i++) {
            hs0[std::min<unsigned long long>(63, (wg[2 * i] - t0) / 500)]++;
            he[std::min<unsigned long long>(63, (wg[2 * i + 1] - t0) / 500)]++;
            hd[std::min<unsigned long long>(63, (wg[2 * i + 1] - wg[2 * i]) / 500)]++;
        }
        printf("bin(5us) starts ends durations\n");
        for (int b = 0; b < 64; b++) if (hs0[b] || he[b] || hd[b]) printf("%3d %5d %5d %5d\n", b * 5, hs0[b], he[b], hd[b]);
    }
#endif
    printf("probe mask %d: %d positions, %.1f us per layer, %.1f TFLOP/s (%.3f of 157.3), stats blocks %d\n", TG_CONV_PROBE, B, us, fl / us / 1e6, fl / us / 1e6 / 157.3, blocks);
    return 0;
}
