// k_conv_pair (the training step's 128 → 128 layer on two wave groups) beside k_conv_halo: launch time, and s_memtime stamps of
// workgroup 0's waves at the phase boundaries of both images (wait for the image to be free | staging + group sync | main loop |
// epilogue).  hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DTG_TOWER_STAMPS -I tak_amd/csrc scripts/probes/conv_pair_stamps.hip -o scripts/probes/_bin/conv_pair_stamps
#include <cstdio>
#include <vector>
#include <algorithm>
#include "../../tak_amd/csrc/net_kernels.hip"
using namespace tg;
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 4000, n = 5, F = 128, M = B * n * n;
    float *in, *out, *w, *b; double* part;
    hipMalloc(&in, (size_t)M * F * 4); hipMalloc(&out, (size_t)M * F * 4);
    hipMalloc(&w, (size_t)9 * F * F * 4); hipMalloc(&b, F * 4); hipMalloc(&part, (size_t)4096 * 2 * F * 8);
    std::vector<float> h((size_t)M * F);
    for (size_t i = 0; i < h.size(); i++) h[i] = std::max(0.0f, 0.001f * (float)((i * 2654435761u) % 1999) - 1.0f);
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    h.resize((size_t)9 * F * F);
    for (size_t i = 0; i < h.size(); i++) h[i] = 0.0001f * (float)((i * 2654435761u) % 197) - 0.01f;
    hipMemcpy(w, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemset(b, 0, F * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int blocks = 0;
    const bool nostats = argc > 2;
    auto launch = [&]() { return launch_conv3x3(nullptr, in, w, b, nullptr, out, M, n, F, F, F, F, false, nostats ? nullptr : part, &blocks); };
    for (int i = 0; i < 3; i++) if (launch() != hipSuccess) { printf("launch failed\n"); return 1; }
    hipDeviceSynchronize();
    const int reps = 20;
    hipEventRecord(e0);
    for (int i = 0; i < reps; i++) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1000 / reps, fl = 2.0 * M * F * F * 9;
    printf("%s: %d positions, %.1f us per layer, %.1f TFLOP/s (%.3f of 157.3), stats blocks %d\n", getenv("TG_NO_PAIR_CONV") ? "k_conv_halo" : "k_conv_pair", B, us,
           fl / us / 1e6, fl / us / 1e6 / 157.3, blocks);
    unsigned long long* stamps; hipMalloc(&stamps, (128 + 2 * 1024) * 8); hipMemset(stamps, 0, (128 + 2 * 1024) * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(g_tower_stamps), &stamps, sizeof(stamps));
    launch(); hipDeviceSynchronize();
    unsigned long long hs[2 * 16 * 8];
    hipMemcpy(hs, stamps, sizeof(hs), hipMemcpyDeviceToHost);
    if (!getenv("TG_NO_PAIR_CONV")) {
        unsigned long long t0 = ~0ull;
        for (int w8 = 0; w8 < 8; w8++) t0 = std::min(t0, hs[w8 * 8]);
        for (int w8 = 0; w8 < 8; w8++) {
            printf("wave %d (group %c):", w8, w8 < 4 ? 'A' : 'B');
            for (int img = 0; img < 2; img++) {
                const unsigned long long* t = &hs[(img * 16 + w8) * 8];
                printf("  image %d at %7llu: wait %6llu  stage+sync %6llu  mainloop %7llu  epilogue %6llu", img, t[0] - t0, t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3]);
            }
            printf("\n");
        }
    }
    return 0;
}
