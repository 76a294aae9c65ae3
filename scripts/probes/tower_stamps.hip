// Where a layer of the fused tower spends its cycles: builds net_kernels.hip with -DTG_TOWER_STAMPS and prints, for
// workgroup 0 of a C2-shaped launch (4096 positions, 6 blocks × 64 filters), the s_memtime deltas between the phase
// boundaries of every layer and wave: main loop | epilogue VALU | wait at barrier 1 | LDS write-back | wait at barrier 2.
// hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DTG_TOWER_STAMPS -I../../tak_amd/csrc tower_stamps.hip -o _bin/tower_stamps
#include <cstdio>
#include <vector>
#include "../../tak_amd/csrc/net_kernels.hip"
#include "probe_env.h"
using namespace tg;
int main(int argc, char** argv) {
    const int variant = argc > 1 ? atoi(argv[1]) : 8;
    const bool c5 = variant == 85;  // variant 85: the C5 network (10 blocks × 128 filters, 8 positions per workgroup), constant planes as a bias
    const int B = 4096, n = 5, F = c5 ? 128 : 64, R = c5 ? 10 : 6, cin_pad = 80, nl = 1 + 2 * R;
    uint8_t* states; hipMalloc(&states, (size_t)B * 256); hipMemset(states, 0, (size_t)B * 256);
    std::vector<uint8_t> hs((size_t)B * 256, 0);
    for (int b = 0; b < B; b++) { uint8_t* h = &hs[(size_t)b * 256 + 240]; h[0] = 5; h[4] = 21; h[5] = 1; h[6] = 21; h[7] = 1; h[8] = 4; }
    hipMemcpy(states, hs.data(), hs.size(), hipMemcpyHostToDevice);
    TowerParams T{};
    T.nlayers = nl; T.cin_pad = cin_pad; T.cin_last_t = 2; T.F = F;
    for (int l = 0; l < nl; l++) {
        size_t wf = (size_t)9 * (l ? F : cin_pad) * F;
        float* w; hipMalloc(&w, wf * 4);
        std::vector<float> hw(wf);
        for (size_t i = 0; i < wf; i++) hw[i] = 0.01f * (float)((i * 2654435761u) % 97) - 0.45f;
        hipMemcpy(w, hw.data(), wf * 4, hipMemcpyHostToDevice);
        float* b; hipMalloc(&b, F * 4); hipMemset(b, 0, F * 4);
        T.w[l] = w; T.b[l] = b;
    }
    float* out; hipMalloc(&out, (size_t)B * 25 * F * 4);
    unsigned long long* stamps; hipMalloc(&stamps, (size_t)nl * 16 * 8 * 8); hipMemset(stamps, 0, (size_t)nl * 16 * 8 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(g_tower_stamps), &stamps, sizeof(stamps));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<uint32_t> map(25 * 16);
    tower_halo_slotmap(5, c5 ? 8 : 16, c5 ? 37 : 36, map.data());
    uint32_t* dmap; hipMalloc(&dmap, map.size() * 4); hipMemcpy(dmap, map.data(), map.size() * 4, hipMemcpyHostToDevice);
    T.slotmap = dmap; T.halo_pw = c5 ? 8 : 16; T.halo_ps = c5 ? 37 : 36;
    // variant 82: the halo tower with the constant input planes as a per-position bias (TowerParams.cb): layer 0 over 32 channels
    if (variant == 82 || c5) {
        size_t wf = (size_t)9 * 32 * F;
        float* w; hipMalloc(&w, wf * 4);
        std::vector<float> hw(wf);
        for (size_t i = 0; i < wf; i++) hw[i] = 0.01f * (float)((i * 2654435761u) % 97) - 0.45f;
        hipMemcpy(w, hw.data(), wf * 4, hipMemcpyHostToDevice);
        float* S; hipMalloc(&S, (size_t)46 * 9 * F * 4); hipMemset(S, 0, (size_t)46 * 9 * F * 4);
        T.cb = 1; T.cb_cin_pad = 32; T.cb_last_t = 3; T.w0_board = w; T.cplane_sums = S;
    }
    auto launch = [&]() {
        if (c5) return launch_tower_halo_t<13, 8, 2, 8, 5, true, true>(nullptr, (const float*)states, T, out, B, 8);
        if (variant == 82) return launch_tower_halo_t<13, 8, 2, 4, 5, true, true>(nullptr, (const float*)states, T, out, B, 4);
        if (variant == 80) return launch_tower_halo_t<13, 8, 5, 4, 5, true>(nullptr, (const float*)states, T, out, B, 4);
        if (variant == 16) return launch_tower_t<7, 16, 5, 4, true>(nullptr, (const float*)states, T, out, B, n, 16, 4);
        if (variant == 4) return launch_tower_t<25, 4, 5, 4, true>(nullptr, (const float*)states, T, out, B, n, 16, 4);
        return launch_tower_t<13, 8, 5, 4, true>(nullptr, (const float*)states, T, out, B, n, 16, 4);
    };
    for (int i = 0; i < 3; i++) launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 10; i++) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("variant %d waves: %.1f us per launch\n", variant, ms * 100);
    std::vector<unsigned long long> h((size_t)nl * 16 * 8);
    hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
    const int nw = variant >= 80 ? 8 : variant;
    printf("layer wave |  mainloop  epilogue  barrier1  writeback  barrier2 | next-layer start - this start\n");
    for (int l = 0; l < nl - 1; l++)
        for (int w = 0; w < nw; w++) {
            const unsigned long long* s = &h[((size_t)l * 16 + w) * 8];
            const unsigned long long* nx = &h[((size_t)(l + 1) * 16 + w) * 8];
            if (l == 0 || l == 5 || l == 6)
                printf("%5d %4d | %9llu %9llu %9llu %10llu %9llu | %llu\n", l, w, s[1] - s[0], s[2] - s[1], s[3] - s[2], s[4] - s[3], s[5] - s[4], nx[0] - s[0]);
        }
    for (int w = 0; w < nw; w++) {
        const unsigned long long* s = &h[((size_t)0 * 16 + w) * 8];
        if (s[6]) printf("layer 0 wave %d: staging (start -> image ready) %llu cycles\n", w, s[0] - s[6]);
    }
    // averages over layers 1..nl-2 and waves
    double acc[6] = {0, 0, 0, 0, 0, 0};
    int cnt = 0;
    for (int l = 1; l < nl - 1; l++)
        for (int w = 0; w < nw; w++) {
            const unsigned long long* s = &h[((size_t)l * 16 + w) * 8];
            const unsigned long long* nx = &h[((size_t)(l + 1) * 16 + w) * 8];
            for (int k = 0; k < 5; k++) acc[k] += (double)(s[k + 1] - s[k]);
            acc[5] += (double)(nx[0] - s[0]);
            cnt++;
        }
    printf("mean over layers 1..%d, all waves: mainloop %.0f  epilogue %.0f  barrier1 %.0f  writeback %.0f  barrier2 %.0f | layer %.0f cycles (s_memtime ticks)\n",
           nl - 2, acc[0] / cnt, acc[1] / cnt, acc[2] / cnt, acc[3] / cnt, acc[4] / cnt, acc[5] / cnt);
    return 0;
}
