// The small-batch towers (k_tower<RTW, 4 waves>, C2 shape with the constant planes as a bias) under s_memtime stamps: where a layer of
// 7 row tiles (1024 positions, 4 per workgroup) spends its cycles against one of 13 (2048 positions, 8 per workgroup) — the plateau
// of the games sweep (profiles/r06_e_tower_pw_sweep.txt).
// hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DTG_TOWER_STAMPS -I../../tak_amd/csrc tower_small_stamps.hip -o _bin/tower_small_stamps
#include <cstdio>
#include <vector>
#include "../../tak_amd/csrc/net_kernels.hip"
#include "probe_env.h"
using namespace tg;
int main(int argc, char** argv) {
    const int pw = argc > 1 ? atoi(argv[1]) : 4;
    const int n = 5, F = 64, R = 6, nl = 1 + 2 * R, B = 256 * pw;
    uint8_t* states; hipMalloc(&states, (size_t)B * 256);
    std::vector<uint8_t> hs((size_t)B * 256, 0);
    for (int b = 0; b < B; b++) { uint8_t* h = &hs[(size_t)b * 256 + 240]; h[0] = 5; h[4] = 21; h[5] = 1; h[6] = 21; h[7] = 1; h[8] = 4; }
    hipMemcpy(states, hs.data(), hs.size(), hipMemcpyHostToDevice);
    TowerParams T{};
    T.nlayers = nl; T.cin_pad = 80; T.cin_last_t = 2; T.F = F;
    for (int l = 0; l < nl; l++) {
        size_t wf = (size_t)9 * (l ? F : 32) * F;
        float* w; hipMalloc(&w, wf * 4);
        std::vector<float> hw(wf);
        for (size_t i = 0; i < wf; i++) hw[i] = 0.01f * (float)((i * 2654435761u) % 97) - 0.45f;
        hipMemcpy(w, hw.data(), wf * 4, hipMemcpyHostToDevice);
        float* b; hipMalloc(&b, F * 4); hipMemset(b, 0, F * 4);
        T.w[l] = w; T.b[l] = b;
    }
    float* S; hipMalloc(&S, (size_t)46 * 9 * F * 4); hipMemset(S, 0, (size_t)46 * 9 * F * 4);
    T.cb = 1; T.cb_cin_pad = 32; T.cb_last_t = 3; T.w0_board = T.w[0]; T.cplane_sums = S;
    float* out; hipMalloc(&out, (size_t)(B + 16) * 25 * F * 4);
    unsigned long long* stamps; hipMalloc(&stamps, (size_t)nl * 16 * 8 * 8); hipMemset(stamps, 0, (size_t)nl * 16 * 8 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(g_tower_stamps), &stamps, sizeof(stamps));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&]() {
        if (pw == 1) return launch_tower_t<2, 4, 2, 4, true, true>(nullptr, (const float*)states, T, out, B, n, 1, 4);
        if (pw == 2) return launch_tower_t<4, 4, 2, 4, true, true>(nullptr, (const float*)states, T, out, B, n, 2, 4);
        if (pw == 4) return launch_tower_t<7, 4, 2, 4, true, true>(nullptr, (const float*)states, T, out, B, n, 4, 4);
        if (pw == 8) return launch_tower_t<13, 4, 2, 4, true, true>(nullptr, (const float*)states, T, out, B, n, 8, 4);
        return launch_tower_t<13, 8, 2, 4, true, true>(nullptr, (const float*)states, T, out, B, n, 16, 4);
    };
    for (int i = 0; i < 3; i++) launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 10; i++) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%d positions per workgroup, %d positions: %.1f us per launch\n", pw, B, ms * 100);
    std::vector<unsigned long long> h((size_t)nl * 16 * 8);
    hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
    const int nw = pw == 16 ? 8 : 4;
    printf("layer wave |  mainloop  epilogue  barrier1  writeback  barrier2 | next-layer start - this start\n");
    for (int l = 0; l < nl - 1; l++)
        for (int w = 0; w < nw; w++) {
            const unsigned long long* s = &h[((size_t)l * 16 + w) * 8];
            const unsigned long long* nx = &h[((size_t)(l + 1) * 16 + w) * 8];
            if (l == 0 || l == 5 || l == 6)
                printf("%5d %4d | %9llu %9llu %9llu %10llu %9llu | %llu\n", l, w, s[1] - s[0], s[2] - s[1], s[3] - s[2], s[4] - s[3], s[5] - s[4], nx[0] - s[0]);
        }
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
    printf("mean over layers 1..%d, all waves: mainloop %.0f  epilogue %.0f  barrier1 %.0f  writeback %.0f  barrier2 %.0f | layer %.0f ticks (s_memtime, 100 MHz)\n",
           nl - 2, acc[0] / cnt, acc[1] / cnt, acc[2] / cnt, acc[3] / cnt, acc[4] / cnt, acc[5] / cnt);
    return 0;
}
