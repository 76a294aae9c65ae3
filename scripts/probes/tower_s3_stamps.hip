// Where a layer of the split-bf16 halo tower spends its cycles: builds net_s3_kernels.hip with -DTG_S3_STAMPS and prints, for
// one workgroup of a 4096-position launch, the s_memtime deltas between the phase boundaries of every layer and wave:
// main loop | bias + ReLU | wait at barrier 1 | skip read + split + write-back | wait at barrier 2.
// usage: tower_s3_stamps [c2|c5|c3]
// hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DTG_S3_STAMPS -I../../tak_amd/csrc tower_s3_stamps.hip ../../tak_amd/csrc/net_kernels.hip -o _bin/tower_s3_stamps
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../tak_amd/csrc/net_s3_kernels.hip"
#include "probe_env.h"
using namespace tg;
int main(int argc, char** argv) {
    const char* cfg = argc > 1 ? argv[1] : "c2";
    const int n = !strcmp(cfg, "c3") ? 6 : 5, F = !strcmp(cfg, "c2") ? 64 : 128, R = !strcmp(cfg, "c2") ? 6 : 10;
    const int B = 4096, nl = 1 + 2 * R, nsq = n * n, sb = n == 5 ? 256 : 352;
    uint8_t* states; hipMalloc(&states, (size_t)B * sb);
    std::vector<uint8_t> hs((size_t)B * sb, 0);
    for (int b = 0; b < B; b++) { uint8_t* h = &hs[(size_t)b * sb + sb - 16]; h[0] = n; h[4] = n == 5 ? 21 : 30; h[5] = 1; h[6] = h[4]; h[7] = 1; h[8] = 4; }
    hipMemcpy(states, hs.data(), hs.size(), hipMemcpyHostToDevice);
    TowerS3Params T{};
    T.nlayers = nl; T.cin_pad = 96; T.F = F;
    for (int l = 0; l < nl; l++) {
        size_t wb = (size_t)9 * (l ? F / 32 : 3) * (F / 16) * 128 * 16;
        void* w; hipMalloc(&w, wb);
        std::vector<uint16_t> hw(wb / 2);
        for (size_t i = 0; i < hw.size(); i++) hw[i] = (uint16_t)(0x3a00 + (i * 2654435761u) % 509 + ((i & 1) << 15));
        hipMemcpy(w, hw.data(), wb, hipMemcpyHostToDevice);
        float* b; hipMalloc(&b, F * 4); hipMemset(b, 0, F * 4);
        T.w[l] = w; T.b[l] = b;
    }
    int pw, ps;
    tower_s3_halo_geometry(n, F, &pw, &ps);
    std::vector<uint32_t> map((size_t)((pw * nsq + 15) / 16) * 16);
    tower_halo_slotmap(n, pw, ps, map.data());
    uint32_t* dmap; hipMalloc(&dmap, map.size() * 4); hipMemcpy(dmap, map.data(), map.size() * 4, hipMemcpyHostToDevice);
    T.slotmap = dmap; T.halo_ps = ps;
    float* out; hipMalloc(&out, (size_t)B * nsq * F * 4);
    unsigned long long* stamps; hipMalloc(&stamps, (size_t)nl * 16 * 8 * 8); hipMemset(stamps, 0, (size_t)nl * 16 * 8 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(g_s3_stamps), &stamps, sizeof(stamps));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&]() { return launch_tower_s3_states(nullptr, states, T, out, B, n, true); };
    for (int i = 0; i < 3; i++) launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 10; i++) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%s: %.1f us per launch\n", cfg, ms * 100);
    std::vector<unsigned long long> h((size_t)nl * 16 * 8);
    hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
    const int nw = F == 64 ? 4 : 8;
    printf("staging (wave 0): %llu cycles\n", h[0] - h[6]);
    printf("layer wave |  mainloop  bias+relu  barrier1  writeback  barrier2 | next-layer start - this start\n");
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
    printf("mean over layers 1..%d, all waves: mainloop %.0f  bias+relu %.0f  barrier1 %.0f  writeback %.0f  barrier2 %.0f | layer %.0f cycles\n",
           nl - 2, acc[0] / cnt, acc[1] / cnt, acc[2] / cnt, acc[3] / cnt, acc[4] / cnt, acc[5] / cnt);
    printf("whole kernel, wave 0: %llu cycles\n", h[((size_t)(nl - 1) * 16) * 8 + 5] - h[6]);
    return 0;
}
