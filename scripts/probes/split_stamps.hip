// k_tower_split under s_memtime stamps (workgroup 0: position 0, channel group 0): per layer the MFMA chain, the slice store up to the
// barrier, the wait for the siblings, the staging of the next image.  Net6 shape: 6x6, 16 blocks x 128 filters, 32 positions.
// hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DTG_TOWER_STAMPS -I../../tak_amd/csrc split_stamps.hip -o _bin/split_stamps
#include <cstdio>
#include <vector>
#include "../../tak_amd/csrc/net_kernels.hip"
#include "probe_env.h"
using namespace tg;
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 6, B = argc > 2 ? atoi(argv[2]) : 32;
    const int F = 128, R = n == 6 ? 16 : 8, nl = 1 + 2 * R, nsq = n * n, sb = n == 6 ? 384 : 256;
    uint8_t* states; hipMalloc(&states, (size_t)B * sb);
    std::vector<uint8_t> hs((size_t)B * sb, 0);
    for (int b = 0; b < B; b++) { uint8_t* h = &hs[(size_t)b * sb + sb - 16]; h[0] = n; h[4] = n == 6 ? 30 : 21; h[5] = 1; h[6] = h[4]; h[7] = 1; h[8] = 4; }
    hipMemcpy(states, hs.data(), hs.size(), hipMemcpyHostToDevice);
    TowerParams T{};
    T.nlayers = nl; T.cin_pad = n == 6 ? 96 : 80; T.cin_last_t = 3; T.F = F;
    for (int l = 0; l < nl; l++) {
        size_t wf = (size_t)9 * (l ? F : 32) * F;
        float* w; hipMalloc(&w, wf * 4);
        std::vector<float> hw(wf);
        for (size_t i = 0; i < wf; i++) hw[i] = 0.01f * (float)((i * 2654435761u) % 97) - 0.45f;
        hipMemcpy(w, hw.data(), wf * 4, hipMemcpyHostToDevice);
        float* b; hipMalloc(&b, F * 4); hipMemset(b, 0, F * 4);
        T.w[l] = w; T.b[l] = b;
    }
    float* S; hipMalloc(&S, (size_t)64 * 9 * F * 4); hipMemset(S, 0, (size_t)64 * 9 * F * 4);
    T.cb = 1; T.cb_cin_pad = 32; T.cb_last_t = 3; T.w0_board = T.w[0]; T.cplane_sums = S;
    unsigned* ctl; hipMalloc(&ctl, TOWER_SPLIT_CTL_WORDS * 4); hipMemset(ctl, 0, TOWER_SPLIT_CTL_WORDS * 4);
    T.split_flags = ctl; T.split_err = (int*)ctl + TOWER_SPLIT_CTL_WORDS - 32;
    float *out, *scratch; hipMalloc(&out, (size_t)(B + 16) * nsq * F * 4); hipMalloc(&scratch, (size_t)2 * TOWER_SPLIT_MAX_BATCH * nsq * F * 4);
    unsigned long long* stamps; hipMalloc(&stamps, (size_t)nl * 16 * 8 * 8); hipMemset(stamps, 0, (size_t)nl * 16 * 8 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(g_tower_stamps), &stamps, sizeof(stamps));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&]() { return launch_tower_states(nullptr, states, T, out, B, n, scratch); };
    for (int i = 0; i < 3; i++) launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 20; i++) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    int err; hipMemcpy(&err, T.split_err, 4, hipMemcpyDeviceToHost);
    printf("%dx%d, %d layers x %d filters, %d positions: %.1f us per launch (error word %d)\n", n, n, nl, F, B, ms * 50, err);
    std::vector<unsigned long long> h((size_t)nl * 16 * 8);
    hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
    const int nw = n == 6 ? 3 : 2;
    printf("layer wave |   chain  epilogue+store+barrier  atomic+wait  staging | layer total   (s_memtime ticks)\n");
    double acc[5] = {0, 0, 0, 0, 0};
    int cnt = 0;
    for (int l = 1; l < nl - 1; l++)
        for (int w = 0; w < nw; w++) {
            const unsigned long long* s = &h[((size_t)l * 16 + w) * 8];
            if (l == 1 || l == 10 || l == 20) printf("%5d %4d | %7llu %14llu %16llu %10llu | %llu\n", l, w, s[1] - s[0], s[2] - s[1], s[3] - s[2], s[4] - s[3], s[4] - s[0]);
            for (int k = 0; k < 4; k++) acc[k] += (double)(s[k + 1] - s[k]);
            acc[4] += (double)(s[4] - s[0]);
            cnt++;
        }
    printf("mean over layers 1..%d: chain %.0f  epilogue+store+barrier %.0f  atomic+wait %.0f  staging %.0f | layer %.0f ticks; 100 MHz ticks? launch/layers = %.2f us per layer\n",
           nl - 2, acc[0] / cnt, acc[1] / cnt, acc[2] / cnt, acc[3] / cnt, acc[4] / cnt, ms * 50 / nl);
    return 0;
}
