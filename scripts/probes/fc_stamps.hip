// Where a K-step of the barrier version of the policy FC (k_fc_lds, C2 shape: 4096 × 1600 → 1664; run with TG_FC_BARRIER=1)
// spends its cycles: s_memtime stamps of workgroup 0.
// hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DTG_TOWER_STAMPS -I../../tak_amd/csrc fc_stamps.hip -o _bin/fc_stamps
#include <cstdio>
#include <vector>
#include "../../tak_amd/csrc/net_kernels.hip"
using namespace tg;
int main() {
    const int M = 4096, K = 1600, NP = 1664;
    float *A, *W, *bias, *out;
    hipMalloc(&A, (size_t)M * K * 4); hipMemset(A, 0, (size_t)M * K * 4);
    hipMalloc(&W, (size_t)K * NP * 4); hipMemset(W, 0, (size_t)K * NP * 4);
    hipMalloc(&bias, NP * 4); hipMemset(bias, 0, NP * 4);
    hipMalloc(&out, (size_t)M * NP * 4);
    unsigned long long* stamps; hipMalloc(&stamps, 32 * 16 * 8 * 8); hipMemset(stamps, 0, 32 * 16 * 8 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(g_tower_stamps), &stamps, sizeof(stamps));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) launch_gemm(nullptr, A, K, W, bias, out, M, K, NP, NP, 1576);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 10; i++) launch_gemm(nullptr, A, K, W, bias, out, M, K, NP, NP, 1576);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("k_fc_lds: %.1f us per launch (ideal at 157.3 TF: %.1f us)\n", ms * 100, 2.0 * M * K * NP / 157.3e12 * 1e6);
    std::vector<unsigned long long> h(32 * 16 * 8);
    hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
    printf("step wave |  compute  stage_store  barrier | step total\n");
    double acc[4] = {0, 0, 0, 0}; int cnt = 0;
    for (int st = 1; st < 24; st++)
        for (int w = 0; w < 8; w++) {
            const unsigned long long* s = &h[((size_t)st * 16 + w) * 8];
            const unsigned long long* nx = &h[((size_t)(st + 1) * 16 + w) * 8];
            if (st == 10) printf("%4d %4d | %8llu %8llu %8llu | %llu\n", st, w, s[1] - s[0], s[2] - s[1], s[3] - s[2], nx[0] - s[0]);
            acc[0] += s[1] - s[0]; acc[1] += s[2] - s[1]; acc[2] += s[3] - s[2]; acc[3] += nx[0] - s[0]; cnt++;
        }
    printf("mean: compute %.0f  stage_store %.0f  barrier %.0f | step %.0f cycles (ideal 2 waves x 208 MFMAs x 32 = 13312)\n", acc[0] / cnt, acc[1] / cnt, acc[2] / cnt, acc[3] / cnt);
    return 0;
}
