// tower_s3_probe.hip — times k_tower_s3 (C2 shape: 4096 positions, 5x5, F = 64, 13 layers) with parts removed, to see
// where the time outside the MFMAs goes.  Build (one binary per variant):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DS3_PROBE=<v> -I tak_amd/csrc scripts/probes/tower_s3_probe.hip -o /tmp/p<v>
// variants: 0 normal · 1 no epilogue store/skip (image untouched) · 2 weights of chunk 0 reused (no weight stream)
//           3 activations of one LDS slot reused (no per-chunk ds_read) · 4 no MFMAs · 5 = 4 + 2 · 6 = 4 + 2 + 3
// (round 5: the S3_PROBE masks were removed from the product kernel — commit 6da8f61 has them; this program now times the shipped kernel)
#ifndef S3_PROBE
#define S3_PROBE 0
#endif
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../tak_amd/csrc/net_s3_kernels.hip"
#include "probe_env.h"

int main() {
    using namespace tg;
    const int B = 4096, n = 5, F = 64, R = 6, L = 1 + 2 * R;
    TowerS3Params T{};
    T.nlayers = L; T.cin_pad = 80; T.F = F;
    std::vector<void*> bufs;
    for (int l = 0; l < L; l++) {
        int KC = l == 0 ? 3 : 2;
        size_t bytes = (size_t)9 * KC * F * 64 * 2;
        void* w; hipMalloc(&w, bytes); hipMemset(w, 0, bytes);
        float* b; hipMalloc((void**)&b, F * 4); hipMemset(b, 0, F * 4);
        T.w[l] = w; T.b[l] = b;
    }
    uint8_t* states; hipMalloc((void**)&states, (size_t)B * 256); hipMemset(states, 0, (size_t)B * 256);
    float* out; hipMalloc((void**)&out, (size_t)B * 25 * F * 4);
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) launch_tower_s3_states(st, states, T, out, B, n, true);
    hipStreamSynchronize(st);
    hipEventRecord(e0, st);
    const int reps = 20;
    for (int i = 0; i < reps; i++) launch_tower_s3_states(st, states, T, out, B, n, true);
    hipEventRecord(e1, st);
    hipStreamSynchronize(st);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("variant %d: %.1f us per launch (%s)\n", S3_PROBE, 1000.0f * ms / reps, hipGetErrorString(hipGetLastError()));
    return 0;
}
