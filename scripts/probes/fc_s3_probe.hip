// fc_s3_probe.hip — times k_fc_s3b (split-bf16 policy FC, C2 shape: 4096 x 1600 -> 1575) with parts removed.
//   -DFC_PROBE: 0 normal · 1 no MFMAs · 2 no activation loads · 3 no weight staging (LDS written once) · 4 = 1 + 2 + 3
// (round 5: the FC_PROBE masks were removed from the product kernel — commit 6da8f61 has them; this program now times the shipped kernel)
#ifndef FC_PROBE
#define FC_PROBE 0
#endif
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../tak_amd/csrc/net_s3_kernels.hip"
#include "probe_env.h"

int main() {
    using namespace tg;
    const int M = 4096, K = 1600, P = 1575, NP = 1680;
    void *A, *W; float *b, *out;
    hipMalloc(&A, (size_t)M * K * 4); hipMemset(A, 0, (size_t)M * K * 4);
    hipMalloc(&W, (size_t)(K / 32) * NP * 64 * 2); hipMemset(W, 0, (size_t)(K / 32) * NP * 64 * 2);
    hipMalloc((void**)&b, NP * 4); hipMemset(b, 0, NP * 4);
    hipMalloc((void**)&out, (size_t)M * NP * 4);
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) launch_fc_s3(st, (const float*)A, W, b, out, M, K, NP, NP, P);
    hipStreamSynchronize(st);
    hipEventRecord(e0, st);
    const int reps = 20;
    for (int i = 0; i < reps; i++) launch_fc_s3(st, (const float*)A, W, b, out, M, K, NP, NP, P);
    hipEventRecord(e1, st);
    hipStreamSynchronize(st);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("FC_PROBE=%d: %.1f us per launch (%s)\n", FC_PROBE, 1000.0f * ms / reps, hipGetErrorString(hipGetLastError()));
    return 0;
}
