// Stand-alone timing of k_fc_reg's main loop (tak_amd/csrc/fc_reg.cuh) on the C2 shape — M = 4096 rows, K = 1600, 99 output
// tiles — with synthetic operands in the product's layouts, s_memtime / s_memrealtime stamps around the loop of every wave.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off [-DTG_FR_PROBE=mask] scripts/probes/fc_reg_probe.hip -o scripts/probes/_bin/fc_reg_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fc_reg.cuh"
using namespace tg;

__global__ __launch_bounds__(512) void k_probe(const float* __restrict__ A, const float* __restrict__ Wp, float* __restrict__ out, int M, int K,
                                               int NP, uint64_t* __restrict__ stamps, int mode) {
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int r16 = lane & 15, q = lane >> 4;
    const int h = wave >> 2, cg = wave & 3;
    const int cb = blockIdx.y;
    const int xrt = cb & 7, ie = xrt & 3;
    const bool has_x = mode == 1 ? false : mode == 2 ? true : (h == (xrt >> 2) && cg < FR_NX);  // 1: no wave / 2: every wave runs the 13-tile loop
    const int nchunks = K >> 4;
    const f32x4* ap[4];
    int rt[4];
    for (int i = 0; i < 4; i++) {
        rt[i] = blockIdx.x * 8 + 4 * h + ((i + ie) & 3);
        ap[i] = (const f32x4*)A + (size_t)rt[i] * nchunks * 64 + r16 * 4 + q;
    }
    const f32x4* wg = (const f32x4*)Wp;
    const size_t wchunk = (size_t)NP * 4;
    const int ct0 = cb * FR_MAIN + cg * 3;
    const f32x4* wb = wg + ((size_t)(ct0 * 16 + r16) * 4 + q);
    const f32x4* wx = wg + ((size_t)((FR_XT0 + (cg < FR_NX ? cg : 0)) * 16 + r16) * 4 + q);
    f32x4 acc[4][3], accx = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 3; j++) acc[i][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    const uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (has_x) fc_reg_loop<true>(ap, 64, wb, wchunk, wx, nchunks, acc, accx);
    else fc_reg_loop<false>(ap, 64, wb, wchunk, wx, nchunks, acc, accx);
    f32x4 s = accx;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 3; j++) s += acc[i][j];
    asm volatile("" : "+v"(s));
    const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 512 + tid] = s[0] + s[1] + s[2] + s[3];
    if (lane == 0) {
        uint64_t* st = stamps + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave) * 4;
        st[0] = t0; st[1] = t1; st[2] = r0; st[3] = (r1 & 0xFFFFFFFFFFFFull) | ((uint64_t)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) << 48);  // HW_REG_HW_ID in the top bits
    }
}

int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const int M = 4096, K = 1600, NP = 1664;
    std::vector<float> hA((size_t)M * K), hW((size_t)K * NP);
    for (auto& v : hA) v = std::max(0.0f, (float)rand() / (float)RAND_MAX - 0.4f);
    for (auto& v : hW) v = ((float)rand() / (float)RAND_MAX - 0.5f) * 0.05f;
    float *A, *W, *out;
    uint64_t* stamps;
    hipMalloc(&A, hA.size() * 4); hipMalloc(&W, hW.size() * 4); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&stamps, 256 * 8 * 4 * 8);
    hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        for (int i = 0; i < 20; i++) k_probe<<<dim3(32, 8), 512>>>(A, W, out, M, K, NP, stamps, mode);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("launch: %.1f us\n", ms * 1000 / 20);
    }
    std::vector<uint64_t> st(256 * 8 * 4);
    hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> cyc, clk;
    for (int w = 0; w < 256 * 8; w++) {
        cyc.push_back((double)(st[w * 4 + 1] - st[w * 4]));
        clk.push_back((double)(st[w * 4 + 1] - st[w * 4]) / (double)((st[w * 4 + 3] & 0xFFFFFFFFFFFFull) - st[w * 4 + 2]) * 100.0);  // MHz
    }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    printf("wave loop cycles: min %.0f median %.0f max %.0f; MFMA issue needs %d per SIMD (2 waves)\n", cyc[0], cyc[cyc.size() / 2], cyc.back(), 2 * 100 * 50 * 32);
    printf("shader clock inside the loop: median %.0f MHz (min %.0f max %.0f)\n", clk[clk.size() / 2], clk[0], clk.back());
    for (int g = 0; g < 256; g += 97)
        for (int w = 0; w < 8; w++) {
            const uint64_t* x = &st[(g * 8 + w) * 4];
            const unsigned hw = (unsigned)(x[3] >> 48);
            printf("  workgroup %3d wave %d: SIMD %u CU %u  start +%6.0f  %.0f cycles\n", g, w, (hw >> 4) & 3, (hw >> 8) & 15, (double)(x[0] - st[g * 8 * 4]), (double)(x[1] - x[0]));
        }
    return 0;
}
