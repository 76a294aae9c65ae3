// Stand-alone timing of tg::conv_mainloop (the tower's MFMA loop) on fake data: 256 workgroups × 8 waves,
// LDS image of 400 rows × 64 channels, weights streamed from a 147 KB L2-resident buffer — the C2 shape.
#include <cstdio>
#include "../../tak_amd/csrc/conv_mainloop.cuh"
using namespace tg;
template <int RTW, int CH>
__global__ __launch_bounds__(512) void probe(float* out, const float* __restrict__ wglob, int layers, int n) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    f32x4* lds4 = (f32x4*)lds;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, q = lane >> 4;
    const int rows = 400, nsq = n * n, LS4 = (CH * 16 + LDS_PAD) >> 2;
    for (int i = tid; i < 401 * LS4; i += 512) lds4[i] = f32x4{0.001f * (i & 255), 0.5f, 0.25f, 1.0f};
    __syncthreads();
    const int ct = wave & 3, rg = wave >> 2;
    const int rho0 = rg * RTW * 16 + r16;
    f32x4 tot = f32x4{0, 0, 0, 0};
    for (int l = 0; l < layers; l++) {
        f32x4 acc[RTW];
        for (int j = 0; j < RTW; j++) acc[j] = f32x4{0, 0, 0, 0};
#ifdef STAG
        if (wave >= 4) __builtin_amdgcn_s_sleep(STAG);
#endif
        const f32x4* wp = (const f32x4*)wglob + ((size_t)(ct * 16 + r16) * 4 + q);
        int vmask[RTW];
        conv_tap_masks<RTW>(rows, n, nsq, rho0, vmask);
        conv_mainloop<RTW, CH>(lds4, wp, (size_t)64 * 4, LS4, rows, n, rho0, q, vmask, acc);
        for (int j = 0; j < RTW; j++) tot += acc[j];
    }
    out[blockIdx.x * 512 + tid] = tot[0] + tot[1] + tot[2] + tot[3];
}
int main() {
    float *d, *w;
    hipMalloc(&d, 256 * 512 * 4);
    hipMalloc(&w, 36 * 64 * 16 * 4 + 4096);
    hipMemset(w, 0, 36 * 64 * 16 * 4 + 4096);
    const int layers = 50;
    size_t lds = 401 * 68 * 4;
    hipFuncSetAttribute((const void*)probe<13, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<13, 4><<<256, 512, lds>>>(d, w, 2, 5);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<13, 4><<<256, 512, lds>>>(d, w, layers, 5);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = 256.0 * 8 * layers * 36.0 * 4 * 13 * 2048.0;  // issued MFMA flops (26 tile slots for 25 tiles)
    printf("conv_mainloop<13,4>: %.2f us per layer, %.1f TFLOP/s issued (%.1f%% of 157.3); useful = x25/26\n", ms * 1000 / layers, flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100);
    return 0;
}
