// Round 5 feasibility probe for a training convolution whose workgroup runs TWO independent wave groups (4 waves each, one wave of
// each group per SIMD) on two sub-images — 5 positions = 8 row tiles and 3 positions = 5 row tiles, two channel tiles per wave — so
// that one group's staging / epilogue hides under the other group's MFMAs.  r04_b_fc_candidates.txt 2 found that two waves of a SIMD
// running DIFFERENT unrolled register-fed loops did not share the matrix pipe (478 k cycles against 310 k / 338 k unmixed).  Is that
// so for the LDS-fed conv loop?  One workgroup per CU, v_mfma_f32_16x16x4_f32, B operands from LDS (one ds_read_b128 per tile and
// 16-k step), weights in registers, per image 72 steps; idle phases (s_sleep) stand in for staging / epilogue.
//   variant 0: all 8 waves the shipped shape (13 tiles x 1 channel tile), 2 images, pauses of 24 k cycles between images  [today]
//   variant 1: waves 0-3: 8 tiles x 2 channel tiles, waves 4-7: 5 tiles x 2 channel tiles, 2 images each, pauses 10 k / 6 k   [proposal]
//   variant 2: as 1 without pauses;  variant 3: as 0 without pauses
// prints cycles per workgroup (s_memtime of wave 0 .. last wave's end) against the MFMA issue floor (2 waves x MFMAs x 32 cycles per SIMD).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 scripts/probes/mixed_loop_probe.hip -o scripts/probes/_bin/mixed_loop_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int RT, int CT>
__device__ __forceinline__ void image(const f32x4* lds, int base, f32x4 (&acc)[RT][CT], const f32x4 (&w)[CT], int steps) {
    constexpr int H1 = (RT + 1) / 2;
    f32x4 a[RT];
#pragma unroll
    for (int j = 0; j < H1; j++) a[j] = lds[base + j * 16 * 33];
#pragma unroll 1
    for (int s3 = 0; s3 < steps; s3 += 8) {
#pragma unroll
        for (int s = 0; s < 8; s++) {
#pragma unroll
            for (int j = H1; j < RT; j++) a[j] = lds[base + j * 16 * 33 + s * 4];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int j = 0; j < H1; j++)
#pragma unroll
                    for (int c = 0; c < CT; c++) acc[j][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[c][t], a[j][t], acc[j][c], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < H1; j++) a[j] = lds[base + j * 16 * 33 + ((s + 1) & 7) * 4];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int j = H1; j < RT; j++)
#pragma unroll
                    for (int c = 0; c < CT; c++) acc[j][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[c][t], a[j][t], acc[j][c], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}
__device__ __forceinline__ void pause(int cycles) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while ((long long)(__builtin_amdgcn_s_memtime() - t0) < cycles) __builtin_amdgcn_s_sleep(8);
}
template <int RT, int CT>
__device__ __forceinline__ float run(const f32x4* lds, int lane, int images, int pause_cycles) {
    f32x4 acc[RT][CT], w[CT];
#pragma unroll
    for (int c = 0; c < CT; c++) w[c] = f32x4{0.01f * lane, 0.02f, 0.03f + c, 0.04f};
#pragma unroll
    for (int j = 0; j < RT; j++)
#pragma unroll
        for (int c = 0; c < CT; c++) acc[j][c] = f32x4{0, 0, 0, 0};
    const int base = (lane & 15) * 33 + (lane >> 4);
    for (int i = 0; i < images; i++) {
        if (pause_cycles) pause(pause_cycles);
        image<RT, CT>(lds, base, acc, w, 72);
    }
    f32x4 s = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < RT; j++)
#pragma unroll
        for (int c = 0; c < CT; c++) s += acc[j][c];
    return s[0] + s[1] + s[2] + s[3];
}
template <int VAR>
__global__ __launch_bounds__(512) void probe(float* out, unsigned long long* stamps) {
    extern __shared__ f32x4 lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < 208 * 33 + 64; i += 512) lds[i] = f32x4{1.0f + (i & 7), 0.5f, 0.25f, 2.0f};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float r;
    if (VAR == 0 || VAR == 3) r = run<13, 1>(lds, lane, 2, VAR == 0 ? 24000 : 0);
    else if (wave < 4) r = run<8, 2>(lds, lane, 2, VAR == 1 ? 10000 : 0);
    else r = run<5, 2>(lds, lane, 2, VAR == 1 ? 6000 : 0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && blockIdx.x == 0) { stamps[2 * wave] = t0; stamps[2 * wave + 1] = t1; }
    out[blockIdx.x * 512 + tid] = r;
}
template <int VAR>
void go(const char* what, long long mfma_per_simd) {
    float* out; unsigned long long* st;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&st, 16 * 8);
    const size_t lds = (208 * 33 + 64) * 16;
    hipFuncSetAttribute((const void*)probe<VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(probe<VAR>, dim3(256), dim3(512), lds, nullptr, out, st);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int i = 0; i < 10; i++) hipLaunchKernelGGL(probe<VAR>, dim3(256), dim3(512), lds, nullptr, out, st);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[16];
    hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
    unsigned long long lo = h[0], hi = h[1];
    for (int w = 0; w < 8; w++) { lo = h[2 * w] < lo ? h[2 * w] : lo; hi = h[2 * w + 1] > hi ? h[2 * w + 1] : hi; }
    printf("%-62s %7.1f us per launch | workgroup 0: %8llu cycles (s_memtime) | waves end at", what, ms * 100, hi - lo);
    for (int w = 0; w < 8; w++) printf(" %llu", h[2 * w + 1] - lo);
    printf(" | MFMA issue floor %lld cycles\n", mfma_per_simd * 32);
    hipFree(out); hipFree(st);
}
int main() {
    // per SIMD: two waves.  shipped: 2 x (2 images x 72 steps x 52 MFMAs); proposal: (2 x 72 x 64) + (2 x 72 x 40)
    go<3>("all waves 13 x 1, no pauses", 2LL * 2 * 72 * 52);
    go<0>("all waves 13 x 1, 24 k-cycle pause before every image", 2LL * 2 * 72 * 52);
    go<2>("waves 0-3: 8 x 2, waves 4-7: 5 x 2, no pauses", 2LL * 72 * 64 + 2LL * 72 * 40);
    go<1>("waves 0-3: 8 x 2 (10 k pauses), waves 4-7: 5 x 2 (6 k pauses)", 2LL * 72 * 64 + 2LL * 72 * 40);
    return 0;
}
