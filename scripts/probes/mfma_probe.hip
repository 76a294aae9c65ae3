// Micro-probe: achievable v_mfma_f32_16x16x4_f32 rate for the wave shapes the conv / FC kernels use.
// variants: 0 = registers only; 1 = + 13 ds_read_b128 per 52 MFMAs (conflict-free); 2 = same with 2-way conflicts
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
using f32x4 = __attribute__((ext_vector_type(4))) float;
template <int VAR, int NT>
__global__ __launch_bounds__(512) void probe(float* out, int iters) {
    __shared__ f32x4 lds[8192];
    int tid = threadIdx.x, lane = tid & 63, r16 = lane & 15, q = lane >> 4;
    for (int i = tid; i < 8192; i += 512) lds[i] = f32x4{1.0f + i, 0.5f, 0.25f, 2.0f};
    __syncthreads();
    f32x4 acc[NT];
    for (int j = 0; j < NT; j++) acc[j] = f32x4{0, 0, 0, 0};
    f32x4 a[NT];
    for (int j = 0; j < NT; j++) a[j] = lds[(j * 16 + r16) * 17 + q];
    f32x4 w = lds[tid & 1023];
    for (int it = 0; it < iters; it++) {
        if (VAR >= 1) {
#pragma unroll
            for (int j = 0; j < NT; j++) {
                int idx = VAR == 1 ? ((q * 1024 + (it & 3) * 208 + j * 16 + r16) & 8191) : ((((j * 16 + r16) * 17 + q + (it & 3) * 4)) & 8191);
                a[j] = lds[idx];
            }
        }
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int j = 0; j < NT; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t], a[j][t], acc[j], 0, 0, 0);
    }
    f32x4 s = f32x4{0, 0, 0, 0};
    for (int j = 0; j < NT; j++) s += acc[j];
    out[blockIdx.x * 512 + tid] = s[0] + s[1] + s[2] + s[3];
}
// closer to the real conv loop: per-lane offset array, weights streamed from global, taps switched every 4 chunks
template <int VAR, int NT>
__global__ __launch_bounds__(512) void probe2(float* out, const f32x4* __restrict__ wglob, int iters, int n) {
    extern __shared__ f32x4 ldsd[];
    int tid = threadIdx.x, lane = tid & 63, r16 = lane & 15, q = lane >> 4, wave = tid >> 6;
    const int LS4 = 17, rows = 400;
    for (int i = tid; i < 401 * LS4; i += 512) ldsd[i] = f32x4{1.0f + i, 0.5f, 0.25f, 2.0f};
    __syncthreads();
    int rho0 = (wave >> 2) * NT * 16 + r16;
    int pyx[NT];
    for (int j = 0; j < NT; j++) { int rho = rho0 + j * 16; int sq = rho % 25; int y = sq / 5, x = sq % 5; pyx[j] = rho < rows ? (y | (x << 8)) : 0x7f7f; }
    f32x4 acc[NT];
    for (int j = 0; j < NT; j++) acc[j] = f32x4{0, 0, 0, 0};
    int aoff[NT];
    const f32x4* wp = wglob + ((wave & 3) * 16 + r16) * 4 + q;
    f32x4 w = wp[0];
    int kk = 0;
    for (int it = 0; it < iters; it++) {
        for (int tap = 0; tap < 9; tap++) {
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
#pragma unroll
            for (int j = 0; j < NT; j++) {
                int yy = (pyx[j] & 0xff) + dy, xx = (pyx[j] >> 8) + dx;
                bool ok = (VAR & 4) ? true : (yy >= 0 && yy < n && xx >= 0 && xx < n);
                aoff[j] = ok ? (rho0 + j * 16 + ((VAR & 4) ? 0 : dy * n + dx)) * LS4 + q : rows * LS4 + q;
            }
            for (int kc = 0; kc < 4; kc++) {
                f32x4 a[NT];
#pragma unroll
                for (int j = 0; j < NT; j++) a[j] = ldsd[aoff[j] + kc * 4];
                f32x4 wn = (VAR & 2) ? wp[(size_t)((kk + 1) % 36) * 256] : w;
#pragma unroll
                for (int t = 0; t < 4; t++)
#pragma unroll
                    for (int j = 0; j < NT; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t], a[j][t], acc[j], 0, 0, 0);
                w = wn; kk++;
            }
        }
    }
    f32x4 s = f32x4{0, 0, 0, 0};
    for (int j = 0; j < NT; j++) s += acc[j];
    out[blockIdx.x * 512 + tid] = s[0] + s[1] + s[2] + s[3];
}
template <int VAR, int NT>
void run2(const char* name) {
    float* d; hipMalloc(&d, 256 * 512 * 4);
    f32x4* wg; hipMalloc(&wg, 36 * 256 * 16 * 2); hipMemset(wg, 0, 36 * 256 * 16 * 2);
    int iters = 100;
    size_t lds = 401 * 17 * 16;
    hipFuncSetAttribute((const void*)probe2<VAR, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe2<VAR, NT><<<256, 512, lds>>>(d, wg, 2, 5);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe2<VAR, NT><<<256, 512, lds>>>(d, wg, iters, 5);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = 256.0 * 8 * iters * 36 * 4 * NT * 2048.0;
    printf("%-48s NT %d: %.3f ms  %.1f TFLOP/s (%.1f%% of 157.3)\n", name, NT, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100);
}
template <int VAR, int NT>
void run(const char* name, int threads) {
    float* d; hipMalloc(&d, 256 * 512 * 4);
    int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<VAR, NT><<<256, threads>>>(d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<VAR, NT><<<256, threads>>>(d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = 256.0 * (threads / 64) * iters * 4 * NT * 2048.0;
    printf("%-40s threads %d NT %d: %.3f ms  %.1f TFLOP/s (%.1f%% of 157.3)\n", name, threads, NT, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100);
    hipFree(d);
}
// register tile of k_fc_reg: 4 activation fragments × 3 weight fragments → 12 accumulators, operands in registers only.
// ORDER 0: for t, for i, for j (B operand fixed for 3 MFMAs); 1: for t, for j, for i (A operand fixed for 4); 2: for i, for j, for t
// (the 4 k-slices of one accumulator back to back: a dependent chain of 4)
template <int ORDER>
__global__ __launch_bounds__(512) void probe3(float* out, const f32x4* __restrict__ src, int iters) {
    const int tid = threadIdx.x;
    f32x4 a[4], w[3], acc[4][3];
    for (int i = 0; i < 4; i++) a[i] = src[tid + 512 * i];
    for (int j = 0; j < 3; j++) w[j] = src[tid + 512 * (4 + j)];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 3; j++) acc[i][j] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; it++) {
        if (ORDER == 0) {
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 3; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j][t], a[i][t], acc[i][j], 0, 0, 0);
        } else if (ORDER == 1) {
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int j = 0; j < 3; j++)
#pragma unroll
                    for (int i = 0; i < 4; i++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j][t], a[i][t], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 3; j++)
#pragma unroll
                    for (int t = 0; t < 4; t++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j][t], a[i][t], acc[i][j], 0, 0, 0);
        }
        asm volatile("" : "+v"(a[0]), "+v"(w[0]));  // (keeps the loop a loop)
    }
    f32x4 s = f32x4{0, 0, 0, 0};
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 3; j++) s += acc[i][j];
    out[blockIdx.x * 512 + tid] = s[0] + s[1] + s[2] + s[3];
}
template <int ORDER>
void run3(const char* name, int threads, bool random) {
    float* d; hipMalloc(&d, 256 * 512 * 4);
    f32x4* src; hipMalloc(&src, 512 * 7 * 16);
    float* h = (float*)malloc(512 * 7 * 16);
    for (int i = 0; i < 512 * 7 * 4; i++) h[i] = random ? ((float)rand() / (float)RAND_MAX - 0.5f) * ((i & 1) ? 0.05f : 1.7f) : 0.0f;  // full-mantissa randomness
    hipMemcpy(src, h, 512 * 7 * 16, hipMemcpyHostToDevice);
    int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe3<ORDER><<<256, threads>>>(d, src, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe3<ORDER><<<256, threads>>>(d, src, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = 256.0 * (threads / 64) * iters * 48 * 2048.0;
    printf("%-44s threads %d %s: %.3f ms  %.1f TFLOP/s (%.1f%% of 157.3)\n", name, threads, random ? "random" : "zeros ", ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100);
    hipFree(d); hipFree(src); free(h);
}
int main() {
    run3<0>("4x3 register tile, order t,i,j", 512, true);
    run3<0>("4x3 register tile, order t,i,j", 512, false);
    run3<1>("4x3 register tile, order t,j,i", 512, true);
    run3<2>("4x3 register tile, order i,j,t (chains of 4)", 512, true);
    run3<0>("4x3 register tile, order t,i,j", 256, true);
    run<0, 13>("regs only", 512);
    run<0, 13>("regs only", 256);
    run<0, 4>("regs only", 512);
    run<1, 13>("+13 ds_read_b128 conflict-free", 512);
    run<2, 13>("+13 ds_read_b128 2-way conflicts", 512);
    run<1, 13>("+13 ds_read_b128 conflict-free", 256);
    run2<0, 13>("conv-like: taps + zero rows, w in regs");
    run2<2, 13>("conv-like: taps + zero rows + global w stream");
    run2<4, 13>("conv-like: no shifts/zero rows (conflict pattern only)");
    run2<6, 13>("conv-like: no shifts + global w stream");
    return 0;
}
