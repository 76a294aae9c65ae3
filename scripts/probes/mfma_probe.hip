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
// ------------------------------------------------------------------------------------------------------------------
// Winograd F(2x2, 3x3) for one F -> F layer of the C2 tower (16 positions per workgroup, 64 -> 64, 5x5 board): a stand-in for
// its MAIN LOOP that keeps every cost the real one has and drops what could only make it slower.  9 output tiles of 2x2 per
// position -> 144 (position, tile) rows = 9 MFMA row tiles; 16 transformed elements x 4 output-channel tiles; K = 64.
// A wave owns (row tile, channel tile) pairs with all 16 elements (64 accumulator registers: the output transform A^T M A needs the
// 16 elements of a row in ONE wave — there is no LDS left to exchange them: the image fills it); 36 pairs: waves 0-3 take 5,
// waves 4-7 take 4 (9 per SIMD).  Per (pair, 16-channel chunk) UNIT: the 4x4 input patch of the lane's row from the LDS image
// (16 ds_read_b128 at immediate offsets), V = B^T d B in registers (32 f32x4 add/sub), the 16 transformed-weight fragments of the
// channel tile (16 x 1 KB from the 256 KB U table of the layer: L1 / L2), 64 MFMAs.  Not modelled: the output transform
// (24 f32x4 add/sub per pair), the wider halo the 6x6 coverage needs (it does not fit beside 16 positions), bias / ReLU / skip.
// VAR bit 0: no weight stream (U fragments stay in registers); bit 1: no input transform; bit 2: no LDS reads.
// ------------------------------------------------------------------------------------------------------------------
template <int VAR>
__global__ __launch_bounds__(512) void probe_wino(float* out, const f32x4* __restrict__ U, int layers) {
    extern __shared__ f32x4 img[];  // 583 cells x 17 slots (the tower's halo image, 158.6 KB)
    const int tid = threadIdx.x, lane = tid & 63, r16 = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 583 * 17; i += 512) img[i] = f32x4{0.001f * (i % 97), 0.5f, -0.25f, 0.002f * (i % 31)};
    __syncthreads();
    const int ct = wave & 3, npairs = wave < 4 ? 5 : 4, rt0 = wave < 4 ? 0 : 5;
    const f32x4* ub = U + ((size_t)(ct * 16 + r16) * 4 + q);  // fragment (element e, chunk c) at ((e*4 + c)*64 + col)*4 + q slots
    f32x4 s = f32x4{0, 0, 0, 0};
    for (int layer = 0; layer < layers; layer++) {
        for (int pr = 0; pr < npairs; pr++) {
            const int row = (rt0 + pr) * 16 + r16;      // (position, tile)
            const int pos = row / 9, tile = row - pos * 9, ty = tile / 3, tx = tile - ty * 3;
            int base = (7 + pos * 36 + (2 * ty) * 6 + 2 * tx) * 17 + q;  // top-left cell of the 4x4 patch
            base = min(base, (583 - 22) * 17);                           // (the last tiles would reach past the image: clamp)
            f32x4 acc[16];
#pragma unroll
            for (int e = 0; e < 16; e++) acc[e] = f32x4{0, 0, 0, 0};
#pragma unroll 1
            for (int c = 0; c < 4; c++) {
                f32x4 d[4][4], w[16];
#pragma unroll
                for (int e = 0; e < 16; e++) w[e] = (VAR & 1) ? f32x4{0.01f * e, 0.02f, 0.03f, 0.04f} : ub[(size_t)((e * 4 + c) * 64) * 4 + (size_t)layer * 0];
#pragma unroll
                for (int y = 0; y < 4; y++)
#pragma unroll
                    for (int x = 0; x < 4; x++) d[y][x] = (VAR & 4) ? f32x4{0.1f * y, 0.2f * x, 0.3f, 0.4f} : img[base + (y * 6 + x) * 17 + c * 4];
                f32x4 v[4][4];
                if (VAR & 2) {
#pragma unroll
                    for (int y = 0; y < 4; y++)
#pragma unroll
                        for (int x = 0; x < 4; x++) v[y][x] = d[y][x];
                } else {  // V = B^T d B,  B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
                    f32x4 t[4][4];
#pragma unroll
                    for (int x = 0; x < 4; x++) {
                        t[0][x] = d[0][x] - d[2][x];
                        t[1][x] = d[1][x] + d[2][x];
                        t[2][x] = d[2][x] - d[1][x];
                        t[3][x] = d[1][x] - d[3][x];
                    }
#pragma unroll
                    for (int y = 0; y < 4; y++) {
                        v[y][0] = t[y][0] - t[y][2];
                        v[y][1] = t[y][1] + t[y][2];
                        v[y][2] = t[y][2] - t[y][1];
                        v[y][3] = t[y][1] - t[y][3];
                    }
                }
#pragma unroll
                for (int t4 = 0; t4 < 4; t4++)
#pragma unroll
                    for (int e = 0; e < 16; e++) acc[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[e][t4], v[e >> 2][e & 3][t4], acc[e], 0, 0, 0);
            }
#pragma unroll
            for (int e = 0; e < 16; e++) s += acc[e];
        }
        asm volatile("" : "+v"(s));
    }
    out[blockIdx.x * 512 + tid] = s[0] + s[1] + s[2] + s[3];
}
template <int VAR>
void run_wino(const char* name) {
    float* d; hipMalloc(&d, 256 * 512 * 4);
    f32x4* U; hipMalloc(&U, 16 * 4 * 64 * 4 * 16);
    float* h = (float*)malloc(16 * 4 * 64 * 4 * 16);
    for (int i = 0; i < 16 * 4 * 64 * 4 * 4; i++) h[i] = ((float)rand() / (float)RAND_MAX - 0.5f) * 0.1f;
    hipMemcpy(U, h, 16 * 4 * 64 * 4 * 16, hipMemcpyHostToDevice);
    const size_t lds = 583 * 17 * 16;
    hipFuncSetAttribute((const void*)probe_wino<VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms[2];
    for (int k = 0; k < 2; k++) {
        const int layers = k == 0 ? 12 : 36;
        probe_wino<VAR><<<256, 512, lds>>>(d, U, 2);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int rep = 0; rep < 10; rep++) probe_wino<VAR><<<256, 512, lds>>>(d, U, layers);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms[k], e0, e1);
        ms[k] /= 10;
    }
    const double per_layer_us = (ms[1] - ms[0]) / 24 * 1000;
    printf("%-58s %.1f us per layer (direct conv_mainloop_halo: 51; MFMA issue alone: %.1f)\n", name, per_layer_us, 9216.0 / 4 * 32 / 2.4e3);
    hipFree(d); hipFree(U); free(h);
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
    run_wino<0>("Winograd F(2x2,3x3) main-loop stand-in, everything");
    run_wino<1>("  without the transformed-weight stream");
    run_wino<2>("  without the input transform");
    run_wino<4>("  without the LDS reads");
    run_wino<7>("  MFMAs only");
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
