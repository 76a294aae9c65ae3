// train_kernels.hip — kernels of the training step (Network::train_inner, reference
// alpha-tak/src/model/network.rs:58-97): BatchNorm in training mode, log-softmax / MSE losses and their
// gradients, the weight-gradient (TN) implicit GEMM on f32 MFMA, Adam, and the re-layout of the master
// parameters (tch layouts) into the fragment layouts of the forward / data-gradient kernels.
//
// The forward convolutions and the data gradients reuse k_conv3x3 / k_conv_pos / k_gemm of net_kernels.hip:
// a data gradient is the same implicit GEMM with the taps flipped and the channel roles swapped, which is
// only a different packing of the weights (k_pack_conv_bwd).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "conv_mainloop.cuh"
#include "kernels.h"

namespace tg {

// ------------------------------------------------------------------------------------------------
// Parameter re-layout (master parameters stay in tch layout: conv OIHW, linear [out, in])
// ------------------------------------------------------------------------------------------------
// forward fragments: dst[((k>>4)·OP + o)·16 + (k&15)], k = tap·Ipad + c
__global__ void k_pack_conv_fwd(const float* __restrict__ W, int O, int I, int Ipad, int OP, float* __restrict__ dst) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)9 * Ipad * OP;
    if (idx >= total) return;
    int kl = (int)(idx & 15);
    size_t t = idx >> 4;
    int o = (int)(t % OP);
    int k = (int)(t / OP) * 16 + kl;
    int tap = k / Ipad, c = k - tap * Ipad;
    dst[idx] = (o < O && c < I) ? W[((size_t)o * I + c) * 9 + tap] : 0.0f;
}
// data-gradient fragments: dX[m][c] = Σ_{tap',o} dZ[nbr(m,tap')][o] · W[o][c][8-tap']
// dst[((k>>4)·IP + c)·16 + (k&15)], k = tap'·Opad + o
__global__ void k_pack_conv_bwd(const float* __restrict__ W, int O, int I, int Opad, int IP, float* __restrict__ dst) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)9 * Opad * IP;
    if (idx >= total) return;
    int kl = (int)(idx & 15);
    size_t t = idx >> 4;
    int c = (int)(t % IP);
    int k = (int)(t / IP) * 16 + kl;
    int tap = k / Opad, o = k - tap * Opad;
    dst[idx] = (o < O && c < I) ? W[((size_t)o * I + c) * 9 + (8 - tap)] : 0.0f;
}
// policy FC forward: dst[((k>>4)·NP + p)·16 + (k&15)] = W[p][c·nsq + sq], k = sq·F + c (NHWC activations)
__global__ void k_pack_fc_fwd(const float* __restrict__ W, int P, int F, int nsq, int NP, float* __restrict__ dst) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int K = F * nsq;
    size_t total = (size_t)K * NP;
    if (idx >= total) return;
    int kl = (int)(idx & 15);
    size_t t = idx >> 4;
    int p = (int)(t % NP);
    int k = (int)(t / NP) * 16 + kl;
    int sq = k / F, c = k - sq * F;
    dst[idx] = p < P ? W[(size_t)p * K + (size_t)c * nsq + sq] : 0.0f;
}
// policy FC data gradient dS = dLogits · W: reduction over p (padded to Pp), output column k (padded to KP)
// dst[((p>>4)·KP + k)·16 + (p&15)]
__global__ void k_pack_fc_bwd(const float* __restrict__ W, int P, int F, int nsq, int Pp, int KP, float* __restrict__ dst) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int K = F * nsq;
    size_t total = (size_t)Pp * KP;
    if (idx >= total) return;
    int pl = (int)(idx & 15);
    size_t t = idx >> 4;
    int k = (int)(t % KP);
    int p = (int)(t / KP) * 16 + pl;
    int sq = k / F, c = k - sq * F;
    dst[idx] = (p < P && k < K) ? W[(size_t)p * K + (size_t)c * nsq + sq] : 0.0f;
}
// value weights: wv[sq·F + c] = W[c·nsq + sq]
__global__ void k_pack_value(const float* __restrict__ W, int F, int nsq, float* __restrict__ dst) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= F * nsq) return;
    int sq = k / F, c = k - sq * F;
    dst[k] = W[c * nsq + sq];
}
__global__ void k_pad_copy(const float* __restrict__ src, int n, int npad, float* __restrict__ dst) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < npad) dst[i] = i < n ? src[i] : 0.0f;
}

// ------------------------------------------------------------------------------------------------
// Column reductions over the rows of an NHWC activation [M][F] (BatchNorm statistics, BN backward sums,
// bias gradients).  Thread = (row lane, float4 of channels); partial sums leave the block as doubles.
// part[block][2][F].
// ------------------------------------------------------------------------------------------------
enum { RED_SUM = 0, RED_VAR = 1, RED_BNBWD = 2 };

template <int MODE>
__global__ __launch_bounds__(256) void k_col_reduce(const float* __restrict__ a, const float* __restrict__ y, const float* __restrict__ z,
                                                    const float* __restrict__ mean, const float* __restrict__ invstd, int M, int F,
                                                    int rows_per_block, double* __restrict__ part) {
    __shared__ double red[2][256][4];
    const int tid = threadIdx.x;
    const int vpr = F >> 2;
    const int lanes_r = 256 / vpr;
    const int cv = tid % vpr, rl = tid / vpr;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(M, r0 + rows_per_block);
    const f32x4* a4 = (const f32x4*)a;
    const f32x4* y4 = (const f32x4*)y;
    const f32x4* z4 = (const f32x4*)z;
    f32x4 mu = {0.f, 0.f, 0.f, 0.f}, is = {0.f, 0.f, 0.f, 0.f};
    if (MODE != RED_SUM) mu = ((const f32x4*)mean)[cv];
    if (MODE == RED_BNBWD) is = ((const f32x4*)invstd)[cv];
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    if (rl < lanes_r) {
        for (int r = r0 + rl; r < r1; r += lanes_r) {
            const size_t o = (size_t)r * vpr + cv;
            f32x4 x = a4[o];
            if (MODE == RED_SUM) s1 += x;
            if (MODE == RED_VAR) { f32x4 d = x - mu; s1 += d * d; }
            if (MODE == RED_BNBWD) {
                f32x4 yy = y4[o], zz = z4[o];
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    float g = yy[t] > 0.0f ? x[t] : 0.0f;
                    s1[t] += g;
                    s2[t] += g * ((zz[t] - mu[t]) * is[t]);
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 4; t++) { red[0][tid][t] = (double)s1[t]; red[1][tid][t] = (double)s2[t]; }
    __syncthreads();
    if (tid < vpr) {
        for (int w = 0; w < 2; w++) {
            double acc[4] = {0, 0, 0, 0};
            for (int l = 0; l < lanes_r; l++)
#pragma unroll
                for (int t = 0; t < 4; t++) acc[t] += red[w][l * vpr + tid][t];
            double* dst = part + ((size_t)blockIdx.x * 2 + w) * F + tid * 4;
#pragma unroll
            for (int t = 0; t < 4; t++) dst[t] = acc[t];
        }
    }
}

// One 256-thread block per channel: fixed-order (thread-strided, then tree) sum of the block partials.
__device__ inline double part_sum(const double* part, int nblk, int F, int w, int c) {
    __shared__ double red[256];
    double s = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 256) s += part[((size_t)b * 2 + w) * F + c];
    __syncthreads();
    red[threadIdx.x] = s;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) red[threadIdx.x] += red[threadIdx.x + d];
        __syncthreads();
    }
    return red[0];
}
__global__ __launch_bounds__(256) void k_bn_mean_finalize(const double* __restrict__ part, int nblk, int F, int M, float* __restrict__ mean) {
    const int c = blockIdx.x;
    double s = part_sum(part, nblk, F, 0, c);
    if (threadIdx.x == 0) mean[c] = (float)(s / (double)M);
}
// biased variance → invstd; running statistics as torch batch_norm(training=True): unbiased variance, momentum
__global__ __launch_bounds__(256) void k_bn_var_finalize(const double* __restrict__ part, int nblk, int F, int M, float eps, float momentum,
                                                         const float* __restrict__ mean, float* __restrict__ invstd,
                                                         float* __restrict__ running_mean, float* __restrict__ running_var) {
    const int c = blockIdx.x;
    double var = part_sum(part, nblk, F, 0, c) / (double)M;
    if (threadIdx.x != 0) return;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    double unbiased = M > 1 ? var * ((double)M / (double)(M - 1)) : var;
    running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * (double)mean[c]);
    running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unbiased);
}
// the same from Σz and Σz² (the conv kernel's epilogue partials, doubles): var = E[z²] − E[z]² in double
__global__ __launch_bounds__(256) void k_bn_moments_finalize(const double* __restrict__ part, int nblk, int F, int M, float eps, float momentum,
                                                             float* __restrict__ mean, float* __restrict__ invstd,
                                                             float* __restrict__ running_mean, float* __restrict__ running_var) {
    const int c = blockIdx.x;
    const double s1 = part_sum(part, nblk, F, 0, c), s2 = part_sum(part, nblk, F, 1, c);
    if (threadIdx.x != 0) return;
    const double mu = s1 / (double)M;
    double var = s2 / (double)M - mu * mu;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)mu;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    const double unbiased = M > 1 ? var * ((double)M / (double)(M - 1)) : var;
    running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * (double)(float)mu);
    running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unbiased);
}
// Σg, Σg·x̂ → grad_beta += , grad_gamma += ; the per-row means used by the apply kernel
// (the two means stay doubles: rounded to f32 they would shift every row's dz of a channel by the same 6e-8-relative amount,
//  and the weight gradients that sum dz over all M rows would collect M times that — 1.5e-3 at the reference chunk size)
__global__ __launch_bounds__(256) void k_bn_bwd_finalize(const double* __restrict__ part, int nblk, int F, int M, double* __restrict__ mean_g,
                                                         double* __restrict__ mean_gx, float* __restrict__ grad_gamma,
                                                         float* __restrict__ grad_beta) {
    const int c = blockIdx.x;
    double sg = part_sum(part, nblk, F, 0, c), sgx = part_sum(part, nblk, F, 1, c);
    if (threadIdx.x != 0) return;
    mean_g[c] = sg / (double)M;
    mean_gx[c] = sgx / (double)M;
    grad_beta[c] += (float)sg;
    grad_gamma[c] += (float)sgx;
}
__global__ __launch_bounds__(256) void k_colsum_finalize(const double* __restrict__ part, int nblk, int Fp, int valid, float* __restrict__ grad) {
    const int c = blockIdx.x;
    double s = part_sum(part, nblk, Fp, 0, c);
    if (threadIdx.x == 0 && c < valid) grad[c] += (float)s;
}

// column sums of a wide row-major matrix (FC policy bias gradient): thread = column, block row = row split
__global__ __launch_bounds__(256) void k_colsum_wide(const float* __restrict__ a, int rows, int stride, int cols, int rows_per_split,
                                                     double* __restrict__ part) {
    int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    int r0 = blockIdx.y * rows_per_split, r1 = min(rows, r0 + rows_per_split);
    double s = 0.0;
    for (int r = r0; r < r1; r++) s += (double)a[(size_t)r * stride + c];
    part[((size_t)blockIdx.y * 2) * cols + c] = s;
}

// y = relu(γ·(z-μ)·invstd + β (+ skip))   (res_block.rs:13-24 with BN in training mode)
__global__ __launch_bounds__(256) void k_bn_fwd_apply(const float* __restrict__ z, const float* __restrict__ mean,
                                                      const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, const float* __restrict__ skip,
                                                      float* __restrict__ y, size_t total4, int vpr) {
    size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total4) return;
    int cv = (int)(idx % vpr);
    f32x4 x = ((const f32x4*)z)[idx];
    f32x4 mu = ((const f32x4*)mean)[cv], is = ((const f32x4*)invstd)[cv], g = ((const f32x4*)gamma)[cv], b = ((const f32x4*)beta)[cv];
    f32x4 v = (x - mu) * is * g + b;
    if (skip) v += ((const f32x4*)skip)[idx];
#pragma unroll
    for (int t = 0; t < 4; t++) v[t] = fmaxf(v[t], 0.0f);
    ((f32x4*)y)[idx] = v;
}

// g = dy·[y>0];  dz = γ·invstd·(g − mean(g) − x̂·mean(g·x̂));  optionally g is also the gradient of the skip path
__global__ __launch_bounds__(256) void k_bn_bwd_apply(const float* __restrict__ dy, const float* __restrict__ y,
                                                      const float* __restrict__ z, const float* __restrict__ mean,
                                                      const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                      const double* __restrict__ mean_g, const double* __restrict__ mean_gx,
                                                      float* __restrict__ dz, float* __restrict__ gskip, size_t total4, int vpr) {
    size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total4) return;
    int cv = (int)(idx % vpr);
    f32x4 d = ((const f32x4*)dy)[idx], yy = ((const f32x4*)y)[idx], zz = ((const f32x4*)z)[idx];
    f32x4 mu = ((const f32x4*)mean)[cv], is = ((const f32x4*)invstd)[cv], ga = ((const f32x4*)gamma)[cv];
    f32x4 g, o;
#pragma unroll
    for (int t = 0; t < 4; t++) {
        g[t] = yy[t] > 0.0f ? d[t] : 0.0f;
        float xh = (zz[t] - mu[t]) * is[t];
        // the centring in double (see k_bn_bwd_finalize), the result back in f32
        const double centred = (double)g[t] - mean_g[4 * cv + t] - (double)xh * mean_gx[4 * cv + t];
        o[t] = (float)((double)(ga[t] * is[t]) * centred);
    }
    ((f32x4*)dz)[idx] = o;
    if (gskip) ((f32x4*)gskip)[idx] = g;
}

// k_bn_bwd_apply with the column sums of dz — the gradient of the bias of the convolution in front of the BatchNorm — taken while
// dz is in registers, in k_col_reduce<RED_SUM>'s thread mapping and order (so the sums are the bits a separate pass over dz gave):
// one pass over [M][F] and two launches less per layer.
__global__ __launch_bounds__(256) void k_bn_bwd_apply_sum(const float* __restrict__ dy, const float* __restrict__ y,
                                                          const float* __restrict__ z, const float* __restrict__ mean,
                                                          const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                          const double* __restrict__ mean_g, const double* __restrict__ mean_gx,
                                                          float* __restrict__ dz, float* __restrict__ gskip, int M, int F, int rows_per_block,
                                                          double* __restrict__ part) {
    __shared__ double red[256][4];
    const int tid = threadIdx.x;
    const int vpr = F >> 2;
    const int lanes_r = 256 / vpr;
    const int cv = tid % vpr, rl = tid / vpr;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(M, r0 + rows_per_block);
    const f32x4 mu = ((const f32x4*)mean)[cv], is = ((const f32x4*)invstd)[cv], ga = ((const f32x4*)gamma)[cv];
    double mg[4], mgx[4];
#pragma unroll
    for (int t = 0; t < 4; t++) { mg[t] = mean_g[4 * cv + t]; mgx[t] = mean_gx[4 * cv + t]; }
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f};
    if (rl < lanes_r) {
        for (int r = r0 + rl; r < r1; r += lanes_r) {
            const size_t idx = (size_t)r * vpr + cv;
            const f32x4 d = ((const f32x4*)dy)[idx], yy = ((const f32x4*)y)[idx], zz = ((const f32x4*)z)[idx];
            f32x4 g, o;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                g[t] = yy[t] > 0.0f ? d[t] : 0.0f;
                const float xh = (zz[t] - mu[t]) * is[t];
                const double centred = (double)g[t] - mg[t] - (double)xh * mgx[t];
                o[t] = (float)((double)(ga[t] * is[t]) * centred);
            }
            ((f32x4*)dz)[idx] = o;
            if (gskip) ((f32x4*)gskip)[idx] = g;
            s1 += o;
        }
    }
#pragma unroll
    for (int t = 0; t < 4; t++) red[tid][t] = (double)s1[t];
    __syncthreads();
    if (tid < vpr) {
        double acc[4] = {0, 0, 0, 0};
        for (int l = 0; l < lanes_r; l++)
#pragma unroll
            for (int t = 0; t < 4; t++) acc[t] += red[l * vpr + tid][t];
        double* dst = part + ((size_t)blockIdx.x * 2) * F + tid * 4;
#pragma unroll
        for (int t = 0; t < 4; t++) dst[t] = acc[t];
    }
}

// ------------------------------------------------------------------------------------------------
// Heads: losses of network.rs:81-84 and their gradients
// ------------------------------------------------------------------------------------------------
__device__ inline float wsum(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}
__device__ inline float wmax(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d));
    return v;
}
__device__ inline float block_sum(float v, float* red) {
    v = wsum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ inline float block_max(float v, float* red) {
    v = wmax(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// log_softmax over all P outputs (net5.rs:113-118), loss_p row = −Σ π·logp, and (when pi is given)
// dLogits = (softmax·Σπ − π)·inv_b  — the backward of −(π·logp).sum()/B through log_softmax.
// One 256-thread block per position; logits / dlogits share the layout of k_softmax (conv head: [sq][ch]).
__global__ __launch_bounds__(256) void k_policy_loss(const float* __restrict__ logits, int row_stride, int conv_head, int nsq,
                                                     int ch_stride, int P, const float* __restrict__ pi, float inv_b,
                                                     float* __restrict__ dlogits, float* __restrict__ logp_out,
                                                     float* __restrict__ loss_rows) {
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* x = logits + (size_t)b * row_stride;
    auto slot = [&](int p) -> int {
        if (!conv_head) return p;
        int ch = p / nsq, sq = p - ch * nsq;
        return sq * ch_stride + ch;
    };
    float mx = -INFINITY;
    for (int p = tid; p < P; p += 256) mx = fmaxf(mx, x[slot(p)]);
    mx = block_max(mx, red);
    float s = 0.0f;
    for (int p = tid; p < P; p += 256) s += expf(x[slot(p)] - mx);
    s = block_sum(s, red);
    const float lse = mx + logf(s);
    float lp = 0.0f, sp = 0.0f;
    const float* t = pi ? pi + (size_t)b * P : nullptr;
    for (int p = tid; p < P; p += 256) {
        float l = x[slot(p)] - lse;
        if (logp_out) logp_out[(size_t)b * P + p] = l;
        if (t) { lp -= t[p] * l; sp += t[p]; }
    }
    if (!t) return;
    lp = block_sum(lp, red);
    sp = block_sum(sp, red);
    if (tid == 0) loss_rows[b] = lp;
    float* d = dlogits + (size_t)b * row_stride;
    for (int p = tid; p < P; p += 256) {
        int o = slot(p);
        d[o] = (expf(x[o] - lse) * sp - t[p]) * inv_b;
    }
}

// value head in training: v = tanh(w·s + b);  loss_z row = (z − v)²;  dpre = −2(z − v)·inv_b·(1 − v²)
__global__ __launch_bounds__(256) void k_value_train(const float* __restrict__ act, const float* __restrict__ wv,
                                                     const float* __restrict__ bv, int B, int len, const float* __restrict__ zt,
                                                     float inv_b, float* __restrict__ eval, float* __restrict__ dpre,
                                                     float* __restrict__ loss_rows) {
    int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    int lane = threadIdx.x & 63;
    const float4* a = (const float4*)(act + (size_t)b * len);
    const float4* w = (const float4*)wv;
    float s = 0.0f;
    for (int k = lane; k < (len >> 2); k += 64) {
        float4 x = a[k], y = w[k];
        s = fmaf(x.x, y.x, s);
        s = fmaf(x.y, y.y, s);
        s = fmaf(x.z, y.z, s);
        s = fmaf(x.w, y.w, s);
    }
    s = wsum(s);
    if (lane == 0) {
        float v = tanhf(s + bv[0]);
        eval[b] = v;
        if (zt) {
            float d = zt[b] - v;
            loss_rows[b] = d * d;
            dpre[b] = -2.0f * d * inv_b * (1.0f - v * v);
        }
    }
}
// dS[b][k] += dpre[b]·wv[k]
__global__ __launch_bounds__(256) void k_value_bwd_ds(float* __restrict__ ds, const float* __restrict__ dpre,
                                                      const float* __restrict__ wv, size_t total4, int len4) {
    size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total4) return;
    size_t b = idx / len4;
    int k = (int)(idx - b * len4);
    float d = dpre[b];
    f32x4 w = ((const f32x4*)wv)[k];
    ((f32x4*)ds)[idx] += w * d;
}
// value weight gradient: part[split][k] = Σ_{b in split} dpre[b]·s[b][k]; part[split][len] = Σ dpre[b] (bias)
__global__ __launch_bounds__(256) void k_value_wgrad(const float* __restrict__ act, const float* __restrict__ dpre, int B, int len,
                                                     int rows_per_split, double* __restrict__ part) {
    int k = blockIdx.x * 256 + threadIdx.x;
    int b0 = blockIdx.y * rows_per_split, b1 = min(B, b0 + rows_per_split);
    if (k > len) return;
    double s = 0.0;
    if (k == len) { for (int b = b0; b < b1; b++) s += (double)dpre[b]; }
    else for (int b = b0; b < b1; b++) s += (double)(dpre[b] * act[(size_t)b * len + k]);
    part[(size_t)blockIdx.y * (len + 1) + k] = s;
}
// grad_w[c·nsq + sq] += Σ_split part[split][sq·F + c];  grad_b += Σ part[split][len]
__global__ void k_value_wgrad_finalize(const double* __restrict__ part, int splits, int F, int nsq, float* __restrict__ grad_w,
                                       float* __restrict__ grad_b) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int len = F * nsq;
    if (k > len) return;
    double s = 0.0;
    for (int i = 0; i < splits; i++) s += part[(size_t)i * (len + 1) + k];
    if (k == len) grad_b[0] += (float)s;
    else { int sq = k / F, c = k - sq * F; grad_w[c * nsq + sq] += (float)s; }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient: dW[co][ci][tap] = Σ_rows G[row][co] · X[nbr(row, tap)][ci]  — a TN GEMM whose reduction
// runs over the rows.  v_mfma_f32_16x16x4_f32 consumes 4 rows per instruction (k = lane>>4); a lane loads
// 16 bytes = 4 consecutive channels of its row from each side and uses component a (G side) × component u
// (X side) for sub-tile (a,u), whose 16×16 outputs are the channels {4i+a}×{4j+u}: 2 LDS reads feed 16 MFMAs
// and nothing is transposed.  A workgroup = 4 waves (one per SIMD) owning a 64(co)×64(ci) block for its rows:
//   CONV: wave w owns G sub-tile a = w for all 9 taps × 4 X sub-tiles (36 accumulator tiles; the X tile is
//         shared, taps are LDS row offsets or the zero row);
//   FC:   wave w owns ci block 4·tile+w (X tile = 256 channels), all 16 sub-tiles.
// Partial blocks (one per row split) go to a workspace; k_wgrad_reduce_* sums them in fixed order into the
// gradient buffer in tch layout — deterministic, and the place where gradients accumulate over chunks.
// part[((split·ntiles + tile)·nsub + sub)·4096 + co_l·64 + ci_l]
// ------------------------------------------------------------------------------------------------
// PF: float4 slots per thread of the register prefetch (CONV): a chunk holds at most 16·PF rows
template <bool CONV, int PF = 4>
__global__ __launch_bounds__(256) void k_wgrad(const float* __restrict__ X, int xs, int xvalid, const float* __restrict__ G, int gs,
                                               int gvalid, int R, int n, int nsq, int rows_chunk, int chunks_per_split, int ncob,
                                               float* __restrict__ part) {
    constexpr int T = CONV ? 9 : 1;      // taps per wave
    constexpr int NA = CONV ? 1 : 4;     // G sub-tiles per wave
    constexpr int XC = CONV ? 64 : 256;
    constexpr int XV = XC / 4;           // float4 per staged X row
    constexpr int XLS4 = XV + 1;         // LDS row pitch (f32x4), +1 shifts banks between rows (pitch XV measured in round 5: no faster here)
    constexpr int GLS4 = 17;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    f32x4* Xt = (f32x4*)lds;                                  // (rows_chunk + 1) rows; the last one stays zero
    const int rows_pad_max = (rows_chunk + 3) & ~3;
    f32x4* Gt = Xt + (size_t)(rows_chunk + 1) * XLS4;         // rows_pad_max rows
    int* tmask = (int*)(Gt + (size_t)rows_pad_max * GLS4);   // rows_pad_max entries

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int j = lane & 15, q = lane >> 4;
    const int tile = blockIdx.y, cib = tile / ncob, cob = tile - cib * ncob;
    const int xc0 = cib * XC, gc0 = cob * 64;
    const int split = blockIdx.x;

    f32x4 acc[T][NA][4];
#pragma unroll
    for (int t = 0; t < T; t++)
#pragma unroll
        for (int a = 0; a < NA; a++)
#pragma unroll
            for (int u = 0; u < 4; u++) acc[t][a][u] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int v = tid; v < XLS4; v += 256) Xt[(size_t)rows_chunk * XLS4 + v] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int row_begin = split * chunks_per_split * rows_chunk;
    // CONV: the next chunk's X / G slots travel through registers (≤ 4 + 4 float4 per thread, unconditional clamped loads)
    // while the current chunk's MFMAs run; they are written to LDS behind the barrier that ends the chunk
    f32x4 px[PF], pg[PF];
    auto prefetch = [&](int r0) {
#pragma unroll
        for (int u = 0; u < PF; u++) {
            const int idx = u * 256 + tid;
            const int r = idx >> 4, v = idx & 15;
            const int rr = min(r0 + r, R - 1);
            const int cx = min(xc0 + 4 * v, xs - 4), cg = min(gc0 + 4 * v, gs - 4);
            px[u] = *(const f32x4*)(X + (size_t)rr * xs + cx);
            pg[u] = *(const f32x4*)(G + (size_t)rr * gs + cg);
        }
    };
    if (CONV && row_begin < R) prefetch(row_begin);
    for (int ch = 0; ch < chunks_per_split; ch++) {
        const int r0 = row_begin + ch * rows_chunk;
        if (r0 >= R) break;
        const int rows = min(rows_chunk, R - r0);
        const int rows_pad = (rows + 3) & ~3;
        __syncthreads();  // the previous chunk has been consumed
        if (CONV) {
            const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < PF; u++) {
                const int idx = u * 256 + tid;
                const int r = idx >> 4, v = idx & 15;
                if (idx < rows * 16) Xt[r * XLS4 + v] = (xc0 + 4 * v < xvalid) ? px[u] : zero;
                if (idx < rows_pad * 16) Gt[r * GLS4 + v] = (r < rows && gc0 + 4 * v < gvalid) ? pg[u] : zero;
            }
        } else {
            for (int idx = tid; idx < rows * XV; idx += 256) {
                int r = idx / XV, v = idx - r * XV;
                int c = xc0 + 4 * v;
                f32x4 val = f32x4{0.f, 0.f, 0.f, 0.f};
                if (c < xvalid) val = *(const f32x4*)(X + (size_t)(r0 + r) * xs + c);
                Xt[r * XLS4 + v] = val;
            }
            for (int idx = tid; idx < rows_pad * 16; idx += 256) {
                int r = idx >> 4, v = idx & 15;
                int c = gc0 + 4 * v;
                f32x4 val = f32x4{0.f, 0.f, 0.f, 0.f};
                if (r < rows && c < gvalid) val = *(const f32x4*)(G + (size_t)(r0 + r) * gs + c);
                Gt[r * GLS4 + v] = val;
            }
        }
        if (CONV) {
            for (int r = tid; r < rows_pad; r += 256) {
                int m = 0;
                if (r < rows) {
                    int sq = r % nsq;  // r0 is a multiple of nsq
                    int y = sq / n, x = sq - y * n;
#pragma unroll
                    for (int t = 0; t < 9; t++) {
                        int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
                        if (yy >= 0 && yy < n && xx >= 0 && xx < n) m |= 1 << t;
                    }
                }
                tmask[r] = m;
            }
        }
        __syncthreads();
        if (CONV && ch + 1 < chunks_per_split && r0 + rows_chunk < R) prefetch(r0 + rows_chunk);
        if (CONV) {
            for (int rr = 0; rr < rows_pad; rr += 4) {
                const int r = rr + q;
                const float g = ((const float*)Gt)[(r * GLS4 + j) * 4 + wave];  // this wave's G sub-tile a = wave: channels {4i + wave}
                const int m = tmask[r];
                f32x4 x[T];
#pragma unroll
                for (int t = 0; t < T; t++) {
                    const int sh = (t / 3 - 1) * n + (t % 3 - 1);
                    const int row = ((m >> t) & 1) ? r + sh : rows_chunk;
                    x[t] = Xt[row * XLS4 + j];
                }
#pragma unroll
                for (int t = 0; t < T; t++)
#pragma unroll
                    for (int u = 0; u < 4; u++) acc[t][0][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(g, x[t][u], acc[t][0][u], 0, 0, 0);
            }
        } else {
            for (int rr = 0; rr < rows_pad; rr += 4) {
                const int r = rr + q;
                const f32x4 g = Gt[r * GLS4 + j];
                const f32x4 x = r < rows ? Xt[r * XLS4 + wave * 16 + j] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int a = 0; a < NA; a++)
#pragma unroll
                    for (int u = 0; u < 4; u++) acc[0][a][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(g[a], x[u], acc[0][a][u], 0, 0, 0);
            }
        }
    }
    // lane (j, q), register v of sub-tile (a,u): co_l = 16q + 4v + a, ci_l = 4j + u
    const int ntiles = gridDim.y;
    constexpr int NSUB = CONV ? 9 : 4;
#pragma unroll
    for (int t = 0; t < T; t++) {
        const int sub = CONV ? t : wave;
        float* dst = part + (((size_t)split * ntiles + tile) * NSUB + sub) * 4096;
#pragma unroll
        for (int a = 0; a < NA; a++)
#pragma unroll
            for (int v = 0; v < 4; v++) {
                const int co_l = 16 * q + 4 * v + (CONV ? wave : a);
                f32x4 o = f32x4{acc[t][a][0][v], acc[t][a][1][v], acc[t][a][2][v], acc[t][a][3][v]};
                *(f32x4*)(dst + co_l * 64 + 4 * j) = o;
            }
    }
}

// ------------------------------------------------------------------------------------------------
// The conv weight gradient with the X tile as a HALO image (round 3; k_wgrad<true> stays for other board sizes and as the
// A/B reference, TG_NO_HALO_WGRAD).  Same decomposition — 64(co)×64(ci) block per workgroup, wave w = G sub-tile a = w × 9
// taps × 4 X sub-tiles, chunks of PWC positions, split-K partials in the same layout — but the staged input rows sit in halo
// cells (one zero cell behind every board row, a zero row behind every position, as in k_tower_halo): a tap is a constant cell
// offset whether or not it stays on the board.  So the per-step work of a wave is 1 + 9 LDS reads at addresses that are a
// per-kernel register (the cell of its row, the same in every chunk) plus an immediate, and 36 MFMAs: no tap masks, no selects,
// no address arithmetic, no LDS read that another LDS read waits for (k_wgrad spent 93 vector instructions per step on those and
// kept the MFMA pipe 65 % busy).  Row r of a chunk ↔ cell LEAD + (r / n²)·PS + (y·RS + x).
// ------------------------------------------------------------------------------------------------
template <int NB, int PWC>
__global__ __launch_bounds__(256, 2) void k_wgrad_halo(const float* __restrict__ X, int xs, int xvalid, const float* __restrict__ G, int gs,
                                                    int gvalid, int R, int chunks_per_split, int ncob, float* __restrict__ part) {
    constexpr int n = NB, nsq = NB * NB, RS = NB + 1, LEAD = NB + 2, PS = (NB + 1) * RS;
    constexpr int ROWS = PWC * nsq, ROWS_PAD = (ROWS + 3) & ~3, NSTEPS = ROWS_PAD / 4;
    constexpr int CELLS = LEAD + PWC * PS + 1;
    // X cells at a pitch of exactly 16 slots: a wave's ds_read_b128 takes slot j of four rows' cells, and the hardware serves lanes
    // {0-3, 12-15} of one row together with {4-11} of the next — at pitch 16 those are 16 different bank quads whatever the two cells
    // are; at 17 (rounds 3 – 4) neighbouring cells met on one quad: SQ_LDS_BANK_CONFLICT was 49 % of the kernel's LDS cycles
    // (profiles/r05_r_pmc_sq_train_kernels.txt).  The LDS is only 23 % busy, so it buys 0.8 % (240.3 → 238.3 µs, r05_s).
    constexpr int XLS4 = 16, GLS4 = 17;
    constexpr int PF = (ROWS * 16 + 255) / 256;  // float4 slots per thread of the register prefetch
    static_assert(ROWS <= 64, "the register prefetch carries at most 64 rows");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    f32x4* Xh = (f32x4*)lds;                       // CELLS cells of XLS4 slots
    f32x4* Gt = Xh + (size_t)CELLS * XLS4;         // ROWS_PAD rows of GLS4 slots
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int j = lane & 15, q = lane >> 4;
    const int tile = blockIdx.y, cib = tile / ncob, cob = tile - cib * ncob;
    const int xc0 = cib * 64, gc0 = cob * 64;
    const int split = blockIdx.x;
    auto cell_of = [](int r) { const int p = r / nsq, sq = r - p * nsq, y = sq / n, x = sq - y * n; return LEAD + p * PS + y * RS + x; };

    f32x4 acc[9][4];
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
        for (int u = 0; u < 4; u++) acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
    // every cell zero once; the squares' cells are overwritten chunk by chunk, the halo never again
    for (int idx = tid; idx < CELLS * XLS4; idx += 256) Xh[idx] = f32x4{0.f, 0.f, 0.f, 0.f};
    // this lane's rows 4s + q of a chunk: X address (cell, slot j) and G address (row, slot j, component `wave`) — per kernel
    int xa[NSTEPS];
#pragma unroll
    for (int s = 0; s < NSTEPS; s++) {
        const int r = 4 * s + q;
        xa[s] = (r < ROWS ? cell_of(r) : LEAD) * XLS4 + j;  // (padding rows read a real cell against a zero gradient)
    }
    const int ga = (q * GLS4 + j) * 4 + wave;

    const int row_begin = split * chunks_per_split * ROWS;
    f32x4 px[PF], pg[PF];
    auto prefetch = [&](int r0) {
#pragma unroll
        for (int u = 0; u < PF; u++) {
            const int idx = u * 256 + tid;
            const int r = idx >> 4, v = idx & 15;
            const int rr = min(r0 + r, R - 1);
            const int cx = min(xc0 + 4 * v, xs - 4), cg = min(gc0 + 4 * v, gs - 4);
            px[u] = *(const f32x4*)(X + (size_t)rr * xs + cx);
            pg[u] = *(const f32x4*)(G + (size_t)rr * gs + cg);
        }
    };
    if (row_begin < R) prefetch(row_begin);
    for (int ch = 0; ch < chunks_per_split; ch++) {
        const int r0 = row_begin + ch * ROWS;
        if (r0 >= R) break;
        const int rows = min(ROWS, R - r0);
        __syncthreads();  // the previous chunk has been consumed (and, the first time, the zero fill is complete)
        {
            const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < PF; u++) {
                const int idx = u * 256 + tid;
                const int r = idx >> 4, v = idx & 15;
                if (idx < ROWS * 16) Xh[cell_of(r) * XLS4 + v] = (r < rows && xc0 + 4 * v < xvalid) ? px[u] : zero;
                if (idx < ROWS_PAD * 16) Gt[r * GLS4 + v] = (r < rows && gc0 + 4 * v < gvalid) ? pg[u] : zero;
            }
        }
        __syncthreads();
        if (ch + 1 < chunks_per_split && r0 + ROWS < R) prefetch(r0 + ROWS);
#pragma unroll
        for (int s = 0; s < NSTEPS; s++) {
            const float g = ((const float*)Gt)[ga + s * 4 * GLS4 * 4];
            f32x4 x[9];
#pragma unroll
            for (int t = 0; t < 9; t++) x[t] = Xh[xa[s] + ((t / 3 - 1) * RS + (t % 3 - 1)) * XLS4];
#pragma unroll
            for (int t = 0; t < 9; t++)
#pragma unroll
                for (int u = 0; u < 4; u++) acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(g, x[t][u], acc[t][u], 0, 0, 0);
            // one step's operands at a time: left alone the scheduler hoists the reads of many steps ahead of the MFMAs (246
            // registers, one wave per SIMD); the second wave of the SIMD covers this wave's LDS latency instead
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // lane (j, q), register v of sub-tile (a = wave, u): co_l = 16q + 4v + wave, ci_l = 4j + u   (k_wgrad's partial layout)
    const int ntiles = gridDim.y;
#pragma unroll
    for (int t = 0; t < 9; t++) {
        float* dst = part + (((size_t)split * ntiles + tile) * 9 + t) * 4096;
#pragma unroll
        for (int v = 0; v < 4; v++) {
            const int co_l = 16 * q + 4 * v + wave;
            f32x4 o = f32x4{acc[t][0][v], acc[t][1][v], acc[t][2][v], acc[t][3][v]};
            *(f32x4*)(dst + co_l * 64 + 4 * j) = o;
        }
    }
}

// conv: grad[(co·I + ci)·9 + tap] += Σ_split part[…]
__global__ __launch_bounds__(256) void k_wgrad_reduce_conv(const float* __restrict__ part, int splits, int ncib, int ncob, int O,
                                                           int I, float* __restrict__ grad) {
    // a thread sums four consecutive input channels (one 16-byte load per split, four splits in flight): the partials are a
    // 70 MB stream, and one dword per thread and split left the loop waiting on latency (36 µs per layer); the splits are
    // still added one after the other, in split order, per element
    size_t idx4 = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int ntiles = ncib * ncob;
    const size_t total = (size_t)ntiles * 9 * 4096, total4 = total >> 2;
    if (idx4 >= total4) return;
    const size_t idx = idx4 << 2;
    int ci_l = (int)(idx & 63), co_l = (int)((idx >> 6) & 63);
    size_t t = idx >> 12;
    int tap = (int)(t % 9);
    int tile = (int)(t / 9);
    int cib = tile / ncob, cob = tile - cib * ncob;
    int co = cob * 64 + co_l, ci = cib * 64 + ci_l;
    if (co >= O || ci >= I) return;
    const f32x4* p4 = (const f32x4*)part + idx4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    int sp = 0;
    for (; sp + 4 <= splits; sp += 4) {
        const f32x4 a = p4[(size_t)sp * total4], b = p4[(size_t)(sp + 1) * total4], c = p4[(size_t)(sp + 2) * total4], d = p4[(size_t)(sp + 3) * total4];
        s += a; s += b; s += c; s += d;
    }
    for (; sp < splits; sp++) s += p4[(size_t)sp * total4];
#pragma unroll
    for (int u = 0; u < 4; u++)
        if (ci + u < I) grad[((size_t)co * I + ci + u) * 9 + tap] += s[u];
}
// policy FC: grad[p·K + c·nsq + sq] += Σ_split part[…],  k = (4·cit + w)·64 + ci_l = sq·F + c
__global__ __launch_bounds__(256) void k_wgrad_reduce_fc(const float* __restrict__ part, int splits, int ncit, int ncob, int P, int F,
                                                         int nsq, float* __restrict__ grad) {
    size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int ntiles = ncit * ncob;
    size_t total = (size_t)ntiles * 4 * 4096;
    if (idx >= total) return;
    int ci_l = (int)(idx & 63), co_l = (int)((idx >> 6) & 63);
    size_t t = idx >> 12;
    int w = (int)(t & 3);
    int tile = (int)(t >> 2);
    int cit = tile / ncob, cob = tile - cit * ncob;
    int p = cob * 64 + co_l, k = (cit * 4 + w) * 64 + ci_l;
    const int K = F * nsq;
    if (p >= P || k >= K) return;
    float s = 0.0f;
    for (int sp = 0; sp < splits; sp++) s += part[(size_t)sp * total + idx];
    int sq = k / F, c = k - sq * F;
    grad[(size_t)p * K + (size_t)c * nsq + sq] += s;
}

// ------------------------------------------------------------------------------------------------
// Adam as tch's nn::Adam{wd} → torch::optim::Adam (L2 weight decay folded into the gradient, no amsgrad),
// network.rs:40-45.  gscale = 1/world_size after the gradient all-reduce.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                              float* __restrict__ v, size_t n, float lr, float b1, float b2, float eps, float wd,
                                              float bc1, float bc2_sqrt, float gscale) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float gi = g[i] * gscale + wd * p[i];
    float mi = b1 * m[i] + (1.0f - b1) * gi;
    float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] -= (lr / bc1) * (mi / denom);
}

__global__ void k_sum_rows(const float* __restrict__ rows, int n, double* __restrict__ out) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += (double)rows[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) red[threadIdx.x] += red[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0];
}

// ---- launchers --------------------------------------------------------------------------------
static inline unsigned blocks_for(size_t total, int bs = 256) { return (unsigned)((total + bs - 1) / bs); }

hipError_t launch_pack_conv_fwd(hipStream_t st, const float* W, int O, int I, int Ipad, int OP, float* dst) {
    hipLaunchKernelGGL(k_pack_conv_fwd, dim3(blocks_for((size_t)9 * Ipad * OP)), dim3(256), 0, st, W, O, I, Ipad, OP, dst);
    return hipGetLastError();
}
hipError_t launch_pack_conv_bwd(hipStream_t st, const float* W, int O, int I, int Opad, int IP, float* dst) {
    hipLaunchKernelGGL(k_pack_conv_bwd, dim3(blocks_for((size_t)9 * Opad * IP)), dim3(256), 0, st, W, O, I, Opad, IP, dst);
    return hipGetLastError();
}
hipError_t launch_pack_fc_fwd(hipStream_t st, const float* W, int P, int F, int nsq, int NP, float* dst) {
    hipLaunchKernelGGL(k_pack_fc_fwd, dim3(blocks_for((size_t)F * nsq * NP)), dim3(256), 0, st, W, P, F, nsq, NP, dst);
    return hipGetLastError();
}
hipError_t launch_pack_fc_bwd(hipStream_t st, const float* W, int P, int F, int nsq, int Pp, int KP, float* dst) {
    hipLaunchKernelGGL(k_pack_fc_bwd, dim3(blocks_for((size_t)Pp * KP)), dim3(256), 0, st, W, P, F, nsq, Pp, KP, dst);
    return hipGetLastError();
}
hipError_t launch_pack_value(hipStream_t st, const float* W, int F, int nsq, float* dst) {
    hipLaunchKernelGGL(k_pack_value, dim3(blocks_for((size_t)F * nsq)), dim3(256), 0, st, W, F, nsq, dst);
    return hipGetLastError();
}
hipError_t launch_pad_copy(hipStream_t st, const float* src, int n, int npad, float* dst) {
    hipLaunchKernelGGL(k_pad_copy, dim3(blocks_for((size_t)npad)), dim3(256), 0, st, src, n, npad, dst);
    return hipGetLastError();
}

int col_reduce_blocks(int M, int F, int* rows_per_block) {
    int lanes_r = 256 / (F >> 2);
    int rpb = lanes_r * 16;
    int nblk = (M + rpb - 1) / rpb;
    if (nblk > 2048) { rpb = ((M + 2047) / 2048 + lanes_r - 1) / lanes_r * lanes_r; nblk = (M + rpb - 1) / rpb; }
    *rows_per_block = rpb;
    return nblk;
}

hipError_t launch_bn_stats(hipStream_t st, const float* z, int M, int F, float eps, float momentum, double* part, float* mean,
                           float* invstd, float* running_mean, float* running_var) {
    int rpb, nblk = col_reduce_blocks(M, F, &rpb);
    hipLaunchKernelGGL((k_col_reduce<RED_SUM>), dim3(nblk), dim3(256), 0, st, z, nullptr, nullptr, nullptr, nullptr, M, F, rpb, part);
    hipLaunchKernelGGL(k_bn_mean_finalize, dim3(F), dim3(256), 0, st, part, nblk, F, M, mean);
    hipLaunchKernelGGL((k_col_reduce<RED_VAR>), dim3(nblk), dim3(256), 0, st, z, nullptr, nullptr, mean, nullptr, M, F, rpb, part);
    hipLaunchKernelGGL(k_bn_var_finalize, dim3(F), dim3(256), 0, st, part, nblk, F, M, eps, momentum, mean, invstd,
                       running_mean, running_var);
    return hipGetLastError();
}
hipError_t launch_bn_stats_from_partials(hipStream_t st, const double* part, int nblk, int M, int F, float eps, float momentum,
                                         float* mean, float* invstd, float* running_mean, float* running_var) {
    hipLaunchKernelGGL(k_bn_moments_finalize, dim3(F), dim3(256), 0, st, part, nblk, F, M, eps, momentum, mean, invstd, running_mean,
                       running_var);
    return hipGetLastError();
}
hipError_t launch_bn_fwd_apply(hipStream_t st, const float* z, const float* mean, const float* invstd, const float* gamma,
                               const float* beta, const float* skip, float* y, int M, int F) {
    size_t total4 = (size_t)M * F / 4;
    hipLaunchKernelGGL(k_bn_fwd_apply, dim3(blocks_for(total4)), dim3(256), 0, st, z, mean, invstd, gamma, beta, skip, y, total4, F / 4);
    return hipGetLastError();
}
// grad[c] += Σ over `rows` partial rows part[(row·2)·F + c]
hipError_t launch_colsum_finalize(hipStream_t st, const double* part, int rows, int F, int valid, float* grad) {
    hipLaunchKernelGGL(k_colsum_finalize, dim3(valid), dim3(256), 0, st, part, rows, F, valid, grad);
    return hipGetLastError();
}
hipError_t launch_bn_bwd(hipStream_t st, const float* dy, const float* y, const float* z, const float* mean, const float* invstd,
                         const float* gamma, int M, int F, double* part, double* mean_g, double* mean_gx, float* grad_gamma,
                         float* grad_beta, float* dz, float* gskip, float* grad_conv_bias, int sums_in_part, double* colsum_part,
                         int* colsum_rows) {
    int rpb, nblk = col_reduce_blocks(M, F, &rpb);
    // sums_in_part > 0: Σg and Σg·x̂ already sit in `part`, that many partial rows — left there by the epilogue of the convolution
    // that produced dy (launch_conv3x3 with ConvBnBwdIn); else a pass over dy, y and z takes them
    if (sums_in_part <= 0) hipLaunchKernelGGL((k_col_reduce<RED_BNBWD>), dim3(nblk), dim3(256), 0, st, dy, y, z, mean, invstd, M, F, rpb, part);
    hipLaunchKernelGGL(k_bn_bwd_finalize, dim3(F), dim3(256), 0, st, part, sums_in_part > 0 ? sums_in_part : nblk, F, M, mean_g, mean_gx, grad_gamma, grad_beta);
    if (grad_conv_bias) {  // dz and its column sums in one pass (the bias gradient of the convolution in front)
        // colsum_part: the partial rows go there and the caller finalises them (launch_colsum_finalize) where it suits it — the training
        // step does it on the weight gradients' stream, off the chain of kernels the next convolution waits for
        hipLaunchKernelGGL(k_bn_bwd_apply_sum, dim3(nblk), dim3(256), 0, st, dy, y, z, mean, invstd, gamma, mean_g, mean_gx, dz, gskip, M, F,
                           rpb, colsum_part ? colsum_part : part);
        if (colsum_part) *colsum_rows = nblk;
        else hipLaunchKernelGGL(k_colsum_finalize, dim3(F), dim3(256), 0, st, part, nblk, F, F, grad_conv_bias);
        return hipGetLastError();
    }
    size_t total4 = (size_t)M * F / 4;
    hipLaunchKernelGGL(k_bn_bwd_apply, dim3(blocks_for(total4)), dim3(256), 0, st, dy, y, z, mean, invstd, gamma, mean_g, mean_gx, dz,
                       gskip, total4, F / 4);
    return hipGetLastError();
}
hipError_t launch_colsum_acc(hipStream_t st, const float* a, int M, int Fp, int valid, double* part, float* grad) {
    if ((Fp >> 2) > 256 || 256 % (Fp >> 2) != 0) {  // wide rows (FC logits): one thread per column
        const int splits = 32;
        int rps = (M + splits - 1) / splits;
        hipLaunchKernelGGL(k_colsum_wide, dim3((Fp + 255) / 256, splits), dim3(256), 0, st, a, M, Fp, Fp, rps, part);
        hipLaunchKernelGGL(k_colsum_finalize, dim3(valid), dim3(256), 0, st, part, splits, Fp, valid, grad);
        return hipGetLastError();
    }
    int rpb, nblk = col_reduce_blocks(M, Fp, &rpb);
    hipLaunchKernelGGL((k_col_reduce<RED_SUM>), dim3(nblk), dim3(256), 0, st, a, nullptr, nullptr, nullptr, nullptr, M, Fp, rpb, part);
    hipLaunchKernelGGL(k_colsum_finalize, dim3(valid), dim3(256), 0, st, part, nblk, Fp, valid, grad);
    return hipGetLastError();
}
hipError_t launch_policy_loss(hipStream_t st, const float* logits, int row_stride, bool conv_head, int nsq, int ch_stride, int P, int B,
                              const float* pi, float inv_b, float* dlogits, float* logp_out, float* loss_rows) {
    hipLaunchKernelGGL(k_policy_loss, dim3(B), dim3(256), 0, st, logits, row_stride, conv_head ? 1 : 0, nsq, ch_stride, P, pi, inv_b,
                       dlogits, logp_out, loss_rows);
    return hipGetLastError();
}
hipError_t launch_value_train(hipStream_t st, const float* act, const float* wv, const float* bv, int B, int len, const float* zt,
                              float inv_b, float* eval, float* dpre, float* loss_rows) {
    hipLaunchKernelGGL(k_value_train, dim3((B + 3) / 4), dim3(256), 0, st, act, wv, bv, B, len, zt, inv_b, eval, dpre, loss_rows);
    return hipGetLastError();
}
hipError_t launch_value_bwd(hipStream_t st, const float* act, const float* dpre, const float* wv, int B, int F, int nsq, float* ds,
                            double* part, float* grad_w, float* grad_b, hipStream_t st_grad) {
    // ds += dpre ⊗ wv on `st`; the head's own gradients (Σ_b dpre·act, Σ_b dpre) on `st_grad` (null: st) — the training step puts them
    // on the weight gradients' stream with a workspace of that stream, off the chain the first data-gradient convolution waits for
    const int len = F * nsq;
    size_t total4 = (size_t)B * len / 4;
    if (!st_grad) st_grad = st;
    hipLaunchKernelGGL(k_value_bwd_ds, dim3(blocks_for(total4)), dim3(256), 0, st, ds, dpre, wv, total4, len / 4);
    const int splits = 32;
    int rps = (B + splits - 1) / splits;
    hipLaunchKernelGGL(k_value_wgrad, dim3((len + 1 + 255) / 256, splits), dim3(256), 0, st_grad, act, dpre, B, len, rps, part);
    hipLaunchKernelGGL(k_value_wgrad_finalize, dim3((len + 1 + 255) / 256), dim3(256), 0, st_grad, part, splits, F, nsq, grad_w, grad_b);
    return hipGetLastError();
}

// workspace floats needed by launch_wgrad_* for the given shape
static void wgrad_plan_conv(int B, int nsq, int ntiles, int* pw, int* cps, int* splits) {
    // positions per chunk: what the register prefetch of PF = 4 carries (64 rows): 5×5 → 2 positions = 50 rows (one position
    // = 25 rows padded to 28 wasted 11 % of the MFMAs and paid a pair of barriers every 7 k-steps: 23.2 → 22.4 ms per chunk of the
    // C5 network); 4 positions = 100 rows need PF = 7, which leaves one wave per SIMD (112 + 144 registers) and is no faster
    // (23.2 ms).  TG_WGRAD_PW overrides (A/B).
    static const int forced = env_int("TG_WGRAD_PW");
    *pw = forced > 0 ? forced : 64 / nsq;
    if (*pw * nsq > 112) *pw = 112 / nsq;
    if (*pw < 1) *pw = 1;
    int chunks = (B + *pw - 1) / *pw;
    int target = 512 / ntiles;  // two resident workgroups per CU (register-limited): one full round, half the partials to reduce
    if (target < 1) target = 1;
    *cps = (chunks + target - 1) / target;
    *splits = (chunks + *cps - 1) / *cps;
}
size_t wgrad_conv_workspace(int B, int n, int I, int O) {
    int ncib = (I + 63) / 64, ncob = (O + 63) / 64, pw, cps, splits;
    wgrad_plan_conv(B, n * n, ncib * ncob, &pw, &cps, &splits);
    return (size_t)splits * ncib * ncob * 9 * 4096;
}
hipError_t launch_wgrad_conv(hipStream_t st, const float* X, int xs, int I, const float* G, int gs, int O, int B, int n, float* part,
                             float* grad) {
    const int nsq = n * n;
    int ncib = (I + 63) / 64, ncob = (O + 63) / 64, pw, cps, splits;
    wgrad_plan_conv(B, nsq, ncib * ncob, &pw, &cps, &splits);
    const int rows_chunk = pw * nsq, rows_pad = (rows_chunk + 3) & ~3;
    size_t lds = ((size_t)(rows_chunk + 1) * 17 + (size_t)rows_pad * 17) * 16 + (size_t)rows_pad * 4;
    const int xvalid = xs < ncib * 64 ? xs : ncib * 64;  // columns that exist in memory
    const int gvalid = gs < ncob * 64 ? gs : ncob * 64;
    static const bool no_halo = env_on("TG_NO_HALO_WGRAD") || env_int("TG_WGRAD_PW") != 0;
    if (!no_halo && n == 5 && pw == 2) {
        constexpr size_t hl = ((size_t)(5 + 2 + 2 * 36 + 1) * 16 + 52 * 17) * 16;
        hipLaunchKernelGGL((k_wgrad_halo<5, 2>), dim3(splits, ncib * ncob), dim3(256), hl, st, X, xs, xvalid, G, gs, gvalid, B * nsq, cps, ncob, part);
    } else if (!no_halo && n == 6 && pw == 1) {
        constexpr size_t hl = ((size_t)(6 + 2 + 1 * 49 + 1) * 16 + 36 * 17) * 16;
        hipLaunchKernelGGL((k_wgrad_halo<6, 1>), dim3(splits, ncib * ncob), dim3(256), hl, st, X, xs, xvalid, G, gs, gvalid, B * nsq, cps, ncob, part);
    } else if (rows_chunk <= 64)
        hipLaunchKernelGGL((k_wgrad<true, 4>), dim3(splits, ncib * ncob), dim3(256), lds, st, X, xs, xvalid, G, gs, gvalid, B * nsq, n, nsq,
                           rows_chunk, cps, ncob, part);
    else
        hipLaunchKernelGGL((k_wgrad<true, 7>), dim3(splits, ncib * ncob), dim3(256), lds, st, X, xs, xvalid, G, gs, gvalid, B * nsq, n, nsq,
                           rows_chunk, cps, ncob, part);
    size_t total = (size_t)ncib * ncob * 9 * 4096;
    hipLaunchKernelGGL(k_wgrad_reduce_conv, dim3(blocks_for(total / 4)), dim3(256), 0, st, part, splits, ncib, ncob, O, I, grad);
    return hipGetLastError();
}
static void wgrad_plan_fc(int B, int* cps, int* splits) {
    const int rows_chunk = 32;
    int chunks = (B + rows_chunk - 1) / rows_chunk;
    *cps = (chunks + 7) / 8;
    *splits = (chunks + *cps - 1) / *cps;
}
size_t wgrad_fc_workspace(int B, int K, int P) {
    int ncit = (K + 255) / 256, ncob = (P + 63) / 64, cps, splits;
    wgrad_plan_fc(B, &cps, &splits);
    return (size_t)splits * ncit * ncob * 4 * 4096;
}
hipError_t launch_wgrad_fc(hipStream_t st, const float* S, int K, const float* G, int gs, int P, int B, int F, int nsq, float* part,
                           float* grad) {
    int ncit = (K + 255) / 256, ncob = (P + 63) / 64, cps, splits;
    wgrad_plan_fc(B, &cps, &splits);
    const int rows_chunk = 32;
    size_t lds = ((size_t)(rows_chunk + 1) * 65 + (size_t)rows_chunk * 17) * 16 + (size_t)rows_chunk * 4;
    const int gvalid = gs < ncob * 64 ? gs : ncob * 64;
    hipLaunchKernelGGL((k_wgrad<false>), dim3(splits, ncit * ncob), dim3(256), lds, st, S, K, K, G, gs, gvalid, B, 1, 1, rows_chunk, cps,
                       ncob, part);
    size_t total = (size_t)ncit * ncob * 4 * 4096;
    hipLaunchKernelGGL(k_wgrad_reduce_fc, dim3(blocks_for(total)), dim3(256), 0, st, part, splits, ncit, ncob, P, F, nsq, grad);
    return hipGetLastError();
}
hipError_t launch_adam(hipStream_t st, float* p, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps,
                       float wd, float bc1, float bc2_sqrt, float gscale) {
    hipLaunchKernelGGL(k_adam, dim3(blocks_for(n)), dim3(256), 0, st, p, g, m, v, n, lr, b1, b2, eps, wd, bc1, bc2_sqrt, gscale);
    return hipGetLastError();
}
hipError_t launch_sum_rows(hipStream_t st, const float* rows, int n, double* out) {
    hipLaunchKernelGGL(k_sum_rows, dim3(1), dim3(256), 0, st, rows, n, out);
    return hipGetLastError();
}

}  // namespace tg
