// rng.cuh — counter-based randomness for self-play.
//
// The reference draws from rand::thread_rng (alpha-tak/src/search/noise.rs:13, play.rs:63,
// train/src/self_play.rs:114), which is not reproducible.  Here every random decision is a pure
// function of (seed; slot, generation, ply, purpose, index, attempt) through Philox4x32-10, and the
// Dirichlet noise is built from f64 arithmetic restricted to + - * / (IEEE, no FMA contraction), so a
// run is reproducible across GPUs, across 1/2/4/8-way sharding, and against a CPU statement of the
// same spec.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tg {

enum : uint32_t { RNG_OPENING = 1, RNG_GAMMA = 2, RNG_PICK = 3 };

struct U4 { uint32_t v[4]; };

__host__ __device__ inline U4 philox4x32_10(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    U4 o;
    o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
    return o;
}

__host__ __device__ inline U4 rng_draw(uint64_t seed, uint32_t slot, uint32_t generation, uint32_t ply, uint32_t purpose,
                                       uint32_t index, uint32_t attempt) {
    return philox4x32_10(seed, slot, generation, ply | (purpose << 16), index | (attempt << 16));
}

__host__ __device__ inline double bits_f64(uint64_t b) {
    union { uint64_t u; double d; } x;
    x.u = b;
    return x.d;
}
__host__ __device__ inline uint64_t f64_bits(double d) {
    union { uint64_t u; double d; } x;
    x.d = d;
    return x.u;
}

// sqrt / log / exp from + - * / only: identical results on every IEEE machine
__host__ __device__ inline double det_sqrt(double a) {
    double x = bits_f64((f64_bits(a) >> 1) + 0x1FF8000000000000ull);
    for (int i = 0; i < 6; i++) x = 0.5 * (x + a / x);
    return x;
}
__host__ __device__ inline double det_log(double x) {
    uint64_t b = f64_bits(x);
    int e = (int)((b >> 52) & 0x7ff) - 1023;
    double m = bits_f64((b & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    double t = (m - 1.0) / (m + 1.0);
    double t2 = t * t;
    double p = 1.0 / 21.0;
    p = p * t2 + 1.0 / 19.0;
    p = p * t2 + 1.0 / 17.0;
    p = p * t2 + 1.0 / 15.0;
    p = p * t2 + 1.0 / 13.0;
    p = p * t2 + 1.0 / 11.0;
    p = p * t2 + 1.0 / 9.0;
    p = p * t2 + 1.0 / 7.0;
    p = p * t2 + 1.0 / 5.0;
    p = p * t2 + 1.0 / 3.0;
    p = p * t2 + 1.0;
    return (double)e * 0.6931471805599453 + 2.0 * t * p;
}
__host__ __device__ inline double det_exp(double y) {
    if (y < -700.0) return 0.0;
    if (y > 700.0) y = 700.0;
    double t = y * 1.4426950408889634;
    long long k = (long long)(t < 0 ? t - 0.5 : t + 0.5);
    double r = (y - (double)k * 0.6931471803691238) - (double)k * 1.9082149292705877e-10;
    double p = 1.0;
    for (int i = 18; i >= 1; i--) p = 1.0 + p * (r / (double)i);
    return bits_f64(f64_bits(p) + ((uint64_t)k << 52));
}
__host__ __device__ inline double u32_unit(uint32_t x) { return ((double)x + 0.5) * (1.0 / 4294967296.0); }

// Gamma(alpha, 1), Marsaglia–Tsang (+ U^(1/alpha) boost below 1); draws: (RNG_GAMMA, index, attempt)
__host__ __device__ inline double gamma_sample(double alpha, uint64_t seed, uint32_t slot, uint32_t generation, uint32_t ply,
                                               uint32_t index) {
    double a = alpha < 1.0 ? alpha + 1.0 : alpha;
    double d = a - 1.0 / 3.0;
    double c = 1.0 / det_sqrt(9.0 * d);
    for (uint32_t attempt = 0; attempt < 65535; attempt++) {
        U4 r = rng_draw(seed, slot, generation, ply, RNG_GAMMA, index, attempt);
        double v1 = 2.0 * u32_unit(r.v[0]) - 1.0;
        double v2 = 2.0 * u32_unit(r.v[1]) - 1.0;
        double s = v1 * v1 + v2 * v2;
        if (s >= 1.0 || s == 0.0) continue;
        double x = v1 * det_sqrt(-2.0 * det_log(s) / s);
        double w = 1.0 + c * x;
        if (w <= 0.0) continue;
        double v = w * w * w;
        double u = u32_unit(r.v[2]);
        if (det_log(u) < 0.5 * x * x + d - d * v + d * det_log(v)) {
            double g = d * v;
            if (alpha < 1.0) g = g * det_exp(det_log(u32_unit(r.v[3])) / alpha);
            return g;
        }
    }
    return d;
}

// test evaluator TG_EVAL_HASH: pseudo-random but exactly representable policy / eval from a state hash
__host__ __device__ inline uint64_t mix64(uint64_t x) {
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27; x *= 0x94d049bb133111ebull;
    x ^= x >> 31;
    return x;
}
__host__ __device__ inline float hash_policy(uint64_t h, uint32_t index) {
    uint64_t v = mix64(h ^ (0x9E3779B97F4A7C15ull * (uint64_t)(index + 1)));
    return (float)((uint32_t)(v >> 40) + 1u) * (1.0f / 16777216.0f);
}
__host__ __device__ inline float hash_eval(uint64_t h) {
    uint64_t v = mix64(h ^ 0xD6E8FEB86659FD93ull);
    return (float)(uint32_t)(v >> 40) * (1.0f / 8388608.0f) - 1.0f;
}

}  // namespace tg
