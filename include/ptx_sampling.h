/* ptx_sampling.h — normative definition of the counter-based per-path random numbers and of the
 * deterministic sine/cosine that the MI355X path tracer uses.
 *
 * The reference draws its samples sequentially from one OS-seeded `SmallRng` per thread
 * (tracing/src/worker.rs:35, tracing/src/sampling.rs:5-48, tracing/src/material.rs:181), so its
 * stream is not reproducible and is deliberately NOT reproduced here (parity unpinned
 * at stream level).  What is kept are the *distributions* and the *consumption order*:
 *
 *   dimension 0,1                      pixel jitter x, y            U[0,1)     sampling.rs:5-7
 *   per bounce b (1-based), base(b) = 2 + (b-1)*(3 + 2*n_lights):
 *     base+0                           lobe pick                    U[0,1)     material.rs:181
 *     base+1, base+2                   concentric disk x, y         U[-1,1]    sampling.rs:28-29
 *     base+3+2l, base+4+2l             light l: z, angle            U[-1,1), U[0,tau)  sampling.rs:10-11
 *
 * A number is a pure function of (seed, pixel index y*W+x, global sample index, dimension), so any
 * conforming implementation (GPU kernels, a CPU checker) traces the same paths whatever the order in
 * which paths are scheduled, and an image does not depend on how samples are sharded over GPUs.
 *
 * Everything is plain IEEE-754 binary32 / integer arithmetic with a fixed evaluation order; code that
 * includes this header must be compiled with floating-point contraction off (-ffp-contract=off).
 */
#ifndef PTX_SAMPLING_H
#define PTX_SAMPLING_H

#include <stdint.h>

#if defined(__HIPCC__)
#define PTX_HD __host__ __device__ static inline
#else
#define PTX_HD static inline
#endif

/* SplitMix64 output function (Steele, Lea, Flood 2014). */
PTX_HD uint64_t ptx_mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* Key of one path = one (pixel, sample) pair. */
PTX_HD uint64_t ptx_path_key(uint64_t seed, uint32_t pixel, uint32_t sample) {
    return ptx_mix64(seed ^ ptx_mix64(((uint64_t)sample << 32) | (uint64_t)pixel));
}

/* 24 uniform bits for dimension `dim` of the path. */
PTX_HD uint32_t ptx_bits24(uint64_t key, uint32_t dim) {
    return (uint32_t)(ptx_mix64(key + (uint64_t)dim * 0xD1342543DE82EF95ull) >> 40);
}

/* U[0,1): k * 2^-24 (the 24-bit precision `rand` gives an f32). */
PTX_HD float ptx_u01(uint64_t key, uint32_t dim) {
    return (float)ptx_bits24(key, dim) * 5.9604644775390625e-8f;
}

/* U[-1,1): sampling.rs:10 `random_range(-1.0..1.0)`. */
PTX_HD float ptx_sym_open(uint64_t key, uint32_t dim) {
    return 2.0f * ptx_u01(key, dim) - 1.0f;
}

/* U[-1,1]: sampling.rs:28 `random_range(-1.0..=1.0)`. */
PTX_HD float ptx_sym_closed(uint64_t key, uint32_t dim) {
    return 2.0f * ((float)ptx_bits24(key, dim) * (1.0f / 16777215.0f)) - 1.0f;
}

/* U[0,tau): sampling.rs:11 `random_range(0.0..TAU)`. */
PTX_HD float ptx_angle(uint64_t key, uint32_t dim) {
    return ptx_u01(key, dim) * 6.28318530717958647692f;
}

PTX_HD uint32_t ptx_dim_base(uint32_t bounce /*1-based*/, uint32_t n_lights) {
    return 2u + (bounce - 1u) * (3u + 2u * n_lights);
}

/* Deterministic sin/cos for |x| <= ~8 (the path tracer only needs [-pi/4, 2*pi]).
 * Three-term Cody-Waite reduction by pi/2, then the Cephes single-precision minimax polynomials on
 * [-pi/4, pi/4]; every operation is a single rounded f32 add or multiply in the order written, so the
 * result is identical on gfx950 and on any IEEE host.  Maximum error ~1 ulp in the stated range. */
PTX_HD void ptx_sincos(float x, float* s_out, float* c_out) {
    float kf = x * 0.636619772367581343f;
    /* round to nearest integer, ties away from zero (|kf| is small, the int cast is exact) */
    int k = (int)(kf + (kf >= 0.0f ? 0.5f : -0.5f));
    float fk = (float)k;
    float r = x - fk * 1.5703125f;
    r = r - fk * 4.837512969970703125e-4f;
    r = r - fk * 7.54978995489188216e-8f;
    float z = r * r;
    float ps = -1.9515295891e-4f * z + 8.3321608736e-3f;
    ps = ps * z - 1.6666654611e-1f;
    float sn = (ps * z) * r + r;
    float pc = 2.443315711809948e-5f * z - 1.388731625493765e-3f;
    pc = pc * z + 4.166664568298827e-2f;
    float cs = (pc * z) * z - 0.5f * z + 1.0f;
    switch (k & 3) {
        case 0: *s_out = sn; *c_out = cs; break;
        case 1: *s_out = cs; *c_out = -sn; break;
        case 2: *s_out = -sn; *c_out = -cs; break;
        default: *s_out = -cs; *c_out = sn; break;
    }
}

#endif /* PTX_SAMPLING_H */
