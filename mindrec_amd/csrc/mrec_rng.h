// mrec_rng.h -- counter-based N(0,1) generator for table / default-row initialisation.
// Spec (shared bit-for-bit with oracle/mrec_oracle.c, which states it independently): columns come in PAIRS, both outputs of
// one Box-Muller transform (half the hashing, logarithm, root and trigonometry per value -- default rows of new hash-table
// keys and the initialisation of a table are ALU-bound on this generator):
//   h  = mix64(seed ^ mix64(row * 0xD1342543DE82EF95 + (col >> 1)))
//   u1 = ((h >> 40) + 1) * 2^-24  in (0,1],   k = (h >> 8) & 0xFFFFFF
//   z(col even) = sqrt(-2 ln u1) * cos(2 pi k / 2^24),   z(col odd) = sqrt(-2 ln u1) * sin(2 pi k / 2^24)
// ln and cos are polynomial kernels whose every multiply-add is an explicit fma, so the result
// does not depend on the compiler's contraction choices (the library is built -ffp-contract=off).
#pragma once
#include "mrec_common.h"

__device__ __forceinline__ float mrec_det_logf(float x) {
    uint32_t xb = __float_as_uint(x);
    int e = (int)((xb >> 23) & 0xFFu) - 126;
    float m = __uint_as_float((xb & 0x007FFFFFu) | 0x3F000000u);
    if (m < 0.70710678118654752440f) {
        e -= 1;
        m = m + m - 1.0f;
    } else {
        m = m - 1.0f;
    }
    float z = m * m;
    float y = 7.0376836292E-2f;
    y = __builtin_fmaf(y, m, -1.1514610310E-1f);
    y = __builtin_fmaf(y, m, 1.1676998740E-1f);
    y = __builtin_fmaf(y, m, -1.2420140846E-1f);
    y = __builtin_fmaf(y, m, 1.4249322787E-1f);
    y = __builtin_fmaf(y, m, -1.6668057665E-1f);
    y = __builtin_fmaf(y, m, 2.0000714765E-1f);
    y = __builtin_fmaf(y, m, -2.4999993993E-1f);
    y = __builtin_fmaf(y, m, 3.3333331174E-1f);
    y = y * m;
    y = y * z;
    float fe = (float)e;
    y = __builtin_fmaf(-2.12194440e-4f, fe, y);
    y = __builtin_fmaf(-0.5f, z, y);
    float r = m + y;
    r = __builtin_fmaf(0.693359375f, fe, r);
    return r;
}

// (cos, sin)(2 pi k / 2^24) for integer k in [0, 2^24): integer quadrant reduction + fma polynomials on [0, pi/4]
__device__ __forceinline__ void mrec_det_sincos2pi_u24(uint32_t k, float& cos_out, float& sin_out) {
    uint32_t q = k >> 22;
    uint32_t r = k & 0x3FFFFFu;
    bool swap = false;
    if (r > 0x200000u) { r = 0x400000u - r; swap = true; }
    float t = (float)r * 3.7450703e-07f;
    float z = t * t;
    float s = -1.9515295891E-4f;
    s = __builtin_fmaf(s, z, 8.3321608736E-3f);
    s = __builtin_fmaf(s, z, -1.6666654611E-1f);
    s = s * z;
    s = __builtin_fmaf(s, t, t);
    float c = 2.443315711809948E-005f;
    c = __builtin_fmaf(c, z, -1.388731625493765E-003f);
    c = __builtin_fmaf(c, z, 4.166664568298827E-002f);
    c = c * z;
    c = c * z;
    c = __builtin_fmaf(-0.5f, z, c);
    c = c + 1.0f;
    const float cs = swap ? s : c;      // of the angle inside the quadrant
    const float sn = swap ? c : s;
    const float co = (q & 1u) ? sn : cs, so = (q & 1u) ? cs : sn;
    cos_out = (q == 1u || q == 2u) ? -co : co;
    sin_out = (q >= 2u) ? -so : so;
}

// columns 2 * pair and 2 * pair + 1 of `row`
__device__ __forceinline__ void mrec_det_normal2(uint64_t seed, int64_t row, int32_t pair, float& z0, float& z1) {
    uint64_t h = mrec_mix64(seed ^ mrec_mix64((uint64_t)row * 0xD1342543DE82EF95ull + (uint64_t)(uint32_t)pair));
    uint32_t a = (uint32_t)(h >> 40);
    uint32_t b = (uint32_t)(h >> 8) & 0xFFFFFFu;
    float u1 = ((float)a + 1.0f) * 5.9604644775390625e-08f;
    float rad = sqrtf(-2.0f * mrec_det_logf(u1));
    float c, s;
    mrec_det_sincos2pi_u24(b, c, s);
    z0 = rad * c;
    z1 = rad * s;
}

__device__ __forceinline__ float mrec_det_normal(uint64_t seed, int64_t row, int32_t col) {
    float z0, z1;
    mrec_det_normal2(seed, row, col >> 1, z0, z1);
    return (col & 1) ? z1 : z0;
}
