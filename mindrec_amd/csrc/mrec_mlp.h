// mrec_mlp.h -- helpers shared by the output-head kernels (mrec_mlp.hip) and the tail launch (mrec_tail.hip): 8-element 16-bit
// pack / unpack in both 16-bit formats, and the fixed-order column sum of per-workgroup partial rows.
#pragma once
#include "mrec_common.h"

namespace {

__device__ __forceinline__ float bf2f(uint16_t x) { return __uint_as_float(((unsigned)x) << 16); }
__device__ __forceinline__ uint16_t f2bf(float x) { __bf16 b = (__bf16)x; return __builtin_bit_cast(uint16_t, b); }

struct bf8 { uint4 u; };   // 8 bf16

__device__ __forceinline__ void unpack8(const uint4& u, float (&f)[8]) {
    f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xFFFF0000u);
    f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xFFFF0000u);
    f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xFFFF0000u);
    f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xFFFF0000u);
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
    uint4 u;
    u.x = (unsigned)f2bf(f[0]) | ((unsigned)f2bf(f[1]) << 16);
    u.y = (unsigned)f2bf(f[2]) | ((unsigned)f2bf(f[3]) << 16);
    u.z = (unsigned)f2bf(f[4]) | ((unsigned)f2bf(f[5]) << 16);
    u.w = (unsigned)f2bf(f[6]) | ((unsigned)f2bf(f[7]) << 16);
    return u;
}

// fp16 forms (the reference's mixed-precision dtype, wide_and_deep.py:119-128) for the output head
template <bool F16> __device__ __forceinline__ void unpack8t(const uint4& u, float (&f)[8]) {
    if (!F16) { unpack8(u, f); return; }
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) { const h2 v = __builtin_bit_cast(h2, w[k]); f[2 * k] = (float)v[0]; f[2 * k + 1] = (float)v[1]; }
}
template <bool F16> __device__ __forceinline__ uint4 pack8t(const float (&f)[8]) {
    if (!F16) return pack8(f);
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    unsigned w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { const h2 v = {(_Float16)f[2 * k], (_Float16)f[2 * k + 1]}; w[k] = __builtin_bit_cast(unsigned, v); }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

constexpr int MB = 256;   // threads per block

// out[c] = sum over blocks of partial[blk][c] in a fixed order.  A block owns 32 adjacent columns and
// splits the partial rows over 8 row-groups; every thread keeps 8 independent loads in flight (a single
// thread walking 512 partials serially paid a memory round trip per partial: 117 us for this "tiny" step).
__device__ __forceinline__ float finish_column(const float* __restrict__ partial, int nblk, int W, int c,
                                               float (*sm)[32]) {
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int chunk = (nblk + 7) / 8;
    const int b0 = ry * chunk, b1 = (b0 + chunk < nblk) ? b0 + chunk : nblk;
    float s = 0.0f;
    if (c < W) {
        int b = b0;
        for (; b + 8 <= b1; b += 8) {
            float t[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) t[k] = partial[(int64_t)(b + k) * W + c];
#pragma unroll
            for (int k = 0; k < 8; ++k) s += t[k];
        }
        for (; b < b1; ++b) s += partial[(int64_t)b * W + c];
    }
    sm[ry][cx] = s;
    __syncthreads();
    float tot = 0.0f;
    if (ry == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) tot += sm[k][cx];
    }
    return tot;   // valid for ry == 0
}

}  // namespace
