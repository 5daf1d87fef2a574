// mrec_common.h -- shared host/device helpers for libmrec_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mrec.h"

#define MREC_API extern "C" __attribute__((visibility("default")))

extern int g_mrec_last_hip_error;

#define MREC_HIP_CHECK(expr)                       \
    do {                                           \
        hipError_t _e = (expr);                    \
        if (_e != hipSuccess) {                    \
            g_mrec_last_hip_error = (int)_e;       \
            return MREC_EHIP;                      \
        }                                          \
    } while (0)

#define MREC_LAUNCH_CHECK() MREC_HIP_CHECK(hipGetLastError())

static inline size_t mrec_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Bump allocator over the caller's workspace; every block 256-B aligned.
struct MrecArena {
    char* base;
    size_t cap;
    size_t off;
    bool ok;
    MrecArena(void* ws, size_t bytes) : base((char*)ws), cap(bytes), off(0), ok(true) {}
    template <class T>
    T* take(size_t count) {
        size_t b = mrec_align_up(count * sizeof(T), 256);
        if (off + b > cap) { ok = false; off += b; return nullptr; }
        T* p = (T*)(base + off);
        off += b;
        return p;
    }
};

static inline int64_t mrec_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

constexpr int kWave = 64;

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

__host__ __device__ __forceinline__ uint64_t mrec_mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ uint32_t mrec_hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

__device__ __forceinline__ uint32_t mrec_hash_key(int32_t k) { return mrec_hash32((uint32_t)k); }
__device__ __forceinline__ uint32_t mrec_hash_key(int64_t k) {
    uint64_t h = mrec_mix64((uint64_t)k);
    return (uint32_t)(h ^ (h >> 32));
}

// Wave-level inclusive scan (sum) over 64 lanes.
__device__ __forceinline__ int wave_incl_scan(int x) {
    const int l = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int y = __shfl_up(x, d, 64);
        if (l >= d) x += y;
    }
    return x;
}

// Block-level exclusive scan for 256-thread blocks; returns exclusive prefix, *total = block sum.
// smem must hold >= 8 ints.
__device__ __forceinline__ int block_excl_scan_256(int x, int* smem, int* total) {
    const int w = threadIdx.x >> 6, l = lane_id();
    int incl = wave_incl_scan(x);
    if (l == 63) smem[w] = incl;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int s = smem[i];
        if (i < w) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + incl - x;
}
