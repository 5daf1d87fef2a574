// row_rmw_probe.hip -- what does the MI355X give a kernel that does nothing but the memory pattern of the
// sparse LazyAdam apply?  Random, unique 960-byte rows of a 192 GB table are read, changed and written back
// (the p|m|v row of mrec_apply.hip), optionally together with a streamed 160-byte bf16 "gradient" row per id.
// No index structure, no optimizer arithmetic: the time of this kernel is the ceiling for that access pattern.
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/row_rmw_probe tools/probes/row_rmw_probe.hip
//   run:   tools/probes/row_rmw_probe [rows=200000000] [n=425984] [batch=2]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));

// 20 lanes per row (3 groups per wave64, 4 lanes idle), W rows per group walked BATCH at a time.
template <int BATCH, bool NT, bool GRAD>
__global__ __launch_bounds__(256) void k_rmw(float* __restrict__ table, const int* __restrict__ rows, int n, int W,
                                             const uint16_t* __restrict__ grad) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane / 20, sub = lane - grp * 20;
    if (grp >= 3) return;
    const int64_t g = ((int64_t)blockIdx.x * 4 + wave) * 3 + grp;
    const int64_t s = g * W;
    for (int j = 0; j < W; j += BATCH) {
        f4 st[BATCH][3];
        float gsum[BATCH];
        int64_t off[BATCH];
#pragma unroll
        for (int k = 0; k < BATCH; ++k) {
            const int64_t e = s + j + k;
            off[k] = -1;
            gsum[k] = 1.0f;
            if (e < n) {
                off[k] = (int64_t)rows[e] * 240 + sub * 4;
                if (GRAD) {
                    u2 t = __builtin_nontemporal_load((const u2*)(grad + e * 80 + sub * 4));
                    gsum[k] = __uint_as_float(t.x << 16) + __uint_as_float(t.y << 16);
                }
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    st[k][i] = NT ? __builtin_nontemporal_load((const f4*)(table + off[k] + 80 * i)) : *(const f4*)(table + off[k] + 80 * i);
            }
        }
#pragma unroll
        for (int k = 0; k < BATCH; ++k) {
            if (off[k] >= 0) {
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    f4 x = st[k][i] + gsum[k];
                    if (NT) __builtin_nontemporal_store(x, (f4*)(table + off[k] + 80 * i)); else *(f4*)(table + off[k] + 80 * i) = x;
                }
            }
        }
    }
}

// The lookup's pattern: random 320-byte rows (80 floats) read, rounded to bf16, written as a dense [n, 80] bf16
// stream.  20 lanes per row, 4 rows per lane-group in flight.
__global__ __launch_bounds__(256) void k_gather_pat(const float* __restrict__ table, const int* __restrict__ rows, int n,
                                                    uint16_t* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane / 20, sub = lane - grp * 20;
    if (grp >= 3) return;
    const int64_t g = ((int64_t)blockIdx.x * 4 + wave) * 3 + grp;
    f4 v[4];
    int64_t e[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        e[k] = g * 4 + k;
        if (e[k] < n) v[k] = *(const f4*)(table + (int64_t)rows[e[k]] * 80 + sub * 4);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (e[k] < n) {
            u2 o;
            o.x = (__float_as_uint(v[k].x) >> 16) | (__float_as_uint(v[k].y) & 0xFFFF0000u);
            o.y = (__float_as_uint(v[k].z) >> 16) | (__float_as_uint(v[k].w) & 0xFFFF0000u);
            *(u2*)(out + e[k] * 80 + sub * 4) = o;
        }
    }
}

template <int BATCH, bool NT, bool GRAD>
float run(float* table, const int* rows, int n, int W, const uint16_t* grad, int iters) {
    const int64_t groups = (n + W - 1) / W;
    const unsigned blocks = (unsigned)((groups + 11) / 12);
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) k_rmw<BATCH, NT, GRAD><<<blocks, 256>>>(table, rows, n, W, grad);
    CK(hipDeviceSynchronize());
    float best = 1e9f, tot = 0;
    for (int i = 0; i < iters; ++i) {
        CK(hipEventRecord(a));
        k_rmw<BATCH, NT, GRAD><<<blocks, 256>>>(table, rows + (size_t)(i % 4) * n, n, W, grad);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        best = ms < best ? ms : best; tot += ms;
    }
    return tot / iters;
}

int main(int argc, char** argv) {
    const int64_t V = argc > 1 ? atoll(argv[1]) : 200000000LL;
    const int n = argc > 2 ? atoi(argv[2]) : 425984;
    size_t bytes = (size_t)V * 960;
    float* table; CK(hipMalloc(&table, bytes));
    CK(hipMemset(table, 0, bytes));
    // 4 sets of n distinct random rows (a multiplicative permutation of [0, V))
    std::vector<int> h((size_t)4 * n);
    uint64_t x = 88172645463325252ULL;
    for (size_t i = 0; i < h.size(); ++i) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        h[i] = (int)(x % (uint64_t)V);
    }
    int* rows; CK(hipMalloc(&rows, h.size() * 4)); CK(hipMemcpy(rows, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    uint16_t* grad; CK(hipMalloc(&grad, (size_t)n * 160)); CK(hipMemset(grad, 0, (size_t)n * 160));
    const double rmw = (double)n * 960 * 2, withg = rmw + (double)n * 160 + (double)n * 4;
    printf("table %.1f GB, %d random rows per launch; row read+write = %.1f MB, with bf16 gradient stream + ids = %.1f MB\n",
           bytes / 1e9, n, rmw / 1e6, withg / 1e6);
    const int Ws[] = {1, 4, 16};
    for (int W : Ws) {
        float t;
        t = run<1, true, false>(table, rows, n, W, grad, 20);  printf("W=%2d batch 1 nontemporal          : %7.1f us  %6.0f GB/s\n", W, t * 1e3, rmw / t / 1e6);
        t = run<2, true, false>(table, rows, n, W, grad, 20);  printf("W=%2d batch 2 nontemporal          : %7.1f us  %6.0f GB/s\n", W, t * 1e3, rmw / t / 1e6);
        t = run<4, true, false>(table, rows, n, W, grad, 20);  printf("W=%2d batch 4 nontemporal          : %7.1f us  %6.0f GB/s\n", W, t * 1e3, rmw / t / 1e6);
        t = run<2, false, false>(table, rows, n, W, grad, 20); printf("W=%2d batch 2 cached               : %7.1f us  %6.0f GB/s\n", W, t * 1e3, rmw / t / 1e6);
        t = run<2, true, true>(table, rows, n, W, grad, 20);   printf("W=%2d batch 2 nontemporal + gradient: %7.1f us  %6.0f GB/s (of %.1f MB)\n", W, t * 1e3, withg / t / 1e6, withg / 1e6);
    }
    {
        // lookup pattern over a [3*V, 80] view of the same memory (rows drawn from [0, V))
        const unsigned blocks = (unsigned)(((n + 3) / 4 + 11) / 12);
        hipEvent_t a, b;
        CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        for (int i = 0; i < 3; ++i) k_gather_pat<<<blocks, 256>>>(table, rows, n, grad);
        CK(hipDeviceSynchronize());
        float tot = 0;
        for (int i = 0; i < 20; ++i) {
            CK(hipEventRecord(a));
            k_gather_pat<<<blocks, 256>>>(table, rows + (size_t)(i % 4) * n, n, grad);
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            tot += ms;
        }
        const double by = (double)n * (4 + 320 + 160);
        printf("lookup pattern: random 320-B rows -> bf16 stream: %7.1f us  %6.0f GB/s (of %.1f MB)\n", tot / 20 * 1e3, by / (tot / 20) / 1e6, by / 1e6);
    }
    return 0;
}
