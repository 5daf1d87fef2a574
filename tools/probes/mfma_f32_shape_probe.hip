// Bare fp32 MFMA loops on random data, fragments re-read from LDS every step, two workgroups of four waves per CU: what rate does the chip
// hold with v_mfma_f32_32x32x2_f32 against v_mfma_f32_16x16x4_f32?  (MI355X guide, DVFS give-back (7): the clock held under MFMA
// load can depend on the instruction's shape.)   hipcc --offload-arch=gfx950 -O3 -o mfma_f32_shape_probe mfma_f32_shape_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int LD = 132, BK = 32;

template <int SHAPE>
__global__ __launch_bounds__(256, 2) void k_loop(const float* __restrict__ src, float* __restrict__ out, int iters) {
    __shared__ float As[BK * LD], Bs[BK * LD];
    const int t = threadIdx.x, l = t & 63, w = t >> 6, wr = w >> 1, wc = w & 1;
    for (int i = t; i < BK * LD; i += 256) { As[i] = src[i]; Bs[i] = src[BK * LD + i]; }
    __syncthreads();
    if (SHAPE == 32) {
        f32x16 acc[2][2];
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        const float* ar = As + (l >> 5) * LD + wr * 64 + (l & 31);
        const float* br = Bs + (l >> 5) * LD + wc * 64 + (l & 31);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int s = 0; s < BK / 2; ++s) {
                const float a0 = ar[2 * s * LD], a1 = ar[2 * s * LD + 32], b0 = br[2 * s * LD], b1 = br[2 * s * LD + 32];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
        }
        float s = 0.f;
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
        out[blockIdx.x * 256 + t] = s;
    } else {
        f32x4 acc[4][4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
        const float* ar = As + (l >> 4) * LD + wr * 64 + (l & 15);
        const float* br = Bs + (l >> 4) * LD + wc * 64 + (l & 15);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int s = 0; s < BK / 4; ++s) {
                float a[4], b[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) { a[i] = ar[4 * s * LD + 16 * i]; b[i] = br[4 * s * LD + 16 * i]; }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
        float s = 0.f;
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
        out[blockIdx.x * 256 + t] = s;
    }
}

int main() {
    const int n = 2 * BK * LD, blocks = 512, iters = 4000;
    std::vector<float> h(n);
    srand(1);
    for (auto& x : h) x = (float)rand() / RAND_MAX * 2.f - 1.f;
    float *src, *out;
    hipMalloc(&src, n * 4); hipMalloc(&out, blocks * 256 * 4);
    hipMemcpy(src, h.data(), n * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep)
        for (int shape : {32, 16}) {
            for (int warm = 0; warm < 2; ++warm) {
                hipEventRecord(e0);
                if (shape == 32) k_loop<32><<<blocks, 256>>>(src, out, iters); else k_loop<16><<<blocks, 256>>>(src, out, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flop = (double)blocks * 4 * iters * (BK / 2) * 4 * 4096.0;     // per wave and iteration: 64 x 64 x 32 x 2
            printf("shape %dx%d: %.3f ms  %.1f TFLOP/s\n", shape, shape, ms, flop / ms * 1e-9);
        }
    return 0;
}
