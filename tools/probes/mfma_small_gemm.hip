// mfma_small_gemm.hip -- prototype: C[M,N] = relu(A[M,K] . Bt[N,K]^T + bias) in bf16 with fp32 accumulation on the
// gfx950 matrix cores (v_mfma_f32_32x32x16_bf16), for the small layers of the Wide&Deep MLP (N = 128 / 256 / 512,
// K = 256 / 512 / 1024, M = 16384) where the GEMM library runs 64-tile kernels on 256 CUs at ~0.2 PFLOP/s.
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/mfma_small_gemm tools/probes/mfma_small_gemm.hip
// Workgroup = 4 waves; wave (wm, wn) owns a 64 x 64 block of C (2 x 2 MFMA tiles, 64 accumulator VGPRs).  A K-tile of
// 64 is staged global -> LDS with coalesced 16-byte loads (row pitch 144 B: conflict-free 16-byte fragment reads), each
// lane then reads its 8-element fragments A[row r][k = 8h + j], B[k = 8h + j][col r] (r = lane & 31, h = lane >> 5).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int KT = 64;            // K per staged tile
constexpr int PITCH = KT + 8;     // bf16 elements per LDS row (144 B)

__device__ __forceinline__ uint16_t f2bf(float x) { __bf16 b = (__bf16)x; return __builtin_bit_cast(uint16_t, b); }

// RW x CW waves per workgroup (RW * CW == 4): the workgroup's C block is (64 RW) x (64 CW).
template <int RW, int CW>
__global__ __launch_bounds__(256) void k_gemm_nt_bias_relu(const uint16_t* __restrict__ A, const uint16_t* __restrict__ Bt,
                                                           const uint16_t* __restrict__ bias, uint16_t* __restrict__ C,
                                                           int M, int N, int K) {
    constexpr int TM = 64 * RW, TN = 64 * CW;
    __shared__ __attribute__((aligned(16))) uint16_t sA[TM * PITCH];
    __shared__ __attribute__((aligned(16))) uint16_t sB[TN * PITCH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave / CW, wn = wave % CW;
    const int64_t m0 = (int64_t)blockIdx.x * TM, n0 = (int64_t)blockIdx.y * TN;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.0f;

    for (int k0 = 0; k0 < K; k0 += KT) {
        // stage: every thread moves 16-byte pieces; a row of the tile is 8 pieces
        for (int p = threadIdx.x; p < TM * 8; p += 256) {
            const int row = p >> 3, c = p & 7;
            *(uint4*)(sA + row * PITCH + c * 8) = *(const uint4*)(A + (m0 + row) * K + k0 + c * 8);
        }
        for (int p = threadIdx.x; p < TN * 8; p += 256) {
            const int row = p >> 3, c = p & 7;
            *(uint4*)(sB + row * PITCH + c * 8) = *(const uint4*)(Bt + (n0 + row) * K + k0 + c * 8);
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < KT / 16; ++kk) {
            bf16x8 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = *(const bf16x8*)(sA + (wm * 64 + i * 32 + r) * PITCH + kk * 16 + 8 * h);
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = *(const bf16x8*)(sB + (wn * 64 + j * 32 + r) * PITCH + kk * 16 + 8 * h);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    // epilogue: C tile element (row = (q&3) + 8*(q>>2) + 4*h, col = r) of each 32x32 tile
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t col = n0 + wn * 64 + j * 32 + r;
            const float bv = __uint_as_float(((unsigned)bias[col]) << 16);
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int64_t row = m0 + wm * 64 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
                float v = acc[i][j][q] + bv;
                v = v > 0.0f ? v : 0.0f;
                C[row * N + col] = f2bf(v);
            }
        }
}

static uint16_t h_f2bf(float x) { uint32_t u; memcpy(&u, &x, 4); uint32_t r = u + 0x7FFF + ((u >> 16) & 1); return (uint16_t)(r >> 16); }
static float h_bf2f(uint16_t x) { uint32_t u = ((uint32_t)x) << 16; float f; memcpy(&f, &u, 4); return f; }

template <int RW, int CW>
void run(const char* tag, int M, int N, int K) {
    std::vector<uint16_t> hA((size_t)M * K), hB((size_t)N * K), hb(N), hC((size_t)M * N);
    srand(1);
    for (auto& x : hA) x = h_f2bf((rand() % 2001 - 1000) / 1000.0f);
    for (auto& x : hB) x = h_f2bf((rand() % 2001 - 1000) / 4000.0f);
    for (auto& x : hb) x = h_f2bf((rand() % 2001 - 1000) / 1000.0f);
    uint16_t *dA, *dB, *db, *dC;
    CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&dB, hB.size() * 2)); CK(hipMalloc(&db, N * 2)); CK(hipMalloc(&dC, hC.size() * 2));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), N * 2, hipMemcpyHostToDevice));
    dim3 grid(M / (64 * RW), N / (64 * CW));
    k_gemm_nt_bias_relu<RW, CW><<<grid, 256>>>(dA, dB, db, dC, M, N, K);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hC.data(), dC, hC.size() * 2, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int t = 0; t < 2000; ++t) {
        const int i = rand() % M, j = rand() % N;
        double s = h_bf2f(hb[j]);
        for (int k = 0; k < K; ++k) s += (double)h_bf2f(hA[(size_t)i * K + k]) * h_bf2f(hB[(size_t)j * K + k]);
        if (s < 0) s = 0;
        const double got = h_bf2f(hC[(size_t)i * N + j]);
        const double err = fabs(got - s) / (fabs(s) + 1e-2);
        if (err > worst) worst = err;
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) k_gemm_nt_bias_relu<RW, CW><<<grid, 256>>>(dA, dB, db, dC, M, N, K);
    CK(hipEventRecord(e0));
    const int it = 50;
    for (int i = 0; i < it; ++i) k_gemm_nt_bias_relu<RW, CW><<<grid, 256>>>(dA, dB, db, dC, M, N, K);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms / it * 1e3, fl = 2.0 * M * N * K;
    printf("%-10s M=%d N=%d K=%d  grid %dx%d  %7.1f us  %6.3f PFLOP/s   worst rel err vs fp64 %.2e (bf16 ulp ~4e-3)\n", tag, M, N, K,
           grid.x, grid.y, us, fl / us / 1e9, worst);
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(db)); CK(hipFree(dC));
}

int main() {
    run<1, 4>("1x4", 16384, 256, 512);      // layer 2 forward
    run<2, 2>("2x2", 16384, 256, 512);
    run<2, 2>("2x2", 16384, 128, 256);      // layer 3 forward
    run<1, 4>("1x4", 16384, 512, 1024);     // layer 1 forward
    run<2, 2>("2x2", 16384, 512, 1024);
    run<2, 2>("2x2", 16384, 1024, 2048);    // about layer 0 forward (its K = 2080 is not a multiple of 64)
    return 0;
}
