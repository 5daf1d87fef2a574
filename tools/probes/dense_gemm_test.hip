// dense_gemm_test.hip -- standalone check + timing of the MFMA GEMM body (mindrec_amd/csrc/mrec_gemm.h) at the
// Wide&Deep MLP shapes, against an fp64 host reference on sampled outputs (asymmetric random operands).
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/probes/dense_gemm_test tools/probes/dense_gemm_test.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../mindrec_amd/csrc/mrec_gemm.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static uint16_t h_f2bf(float x) { uint32_t u; memcpy(&u, &x, 4); uint32_t r = u + 0x7FFF + ((u >> 16) & 1); return (uint16_t)(r >> 16); }
static float h_bf2f(uint16_t x) { uint32_t u = ((uint32_t)x) << 16; float f; memcpy(&f, &u, 4); return f; }
static uint32_t rng_state = 12345;
static float frand() { rng_state = rng_state * 1664525u + 1013904223u; return ((rng_state >> 8) & 0xFFFF) / 32768.0f - 1.0f; }

using namespace mgemm;

struct Timer {
    hipEvent_t e0, e1;
    Timer() { CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); }
    template <class F> double us(F f, int it = 30) {
        for (int i = 0; i < 3; ++i) f();
        CK(hipEventRecord(e0));
        for (int i = 0; i < it; ++i) f();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        return ms / it * 1e3;
    }
};

// forward: C[M,N] = relu(X[M,K] . W[K,N] + b)
template <int VAR = 0, int MR = 8>
static void test_fwd(int M, int N, int K) {
    std::vector<uint16_t> hX((size_t)M * K), hW((size_t)K * N), hC((size_t)M * N);
    std::vector<float> hb(N);
    for (auto& x : hX) x = h_f2bf(frand());
    for (auto& x : hW) x = h_f2bf(frand() * 0.05f);
    for (auto& x : hb) x = frand() * 0.1f;
    uint16_t *dX, *dW, *dC; float* db;
    CK(hipMalloc(&dX, hX.size() * 2)); CK(hipMalloc(&dW, hW.size() * 2)); CK(hipMalloc(&dC, hC.size() * 2)); CK(hipMalloc(&db, N * 4));
    CK(hipMemcpy(dX, hX.data(), hX.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dC, 0xFF, hC.size() * 2));
    Args a{};
    a.P = dX; a.Q = dW; a.C = dC; a.bias = db; a.ldp = K; a.ldq = N; a.ldc = N; a.Pext = M; a.Qext = N; a.K = K;
    a.nTp = (M + MR * 32 - 1) / (MR * 32); a.nTq = (N + 255) / 256; a.relu = 1; a.kt_per_slab = (K + 63) / 64;
    const int grid = a.nTp * a.nTq;
    auto run = [&] { k_gemm256<false, true, EPI_FWD, false, VAR, MR><<<grid, kThreads>>>(a); };
    run();
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hC.data(), dC, hC.size() * 2, hipMemcpyDeviceToHost));
    double worst = 0; int bad = 0;
    for (int t = 0; t < 4000; ++t) {
        int i = rand() % M, j = rand() % N;
        if (t < 64) { i = (t & 1) ? M - 1 - (t >> 1) : (t >> 1); }
        if (t >= 64 && t < 128) { j = (t & 1) ? N - 1 - ((t - 64) >> 1) : ((t - 64) >> 1); }
        double s = hb[j];
        for (int k = 0; k < K; ++k) s += (double)h_bf2f(hX[(size_t)i * K + k]) * h_bf2f(hW[(size_t)k * N + j]);
        if (s < 0) s = 0;
        const double got = h_bf2f(hC[(size_t)i * N + j]);
        const double err = fabs(got - s) / (fabs(s) + 0.05);
        if (err > worst) worst = err;
        if (err > 1e-2) { if (bad < 5) printf("   BAD fwd (%d,%d): got %g want %g\n", i, j, got, s); ++bad; }
    }
    Timer tm;
    const double us = tm.us(run);
    printf("fwd%d mr%d M=%d N=%d K=%d grid %d: %7.1f us %6.3f PF/s  worst rel err %.2e  %s\n", VAR, MR, M, N, K, grid, us, 2.0 * M * N * K / us / 1e9, worst,
           bad ? "FAIL" : "ok");
    CK(hipFree(dX)); CK(hipFree(dW)); CK(hipFree(dC)); CK(hipFree(db));
}

// dgrad: dX[M,Kin] = (dY[M,N] . W[Kin,N]^T) masked by H[M,Kin] > 0, + column sums
template <int MR = 8>
static void test_dgrad(int M, int Kin, int N, bool mask) {
    std::vector<uint16_t> hdY((size_t)M * N), hW((size_t)Kin * N), hH((size_t)M * Kin), hC((size_t)M * Kin);
    for (auto& x : hdY) x = h_f2bf(frand());
    for (auto& x : hW) x = h_f2bf(frand() * 0.05f);
    for (auto& x : hH) { float v = frand(); x = h_f2bf(v > 0.2f ? v : 0.f); }
    uint16_t *ddY, *dW, *dH, *dC; float* dws;
    const int nTp = (M + MR * 32 - 1) / (MR * 32), nTq = (Kin + 255) / 256;
    CK(hipMalloc(&ddY, hdY.size() * 2)); CK(hipMalloc(&dW, hW.size() * 2)); CK(hipMalloc(&dH, hH.size() * 2)); CK(hipMalloc(&dC, hC.size() * 2));
    CK(hipMalloc(&dws, (size_t)nTp * Kin * 4));
    CK(hipMemcpy(ddY, hdY.data(), hdY.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dH, hH.data(), hH.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(dC, 0xFF, hC.size() * 2));
    Args a{};
    a.P = ddY; a.Q = dW; a.C = dC; a.H = mask ? dH : nullptr; a.colsum_ws = mask ? dws : nullptr;
    a.ldp = N; a.ldq = N; a.ldc = Kin; a.Pext = M; a.Qext = Kin; a.K = N; a.nTp = nTp; a.nTq = nTq; a.kt_per_slab = (N + 63) / 64;
    const int grid = nTp * nTq;
    auto run = [&] { k_gemm256<false, false, EPI_DGRAD, false, 0, MR><<<grid, kThreads>>>(a); };
    run();
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hC.data(), dC, hC.size() * 2, hipMemcpyDeviceToHost));
    double worst = 0; int bad = 0;
    for (int t = 0; t < 4000; ++t) {
        int i = rand() % M, j = rand() % Kin;
        if (t < 64) j = Kin - 1 - t;
        double s = 0;
        for (int k = 0; k < N; ++k) s += (double)h_bf2f(hdY[(size_t)i * N + k]) * h_bf2f(hW[(size_t)j * N + k]);
        if (mask && !(h_bf2f(hH[(size_t)i * Kin + j]) > 0)) s = 0;
        const double got = h_bf2f(hC[(size_t)i * Kin + j]);
        const double err = fabs(got - s) / (fabs(s) + 0.05);
        if (err > worst) worst = err;
        if (err > 1e-2) { if (bad < 5) printf("   BAD dgrad (%d,%d): got %g want %g\n", i, j, got, s); ++bad; }
    }
    double cworst = 0;
    if (mask) {
        std::vector<float> hws((size_t)nTp * Kin);
        CK(hipMemcpy(hws.data(), dws, hws.size() * 4, hipMemcpyDeviceToHost));
        for (int t = 0; t < 40; ++t) {
            const int j = t < 8 ? Kin - 1 - t : rand() % Kin;
            double want = 0, got = 0;
            for (int i = 0; i < M; ++i) want += h_bf2f(hC[(size_t)i * Kin + j]);
            for (int tp = 0; tp < nTp; ++tp) got += hws[(size_t)tp * Kin + j];
            const double err = fabs(got - want) / (fabs(want) + 1.0);
            if (err > cworst) cworst = err;
            if (err > 1e-3) { if (bad < 5) printf("   BAD colsum %d: got %g want %g\n", j, got, want); ++bad; }
        }
    }
    Timer tm;
    const double us = tm.us(run);
    printf("dgrad mr%d M=%d Kin=%d N=%d mask=%d grid %d: %7.1f us %6.3f PF/s  worst rel err %.2e colsum %.2e  %s\n", MR, M, Kin, N, (int)mask, grid, us,
           2.0 * M * N * Kin / us / 1e9, worst, cworst, bad ? "FAIL" : "ok");
    CK(hipFree(ddY)); CK(hipFree(dW)); CK(hipFree(dH)); CK(hipFree(dC)); CK(hipFree(dws));
}

// wgrad: dW[Kin,N] = X[M,Kin]^T . dY[M,N], split over M in S slabs (fp32)
template <int MR = 8>
static void test_wgrad(int M, int Kin, int N, int S) {
    std::vector<uint16_t> hX((size_t)M * Kin), hdY((size_t)M * N);
    for (auto& x : hX) x = h_f2bf(frand());
    for (auto& x : hdY) x = h_f2bf(frand() * 0.05f);
    uint16_t *dX, *ddY; float* dC;
    const int nTp = (Kin + MR * 32 - 1) / (MR * 32), nTq = (N + 255) / 256;
    CK(hipMalloc(&dX, hX.size() * 2)); CK(hipMalloc(&ddY, hdY.size() * 2)); CK(hipMalloc(&dC, (size_t)S * Kin * N * 4));
    CK(hipMemcpy(dX, hX.data(), hX.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(ddY, hdY.data(), hdY.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(dC, 0xFF, (size_t)S * Kin * N * 4));
    Args a{};
    a.P = dX; a.Q = ddY; a.C = dC; a.ldp = Kin; a.ldq = N; a.ldc = N; a.Pext = Kin; a.Qext = N; a.K = M; a.nTp = nTp; a.nTq = nTq;
    const int Ttot = (M + 63) / 64;
    a.kt_per_slab = (Ttot + S - 1) / S; a.slab_stride = (int64_t)Kin * N;
    const int grid = nTp * nTq * S;
    auto run = [&] { k_gemm256<true, true, EPI_F32, false, 0, MR><<<grid, kThreads>>>(a); };
    run();
    CK(hipDeviceSynchronize());
    std::vector<float> hC((size_t)S * Kin * N);
    CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0; int bad = 0;
    for (int t = 0; t < 600; ++t) {
        int i = rand() % Kin, j = rand() % N;
        if (t < 40) i = Kin - 1 - t;
        double s = 0;
        for (int m = 0; m < M; ++m) s += (double)h_bf2f(hX[(size_t)m * Kin + i]) * h_bf2f(hdY[(size_t)m * N + j]);
        double got = 0;
        for (int z = 0; z < S; ++z) got += hC[(size_t)z * Kin * N + (size_t)i * N + j];
        const double err = fabs(got - s) / (fabs(s) + 0.5);
        if (err > worst) worst = err;
        if (err > 1e-3) { if (bad < 5) printf("   BAD wgrad (%d,%d): got %g want %g\n", i, j, got, s); ++bad; }
    }
    Timer tm;
    const double us = tm.us(run);
    printf("wgrad mr%d M=%d Kin=%d N=%d S=%d grid %d: %7.1f us %6.3f PF/s  worst rel err %.2e  %s\n", MR, M, Kin, N, S, grid, us, 2.0 * M * N * Kin / us / 1e9,
           worst, bad ? "FAIL" : "ok");
    CK(hipFree(dX)); CK(hipFree(ddY)); CK(hipFree(dC));
}


// Main-loop ablations of the forward kernel as the product launches it (both operands K-contiguous, fp16; timing only: the
// ablated variants compute garbage).  VAR bits: mrec_gemm.h.  VAR & 64: prints where a phase's cycles go.
template <int VAR>
static void time_fwd_kc(int M, int N, int K, const char* what) {
    std::vector<uint16_t> hX((size_t)M * K), hW((size_t)N * K);
    for (auto& x : hX) x = h_f2bf(frand());         // (bit patterns only matter for the clock the chip holds: full-range random)
    for (auto& x : hW) x = h_f2bf(frand());
    uint16_t *dX, *dW, *dC; float *db, *dws;
    CK(hipMalloc(&dX, hX.size() * 2)); CK(hipMalloc(&dW, hW.size() * 2)); CK(hipMalloc(&dC, (size_t)M * N * 2)); CK(hipMalloc(&db, N * 4));
    CK(hipMemcpy(dX, hX.data(), hX.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(db, 0, N * 4));
    Args a{};
    a.P = dX; a.Q = dW; a.C = dC; a.bias = db; a.ldp = K; a.ldq = K; a.ldc = N; a.Pext = M; a.Qext = N; a.K = K;
    a.nTp = (M + 255) / 256; a.nTq = (N + 255) / 256; a.relu = 1; a.kt_per_slab = (K + 63) / 64;
    const int grid = a.nTp * a.nTq;
    CK(hipMalloc(&dws, (size_t)grid * 8 * 36 * 4));
    CK(hipMemset(dws, 0, (size_t)grid * 8 * 36 * 4));
    if (VAR & 64) a.colsum_ws = dws;
    auto run = [&] { k_gemm256<false, false, EPI_FWD, true, VAR, 8><<<grid, kThreads>>>(a); };
    run();
    CK(hipDeviceSynchronize());
    Timer tm;
    double best = 1e30, sum = 0;
    for (int r = 0; r < 5; ++r) { const double us = tm.us(run, 20); best = us < best ? us : best; sum += us; }
    printf("var %3d %-34s M=%d N=%d K=%d grid %d: mean %7.1f us  best %7.1f us  %6.3f PF/s\n", VAR, what, M, N, K, grid, sum / 5, best,
           2.0 * M * N * K / best / 1e9);
    if (VAR & 64) {
        std::vector<float> h((size_t)grid * 8 * 36);
        CK(hipMemcpy(h.data(), dws, h.size() * 4, hipMemcpyDeviceToHost));
        const int T = (K + 63) / 64;
        for (int half = 0; half < 2; ++half) {
            double acc[32] = {};
            double cyc = 0, rt = 0;
            for (int b = 0; b < grid; ++b)
                for (int w = half * 4; w < half * 4 + 4; ++w) { cyc += h[((size_t)b * 8 + w) * 36 + 32]; rt += h[((size_t)b * 8 + w) * 36 + 33]; }
            for (int b = 0; b < grid; ++b)
                for (int w = half * 4; w < half * 4 + 4; ++w)
                    for (int i = 0; i < 32; ++i) acc[i] += h[((size_t)b * 8 + w) * 36 + i];
            printf("    waves %d-%d, cycles per phase (issue+waits | barrier 1 | MFMAs | barrier 2):", half * 4, half * 4 + 3);
            double tot = 0;
            for (int ph = 0; ph < 8; ++ph) {
                if (ph == 4) printf("\n            odd K-tiles:");
                printf("  [");
                for (int sg = 0; sg < 4; ++sg) { const double v = acc[ph * 4 + sg] / (grid * 4.0) / (T / 2); tot += v; printf(" %5.0f", v); }
                printf(" ]");
            }
            printf("  two K-tiles %6.0f  clock %.2f GHz\n", tot, cyc / rt * 0.1);
        }
    }
    CK(hipFree(dX)); CK(hipFree(dW)); CK(hipFree(dC)); CK(hipFree(db)); CK(hipFree(dws));
}
static void ablations(int B) {
    for (int rep = 0; rep < 2; ++rep) {
        const int K = rep == 0 ? 2080 : 8320;
        time_fwd_kc<0>(B, 1024, K, "as shipped");
        time_fwd_kc<1>(B, 1024, K, "no stagger");
        time_fwd_kc<2>(B, 1024, K, "no priorities");
        time_fwd_kc<8>(B, 1024, K, "no staging past K-tile 1");
        time_fwd_kc<16>(B, 1024, K, "no fragment reads past K-tile 0");
        time_fwd_kc<24>(B, 1024, K, "neither (MFMAs + barriers)");
        time_fwd_kc<32>(B, 1024, K, "staging from K-tile 0's addresses");
        time_fwd_kc<128>(B, 1024, K, "reads dealt 6/6/6/6 (wrong results)");
        time_fwd_kc<16384>(B, 1024, K, "reads dealt 8/4/8/4 (wrong results)");
        time_fwd_kc<65536>(B, 1024, K, "reads 8/4/8/4 (correct: bit-identical)");
        time_fwd_kc<256>(B, 1024, K, "four K-tiles per loop trip");
        time_fwd_kc<512>(B, 1024, K, "eight K-tiles per loop trip");
        time_fwd_kc<8192>(B, 1024, K, "no output stores");
        time_fwd_kc<0>(B, 1024, K, "as shipped (again)");
        time_fwd_kc<64>(B, 1024, K, "stamped");
        time_fwd_kc<72>(B, 1024, K, "stamped, no staging past K-tile 1");
        time_fwd_kc<80>(B, 1024, K, "stamped, no fragment reads");
        time_fwd_kc<88>(B, 1024, K, "stamped, neither");
        time_fwd_kc<96>(B, 1024, K, "stamped, staging from K-tile 0");
    }
    time_fwd_kc<0>(4096, 4096, 4096, "4096^3");
    time_fwd_kc<0>(8192, 8192, 8192, "8192^3");
}

// The round-5 read schedule of the 256 x 256 body against the shipped one (VAR & 65536 selects it): the same MFMAs in the same order, so every
// output bit must agree -- forward (both operand layouts), input gradient with mask and column sums, weight gradient slabs; ragged
// shapes and K tails included; several launches each (a race between the early fragment reads and the staging DMA would show as a
// difference that comes and goes).
template <bool PT, bool QT, int EPI>
static int compare_old_new(int M, int N, int K, int S, const char* what) {
    // forward / dgrad: C[M, N] 16-bit = P[M, K] . Q (QT ? [K, N] : [N, K]);  wgrad (PT && QT): C[S][Kin = N?]...  (sizes as the tests above)
    const int Pext = PT ? N : M;            // wgrad: P = x [M, Kin] read as [k = m][p = kin]: Pext = Kin (passed as N), Qext = Nout (passed as K)
    (void)Pext;
    std::vector<uint16_t> hP, hQ, hH;
    size_t nC;
    Args a{};
    uint16_t *dP, *dQ, *dH = nullptr;
    float* dws = nullptr;
    if (!(PT && QT)) {
        hP.resize((size_t)M * K); hQ.resize((size_t)N * K);
        for (auto& x : hP) x = h_f2bf(frand());
        for (auto& x : hQ) x = h_f2bf(frand() * 0.05f);
        a.ldp = K; a.ldq = QT ? N : K; a.ldc = N; a.Pext = M; a.Qext = N; a.K = K;
        a.nTp = (M + 255) / 256; a.nTq = (N + 255) / 256; a.kt_per_slab = (K + 63) / 64; a.relu = 1;
        nC = (size_t)M * N * 2;
    } else {
        // dW[Kin = N, Nout = K] = x[M, Kin]^T . dy[M, Nout], reduction over M in S slabs
        hP.resize((size_t)M * N); hQ.resize((size_t)M * K);
        for (auto& x : hP) x = h_f2bf(frand());
        for (auto& x : hQ) x = h_f2bf(frand() * 0.05f);
        a.ldp = N; a.ldq = K; a.ldc = K; a.Pext = N; a.Qext = K; a.K = M;
        a.nTp = (N + 255) / 256; a.nTq = (K + 255) / 256;
        const int Tt = (M + 63) / 64;
        a.kt_per_slab = (Tt + S - 1) / S; a.slab_stride = (int64_t)N * K;
        nC = (size_t)S * N * K * 4;
    }
    CK(hipMalloc(&dP, hP.size() * 2)); CK(hipMalloc(&dQ, hQ.size() * 2));
    CK(hipMemcpy(dP, hP.data(), hP.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dQ, hQ.data(), hQ.size() * 2, hipMemcpyHostToDevice));
    a.P = dP; a.Q = dQ;
    if (EPI == EPI_DGRAD) {
        hH.resize((size_t)M * N);
        for (auto& x : hH) { float v = frand(); x = h_f2bf(v > 0.2f ? v : 0.f); }
        CK(hipMalloc(&dH, hH.size() * 2));
        CK(hipMemcpy(dH, hH.data(), hH.size() * 2, hipMemcpyHostToDevice));
        CK(hipMalloc(&dws, (size_t)a.nTp * N * 4));
        a.H = dH; a.colsum_ws = dws;
    }
    float* db = nullptr;
    if (EPI == EPI_FWD) { CK(hipMalloc(&db, N * 4)); CK(hipMemset(db, 0, N * 4)); a.bias = db; }
    void *c0, *c1;
    CK(hipMalloc(&c0, nC)); CK(hipMalloc(&c1, nC));
    const int grid = a.nTp * a.nTq * ((PT && QT) ? S : 1);
    std::vector<unsigned char> h0(nC), h1(nC);
    int bad = 0;
    for (int rep = 0; rep < 6; ++rep) {
        CK(hipMemset(c0, 0xEE, nC)); CK(hipMemset(c1, 0x11, nC));
        a.C = c0; k_gemm256<PT, QT, EPI, false, 0, 8><<<grid, kThreads>>>(a);
        a.C = c1; k_gemm256<PT, QT, EPI, false, 65536, 8><<<grid, kThreads>>>(a);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h0.data(), c0, nC, hipMemcpyDeviceToHost));
        CK(hipMemcpy(h1.data(), c1, nC, hipMemcpyDeviceToHost));
        if (memcmp(h0.data(), h1.data(), nC)) ++bad;
    }
    printf("shipped vs 8/4/8/4 schedule, %-22s M=%d N=%d K=%d S=%d grid %d: %s\n", what, M, N, K, S, grid, bad ? "DIFFERENT" : "bit-identical (6 launches)");
    CK(hipFree(dP)); CK(hipFree(dQ)); CK(hipFree(c0)); CK(hipFree(c1));
    if (dH) CK(hipFree(dH));
    if (dws) CK(hipFree(dws));
    if (db) CK(hipFree(db));
    return bad;
}
static int compare_all(int B) {
    int bad = 0;
    bad += compare_old_new<false, false, EPI_FWD>(B, 1024, 2080, 1, "forward (W transposed)");
    bad += compare_old_new<false, true, EPI_FWD>(B, 1024, 2080, 1, "forward (W as stored)");
    bad += compare_old_new<false, false, EPI_FWD>(700, 264, 3120, 1, "forward, ragged");
    bad += compare_old_new<false, true, EPI_FWD>(300, 200, 96, 1, "forward, short K");
    bad += compare_old_new<false, false, EPI_FWD>(512, 256, 72, 1, "forward, K tail 8");
    bad += compare_old_new<false, false, EPI_FWD>(256, 256, 32, 1, "forward, half tile");
    bad += compare_old_new<false, false, EPI_DGRAD>(B, 2080, 1024, 1, "input gradient");
    bad += compare_old_new<false, false, EPI_DGRAD>(300, 200, 128, 1, "input gradient, ragged");
    bad += compare_old_new<true, true, EPI_F32>(B, 2080, 1024, 7, "weight gradient");
    bad += compare_old_new<true, true, EPI_F32>(1000, 3120, 264, 3, "weight gradient, ragged");
    bad += compare_old_new<true, true, EPI_F32>(96, 264, 72, 1, "weight gradient, short");
    return bad;
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 16384;
    if (argc > 2 && !strcmp(argv[2], "ablate")) { ablations(B); return 0; }
    if (argc > 2 && !strcmp(argv[2], "compare")) { return compare_all(B) ? 1 : 0; }
    if (argc > 2 && !strcmp(argv[2], "prof")) {        // one long dispatch for counter passes (clock, LDS conflicts)
        test_fwd(B, 1024, 33280);
        test_dgrad<8>(B, 1024, 33280, false);
        test_dgrad<4>(B, 1024, 33280, false);
        return 0;
    }
    // small odd shapes first (tails, masks), then the Wide&Deep MLP shapes (2080-1024-512-256-128)
    test_fwd(300, 200, 96);
    test_fwd(512, 256, 64);
    test_fwd(256, 256, 32);
    test_dgrad(300, 200, 128, true);
    test_wgrad(256, 200, 136, 2);
    test_wgrad(128, 256, 256, 1);
    test_wgrad(96, 264, 72, 1);
    test_wgrad(480, 264, 72, 3);
    test_fwd(100, 520, 160);
    test_fwd(700, 264, 3120);
    test_fwd(512, 256, 72);
    test_dgrad(300, 3120, 136, false);
    test_dgrad(300, 200, 72, true);
    test_wgrad(1000, 3120, 264, 3);
    test_wgrad(50, 264, 72, 1);
    // the 128 x 256 configuration: odd shapes, tails, then the narrow layers
    test_fwd<0, 4>(300, 200, 96);
    test_fwd<0, 4>(130, 520, 160);
    test_fwd<0, 4>(700, 264, 3120);
    test_fwd<0, 4>(512, 256, 72);
    test_fwd<0, 4>(128, 256, 64);
    test_dgrad<4>(300, 200, 128, true);
    test_dgrad<4>(300, 3120, 136, false);
    test_dgrad<4>(1000, 200, 72, true);
    test_wgrad<4>(256, 200, 136, 2);
    test_wgrad<4>(1000, 3120, 264, 3);
    test_wgrad<4>(50, 264, 72, 1);
    test_wgrad<4>(96, 136, 72, 1);
    test_fwd<0, 4>(B, 512, 1024);
    test_fwd<0, 4>(B, 256, 512);
    test_fwd<0, 4>(B, 128, 256);
    test_dgrad<4>(B, 1024, 512, true);
    test_dgrad<4>(B, 512, 256, true);
    test_dgrad<4>(B, 256, 128, true);
    test_wgrad<4>(B, 1024, 512, 16);
    test_wgrad<4>(B, 512, 256, 32);
    test_wgrad<4>(B, 256, 128, 64);
    test_fwd(B, 1024, 2080);
    test_fwd(B, 512, 1024);
    test_fwd(B, 256, 512);
    test_fwd(B, 128, 256);
    test_dgrad(B, 2080, 1024, false);
    test_dgrad(B, 1024, 512, true);
    test_dgrad(B, 512, 256, true);
    test_dgrad(B, 256, 128, true);
    test_wgrad(B, 2080, 1024, 7);
    test_wgrad(B, 2080, 1024, 8);
    test_wgrad(B, 1024, 512, 16);
    test_wgrad(B, 1024, 512, 32);
    test_wgrad(B, 512, 256, 32);
    test_wgrad(B, 256, 128, 32);
    // main-loop experiments: slope per K-tile, stagger / priority ablations
    test_fwd(B, 1024, 8320);
    return 0;
}
