// mrec_dense.hip -- DenseLayer (models/wide_deep/src/wide_and_deep.py:113-133; models/deep_and_cross/src/
// deep_and_cross.py:94-114) on the gfx950 matrix cores: the C ABI over the MFMA GEMM body of mrec_gemm.h.
//
//   forward     y  = act(x . W + b)                     MatMul + BiasAdd + ReLU, 16-bit operands, fp32 accumulate
//   bprop/input dx = (dy . W^T) [masked by h > 0]       MatMul bprop + the ReLU bprop of the layer below, and
//               db[k] = sum_m dx[m, k]                  that layer's BiasAdd bprop (column sums, fixed order)
//   bprop/weight dW = x^T . dy                          reduction over the batch, split in S slabs of fp32
//                                                       partial sums (never rounded to 16 bits)
// Everything is reproducible run to run: no atomics, fixed summation orders.
#include "mrec_common.h"
#include "mrec_gemm.h"

namespace {

using mgemm::Args;

inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

int g_cu_count = 0;
int cu_count() {
    if (g_cu_count == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
            g_cu_count = n;
        else
            g_cu_count = 256;
    }
    return g_cu_count;
}

// Tile configuration: the 256 x 256 tile when its workgroups fill the chip, else the 128 x 256 one (twice the
// workgroups, half the time per K-tile: the narrow layers are bound by the latency of one workgroup's K loop).
inline int pick_mr(int64_t blocks_256) { return blocks_256 * 4 >= (int64_t)cu_count() * 3 ? 8 : 4; }
int bwd_input_mr(int64_t M, int32_t K);
int bwd_mr(int64_t M, int32_t K, int32_t N);

// db[k] = sum over tile rows of the per-tile column sums.  Thread (c, q) of a block adds the tile rows t = q, q + 4, ...
// of column c with all its loads in flight (a serial loop over the tile rows is one dependent L2 round trip per row:
// 17 us for 64 rows); the four partial sums are then added in q order: a fixed order, reproducible run to run.
__global__ __launch_bounds__(256) void k_colsum_tiles(const float* __restrict__ ws, int nT, int K, float* __restrict__ db) {
    __shared__ float part[4][64];
    const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int k = blockIdx.x * 64 + c;
    float s = 0.f;
    if (k < K) {
        for (int t0 = q; t0 < nT; t0 += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = t0 + 4 * u;
                v[u] = t < nT ? ws[(int64_t)t * K + k] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
    }
    part[q][c] = s;
    __syncthreads();
    if (q == 0 && k < K) db[k] = ((part[0][c] + part[1][c]) + part[2][c]) + part[3][c];
}

// out[e] = sum_s slabs[s * len + e] in slab order (float4 lanes)
__global__ __launch_bounds__(256) void k_sum_slabs(const float4* __restrict__ slabs, int S, int64_t len4, float4* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < len4; i += (int64_t)gridDim.x * 256) {
        float4 a = slabs[i];
        for (int s = 1; s < S; ++s) {
            const float4 u = slabs[(int64_t)s * len4 + i];
            a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
        }
        out[i] = a;
    }
}

template <bool F16, bool WT = false>      // WT: w is the TRANSPOSED weight [N, K] (both operands K-contiguous)
int fwd_impl(const uint16_t* x, int64_t ldx, const uint16_t* w, const float* bias, int64_t M, int32_t K, int32_t N, int relu,
             uint16_t* y, int64_t ldy, const mrec_dropout_t* drop, void* stream) {
    if (M < 0 || K <= 0 || N <= 0 || ldx < K || ldy < N) return MREC_EINVAL;
    DropArgs da;
    if (!drop_from(drop, N, &da)) return MREC_EINVAL;
    if (M == 0) return MREC_OK;
    if (!x || !w || !y) return MREC_EINVAL;
    if (K % 8 || N % 8 || ldx % 8 || ldy % 4 || !al16(x) || !al16(w) || (((uintptr_t)y) & 7)) return MREC_EUNSUPPORTED;
    if (M * ldx * 2 >= (int64_t(1) << 31) || (int64_t)K * N * 2 >= (int64_t(1) << 31) || M > (int64_t(1) << 30)) return MREC_EUNSUPPORTED;
    // (128 x 256 tiles for layer 0 of the reference's net too -- 512 workgroups instead of exactly one round of 256, to lose less
    // to the plan kernels that take CUs beside it -- was measured: 0.674 vs 0.654 ms/step)
    const int mr = pick_mr(mrec_cdiv(M, 256) * mrec_cdiv(N, 256));
    Args a{};
    a.P = x; a.Q = w; a.C = y; a.bias = bias;
    a.ldp = ldx; a.ldq = WT ? K : N; a.ldc = ldy;
    a.Pext = (int)M; a.Qext = N; a.K = K;
    a.nTp = (int)mrec_cdiv(M, mr * 32); a.nTq = (int)mrec_cdiv(N, 256);
    a.kt_per_slab = (K + 63) / 64;
    a.relu = relu;
    a.drop = da;
    if (mr == 8) mgemm::k_gemm256<false, !WT, mgemm::EPI_FWD, F16, 0, 8><<<a.nTp * a.nTq, mgemm::kThreads, 0, (hipStream_t)stream>>>(a);
    else mgemm::k_gemm256<false, !WT, mgemm::EPI_FWD, F16, 0, 4><<<a.nTp * a.nTq, mgemm::kThreads, 0, (hipStream_t)stream>>>(a);
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}

// argument checks + Args of the two bprops (shared by the separate and the fused entry points)
int bwd_input_args(const uint16_t* dy, int64_t lddy, const uint16_t* w, const uint16_t* h, int64_t M, int32_t K, int32_t N,
                   uint16_t* dx, int64_t lddx, float* db, void* ws, size_t ws_bytes, int mr, const mrec_dropout_t* drop, Args* out) {
    if (M <= 0 || K <= 0 || N <= 0 || lddy < N || lddx < K) return MREC_EINVAL;
    DropArgs da;
    if (!drop_from(drop, K, &da)) return MREC_EINVAL;
    if (!dy || !w || !dx) return MREC_EINVAL;
    if (N % 8 || K % 4 || lddy % 8 || lddx % 4 || !al16(dy) || !al16(w) || (((uintptr_t)dx) & 7) || (h && (((uintptr_t)h) & 7)))
        return MREC_EUNSUPPORTED;
    if (M * lddy * 2 >= (int64_t(1) << 31) || (int64_t)K * N * 2 >= (int64_t(1) << 31) || M > (int64_t(1) << 30)) return MREC_EUNSUPPORTED;
    const int nTp = (int)mrec_cdiv(M, mr * 32);
    float* part = nullptr;
    if (db || ws) {             // ws without db: leave the per-tile-row column sums there for the caller (the dense Adam adds them)
        if (!ws || ws_bytes < (size_t)nTp * K * 4) return MREC_EWORKSPACE;
        part = (float*)ws;
    }
    Args a{};
    a.P = dy; a.Q = w; a.C = dx; a.H = h; a.colsum_ws = part;
    a.ldp = lddy; a.ldq = N; a.ldc = lddx;
    a.Pext = (int)M; a.Qext = K; a.K = N;
    a.nTp = nTp; a.nTq = (int)mrec_cdiv(K, 256);
    a.kt_per_slab = (N + 63) / 64;
    a.drop = da;
    *out = a;
    return MREC_OK;
}

int bwd_weight_args(const uint16_t* x, int64_t ldx, const uint16_t* dy, int64_t lddy, int64_t M, int32_t K, int32_t N, int32_t S,
                    float* dw, int mr, Args* out) {
    if (M <= 0 || K <= 0 || N <= 0 || S <= 0 || ldx < K || lddy < N) return MREC_EINVAL;
    if (!dw || !x || !dy) return MREC_EINVAL;
    if (K % 8 || N % 8 || ldx % 8 || lddy % 8 || !al16(x) || !al16(dy) || !al16(dw)) return MREC_EUNSUPPORTED;
    if (M * ldx * 2 >= (int64_t(1) << 31) || M * lddy * 2 >= (int64_t(1) << 31)) return MREC_EUNSUPPORTED;
    Args a{};
    a.P = x; a.Q = dy; a.C = dw;
    a.ldp = ldx; a.ldq = lddy; a.ldc = N;
    a.Pext = K; a.Qext = N; a.K = (int)M;
    a.nTp = (int)mrec_cdiv(K, mr * 32); a.nTq = (int)mrec_cdiv(N, 256);
    const int Ttot = (int)mrec_cdiv(M, 64);
    a.kt_per_slab = (Ttot + S - 1) / S;
    a.slab_stride = (int64_t)K * N;
    *out = a;
    return MREC_OK;
}

template <bool F16>
int bwd_input_impl(const uint16_t* dy, int64_t lddy, const uint16_t* w, const uint16_t* h, int64_t M, int32_t K, int32_t N,
                   uint16_t* dx, int64_t lddx, float* db, void* ws, size_t ws_bytes, const mrec_dropout_t* drop, void* stream) {
    if (M == 0 && K > 0) {
        if (db) MREC_HIP_CHECK(hipMemsetAsync(db, 0, (size_t)K * 4, (hipStream_t)stream));
        return MREC_OK;
    }
    Args a;
    const int mr = bwd_input_mr(M, K);
    const int rc = bwd_input_args(dy, lddy, w, h, M, K, N, dx, lddx, db, ws, ws_bytes, mr, drop, &a);
    if (rc != MREC_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (mr == 8) mgemm::k_gemm256<false, false, mgemm::EPI_DGRAD, F16, 0, 8><<<a.nTp * a.nTq, mgemm::kThreads, 0, st>>>(a);
    else mgemm::k_gemm256<false, false, mgemm::EPI_DGRAD, F16, 0, 4><<<a.nTp * a.nTq, mgemm::kThreads, 0, st>>>(a);
    if (db) k_colsum_tiles<<<(unsigned)mrec_cdiv(K, 64), 256, 0, st>>>(a.colsum_ws, a.nTp, K, db);
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}

// the configurations of a layer's two bprops, each by the workgroups it would have with 256 x 256 tiles
int weight_slabs_mr(int64_t M, int32_t K, int32_t N, int mr);
int bwd_mr(int64_t M, int32_t K, int32_t N) {            // weight gradient
    return pick_mr(mrec_cdiv(K, 256) * mrec_cdiv(N, 256) * weight_slabs_mr(M, K, N, 8));
}
int bwd_input_mr(int64_t M, int32_t K) { return pick_mr(mrec_cdiv(M, 256) * mrec_cdiv(K, 256)); }

// slabs with which this weight gradient ALONE would fill the chip: decides its tile configuration
int weight_slabs_mr(int64_t M, int32_t K, int32_t N, int mr) {
    const int64_t tiles = mrec_cdiv(K, mr * 32) * mrec_cdiv(N, 256);
    const int64_t Ttot = mrec_cdiv(M, 64);
    int64_t S = cu_count() / tiles;
    int64_t cap = ((int64_t)16 << 20) / ((int64_t)K * N * 4);
    if (cap < 16) cap = 16;
    if (cap > 64) cap = 64;
    if (S > cap) S = cap;
    if (S > Ttot) S = Ttot;
    if (S < 1) S = 1;
    return (int)S;
}
// The slab count proposed to the caller.  The weight gradient runs in ONE launch with the layer's input gradient (>= one
// workgroup per CU of its own), so it needs about half a chip of workgroups, not a full one: fewer, longer workgroups pay the
// prologue / epilogue (a 128-KB fp32 tile each) less often and leave half the slab bytes for the optimizer to read.  A layer whose
// weight gradient is a full round of 256 x 256 tiles (layer 0 of the reference's net: 36 tiles) keeps ~0.85 of a round, so
// that the input-gradient workgroups dispatched behind it pack into the CUs it leaves free.  Measured on the reference's net at
// batch 16384 (a sweep over per-layer counts): slabs 7 / 16 / 32 / 64 -> 6 / 8 / 16 / 32 took the step from 0.695 to 0.652 ms.
int weight_slabs(int64_t M, int32_t K, int32_t N) {
    const int mr = bwd_mr(M, K, N);
    const int64_t tiles = mrec_cdiv(K, mr * 32) * mrec_cdiv(N, 256);
    const int64_t Ttot = mrec_cdiv(M, 64);
    int64_t S = tiles >= 32 ? (int64_t)(0.85 * cu_count() / tiles + 0.5) : cu_count() / (2 * tiles);
    int64_t cap = ((int64_t)8 << 20) / ((int64_t)K * N * 4);        // slabs are written once and read once by the optimizer
    if (cap < 8) cap = 8;
    if (cap > 32) cap = 32;
    if (S > cap) S = cap;
    if (S > Ttot) S = Ttot;
    if (S < 1) S = 1;
    return (int)S;
}

template <bool F16>
int bwd_weight_impl(const uint16_t* x, int64_t ldx, const uint16_t* dy, int64_t lddy, int64_t M, int32_t K, int32_t N, int32_t S,
                    float* dw, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (M == 0 && K > 0 && N > 0 && S > 0 && dw) {
        MREC_HIP_CHECK(hipMemsetAsync(dw, 0, (size_t)S * K * N * 4, st));
        return MREC_OK;
    }
    Args a;
    const int mr = bwd_mr(M, K, N);
    const int rc = bwd_weight_args(x, ldx, dy, lddy, M, K, N, S, dw, mr, &a);
    if (rc != MREC_OK) return rc;
    if (mr == 8) mgemm::k_gemm256<true, true, mgemm::EPI_F32, F16, 0, 8><<<a.nTp * a.nTq * S, mgemm::kThreads, 0, st>>>(a);
    else mgemm::k_gemm256<true, true, mgemm::EPI_F32, F16, 0, 4><<<a.nTp * a.nTq * S, mgemm::kThreads, 0, st>>>(a);
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}

template <bool F16>
int bwd_impl(const uint16_t* dy, int64_t lddy, const uint16_t* w, const uint16_t* h, const uint16_t* x, int64_t ldx, int64_t M,
             int32_t K, int32_t N, uint16_t* dx, int64_t lddx, void* db_ws, size_t db_ws_bytes, int32_t S, float* dw,
             const mrec_dropout_t* drop, const mrec_wgrad_t* extra, int32_t n_extra, void* stream) {
    Args ad, aw;
    if (n_extra < 0 || n_extra > 2 || (n_extra && !extra)) return MREC_EINVAL;
    const int mrd = bwd_input_mr(M, K), mrw = bwd_mr(M, K, N);
    int rc = bwd_input_args(dy, lddy, w, h, M, K, N, dx, lddx, nullptr, db_ws, db_ws_bytes, mrd, drop, &ad);
    if (rc != MREC_OK) return rc;
    rc = bwd_weight_args(x, ldx, dy, lddy, M, K, N, S, dw, mrw, &aw);
    if (rc != MREC_OK) return rc;
    const int nd = ad.nTp * ad.nTq, nw = aw.nTp * aw.nTq * S;
    // other layers' weight gradients riding this launch: the same batch, tiled in this launch's weight-gradient configuration
    mgemm::ExtraW ex{};
    for (int e = 0; e < n_extra; ++e) {
        rc = bwd_weight_args(extra[e].x, extra[e].ldx, extra[e].dy, extra[e].lddy, M, extra[e].K, extra[e].N, extra[e].S, extra[e].dw_slabs,
                             mrw, &ex.a[e]);
        if (rc != MREC_OK) return rc;
        ex.n[e] = ex.a[e].nTp * ex.a[e].nTq * extra[e].S;
    }
    const int ntot = nd + nw + ex.n[0] + ex.n[1];
    // longer workgroups first (time per K-tile goes with the tile height)
    const int wfirst = (int64_t)aw.kt_per_slab * mrw >= (int64_t)ad.kt_per_slab * mrd;
    hipStream_t st = (hipStream_t)stream;
    const int n1 = wfirst ? nw : nd;
    if (mrd == 8 && mrw == 8) mgemm::k_gemm256_bwd<F16, 8, 8><<<ntot, mgemm::kThreads, 0, st>>>(ad, aw, n1, wfirst, ex);
    else if (mrd == 8) mgemm::k_gemm256_bwd<F16, 8, 4><<<ntot, mgemm::kThreads, 0, st>>>(ad, aw, n1, wfirst, ex);
    else if (mrw == 8) mgemm::k_gemm256_bwd<F16, 4, 8><<<ntot, mgemm::kThreads, 0, st>>>(ad, aw, n1, wfirst, ex);
    else mgemm::k_gemm256_bwd<F16, 4, 4><<<ntot, mgemm::kThreads, 0, st>>>(ad, aw, n1, wfirst, ex);
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}

}  // namespace

MREC_API int mrec_dense_fwd_bf16(const uint16_t* x, int64_t ldx, const uint16_t* w, const float* bias, int64_t M, int32_t K,
                                 int32_t N, int relu, uint16_t* y, int64_t ldy, const mrec_dropout_t* drop_next, void* stream) {
    return fwd_impl<false>(x, ldx, w, bias, M, K, N, relu, y, ldy, drop_next, stream);
}
MREC_API int mrec_dense_fwd_f16(const uint16_t* x, int64_t ldx, const uint16_t* w, const float* bias, int64_t M, int32_t K,
                                int32_t N, int relu, uint16_t* y, int64_t ldy, const mrec_dropout_t* drop_next, void* stream) {
    return fwd_impl<true>(x, ldx, w, bias, M, K, N, relu, y, ldy, drop_next, stream);
}

MREC_API int mrec_dense_fwd_wt_bf16(const uint16_t* x, int64_t ldx, const uint16_t* wt, const float* bias, int64_t M, int32_t K,
                                    int32_t N, int relu, uint16_t* y, int64_t ldy, const mrec_dropout_t* drop_next, void* stream) {
    return fwd_impl<false, true>(x, ldx, wt, bias, M, K, N, relu, y, ldy, drop_next, stream);
}
MREC_API int mrec_dense_fwd_wt_f16(const uint16_t* x, int64_t ldx, const uint16_t* wt, const float* bias, int64_t M, int32_t K,
                                   int32_t N, int relu, uint16_t* y, int64_t ldy, const mrec_dropout_t* drop_next, void* stream) {
    return fwd_impl<true, true>(x, ldx, wt, bias, M, K, N, relu, y, ldy, drop_next, stream);
}

MREC_API int mrec_dense_bwd_input_workspace_bytes(int64_t M, int32_t K, size_t* out) {
    if (!out || M < 0 || K <= 0) return MREC_EINVAL;
    *out = mrec_align_up((size_t)mrec_cdiv(M > 0 ? M : 1, 128) * K * 4, 256);        // enough for either tile configuration
    return MREC_OK;
}
/* rows of the bias-gradient slabs mrec_dense_bwd_* (fused = 1) / mrec_dense_bwd_input_* (fused = 0) leave in ws */
MREC_API int mrec_dense_bwd_bias_slabs(int64_t M, int32_t K, int32_t N, int fused, int32_t* rows_out) {
    if (!rows_out || M < 0 || K <= 0 || N <= 0) return MREC_EINVAL;
    (void)fused; (void)N;                      // both entry points tile the input gradient the same way
    const int mr = bwd_input_mr(M, K);
    *rows_out = (int32_t)mrec_cdiv(M > 0 ? M : 1, mr * 32);
    return MREC_OK;
}
MREC_API int mrec_dense_bwd_input_bf16(const uint16_t* dy, int64_t lddy, const uint16_t* w, const uint16_t* h, int64_t M,
                                       int32_t K, int32_t N, uint16_t* dx, int64_t lddx, float* db, void* ws, size_t ws_bytes,
                                       const mrec_dropout_t* drop_in, void* stream) {
    return bwd_input_impl<false>(dy, lddy, w, h, M, K, N, dx, lddx, db, ws, ws_bytes, drop_in, stream);
}
MREC_API int mrec_dense_bwd_input_f16(const uint16_t* dy, int64_t lddy, const uint16_t* w, const uint16_t* h, int64_t M,
                                      int32_t K, int32_t N, uint16_t* dx, int64_t lddx, float* db, void* ws, size_t ws_bytes,
                                      const mrec_dropout_t* drop_in, void* stream) {
    return bwd_input_impl<true>(dy, lddy, w, h, M, K, N, dx, lddx, db, ws, ws_bytes, drop_in, stream);
}

MREC_API int mrec_dense_bwd_weight_slabs(int64_t M, int32_t K, int32_t N, int32_t* S_out) {
    if (!S_out || M < 0 || K <= 0 || N <= 0) return MREC_EINVAL;
    *S_out = weight_slabs(M, K, N);
    return MREC_OK;
}
MREC_API int mrec_dense_bwd_weight_bf16(const uint16_t* x, int64_t ldx, const uint16_t* dy, int64_t lddy, int64_t M, int32_t K,
                                        int32_t N, int32_t S, float* dw_slabs, void* stream) {
    return bwd_weight_impl<false>(x, ldx, dy, lddy, M, K, N, S, dw_slabs, stream);
}
MREC_API int mrec_dense_bwd_weight_f16(const uint16_t* x, int64_t ldx, const uint16_t* dy, int64_t lddy, int64_t M, int32_t K,
                                       int32_t N, int32_t S, float* dw_slabs, void* stream) {
    return bwd_weight_impl<true>(x, ldx, dy, lddy, M, K, N, S, dw_slabs, stream);
}

MREC_API int mrec_dense_sum_slabs_f32(const float* slabs, int32_t S, int64_t len, float* out, void* stream) {
    if (S <= 0 || len < 0) return MREC_EINVAL;
    if (len == 0) return MREC_OK;
    if (!slabs || !out) return MREC_EINVAL;
    if (len % 4 || !al16(slabs) || !al16(out)) return MREC_EUNSUPPORTED;
    const int64_t len4 = len / 4;
    const unsigned g = (unsigned)(mrec_cdiv(len4, 256) < 2048 ? mrec_cdiv(len4, 256) : 2048);
    k_sum_slabs<<<g, 256, 0, (hipStream_t)stream>>>((const float4*)slabs, S, len4, (float4*)out);
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}

/* both bprops of one layer in one launch (see include/mrec.h) */
MREC_API int mrec_dense_bwd_bf16(const uint16_t* dy, int64_t lddy, const uint16_t* w, const uint16_t* h, const uint16_t* x,
                                 int64_t ldx, int64_t M, int32_t K, int32_t N, uint16_t* dx, int64_t lddx, void* db_slabs,
                                 size_t db_slabs_bytes, int32_t S, float* dw_slabs, const mrec_dropout_t* drop_in, const mrec_wgrad_t* extra,
                                 int32_t n_extra, void* stream) {
    return bwd_impl<false>(dy, lddy, w, h, x, ldx, M, K, N, dx, lddx, db_slabs, db_slabs_bytes, S, dw_slabs, drop_in, extra, n_extra, stream);
}
MREC_API int mrec_dense_bwd_f16(const uint16_t* dy, int64_t lddy, const uint16_t* w, const uint16_t* h, const uint16_t* x,
                                int64_t ldx, int64_t M, int32_t K, int32_t N, uint16_t* dx, int64_t lddx, void* db_slabs,
                                size_t db_slabs_bytes, int32_t S, float* dw_slabs, const mrec_dropout_t* drop_in, const mrec_wgrad_t* extra,
                                int32_t n_extra, void* stream) {
    return bwd_impl<true>(dy, lddy, w, h, x, ldx, M, K, N, dx, lddx, db_slabs, db_slabs_bytes, S, dw_slabs, drop_in, extra, n_extra, stream);
}
