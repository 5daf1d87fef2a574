// mrec_gemm_f32.hip -- DenseLayer in fp32 (convert_dtype=False: Deep&Cross, models/deep_and_cross/src/deep_and_cross.py:94-114,
// 293-309) and its two bprops on the fp32-input matrix instruction of gfx950, v_mfma_f32_32x32x2_f32.
//
// The fp32 MFMA runs at the fp32 vector rate (157 TFLOP/s, 1/16 of the 16-bit forms) and is bit-for-bit a k-ordered chain of
// fmaf's: one rounding per product, no wider accumulation.  At that rate nothing else on the CU is near its limit, so the
// structure is the plain one: a 128 x 128 output tile per 256-thread workgroup, four waves as 2 x 2 with 64 x 64 (2 x 2
// accumulators of 32 x 32) each, a 32-deep K-tile staged through registers into LDS in REDUCTION-MAJOR order (rows = k, 132
// floats apart: conflict-free ds_read_b32 fragments: the instruction wants A[i = lane & 31][k = lane >> 5]), the next
// K-tile's global loads in flight while the current one is multiplied.
//
// Operands are either reduction-contiguous (X[r, k]: a row of the operand is a row of the matrix: transposed on its way into
// LDS, four ds_write_b32 per 16-byte load) or reduction-strided (Y[k, c]: written as loaded):
//   forward   y  = x . W        A = x  [M, K] contiguous      B = W  [K, N] strided
//   dgrad     dx = dy . W^T     A = dy [M, N] contiguous      B = W  [K, N] contiguous (its rows ARE the outputs)
//   wgrad     dW = x^T . dy     A = x  [M, K] strided         B = dy [M, N] strided, reduction over the batch, split in slabs
// Epilogues: bias + ReLU; mask by the activation below > 0 + per-tile-row column sums (the layer below's BiasAdd bprop);
// plain fp32 slabs.  Rows need only 8-byte alignment (the DCN input is 39 x 30 = 1170 floats wide): VEC = 4, 2 or 1 floats per
// load is picked per operand.
#include "mrec_common.h"

namespace gf32 {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32, LD = 132;      // tile; LDS row stride in floats (528 B: 16-byte aligned rows)
enum { EPI_FWD = 0, EPI_DGRAD = 1, EPI_PLAIN = 2 };

struct Args {
    const float* A; const float* B; float* C;
    int64_t lda, ldb, ldc;
    int M, N, K;               // output extents and the reduction extent
    int tiles_m, tiles_n;
    int k_per_slab;            // reduction elements per split slab (multiple of BK; >= K: no split)
    int64_t slab_stride;       // floats between consecutive slabs of C
    const float* bias;         // EPI_FWD: [N] (nullable)
    int relu;
    const float* H; int64_t ldh;   // EPI_DGRAD: [M, N] activations of the layer below (nullable: no mask)
    float* colsum;             // EPI_DGRAD: [tiles_m, N] per-tile-row column sums (nullable)
};

// one K-tile of an operand, global -> registers.  KC: stored [R, K] (reduction contiguous), R = the tile's 128 output
// rows / columns; else stored [K, R].  VEC floats per load.
template <bool KC, int VEC>
struct Stage {
    static constexpr int NV = BM * BK / 256 / VEC;       // loads per thread
    float v[NV][VEC];
    __device__ __forceinline__ void load(const float* __restrict__ X, int64_t ldx, int r0, int R, int k0, int k_end, int t) {
#pragma unroll
        for (int it = 0; it < NV; ++it) {
            int r, k;
            if (KC) {
                constexpr int QK = BK / VEC;             // loads per row
                const int q = t % QK, rr = t / QK + (256 / QK) * it;
                r = r0 + rr; k = k0 + q * VEC;
            } else {
                constexpr int QR = BM / VEC;             // loads per k-row
                const int q = t % QR, kk = t / QR + (256 / QR) * it;
                r = r0 + q * VEC; k = k0 + kk;
            }
            const bool ok = KC ? (r < R && k + VEC <= k_end) : (k < k_end && r + VEC <= R);
            const float* p = KC ? X + (int64_t)r * ldx + k : X + (int64_t)k * ldx + r;
            if (ok) {
                if (VEC == 4) { const float4 x = *(const float4*)p; v[it][0] = x.x; v[it][1 % VEC] = x.y; v[it][2 % VEC] = x.z; v[it][3 % VEC] = x.w; }
                else if (VEC == 2) { const float2 x = *(const float2*)p; v[it][0] = x.x; v[it][1 % VEC] = x.y; }
                else v[it][0] = *p;
            } else {
                // the ragged edge: element by element (k_end / R not multiples of VEC)
#pragma unroll
                for (int c = 0; c < VEC; ++c) {
                    const bool in = KC ? (r < R && k + c < k_end) : (k < k_end && r + c < R);
                    v[it][c] = in ? p[c] : 0.0f;
                }
            }
        }
    }
    __device__ __forceinline__ void store(float* __restrict__ S, int t) const {      // S: [BK][LD] reduction-major
#pragma unroll
        for (int it = 0; it < NV; ++it) {
            if (KC) {
                constexpr int QK = BK / VEC;
                const int q = t % QK, rr = t / QK + (256 / QK) * it;
#pragma unroll
                for (int c = 0; c < VEC; ++c) S[(q * VEC + c) * LD + rr] = v[it][c];
            } else {
                constexpr int QR = BM / VEC;
                const int q = t % QR, kk = t / QR + (256 / QR) * it;
                float* d = S + kk * LD + q * VEC;
                if (VEC == 4) *(float4*)d = make_float4(v[it][0], v[it][1 % VEC], v[it][2 % VEC], v[it][3 % VEC]);
                else if (VEC == 2) *(float2*)d = make_float2(v[it][0], v[it][1 % VEC]);
                else *d = v[it][0];
            }
        }
    }
};

template <bool AKC, bool BKC, int EPI, int VA, int VB>
__global__ __launch_bounds__(256, 2) void k_gemm_f32(const Args a) {
    __shared__ __attribute__((aligned(16))) float As[BK * LD];
    __shared__ __attribute__((aligned(16))) float Bs[BK * LD];
    __shared__ float red[2][BN];
    const int t = threadIdx.x, l = t & 63, w = t >> 6;
    const int wr = w >> 1, wc = w & 1;
    // workgroup -> (tn, tm, z); column tiles fastest: neighbours share the A panel
    int bid = blockIdx.x;
    const int tn = bid % a.tiles_n; bid /= a.tiles_n;
    const int tm = bid % a.tiles_m;
    const int z = bid / a.tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = z * a.k_per_slab;
    const int kend = min(a.K, kbeg + a.k_per_slab);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    Stage<AKC, VA> sa;
    Stage<BKC, VB> sb;
    sa.load(a.A, a.lda, m0, a.M, kbeg, kend, t);
    sb.load(a.B, a.ldb, n0, a.N, kbeg, kend, t);
    const int ai = wr * 64 + (l & 31), bj = wc * 64 + (l & 31), kh = l >> 5;
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        __syncthreads();                 // everybody is done reading the previous K-tile
        sa.store(As, t);
        sb.store(Bs, t);
        __syncthreads();
        if (k0 + BK < kend) {            // the next K-tile's loads fly while this one is multiplied
            sa.load(a.A, a.lda, m0, a.M, k0 + BK, kend, t);
            sb.load(a.B, a.ldb, n0, a.N, k0 + BK, kend, t);
        }
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
            const float* ar = As + (2 * s + kh) * LD + ai;
            const float* br = Bs + (2 * s + kh) * LD + bj;
            const float a0 = ar[0], a1 = ar[32], b0 = br[0], b1 = br[32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    }

    // ---- epilogue: lane owns column j = n0 + wc*64 + nj*32 + (l & 31), rows i = m0 + wr*64 + mi*32 + (r & 3) + 8 (r >> 2) + 4 kh
    float* C = a.C + (EPI == EPI_PLAIN ? (int64_t)z * a.slab_stride : 0);
    float cs[2] = {0.0f, 0.0f};
#pragma unroll
    for (int nj = 0; nj < 2; ++nj) {
        const int j = n0 + wc * 64 + nj * 32 + (l & 31);
        const bool jok = j < a.N;
        const float bv = (EPI == EPI_FWD && a.bias != nullptr && jok) ? a.bias[j] : 0.0f;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = m0 + wr * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                float v = acc[mi][nj][r];
                if (EPI == EPI_FWD) {
                    v = v + bv;
                    if (a.relu) v = v > 0.0f ? v : 0.0f;
                }
                const bool ok = jok && i < a.M;
                if (EPI == EPI_DGRAD) {
                    if (a.H != nullptr && ok && !(a.H[(int64_t)i * a.ldh + j] > 0.0f)) v = 0.0f;
                    if (ok) cs[nj] += v;         // rows in register order, then the fixed tree below
                }
                if (ok) C[(int64_t)i * a.ldc + j] = v;
            }
        }
    }
    if (EPI == EPI_DGRAD && a.colsum != nullptr) {
        // the lane halves (rows 4 kh), then the two wave rows through LDS: one partial per 128-row tile and column
#pragma unroll
        for (int nj = 0; nj < 2; ++nj) {
            cs[nj] += __shfl_xor(cs[nj], 32, 64);
            if (kh == 0) red[wr][wc * 64 + nj * 32 + (l & 31)] = cs[nj];
        }
        __syncthreads();
        if (t < BN && n0 + t < a.N) a.colsum[(int64_t)tm * a.N + n0 + t] = red[0][t] + red[1][t];
    }
}

inline int vec_of(const float* p, int64_t ld, int ext_c) {      // widest aligned load for rows `ld` apart whose contiguous extent is ext_c
    const uintptr_t u = (uintptr_t)p;
    if ((u & 15) == 0 && ld % 4 == 0) return 4;
    if ((u & 7) == 0 && ld % 2 == 0) return 2;
    (void)ext_c;
    return 1;
}

template <bool AKC, bool BKC, int EPI>
int launch(const Args& a, int va, int vb, int S, hipStream_t st) {
    const unsigned grid = (unsigned)((int64_t)a.tiles_m * a.tiles_n * S);
#define GF_GO(VA, VB) k_gemm_f32<AKC, BKC, EPI, VA, VB><<<grid, 256, 0, st>>>(a)
    if (va == 4 && vb == 4) GF_GO(4, 4);
    else if (va == 4 && vb == 2) GF_GO(4, 2);
    else if (va == 2 && vb == 4) GF_GO(2, 4);
    else if (va == 2 && vb == 2) GF_GO(2, 2);
    else if (va == 1 && vb == 1) GF_GO(1, 1);
    else if (va == 1) { if (vb == 4) GF_GO(1, 4); else GF_GO(1, 2); }
    else { if (va == 4) GF_GO(4, 1); else GF_GO(2, 1); }
#undef GF_GO
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}

}  // namespace gf32

using gf32::Args;

/* y = act(x . w + bias): x [M, K] (ldx), w [K, N] contiguous rows (ldw), bias [N] (nullable), y [M, N] (ldy) */
MREC_API int mrec_dense32_fwd(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, int64_t M, int32_t K,
                              int32_t N, int relu, float* y, int64_t ldy, void* stream) {
    if (M < 0 || K <= 0 || N <= 0 || ldx < K || ldw < N || ldy < N) return MREC_EINVAL;
    if (M == 0) return MREC_OK;
    if (!x || !w || !y) return MREC_EINVAL;
    if (M > (int64_t(1) << 30)) return MREC_EUNSUPPORTED;
    Args a{};
    a.A = x; a.lda = ldx; a.B = w; a.ldb = ldw; a.C = y; a.ldc = ldy;
    a.M = (int)M; a.N = N; a.K = K;
    a.tiles_m = (int)mrec_cdiv(M, gf32::BM); a.tiles_n = (int)mrec_cdiv(N, gf32::BN);
    a.k_per_slab = (int)mrec_align_up((size_t)K, gf32::BK);
    a.bias = bias; a.relu = relu;
    return gf32::launch<true, false, gf32::EPI_FWD>(a, gf32::vec_of(x, ldx, K), gf32::vec_of(w, ldw, N), 1, (hipStream_t)stream);
}

/* dx = (dy . w^T) * (h > 0): dy [M, N] (lddy), w [K, N] (ldw), h [M, K] (ldh, nullable), dx [M, K] (lddx);
 * colsum_ws (nullable): [ceil(M / 128), K] per-tile-row column sums of dx = the partials of the layer below's bias gradient */
MREC_API int mrec_dense32_bwd_input(const float* dy, int64_t lddy, const float* w, int64_t ldw, const float* h, int64_t ldh, int64_t M,
                                    int32_t K, int32_t N, float* dx, int64_t lddx, float* colsum_ws, void* stream) {
    if (M < 0 || K <= 0 || N <= 0 || lddy < N || ldw < N || lddx < K || (h && ldh < K)) return MREC_EINVAL;
    if (M == 0) return MREC_OK;
    if (!dy || !w || !dx) return MREC_EINVAL;
    if (M > (int64_t(1) << 30)) return MREC_EUNSUPPORTED;
    Args a{};
    a.A = dy; a.lda = lddy; a.B = w; a.ldb = ldw; a.C = dx; a.ldc = lddx;
    a.M = (int)M; a.N = K; a.K = N;                       // outputs [M, K], reduction over N
    a.tiles_m = (int)mrec_cdiv(M, gf32::BM); a.tiles_n = (int)mrec_cdiv(K, gf32::BN);
    a.k_per_slab = (int)mrec_align_up((size_t)N, gf32::BK);
    a.H = h; a.ldh = ldh; a.colsum = colsum_ws;
    return gf32::launch<true, true, gf32::EPI_DGRAD>(a, gf32::vec_of(dy, lddy, N), gf32::vec_of(w, ldw, N), 1, (hipStream_t)stream);
}

/* dw_slabs[s] = x[slab s]^T . dy[slab s]: x [M, K] (ldx), dy [M, N] (lddy), dw_slabs [S, K, N] contiguous; the batch is cut
 * into S slabs of ceil(M / S / 32) * 32 rows (mrec_dense32_bwd_weight_slabs proposes S) */
MREC_API int mrec_dense32_bwd_weight(const float* x, int64_t ldx, const float* dy, int64_t lddy, int64_t M, int32_t K, int32_t N,
                                     int32_t S, float* dw_slabs, void* stream) {
    if (M < 0 || K <= 0 || N <= 0 || S <= 0 || ldx < K || lddy < N) return MREC_EINVAL;
    if (!dw_slabs) return MREC_EINVAL;
    if (M == 0) {
        MREC_HIP_CHECK(hipMemsetAsync(dw_slabs, 0, (size_t)S * K * N * sizeof(float), (hipStream_t)stream));
        return MREC_OK;
    }
    if (!x || !dy) return MREC_EINVAL;
    if (M > (int64_t(1) << 30)) return MREC_EUNSUPPORTED;
    Args a{};
    a.A = x; a.lda = ldx; a.B = dy; a.ldb = lddy; a.C = dw_slabs; a.ldc = N;
    a.M = K; a.N = N; a.K = (int)M;                       // outputs [K, N], reduction over the batch
    a.tiles_m = (int)mrec_cdiv(K, gf32::BM); a.tiles_n = (int)mrec_cdiv(N, gf32::BN);
    a.k_per_slab = (int)mrec_align_up((size_t)mrec_cdiv(M, S), gf32::BK);
    a.slab_stride = (int64_t)K * N;
    if ((int64_t)a.k_per_slab * (S - 1) >= M && S > 1) return MREC_EINVAL;      // an empty slab: S too large for this batch
    return gf32::launch<false, false, gf32::EPI_PLAIN>(a, gf32::vec_of(x, ldx, K), gf32::vec_of(dy, lddy, N), S, (hipStream_t)stream);
}

/* Batch slabs that fill the chip about twice: the output has only ceil(K / 128) * ceil(N / 128) tiles */
MREC_API int mrec_dense32_bwd_weight_slabs(int64_t M, int32_t K, int32_t N, int32_t* out) {
    if (!out || M < 0 || K <= 0 || N <= 0) return MREC_EINVAL;
    const int64_t tiles = mrec_cdiv(K, gf32::BM) * mrec_cdiv(N, gf32::BN);
    int64_t S = mrec_cdiv(512, tiles);
    const int64_t smax = M / 256 > 0 ? M / 256 : 1;       // at least 256 batch rows per slab
    if (S > smax) S = smax;
    if (S > 64) S = 64;
    while (S > 1 && (int64_t)mrec_align_up((size_t)mrec_cdiv(M, S), gf32::BK) * (S - 1) >= M) --S;
    *out = (int32_t)S;
    return MREC_OK;
}
