// mrec_gemm_f32.hip -- DenseLayer in fp32 (convert_dtype=False: Deep&Cross, models/deep_and_cross/src/deep_and_cross.py:94-114,
// 293-309) and its two bprops on the fp32-input matrix instruction of gfx950, v_mfma_f32_32x32x2_f32.
//
// The fp32 MFMA runs at the fp32 vector rate (157 TFLOP/s, 1/16 of the 16-bit forms) and is bit-for-bit a k-ordered chain of
// fmaf's: one rounding per product, no wider accumulation.  At that rate nothing else on the CU is near its limit -- what
// costs is an idle matrix pipe: rounds of workgroups that do not fill the chip, and the waits around the staging.  So:
//   * a BM x 128 output tile per 256-thread workgroup, BM = 128 or 64 picked per problem so that the tiles come out in full
//     rounds of the 512 workgroups the chip holds (2 per CU); four waves as 2 x 2, each (BM / 2) x 64 = BM / 64 x 2
//     accumulators of 32 x 32;
//   * a 32-deep K-tile in LDS in REDUCTION-MAJOR order (rows = k, 132 floats apart; the instruction wants
//     A[i = lane & 31][k = lane >> 5], so a fragment is one conflict-free ds_read_b32), two buffers: the next K-tile's global
//     loads are issued before the current one is multiplied and written to the other buffer behind its MFMAs -- ONE barrier
//     per K-tile, with the second workgroup of the CU computing across it;
//   * branch-free staging: a load whose vector lies outside the operand reads the operand's first element instead and is
//     zeroed by a select (rows need only 8-byte alignment -- the DCN input is 39 x 30 = 1170 floats wide -- so VEC = 4, 2 or 1
//     floats per load is picked per operand such that no vector straddles an edge);
//   * a tile that hangs over the output's edge skips the MFMAs of its 32 x 32 blocks that lie wholly outside.
//
// Operands are either reduction-contiguous (X[r, k]: a row of the operand is a row of the matrix: transposed on its way into
// LDS, VEC ds_write_b32 per load) or reduction-strided (Y[k, c]: written as loaded):
//   forward   y  = x . W        A = x  [M, K] contiguous      B = W  [K, N] strided
//   dgrad     dx = dy . W^T     A = dy [M, N] contiguous      B = W  [K, N] contiguous (its rows ARE the outputs)
//   wgrad     dW = x^T . dy     A = x  [M, K] strided         B = dy [M, N] strided, reduction over the batch, split in slabs
// Epilogues: bias + ReLU; mask by the activation below > 0 + column sums per 64 output rows (the layer below's BiasAdd
// bprop; the same partials in the same order whichever BM ran); plain fp32 slabs.
#include "mrec_common.h"

namespace gf32 {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BN = 128, BK = 32, LD = 132;      // column tile, K-tile; LDS row stride in floats (528 B: 16-byte aligned rows)
constexpr int TILE_F = BK * LD;                 // floats per operand tile in LDS
constexpr int LDS_BYTES = (4 * TILE_F + 2 * BN) * (int)sizeof(float);
constexpr int SLOTS = 512;                      // workgroups the chip holds at 2 per CU
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
    float* colsum;             // EPI_DGRAD: [ceil(M / 64), N] column sums per 64 output rows (nullable)
};

// one K-tile of an operand, global -> registers -> LDS.  KC: stored [R, K] (reduction contiguous), R = the tile's ROWS output
// rows / columns; else stored [K, R].  VEC floats per load; the host guarantees that no vector straddles an edge.
template <bool KC, int VEC, int ROWS>
struct Stage {
    static constexpr int NV = ROWS * BK / 256 / VEC;       // loads per thread
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
                constexpr int QR = ROWS / VEC;           // loads per k-row
                const int q = t % QR, kk = t / QR + (256 / QR) * it;
                r = r0 + q * VEC; k = k0 + kk;
            }
            const bool ok = r < R && k < k_end;
            const float* p = ok ? (KC ? X + (int64_t)r * ldx + k : X + (int64_t)k * ldx + r) : X;
            if (VEC == 4) {
                const float4 x = *(const float4*)p;
                v[it][0] = ok ? x.x : 0.0f; v[it][1 % VEC] = ok ? x.y : 0.0f; v[it][2 % VEC] = ok ? x.z : 0.0f; v[it][3 % VEC] = ok ? x.w : 0.0f;
            } else if (VEC == 2) {
                const float2 x = *(const float2*)p;
                v[it][0] = ok ? x.x : 0.0f; v[it][1 % VEC] = ok ? x.y : 0.0f;
            } else {
                const float x = *p;
                v[it][0] = ok ? x : 0.0f;
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
                constexpr int QR = ROWS / VEC;
                const int q = t % QR, kk = t / QR + (256 / QR) * it;
                float* d = S + kk * LD + q * VEC;
                if (VEC == 4) *(float4*)d = make_float4(v[it][0], v[it][1 % VEC], v[it][2 % VEC], v[it][3 % VEC]);
                else if (VEC == 2) *(float2*)d = make_float2(v[it][0], v[it][1 % VEC]);
                else *d = v[it][0];
            }
        }
    }
};

// the 16 k-steps of one K-tile: fragments of step s + 1 are requested before the MFMAs of step s
template <int MI, bool FULL>
__device__ __forceinline__ void ktile(f32x16 (&acc)[MI][2], const float* __restrict__ As, const float* __restrict__ Bs, int ai, int bj, int kh,
                                      const bool (&mv)[MI], const bool (&nv)[2]) {
    float fa[2][MI], fb[2][2];
    const float* ar = As + kh * LD + ai;
    const float* br = Bs + kh * LD + bj;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) fa[0][mi] = ar[mi * 32];
#pragma unroll
    for (int nj = 0; nj < 2; ++nj) fb[0][nj] = br[nj * 32];
#pragma unroll
    for (int s = 0; s < BK / 2; ++s) {
        const int cur = s & 1, nxt = cur ^ 1;
        if (s + 1 < BK / 2) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) fa[nxt][mi] = ar[(2 * s + 2) * LD + mi * 32];
#pragma unroll
            for (int nj = 0; nj < 2; ++nj) fb[nxt][nj] = br[(2 * s + 2) * LD + nj * 32];
        }
        __builtin_amdgcn_sched_barrier(0);      // (left alone the scheduler sinks the reads to just in front of their MFMAs)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int nj = 0; nj < 2; ++nj)
                if (FULL || (mv[mi] && nv[nj]))
                    acc[mi][nj] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][mi], fb[cur][nj], acc[mi][nj], 0, 0, 0);
    }
}

template <int BM, bool AKC, bool BKC, int EPI, int VA, int VB>
__global__ __launch_bounds__(256, 2) void k_gemm_f32(const Args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];      // As[2] | Bs[2] | red[2][BN]
    constexpr int MI = BM / 64;
    float* const As = smem;
    float* const Bs = smem + 2 * TILE_F;
    float* const red = smem + 4 * TILE_F;
    const int t = threadIdx.x, l = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = w >> 1, wc = w & 1;
    // workgroup -> (tn, tm, z); column tiles fastest: neighbours share the A panel
    int bid = blockIdx.x;
    const int tn = bid % a.tiles_n; bid /= a.tiles_n;
    const int tm = bid % a.tiles_m;
    const int z = bid / a.tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = z * a.k_per_slab;
    const int kend = min(a.K, kbeg + a.k_per_slab);

    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // 32 x 32 blocks of this wave that hold any output at all (wave-uniform)
    bool mv[MI], nv[2];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) mv[mi] = m0 + wr * (BM / 2) + mi * 32 < a.M;
#pragma unroll
    for (int nj = 0; nj < 2; ++nj) nv[nj] = n0 + wc * 64 + nj * 32 < a.N;
    const bool full = (m0 + BM <= a.M) && (n0 + BN <= a.N);

    Stage<AKC, VA, BM> sa;
    Stage<BKC, VB, BN> sb;
    const int ai = wr * (BM / 2) + (l & 31), bj = wc * 64 + (l & 31), kh = l >> 5;
    if (kbeg < kend) {
        sa.load(a.A, a.lda, m0, a.M, kbeg, kend, t);
        sb.load(a.B, a.ldb, n0, a.N, kbeg, kend, t);
        sa.store(As, t);
        sb.store(Bs, t);
    }
    __syncthreads();
    int cur = 0;
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        const bool more = k0 + BK < kend;
        if (more) {                      // the next K-tile's loads fly while this one is multiplied
            sa.load(a.A, a.lda, m0, a.M, k0 + BK, kend, t);
            sb.load(a.B, a.ldb, n0, a.N, k0 + BK, kend, t);
        }
        if (full) ktile<MI, true>(acc, As + cur * TILE_F, Bs + cur * TILE_F, ai, bj, kh, mv, nv);
        else ktile<MI, false>(acc, As + cur * TILE_F, Bs + cur * TILE_F, ai, bj, kh, mv, nv);
        if (more) {                      // the other buffer was last read one K-tile ago, in front of the barrier below
            sa.store(As + (cur ^ 1) * TILE_F, t);
            sb.store(Bs + (cur ^ 1) * TILE_F, t);
        }
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: lane owns column j = n0 + wc*64 + nj*32 + (l & 31), rows i = m0 + wr*(BM/2) + mi*32 + (r & 3) + 8 (r >> 2) + 4 kh
    float* C = a.C + (EPI == EPI_PLAIN ? (int64_t)z * a.slab_stride : 0);
    float cs[MI][2];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) cs[mi][0] = cs[mi][1] = 0.0f;
#pragma unroll
    for (int nj = 0; nj < 2; ++nj) {
        const int j = n0 + wc * 64 + nj * 32 + (l & 31);
        const bool jok = j < a.N;
        const float bv = (EPI == EPI_FWD && a.bias != nullptr && jok) ? a.bias[j] : 0.0f;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = m0 + wr * (BM / 2) + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                float v = acc[mi][nj][r];
                if (EPI == EPI_FWD) {
                    v = v + bv;
                    if (a.relu) v = v > 0.0f ? v : 0.0f;
                }
                const bool ok = jok && i < a.M;
                if (EPI == EPI_DGRAD) {
                    if (a.H != nullptr && ok && !(a.H[(int64_t)i * a.ldh + j] > 0.0f)) v = 0.0f;
                    if (ok) cs[mi][nj] += v;     // rows in register order, then the fixed tree below
                }
                if (ok) C[(int64_t)i * a.ldc + j] = v;
            }
        }
    }
    if (EPI == EPI_DGRAD && a.colsum != nullptr) {
        // per 32-row block: the two lane halves (rows 4 kh); per 64 rows: block 0 + block 1 -- the two blocks of a wave
        // (BM = 128) or of the two wave rows (BM = 64, through LDS): the same partials in the same order either way
#pragma unroll
        for (int nj = 0; nj < 2; ++nj) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) cs[mi][nj] += __shfl_xor(cs[mi][nj], 32, 64);
            const int jl = wc * 64 + nj * 32 + (l & 31);
            if (MI == 2) {
                const int row = 2 * tm + wr;
                if (kh == 0 && n0 + jl < a.N && (int64_t)row * 64 < a.M) a.colsum[(int64_t)row * a.N + n0 + jl] = cs[0][nj] + cs[MI - 1][nj];
            } else if (kh == 0) {
                red[wr * BN + jl] = cs[0][nj];
            }
        }
        if (MI == 1) {
            __syncthreads();
            if (t < BN && n0 + t < a.N) a.colsum[(int64_t)tm * a.N + n0 + t] = red[t] + red[BN + t];
        }
    }
}

// widest aligned load for rows `ld` apart whose contiguous extent is ext_c: no vector may straddle the extent's end
inline int vec_of(const float* p, int64_t ld, int ext_c) {
    const uintptr_t u = (uintptr_t)p;
    if ((u & 15) == 0 && ld % 4 == 0 && ext_c % 4 == 0) return 4;
    if ((u & 7) == 0 && ld % 2 == 0 && ext_c % 2 == 0) return 2;
    return 1;
}

// BM = 64 when its tiles come out in fewer (half-length) rounds of the chip than those of BM = 128
inline int pick_bm(int64_t M, int64_t N, int64_t S) {
    const int64_t tn = mrec_cdiv(N, BN);
    const int64_t c128 = mrec_cdiv(mrec_cdiv(M, 128) * tn * S, SLOTS) * 2;
    const int64_t c64 = mrec_cdiv(mrec_cdiv(M, 64) * tn * S, SLOTS);
    return c64 < c128 ? 64 : 128;
}

template <int BM, bool AKC, bool BKC, int EPI, int VA, int VB>
int launch_one(const Args& a, unsigned grid, hipStream_t st) {
    static bool attr_done = false;      // (per instantiation; racing threads set the same value)
    if (!attr_done) {
        MREC_HIP_CHECK(hipFuncSetAttribute((const void*)k_gemm_f32<BM, AKC, BKC, EPI, VA, VB>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr_done = true;
    }
    k_gemm_f32<BM, AKC, BKC, EPI, VA, VB><<<grid, 256, LDS_BYTES, st>>>(a);
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}

template <int BM, bool AKC, bool BKC, int EPI>
int launch_bm(const Args& a, int va, int vb, unsigned grid, hipStream_t st) {
#define GF_GO(VA, VB) return launch_one<BM, AKC, BKC, EPI, VA, VB>(a, grid, st)
    if (va == 4 && vb == 4) GF_GO(4, 4);
    else if (va == 4 && vb == 2) GF_GO(4, 2);
    else if (va == 2 && vb == 4) GF_GO(2, 4);
    else if (va == 2 && vb == 2) GF_GO(2, 2);
    else if (va == 1 && vb == 1) GF_GO(1, 1);
    else if (va == 1) { if (vb == 4) GF_GO(1, 4); else GF_GO(1, 2); }
    else { if (va == 4) GF_GO(4, 1); else GF_GO(2, 1); }
#undef GF_GO
}

// a: M, N, K, operands and epilogue set; tiles and the grid are filled in here
template <bool AKC, bool BKC, int EPI>
int launch(Args a, int va, int vb, int S, hipStream_t st) {
    const int bm = pick_bm(a.M, a.N, S);
    a.tiles_m = (int)mrec_cdiv(a.M, bm);
    a.tiles_n = (int)mrec_cdiv(a.N, BN);
    const unsigned grid = (unsigned)((int64_t)a.tiles_m * a.tiles_n * S);
    return bm == 64 ? launch_bm<64, AKC, BKC, EPI>(a, va, vb, grid, st) : launch_bm<128, AKC, BKC, EPI>(a, va, vb, grid, st);
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
    a.k_per_slab = (int)mrec_align_up((size_t)K, gf32::BK);
    a.bias = bias; a.relu = relu;
    return gf32::launch<true, false, gf32::EPI_FWD>(a, gf32::vec_of(x, ldx, K), gf32::vec_of(w, ldw, N), 1, (hipStream_t)stream);
}

/* dx = (dy . w^T) * (h > 0): dy [M, N] (lddy), w [K, N] (ldw), h [M, K] (ldh, nullable), dx [M, K] (lddx);
 * colsum_ws (nullable): [ceil(M / 64), K] column sums of dx per 64 rows = the partials of the layer below's bias gradient */
MREC_API int mrec_dense32_bwd_input(const float* dy, int64_t lddy, const float* w, int64_t ldw, const float* h, int64_t ldh, int64_t M,
                                    int32_t K, int32_t N, float* dx, int64_t lddx, float* colsum_ws, void* stream) {
    if (M < 0 || K <= 0 || N <= 0 || lddy < N || ldw < N || lddx < K || (h && ldh < K)) return MREC_EINVAL;
    if (M == 0) return MREC_OK;
    if (!dy || !w || !dx) return MREC_EINVAL;
    if (M > (int64_t(1) << 30)) return MREC_EUNSUPPORTED;
    Args a{};
    a.A = dy; a.lda = lddy; a.B = w; a.ldb = ldw; a.C = dx; a.ldc = lddx;
    a.M = (int)M; a.N = K; a.K = N;                       // outputs [M, K], reduction over N
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
    a.k_per_slab = (int)mrec_align_up((size_t)mrec_cdiv(M, S), gf32::BK);
    a.slab_stride = (int64_t)K * N;
    if ((int64_t)a.k_per_slab * (S - 1) >= M && S > 1) return MREC_EINVAL;      // an empty slab: S too large for this batch
    return gf32::launch<false, false, gf32::EPI_PLAIN>(a, gf32::vec_of(x, ldx, K), gf32::vec_of(dy, lddy, N), S, (hipStream_t)stream);
}

/* Batch slabs: the output has only ceil(K / BM) * ceil(N / 128) tiles.  The count (and with it the tile height the launch will pick)
 * that takes the fewest rounds of the chip's 512 workgroup slots x rows per workgroup; among equals the smallest (the slabs are
 * the optimizer's to read). */
MREC_API int mrec_dense32_bwd_weight_slabs(int64_t M, int32_t K, int32_t N, int32_t* out) {
    if (!out || M < 0 || K <= 0 || N <= 0) return MREC_EINVAL;
    int64_t smax = M / 256 > 0 ? M / 256 : 1;             // at least 256 batch rows per slab
    if (smax > 64) smax = 64;
    int64_t best = 1, best_cost = -1;
    for (int64_t S = 1; S <= smax; ++S) {
        const int64_t rows = (int64_t)mrec_align_up((size_t)mrec_cdiv(M, S), gf32::BK);
        if (S > 1 && rows * (S - 1) >= M) continue;       // an empty slab
        const int bm = gf32::pick_bm(K, N, S);
        const int64_t rounds = mrec_cdiv(mrec_cdiv(K, bm) * mrec_cdiv(N, gf32::BN) * S, gf32::SLOTS);
        const int64_t cost = rounds * bm * rows;
        if (best_cost < 0 || cost < best_cost) { best = S; best_cost = cost; }
    }
    *out = (int32_t)best;
    return MREC_OK;
}
