// mrec_gemm_x3.hip -- fp32 DenseLayers (models/deep_and_cross/src/deep_and_cross.py:94-114, convert_dtype=False; the reference's
// benchmark net, benchmarks/wide_deep/default_config.yaml:16 use_mixed_precision: False) at the 16-bit matrix rate.
//
// gfx950's fp32-input MFMA runs at 1/16 of the bf16 one (157 vs 2500 TFLOP/s), and the exact-fp32 kernel of mrec_gemm_f32.hip
// holds 62-71 % of that.  Here every fp32 operand is held as THREE bf16 parts, x = x1 + x2 + x3 with x1 = bf16(x), x2 = bf16(x - x1),
// x3 = bf16(x - x1 - x2): 3 x 8 = 24 mantissa bits, the residuals exact in fp32 -- and a product a . b becomes the six bf16
// products a1 b1, a1 b2, a2 b1, a1 b3, a2 b2, a3 b1 (the three left out are below 2^-24 of a . b: under fp32's own rounding),
// all accumulated in the MFMA's fp32 accumulators as six SEGMENTS of ONE reduction (smallest terms first) through the 16-bit
// GEMM body of mrec_gemm.h (VAR & 4: a K-tile's operand offset comes from its segment): 6/16 of the fp32-MFMA time at the same
// accuracy class as an fp32 GEMM (error <= ~2^-22 of sum |a b|, measured against float64 in tests/test_dense32_gpu.py).
//
//   mrec_x3_split         fp32 [R, C] -> bf16 parts [3][Rp][Cp] (Rp, Cp = R, C rounded up to 64, zero padded)
//   mrec_x3_gemm          form 0: C[M, N] = P[M, K] . Q[K, N]      (DenseLayer forward)
//                         form 1: C[M, K] = P[M, N] . Q[K, N]^T    (input gradient)
//                         form 2: C[S][K, N] = P[M, K]^T . Q[M, N] (weight gradient, S batch slabs of fp32 partial sums)
//   mrec_x3_bias_relu     y = relu?(acc + bias) in place, and (optionally) y's own three parts for the next GEMM, one pass
//   mrec_x3_mask_colsum   dx = acc masked by h > 0 in place, column sums per 64 rows (the bias gradient of the layer below),
//                         and (optionally) dx's parts, one pass
#include "mrec_common.h"
#include <cstdlib>
#include "mrec_gemm.h"
#include "mrec_dropout.h"

namespace {

using mgemm::Args;

// the six products, smallest first: parts (a, b) = (2, 0) (1, 1) (0, 2) (1, 0) (0, 1) (0, 0); two bits per segment
constexpr uint32_t kCodeP = 2u | (1u << 2) | (0u << 4) | (1u << 6) | (0u << 8) | (0u << 10);
constexpr uint32_t kCodeQ = 0u | (1u << 2) | (2u << 4) | (0u << 6) | (1u << 8) | (0u << 10);

inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
inline int64_t up64(int64_t x) { return (x + 63) / 64 * 64; }

__device__ __forceinline__ void split3(float x, uint16_t& a, uint16_t& b, uint16_t& c) {
    const __bf16 x1 = (__bf16)x;
    const float r1 = x - (float)x1;
    const __bf16 x2 = (__bf16)r1;
    const float r2 = r1 - (float)x2;
    const __bf16 x3 = (__bf16)r2;
    a = __builtin_bit_cast(uint16_t, x1); b = __builtin_bit_cast(uint16_t, x2); c = __builtin_bit_cast(uint16_t, x3);
}

// 8 consecutive values -> the three parts' 16-byte groups
__device__ __forceinline__ void split8(const float (&v)[8], uint4& a, uint4& b, uint4& c) {
    uint16_t pa[8], pb[8], pc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) split3(v[k], pa[k], pb[k], pc[k]);
    a = make_uint4(pa[0] | ((unsigned)pa[1] << 16), pa[2] | ((unsigned)pa[3] << 16), pa[4] | ((unsigned)pa[5] << 16), pa[6] | ((unsigned)pa[7] << 16));
    b = make_uint4(pb[0] | ((unsigned)pb[1] << 16), pb[2] | ((unsigned)pb[3] << 16), pb[4] | ((unsigned)pb[5] << 16), pb[6] | ((unsigned)pb[7] << 16));
    c = make_uint4(pc[0] | ((unsigned)pc[1] << 16), pc[2] | ((unsigned)pc[3] << 16), pc[4] | ((unsigned)pc[5] << 16), pc[6] | ((unsigned)pc[7] << 16));
}

// One thread per 8 columns of the PADDED image.  MODE 0: plain split; 1: y = relu?(x + bias) written back + split;
// 2: dx = h > 0 ? x : 0 written back + split, column sums per 64 rows.
template <int MODE>
__global__ __launch_bounds__(256) void k_x3_post(float* __restrict__ x, int64_t ldx, int64_t R, int C, const float* __restrict__ bias,
                                                 int relu, const float* __restrict__ h, int64_t ldh, float* __restrict__ colsum,
                                                 uint16_t* __restrict__ parts, int64_t Rp, int64_t Cp, float scale) {
    // block = 64 rows x 32 column groups (256 columns); thread (rq, cg): rows rq, rq + 8, ... of column group cg
    const int cg = threadIdx.x & 31, rq = threadIdx.x >> 5;
    const int64_t r0 = (int64_t)blockIdx.y * 64;
    const int c0 = (blockIdx.x * 32 + cg) * 8;
    if (c0 >= Cp) return;
    float b8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (MODE == 1 && bias) {
#pragma unroll
        for (int k = 0; k < 8; ++k) b8[k] = (c0 + k < C) ? bias[c0 + k] : 0.f;
    }
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bool vec = (ldx % 4 == 0) && (c0 + 8 <= C);
    for (int rr = rq; rr < 64; rr += 8) {
        const int64_t r = r0 + rr;
        if (r >= Rp) break;
        float v[8];
        const bool live = r < R;
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = 0.f;
        if (live) {
            if (vec) {
                const float4 lo = *(const float4*)(x + r * ldx + c0), hi = *(const float4*)(x + r * ldx + c0 + 4);
                v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) if (c0 + k < C) v[k] = x[r * ldx + c0 + k];
            }
            if (MODE == 1) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    v[k] = v[k] + b8[k];
                    if (relu) v[k] = v[k] > 0.f ? v[k] : 0.f;
                    if (c0 + k >= C) v[k] = 0.f;
                }
            }
            if (MODE == 2) {
                if (h) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) if (c0 + k < C && !(h[r * ldh + c0 + k] > 0.f)) v[k] = 0.f;
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] *= scale;            // (1 / keep of a Dropout on the layer's input; 1 otherwise)
#pragma unroll
                for (int k = 0; k < 8; ++k) cs[k] += v[k];
            }
            if (MODE != 0) {
                if (vec) {
                    *(float4*)(x + r * ldx + c0) = make_float4(v[0], v[1], v[2], v[3]);
                    *(float4*)(x + r * ldx + c0 + 4) = make_float4(v[4], v[5], v[6], v[7]);
                } else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) if (c0 + k < C) x[r * ldx + c0 + k] = v[k];
                }
            }
        }
        if (parts) {
            uint4 a, b, c;
            split8(v, a, b, c);
            const int64_t o = r * Cp + c0;
            *(uint4*)(parts + o) = a;
            *(uint4*)(parts + Rp * Cp + o) = b;
            *(uint4*)(parts + 2 * Rp * Cp + o) = c;
        }
    }
    if (MODE == 2 && colsum) {
        __shared__ float red[8][32][8];
#pragma unroll
        for (int k = 0; k < 8; ++k) red[rq][cg][k] = cs[k];
        __syncthreads();
        if (rq == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float s = red[0][cg][k];
                for (int q = 1; q < 8; ++q) s += red[q][cg][k];
                if (c0 + k < C && r0 < R) colsum[(int64_t)blockIdx.y * C + c0 + k] = s;
            }
        }
    }
}

// Two problems of the same kind in one launch: workgroups [0, n1) the first, the rest the second (uniform selects: the arguments stay in
// SGPRs).  The input gradient into a width that leaves a narrow last column tile (Deep&Cross's 1170 = 4 x 256 + 146) runs its full tiles
// as one round over the chip and the narrow tile as S short slabs of the reduction behind them, instead of two rounds at 62 %.
template <bool PT, bool QT, int MR>
__global__ __launch_bounds__(mgemm::kThreads, 2) void k_x3_pair(const Args a1, const Args a2, const int n1) {
    __shared__ __attribute__((aligned(1024))) char smem[mgemm::Lds<MR>::bytes];
    Args sel = a1;
    int b = blockIdx.x, nb = n1;
    if (b >= n1) { sel = a2; b -= n1; nb = (int)gridDim.x - n1; }
    mgemm::gemm256_body<MR, PT, QT, mgemm::EPI_F32, false, 4>(sel, b, nb, (MGEMM_LDS char*)smem);
}

// dx[r, c0 + c] = sum_s ws[s][r][c]  (c < rem; ws rows of ldw floats, ldw % 4 == 0; dx rows 8-byte aligned)
__global__ __launch_bounds__(256) void k_x3_fold_cols(const float* __restrict__ ws, int S, int64_t M, int rem, int ldw, float* __restrict__ dx,
                                                      int64_t lddx, int c0) {
    const int per = ldw / 4;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M * per) return;
    const int64_t r = i / per;
    const int c = (int)(i - r * per) * 4;
    float4 acc = *(const float4*)(ws + r * ldw + c);
    for (int s = 1; s < S; ++s) {
        const float4 v = *(const float4*)(ws + ((int64_t)s * M + r) * ldw + c);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    float* d = dx + r * lddx + c0 + c;
    if (c + 2 <= rem) *(float2*)d = make_float2(acc.x, acc.y); else if (c < rem) d[0] = acc.x;
    if (c + 4 <= rem) *(float2*)(d + 2) = make_float2(acc.z, acc.w); else if (c + 2 < rem) d[2] = acc.z;
}

template <int MODE>
int post_launch(float* x, int64_t ldx, int64_t R, int32_t C, const float* bias, int relu, const float* h, int64_t ldh, float* colsum,
                uint16_t* parts, void* stream, float scale = 1.0f) {
    if (R < 0 || C <= 0 || ldx < C || !x) return MREC_EINVAL;
    if (R == 0) return MREC_OK;
    if (ldx % 2 || (parts && !al16(parts)) || (((uintptr_t)x) & 7)) return MREC_EUNSUPPORTED;
    const int64_t Rp = up64(R), Cp = up64(C);
    const int64_t rows = parts ? Rp : R;
    dim3 grid((unsigned)mrec_cdiv(Cp, 256), (unsigned)mrec_cdiv(rows, 64));
    k_x3_post<MODE><<<grid, 256, 0, (hipStream_t)stream>>>(x, ldx, R, C, bias, relu, h, ldh, colsum, parts, Rp, Cp, scale);
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}

}  // namespace

MREC_API int mrec_x3_parts_elems(int64_t rows, int64_t cols, int64_t* out) {
    if (!out || rows < 0 || cols < 0) return MREC_EINVAL;
    *out = 3 * up64(rows) * up64(cols);
    return MREC_OK;
}

// Batch slabs of the weight gradient: as many as put one 256 x 256 tile x slab on each of the 256 CUs (measured at M = 16384: 1170 x 1024
// in 12 slabs of big tiles 260 us, in 6 slabs of 128-row tiles 298 us; 1024 x 1024 in 16 slabs 181 us, in 8 slabs of 128-row tiles 219 us),
// every slab with a non-empty share of the 6 * ceil(M / 64) reduction tiles.
MREC_API int mrec_x3_wgrad_slabs(int64_t M, int32_t K, int32_t N, int32_t* out) {
    if (!out || M <= 0 || K <= 0 || N <= 0) return MREC_EINVAL;
    const int64_t tiles = mrec_cdiv(K, 256) * mrec_cdiv(N, 256), T = 6 * (up64(M) / 64);
    int64_t S = 256 / tiles;
    S = S < 1 ? 1 : (S > 32 ? 32 : S);
    S = S > T ? T : S;
    while (S > 1 && mrec_cdiv(T, S) * (S - 1) >= T) --S;
    *out = (int32_t)S;
    return MREC_OK;
}

MREC_API int mrec_x3_split(const float* x, int64_t ldx, int64_t R, int32_t C, uint16_t* parts, void* stream) {
    if (!parts) return MREC_EINVAL;
    return post_launch<0>(const_cast<float*>(x), ldx, R, C, nullptr, 0, nullptr, 0, nullptr, parts, stream);
}

MREC_API int mrec_x3_bias_relu(float* acc, int64_t ld, int64_t M, int32_t N, const float* bias, int relu, uint16_t* parts_out, void* stream) {
    return post_launch<1>(acc, ld, M, N, bias, relu, nullptr, 0, nullptr, parts_out, stream);
}

MREC_API int mrec_x3_mask_colsum(float* acc, int64_t ld, int64_t M, int32_t K, const float* h, int64_t ldh, float scale, float* colsum,
                                 uint16_t* parts_out, void* stream) {
    if ((h && ldh < K) || !(scale > 0.0f)) return MREC_EINVAL;
    return post_launch<2>(acc, ld, M, K, nullptr, 0, h, ldh, colsum, parts_out, stream, scale);
}

namespace {
struct X3Epi {                // the fused output end (EPI_X3); mode 0: plain fp32 out
    int mode; const float* bias; int relu; const float* h; int64_t ldh; float scale; float* colsum; uint16_t* parts;
    DropArgs drop;            // mode 1: Dropout on the output (thresh 0: none)
};
int x3_gemm(int form, const uint16_t* Pparts, const uint16_t* Qparts, int64_t M, int32_t K, int32_t N, float* C, int64_t ldc, int32_t S,
            const X3Epi& e, void* stream);
}  // namespace

MREC_API int mrec_x3_gemm(int form, const uint16_t* Pparts, const uint16_t* Qparts, int64_t M, int32_t K, int32_t N, float* C, int64_t ldc,
                          int32_t S, void* stream) {
    return x3_gemm(form, Pparts, Qparts, M, K, N, C, ldc, S, X3Epi{}, stream);
}

MREC_API int mrec_x3_gemm_fwd(const uint16_t* xparts, const uint16_t* wparts, int64_t M, int32_t K, int32_t N, float* y, int64_t ldy,
                              const float* bias, int relu, const mrec_dropout_t* drop_next, uint16_t* parts_out, void* stream) {
    if (parts_out && !al16(parts_out)) return MREC_EUNSUPPORTED;
    X3Epi e{1, bias, relu, nullptr, 0, 1.0f, nullptr, parts_out, DropArgs{}};
    if (!drop_from(drop_next, N, &e.drop)) return MREC_EINVAL;
    return x3_gemm(0, xparts, wparts, M, K, N, y, ldy, 1, e, stream);
}

namespace {
// The plain input gradient whose width leaves a narrow last column tile AND whose full tiles fit one round fewer over the 256 CUs
// without it: S > 0 = reduction slabs for the narrow tile (through a workspace), 0 = one ordinary launch.
struct DgradSplit { int S; int c0, rem, ldw; };
DgradSplit dgrad_split(int64_t M, int32_t K, int32_t N) {
    DgradSplit d{0, 0, 0, 0};
    static const int off = [] { const char* e = getenv("MREC_X3_NOSPLIT"); return e ? atoi(e) : 0; }();      // (tools/probes/x3_bench.py)
    if (off) return d;
    const int64_t nTp = mrec_cdiv(M, 256), nTq = mrec_cdiv(K, 256);
    const int rem = (int)(K - 256 * (nTq - 1));
    if (nTq < 2 || rem >= 224 || mrec_cdiv(nTp * nTq, 256) == mrec_cdiv(nTp * (nTq - 1), 256)) return d;
    int64_t S = 256 / nTp;
    S = S > 4 ? 4 : S;
    const int64_t T = 6 * (up64(N) / 64);
    while (S > 1 && mrec_cdiv(T, S) * (S - 1) >= T) --S;
    if (S < 2) return d;
    d.S = (int)S; d.c0 = (int)(256 * (nTq - 1)); d.rem = rem; d.ldw = (rem + 3) / 4 * 4;
    return d;
}
int x3_dgrad_split(const uint16_t* dyparts, const uint16_t* wparts, int64_t M, int32_t K, int32_t N, float* dx, int64_t lddx, float* ws,
                   const DgradSplit& d, void* stream);
}  // namespace

MREC_API int mrec_x3_gemm_dgrad_workspace_bytes(int64_t M, int32_t K, int32_t N, size_t* out) {
    if (!out || M <= 0 || K <= 0 || N <= 0) return MREC_EINVAL;
    const DgradSplit d = dgrad_split(M, K, N);
    *out = d.S ? (size_t)d.S * M * d.ldw * sizeof(float) + 256 : 0;
    return MREC_OK;
}

MREC_API int mrec_x3_gemm_dgrad(const uint16_t* dyparts, const uint16_t* wparts, int64_t M, int32_t K, int32_t N, float* dx, int64_t lddx,
                                const float* h, int64_t ldh, float scale, float* colsum, uint16_t* parts_out, void* ws, size_t ws_bytes,
                                void* stream) {
    if ((parts_out && !al16(parts_out)) || (h && (ldh < K || ldh % 2 || (((uintptr_t)h) & 7)))) return MREC_EUNSUPPORTED;
    if (!(scale > 0.0f)) return MREC_EINVAL;
    if (!h && !colsum && !parts_out && scale == 1.0f && ws && M > 0 && K > 0 && N > 0) {
        const DgradSplit d = dgrad_split(M, K, N);
        if (d.S && al16(ws) && ws_bytes >= (size_t)d.S * M * d.ldw * sizeof(float))
            return x3_dgrad_split(dyparts, wparts, M, K, N, dx, lddx, (float*)ws, d, stream);
    }
    return x3_gemm(1, dyparts, wparts, M, K, N, dx, lddx, 1, X3Epi{2, nullptr, 0, h, ldh, scale, colsum, parts_out, DropArgs{}}, stream);
}

namespace {
int x3_gemm(int form, const uint16_t* Pparts, const uint16_t* Qparts, int64_t M, int32_t K, int32_t N, float* C, int64_t ldc, int32_t S,
            const X3Epi& e, void* stream) {
    if (form < 0 || form > 2 || M <= 0 || K <= 0 || N <= 0 || !Pparts || !Qparts || !C || S <= 0) return MREC_EINVAL;
    if (!al16(Pparts) || !al16(Qparts) || (((uintptr_t)C) & 7) || ldc % 2) return MREC_EUNSUPPORTED;
    const int64_t Mp = up64(M), Kp = up64(K), Np = up64(N);
    Args a{};
    a.P = Pparts; a.Q = Qparts; a.C = C; a.ldc = ldc;
    int64_t partP, partQ, red;                    // elements per part of P / Q; the reduction extent (padded)
    if (form == 0) {          // P = x parts [Mp][Kp], Q = w parts [Kp][Np]
        a.ldp = Kp; a.ldq = Np; a.Pext = (int)M; a.Qext = N; red = Kp; partP = Mp * Kp; partQ = Kp * Np;
        if (ldc < N || S != 1) return MREC_EINVAL;
    } else if (form == 1) {   // P = dy parts [Mp][Np], Q = w parts [Kp][Np] (rows = outputs)
        a.ldp = Np; a.ldq = Np; a.Pext = (int)M; a.Qext = K; red = Np; partP = Mp * Np; partQ = Kp * Np;
        if (ldc < K || S != 1) return MREC_EINVAL;
    } else {                  // P = x parts [Mp][Kp] (reduction over rows), Q = dy parts [Mp][Np]
        a.ldp = Kp; a.ldq = Np; a.Pext = K; a.Qext = N; red = Mp; partP = Mp * Kp; partQ = Mp * Np;
        if (ldc < N) return MREC_EINVAL;
        a.slab_stride = (int64_t)K * ldc;
    }
    if (3 * partP * 2 >= (int64_t(1) << 31) || 3 * partQ * 2 >= (int64_t(1) << 31)) return MREC_EUNSUPPORTED;
    a.K = (int)(6 * red);
    a.seg_tiles = (int)(red / 64);
    static const int inter = [] { const char* e = getenv("MREC_X3_INTER"); return e ? atoi(e) : 1; }();
    a.seg_inter = inter;
    a.seg_codeP = kCodeP; a.seg_codeQ = kCodeQ; a.seg_partP = (uint32_t)(partP * 2); a.seg_partQ = (uint32_t)(partQ * 2);
    a.rangeP = 3 * partP * 2; a.rangeQ = 3 * partQ * 2;
    const int Ttot = 6 * a.seg_tiles;
    a.kt_per_slab = (Ttot + S - 1) / S;
    if ((int64_t)a.kt_per_slab * (S - 1) >= Ttot && S > 1) return MREC_EINVAL;
    // 256- or 128-row tiles: whichever fills the 256 CUs in fewer tile-times (a 128-row tile costs ~0.55 of a 256-row one; the input
    // gradient into 1170 columns is 320 big tiles = two rounds at 62 %, or 640 small ones = 2.5 rounds of half the length)
    const int64_t t8 = mrec_cdiv(a.Pext, 256) * mrec_cdiv(a.Qext, 256) * S, t4 = mrec_cdiv(a.Pext, 128) * mrec_cdiv(a.Qext, 256) * S;
    int mr = (double)mrec_cdiv(t4, 256) * 0.55 < (double)mrec_cdiv(t8, 256) ? 4 : 8;
    static const int force_mr = [] { const char* e = getenv("MREC_X3_MR"); return e ? atoi(e) : 0; }();      // (tools/probes/x3_bench.py)
    if (force_mr == 4 || force_mr == 8) mr = force_mr;
    a.nTp = (int)mrec_cdiv(a.Pext, mr * 32); a.nTq = (int)mrec_cdiv(a.Qext, 256);
    const unsigned grid = (unsigned)(a.nTp * a.nTq * S);
    hipStream_t st = (hipStream_t)stream;
#define MREC_X3(PT, QT, EPI)                                                                                         \
    do {                                                                                                              \
        if (mr == 8) mgemm::k_gemm256<PT, QT, EPI, false, 4, 8><<<grid, mgemm::kThreads, 0, st>>>(a);                 \
        else mgemm::k_gemm256<PT, QT, EPI, false, 4, 4><<<grid, mgemm::kThreads, 0, st>>>(a);                         \
    } while (0)
    if (e.mode != 0) {            // the layer's output end in the GEMM's epilogue (form 0: bias + ReLU; 1: ReLU mask, scale, bias gradient)
        if (form == 2 || (e.mode == 1) != (form == 0)) return MREC_EINVAL;
        a.x3_mode = e.mode; a.bias = e.bias; a.relu = e.relu; a.H = e.h; a.ldh = e.ldh; a.x3_scale = e.scale; a.colsum_ws = e.colsum;
        a.parts = e.parts; a.drop = e.drop;
        const int64_t outc = form == 0 ? N : K;
        a.parts_ld = up64(outc); a.parts_stride = Mp * up64(outc);
        if (form == 0) MREC_X3(false, true, mgemm::EPI_X3);
        else MREC_X3(false, false, mgemm::EPI_X3);
    } else if (form == 0) MREC_X3(false, true, mgemm::EPI_F32);
    else if (form == 1) MREC_X3(false, false, mgemm::EPI_F32);
    else MREC_X3(true, true, mgemm::EPI_F32);
#undef MREC_X3
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}

int x3_dgrad_split(const uint16_t* dyparts, const uint16_t* wparts, int64_t M, int32_t K, int32_t N, float* dx, int64_t lddx, float* ws,
                   const DgradSplit& d, void* stream) {
    if (!dyparts || !wparts || !dx || lddx < K) return MREC_EINVAL;
    if (!al16(dyparts) || !al16(wparts) || (((uintptr_t)dx) & 7) || lddx % 2) return MREC_EUNSUPPORTED;
    const int64_t Mp = up64(M), Kp = up64(K), Np = up64(N);
    const int64_t partP = Mp * Np, partQ = Kp * Np;
    if (3 * partP * 2 >= (int64_t(1) << 31) || 3 * partQ * 2 >= (int64_t(1) << 31)) return MREC_EUNSUPPORTED;
    Args a{};
    a.P = dyparts; a.Q = wparts; a.C = dx; a.ldc = lddx;
    a.ldp = Np; a.ldq = Np; a.Pext = (int)M; a.Qext = d.c0;
    a.K = (int)(6 * Np); a.seg_tiles = (int)(Np / 64); a.seg_inter = 1;
    a.seg_codeP = kCodeP; a.seg_codeQ = kCodeQ; a.seg_partP = (uint32_t)(partP * 2); a.seg_partQ = (uint32_t)(partQ * 2);
    a.rangeP = 3 * partP * 2; a.rangeQ = 3 * partQ * 2;
    const int Ttot = 6 * a.seg_tiles;
    a.kt_per_slab = Ttot;
    a.nTp = (int)mrec_cdiv(M, 256); a.nTq = d.c0 / 256;
    Args b = a;                     // the narrow tile: rows c0.. of every part of w, S slabs of the reduction into the workspace
    b.Q = wparts + (int64_t)d.c0 * Np; b.rangeQ = a.rangeQ - (int64_t)d.c0 * Np * 2;
    b.C = ws; b.ldc = d.ldw; b.slab_stride = M * (int64_t)d.ldw;
    b.Qext = d.rem; b.nTq = 1;
    b.kt_per_slab = (Ttot + d.S - 1) / d.S;
    const int n1 = a.nTp * a.nTq, n2 = b.nTp * d.S;
    hipStream_t st = (hipStream_t)stream;
    k_x3_pair<false, false, 8><<<(unsigned)(n1 + n2), mgemm::kThreads, 0, st>>>(a, b, n1);
    const int64_t items = M * (d.ldw / 4);
    k_x3_fold_cols<<<(unsigned)mrec_cdiv(items, 256), 256, 0, st>>>(ws, d.S, M, d.rem, d.ldw, dx, lddx, d.c0);
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}
}  // namespace
