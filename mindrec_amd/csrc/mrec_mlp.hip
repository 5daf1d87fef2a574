// mrec_mlp.hip -- the elementwise / reduction ends of the Wide&Deep dense net, fused, for gfx950.
//
// The hidden DenseLayers (models/wide_deep/src/wide_and_deep.py:113-133) run on the matrix cores (mrec_dense.hip; their
// ReLU / BiasAdd bprops are GEMM epilogues there); what is left is the output end of the net, HBM-bound byte work that
// MindSpore runs as separate primitives (the 128 -> 1 output layer, wide + deep add :315,
// SigmoidCrossEntropyWithLogits + ReduceMean :352-354 and their bprops).  One kernel covers it:
//
//  k_head_fwd_bwd    : logit = h4 . W5 + b5 + wide;  loss terms;  dlogit = (sigmoid(logit) - y) * scale;
//                      dh4 = dlogit * W5 masked by h4 > 0;  dW5 += h4 * dlogit;  db5 += dlogit
//                      -- forward AND backward of the output layer and the loss in one pass over h4.
//
// It reduces over the batch with per-block partials written to a workspace and a second tiny kernel
// that adds the partials in block order: bitwise reproducible, no float atomics.
#include "mrec_common.h"
#include "mrec_dropout.h"
#include "mrec_mlp.h"

namespace {

// Output head.  K5 = width of the last hidden layer (multiple of 8, K5/8 a power of two <= 64).
// partial layout per block: [K5] dW5 partials, [K5] column sums of dh4 (= bias gradient of the last
// hidden layer), then db5 partial, then loss partial.
// the row's 8 values of this lane: 16 bytes of 16-bit values, or 32 bytes of fp32 (KIND 2: the fp32 net)
template <int KIND> __device__ __forceinline__ void load8k(const uint4* __restrict__ h, int64_t idx, float (&f)[8]) {
    if (KIND == 2) {
        const float4 a = ((const float4*)h)[2 * idx], b = ((const float4*)h)[2 * idx + 1];
        f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
    } else {
        unpack8t<KIND == 1>(h[idx], f);
    }
}
template <int KIND> __device__ __forceinline__ void store8k(uint4* __restrict__ h, int64_t idx, const float (&f)[8]) {
    if (KIND == 2) {
        ((float4*)h)[2 * idx] = make_float4(f[0], f[1], f[2], f[3]);
        ((float4*)h)[2 * idx + 1] = make_float4(f[4], f[5], f[6], f[7]);
    } else {
        h[idx] = pack8t<KIND == 1>(f);
    }
}

template <int KIND>      // 0: bfloat16, 1: IEEE half, 2: fp32 activations
__global__ __launch_bounds__(MB) void k_head_fwd_bwd(const uint4* __restrict__ h4, const float* __restrict__ w5,
                                                     const float* __restrict__ b5, const float* __restrict__ wide,
                                                     const float* __restrict__ label, int64_t B, int CG,
                                                     int rows_per_block, float dscale, float* __restrict__ logit_out,
                                                     float* __restrict__ dlogit_out, uint4* __restrict__ dh4,
                                                     float* __restrict__ partial, const float* __restrict__ wprod, int F,
                                                     const float* __restrict__ wide_bias, float dh_scale) {
    __shared__ float red[MB][8];
    __shared__ float red2[MB][2];
    const int cg = threadIdx.x % CG, rl = threadIdx.x / CG, RP = MB / CG;
    const int64_t r_begin = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r_end = (r_begin + rows_per_block < B) ? r_begin + rows_per_block : B;
    float w[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) w[k] = w5[cg * 8 + k];
    const float bias = *b5;
    float accw[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float accd[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float accb = 0.0f, accl = 0.0f;
    // all threads of a row-group iterate together (the shuffles below need the CG lanes of a row converged)
    for (int64_t r0 = r_begin; r0 < r_end; r0 += RP) {
        const int64_t r = r0 + rl;
        const bool valid = r < r_end;
        float fh[8];
        if (valid) load8k<KIND>(h4, r * CG + cg, fh);
        else {
#pragma unroll
            for (int k = 0; k < 8; ++k) fh[k] = 0.0f;
        }
        float part = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) part += fh[k] * w[k];
        // reduce over the CG lanes of this row (CG is a power of two <= 64, lanes of a row are adjacent)
        for (int d = CG >> 1; d >= 1; d >>= 1) part += __shfl_xor(part, d, 64);
        // the wide branch: either given per sample, or as the per-field products the fused lookup wrote ([B, F, 2]: product,
        // pad).  The CG lanes of the row load the products side by side (one pass of loads, not F dependent ones), then every
        // lane adds them up in FIELD ORDER through shuffles -- ReduceSum over the fields, then + Wide_b: the same adds in the
        // same order as mrec_wide_sum / the CPU restatement.
        float wv = 0.0f;
        if (wprod) {
            const int lane_row0 = (threadIdx.x & 63) - cg;           // first lane of this row's group
            float acc_w = 0.0f;
            for (int f0 = 0; f0 < F; f0 += CG) {
                const int f = f0 + cg;
                const float mine = (valid && f < F) ? wprod[2 * (r * F + f)] : 0.0f;
                const int nf = (F - f0 < CG) ? F - f0 : CG;
                for (int c = 0; c < nf; ++c) acc_w = acc_w + __shfl(mine, lane_row0 + c, 64);
            }
            wv = acc_w + *wide_bias;
        } else if (valid) {
            wv = wide[r];
        }
        float dl = 0.0f;
        if (valid) {
            float wsum = wv;
            const float z = part + bias + wsum;
            const float y = label[r];
            // SigmoidCrossEntropyWithLogits: max(z,0) - z*y + log(1 + exp(-|z|))
            const float loss = fmaxf(z, 0.0f) - z * y + log1pf(expf(-fabsf(z)));
            const float sg = 1.0f / (1.0f + expf(-z));
            dl = (sg - y) * dscale;
            if (cg == 0) {
                logit_out[r] = z;
                dlogit_out[r] = dl;
                accl += loss;
                accb += dl;
            }
            float o[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                // d h4 through the last layer, masked by its ReLU (and, where h4 went through Dropout, by that mask: its zeros)
                o[k] = fh[k] > 0.0f ? (dl * w[k]) * dh_scale : 0.0f;
                accw[k] += fh[k] * dl;
                accd[k] += o[k];
            }
            store8k<KIND>(dh4, r * CG + cg, o);
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) red[threadIdx.x][k] = accw[k];
    red2[threadIdx.x][0] = accb;
    red2[threadIdx.x][1] = accl;
    __syncthreads();
    const int K5 = CG * 8;
    float* p = partial + (int64_t)blockIdx.x * (2 * K5 + 2);
    if (rl == 0) {
        float s[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) s[k] = red[cg][k];
        for (int q = 1; q < RP; ++q) {
#pragma unroll
            for (int k = 0; k < 8; ++k) s[k] += red[q * CG + cg][k];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) p[cg * 8 + k] = s[k];
    }
    if (threadIdx.x == 0) {
        float sb = 0.0f, sl = 0.0f;
        for (int q = 0; q < RP; ++q) { sb += red2[q * CG][0]; sl += red2[q * CG][1]; }
        p[2 * K5] = sb;
        p[2 * K5 + 1] = sl;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) red[threadIdx.x][k] = accd[k];
    __syncthreads();
    if (rl == 0) {
        float s[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) s[k] = red[cg][k];
        for (int q = 1; q < RP; ++q) {
#pragma unroll
            for (int k = 0; k < 8; ++k) s[k] += red[q * CG + cg][k];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) p[K5 + cg * 8 + k] = s[k];
    }
}

__global__ __launch_bounds__(MB) void k_head_finish(const float* __restrict__ partial, int nblk, int K5, float inv_B,
                                                    float* __restrict__ dw5, float* __restrict__ db4,
                                                    float* __restrict__ db5, float* __restrict__ loss,
                                                    float* __restrict__ dwide_b) {
    __shared__ float sm[8][32];
    const int W = 2 * K5 + 2;
    const int c = blockIdx.x * 32 + (threadIdx.x & 31);
    const float s = finish_column(partial, nblk, W, c, sm);
    if ((threadIdx.x >> 5) != 0 || c >= W) return;
    if (c < K5) dw5[c] = s;
    else if (c < 2 * K5) db4[c - K5] = s;
    else if (c == 2 * K5) {
        *db5 = s;
        if (dwide_b) *dwide_b = s;       // d loss / d Wide_b is the same sum of dlogit
    } else *loss = s * inv_B;
}

inline bool pow2(int x) { return x > 0 && (x & (x - 1)) == 0; }
inline int pick_blocks(int64_t B, int RP) {
    int64_t nb = 256;
    while (nb > 1 && B / nb < 4 * RP) nb >>= 1;   // at least a few passes per block
    return (int)nb;
}

}  // namespace

MREC_API int mrec_head_workspace_bytes(int64_t B, int32_t K5, size_t* out) {
    if (!out || B < 0 || K5 <= 0) return MREC_EINVAL;
    *out = (size_t)512 * (2 * K5 + 2) * sizeof(float) + 256;
    return MREC_OK;
}

static int head_impl(int kind, const void* h4, const float* w5, const float* b5, const float* wide,
                     const float* label, int64_t B, int32_t K5, float dscale, float dh_scale, float* logit,
                     float* dlogit, void* dh4, float* dw5, float* db4, float* db5, float* loss,
                     void* ws, size_t ws_bytes, void* stream, const float* wprod = nullptr, int F = 0,
                     const float* wide_bias = nullptr, float* dwide_b = nullptr) {
    if (B <= 0 || K5 <= 0 || !(dh_scale >= 1.0f)) return MREC_EINVAL;
    if (wprod && (F <= 0 || !wide_bias)) return MREC_EINVAL;
    if (!h4 || !w5 || !b5 || (!wide && !wprod) || !label || !logit || !dlogit || !dh4 || !dw5 || !db4 || !db5 || !loss || !ws) return MREC_EINVAL;
    if (K5 % 8 || !pow2(K5 / 8) || K5 / 8 > 64) return MREC_EUNSUPPORTED;
    if ((((uintptr_t)h4 | (uintptr_t)dh4) & 15) != 0) return MREC_EINVAL;
    const int CG = K5 / 8, RP = MB / CG;
    const int nblk = pick_blocks(B, RP);
    if (ws_bytes < (size_t)nblk * (2 * K5 + 2) * sizeof(float)) return MREC_EWORKSPACE;
    const int rows_per_block = (int)mrec_cdiv(mrec_cdiv(B, nblk), RP) * RP;
    const int nb = (int)mrec_cdiv(B, rows_per_block);
    hipStream_t st = (hipStream_t)stream;
#define MREC_HEAD_GO(KIND)                                                                                                 \
    k_head_fwd_bwd<KIND><<<nb, MB, 0, st>>>((const uint4*)h4, w5, b5, wide, label, B, CG, rows_per_block, dscale, logit, dlogit, \
                                            (uint4*)dh4, (float*)ws, wprod, F, wide_bias, dh_scale)
    if (kind == 2) MREC_HEAD_GO(2);
    else if (kind == 1) MREC_HEAD_GO(1);
    else MREC_HEAD_GO(0);
#undef MREC_HEAD_GO
    k_head_finish<<<(unsigned)mrec_cdiv(2 * K5 + 2, 32), MB, 0, st>>>((const float*)ws, nb, K5, 1.0f / (float)B, dw5, db4, db5, loss, dwide_b);
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}

MREC_API int mrec_head_fwd_bwd_bf16(const uint16_t* h4, const float* w5, const float* b5, const float* wide,
                                    const float* label, int64_t B, int32_t K5, float dscale, float dh_scale, float* logit,
                                    float* dlogit, uint16_t* dh4, float* dw5, float* db4, float* db5, float* loss,
                                    void* ws, size_t ws_bytes, void* stream) {
    return head_impl(0, h4, w5, b5, wide, label, B, K5, dscale, dh_scale, logit, dlogit, dh4, dw5, db4, db5, loss, ws, ws_bytes, stream);
}

MREC_API int mrec_head_fwd_bwd_f16(const uint16_t* h4, const float* w5, const float* b5, const float* wide,
                                   const float* label, int64_t B, int32_t K5, float dscale, float dh_scale, float* logit,
                                   float* dlogit, uint16_t* dh4, float* dw5, float* db4, float* db5, float* loss,
                                   void* ws, size_t ws_bytes, void* stream) {
    return head_impl(1, h4, w5, b5, wide, label, B, K5, dscale, dh_scale, logit, dlogit, dh4, dw5, db4, db5, loss, ws, ws_bytes, stream);
}

/* fp32 activations (the fp32 net: mlp_dtype = fp32 of the Wide&Deep / DeepFM engines): h4, dh4 float32 [B, K5] */
MREC_API int mrec_head_fwd_bwd_f32(const float* h4, const float* w5, const float* b5, const float* wide,
                                   const float* label, int64_t B, int32_t K5, float dscale, float dh_scale, float* logit,
                                   float* dlogit, float* dh4, float* dw5, float* db4, float* db5, float* loss,
                                   void* ws, size_t ws_bytes, void* stream) {
    return head_impl(2, h4, w5, b5, wide, label, B, K5, dscale, dh_scale, logit, dlogit, dh4, dw5, db4, db5, loss, ws, ws_bytes, stream);
}

/* The same head with the wide branch given as per-field products [B, F] + the wide bias (see include/mrec.h). */
MREC_API int mrec_head_fwd_bwd_wide(int32_t f16, const uint16_t* h4, const float* w5, const float* b5, const float* wide_prod,
                                    int32_t F, const float* wide_bias, const float* label, int64_t B, int32_t K5, float dscale,
                                    float dh_scale, float* logit, float* dlogit, uint16_t* dh4, float* dw5, float* db4, float* db5, float* dwide_bias,
                                    float* loss, void* ws, size_t ws_bytes, void* stream) {
    return head_impl(f16 != 0 ? 1 : 0, h4, w5, b5, nullptr, label, B, K5, dscale, dh_scale, logit, dlogit, dh4, dw5, db4, db5, loss, ws, ws_bytes,
                     stream, wide_prod, F, wide_bias, dwide_bias);
}

// ---- Dropout as a pass of its own (spec: mrec_dropout.h): the first DenseLayer's input (the looked-up rows) and the fp32 net --
namespace {
template <int KIND>      // 0: fp32, 1: bfloat16, 2: IEEE half
__global__ __launch_bounds__(256) void k_dropout(const void* __restrict__ x, int64_t ldx, void* __restrict__ y, int64_t ldy, int64_t M,
                                                 int W, DropArgs d) {
    const uint64_t key = drop_key(d);
    const int W4 = W >> 2;
    const int64_t nq = M * W4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nq; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / W4;
        const int c = (int)(i - r * W4) * 4;
        const uint64_t qd = drop_quad(key, d.row0 + r, W, c);
        float v[4];
        if (KIND == 0) {
            const float4 t = *(const float4*)((const float*)x + r * ldx + c);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else {
            const uint2 t = *(const uint2*)((const uint16_t*)x + r * ldx + c);
            const uint32_t b[4] = {t.x & 0xFFFFu, t.x >> 16, t.y & 0xFFFFu, t.y >> 16};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                v[j] = KIND == 1 ? __uint_as_float(b[j] << 16) : (float)__builtin_bit_cast(_Float16, (uint16_t)b[j]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = drop_keep(qd, j, d.thresh) ? v[j] * d.scale : 0.0f;
        if (KIND == 0) {
            *(float4*)((float*)y + r * ldy + c) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            uint32_t b[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (KIND == 1) b[j] = f2bf(v[j]);
                else b[j] = __builtin_bit_cast(uint16_t, (_Float16)v[j]);
            }
            *(uint2*)((uint16_t*)y + r * ldy + c) = make_uint2(b[0] | (b[1] << 16), b[2] | (b[3] << 16));
        }
    }
}
__global__ __launch_bounds__(256) void k_dropout_mask(float* __restrict__ mask, int64_t ld, int64_t M, int W, DropArgs d) {
    const uint64_t key = drop_key(d);
    const int W4 = W >> 2;
    const int64_t nq = M * W4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nq; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / W4;
        const int c = (int)(i - r * W4) * 4;
        const uint64_t qd = d.thresh ? drop_quad(key, d.row0 + r, W, c) : 0ull;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = d.thresh == 0 ? 1.0f : (drop_keep(qd, j, d.thresh) ? d.scale : 0.0f);
        *(float4*)(mask + r * ld + c) = make_float4(v[0], v[1], v[2], v[3]);
    }
}
}  // namespace

MREC_API int mrec_dropout(const void* x, int64_t ldx, void* y, int64_t ldy, int32_t kind, int64_t M, int32_t W,
                          const mrec_dropout_t* drop, void* stream) {
    if (M < 0 || W <= 0 || ldx < W || ldy < W || kind < 0 || kind > 2 || !drop) return MREC_EINVAL;
    DropArgs d;
    if (!drop_from(drop, W, &d)) return MREC_EINVAL;
    if (M == 0) return MREC_OK;
    if (!x || !y) return MREC_EINVAL;
    const int al = kind == 0 ? 15 : 7;
    if (ldx % 4 || ldy % 4 || (((uintptr_t)x | (uintptr_t)y) & al)) return MREC_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (d.thresh == 0) {             // keep_prob 1: the identity
        if (x != y) MREC_HIP_CHECK(hipMemcpy2DAsync(y, (size_t)ldy * (kind ? 2 : 4), x, (size_t)ldx * (kind ? 2 : 4),
                                                    (size_t)W * (kind ? 2 : 4), (size_t)M, hipMemcpyDeviceToDevice, st));
        return MREC_OK;
    }
    const int64_t nq = M * (W / 4);
    const unsigned g = (unsigned)(mrec_cdiv(nq, 256) < 8192 ? mrec_cdiv(nq, 256) : 8192);
    if (kind == 0) k_dropout<0><<<g, 256, 0, st>>>(x, ldx, y, ldy, M, W, d);
    else if (kind == 1) k_dropout<1><<<g, 256, 0, st>>>(x, ldx, y, ldy, M, W, d);
    else k_dropout<2><<<g, 256, 0, st>>>(x, ldx, y, ldy, M, W, d);
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}

MREC_API int mrec_dropout_mask_f32(float* mask, int64_t ld, int64_t M, int32_t W, const mrec_dropout_t* drop, void* stream) {
    if (M < 0 || W <= 0 || ld < W || !drop) return MREC_EINVAL;
    DropArgs d;
    if (!drop_from(drop, W, &d)) return MREC_EINVAL;
    if (M == 0) return MREC_OK;
    if (!mask) return MREC_EINVAL;
    if (ld % 4 || (((uintptr_t)mask) & 15)) return MREC_EUNSUPPORTED;
    const int64_t nq = M * (W / 4);
    const unsigned g = (unsigned)(mrec_cdiv(nq, 256) < 8192 ? mrec_cdiv(nq, 256) : 8192);
    k_dropout_mask<<<g, 256, 0, (hipStream_t)stream>>>(mask, ld, M, W, d);
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}
