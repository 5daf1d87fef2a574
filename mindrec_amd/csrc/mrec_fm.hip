// mrec_fm.hip -- DeepFM second-order (FM) term over the gathered, masked embeddings, for gfx950.
//
// Reference: DeepFMModel.construct, models/deepfm/src/deepfm.py:221-228
//     vx = V[ids] * mask;  fm_out = 0.5 * ReduceSum( Square(ReduceSum(vx, 1)) - ReduceSum(Square(vx), 1), 1 )
// MindSpore runs two Squares, three ReduceSums and a Sub over the [B, F, D] tensor; here one wave64
// owns a sample, keeps the D column sums in registers while it streams the sample's F rows once, and
// writes fm_out[b] plus the column sums (the only thing the backward needs besides vx):
//     d fm / d vx[b, f, d] = sum_f' vx[b, f', d] - vx[b, f, d].
// HBM-bound: B*F*D*4 bytes read forward; backward reads vx, adds into the MLP's input gradient.
#include "mrec_common.h"

namespace {

constexpr int FM_MAXC = 4;   // columns per lane: D <= 256

__device__ __forceinline__ float wave_sum_f(float x) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d, 64);
    return x;
}

__global__ __launch_bounds__(256) void k_fm_fwd(const float* __restrict__ vx, int64_t B, int F, int D,
                                                float* __restrict__ fm_out, float* __restrict__ colsum,
                                                const float* __restrict__ add = nullptr) {
    const int lane = threadIdx.x & 63;
    const int64_t nw = (int64_t)gridDim.x * 4;
    for (int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); b < B; b += nw) {
        float s[FM_MAXC], q[FM_MAXC];
#pragma unroll
        for (int j = 0; j < FM_MAXC; ++j) { s[j] = 0.0f; q[j] = 0.0f; }
        const float* row = vx + b * (int64_t)F * D;
        for (int f = 0; f < F; ++f) {
#pragma unroll
            for (int j = 0; j < FM_MAXC; ++j) {
                const int c = lane + 64 * j;
                if (c < D) {
                    const float x = row[(int64_t)f * D + c];
                    s[j] = s[j] + x;          // ReduceSum(vx, 1), field order
                    q[j] = q[j] + x * x;      // ReduceSum(Square(vx), 1)
                }
            }
        }
        float part = 0.0f;
#pragma unroll
        for (int j = 0; j < FM_MAXC; ++j) {
            const int c = lane + 64 * j;
            if (c < D) {
                part += s[j] * s[j] - q[j];
                colsum[b * D + c] = s[j];
            }
        }
        const float tot = wave_sum_f(part);
        if (lane == 0) fm_out[b] = add ? add[b] + 0.5f * tot : 0.5f * tot;
    }
}

// 16-byte-lane variant (D % 4 == 0, 16-byte aligned rows): a lane-group of lpr = D/4 lanes owns a sample (three
// samples per wave64 at D = 80), four fields' loads are issued before they are consumed, no per-column guards.
// Column sums are accumulated in field order exactly as above (colsum stays bit-identical); the sum over the
// columns is a fixed shuffle tree inside the lane-group.
// KIND16 1 / 2: the rows also leave as a bf16 / f16 copy (x16 [B, F, D]) -- the dense net's 16-bit input, rounded from the very values
// the FM term reads (DeepFM looked the table up a second time for it).
__device__ __forceinline__ uint2 fm_pack16(const float4 x, const int kind) {
    if (kind == 1) {
        typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
        const bf2 lo = {(__bf16)x.x, (__bf16)x.y}, hi = {(__bf16)x.z, (__bf16)x.w};
        return make_uint2(__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi));
    }
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 lo = {(_Float16)x.x, (_Float16)x.y}, hi = {(_Float16)x.z, (_Float16)x.w};
    return make_uint2(__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi));
}

__global__ __launch_bounds__(256) void k_fm_fwd4(const float4* __restrict__ vx, int64_t B, int F, int lpr, int G,
                                                 float* __restrict__ fm_out, float4* __restrict__ colsum,
                                                 const float* __restrict__ add = nullptr, uint2* __restrict__ x16 = nullptr, int kind16 = 0) {
    const int lane = threadIdx.x & 63;
    const int grp = lane / lpr, sub = lane - grp * lpr;
    const bool act = grp < G;
    const int64_t ng = (int64_t)gridDim.x * 4 * G;
    for (int64_t b0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * G; b0 < B; b0 += ng) {
        const int64_t b = b0 + grp;
        const bool live = act && b < B;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = s;
        const float4* row = vx + (live ? b : 0) * (int64_t)F * lpr + sub;
        int f = 0;
        for (; f + 4 <= F; f += 4) {
            float4 x[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) x[k] = live ? row[(int64_t)(f + k) * lpr] : make_float4(0.f, 0.f, 0.f, 0.f);
            if (x16 && live) {
#pragma unroll
                for (int k = 0; k < 4; ++k) x16[(b * F + f + k) * (int64_t)lpr + sub] = fm_pack16(x[k], kind16);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                s.x = s.x + x[k].x; s.y = s.y + x[k].y; s.z = s.z + x[k].z; s.w = s.w + x[k].w;
                q.x = q.x + x[k].x * x[k].x; q.y = q.y + x[k].y * x[k].y; q.z = q.z + x[k].z * x[k].z; q.w = q.w + x[k].w * x[k].w;
            }
        }
        for (; f < F; ++f) {
            const float4 x = live ? row[(int64_t)f * lpr] : make_float4(0.f, 0.f, 0.f, 0.f);
            if (x16 && live) x16[(b * F + f) * (int64_t)lpr + sub] = fm_pack16(x, kind16);
            s.x = s.x + x.x; s.y = s.y + x.y; s.z = s.z + x.z; s.w = s.w + x.w;
            q.x = q.x + x.x * x.x; q.y = q.y + x.y * x.y; q.z = q.z + x.z * x.z; q.w = q.w + x.w * x.w;
        }
        float part = ((s.x * s.x - q.x) + (s.y * s.y - q.y)) + ((s.z * s.z - q.z) + (s.w * s.w - q.w));
        if (live) colsum[b * lpr + sub] = s;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float o = __shfl_down(part, off, 64);
            if (sub + off < lpr) part += o;
        }
        if (live && sub == 0) fm_out[b] = add ? add[b] + 0.5f * part : 0.5f * part;
    }
}

// g[b, f, d] += dout[b] * (colsum[b, d] - vx[b, f, d])
__global__ __launch_bounds__(256) void k_fm_bwd(const float* __restrict__ vx, const float* __restrict__ colsum,
                                                const float* __restrict__ dout, int64_t B, int F, int D,
                                                float* __restrict__ g) {
    const int64_t total = B * (int64_t)F * D;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / ((int64_t)F * D);
        const int d = (int)(i % D);
        g[i] = g[i] + dout[b] * (colsum[b * D + d] - vx[i]);
    }
}

// float4 variant with 32-bit index arithmetic (B*F*D/4 < 2^31)
__global__ __launch_bounds__(256) void k_fm_bwd4(const float4* __restrict__ vx, const float4* __restrict__ colsum,
                                                 const float* __restrict__ dout, unsigned total4, unsigned FD4, unsigned D4,
                                                 float4* __restrict__ g) {
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total4; i += gridDim.x * 256u) {
        const unsigned b = i / FD4;
        const unsigned d4 = i % D4;
        const float w = dout[b];
        const float4 c = colsum[b * D4 + d4], x = vx[i];
        float4 y = g[i];
        y.x = y.x + w * (c.x - x.x); y.y = y.y + w * (c.y - x.y); y.z = y.z + w * (c.z - x.z); y.w = y.w + w * (c.w - x.w);
        g[i] = y;
    }
}

// g_out[i] = widen(g16[i]) + dout[b] * (colsum[b, d] - vx[i]): the FM term's gradient added onto the 16-bit input gradient of the
// mixed-precision MLP (DenseLayer with convert_dtype, deepfm.py:135-145), written as the fp32 row gradient of the lookup
template <bool F16>
__global__ __launch_bounds__(256) void k_fm_bwd4_mix(const float4* __restrict__ vx, const float4* __restrict__ colsum,
                                                     const float* __restrict__ dout, const uint2* __restrict__ g16, unsigned total4,
                                                     unsigned FD4, unsigned D4, float4* __restrict__ g) {
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total4; i += gridDim.x * 256u) {
        const unsigned b = i / FD4;
        const unsigned d4 = i % D4;
        const float w = dout[b];
        const float4 c = colsum[b * D4 + d4], x = vx[i];
        const uint2 u = g16[i];
        float4 y;
        if (F16) {
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            const h2 a = __builtin_bit_cast(h2, u.x), bb = __builtin_bit_cast(h2, u.y);
            y = make_float4((float)a[0], (float)a[1], (float)bb[0], (float)bb[1]);
        } else {
            y = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16),
                            __uint_as_float(u.y & 0xFFFF0000u));
        }
        y.x = y.x + w * (c.x - x.x); y.y = y.y + w * (c.y - x.y); y.z = y.z + w * (c.z - x.z); y.w = y.w + w * (c.w - x.w);
        g[i] = y;
    }
}

__global__ __launch_bounds__(256) void k_scatter_add_rows(float* __restrict__ table, int64_t ld, int D,
                                                          const int* __restrict__ rows, int64_t n,
                                                          const float* __restrict__ vals) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const int r = rows[i];
    if (r < 0) return;
    for (int c = lane; c < D; c += 64) table[(int64_t)r * ld + c] += vals[i * D + c];
}

}  // namespace

MREC_API int mrec_fm_fwd_f32(const float* vx, int64_t B, int32_t F, int32_t D, float* fm_out, float* colsum,
                             void* stream) {
    return mrec_fm_fwd_add_f32(vx, B, F, D, nullptr, fm_out, colsum, stream);
}

MREC_API int mrec_fm_fwd_add_f32(const float* vx, int64_t B, int32_t F, int32_t D, const float* addend, float* fm_out, float* colsum,
                                 void* stream) {
    return mrec_fm_fwd_add16_f32(vx, B, F, D, addend, fm_out, colsum, nullptr, 0, stream);
}

MREC_API int mrec_fm_fwd_add16_f32(const float* vx, int64_t B, int32_t F, int32_t D, const float* addend, float* fm_out, float* colsum,
                                   void* x16, int32_t kind16, void* stream) {
    if (B < 0 || F <= 0 || D <= 0 || (x16 && kind16 != 1 && kind16 != 2)) return MREC_EINVAL;
    if (x16 && (D % 4 || (((uintptr_t)x16) & 7) || ((((uintptr_t)vx) | ((uintptr_t)colsum)) & 15))) return MREC_EUNSUPPORTED;
    if (D > 64 * FM_MAXC) return MREC_EUNSUPPORTED;
    if (B == 0) return MREC_OK;
    if (!vx || !fm_out || !colsum) return MREC_EINVAL;
    const bool al = ((((uintptr_t)vx) | ((uintptr_t)colsum)) & 15) == 0;
    if (D % 4 == 0 && al) {
        const int lpr = D / 4, G = 64 / lpr;
        int64_t blocks = mrec_cdiv(B, (int64_t)4 * G);
        if (blocks > 256 * 8) blocks = 256 * 8;
        k_fm_fwd4<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>((const float4*)vx, B, F, lpr, G, fm_out, (float4*)colsum, addend,
                                                                      (uint2*)x16, kind16);
    } else {
        int64_t blocks = mrec_cdiv(B, 4);
        if (blocks > 256 * 8) blocks = 256 * 8;
        k_fm_fwd<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(vx, B, F, D, fm_out, colsum, addend);
    }
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}

MREC_API int mrec_fm_bwd_f32(const float* vx, const float* colsum, const float* dout, int64_t B, int32_t F, int32_t D,
                             float* g, void* stream) {
    if (B < 0 || F <= 0 || D <= 0) return MREC_EINVAL;
    if (B == 0) return MREC_OK;
    if (!vx || !colsum || !dout || !g) return MREC_EINVAL;
    const int64_t total = B * (int64_t)F * D;
    const bool al = ((((uintptr_t)vx) | ((uintptr_t)colsum) | ((uintptr_t)g)) & 15) == 0;
    if (D % 4 == 0 && al && total / 4 < (int64_t(1) << 31)) {
        int64_t blocks = mrec_cdiv(total / 4, 256);
        if (blocks > 256 * 16) blocks = 256 * 16;
        k_fm_bwd4<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>((const float4*)vx, (const float4*)colsum, dout,
                                                                     (unsigned)(total / 4), (unsigned)(F * (D / 4)),
                                                                     (unsigned)(D / 4), (float4*)g);
    } else {
        int64_t blocks = mrec_cdiv(total, 256);
        if (blocks > 256 * 16) blocks = 256 * 16;
        k_fm_bwd<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(vx, colsum, dout, B, F, D, g);
    }
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}

MREC_API int mrec_fm_bwd_mix_f32(const float* vx, const float* colsum, const float* dout, const void* g16, int32_t g16_kind, int64_t B,
                                 int32_t F, int32_t D, float* g_out, void* stream) {
    if (B < 0 || F <= 0 || D <= 0 || (g16_kind != 1 && g16_kind != 2)) return MREC_EINVAL;
    if (B == 0) return MREC_OK;
    if (!vx || !colsum || !dout || !g16 || !g_out) return MREC_EINVAL;
    const int64_t total = B * (int64_t)F * D;
    const bool al = ((((uintptr_t)vx) | ((uintptr_t)colsum) | ((uintptr_t)g_out)) & 15) == 0 && (((uintptr_t)g16) & 7) == 0;
    if (D % 4 || !al || total / 4 >= (int64_t(1) << 31)) return MREC_EUNSUPPORTED;
    int64_t blocks = mrec_cdiv(total / 4, 256);
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (g16_kind == 2)
        k_fm_bwd4_mix<true><<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>((const float4*)vx, (const float4*)colsum, dout, (const uint2*)g16,
                                                                              (unsigned)(total / 4), (unsigned)(F * (D / 4)), (unsigned)(D / 4), (float4*)g_out);
    else
        k_fm_bwd4_mix<false><<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>((const float4*)vx, (const float4*)colsum, dout, (const uint2*)g16,
                                                                               (unsigned)(total / 4), (unsigned)(F * (D / 4)), (unsigned)(D / 4), (float4*)g_out);
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}

MREC_API int mrec_scatter_add_rows_f32(float* table, int64_t ld, int32_t D, const int32_t* rows, int64_t n,
                                       const float* vals, void* stream) {
    if (n < 0 || D <= 0 || ld < D) return MREC_EINVAL;
    if (n == 0) return MREC_OK;
    if (!table || !rows || !vals) return MREC_EINVAL;
    k_scatter_add_rows<<<(unsigned)mrec_cdiv(n, 4), 256, 0, (hipStream_t)stream>>>(table, ld, D, rows, n, vals);
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}
