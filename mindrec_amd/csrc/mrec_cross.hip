// mrec_cross.hip -- DCN-v1 cross layers, all L layers in one HBM pass, for gfx950.
//
// Reference: CrossLayer.construct, models/deep_and_cross/src/deep_and_cross.py:139-149
//     y = x0 * (x_l . w) + b + x_l          (w, b in R^D: a GEMV + rank-1 update, HBM-bound)
// applied six times in DeepCrossModel.construct (:300-306).  MindSpore runs 6 x (tensor_dot,
// MatMul, two adds), re-reading the [B, D] activations every layer; here one wave64 owns a row,
// keeps x0 and x_l in registers (D = 1170 -> 19 floats per lane) and streams x0 in / y out once:
// 2*B*D*4 bytes for the whole stack.
//
// Backward.  The stack is affine in x0 with per-row scalar coefficients:
//     x_l = a_l * x0 + beta_l,   a_l = 1 + sum_{l'<l} s_l',   beta_l = sum_{l'<l} b_l',  s_l = x_l . w_l
// so with t_l = gy_{l+1} . x0 (a scalar recursion t_l = dy.x0 + sum_{l'>l} t_l' (w_l'.x0)) and
// u_l = t_l * a_l:
//     dx0  = a_L * dy + sum_l u_l w_l
//     dw_l = sum_rows u_l x0 + beta_l * sum_rows t_l
//     db_l = colsum(dy) + sum_{l'>l} w_l' * sum_rows t_l'
// One pass reads x0 and dy, writes dx0, and keeps the [L, D] batch sums sum_rows u_l x0 in
// registers per wave; per-wave slabs are then added in wave order (bitwise reproducible).
#include "mrec_common.h"

namespace {

constexpr int LMAX = 8;

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d, 64);
    return x;
}

template <int NPL>
__global__ __launch_bounds__(256) void k_cross_fwd(const float* __restrict__ x0, const float* __restrict__ w,
                                                   const float* __restrict__ b, int L, int64_t B, int D,
                                                   float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t nw = (int64_t)gridDim.x * 4;
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < B; row += nw) {
        float x[NPL], xl[NPL];
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            const int c = lane + 64 * j;
            x[j] = c < D ? x0[row * D + c] : 0.0f;
            xl[j] = x[j];
        }
        for (int l = 0; l < L; ++l) {
            float part = 0.0f;
#pragma unroll
            for (int j = 0; j < NPL; ++j) {
                const int c = lane + 64 * j;
                if (c < D) part += xl[j] * w[l * D + c];
            }
            const float s = wave_sum(part);
#pragma unroll
            for (int j = 0; j < NPL; ++j) {
                const int c = lane + 64 * j;
                if (c < D) xl[j] = (x[j] * s + b[l * D + c]) + xl[j];
            }
        }
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            const int c = lane + 64 * j;
            if (c < D) out[row * D + c] = xl[j];
        }
    }
}

// slab layout per wave: [LMAX][D] sums of u_l*x0, then [D] colsum(dy), then [LMAX] sums of t_l.
__host__ __device__ inline int64_t slab_floats(int D) { return (int64_t)(LMAX + 1) * D + LMAX; }

template <int NPL>
__global__ __launch_bounds__(256) void k_cross_bwd(const float* __restrict__ x0, const float* __restrict__ w,
                                                   const float* __restrict__ b, int L, int64_t B, int D,
                                                   const float* __restrict__ dy, float* __restrict__ dx0,
                                                   float* __restrict__ slabs) {
    const int lane = threadIdx.x & 63;
    const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * 4;
    float accw[LMAX][NPL];
    float accd[NPL];
    float accT[LMAX];
#pragma unroll
    for (int l = 0; l < LMAX; ++l) {
        accT[l] = 0.0f;
#pragma unroll
        for (int j = 0; j < NPL; ++j) accw[l][j] = 0.0f;
    }
#pragma unroll
    for (int j = 0; j < NPL; ++j) accd[j] = 0.0f;

    for (int64_t row = wid; row < B; row += nw) {
        float x[NPL], xl[NPL], g[NPL];
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            const int c = lane + 64 * j;
            x[j] = c < D ? x0[row * D + c] : 0.0f;
            g[j] = c < D ? dy[row * D + c] : 0.0f;
            xl[j] = x[j];
        }
        float a[LMAX + 1], P[LMAX], t[LMAX];
        a[0] = 1.0f;
        float qp = 0.0f;
#pragma unroll
        for (int j = 0; j < NPL; ++j) qp += g[j] * x[j];
        const float q = wave_sum(qp);
#pragma unroll
        for (int l = 0; l < LMAX; ++l) {
            if (l < L) {
                float ps = 0.0f, pp = 0.0f;
#pragma unroll
                for (int j = 0; j < NPL; ++j) {
                    const int c = lane + 64 * j;
                    if (c < D) {
                        const float wv = w[l * D + c];
                        ps += xl[j] * wv;
                        pp += x[j] * wv;
                    }
                }
                const float s = wave_sum(ps);
                P[l] = wave_sum(pp);
                a[l + 1] = a[l] + s;
#pragma unroll
                for (int j = 0; j < NPL; ++j) {
                    const int c = lane + 64 * j;
                    if (c < D) xl[j] = (x[j] * s + b[l * D + c]) + xl[j];
                }
            } else {
                P[l] = 0.0f;
                a[l + 1] = a[l];
            }
        }
        // t_l = q + sum_{l' > l} t_l' P_l'
        float run = 0.0f;
#pragma unroll
        for (int l = LMAX - 1; l >= 0; --l) {
            if (l < L) {
                t[l] = q + run;
                run += t[l] * P[l];
            } else {
                t[l] = 0.0f;
            }
        }
        // dx0 = a_L * dy + sum_l u_l w_l ; accumulate batch sums
        float o[NPL];
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            o[j] = a[LMAX] * g[j];  // a[LMAX] == a[L]: layers past L add nothing
            accd[j] += g[j];
        }
#pragma unroll
        for (int l = 0; l < LMAX; ++l) {
            if (l < L) {
                const float u = t[l] * a[l];
                accT[l] += t[l];
#pragma unroll
                for (int j = 0; j < NPL; ++j) {
                    const int c = lane + 64 * j;
                    if (c < D) {
                        o[j] += u * w[l * D + c];
                        accw[l][j] += u * x[j];
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            const int c = lane + 64 * j;
            if (c < D) dx0[row * D + c] = o[j];
        }
    }
    float* sl = slabs + wid * slab_floats(D);
#pragma unroll
    for (int l = 0; l < LMAX; ++l) {
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            const int c = lane + 64 * j;
            if (c < D) sl[(int64_t)l * D + c] = accw[l][j];
        }
        if (lane == 0) sl[(int64_t)(LMAX + 1) * D + l] = accT[l];
    }
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
        const int c = lane + 64 * j;
        if (c < D) sl[(int64_t)LMAX * D + c] = accd[j];
    }
}

// One thread per column: adds the per-wave slabs in wave order, then composes dw and db.
__global__ __launch_bounds__(256) void k_cross_bwd_reduce(const float* __restrict__ slabs, int64_t nslabs,
                                                          const float* __restrict__ w, const float* __restrict__ b,
                                                          int L, int D, float* __restrict__ dw,
                                                          float* __restrict__ db) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= D) return;
    const int64_t sf = slab_floats(D);
    float sw[LMAX], T[LMAX], cs = 0.0f;
#pragma unroll
    for (int l = 0; l < LMAX; ++l) { sw[l] = 0.0f; T[l] = 0.0f; }
    for (int64_t k = 0; k < nslabs; ++k) {
        const float* sl = slabs + k * sf;
#pragma unroll
        for (int l = 0; l < LMAX; ++l) {
            if (l < L) {
                sw[l] += sl[(int64_t)l * D + c];
                T[l] += sl[(int64_t)(LMAX + 1) * D + l];
            }
        }
        cs += sl[(int64_t)LMAX * D + c];
    }
    float beta = 0.0f;  // beta_l[c]
    float tail = 0.0f;  // sum_{l' > l} w_l'[c] * T_l'
    float dbv[LMAX];
#pragma unroll
    for (int l = LMAX - 1; l >= 0; --l) {
        if (l < L) {
            dbv[l] = cs + tail;
            tail += w[l * D + c] * T[l];
        }
    }
#pragma unroll
    for (int l = 0; l < LMAX; ++l) {
        if (l < L) {
            dw[l * D + c] = sw[l] + beta * T[l];
            db[l * D + c] = dbv[l];
            beta += b[l * D + c];
        }
    }
}

inline int npl_bucket(int D) {
    const int need = (D + 63) / 64;
    const int buckets[] = {1, 2, 4, 8, 12, 16, 20, 24, 32};
    for (int v : buckets) if (need <= v) return v;
    return -1;
}

inline unsigned bwd_blocks(int64_t B) {
    int64_t blocks = mrec_cdiv(B, 4 * 8);  // >= 8 rows per wave
    if (blocks > 512) blocks = 512;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

}  // namespace

#define MREC_NPL_DISPATCH(NPLV, CALL)                                                              \
    switch (NPLV) {                                                                                \
        case 1: { constexpr int N_ = 1; CALL; } break;                                             \
        case 2: { constexpr int N_ = 2; CALL; } break;                                             \
        case 4: { constexpr int N_ = 4; CALL; } break;                                             \
        case 8: { constexpr int N_ = 8; CALL; } break;                                             \
        case 12: { constexpr int N_ = 12; CALL; } break;                                           \
        case 16: { constexpr int N_ = 16; CALL; } break;                                           \
        case 20: { constexpr int N_ = 20; CALL; } break;                                           \
        case 24: { constexpr int N_ = 24; CALL; } break;                                           \
        default: { constexpr int N_ = 32; CALL; } break;                                           \
    }

MREC_API int mrec_cross_layers_f32(const float* x0, const float* w, const float* b, int32_t L, int64_t B, int32_t D,
                                   float* out, void* stream) {
    if (B < 0 || D <= 0 || L < 0) return MREC_EINVAL;
    if (B == 0) return MREC_OK;
    if (!x0 || !out || (L > 0 && (!w || !b))) return MREC_EINVAL;
    const int npl = npl_bucket(D);
    if (npl < 0) return MREC_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    int64_t blocks = mrec_cdiv(B, 4);
    if (blocks > 256 * 8) blocks = 256 * 8;
    MREC_NPL_DISPATCH(npl, (k_cross_fwd<N_><<<(unsigned)blocks, 256, 0, st>>>(x0, w, b, L, B, D, out)));
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}

MREC_API int mrec_cross_layers_bwd_workspace_bytes(int32_t L, int64_t B, int32_t D, size_t* out) {
    if (!out || B < 0 || D <= 0 || L < 0) return MREC_EINVAL;
    *out = (size_t)bwd_blocks(B) * 4 * slab_floats(D) * sizeof(float) + 256;
    return MREC_OK;
}

MREC_API int mrec_cross_layers_bwd_f32(const float* x0, const float* w, const float* b, int32_t L, int64_t B,
                                       int32_t D, const float* dy, float* dx0, float* dw, float* db, void* ws,
                                       size_t ws_bytes, void* stream) {
    if (B < 0 || D <= 0 || L < 0) return MREC_EINVAL;
    if (L > LMAX) return MREC_EUNSUPPORTED;
    if (!x0 || !dy || !dx0 || !ws || (L > 0 && (!w || !b || !dw || !db))) return MREC_EINVAL;
    const int npl = npl_bucket(D);
    if (npl < 0) return MREC_EUNSUPPORTED;
    const unsigned blocks = bwd_blocks(B);
    const int64_t nslabs = (int64_t)blocks * 4;
    if (ws_bytes < (size_t)nslabs * slab_floats(D) * sizeof(float)) return MREC_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* slabs = (float*)ws;
    MREC_NPL_DISPATCH(npl, (k_cross_bwd<N_><<<blocks, 256, 0, st>>>(x0, w, b, L, B, D, dy, dx0, slabs)));
    if (L > 0)
        k_cross_bwd_reduce<<<(unsigned)mrec_cdiv(D, 256), 256, 0, st>>>(slabs, nslabs, w, b, L, D, dw, db);
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}
