// mrec_dcn.hip -- the output end of Deep&Cross in one pass (models/deep_and_cross/src/deep_and_cross.py:306-309 + the loss
// :326-331): logit = concat([deep, cross]) . W3 + b3 without materialising the [B, 2194] concat, SigmoidCrossEntropyWithLogits +
// ReduceMean, and every bprop that hangs off the logit:
//     dlogit = (sigmoid(logit) - label) * dscale
//     dd2 = dlogit * W3[:H] where d2 > 0        (through the ReLU of dense_layer_2: the gradient at its pre-activation)
//     dc  = dlogit * W3[H:]                     (the cross stack's output gradient)
//     dW3 = [d2 ; c]^T . dlogit,  db3 = sum dlogit,  db2 = column sums of dd2  (dense_layer_2's BiasAdd bprop)
// HBM-bound: d2 and c are read once (a wave keeps a row in registers between the dot product and the bprops), dd2 and dc are
// written once.  A wave owns a run of rows; its lanes own fixed columns (float4 of d2, float2 of c: rows of c are 1170 floats,
// 8-byte aligned), so the per-column batch sums (dW3, db2) accumulate in registers in row order; the four waves of a
// workgroup, then the workgroups (a second launch, in workgroup order) are added in a fixed order: reproducible.
#include "mrec_common.h"

namespace dcn {

constexpr int NA = 4;        // float4 per lane of the deep half:  H <= 64 * 4 * NA = 1024
constexpr int NC = 10;       // float2 per lane of the cross half: X <= 64 * 2 * NC = 1280
constexpr int ROWS = 32;     // rows per workgroup (8 per wave)

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d, 64);
    return x;
}

// partials: [nblk][P] with P = 2 H + X + 2 laid out [dW3 deep (H) | db2 (H) | dW3 cross (X) | db3, loss sum] (the 16-byte
// lanes first: every region starts aligned for its lanes' stores)
__global__ __launch_bounds__(256) void k_dcn_head(const float* __restrict__ d2, int64_t ldd, const float* __restrict__ c, int64_t ldc,
                                                  const float* __restrict__ w3, const float* __restrict__ b3,
                                                  const float* __restrict__ label, int64_t B, int H, int X, float dscale,
                                                  float* __restrict__ logit_out, float* __restrict__ dd2, int64_t lddd,
                                                  float* __restrict__ dc, int64_t lddc, float* __restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) float red[];      // [4][Pw]
    const int t = threadIdx.x, l = t & 63, w = t >> 6;
    const int P = 2 * H + X + 2, Pw = (P + 3) & ~3;
    float4 wa[NA];
    float2 wc[NC];
#pragma unroll
    for (int q = 0; q < NA; ++q) {
        const int j = (q * 64 + l) * 4;
        wa[q] = j < H ? *(const float4*)(w3 + j) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int q = 0; q < NC; ++q) {
        const int j = (q * 64 + l) * 2;
        wc[q] = j < X ? *(const float2*)(w3 + H + j) : make_float2(0.f, 0.f);
    }
    const float bias = b3[0];
    float4 gwa[NA], gb2[NA];
    float2 gwc[NC];
#pragma unroll
    for (int q = 0; q < NA; ++q) { gwa[q] = make_float4(0.f, 0.f, 0.f, 0.f); gb2[q] = gwa[q]; }
#pragma unroll
    for (int q = 0; q < NC; ++q) gwc[q] = make_float2(0.f, 0.f);
    float gb3 = 0.0f, lsum = 0.0f;

    const int64_t r0 = (int64_t)blockIdx.x * ROWS + w * (ROWS / 4);
    for (int rr = 0; rr < ROWS / 4; ++rr) {
        const int64_t r = r0 + rr;
        if (r >= B) break;                   // (wave-uniform)
        float4 xa[NA];
        float2 xc[NC];
        float dot = 0.0f;
#pragma unroll
        for (int q = 0; q < NA; ++q) {
            const int j = (q * 64 + l) * 4;
            xa[q] = j < H ? *(const float4*)(d2 + r * ldd + j) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int q = 0; q < NC; ++q) {
            const int j = (q * 64 + l) * 2;
            xc[q] = j < X ? *(const float2*)(c + r * ldc + j) : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int q = 0; q < NA; ++q) dot += (xa[q].x * wa[q].x + xa[q].y * wa[q].y) + (xa[q].z * wa[q].z + xa[q].w * wa[q].w);
#pragma unroll
        for (int q = 0; q < NC; ++q) dot += xc[q].x * wc[q].x + xc[q].y * wc[q].y;
        const float z = wave_sum(dot) + bias;
        const float y = label[r];
        // SigmoidCrossEntropyWithLogits: max(z, 0) - z y + log(1 + exp(-|z|)); its bprop: sigmoid(z) - y
        const float e = expf(-fabsf(z));
        const float loss = fmaxf(z, 0.0f) - z * y + log1pf(e);
        const float sig = z >= 0.0f ? 1.0f / (1.0f + e) : e / (1.0f + e);
        const float dl = (sig - y) * dscale;
        if (l == 0) {
            logit_out[r] = z;
            gb3 += dl;
            lsum += loss;
        }
#pragma unroll
        for (int q = 0; q < NA; ++q) {
            const int j = (q * 64 + l) * 4;
            if (j < H) {
                float4 g = make_float4(xa[q].x > 0.f ? dl * wa[q].x : 0.f, xa[q].y > 0.f ? dl * wa[q].y : 0.f,
                                       xa[q].z > 0.f ? dl * wa[q].z : 0.f, xa[q].w > 0.f ? dl * wa[q].w : 0.f);
                *(float4*)(dd2 + r * lddd + j) = g;
                gb2[q].x += g.x; gb2[q].y += g.y; gb2[q].z += g.z; gb2[q].w += g.w;
                gwa[q].x += xa[q].x * dl; gwa[q].y += xa[q].y * dl; gwa[q].z += xa[q].z * dl; gwa[q].w += xa[q].w * dl;
            }
        }
#pragma unroll
        for (int q = 0; q < NC; ++q) {
            const int j = (q * 64 + l) * 2;
            if (j < X) {
                *(float2*)(dc + r * lddc + j) = make_float2(dl * wc[q].x, dl * wc[q].y);
                gwc[q].x += xc[q].x * dl; gwc[q].y += xc[q].y * dl;
            }
        }
    }
    // the four waves' column sums through LDS, added in wave order
    float* mine = red + w * Pw;
#pragma unroll
    for (int q = 0; q < NA; ++q) {
        const int j = (q * 64 + l) * 4;
        if (j < H) { *(float4*)(mine + j) = gwa[q]; *(float4*)(mine + H + j) = gb2[q]; }
    }
#pragma unroll
    for (int q = 0; q < NC; ++q) {
        const int j = (q * 64 + l) * 2;
        if (j < X) *(float2*)(mine + 2 * H + j) = gwc[q];
    }
    if (l == 0) { mine[2 * H + X] = gb3; mine[2 * H + X + 1] = lsum; }
    __syncthreads();
    for (int j = t; j < P; j += 256)
        partials[(int64_t)blockIdx.x * P + j] = ((red[j] + red[Pw + j]) + red[2 * Pw + j]) + red[3 * Pw + j];
}

// out[j] = sum over the workgroups' partials in workgroup order
__global__ __launch_bounds__(256) void k_dcn_head_finish(const float* __restrict__ partials, int nblk, int P, int H, int X, float inv_B,
                                                         float* __restrict__ dw3, float* __restrict__ db2, float* __restrict__ db3,
                                                         float* __restrict__ loss) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= P) return;
    float s = 0.0f;
    int b = 0;
    for (; b + 8 <= nblk; b += 8) {
        float u[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) u[q] = partials[(int64_t)(b + q) * P + j];
#pragma unroll
        for (int q = 0; q < 8; ++q) s += u[q];
    }
    for (; b < nblk; ++b) s += partials[(int64_t)b * P + j];
    if (j < H) dw3[j] = s;
    else if (j < 2 * H) db2[j - H] = s;
    else if (j < 2 * H + X) dw3[H + (j - 2 * H)] = s;
    else if (j == 2 * H + X) db3[0] = s;
    else loss[0] = s * inv_B;
}

}  // namespace dcn

MREC_API int mrec_dcn_head_workspace_bytes(int64_t B, int32_t H, int32_t X, size_t* out) {
    if (!out || B < 0 || H <= 0 || X <= 0) return MREC_EINVAL;
    *out = (size_t)mrec_cdiv(B ? B : 1, dcn::ROWS) * (2 * (size_t)H + X + 2) * sizeof(float);
    return MREC_OK;
}

MREC_API int mrec_dcn_head_fwd_bwd(const float* d2, int64_t ldd, const float* c, int64_t ldc, const float* w3, const float* b3,
                                   const float* label, int64_t B, int32_t H, int32_t X, float dscale, float* logit_out, float* dd2,
                                   int64_t lddd, float* dc, int64_t lddc, float* dw3_out, float* db2_out, float* db3_out,
                                   float* loss_out, void* ws, size_t ws_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (B <= 0 || H <= 0 || X <= 0 || ldd < H || ldc < X || lddd < H || lddc < X) return MREC_EINVAL;
    if (!d2 || !c || !w3 || !b3 || !label || !logit_out || !dd2 || !dc || !dw3_out || !db2_out || !db3_out || !loss_out || !ws) return MREC_EINVAL;
    if (H % 4 || X % 2 || H > 64 * 4 * dcn::NA || X > 64 * 2 * dcn::NC || ldd % 4 || lddd % 4 || ldc % 2 || lddc % 2 ||
        ((((uintptr_t)d2) | ((uintptr_t)dd2) | ((uintptr_t)w3)) & 15) || ((((uintptr_t)c) | ((uintptr_t)dc)) & 7) || (H % 2))
        return MREC_EUNSUPPORTED;
    const int P = 2 * H + X + 2;
    const int nblk = (int)mrec_cdiv(B, dcn::ROWS);
    if (ws_bytes < (size_t)nblk * P * sizeof(float)) return MREC_EWORKSPACE;
    const size_t lds = (size_t)4 * ((P + 3) & ~3) * sizeof(float);
    if (lds > 160 * 1024) return MREC_EUNSUPPORTED;
    if (lds > 64 * 1024)
        MREC_HIP_CHECK(hipFuncSetAttribute((const void*)dcn::k_dcn_head, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dcn::k_dcn_head<<<nblk, 256, lds, st>>>(d2, ldd, c, ldc, w3, b3, label, B, H, X, dscale, logit_out, dd2, lddd, dc, lddc, (float*)ws);
    dcn::k_dcn_head_finish<<<(unsigned)mrec_cdiv(P, 256), 256, 0, st>>>((const float*)ws, nblk, P, H, X, 1.0f / (float)B, dw3_out, db2_out,
                                                                      db3_out, loss_out);
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}
