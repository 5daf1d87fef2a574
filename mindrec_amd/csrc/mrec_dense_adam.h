// mrec_dense_adam.h -- the dense Adam over a flat buffer whose gradient is partly fp32 slabs (the dense net's optimizer step),
// as a device body shared by its own kernel (mrec_gather.hip) and by the launch that also finishes the sparse apply's
// window-crossing runs (k_finish_dense_adam, mrec_apply.hip).
#pragma once
#include "mrec_common.h"
#include "mrec_optim.h"

namespace {

__device__ __forceinline__ uint16_t f2bf(float x) { __bf16 b = (__bf16)x; return __builtin_bit_cast(uint16_t, b); }

// One element of a dense buffer may belong to FTRL instead of Adam (Ftrl1: its index, or -1): Wide&Deep's `wide_b`, which
// TrainStepWrap hands to the FTRL optimizer with the wide table (wide_and_deep.py:407-411) while it lives in the dense net's flat
// buffer here -- its `m` word is FTRL's accum, its `v` word FTRL's linear.  A uniform compare per vector; no second launch.
struct Ftrl1 { int64_t idx; FtrlH h; };

__device__ __forceinline__ void adam_or_ftrl(float& p, float& m, float& v, float g, const AdamH& h, bool ftrl, const FtrlH& fh) {
    if (ftrl) ftrl_elem(p, m, v, g, fh);
    else adam_elem(p, m, v, g, h);
}


// The same, for the hand-written MFMA weight-gradient kernel (mrec_dense.hip): its split slabs are fp32 partial
// sums (never rounded), added here in slab order.  SHK: 0 = no shadow, 1 = bf16 shadow, 2 = fp16 shadow.
struct SlabSegs {
    const float4* part[16];
    int64_t start4[16], len4[16];
    int S[16];
    int n;
};

__device__ __forceinline__ unsigned pack_shadow2(float lo, float hi, int shk) {
    if (shk == 2) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        h2 v = {(_Float16)lo, (_Float16)hi};
        return __builtin_bit_cast(unsigned, v);
    }
    return (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16);
}

struct DenseAdamSlabArgs { float4* p; float4* m; float4* v; const float4* g; int64_t n4; AdamH h; uint2* shadow; const StepState* ss; Ftrl1 f1; };

// (tid0, nthreads): this thread's first vector and the stride -- the kernel below passes its own grid; a launch that also carries
// other work (k_finish_dense_adam, mrec_apply.hip) passes what is left of its grid)
template <int SHK>
__device__ __forceinline__ void dense_adam4_slabs_body(float4* __restrict__ p, float4* __restrict__ m,
                                                       float4* __restrict__ v, const float4* __restrict__ g,
                                                       int64_t n4, AdamH h, uint2* __restrict__ shadow, const SlabSegs& sg,
                                                       const StepState* ss, const Ftrl1& f1, int64_t tid0, int64_t nthreads) {
    if (ss) h.lr_t = ss->lr_t;             // this step's bias-corrected step size from device memory (mrec_step_advance)
    for (int64_t i = tid0; i < n4; i += nthreads) {
        float4 pp = p[i], mm = m[i], vv = v[i];
        float4 gg;
        int k = -1;
        for (int q = 0; q < sg.n; ++q)
            if (i >= sg.start4[q] && i < sg.start4[q] + sg.len4[q]) k = q;
        if (k >= 0) {
            // slabs added in slab order, eight loads in flight at a time (a serial loop over 64 bias-gradient slabs is 64
            // dependent L2 round trips for the threads that own bias elements: it doubled the kernel's time)
            const float4* src = sg.part[k] + (i - sg.start4[k]);
            const int S = sg.S[k];
            const int64_t L = sg.len4[k];
            gg = src[0];
            for (int s0 = 1; s0 < S; s0 += 8) {
                float4 u[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) u[q] = (s0 + q < S) ? src[(int64_t)(s0 + q) * L] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (s0 + q < S) { gg.x += u[q].x; gg.y += u[q].y; gg.z += u[q].z; gg.w += u[q].w; }
            }
        } else {
            gg = g[i];
        }
        const int fq = (f1.idx >> 2) == i ? (int)(f1.idx & 3) : -1;
        adam_or_ftrl(pp.x, mm.x, vv.x, gg.x * h.gscale, h, fq == 0, f1.h);
        adam_or_ftrl(pp.y, mm.y, vv.y, gg.y * h.gscale, h, fq == 1, f1.h);
        adam_or_ftrl(pp.z, mm.z, vv.z, gg.z * h.gscale, h, fq == 2, f1.h);
        adam_or_ftrl(pp.w, mm.w, vv.w, gg.w * h.gscale, h, fq == 3, f1.h);
        p[i] = pp; m[i] = mm; v[i] = vv;
        if (SHK) shadow[i] = make_uint2(pack_shadow2(pp.x, pp.y, SHK), pack_shadow2(pp.z, pp.w, SHK));
    }
}

template <int SHK>
__global__ __launch_bounds__(256) void k_dense_adam4_slabs(float4* __restrict__ p, float4* __restrict__ m,
                                                           float4* __restrict__ v, const float4* __restrict__ g,
                                                           int64_t n4, AdamH h, uint2* __restrict__ shadow, SlabSegs sg,
                                                           const StepState* ss, Ftrl1 f1) {
    dense_adam4_slabs_body<SHK>(p, m, v, g, n4, h, shadow, sg, ss, f1, (int64_t)blockIdx.x * 256 + threadIdx.x, (int64_t)gridDim.x * 256);
}

}  // namespace
