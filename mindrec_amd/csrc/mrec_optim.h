// mrec_optim.h -- per-element optimizer updates shared by the dense and the sparse-apply kernels.
// Formulas restate SURVEY.md Appendix A.4 (Adam / LazyAdam) and A.5 (FTRL); operation order is
// kept identical to oracle/mrec_oracle.c so fp32 results agree bit for bit when lr_power == -0.5.
#pragma once
#include "mrec_common.h"

struct AdamH { float lr_t, b1, b2, omb1, omb2, eps, gscale; int nesterov; };

__device__ __forceinline__ void adam_elem(float& p, float& m, float& v, float g, const AdamH& h) {
    const float mn = h.b1 * m + h.omb1 * g;
    const float vn = h.b2 * v + h.omb2 * (g * g);
    const float num = h.nesterov ? (h.b1 * mn + h.omb1 * g) : mn;
    p = p - (h.lr_t * num) / (sqrtf(vn) + h.eps);
    m = mn;
    v = vn;
}

// Step scalars in device memory (mrec_step_state_t of include/mrec.h): the Adam bias-correction powers advance by a kernel
// (mrec_step_advance), so a whole training step -- optimizer included -- replays as one HIP graph with constant arguments.
constexpr int kStampRing = 256;
struct StepState {
    float b1p, b2p, lr_t, pad0;
    long long step;
    unsigned long long pad1;
    unsigned long long stamps[kStampRing][2];   // [step % ring] = {first workgroup start, last wave end} of k_apply_main, wall clock ticks
    // [step % ring] = {begin, end of the step's fused lookup kernel (k_gather_rows with the wide lane), end of k_apply_long, 0}
    unsigned long long aux[kStampRing][4];
};

struct FtrlH { float lr, l1, l2, lr_power, gscale; };

__device__ __forceinline__ void ftrl_elem(float& w, float& a, float& lin, float g, const FtrlH& h) {
    const float an = a + g * g;
    float y, y0;
    if (h.lr_power == -0.5f) { y = sqrtf(an); y0 = sqrtf(a); }
    else { y = powf(an, -h.lr_power); y0 = powf(a, -h.lr_power); }
    const float sigma = (y - y0) / h.lr;
    const float ln = lin + (g - sigma * w);
    const float cl = ln < -h.l1 ? -h.l1 : (ln > h.l1 ? h.l1 : ln);
    const float x = cl - ln;
    const float q = y / h.lr + 2.0f * h.l2;
    w = x / q;
    lin = ln;
    a = an;
}

