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
// The kernels' own wall-clock stamps (how bench.py times kernels INSIDE the captured step: events in a graph cannot be timed on this
// stack): workgroup 0 stores the begin, the last wave of the last-dispatched workgroups the end; only with a step state.
// -DMREC_STAMPS=0 compiles them out: same results bit for bit, and ~6 us per step less (0.9 %: the end stamp sits on the critical tail
// of k_apply_main, 3.4 us, and of the lookup, 1.6 us -- profiles/r05_stamps_ab.txt; one atomicMax per workgroup, one plain store per
// workgroup and stamps by the last round of residency only were all measured: the same).
#ifndef MREC_STAMPS
#define MREC_STAMPS 7      // bits: 1 k_apply_main, 2 the lookup kernels, 4 the finishing pass
#endif

struct StepState {
    float b1p, b2p, lr_t, pad0;
    long long step;
    unsigned long long stamps_off;              // nonzero: no kernel stamps in the steps that run with this state (a training run: 0 costs
                                                // the step ~6 us; bench.py turns them on for the blocks its roofline figures come from)
    unsigned long long stamps[kStampRing][2];   // [step % ring] = {first workgroup start, last wave end} of k_apply_main, wall clock ticks
    // [step % ring] = {begin, end of the step's fused lookup kernel (k_gather_rows with the wide lane), end of k_apply_long, 0}
    unsigned long long aux[kStampRing][4];
    // [step % ring][workgroup % 64] = end of that workgroup's last wave in k_apply_main (the kernel's end = the maximum): PLAIN stores,
    // the later finisher of a slot overwriting the earlier -- one atomicMax per workgroup on stamps[.][1] (4096 of them on one word)
    // cost the step 4 us (profiles/r05_stamps_ab.txt)
    unsigned long long ends[kStampRing][64];
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

