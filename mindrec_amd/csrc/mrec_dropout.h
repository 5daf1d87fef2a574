// mrec_dropout.h -- counter-based Dropout mask (DenseLayer.construct, models/wide_deep/src/wide_and_deep.py:98,117-118:
// `x = self.dropout(x)` on the INPUT of every DenseLayer while training; Dropout(p = 1 - keep_prob), kept elements scaled by
// 1 / keep_prob).  MindSpore draws the mask from its own generator; here the mask is a pure function of (seed, step, layer,
// sample row, column), so that forward and backward, eager and graph replay, one GPU and N data-parallel ranks -- and the CPU
// oracle, which states the same function independently (oracle/mrec_oracle.c) -- all see the same mask without storing it.
//
// Spec, for element (r, c) of the [M, W] input of DenseLayer `layer` (0-based) at training step `step` (0-based), W % 4 == 0:
//   key   = mix64(seed ^ mix64(step * 16 + layer))
//   quad  = mix64(key + ((r * W + c) >> 2))
//   bits  = (quad >> (16 * (c & 3))) & 0xFFFF
//   keep  = bits < thresh,   thresh = round(keep_prob * 65536);   kept: x * (1 / keep_prob) in fp32, else 0
// r is the GLOBAL sample row (a data-parallel rank passes row0 = rank * local batch).
#pragma once
#include "mrec_common.h"

struct DropArgs {
    const long long* step_dev;   // &mrec_step_state_t.step in device memory (a captured step replays with a moving step), or null
    uint64_t seed;
    long long step;              // used when step_dev is null
    int64_t row0;
    int layer;
    uint32_t thresh;             // 0: no dropout
    float scale;
};

__device__ __forceinline__ uint64_t drop_key(const DropArgs& d) {
    const long long t = d.step_dev ? *d.step_dev : d.step;
    return mrec_mix64(d.seed ^ mrec_mix64((uint64_t)t * 16ull + (uint64_t)d.layer));
}
// the 4 x 16 mask bits of the aligned quad of columns that holds column c of row r
__device__ __forceinline__ uint64_t drop_quad(uint64_t key, int64_t r, int64_t W, int64_t c) {
    return mrec_mix64(key + (uint64_t)((r * W + c) >> 2));
}
__device__ __forceinline__ bool drop_keep(uint64_t quad, int j, uint32_t thresh) {
    return ((uint32_t)(quad >> (16 * j)) & 0xFFFFu) < thresh;
}

// host: the C-ABI descriptor -> kernel arguments (thresh 0 = no dropout; keep_prob 1 is the identity).  false: bad descriptor.
static inline bool drop_from(const mrec_dropout_t* d, int64_t W, DropArgs* out) {
    DropArgs a{};
    if (d) {
        if (!(d->keep_prob > 0.f) || d->keep_prob > 1.f || d->layer < 0 || d->layer > 15 || d->step < 0 || d->row0 < 0 || W % 4) return false;
        const long t = lrintf(d->keep_prob * 65536.f);
        if (t < 65536) {
            a.thresh = (uint32_t)(t < 1 ? 1 : t);
            a.scale = 1.0f / d->keep_prob;
            a.seed = d->seed;
            a.step = d->step;
            a.row0 = d->row0;
            a.layer = d->layer;
            a.step_dev = d->step_state ? (const long long*)((const char*)d->step_state + offsetof(mrec_step_state_t, step)) : nullptr;
        }
    }
    *out = a;
    return true;
}
