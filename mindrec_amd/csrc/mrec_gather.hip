// mrec_gather.hip -- row gather (Gather / SparseGatherV2 / EmbeddingLookup), the fused wide-branch
// reduction, on-device table initialisation and the dense (whole-tensor) optimizers, for gfx950.
//
// Reference call sites: mindspore_rec/ops/embedding.py:150,194; models/wide_deep/src/
// wide_and_deep.py:277-290,300-309,435-445; models/deep_and_cross/src/deep_and_cross.py:199,342-344.
//
// All of these are HBM-bound byte movers.  A row of D floats is covered by lpr = D/4 lanes holding
// one float4 each (D = 80 -> 20 lanes, three rows per wave64 instruction, 60/64 lanes busy); each
// lane-group keeps GB independent row loads in flight before the first store.
#include <stdlib.h>

#include "mrec_common.h"
#include <cstring>
#include "mrec_rng.h"
#include "mrec_optim.h"
#include "mrec_dropout.h"
#include "mrec_dense_adam.h"

int g_mrec_last_hip_error = 0;

namespace {

template <int VEC> struct Vf;
template <> struct Vf<4> { float4 v; };
template <> struct Vf<1> { float v; };

__device__ __forceinline__ Vf<4> vload(const float* p, Vf<4>*) { Vf<4> r; r.v = *(const float4*)p; return r; }
__device__ __forceinline__ Vf<1> vload(const float* p, Vf<1>*) { Vf<1> r; r.v = *p; return r; }
__device__ __forceinline__ void vstore(float* p, const Vf<4>& x) { *(float4*)p = x.v; }
__device__ __forceinline__ void vstore(float* p, const Vf<1>& x) { *p = x.v; }
__device__ __forceinline__ Vf<4> vscale(Vf<4> x, float s) { x.v.x *= s; x.v.y *= s; x.v.z *= s; x.v.w *= s; return x; }
__device__ __forceinline__ Vf<1> vscale(Vf<1> x, float s) { x.v *= s; return x; }
// "this value is needed here" (see k_gather_rows)
__device__ __forceinline__ void vtouch(Vf<4>& r) { asm volatile("" : "+v"(r.v.x), "+v"(r.v.y), "+v"(r.v.z), "+v"(r.v.w)); }
__device__ __forceinline__ void vtouch(Vf<1>& r) { asm volatile("" : "+v"(r.v)); }
__device__ __forceinline__ Vf<4> vzero(Vf<4>*) { Vf<4> r; r.v = make_float4(0.f, 0.f, 0.f, 0.f); return r; }
__device__ __forceinline__ Vf<1> vzero(Vf<1>*) { Vf<1> r; r.v = 0.f; return r; }

// bf16 output rows (round-to-nearest-even, the cast the reference applies to the masked embeddings
// before its mixed-precision MLP, wide_and_deep.py:113-133): halves the gather's write stream.
struct bf16o_t { uint16_t v; };
__device__ __forceinline__ void vstore(bf16o_t* p, const Vf<4>& x) {
    uint2 u;
    u.x = (unsigned)f2bf(x.v.x) | ((unsigned)f2bf(x.v.y) << 16);
    u.y = (unsigned)f2bf(x.v.z) | ((unsigned)f2bf(x.v.w) << 16);
    *(uint2*)p = u;
}
__device__ __forceinline__ void vstore(bf16o_t* p, const Vf<1>& x) { p->v = f2bf(x.v); }

// ... and IEEE half output rows (the reference's own mixed-precision dtype)
struct f16o_t { uint16_t v; };
__device__ __forceinline__ unsigned f2h2(float lo, float hi) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 v = {(_Float16)lo, (_Float16)hi};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ void vstore(f16o_t* p, const Vf<4>& x) { *(uint2*)p = make_uint2(f2h2(x.v.x, x.v.y), f2h2(x.v.z, x.v.w)); }
__device__ __forceinline__ void vstore(f16o_t* p, const Vf<1>& x) { p->v = __builtin_bit_cast(uint16_t, (_Float16)x.v); }

#ifndef MREC_GB
#define MREC_GB 4
#endif
constexpr int GB = MREC_GB;  // rows in flight per lane-group (2 / 4 / 8 in the step: lookup 49.3 / 48.2 / 49.7 us)

// Lane-group geometry shared by the row kernels: lpr lanes per row, G = 64/lpr groups per wave.
struct RowGeom { int lpr; int G; };

// wprod (nullable; float4 path with 16-bit output only): the wide branch's products ride along.  The lane-group gets one
// lane more (sub == lpr - 1, column D): it loads the float4 [w accum linear pad] that sits right behind the D deep columns of
// a fused row with the instruction its neighbours load deep columns with, and stores (w * scale, 0) as 8 bytes to
// wprod[2 i] with the instruction they store four 16-bit values with -- no extra memory instructions (a lane-masked scalar
// load + store for it cost +13 us).  The output head adds the products up per sample in field order
// (mrec_head_fwd_bwd_wide), which replaces mrec_wide_sum.
__device__ __forceinline__ uint2 pack16(const bf16o_t*, const float4& v) {
    return make_uint2((unsigned)f2bf(v.x) | ((unsigned)f2bf(v.y) << 16), (unsigned)f2bf(v.z) | ((unsigned)f2bf(v.w) << 16));
}
__device__ __forceinline__ uint2 pack16(const f16o_t*, const float4& v) { return make_uint2(f2h2(v.x, v.y), f2h2(v.z, v.w)); }
__device__ __forceinline__ uint2 pack16(const float*, const float4&) { return make_uint2(0u, 0u); }

// Dropout on looked-up 16-bit rows (the first DenseLayer's input, wide_and_deep.py:117-118) inside the lookup: position i of
// the id list is sample i / F, columns (i % F) * D + col .. + 3 of that sample's [F * D] input; x * (1 / keep) rounded again.
struct GatherDrop { DropArgs d; int F; };
__device__ __forceinline__ float widen16(const bf16o_t*, uint32_t b) { return __uint_as_float(b << 16); }
__device__ __forceinline__ float widen16(const f16o_t*, uint32_t b) { return (float)__builtin_bit_cast(_Float16, (uint16_t)b); }
__device__ __forceinline__ float widen16(const float*, uint32_t) { return 0.0f; }
template <class OT>
__device__ __forceinline__ uint2 drop16(uint2 u, const GatherDrop& gd, uint64_t key, int64_t i, int D, int col) {
    const int64_t r = i / gd.F;
    const int64_t c = (i - r * gd.F) * D + col;
    const uint64_t qd = drop_quad(key, gd.d.row0 + r, (int64_t)gd.F * D, c);
    float v[4] = {widen16((const OT*)nullptr, u.x & 0xFFFFu), widen16((const OT*)nullptr, u.x >> 16),
                  widen16((const OT*)nullptr, u.y & 0xFFFFu), widen16((const OT*)nullptr, u.y >> 16)};
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = drop_keep(qd, j, gd.d.thresh) ? v[j] * gd.d.scale : 0.0f;
    return pack16((const OT*)nullptr, make_float4(v[0], v[1], v[2], v[3]));
}

template <int VEC, class K, class OT = float>
__global__ __launch_bounds__(256) void k_gather_rows(const float* __restrict__ table, int64_t V, int64_t ld,
                                                     const K* __restrict__ ids, int64_t n,
                                                     const float* __restrict__ row_scale,
                                                     OT* __restrict__ out, int D, RowGeom gm,
                                                     float* __restrict__ wprod = nullptr, int64_t ldo = 0, int64_t ldw = 2,
                                                     GatherDrop gd = GatherDrop{}, int ids_stride = 1, int rs_stride = 1,
                                                     bool skip_invalid = false, StepState* ss = nullptr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane / gm.lpr, sub = lane - grp * gm.lpr;
    // Stamps (measurement only; ss == nullptr: none): workgroup 0 stores the begin, the last wave of each of the LAST-dispatched
    // 512 workgroups raises the end -- this lookup belongs to step ss->step + 1 (the step's advance runs behind it).
    __shared__ int waves_done;
    unsigned long long* stamp = ((MREC_STAMPS & 2) && ss && !ss->stamps_off) ? ss->aux[(unsigned)(ss->step + 1) % kStampRing] : nullptr;
    if (stamp) {
        if (threadIdx.x == 0) {
            waves_done = 0;
            if (blockIdx.x == 0) stamp[0] = (unsigned long long)wall_clock64();
        }
        __syncthreads();
    }
    if (grp < gm.G) {
    if (ldo == 0) ldo = D;                 // row stride of `out` in elements; ldw: stride of the wide products in floats
    const int col = sub * VEC;
    const bool wl = VEC == 4 && wprod != nullptr && col >= D;      // the wide lane (column D)
    const int64_t wave_row0 = ((int64_t)blockIdx.x * 4 + wave) * (gm.G * GB);
    Vf<VEC> x[GB];
    float sc[GB];
    int64_t row[GB];
    // Every load is requested unconditionally (a position past the end reads the last id, an id out of range reads row 0; both
    // values are dropped) and the rows are "needed" in straight-line code before the first store: a guarded load that is
    // converted inside its block is waited for there -- the GB ids were GB dependent round trips -- and a wait at a join
    // behind a store waits for that store as well (vmcnt counts in order): GB - 1 store round trips more per wave.
    const int64_t nlast = n - 1;
    uint64_t dkey = 0;                                  // Dropout's key of this step (a device word): read once, with the ids
    if (sizeof(OT) == 2 && gd.d.thresh) dkey = drop_key(gd.d);
#pragma unroll
    for (int k = 0; k < GB; ++k) {
        const int64_t i = wave_row0 + (int64_t)k * gm.G + grp;
        const int64_t ic = i < n ? i : nlast;
        row[k] = (int64_t)ids[ic * ids_stride];          // (strides: the ids / weights of a shard's request message, read in place)
        sc[k] = row_scale ? row_scale[ic * rs_stride] : 1.0f;
    }
    bool okr[GB];
#pragma unroll
    for (int k = 0; k < GB; ++k) {
        const int64_t i = wave_row0 + (int64_t)k * gm.G + grp;
        if (i >= n) row[k] = -1;
        okr[k] = row[k] >= 0 && row[k] < V;
#if defined(MREC_GATHER_ABL) && MREC_GATHER_ABL == 2      // ablation: no row loads (every lane-group reads row 0)
        x[k] = vload(table + col, (Vf<VEC>*)nullptr);
#else
        x[k] = vload(table + (okr[k] ? row[k] : 0) * ld + col, (Vf<VEC>*)nullptr);
#endif
    }
#pragma unroll
    for (int k = 0; k < GB; ++k) {
        vtouch(x[k]);
        if (!okr[k]) x[k] = vzero((Vf<VEC>*)nullptr);
    }
#pragma unroll
    for (int k = 0; k < GB; ++k) {
        const int64_t i = wave_row0 + (int64_t)k * gm.G + grp;
#if defined(MREC_GATHER_ABL) && MREC_GATHER_ABL == 1      // ablation: no stores (a condition the compiler cannot fold)
        if (i < n && sc[k] == 1.2345e30f) {
#else
        if (i < n && !(skip_invalid && row[k] < 0)) {       // (skip_invalid: padding slots of a message are left alone)
#endif
            const Vf<VEC> y = row_scale ? vscale(x[k], sc[k]) : x[k];
            if constexpr (VEC == 4) {
              if (wprod != nullptr) {
                const float4 yv = y.v;
                if constexpr (sizeof(OT) == 2) {
                    // ONE store instruction for the whole lane-group: the wide lane's address points into wprod
                    uint2 u = pack16((const OT*)nullptr, yv);
                    if (gd.d.thresh && !wl) u = drop16<OT>(u, gd, dkey, i, D, col);
                    if (wl) u = make_uint2(__float_as_uint(yv.x), 0u);
                    uint2* dst = wl ? (uint2*)(wprod + ldw * i) : (uint2*)(out + i * ldo + col);
                    *dst = u;
                } else if (wl) {
                    *(uint2*)(wprod + ldw * i) = make_uint2(__float_as_uint(yv.x), 0u);      // fp32 rows (the fp32 wire format)
                } else {
                    vstore(out + i * ldo + col, y);
                }
              } else {
                vstore(out + i * ldo + col, y);
              }
            } else {
                vstore(out + i * ldo + col, y);
            }
        }
    }
    }
    if (stamp && (int)blockIdx.x + 512 >= (int)gridDim.x && lane == 0 && atomicAdd(&waves_done, 1) == 3)
        atomicMax(&stamp[1], (unsigned long long)wall_clock64());
}

// ---- the lookup with 16-byte stores (16-bit rows, D % 8 == 0) ---------------------------------------------------------
// What the kernel above is bound by turned out to be its STORES, not its row reads (profiles/r04_gather_ablation.txt: without
// stores 33 us, without row loads 24 us, both 45 us on uniform ids; on Zipf ids x 39 fields 28 / 32 / 52): a lane holds 4 columns
// = 8 bytes of 16-bit output, and 8-byte-per-lane stores issue at about half the rate of 16-byte ones.  Here neighbouring lanes
// of a row trade halves of TWO rows through the DPP network (quad_perm [1,0,3,2]: one VALU move per dword, no LDS) so that the
// even lane stores 16 bytes (8 columns) of row k and the odd lane 16 bytes of row k + 1 -- ONE store instruction per pair of
// rows, half as many as before, every one 16 bytes wide; the wide lane stores the products of both rows as one 16-byte word.
// A lane-group takes 4 CONSECUTIVE positions (their ids and weights are one 16-byte load each instead of four scalar ones).
// Odd lane-groups keep their wide lane FIRST so that every deep pair starts on an even hardware lane (lpr = D / 4 + 1 is odd).
__device__ __forceinline__ unsigned dpp_swap1(unsigned x) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true);
}
template <class K> struct Ids4;
template <> struct Ids4<int32_t> {
    int4 v;
    __device__ __forceinline__ void load(const int32_t* p) { v = *(const int4*)p; }
    __device__ __forceinline__ int64_t at(int k) const { return (int64_t)(k == 0 ? v.x : k == 1 ? v.y : k == 2 ? v.z : v.w); }
};
template <> struct Ids4<int64_t> {
    longlong2 a, b;
    __device__ __forceinline__ void load(const int64_t* p) { a = *(const longlong2*)p; b = *(const longlong2*)(p + 2); }
    __device__ __forceinline__ int64_t at(int k) const { return k == 0 ? a.x : k == 1 ? a.y : k == 2 ? b.x : b.y; }
};

template <class K, class OT>
__global__ __launch_bounds__(256) void k_gather_rows_w16(const float* __restrict__ table, int64_t V, int64_t ld,
                                                         const K* __restrict__ ids, int64_t n, const float* __restrict__ row_scale,
                                                         OT* __restrict__ out, int D, RowGeom gm, float* __restrict__ wprod,
                                                         int64_t ldo, GatherDrop gd, StepState* ss) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane / gm.lpr, sub = lane - grp * gm.lpr;
    __shared__ int waves_done;
    unsigned long long* stamp = ((MREC_STAMPS & 2) && ss && !ss->stamps_off) ? ss->aux[(unsigned)(ss->step + 1) % kStampRing] : nullptr;      // (as k_gather_rows)
    if (stamp) {
        if (threadIdx.x == 0) {
            waves_done = 0;
            if (blockIdx.x == 0) stamp[0] = (unsigned long long)wall_clock64();
        }
        __syncthreads();
    }
    if (grp < gm.G) {
        const bool has_w = wprod != nullptr;
        const bool wfirst = has_w && (grp & 1);
        const bool wl = has_w && (wfirst ? sub == 0 : sub == gm.lpr - 1);
        const int ds = wfirst ? sub - 1 : sub;                     // deep lane number (4 columns each); parity = the hardware lane's
        const int col = wl ? D : ds * 4;
        const int64_t i0 = ((int64_t)blockIdx.x * 4 + wave) * (gm.G * 4) + grp * 4;
        const bool live = i0 < n;                                  // (n % 4 == 0: a lane-group's 4 positions are all inside or all outside)
        const int64_t ic = live ? i0 : n - 4;
        Ids4<K> idv;
        idv.load(ids + ic);
        float4 scv = make_float4(1.f, 1.f, 1.f, 1.f);
        if (row_scale) scv = *(const float4*)(row_scale + ic);
        uint64_t dkey = 0;
        if (gd.d.thresh) dkey = drop_key(gd.d);
        Vf<4> x[4];
        bool okr[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t r = idv.at(k);
            okr[k] = r >= 0 && r < V;
            x[k] = vload(table + (okr[k] ? r : 0) * ld + col, (Vf<4>*)nullptr);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            vtouch(x[k]);
            if (!okr[k]) x[k] = vzero((Vf<4>*)nullptr);
        }
        const float sc[4] = {scv.x, scv.y, scv.z, scv.w};
        uint2 pk[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const Vf<4> y = row_scale ? vscale(x[k], sc[k]) : x[k];
            pk[k] = pack16((const OT*)nullptr, y.v);
            if (gd.d.thresh && !wl) pk[k] = drop16<OT>(pk[k], gd, dkey, i0 + k, D, col);
            if (wl) pk[k] = make_uint2(__float_as_uint(y.v.x), 0u);
        }
        const bool odd = (ds & 1) != 0;
#pragma unroll
        for (int kp = 0; kp < 4; kp += 2) {
            const uint2 pa = make_uint2(dpp_swap1(pk[kp].x), dpp_swap1(pk[kp].y));              // the partner's half of row kp
            const uint2 pb = make_uint2(dpp_swap1(pk[kp + 1].x), dpp_swap1(pk[kp + 1].y));      // ... of row kp + 1
            uint4 o;
            void* dst;
            if (wl) {
                o = make_uint4(pk[kp].x, 0u, pk[kp + 1].x, 0u);
                dst = wprod + 2 * (i0 + kp);
            } else if (odd) {
                o = make_uint4(pb.x, pb.y, pk[kp + 1].x, pk[kp + 1].y);
                dst = out + (i0 + kp + 1) * ldo + (col - 4);
            } else {
                o = make_uint4(pk[kp].x, pk[kp].y, pa.x, pa.y);
                dst = out + (i0 + kp) * ldo + col;
            }
            if (live) *(uint4*)dst = o;
        }
    }
    if (stamp && (int)blockIdx.x + 512 >= (int)gridDim.x && lane == 0 && atomicAdd(&waves_done, 1) == 3)
        atomicMax(&stamp[1], (unsigned long long)wall_clock64());
}

// Generic-D fallback (D not a multiple of 4, or misaligned): one wave per GB rows, lanes stride columns.
template <class K>
__global__ __launch_bounds__(256) void k_gather_rows_generic(const float* __restrict__ table, int64_t V,
                                                             int64_t ld, const K* __restrict__ ids, int64_t n,
                                                             const float* __restrict__ row_scale,
                                                             float* __restrict__ out, int D) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const int64_t r = (int64_t)ids[i];
    const bool ok = r >= 0 && r < V;
    const float s = row_scale ? row_scale[i] : 1.0f;
    for (int c = lane; c < D; c += 64) {
        float x = ok ? table[r * ld + c] : 0.0f;
        out[i * D + c] = row_scale ? x * s : x;
    }
}

// Wide branch: out[b] = sum_f w[ids[b,f]] * wts[b,f] + bias.  A block takes S whole samples
// (S*F <= WS_TILE entries): every thread gathers WS_PER entries (coalesced id / weight reads, all the
// 4-byte random gathers of the tile in flight at once), products go to LDS, then one thread per sample
// adds its F products in field order -- the same order, multiply and add as the CPU reference, so the
// result is bit-identical; only the memory-level parallelism changed (one thread per sample left the
// chip at one wave per CU: 36 us; this form ~3x faster).
constexpr int WS_PER = 8;
constexpr int WS_TILE = 256 * WS_PER;
template <class K>
__global__ __launch_bounds__(256) void k_wide_sum(const float* __restrict__ w, int64_t V, int64_t ldw,
                                                  const K* __restrict__ ids, const float* __restrict__ wts,
                                                  int64_t B, int F, int S, const float* __restrict__ bias,
                                                  float* __restrict__ out) {
    __shared__ float prod[WS_TILE];
    const int64_t b0 = (int64_t)blockIdx.x * S;
    const int64_t e0 = b0 * F;
    const int64_t nb = (B - b0 < S) ? B - b0 : S;       // samples in this block
    const int ne = (int)(nb * F);                         // entries in this block
    float x[WS_PER], t[WS_PER];
#pragma unroll
    for (int k = 0; k < WS_PER; ++k) {
        const int j = k * 256 + threadIdx.x;
        x[k] = 0.0f; t[k] = 0.0f;
        if (j < ne) {
            const int64_t r = (int64_t)ids[e0 + j];
            t[k] = wts[e0 + j];
            if (r >= 0 && r < V) x[k] = w[r * ldw];
        }
    }
#pragma unroll
    for (int k = 0; k < WS_PER; ++k) {
        const int j = k * 256 + threadIdx.x;
        if (j < ne) prod[j] = x[k] * t[k];
    }
    __syncthreads();
    if ((int)threadIdx.x < nb) {
        float acc = 0.0f;
        const float* p = prod + threadIdx.x * F;
        for (int f = 0; f < F; ++f) acc = acc + p[f];
        out[b0 + threadIdx.x] = acc + (bias ? *bias : 0.0f);
    }
}

// fallback for F > WS_TILE: one thread per sample
template <class K>
__global__ __launch_bounds__(256) void k_wide_sum_wideF(const float* __restrict__ w, int64_t V, int64_t ldw,
                                                        const K* __restrict__ ids, const float* __restrict__ wts,
                                                        int64_t B, int F, const float* __restrict__ bias,
                                                        float* __restrict__ out) {
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    float acc = 0.0f;
    for (int f = 0; f < F; ++f) {
        const int64_t r = (int64_t)ids[b * F + f];
        const float x = (r >= 0 && r < V) ? w[r * ldw] : 0.0f;
        acc = acc + x * wts[b * F + f];
    }
    out[b] = acc + (bias ? *bias : 0.0f);
}

__global__ __launch_bounds__(256) void k_fill_normal(float* __restrict__ out, int64_t nrows, int D, int64_t ld,
                                                     uint64_t seed, int64_t row0, int64_t row_stride, float sigma) {
    // one thread per (row, 4-column chunk); chunks per row = ceil(D/4)
    const int cpr = (D + 3) >> 2;
    const int64_t total = nrows * cpr;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int64_t r = t / cpr;
        const int c0 = (int)(t - r * cpr) * 4;
        float* o = out + r * ld + c0;
        float x[4];
        mrec_det_normal2(seed, row0 + r * row_stride, c0 >> 1, x[0], x[1]);            // (c0 is a multiple of 4: two column pairs)
        mrec_det_normal2(seed, row0 + r * row_stride, (c0 >> 1) + 1, x[2], x[3]);
#pragma unroll
        for (int k = 0; k < 4; ++k) x[k] = (c0 + k < D) ? sigma * x[k] : 0.0f;
        if (c0 + 4 <= D && ((ld & 3) == 0) && ((((uintptr_t)out) & 15) == 0)) {
            *(float4*)o = make_float4(x[0], x[1], x[2], x[3]);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) if (c0 + k < D) o[k] = x[k];
        }
    }
}

// Each wave takes 64 entries at a time: one coalesced read of the new-flags / rows / keys, a ballot of the
// entries that need a row, then the wave fills those rows together (coalesced stores).  When nothing is new
// -- every lookup of resident keys -- the kernel is a single pass over the flags.  Rows of <= 16 columns
// (the wide table's 1-4 floats) are filled by their own lane instead.
template <class K>
__global__ __launch_bounds__(256) void k_init_rows(float* __restrict__ table, int64_t ld, int D,
                                                   const int* __restrict__ rows, const K* __restrict__ keys,
                                                   const uint8_t* __restrict__ is_new, int64_t n_max,
                                                   const int64_t* __restrict__ n_dev, uint64_t seed, float sigma,
                                                   float fill) {
    int64_t n = n_max;
    if (n_dev) { const int64_t nd = *n_dev; n = nd < n ? nd : n; }
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    for (int64_t base = wave * 64; base < n; base += nwaves * 64) {
        const int64_t i = base + lane;
        int r = -1;
        int64_t key = 0;
        if (i < n && (!is_new || is_new[i])) {
            r = rows[i];
            if (r >= 0) key = (int64_t)keys[i];
        }
        if (D <= 16) {
            if (r >= 0)
                for (int c = 0; c < D; ++c)
                    table[(int64_t)r * ld + c] = sigma >= 0.0f ? sigma * mrec_det_normal(seed, key, c) : fill;
            continue;
        }
        uint64_t mask = __ballot(r >= 0);
        while (mask) {
            const int src = __ffsll((unsigned long long)mask) - 1;
            mask &= mask - 1;
            const int rr = __shfl(r, src, 64);
            const int64_t kk = ((int64_t)__shfl((int)(key >> 32), src, 64) << 32) | (uint32_t)__shfl((int)key, src, 64);
            for (int c = 2 * lane; c < D; c += 128) {          // a lane fills a pair of columns: one transform
                float z0 = fill, z1 = fill;
                if (sigma >= 0.0f) {
                    mrec_det_normal2(seed, kk, c >> 1, z0, z1);
                    z0 *= sigma; z1 *= sigma;
                }
                table[(int64_t)rr * ld + c] = z0;
                if (c + 1 < D) table[(int64_t)rr * ld + c + 1] = z1;
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_scatter_rows(float* __restrict__ table, int64_t ld, int D,
                                                      const int* __restrict__ rows, int64_t n,
                                                      const float* __restrict__ vals) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const int r = rows[i];
    if (r < 0) return;
    for (int c = lane; c < D; c += 64) table[(int64_t)r * ld + c] = vals[i * D + c];
}

// dst[dst_rows[i], :] = src[src_rows[i], :] for i < min(n, *n_dev): rows between a device cache and its pinned host home (either side
// may be host memory read / written over PCIe), lists compacted on the device, their length known there only.  A wave per
// row, 16-byte lanes when the geometry allows; i beyond the device count and negative rows cost nothing.
template <bool V4>
__global__ __launch_bounds__(256) void k_move_rows(const float* __restrict__ src, int64_t lds_, const int64_t* __restrict__ src_rows,
                                                   float* __restrict__ dst, int64_t ldd, const int64_t* __restrict__ dst_rows,
                                                   int64_t n, const int64_t* __restrict__ n_dev, int W) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    int64_t lim = n;
    if (n_dev) { const int64_t x = *n_dev; lim = x < 0 ? 0 : (x < n ? x : n); }
    if (i >= lim) return;
    const int64_t rs = src_rows[i], rd = dst_rows[i];
    if (rs < 0 || rd < 0) return;
    if (V4) {
        const float4* s4 = (const float4*)(src + rs * lds_);
        float4* d4 = (float4*)(dst + rd * ldd);
        for (int c = lane; c < W / 4; c += 64) d4[c] = s4[c];
    } else {
        for (int c = lane; c < W; c += 64) dst[rd * ldd + c] = src[rs * lds_ + c];
    }
}

// ---- dense optimizers ------------------------------------------------------------------------
// Dense Adam over n elements.  GT = float or bf16 gradients (the GEMM that produced a weight gradient
// in a bf16 MLP writes bf16; widening on load is exact).  SH: also write the updated parameter, rounded
// to bf16, into a shadow buffer -- the operand copy the next forward's bf16 GEMMs read, so no separate
// cast pass over the weights is needed.
__device__ __forceinline__ float g_widen(float x) { return x; }
__device__ __forceinline__ float g_widen(uint16_t x) { return __uint_as_float(((unsigned)x) << 16); }

template <class GT, bool SH>
__global__ __launch_bounds__(256) void k_dense_adam(float* __restrict__ p, float* __restrict__ m,
                                                    float* __restrict__ v, const GT* __restrict__ g, int64_t n,
                                                    AdamH h, uint16_t* __restrict__ shadow, Ftrl1 f1) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float pp = p[i], mm = m[i], vv = v[i];
        adam_or_ftrl(pp, mm, vv, g_widen(g[i]) * h.gscale, h, i == f1.idx, f1.h);
        p[i] = pp; m[i] = mm; v[i] = vv;
        if (SH) shadow[i] = f2bf(pp);
    }
}

template <bool SH>
__global__ __launch_bounds__(256) void k_dense_adam4(float4* __restrict__ p, float4* __restrict__ m,
                                                     float4* __restrict__ v, const float4* __restrict__ g,
                                                     int64_t n4, AdamH h, uint2* __restrict__ shadow, Ftrl1 f1) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 pp = p[i], mm = m[i], vv = v[i];
        const float4 gg = g[i];
        const int fq = (f1.idx >> 2) == i ? (int)(f1.idx & 3) : -1;
        adam_or_ftrl(pp.x, mm.x, vv.x, gg.x * h.gscale, h, fq == 0, f1.h);
        adam_or_ftrl(pp.y, mm.y, vv.y, gg.y * h.gscale, h, fq == 1, f1.h);
        adam_or_ftrl(pp.z, mm.z, vv.z, gg.z * h.gscale, h, fq == 2, f1.h);
        adam_or_ftrl(pp.w, mm.w, vv.w, gg.w * h.gscale, h, fq == 3, f1.h);
        p[i] = pp; m[i] = mm; v[i] = vv;
        if (SH) shadow[i] = make_uint2((unsigned)f2bf(pp.x) | ((unsigned)f2bf(pp.y) << 16),
                                       (unsigned)f2bf(pp.z) | ((unsigned)f2bf(pp.w) << 16));
    }
}

// Dense Adam whose gradient carries the L2 term of the loss: g_i = sums_i + l2s * p_i (l2s = l2_coef * sens: d/dp of
// l2_coef * sum(p^2) / 2 at the loss scale; NetWithLossClass.construct, wide_and_deep.py:356-360; deepfm.py:252-259), with the
// loss term's own by-product -- sum(p^2) BEFORE the update -- reduced in the same pass: per-thread fp64 partials, wave shuffle,
// LDS, one fp64 word per workgroup; k_sumsq_finish adds them in workgroup order.  Replaces three torch passes over the table
// (the product, the add onto the scattered sums, the sum of squares).
__global__ __launch_bounds__(256) void k_dense_adam4_l2(float4* __restrict__ p, float4* __restrict__ m, float4* __restrict__ v,
                                                        const float4* __restrict__ g, int64_t n4, AdamH h, float l2s,
                                                        double* __restrict__ partial, int tail) {
    __shared__ double red[4];
    double acc = 0.0;
    if (tail && blockIdx.x == 0 && threadIdx.x == 0) {          // the n % 4 elements behind the last whole float4 (DeepFM's [184 965, 1] table)
        float* ps = (float*)(p + n4); float* ms = (float*)(m + n4); float* vs = (float*)(v + n4);
        const float* gs = (const float*)(g + n4);
        for (int t = 0; t < tail; ++t) {
            float pp = ps[t], mm = ms[t], vv = vs[t], gg = gs[t];
            acc += (double)pp * pp;
            gg += l2s * pp;
            adam_elem(pp, mm, vv, gg * h.gscale, h);
            ps[t] = pp; ms[t] = mm; vs[t] = vv;
        }
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 pp = p[i], mm = m[i], vv = v[i];
        float4 gg = g[i];
        acc += (double)pp.x * pp.x + (double)pp.y * pp.y + (double)pp.z * pp.z + (double)pp.w * pp.w;
        gg.x += l2s * pp.x; gg.y += l2s * pp.y; gg.z += l2s * pp.z; gg.w += l2s * pp.w;       // product rounded, then added (no fma)
        adam_elem(pp.x, mm.x, vv.x, gg.x * h.gscale, h);
        adam_elem(pp.y, mm.y, vv.y, gg.y * h.gscale, h);
        adam_elem(pp.z, mm.z, vv.z, gg.z * h.gscale, h);
        adam_elem(pp.w, mm.w, vv.w, gg.w * h.gscale, h);
        p[i] = pp; m[i] = mm; v[i] = vv;
    }
    if (!partial) return;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) acc += __shfl_down(acc, d, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

// One workgroup adds the per-workgroup partials in a FIXED order (thread t takes partials t, t + 256, ...; then a tree over the
// threads): the same bits every run.  (One thread walking all of them was a chain of ~2000 dependent loads: 110 us per table.)
__global__ __launch_bounds__(256) void k_sumsq_finish(const double* __restrict__ partial, int nb, double* __restrict__ out, int accumulate) {
    __shared__ double red[256];
    double s = 0.0;
    for (int b = threadIdx.x; b < nb; b += 256) s += partial[b];
    red[threadIdx.x] = s;
    __syncthreads();
#pragma unroll
    for (int d = 128; d > 0; d >>= 1) {
        if (threadIdx.x < d) red[threadIdx.x] += red[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = (accumulate ? *out : 0.0) + red[0];
}

// nn.Adam over a WHOLE table whose gradient is nonzero on the step's touched rows only (the bprop of a dense Gather:
// UnsortedSegmentSum into a [V, D] tensor, deep_and_cross.py:199,342-344; deepfm.py:198,267; wide_and_deep.py:434-437 with sparse False):
// the row sums stay where the segment-sum left them ([U, D], group order) and every row looks its group up in `row_group` (-1: no
// gradient) -- instead of zeroing a [V, D] gradient, scattering the sums into it and reading it back (three passes over the table's
// size per step).  VEC = 4: D % 4 == 0, a thread owns 4 columns of a row; VEC = 1: any D.  Same arithmetic, element for element, as
// k_dense_adam4_l2 on the scattered gradient.
template <int VEC>
__global__ __launch_bounds__(256) void k_dense_adam_rows_l2(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v, int64_t V,
                                                            int D, const int* __restrict__ row_group, const float* __restrict__ sums,
                                                            AdamH h, float l2s, double* __restrict__ partial, const StepState* __restrict__ ss) {
    __shared__ double red[4];
    if (ss) h.lr_t = ss->lr_t;
    const int DV = D / VEC;
    const int64_t n = V * DV, stride = (int64_t)gridDim.x * 256;
    const int64_t sr = stride / DV;
    const int sc = (int)(stride - sr * DV);
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t r = i / DV;
    int c = (int)(i - r * DV);
    double acc = 0.0;
    for (; i < n; i += stride) {
        const int g = row_group[r];
        const int64_t e = (r * DV + c) * VEC;
        if (VEC == 4) {
            float4 pp = *(float4*)(p + e), mm = *(float4*)(m + e), vv = *(float4*)(v + e);
            float4 gg = g >= 0 ? *(const float4*)(sums + ((int64_t)g * DV + c) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            acc += (double)pp.x * pp.x + (double)pp.y * pp.y + (double)pp.z * pp.z + (double)pp.w * pp.w;
            gg.x += l2s * pp.x; gg.y += l2s * pp.y; gg.z += l2s * pp.z; gg.w += l2s * pp.w;
            adam_elem(pp.x, mm.x, vv.x, gg.x * h.gscale, h);
            adam_elem(pp.y, mm.y, vv.y, gg.y * h.gscale, h);
            adam_elem(pp.z, mm.z, vv.z, gg.z * h.gscale, h);
            adam_elem(pp.w, mm.w, vv.w, gg.w * h.gscale, h);
            *(float4*)(p + e) = pp; *(float4*)(m + e) = mm; *(float4*)(v + e) = vv;
        } else {
            float pp = p[e], mm = m[e], vv = v[e];
            float gg = g >= 0 ? sums[(int64_t)g * D + c] : 0.f;
            acc += (double)pp * pp;
            gg += l2s * pp;
            adam_elem(pp, mm, vv, gg * h.gscale, h);
            p[e] = pp; m[e] = mm; v[e] = vv;
        }
        r += sr; c += sc;
        if (c >= DV) { c -= DV; ++r; }
    }
    if (!partial) return;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) acc += __shfl_down(acc, d, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

// row_group[uniq[g]] = g for the step's groups (row_group was set to -1 everywhere before; a negative / out-of-range row: no entry)
__global__ __launch_bounds__(256) void k_rows_to_groups(const int* __restrict__ uniq, const int64_t* __restrict__ n_uniq_dev, int64_t U, int64_t V,
                                                        int* __restrict__ row_group) {
    const int64_t n = n_uniq_dev ? min(U, *n_uniq_dev) : U;
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < n; g += (int64_t)gridDim.x * 256) {
        const int r = uniq[g];
        if (r >= 0 && r < V) row_group[r] = (int)g;
    }
}

template <bool SH>
__global__ __launch_bounds__(256) void k_dense_adam4_g16(float4* __restrict__ p, float4* __restrict__ m,
                                                         float4* __restrict__ v, const uint2* __restrict__ g,
                                                         int64_t n4, AdamH h, uint2* __restrict__ shadow) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 pp = p[i], mm = m[i], vv = v[i];
        const uint2 u = g[i];
        adam_elem(pp.x, mm.x, vv.x, __uint_as_float(u.x << 16) * h.gscale, h);
        adam_elem(pp.y, mm.y, vv.y, __uint_as_float(u.x & 0xFFFF0000u) * h.gscale, h);
        adam_elem(pp.z, mm.z, vv.z, __uint_as_float(u.y << 16) * h.gscale, h);
        adam_elem(pp.w, mm.w, vv.w, __uint_as_float(u.y & 0xFFFF0000u) * h.gscale, h);
        p[i] = pp; m[i] = mm; v[i] = vv;
        if (SH) shadow[i] = make_uint2((unsigned)f2bf(pp.x) | ((unsigned)f2bf(pp.y) << 16),
                                       (unsigned)f2bf(pp.z) | ((unsigned)f2bf(pp.w) << 16));
    }
}

// g[start_q + e] = sum over s of slabs_q[s * len_q + e], slab order, for every segment q in ONE launch (the data-parallel
// all-reduce needs the summed gradient; 7 launches of a one-segment kernel whose bias segments had 256 lanes of work and 64
// dependent loads each took 210 us).  A thread owns one float4 of one segment; 8 slab loads in flight.
__global__ __launch_bounds__(256) void k_sum_slab_segments(float4* __restrict__ g, SlabSegs sg, int64_t total4) {
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total4; t += (int64_t)gridDim.x * 256) {
        int k = 0;
        int64_t off = t;
        while (k < sg.n - 1 && off >= sg.len4[k]) { off -= sg.len4[k]; ++k; }
        const float4* src = sg.part[k] + off;
        const int S = sg.S[k];
        const int64_t L = sg.len4[k];
        float4 gg = src[0];
        for (int s0 = 1; s0 < S; s0 += 8) {
            float4 u[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) u[q] = (s0 + q < S) ? src[(int64_t)(s0 + q) * L] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (s0 + q < S) { gg.x += u[q].x; gg.y += u[q].y; gg.z += u[q].z; gg.w += u[q].w; }
        }
        g[sg.start4[k] + off] = gg;
    }
}

__global__ __launch_bounds__(256) void k_dense_ftrl(float* __restrict__ w, float* __restrict__ a,
                                                    float* __restrict__ lin, const float* __restrict__ g,
                                                    int64_t n, FtrlH h) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float ww = w[i], aa = a[i], ll = lin[i];
        ftrl_elem(ww, aa, ll, g[i] * h.gscale, h);
        w[i] = ww; a[i] = aa; lin[i] = ll;
    }
}

inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

template <class K, class OT = bf16o_t>
int gather_bf16_impl(const float* table, int64_t V, int64_t ld, int32_t D, const K* ids, int64_t n,
                     const float* row_scale, uint16_t* out, void* stream, int wcol = 0, float* wprod = nullptr, int64_t ldo = 0,
                     int64_t ldw = 2, GatherDrop gd = GatherDrop{}, int ids_stride = 1, int rs_stride = 1, bool skip_invalid = false,
                     StepState* ss = nullptr) {
    constexpr int ob = (int)sizeof(OT);                    // bytes per output element (2: bf16 / f16, 4: fp32 rows + the wide lane)
    const int oa = ob == 2 ? 7 : 15;                       // alignment of an output row's 4-element quad
    if ((ldo != 0 && (ldo < D || ldo % 4)) || (wprod && (ldw < 2 || ldw % 2)) || ids_stride < 1 || rs_stride < 1) return MREC_EINVAL;
    if (wprod && (wcol != D || D % 4 || D > 252 || ld % 4 || ld < D + 4 || !al16(table) || (((uintptr_t)out) & oa) || (((uintptr_t)wprod) & 7)))
        return MREC_EUNSUPPORTED;       // the wide word must sit right behind the deep columns of 16-byte aligned rows
    if (ob == 4 && !wprod) return MREC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (n < 0 || D <= 0 || V < 0 || ld < D) return MREC_EINVAL;
    if (n == 0) return MREC_OK;
    if (V == 0) return MREC_EINVAL;      // rows are read unconditionally at clamped addresses: an empty table has no valid one
    if (!table || !ids || !out) return MREC_EINVAL;
    const bool vec = (D % 4 == 0) && (D <= 256) && (ld % 4 == 0) && al16(table) && ((((uintptr_t)out) & oa) == 0);
    static const bool no_w16 = getenv("MREC_GATHER_W8") != nullptr;      // (debug knob: read once, not per lookup)
    const int64_t ldo_e = ldo ? ldo : D;
    const bool w16 = vec && ob == 2 && D % 8 == 0 && n % 4 == 0 && n >= 4 && ids_stride == 1 && rs_stride == 1 && !skip_invalid && ldw == 2 &&
                     ldo_e % 8 == 0 && al16(out) && (!wprod || al16(wprod)) && al16(ids) && (!row_scale || al16(row_scale)) &&
                     !no_w16;
    if (w16) {
        if constexpr (ob == 2) {
            const int lpr = D / 4 + (wprod ? 1 : 0);
            RowGeom gm{lpr, 64 / lpr};
            k_gather_rows_w16<K, OT><<<(unsigned)mrec_cdiv(n, (int64_t)4 * gm.G * 4), 256, 0, st>>>(table, V, ld, ids, n, row_scale, (OT*)out, D, gm,
                                                                                                 wprod, ldo_e, gd, ss);
        }
    } else if (vec) {
        const int lpr = D / 4 + (wprod ? 1 : 0);
        RowGeom gm{lpr, 64 / lpr};
        k_gather_rows<4, K, OT><<<(unsigned)mrec_cdiv(n, (int64_t)4 * gm.G * GB), 256, 0, st>>>(
            table, V, ld, ids, n, row_scale, (OT*)out, D, gm, wprod, ldo, ldw, gd, ids_stride, rs_stride, skip_invalid, ss);
    } else if (D <= 64 && !wprod && ldo == 0 && ids_stride == 1 && rs_stride == 1 && !skip_invalid) {
        RowGeom gm{D, 64 / D};
        k_gather_rows<1, K, OT><<<(unsigned)mrec_cdiv(n, (int64_t)4 * gm.G * GB), 256, 0, st>>>(
            table, V, ld, ids, n, row_scale, (OT*)out, D, gm);
    } else {
        return MREC_EUNSUPPORTED;
    }
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}

template <class K>
int gather_impl(const float* table, int64_t V, int64_t ld, int32_t D, const K* ids, int64_t n,
                const float* row_scale, float* out, void* stream, bool skip_invalid = false) {
    hipStream_t st = (hipStream_t)stream;
    if (n < 0 || D <= 0 || V < 0 || ld < D) return MREC_EINVAL;
    if (n == 0) return MREC_OK;
    if (V == 0) return MREC_EINVAL;      // rows are read unconditionally at clamped addresses: an empty table has no valid one
    if (!table || !ids || !out) return MREC_EINVAL;
    const bool vec = (D % 4 == 0) && (D <= 256) && (ld % 4 == 0) && al16(table) && al16(out);
    if (vec) {
        RowGeom gm{D / 4, 64 / (D / 4)};
        const int64_t rows_per_block = (int64_t)4 * gm.G * GB;
        k_gather_rows<4, K><<<(unsigned)mrec_cdiv(n, rows_per_block), 256, 0, st>>>(table, V, ld, ids, n, row_scale,
                                                                                  out, D, gm, nullptr, 0, 2, GatherDrop{}, 1, 1, skip_invalid);
    } else if (skip_invalid) {
        return MREC_EUNSUPPORTED;
    } else if (D <= 64) {
        RowGeom gm{D, 64 / D};
        const int64_t rows_per_block = (int64_t)4 * gm.G * GB;
        k_gather_rows<1, K><<<(unsigned)mrec_cdiv(n, rows_per_block), 256, 0, st>>>(table, V, ld, ids, n, row_scale,
                                                                                  out, D, gm);
    } else {
        k_gather_rows_generic<K><<<(unsigned)mrec_cdiv(n, 4), 256, 0, st>>>(table, V, ld, ids, n, row_scale, out, D);
    }
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}

template <class K>
int wide_sum_impl(const float* w, int64_t V, int64_t ldw, const K* ids, const float* wts, int64_t B, int32_t F,
                  const float* bias_dev, float* out, void* stream) {
    if (B < 0 || F <= 0 || V < 0 || ldw < 1) return MREC_EINVAL;
    if (B == 0) return MREC_OK;
    if (!w || !ids || !wts || !out) return MREC_EINVAL;
    if (F <= WS_TILE) {
        int S = WS_TILE / F;
        if (S > 256) S = 256;
        k_wide_sum<K><<<(unsigned)mrec_cdiv(B, S), 256, 0, (hipStream_t)stream>>>(w, V, ldw, ids, wts, B, F, S, bias_dev, out);
    } else {
        k_wide_sum_wideF<K><<<(unsigned)mrec_cdiv(B, 256), 256, 0, (hipStream_t)stream>>>(w, V, ldw, ids, wts, B, F,
                                                                                          bias_dev, out);
    }
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}

inline unsigned stream_grid(int64_t work_items) {
    int64_t b = mrec_cdiv(work_items, 256);
    if (b > 256 * 16) b = 256 * 16;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

MREC_API const char* mrec_strerror(int code) {
    switch (code) {
        case MREC_OK: return "ok";
        case MREC_EINVAL: return "invalid argument";
        case MREC_EWORKSPACE: return "workspace too small";
        case MREC_EUNSUPPORTED: return "unsupported shape";
        case MREC_EHIP: return "HIP runtime error";
        case MREC_ENODEVICE: return "no usable gfx950 device";
        default: return "unknown error";
    }
}

// CRC-32C (Castagnoli, reflected polynomial 0x82F63B78) of a host buffer: the checksum of TFRecord files (the reference's second data
// format, models/wide_deep/src/datasets.py:226-271; mindrec_amd/tfrecord.py).  Host code: data preparation, not the hot path.
MREC_API int mrec_crc32c_host(const void* data, size_t n, uint32_t* out) {
    if (!out || (!data && n)) return MREC_EINVAL;
    static uint32_t table[8][256];
    static const bool init = [] {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
            table[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int t = 1; t < 8; ++t) table[t][i] = (table[t - 1][i] >> 8) ^ table[0][table[t - 1][i] & 0xFF];
        return true;
    }();
    (void)init;
    const unsigned char* p = (const unsigned char*)data;
    uint32_t c = 0xFFFFFFFFu;
    while (n >= 8) {               // slicing by 8
        uint32_t lo, hi;
        memcpy(&lo, p, 4);
        memcpy(&hi, p + 4, 4);
        lo ^= c;
        c = table[7][lo & 0xFF] ^ table[6][(lo >> 8) & 0xFF] ^ table[5][(lo >> 16) & 0xFF] ^ table[4][lo >> 24] ^
            table[3][hi & 0xFF] ^ table[2][(hi >> 8) & 0xFF] ^ table[1][(hi >> 16) & 0xFF] ^ table[0][hi >> 24];
        p += 8;
        n -= 8;
    }
    while (n--) c = table[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
    *out = c ^ 0xFFFFFFFFu;
    return MREC_OK;
}

MREC_API int mrec_last_hip_error(void) { return g_mrec_last_hip_error; }
MREC_API int mrec_version(void) { return 100; }

MREC_API int mrec_device_ok(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return MREC_ENODEVICE;
    hipFuncAttributes fa;
    if (hipFuncGetAttributes(&fa, (const void*)k_fill_normal) != hipSuccess) {
        (void)hipGetLastError();
        return MREC_ENODEVICE;
    }
    return MREC_OK;
}

MREC_API int mrec_fill_normal_f32(float* out, int64_t nrows, int32_t D, int64_t ld, uint64_t seed, int64_t row0,
                                  int64_t row_stride, float sigma, void* stream) {
    if (nrows < 0 || D <= 0 || ld < D) return MREC_EINVAL;
    if (nrows == 0) return MREC_OK;
    if (!out) return MREC_EINVAL;
    const int64_t total = nrows * ((D + 3) / 4);
    k_fill_normal<<<stream_grid(total), 256, 0, (hipStream_t)stream>>>(out, nrows, D, ld, seed, row0, row_stride, sigma);
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}

MREC_API int mrec_gather_rows_f32_i32(const float* table, int64_t V, int64_t ld, int32_t D, const int32_t* ids,
                                      int64_t n, const float* row_scale, float* out, void* stream) {
    return gather_impl<int32_t>(table, V, ld, D, ids, n, row_scale, out, stream);
}
MREC_API int mrec_gather_rows_f32_i64(const float* table, int64_t V, int64_t ld, int32_t D, const int64_t* ids,
                                      int64_t n, const float* row_scale, float* out, void* stream) {
    return gather_impl<int64_t>(table, V, ld, D, ids, n, row_scale, out, stream);
}
/* ... leaving the rows of ids outside [0, V) ALONE instead of writing zeros (MapTensorGet over new keys: mrec_map_lookup_out has
 * written their default rows into `out` already; 16-byte aligned rows, D % 4 == 0, D <= 256). */
MREC_API int mrec_gather_rows_f32_skip_i32(const float* table, int64_t V, int64_t ld, int32_t D, const int32_t* ids,
                                           int64_t n, float* out, void* stream) {
    return gather_impl<int32_t>(table, V, ld, D, ids, n, nullptr, out, stream, true);
}

MREC_API int mrec_gather_rows_bf16_i32(const float* table, int64_t V, int64_t ld, int32_t D, const int32_t* ids,
                                       int64_t n, const float* row_scale, uint16_t* out, void* stream) {
    return gather_bf16_impl<int32_t>(table, V, ld, D, ids, n, row_scale, out, stream);
}
MREC_API int mrec_gather_rows_bf16_i64(const float* table, int64_t V, int64_t ld, int32_t D, const int64_t* ids,
                                       int64_t n, const float* row_scale, uint16_t* out, void* stream) {
    return gather_bf16_impl<int64_t>(table, V, ld, D, ids, n, row_scale, out, stream);
}

MREC_API int mrec_gather_rows_f16_i32(const float* table, int64_t V, int64_t ld, int32_t D, const int32_t* ids,
                                      int64_t n, const float* row_scale, uint16_t* out, void* stream) {
    return gather_bf16_impl<int32_t, f16o_t>(table, V, ld, D, ids, n, row_scale, out, stream);
}
MREC_API int mrec_gather_rows_f16_i64(const float* table, int64_t V, int64_t ld, int32_t D, const int64_t* ids,
                                      int64_t n, const float* row_scale, uint16_t* out, void* stream) {
    return gather_bf16_impl<int64_t, f16o_t>(table, V, ld, D, ids, n, row_scale, out, stream);
}

/* Gather + the wide branch's products in one pass (see include/mrec.h): out_kind 0 = fp32, 1 = bf16, 2 = f16 rows. */
MREC_API int mrec_gather_rows_wide_ex(const float* table, int64_t V, int64_t ld, int32_t D, const void* ids, int32_t id_bytes,
                                      int64_t id_stride, int64_t n, const float* row_scale, int64_t scale_stride, void* out,
                                      int32_t out_kind, int64_t ldo, int32_t wide_col, float* wide_prod, int64_t ldw,
                                      const mrec_dropout_t* drop, int32_t fields, uint32_t flags, void* step_state, void* stream) {
    if ((id_bytes != 4 && id_bytes != 8) || out_kind < 0 || out_kind > 2 || id_stride < 1 || id_stride > (1 << 20) || scale_stride < 1 ||
        scale_stride > (1 << 20))
        return MREC_EINVAL;
    if (!wide_prod || wide_col < 0 || wide_col >= ld) return MREC_EINVAL;
    GatherDrop gd{};
    if (drop) {
        if (out_kind == 0 || fields <= 0 || n % fields || !drop_from(drop, (int64_t)fields * D, &gd.d)) return MREC_EINVAL;
        gd.F = fields;
    }
    if (ldo == D) ldo = 0;
    const int is = (int)id_stride, rs = (int)scale_stride;
    const bool skip = (flags & MREC_GATHER_SKIP_INVALID) != 0;
#define MREC_GW(KT, OT) return gather_bf16_impl<KT, OT>(table, V, ld, D, (const KT*)ids, n, row_scale, (uint16_t*)out, stream, wide_col, \
                                                        wide_prod, ldo, ldw, gd, is, rs, skip, (StepState*)step_state)
    if (id_bytes == 4) {
        if (out_kind == 0) { MREC_GW(int32_t, float); }
        if (out_kind == 1) { MREC_GW(int32_t, bf16o_t); }
        MREC_GW(int32_t, f16o_t);
    }
    if (out_kind == 0) { MREC_GW(int64_t, float); }
    if (out_kind == 1) { MREC_GW(int64_t, bf16o_t); }
    MREC_GW(int64_t, f16o_t);
#undef MREC_GW
}
MREC_API int mrec_gather_rows_wide(const float* table, int64_t V, int64_t ld, int32_t D, const void* ids, int32_t id_bytes,
                                   int64_t n, const float* row_scale, void* out, int32_t out_kind, int64_t ldo, int32_t wide_col,
                                   float* wide_prod, int64_t ldw, const mrec_dropout_t* drop, int32_t fields, void* stream) {
    if (out_kind != 1 && out_kind != 2) return MREC_EINVAL;
    return mrec_gather_rows_wide_ex(table, V, ld, D, ids, id_bytes, 1, n, row_scale, 1, out, out_kind, ldo, wide_col, wide_prod, ldw, drop,
                                    fields, 0u, nullptr, stream);
}

MREC_API int mrec_wide_sum_f32_i32(const float* w, int64_t V, int64_t ldw, const int32_t* ids, const float* wts,
                                   int64_t B, int32_t F, const float* bias_dev, float* out, void* stream) {
    return wide_sum_impl<int32_t>(w, V, ldw, ids, wts, B, F, bias_dev, out, stream);
}
MREC_API int mrec_wide_sum_f32_i64(const float* w, int64_t V, int64_t ldw, const int64_t* ids, const float* wts,
                                   int64_t B, int32_t F, const float* bias_dev, float* out, void* stream) {
    return wide_sum_impl<int64_t>(w, V, ldw, ids, wts, B, F, bias_dev, out, stream);
}

__global__ __launch_bounds__(256) void k_compose_i32(const int* __restrict__ table, const int* __restrict__ idx,
                                                     int64_t n, int* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { const int j = idx[i]; out[i] = j >= 0 ? table[j] : -1; }
}
__global__ __launch_bounds__(256) void k_widen(const int* __restrict__ in, int64_t n, int64_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (int64_t)in[i];
}

// three device-to-device copies in one launch (a step's ids / weights / labels into the static inputs of its HIP graph)
struct Copy3 { uint4* dst[3]; const uint4* src[3]; int64_t n16[3]; };
__global__ __launch_bounds__(256) void k_copy3(Copy3 c) {
    const int64_t stride = (int64_t)gridDim.x * 256;
#pragma unroll
    for (int k = 0; k < 3; ++k)
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < c.n16[k]; i += stride) c.dst[k][i] = c.src[k][i];
}

MREC_API int mrec_copy3(void* dst0, const void* src0, int64_t bytes0, void* dst1, const void* src1, int64_t bytes1, void* dst2,
                        const void* src2, int64_t bytes2, void* stream) {
    void* d[3] = {dst0, dst1, dst2};
    const void* sp[3] = {src0, src1, src2};
    const int64_t b[3] = {bytes0, bytes1, bytes2};
    Copy3 c;
    int64_t most = 0;
    for (int k = 0; k < 3; ++k) {
        if (b[k] < 0 || (b[k] > 0 && (!d[k] || !sp[k]))) return MREC_EINVAL;
        if (b[k] % 16 || (((uintptr_t)d[k] | (uintptr_t)sp[k]) & 15)) return MREC_EUNSUPPORTED;
        c.dst[k] = (uint4*)d[k]; c.src[k] = (const uint4*)sp[k]; c.n16[k] = b[k] / 16;
        if (c.n16[k] > most) most = c.n16[k];
    }
    if (most == 0) return MREC_OK;
    const unsigned g = (unsigned)(mrec_cdiv(most, 256) < 1024 ? mrec_cdiv(most, 256) : 1024);
    k_copy3<<<g, 256, 0, (hipStream_t)stream>>>(c);
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}

// the same for up to 24 tensors (a sink of steps stages all its inputs with one launch)
struct CopyMany { uint4* dst[24]; const uint4* src[24]; int64_t n16[24]; int n; };
__global__ __launch_bounds__(256) void k_copy_many(CopyMany c) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int k = 0; k < c.n; ++k)
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < c.n16[k]; i += stride) c.dst[k][i] = c.src[k][i];
}
MREC_API int mrec_copy_many(int32_t n, void* const* dst, const void* const* src, const int64_t* bytes, void* stream) {
    if (n < 0 || n > 24 || (n && (!dst || !src || !bytes))) return MREC_EINVAL;
    CopyMany c{};
    int64_t most = 0;
    for (int k = 0; k < n; ++k) {
        if (bytes[k] < 0 || (bytes[k] > 0 && (!dst[k] || !src[k]))) return MREC_EINVAL;
        if (bytes[k] % 16 || (((uintptr_t)dst[k] | (uintptr_t)src[k]) & 15)) return MREC_EUNSUPPORTED;
        c.dst[k] = (uint4*)dst[k]; c.src[k] = (const uint4*)src[k]; c.n16[k] = bytes[k] / 16;
        if (c.n16[k] > most) most = c.n16[k];
    }
    c.n = n;
    if (most == 0) return MREC_OK;
    const unsigned g = (unsigned)(mrec_cdiv(most, 256) < 1024 ? mrec_cdiv(most, 256) : 1024);
    k_copy_many<<<g, 256, 0, (hipStream_t)stream>>>(c);
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}

MREC_API int mrec_compose_i32(const int32_t* table, const int32_t* idx, int64_t n, int32_t* out, void* stream) {
    if (n < 0) return MREC_EINVAL;
    if (n == 0) return MREC_OK;
    if (!table || !idx || !out) return MREC_EINVAL;
    k_compose_i32<<<(unsigned)mrec_cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(table, idx, n, out);
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}
MREC_API int mrec_widen_i32_i64(const int32_t* in, int64_t n, int64_t* out, void* stream) {
    if (n < 0) return MREC_EINVAL;
    if (n == 0) return MREC_OK;
    if (!in || !out) return MREC_EINVAL;
    k_widen<<<(unsigned)mrec_cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(in, n, out);
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}

MREC_API int mrec_init_rows_f32(float* table, int64_t ld, int32_t D, const int32_t* rows, const int64_t* keys,
                                const uint8_t* is_new, int64_t n, const int64_t* n_dev, uint64_t seed, float sigma,
                                float fill, void* stream) {
    if (n < 0 || D <= 0 || ld < D) return MREC_EINVAL;
    if (n == 0) return MREC_OK;
    if (!table || !rows || !keys) return MREC_EINVAL;
    k_init_rows<int64_t><<<stream_grid(n), 256, 0, (hipStream_t)stream>>>(table, ld, D, rows, keys, is_new, n,
                                                                           n_dev, seed, sigma, fill);
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}

MREC_API int mrec_scatter_rows_f32(float* table, int64_t ld, int32_t D, const int32_t* rows, int64_t n,
                                   const float* vals, void* stream) {
    if (n < 0 || D <= 0 || ld < D) return MREC_EINVAL;
    if (n == 0) return MREC_OK;
    if (!table || !rows || !vals) return MREC_EINVAL;
    k_scatter_rows<<<(unsigned)mrec_cdiv(n, 4), 256, 0, (hipStream_t)stream>>>(table, ld, D, rows, n, vals);
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}

MREC_API int mrec_move_rows_f32(const float* src, int64_t ld_src, const int64_t* src_rows, float* dst, int64_t ld_dst,
                                const int64_t* dst_rows, int64_t n, const int64_t* n_dev, int32_t W, void* stream) {
    if (n < 0 || W <= 0 || ld_src < W || ld_dst < W) return MREC_EINVAL;
    if (n == 0) return MREC_OK;
    if (!src || !dst || !src_rows || !dst_rows) return MREC_EINVAL;
    const bool v4 = W % 4 == 0 && ld_src % 4 == 0 && ld_dst % 4 == 0 && al16(src) && al16(dst);
    const unsigned grid = (unsigned)mrec_cdiv(n, 4);
    if (v4) k_move_rows<true><<<grid, 256, 0, (hipStream_t)stream>>>(src, ld_src, src_rows, dst, ld_dst, dst_rows, n, n_dev, W);
    else k_move_rows<false><<<grid, 256, 0, (hipStream_t)stream>>>(src, ld_src, src_rows, dst, ld_dst, dst_rows, n, n_dev, W);
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}

static int ftrl1_from(const mrec_ftrl1_t* one, int64_t n, float grad_scale, Ftrl1* out) {
    out->idx = -1;
    out->h.lr = 1.0f; out->h.l1 = 0.0f; out->h.l2 = 0.0f; out->h.lr_power = -0.5f; out->h.gscale = grad_scale;
    if (!one || one->index < 0) return MREC_OK;
    if (one->index >= n || !(one->lr > 0.0f) || one->l1 < 0.0f || one->l2 < 0.0f || one->lr_power > 0.0f) return MREC_EINVAL;
    out->idx = one->index;
    out->h.lr = one->lr; out->h.l1 = one->l1; out->h.l2 = one->l2; out->h.lr_power = one->lr_power;
    return MREC_OK;
}

static int dense_adam_launch(float* p, float* m, float* v, const void* g, int g_bf16, uint16_t* shadow, int64_t n,
                             float lr, float b1, float b2, float eps, float b1_pow, float b2_pow, float grad_scale,
                             int nesterov, void* stream, const mrec_ftrl1_t* one = nullptr) {
    if (n < 0) return MREC_EINVAL;
    Ftrl1 f1;
    if (int rc = ftrl1_from(one, n, grad_scale, &f1)) return rc;
    if (f1.idx >= 0 && g_bf16) return MREC_EUNSUPPORTED;
    if (n == 0) return MREC_OK;
    if (!p || !m || !v || !g) return MREC_EINVAL;
    AdamH h;
    h.lr_t = lr * sqrtf(1.0f - b2_pow) / (1.0f - b1_pow);
    h.b1 = b1; h.b2 = b2; h.omb1 = 1.0f - b1; h.omb2 = 1.0f - b2; h.eps = eps; h.gscale = grad_scale;
    h.nesterov = nesterov;
    hipStream_t st = (hipStream_t)stream;
    const bool al = al16(p) && al16(m) && al16(v) && ((((uintptr_t)g) & (g_bf16 ? 7 : 15)) == 0) &&
                    (!shadow || ((((uintptr_t)shadow) & 7) == 0));
    const int64_t n4 = al ? n / 4 : 0;
    if (n4 > 0) {
        const unsigned gr = stream_grid(n4);
        if (g_bf16) {
            if (shadow) k_dense_adam4_g16<true><<<gr, 256, 0, st>>>((float4*)p, (float4*)m, (float4*)v, (const uint2*)g, n4, h, (uint2*)shadow);
            else k_dense_adam4_g16<false><<<gr, 256, 0, st>>>((float4*)p, (float4*)m, (float4*)v, (const uint2*)g, n4, h, nullptr);
        } else {
            if (shadow) k_dense_adam4<true><<<gr, 256, 0, st>>>((float4*)p, (float4*)m, (float4*)v, (const float4*)g, n4, h, (uint2*)shadow, f1);
            else k_dense_adam4<false><<<gr, 256, 0, st>>>((float4*)p, (float4*)m, (float4*)v, (const float4*)g, n4, h, nullptr, f1);
        }
    }
    const int64_t done = n4 * 4, rem = n - done;
    if (rem > 0) {
        const unsigned gr = stream_grid(rem);
        uint16_t* sh = shadow ? shadow + done : nullptr;
        Ftrl1 fr = f1;
        fr.idx = f1.idx >= done ? f1.idx - done : -1;
        if (g_bf16) {
            const uint16_t* gg = (const uint16_t*)g + done;
            if (sh) k_dense_adam<uint16_t, true><<<gr, 256, 0, st>>>(p + done, m + done, v + done, gg, rem, h, sh, fr);
            else k_dense_adam<uint16_t, false><<<gr, 256, 0, st>>>(p + done, m + done, v + done, gg, rem, h, nullptr, fr);
        } else {
            const float* gg = (const float*)g + done;
            if (sh) k_dense_adam<float, true><<<gr, 256, 0, st>>>(p + done, m + done, v + done, gg, rem, h, sh, fr);
            else k_dense_adam<float, false><<<gr, 256, 0, st>>>(p + done, m + done, v + done, gg, rem, h, nullptr, fr);
        }
    }
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}

MREC_API int mrec_dense_adam_f32(float* p, float* m, float* v, const float* g, int64_t n, float lr, float b1,
                                 float b2, float eps, float b1_pow, float b2_pow, float grad_scale, int nesterov,
                                 void* stream) {
    return dense_adam_launch(p, m, v, g, 0, nullptr, n, lr, b1, b2, eps, b1_pow, b2_pow, grad_scale, nesterov, stream);
}

MREC_API int mrec_dense_adam_ex_f32(float* p, float* m, float* v, const void* g, int g_is_bf16, uint16_t* shadow_bf16,
                                    int64_t n, float lr, float b1, float b2, float eps, float b1_pow, float b2_pow,
                                    float grad_scale, int nesterov, void* stream) {
    return dense_adam_launch(p, m, v, g, g_is_bf16, shadow_bf16, n, lr, b1, b2, eps, b1_pow, b2_pow, grad_scale, nesterov,
                             stream);
}

MREC_API int mrec_dense_adam_l2_workspace_bytes(int64_t n, size_t* out) {
    if (!out || n < 0) return MREC_EINVAL;
    *out = (size_t)stream_grid(n / 4 > 0 ? n / 4 : 1) * sizeof(double) + 256;
    return MREC_OK;
}

MREC_API int mrec_dense_adam_l2_f32(float* p, float* m, float* v, const float* g, int64_t n, float lr, float b1, float b2, float eps,
                                    float b1_pow, float b2_pow, float grad_scale, int nesterov, float l2_scaled, double* sumsq,
                                    int sumsq_accumulate, void* ws, size_t ws_bytes, void* stream) {
    if (n < 0) return MREC_EINVAL;
    if (n == 0) return MREC_OK;
    if (!p || !m || !v || !g) return MREC_EINVAL;
    if (!al16(p) || !al16(m) || !al16(v) || !al16(g)) return MREC_EUNSUPPORTED;
    const int64_t n4 = n / 4;
    const unsigned gr = stream_grid(n4 > 0 ? n4 : 1);
    double* partial = nullptr;
    if (sumsq) {
        MrecArena a(ws, ws_bytes);
        partial = a.take<double>(gr);
        if (!a.ok || !partial) return MREC_EWORKSPACE;
    }
    AdamH h;
    h.lr_t = lr * sqrtf(1.0f - b2_pow) / (1.0f - b1_pow);
    h.b1 = b1; h.b2 = b2; h.omb1 = 1.0f - b1; h.omb2 = 1.0f - b2; h.eps = eps; h.gscale = grad_scale;
    h.nesterov = nesterov;
    hipStream_t st = (hipStream_t)stream;
    k_dense_adam4_l2<<<gr, 256, 0, st>>>((float4*)p, (float4*)m, (float4*)v, (const float4*)g, n4, h, l2_scaled, partial, (int)(n % 4));
    if (sumsq) k_sumsq_finish<<<1, 256, 0, st>>>(partial, (int)gr, sumsq, sumsq_accumulate);
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}

MREC_API int mrec_dense_adam_rows_l2_workspace_bytes(int64_t V, int32_t D, size_t* out) {
    if (!out || V < 0 || D <= 0) return MREC_EINVAL;
    const int64_t items = V * (D % 4 == 0 ? D / 4 : D);
    *out = mrec_align_up((size_t)(V > 0 ? V : 1) * sizeof(int), 256) + (size_t)stream_grid(items > 0 ? items : 1) * sizeof(double) + 512;
    return MREC_OK;
}

MREC_API int mrec_dense_adam_rows_l2_f32(float* p, float* m, float* v, int64_t V, int32_t D, const int32_t* uniq_rows, int64_t U,
                                         const int64_t* n_uniq_dev, const float* sums, float lr, float b1, float b2, float eps, float b1_pow,
                                         float b2_pow, float grad_scale, int nesterov, float l2_scaled, double* sumsq, int sumsq_accumulate,
                                         const void* step_state, void* ws, size_t ws_bytes, void* stream) {
    if (V < 0 || D <= 0 || U < 0) return MREC_EINVAL;
    if (V == 0) return MREC_OK;
    if (!p || !m || !v || !ws || (U > 0 && (!uniq_rows || !sums))) return MREC_EINVAL;
    const bool v4 = D % 4 == 0;
    if (v4 && (!al16(p) || !al16(m) || !al16(v) || (sums && !al16(sums)))) return MREC_EUNSUPPORTED;
    if (V * (int64_t)D >= (int64_t(1) << 40)) return MREC_EUNSUPPORTED;
    const int64_t items = V * (v4 ? D / 4 : D);
    const unsigned gr = stream_grid(items);
    MrecArena a(ws, ws_bytes);
    int* row_group = a.take<int>(V);
    double* partial = sumsq ? a.take<double>(gr) : nullptr;
    if (!a.ok || !row_group || (sumsq && !partial)) return MREC_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    MREC_HIP_CHECK(hipMemsetAsync(row_group, 0xFF, (size_t)V * sizeof(int), st));          // -1 everywhere
    if (U > 0) k_rows_to_groups<<<stream_grid(U), 256, 0, st>>>(uniq_rows, n_uniq_dev, U, V, row_group);
    AdamH h;
    h.lr_t = lr * sqrtf(1.0f - b2_pow) / (1.0f - b1_pow);
    h.b1 = b1; h.b2 = b2; h.omb1 = 1.0f - b1; h.omb2 = 1.0f - b2; h.eps = eps; h.gscale = grad_scale;
    h.nesterov = nesterov;
    if (v4) k_dense_adam_rows_l2<4><<<gr, 256, 0, st>>>(p, m, v, V, D, row_group, sums, h, l2_scaled, partial, (const StepState*)step_state);
    else k_dense_adam_rows_l2<1><<<gr, 256, 0, st>>>(p, m, v, V, D, row_group, sums, h, l2_scaled, partial, (const StepState*)step_state);
    if (sumsq) k_sumsq_finish<<<1, 256, 0, st>>>(partial, (int)gr, sumsq, sumsq_accumulate);
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}

MREC_API int mrec_dense_adam_one_ftrl_f32(float* p, float* m, float* v, const float* g, int64_t n, float lr, float b1, float b2,
                                          float eps, float b1_pow, float b2_pow, float grad_scale, int nesterov,
                                          const mrec_ftrl1_t* one_ftrl, void* stream) {
    return dense_adam_launch(p, m, v, g, 0, nullptr, n, lr, b1, b2, eps, b1_pow, b2_pow, grad_scale, nesterov, stream, one_ftrl);
}

static int dense_adam_slabs_launch(float* p, float* m, float* v, const float* g, void* shadow16, int shadow_kind,
                                   int64_t n, int32_t nseg, const float* const* slabs, const int64_t* starts,
                                   const int64_t* lens, const int32_t* splits, float lr, float b1, float b2, float eps,
                                   float b1_pow, float b2_pow, float grad_scale, int nesterov, void* step_state,
                                   const mrec_ftrl1_t* one, void* stream) {
    if (n < 0 || nseg < 0 || nseg > 16 || shadow_kind < 0 || shadow_kind > 2) return MREC_EINVAL;
    Ftrl1 f1;
    if (int rc = ftrl1_from(one, n, grad_scale, &f1)) return rc;
    const StepState* ss = (const StepState*)step_state;
    if (n == 0) return MREC_OK;
    if (!p || !m || !v || !g || (nseg > 0 && (!slabs || !starts || !lens || !splits)) || (shadow_kind && !shadow16))
        return MREC_EINVAL;
    if (n % 4 || !al16(p) || !al16(m) || !al16(v) || !al16(g) || (shadow16 && (((uintptr_t)shadow16) & 7)))
        return MREC_EUNSUPPORTED;
    SlabSegs sg;
    sg.n = nseg;
    for (int q = 0; q < nseg; ++q) {
        if (!slabs[q] || starts[q] < 0 || lens[q] <= 0 || starts[q] % 4 || lens[q] % 4 || starts[q] + lens[q] > n ||
            splits[q] <= 0 || !al16(slabs[q]))
            return MREC_EINVAL;
        sg.part[q] = (const float4*)slabs[q];
        sg.start4[q] = starts[q] / 4;
        sg.len4[q] = lens[q] / 4;
        sg.S[q] = splits[q];
    }
    AdamH h;
    h.lr_t = lr * sqrtf(1.0f - b2_pow) / (1.0f - b1_pow);
    h.b1 = b1; h.b2 = b2; h.omb1 = 1.0f - b1; h.omb2 = 1.0f - b2; h.eps = eps; h.gscale = grad_scale;
    h.nesterov = nesterov;
    const int64_t n4 = n / 4;
    hipStream_t st = (hipStream_t)stream;
    const unsigned gr = stream_grid(n4);
    if (shadow_kind == 1) k_dense_adam4_slabs<1><<<gr, 256, 0, st>>>((float4*)p, (float4*)m, (float4*)v, (const float4*)g, n4, h, (uint2*)shadow16, sg, ss, f1);
    else if (shadow_kind == 2) k_dense_adam4_slabs<2><<<gr, 256, 0, st>>>((float4*)p, (float4*)m, (float4*)v, (const float4*)g, n4, h, (uint2*)shadow16, sg, ss, f1);
    else k_dense_adam4_slabs<0><<<gr, 256, 0, st>>>((float4*)p, (float4*)m, (float4*)v, (const float4*)g, n4, h, nullptr, sg, ss, f1);
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}

MREC_API int mrec_dense_adam_slabs_f32(float* p, float* m, float* v, const float* g, void* shadow16, int shadow_kind,
                                       int64_t n, int32_t nseg, const float* const* slabs, const int64_t* starts,
                                       const int64_t* lens, const int32_t* splits, float lr, float b1, float b2, float eps,
                                       float b1_pow, float b2_pow, float grad_scale, int nesterov, void* step_state,
                                       void* stream) {
    return dense_adam_slabs_launch(p, m, v, g, shadow16, shadow_kind, n, nseg, slabs, starts, lens, splits, lr, b1, b2, eps, b1_pow,
                                   b2_pow, grad_scale, nesterov, step_state, nullptr, stream);
}

MREC_API int mrec_dense_adam_slabs_one_ftrl_f32(float* p, float* m, float* v, const float* g, void* shadow16, int shadow_kind,
                                                int64_t n, int32_t nseg, const float* const* slabs, const int64_t* starts,
                                                const int64_t* lens, const int32_t* splits, float lr, float b1, float b2, float eps,
                                                float b1_pow, float b2_pow, float grad_scale, int nesterov, void* step_state,
                                                const mrec_ftrl1_t* one_ftrl, void* stream) {
    return dense_adam_slabs_launch(p, m, v, g, shadow16, shadow_kind, n, nseg, slabs, starts, lens, splits, lr, b1, b2, eps, b1_pow,
                                   b2_pow, grad_scale, nesterov, step_state, one_ftrl, stream);
}

MREC_API int mrec_dense_sum_slab_segments_f32(float* g, int64_t n, int32_t nseg, const float* const* slabs, const int64_t* starts,
                                              const int64_t* lens, const int32_t* splits, void* stream) {
    if (n < 0 || nseg < 0 || nseg > 16) return MREC_EINVAL;
    if (nseg == 0) return MREC_OK;
    if (!g || !slabs || !starts || !lens || !splits) return MREC_EINVAL;
    if (!al16(g)) return MREC_EUNSUPPORTED;
    SlabSegs sg;
    sg.n = nseg;
    int64_t total4 = 0;
    for (int q = 0; q < nseg; ++q) {
        if (!slabs[q] || starts[q] < 0 || lens[q] <= 0 || starts[q] % 4 || lens[q] % 4 || starts[q] + lens[q] > n || splits[q] <= 0 ||
            !al16(slabs[q]))
            return MREC_EINVAL;
        sg.part[q] = (const float4*)slabs[q];
        sg.start4[q] = starts[q] / 4;
        sg.len4[q] = lens[q] / 4;
        sg.S[q] = splits[q];
        total4 += lens[q] / 4;
    }
    k_sum_slab_segments<<<stream_grid(total4), 256, 0, (hipStream_t)stream>>>((float4*)g, sg, total4);
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}

MREC_API int mrec_dense_ftrl_f32(float* var, float* accum, float* linear, const float* g, int64_t n, float lr,
                                 float l1, float l2, float lr_power, float grad_scale, void* stream) {
    if (n < 0) return MREC_EINVAL;
    if (n == 0) return MREC_OK;
    if (!var || !accum || !linear || !g) return MREC_EINVAL;
    FtrlH h{lr, l1, l2, lr_power, grad_scale};
    k_dense_ftrl<<<stream_grid(n), 256, 0, (hipStream_t)stream>>>(var, accum, linear, g, n, h);
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}
