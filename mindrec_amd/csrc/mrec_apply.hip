// mrec_apply.hip -- sparse gradient apply: UnsortedSegmentSum fused with the LazyAdam / FTRL row
// update (or a plain store), for gfx950.
//
// Replaces, in one HBM pass, what the reference gets from MindSpore as
//   RowTensor dedup (Unique + UnsortedSegmentSum)  ->  FusedSparseLazyAdam / FusedSparseFtrl
// for the optimizers built at models/wide_deep/src/wide_and_deep.py:415-433 and applied at :490-492
// (formulas: SURVEY.md A.4, A.5).
//
// Why no atomics: float atomics run at ~1.3 TB/s chip-wide on MI355X against ~6 TB/s for plain
// loads/stores, and their sum order is not reproducible.  Instead the step's inverted index
// (mrec_group_by_inverse) lists, per unique id, the positions that reference it, in ascending
// order.  The sorted list is cut into windows of AW entries; a lane-group of lpr = D/4 lanes
// (float4 per lane; D = 80 -> 20 lanes, 3 groups per wave64) walks one window sequentially,
// accumulating the gathered gradient rows in registers, and at the end of every id's run reads
// p/m/v (or var/accum/linear) once, updates, and writes them back once.  Loads for AB entries
// (gradient rows and state rows) are issued together before the first dependent op, so every
// lane-group keeps up to 4*AB 16-byte loads in flight.
//
// (Tried in round 4 and not kept: the gradient rows of a batch's two entries fetched by ONE 16-byte-per-lane load -- even lanes 8
// columns of the first entry's row, odd lanes of the second's -- with the halves traded over the DPP network, the trick that took
// the lookup's stores from 8 to 16 bytes per lane (mrec_gather.hip, k_gather_rows_w16).  Bit-identical, and slower: uniform ids
// 188 -> 201 us, Zipf x 39 fields 122 -> 128, Zipf x 26 fields 109 -> 127 (kernel alone, HIP events): at 128 VGPRs the exchange's
// registers and selects cost more than the halved load count returns.)
//
// Runs that cross a window boundary (duplicate-heavy ids, e.g. Criteo's 13 constant dense-field ids
// with 16384 copies each) leave per-window partial sums in a carry buffer, and every window writes a
// flag (0 = no run continues past it, 1 = it owns a run with few partials, 2 = a long run): no list,
// no counter, no atomic.  A second launch (k_apply_long) finishes the short runs with one lane-group
// each, partials in order, and the long ones with a whole 256-thread block in a fixed tree order.
// Both passes are bitwise reproducible.
#include <type_traits>

#include "mrec_common.h"
#include "mrec_optim.h"
#include "mrec_dense_adam.h"

namespace {

// Per-path tuning (measured on MI355X, V = 200 M, D = 80, batch 16384 x 26, uniform ids):
//   float4 path : nontemporal loads/stores +10 % (rows are touched once per step: keep them out of
//                 L2/MALL), 2 entries per batch +4 % over 4 (8 is 15 % slower: past saturation more
//                 requests in flight only lengthen the queues), 8-entry windows (16 were best while crossing runs were
//                 collected through an atomic list; with per-window flags 8 wins: 178 -> 172 us).
//   scalar path (D = 1 wide table, odd D): cached accesses (the three 4-byte state words of a
//                 fused w|accum|linear record share one 64-B sector), 8-entry windows.
#ifndef MREC_AW4
#define MREC_AW4 8
#endif
#ifndef MREC_AB4
#define MREC_AB4 2
#endif
#ifndef MREC_NT4
#define MREC_NT4 1
#endif
#ifndef MREC_GP4
#define MREC_GP4 2
#endif
#ifndef MREC_WPS4
#define MREC_WPS4 4         // waves per SIMD asked of the register allocator for the float4 + wide-lane kernel (128 VGPRs: the
#endif                      // window's index words live in registers; 5 waves measured 5 % slower; 3 the same on uniform ids but Zipf ids x 39
                            // fields 96 -> 106 us -- round 5, a variant that needed 139 registers)
#ifndef MREC_PAIRS
#define MREC_PAIRS 1                 // a run of two entries that straddles a window boundary is summed by the window its first entry lies in (k_apply_main)
#endif
#ifndef MREC_PAIRS_DUP_DIV
#define MREC_PAIRS_DUP_DIV 16        // ... in batches with at most n / 16 duplicate positions
#endif
#ifndef MREC_APPLY_MAXB
#define MREC_APPLY_MAXB 4096u        // workgroups of k_apply_main (4 waves each; 1024 are resident at 4 waves per SIMD).  Sweep in profiles/r03_apply_chain.txt:
                                     // duplicate-heavy ids want a cap (Zipf x 39 fields: 121 us uncapped, 94-95 at 1024-4096), uniform ids none (172 / 176 us at 4096 / 2048)
#endif
#ifndef MREC_LONG_AB
#define MREC_LONG_AB 16              // partials in flight per lane-group in k_apply_long's pass over a long run (32: 13.5 -> 36 us on Zipf x 39 fields)
#endif
#ifndef MREC_GP1
#define MREC_GP1 4
#endif
#ifndef MREC_AW1
#define MREC_AW1 8
#endif
#ifndef MREC_AB1
#define MREC_AB1 4
#endif
#ifndef MREC_NT1
#define MREC_NT1 0
#endif
template <int VEC> struct ACfg;
template <> struct ACfg<4> { static constexpr int AW = MREC_AW4, AB = MREC_AB4, GP = MREC_GP4; static constexpr bool NT = MREC_NT4; };
template <> struct ACfg<2> { static constexpr int AW = MREC_AW1, AB = MREC_AB1, GP = MREC_GP1; static constexpr bool NT = MREC_NT1; };  // 8-byte path: same window as scalar
template <> struct ACfg<1> { static constexpr int AW = MREC_AW1, AB = MREC_AB1, GP = MREC_GP1; static constexpr bool NT = MREC_NT1; };
static_assert(MREC_GP4 % MREC_AB4 == 0 && MREC_GP1 % MREC_AB1 == 0, "gradient prefetch depth must be a multiple of the state batch depth");
constexpr int AW_MIN = (MREC_AW4 < MREC_AW1) ? MREC_AW4 : MREC_AW1;

template <int VEC> struct Vf;
template <> struct Vf<4> { float4 v; };
template <> struct Vf<2> { float2 v; };
template <> struct Vf<1> { float v; };

typedef float mrec_f4 __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ void vload(Vf<4>& r, const float* p) {
    if (NT) {
        mrec_f4 t = __builtin_nontemporal_load((const mrec_f4*)p);
        r.v = make_float4(t.x, t.y, t.z, t.w);
    } else {
        r.v = *(const float4*)p;
    }
}
typedef float mrec_f2 __attribute__((ext_vector_type(2)));
template <bool NT> __device__ __forceinline__ void vload(Vf<2>& r, const float* p) {
    if (NT) {
        mrec_f2 t = __builtin_nontemporal_load((const mrec_f2*)p);
        r.v = make_float2(t.x, t.y);
    } else {
        r.v = *(const float2*)p;
    }
}
template <bool NT> __device__ __forceinline__ void vload(Vf<1>& r, const float* p) {
    r.v = NT ? __builtin_nontemporal_load(p) : *p;
}
// bf16 row gradients (what a mixed-precision MLP backward produces): widened exactly (bits << 16),
// so summing them is bit-identical to casting the gradient tensor to fp32 first -- at half the bytes.
struct bf16_t { uint16_t v; };
template <bool NT> __device__ __forceinline__ void vload(Vf<4>& r, const bf16_t* p) {
    uint2 u;
    if (NT) {
        typedef unsigned int mrec_u2 __attribute__((ext_vector_type(2)));
        mrec_u2 t = __builtin_nontemporal_load((const mrec_u2*)p);
        u = make_uint2(t.x, t.y);
    } else {
        u = *(const uint2*)p;
    }
    r.v = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16),
                      __uint_as_float(u.y & 0xFFFF0000u));
}
template <bool NT> __device__ __forceinline__ void vload(Vf<2>& r, const bf16_t* p) {
    const unsigned int u = NT ? __builtin_nontemporal_load((const unsigned int*)p) : *(const unsigned int*)p;
    r.v = make_float2(__uint_as_float(u << 16), __uint_as_float(u & 0xFFFF0000u));
}
template <bool NT> __device__ __forceinline__ void vload(Vf<1>& r, const bf16_t* p) {
    r.v = __uint_as_float(((unsigned int)p->v) << 16);
}
// IEEE half row gradients (the reference's mixed-precision dtype): widened exactly by v_cvt_f32_f16.
struct f16_t { uint16_t v; };
__device__ __forceinline__ float2 mrec_h2f2(unsigned int u) {
    typedef _Float16 mrec_h2 __attribute__((ext_vector_type(2)));
    const mrec_h2 h = __builtin_bit_cast(mrec_h2, u);
    return make_float2((float)h[0], (float)h[1]);
}
template <bool NT> __device__ __forceinline__ void vload(Vf<4>& r, const f16_t* p) {
    uint2 u;
    if (NT) {
        typedef unsigned int mrec_u2 __attribute__((ext_vector_type(2)));
        mrec_u2 t = __builtin_nontemporal_load((const mrec_u2*)p);
        u = make_uint2(t.x, t.y);
    } else {
        u = *(const uint2*)p;
    }
    const float2 a = mrec_h2f2(u.x), b = mrec_h2f2(u.y);
    r.v = make_float4(a.x, a.y, b.x, b.y);
}
template <bool NT> __device__ __forceinline__ void vload(Vf<2>& r, const f16_t* p) {
    const unsigned int u = NT ? __builtin_nontemporal_load((const unsigned int*)p) : *(const unsigned int*)p;
    r.v = mrec_h2f2(u);
}
template <bool NT> __device__ __forceinline__ void vload(Vf<1>& r, const f16_t* p) {
    r.v = (float)__builtin_bit_cast(_Float16, p->v);
}
// The bytes of VEC gradient values as loaded, widened where they are USED: a load whose value is converted inside the
// conditional block that issues it is waited for there (s_waitcnt vmcnt(0) in every such block -- the two gradient rows and the
// state rows of a batch were three dependent round trips, not one).
template <int VEC, class GT> struct GBits { Vf<VEC> f; };
template <> struct GBits<4, bf16_t> { unsigned x, y; };
template <> struct GBits<2, bf16_t> { unsigned x; };
template <> struct GBits<1, bf16_t> { unsigned x; };
template <> struct GBits<4, f16_t> { unsigned x, y; };
template <> struct GBits<2, f16_t> { unsigned x; };
template <> struct GBits<1, f16_t> { unsigned x; };
template <bool NT, int VEC> __device__ __forceinline__ void gload(GBits<VEC, float>& b, const float* p) { vload<NT>(b.f, p); }
#define MREC_GLOAD16(G16)                                                                                              \
    template <bool NT> __device__ __forceinline__ void gload(GBits<4, G16>& b, const G16* p) {                          \
        typedef unsigned int mrec_u2 __attribute__((ext_vector_type(2)));                                               \
        const mrec_u2 t = NT ? __builtin_nontemporal_load((const mrec_u2*)p) : *(const mrec_u2*)p;                      \
        b.x = t.x; b.y = t.y;                                                                                           \
    }                                                                                                                   \
    template <bool NT> __device__ __forceinline__ void gload(GBits<2, G16>& b, const G16* p) {                          \
        b.x = NT ? __builtin_nontemporal_load((const unsigned int*)p) : *(const unsigned int*)p;                        \
    }                                                                                                                   \
    template <bool NT> __device__ __forceinline__ void gload(GBits<1, G16>& b, const G16* p) { b.x = p->v; }
MREC_GLOAD16(bf16_t)
MREC_GLOAD16(f16_t)
#undef MREC_GLOAD16
template <int VEC> __device__ __forceinline__ void gwiden(Vf<VEC>& r, const GBits<VEC, float>& b) { r = b.f; }
__device__ __forceinline__ void gwiden(Vf<4>& r, const GBits<4, bf16_t>& b) {
    r.v = make_float4(__uint_as_float(b.x << 16), __uint_as_float(b.x & 0xFFFF0000u), __uint_as_float(b.y << 16),
                      __uint_as_float(b.y & 0xFFFF0000u));
}
__device__ __forceinline__ void gwiden(Vf<2>& r, const GBits<2, bf16_t>& b) {
    r.v = make_float2(__uint_as_float(b.x << 16), __uint_as_float(b.x & 0xFFFF0000u));
}
__device__ __forceinline__ void gwiden(Vf<1>& r, const GBits<1, bf16_t>& b) { r.v = __uint_as_float(b.x << 16); }
__device__ __forceinline__ void gwiden(Vf<4>& r, const GBits<4, f16_t>& b) {
    const float2 lo = mrec_h2f2(b.x), hi = mrec_h2f2(b.y);
    r.v = make_float4(lo.x, lo.y, hi.x, hi.y);
}
__device__ __forceinline__ void gwiden(Vf<2>& r, const GBits<2, f16_t>& b) { r.v = mrec_h2f2(b.x); }
__device__ __forceinline__ void gwiden(Vf<1>& r, const GBits<1, f16_t>& b) { r.v = (float)__builtin_bit_cast(_Float16, (uint16_t)b.x); }

// Gradient load of the float4 path with a wide lane in the group: ONE load instruction for all lanes (the wide lane's
// address points into gw); the wide lane keeps the first dword as the fp32 it is, the others widen their 16-bit values.
template <bool NT> __device__ __forceinline__ void vload_w(Vf<4>& r, const float* p, bool) { vload<NT>(r, p); }
template <bool NT> __device__ __forceinline__ void vload_w(Vf<4>& r, const bf16_t* p, bool wl) {
    typedef unsigned int mrec_u2 __attribute__((ext_vector_type(2)));
    const mrec_u2 t = NT ? __builtin_nontemporal_load((const mrec_u2*)p) : *(const mrec_u2*)p;
    r.v = make_float4(__uint_as_float(wl ? t.x : (t.x << 16)), __uint_as_float(t.x & 0xFFFF0000u), __uint_as_float(t.y << 16),
                      __uint_as_float(t.y & 0xFFFF0000u));
}
template <bool NT> __device__ __forceinline__ void vload_w(Vf<4>& r, const f16_t* p, bool wl) {
    typedef unsigned int mrec_u2 __attribute__((ext_vector_type(2)));
    const mrec_u2 t = NT ? __builtin_nontemporal_load((const mrec_u2*)p) : *(const mrec_u2*)p;
    const float2 a = mrec_h2f2(t.x), b = mrec_h2f2(t.y);
    r.v = make_float4(wl ? __uint_as_float(t.x) : a.x, a.y, b.x, b.y);
}
template <bool NT, class G> __device__ __forceinline__ void vload_w(Vf<2>&, const G*, bool) {}
template <bool NT, class G> __device__ __forceinline__ void vload_w(Vf<1>&, const G*, bool) {}
template <bool NT> __device__ __forceinline__ void vstore(float* p, const Vf<4>& x) {
    if (NT) {
        mrec_f4 t = {x.v.x, x.v.y, x.v.z, x.v.w};
        __builtin_nontemporal_store(t, (mrec_f4*)p);
    } else {
        *(float4*)p = x.v;
    }
}
template <bool NT> __device__ __forceinline__ void vstore(float* p, const Vf<2>& x) {
    if (NT) {
        mrec_f2 t = {x.v.x, x.v.y};
        __builtin_nontemporal_store(t, (mrec_f2*)p);
    } else {
        *(float2*)p = x.v;
    }
}
template <bool NT> __device__ __forceinline__ void vstore(float* p, const Vf<1>& x) {
    if (NT) __builtin_nontemporal_store(x.v, p); else *p = x.v;
}
__device__ __forceinline__ void vzero(Vf<4>& r) { r.v = make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void vzero(Vf<2>& r) { r.v = make_float2(0.f, 0.f); }
__device__ __forceinline__ void vzero(Vf<1>& r) { r.v = 0.f; }
__device__ __forceinline__ void vmul(Vf<4>& x, float s) { x.v.x *= s; x.v.y *= s; x.v.z *= s; x.v.w *= s; }
__device__ __forceinline__ void vmul(Vf<2>& x, float s) { x.v.x *= s; x.v.y *= s; }
__device__ __forceinline__ void vmul(Vf<1>& x, float s) { x.v *= s; }
__device__ __forceinline__ void vadd(Vf<4>& a, const Vf<4>& b) {
    a.v.x = a.v.x + b.v.x; a.v.y = a.v.y + b.v.y; a.v.z = a.v.z + b.v.z; a.v.w = a.v.w + b.v.w;
}
__device__ __forceinline__ void vadd(Vf<2>& a, const Vf<2>& b) { a.v.x = a.v.x + b.v.x; a.v.y = a.v.y + b.v.y; }
__device__ __forceinline__ void vadd(Vf<1>& a, const Vf<1>& b) { a.v = a.v + b.v; }

// "This value is needed HERE": an empty asm with the registers as in-out operands.  The compiler places its s_waitcnt for a
// load at the first use of the value; with the first use inside a conditional block, the wait at the next join also covers
// every store issued on the way (vmcnt counts loads and stores in order) -- a store round trip in the middle of a batch.
__device__ __forceinline__ void vtouch(Vf<4>& r) { asm volatile("" : "+v"(r.v.x), "+v"(r.v.y), "+v"(r.v.z), "+v"(r.v.w)); }
__device__ __forceinline__ void vtouch(Vf<2>& r) { asm volatile("" : "+v"(r.v.x), "+v"(r.v.y)); }
__device__ __forceinline__ void vtouch(Vf<1>& r) { asm volatile("" : "+v"(r.v)); }

template <int VEC> __device__ __forceinline__ void gtouch(GBits<VEC, float>& b) { vtouch(b.f); }
#define MREC_GTOUCH16(G16)                                                                                         \
    __device__ __forceinline__ void gtouch(GBits<4, G16>& b) { asm volatile("" : "+v"(b.x), "+v"(b.y)); }         \
    __device__ __forceinline__ void gtouch(GBits<2, G16>& b) { asm volatile("" : "+v"(b.x)); }                     \
    __device__ __forceinline__ void gtouch(GBits<1, G16>& b) { asm volatile("" : "+v"(b.x)); }
MREC_GTOUCH16(bf16_t)
MREC_GTOUCH16(f16_t)
#undef MREC_GTOUCH16

// ---- updaters: NS state arrays, each [V, ld]; apply() sees one lane's VEC elements -------------
struct UpdAdam {
    static constexpr int NS = 3;
    static constexpr bool kLoad = true;
    float* s[3];
    AdamH h;
    __device__ __forceinline__ void elem(float* st, float g) const { adam_elem(st[0], st[1], st[2], g, h); }
};
struct UpdFtrl {
    static constexpr int NS = 3;
    static constexpr bool kLoad = true;
    float* s[3];
    FtrlH h;
    __device__ __forceinline__ void elem(float* st, float g) const { ftrl_elem(st[0], st[1], st[2], g, h); }
};
struct UpdStore {  // UnsortedSegmentSum: out[u,:] = sum
    static constexpr int NS = 1;
    static constexpr bool kLoad = false;
    float* s[1];
    __device__ __forceinline__ void elem(float* st, float g) const { st[0] = g; }
};

template <class Upd>
__device__ __forceinline__ void upd_apply(const Upd& u, Vf<4> (&st)[Upd::NS], const Vf<4>& g) {
    float a[Upd::NS];
#define MREC_COMP(c)                                              \
    for (int i = 0; i < Upd::NS; ++i) a[i] = st[i].v.c;           \
    u.elem(a, g.v.c);                                             \
    for (int i = 0; i < Upd::NS; ++i) st[i].v.c = a[i];
    MREC_COMP(x) MREC_COMP(y) MREC_COMP(z) MREC_COMP(w)
#undef MREC_COMP
}
template <class Upd>
__device__ __forceinline__ void upd_apply(const Upd& u, Vf<2> (&st)[Upd::NS], const Vf<2>& g) {
    float a[Upd::NS];
    for (int i = 0; i < Upd::NS; ++i) a[i] = st[i].v.x;
    u.elem(a, g.v.x);
    for (int i = 0; i < Upd::NS; ++i) st[i].v.x = a[i];
    for (int i = 0; i < Upd::NS; ++i) a[i] = st[i].v.y;
    u.elem(a, g.v.y);
    for (int i = 0; i < Upd::NS; ++i) st[i].v.y = a[i];
}
template <class Upd>
__device__ __forceinline__ void upd_apply(const Upd& u, Vf<1> (&st)[Upd::NS], const Vf<1>& g) {
    float a[Upd::NS];
    for (int i = 0; i < Upd::NS; ++i) a[i] = st[i].v;
    u.elem(a, g.v);
    for (int i = 0; i < Upd::NS; ++i) st[i].v = a[i];
}

__device__ __forceinline__ void vset_x(Vf<4>& r, float x) { r.v.x = x; }
__device__ __forceinline__ void vset_x(Vf<2>& r, float x) { r.v.x = x; }
__device__ __forceinline__ void vset_x(Vf<1>& r, float x) { r.v = x; }
// FTRL on the wide record held as one float4 [w, accum, linear, pad]; g = the summed gradient in .x
template <int VEC> __device__ __forceinline__ void wide_apply(Vf<VEC>&, const Vf<VEC>&, const FtrlH&) {}
template <> __device__ __forceinline__ void wide_apply<4>(Vf<4>& st, const Vf<4>& g, const FtrlH& h) {
    ftrl_elem(st.v.x, st.v.y, st.v.z, g.v.x, h);
}

struct ApplyGeom { int lpr; int G; int D; };

// The "wide lane": with WIDE the lane-group gets one lane more (sub == lpr - 1), which owns the wide table's FTRL record
// [w | accum | linear | pad] (one float4 at column wcol of the SAME fused row, s[0] + row * ld + wcol) instead of four
// columns of p / m / v.  Its gradient per position is gw[pos / F] (the head's dlogit of the sample, wide_and_deep.py:304-306
// bprop), scaled and summed exactly like the other lanes' columns -- so Unique + UnsortedSegmentSum + FusedSparseFtrl of
// the wide table (wide_and_deep.py:423-430) ride the deep table's pass: one row visit instead of two kernels.
// Cost model (measured): the kernel is bound by the cache-line requests it makes per row, not by bytes or instructions.
//   * the wide record is loaded / stored by the instruction that loads / stores p: wcol == D puts it right behind p's last
//     lane (contiguous 16 bytes more, the same 128-byte line when rows are 128-byte aligned) -- no extra request;
//   * for m and v the wide lane is simply masked off;
//   * the per-sample gradient gw[pos / F] is one group-uniform, cached load (pos / F by multiply-high with `magic`).
// Two earlier forms cost +35 % (separate nontemporal load + store of the record: the line is fetched again) and +55 %
// (the wide lane pointed at dummy lines for m / v: two more line requests per load and store).
struct WideArgs { const float* gw; int F; int wcol; FtrlH h; unsigned magic; float* dummy; unsigned gws; };

// uniq == nullptr means "row = group number" (segment-sum into a dense [U, D] output).
template <class K>
__device__ __forceinline__ int64_t seg_row(const K* uniq, int seg) {
    return uniq ? (int64_t)uniq[seg] : (int64_t)seg;
}

// ---- constant columns ------------------------------------------------------------------------------------------------------
// Criteo's 13 dense features are fields whose id is the SAME in every sample (datasets/criteo_1tb/process_data.py:138-147: one id per
// dense field, the feature's value is the weight): a run of B entries in the sorted index, 2048 window partials and a block's worth
// of finishing work each -- a third of the positions of a 39-field batch.  For such a column the sum over the batch needs no index
// at all: row b of the gradient matrix holds the column's 160 bytes at a fixed offset, neighbouring constant columns next to it.
// k_const_cols (below) finds the columns whose id is one and the same in all B samples and occurs in no other column; the windows
// of k_apply_main skip their entries (position -> field -> a bit of the mask), the first `cblocks` workgroups of the SAME launch sum
// the columns sample by sample (a lane-group walks ONE column over a chunk of `rr` consecutive samples, eight in flight; products and
// order of operations as in the windows), and the finishing launch adds a column's chunk sums in chunk order and updates its one
// row.  Fixed order: bitwise reproducible.  (First form, measured on Zipf ids x 39 fields: a lane-group = 16 samples x all columns,
// 32 dependent round trips -- as long as the whole launch -- and 1024 partial rows per column for the finishing pass: k_apply_main
// no faster, the finishing pass 44 -> 66 us.  As built: k_apply_main 91.9 -> 84.5 us, finishing pass 42.9 -> 27.4 us,
// profiles/r05_const_cols_ab.txt.)
constexpr int kConstMax = 64;        // hot columns handled: every field can be one (F <= 64)
constexpr int kConstLG = 1024;       // chunks of 64 consecutive samples the partial-sum pass cuts the batch into, at most (B <= 65536)
struct ConstCols { const unsigned long long* mask; const long long* hid; const void* ids0; float* part; int B, rr, nlg; unsigned cblocks; int id_bytes; };

// bit f: field f is a constant column (at most kConstMax bits; written by k_const_cols' last workgroup: one scalar load here)
__device__ __forceinline__ unsigned long long const_mask(const unsigned long long* __restrict__ mask) { return *mask; }

template <class K, class GT>
__device__ __forceinline__ void const_part_body(const ConstCols& cc, unsigned long long cmask, const GT* __restrict__ g, int64_t ldg,
                                                const float* __restrict__ rscale, float gscale, const WideArgs& wa, ApplyGeom gm) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane / gm.lpr, sub = lane - grp * gm.lpr;
    // a lane-group = (chunk of rr <= 64 consecutive samples, hot column j): neighbouring lane-groups take neighbouring columns of the
    // same samples (one contiguous stretch of a gradient row where the hot columns are adjacent fields).  First which of the chunk's
    // samples hold the column's hot id (every lane looks at a few, a ballot collects them: all of them in a constant column), then
    // those samples' gradient rows, eight in flight, in sample order.
    const int nc = __popcll(cmask);
    const int lg = ((int)blockIdx.x * 4 + wave) * gm.G + grp;
    const bool live = grp < gm.G && lg < cc.nlg * nc;
    const int chunk = live ? lg / nc : 0, j = live ? lg - chunk * nc : 0;
    int f = 0;
    {
        unsigned long long m = cmask;
        for (int q = 0; q < j; ++q) m &= m - 1ull;
        f = (int)__ffsll((long long)m) - 1;
    }
    const long long hid = cc.hid[f];
    const int b0 = chunk * cc.rr, b1 = (b0 + cc.rr < cc.B) ? b0 + cc.rr : cc.B;
    unsigned long long mm = 0ull;                        // bit i: sample b0 + i holds the hot id
    for (int r = 0; r * gm.lpr < cc.rr; ++r) {
        const int i = r * gm.lpr + sub;
        const bool pred = live && i < cc.rr && b0 + i < b1 && (long long)((const K*)cc.ids0)[(int64_t)(b0 + i) * wa.F + f] == hid;
        const unsigned long long bal = __ballot(pred);    // (every lane of the wave arrives here: no early return above)
        mm |= ((bal >> (grp * gm.lpr)) & ((1ull << gm.lpr) - 1ull)) << (r * gm.lpr);
    }
    if (!live) return;
    const bool wl = sub == gm.lpr - 1;
    const int ccol = sub * 4;
    Vf<4> acc;
    vzero(acc);
    bool first = true;
    while (mm) {
        int bq[8];
        unsigned long long m2 = mm;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            bq[q] = m2 ? b0 + (int)__ffsll((long long)m2) - 1 : -1;
            m2 &= m2 - 1ull;
        }
        GBits<4, GT> gb[8];
        float rs[8], gwv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int b = bq[q] >= 0 ? bq[q] : b0;                        // (fewer than eight left: the chunk's first sample again, dropped)
            const int64_t pos = (int64_t)b * wa.F + f;
            gload<true>(gb[q], g + pos * ldg + (wl ? 0 : ccol));
            rs[q] = rscale ? rscale[pos] : 1.0f;
            gwv[q] = wa.gw[(unsigned)b * wa.gws];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (bq[q] < 0) continue;
            Vf<4> x;
            gwiden(x, gb[q]);
            if (wl) { vzero(x); vset_x(x, gwv[q]); }
            if (rscale) vmul(x, rs[q]);
            vmul(x, gscale);
            if (first) { acc = x; first = false; } else vadd(acc, x);
        }
        mm = m2;
    }
    vstore<false>(cc.part + ((int64_t)chunk * kConstMax + j) * gm.D + ccol, acc);
}

// (WIDE: MREC_WPS4 waves per SIMD asked of the register allocator)
template <int VEC, class K, class Upd, class GT, bool WIDE = false, bool HOT = false>
__device__ __forceinline__ void apply_main_body(const Upd& upd, int64_t V, int64_t ld, const K* __restrict__ uniq,
                                                    const int* __restrict__ spos, const int* __restrict__ sseg,
                                                    int n, const GT* __restrict__ g, int64_t ldg,
                                                    const float* __restrict__ rscale, float gscale, ApplyGeom gm,
                                                    float* __restrict__ carry_head, float* __restrict__ carry_tail,
                                                    int* __restrict__ owners, const int* __restrict__ seg_offsets,
                                                    const WideArgs& wa, const unsigned long long cmask = 0ull, const unsigned bid0 = 0u,
                                                    const int* s_hid = nullptr) {
    constexpr int AW = ACfg<VEC>::AW, AB = ACfg<VEC>::AB, GP = ACfg<VEC>::GP;
    constexpr bool NT = ACfg<VEC>::NT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane / gm.lpr, sub = lane - grp * gm.lpr;
    if (grp >= gm.G) return;  // spare lanes; this kernel has no barriers
    // (Handing the windows out in a strided order instead -- so that the waves in flight sample the list's duplicate-heavy
    // head and its unique tail at once -- was measured: Zipf x 39 fields 0.133 -> 0.188 ms; neighbouring windows share lines.)
    const bool wl = WIDE && sub == gm.lpr - 1;          // this lane owns the wide record
    const int ccol = sub * VEC;                         // column in the carry rows
    const int col = wl ? wa.wcol : ccol;                // column in the table rows
    // A wave walks several windows (the grid is capped at MREC_APPLY_MAXB workgroups, the waves in flight at any moment cover a
    // contiguous stretch of the list): a new wave pays a chain of scalar round trips -- kernel arguments in pieces, the step
    // state, the device-side length -- before its first index word, as long as the window's own chain.
    const int64_t sw_stride = (int64_t)(gridDim.x - bid0) * 4 * gm.G;      // (bid0: workgroups in front of this pass in the launch)
    // Straddling pairs are summed in place (below) only where duplicates are rare.  The same decision in every wave of every
    // window: the number of groups is the last entry's group number + 1.
    const int n_groups = n > 0 ? sseg[n - 1] + 1 : 0;
    const bool pairs_on = MREC_PAIRS && (int64_t)(n - n_groups) * MREC_PAIRS_DUP_DIV <= (int64_t)n;
    for (int64_t sw = ((int64_t)(blockIdx.x - bid0) * 4 + wave) * gm.G + grp; sw * AW < n; sw += sw_stride) {
    const int s = (int)(sw * AW);
    const int e_end = (s + AW < n) ? s + AW : n;

    // The window's index entries all at once (one round trip for the 2 x AW + 1 words, a second one for the rows of its run
    // ends) instead of batch by batch: a window's time is the length of its chain of dependent loads -- index -> row number
    // -> row -- which this takes from three round trips per batch to one (uniform ids 0.186 -> 0.179 ms, Zipf x 39 fields
    // 0.141 -> 0.133; deeper gradient prefetch on top of it changed nothing: what remains is the AB-deep chain of row
    // read-modify-writes).
    // Every load below is REQUESTED unconditionally -- a position past the window's end reads entry n - 1 / position 0 / group 0,
    // valid addresses whose values are dropped by selects -- and converted only where it is used.  Guarded loads compile to one
    // basic block each, and a value that is converted (sign-extended, widened) inside its block is waited for inside it: the
    // eight row numbers of a window were eight dependent round trips, the two gradient rows and the state rows of a batch
    // three.  One wave's chain on an idle chip: 14.6 us before, see profiles/r03_apply_chain.txt.
    // PAIRS: a run of exactly two entries that straddles a window boundary (what a nearly duplicate-free batch has instead of
    // long runs: uniform ids) is summed by the window its first entry lies in -- entry AW, the first one behind the window, is that
    // window's ninth entry -- and skipped by the next one.  Both decide from index words alone (the run's other neighbours), so
    // they agree; the pair is summed in position order like every run inside a window, and leaves no partial sums: such a batch
    // needs no finishing pass at all.  Only where duplicates are rare (`pairs_on`, the same in every wave): on duplicate-heavy ids
    // a wave would hold one lane-group that walks a ninth entry while the other two wait.
    int posw[AW + GP], segw[AW + GP];          // (the loop below walks whole batches of GP entries: AW + 1 rounded up)
    const int nlast = n - 1;
#pragma unroll
    for (int q = 0; q < AW; ++q) {
        const int e = s + q < nlast ? s + q : nlast;
        posw[q] = spos[e];
        segw[q] = sseg[e];
    }
    segw[AW] = sseg[s + AW < nlast ? s + AW : nlast];
    posw[AW] = spos[s + AW < nlast ? s + AW : nlast];
    const int seg_before = sseg[s > 0 ? s - 1 : 0];
    const int seg_before2 = sseg[s > 1 ? s - 2 : 0];
    const int seg_after = sseg[s + AW + 1 < nlast ? s + AW + 1 : nlast];
#pragma unroll
    for (int q = AW + 1; q < AW + GP; ++q) { posw[q] = 0; segw[q] = -2; }
    // HOT (the kernel variant a stream with hot columns is captured with; the plain one is the round's earlier code, instruction for
    // instruction): entries whose id is a field's hot id -- cmask: which fields have one, s_hid: that id, the launch's first
    // workgroups sum them sample by sample -- are not this pass's.  Such a run is far longer than a window, so its entries are a
    // window's head, its tail or all of it: they become entries past the end (group -2), which every test below already treats as
    // nobody's.  The test needs the entry's id = its group's row number, which the window requests anyway: here they are requested
    // right behind the index words, with the end of the window's last run, and the window's logic follows them.
    // (The mask of such entries is used up right here -- the kernel sits at its register budget: kept across the window's body it cost
    // ten spilled registers and 45 us.  Only 32-bit keys under LazyAdam have the registers: no HOT variant of the others exists.)
    constexpr bool CONSTC = HOT;
    static_assert(!HOT || (WIDE && sizeof(K) == 4 && std::is_same<Upd, UpdAdam>::value), "HOT: the wide-folded LazyAdam apply over 32-bit keys");
    typedef typename std::conditional<sizeof(K) == 4, int, int64_t>::type RowT;
    RowT rowv[AW + GP];                                   // table row of the run that ends at entry q (-1: none / out of range)
    int last_end = 0;
#pragma unroll
    for (int q = 0; q < AW; ++q) {
        const bool valid = s + q < e_end;
        posw[q] = valid ? posw[q] : 0;
        segw[q] = valid ? segw[q] : -2;
    }
    if constexpr (HOT) {
        int last_seg0 = segw[0];
#pragma unroll
        for (int q = 1; q < AW; ++q) last_seg0 = (s + q < e_end) ? segw[q] : last_seg0;
#pragma unroll
        for (int q = 0; q < AW; ++q) rowv[q] = (RowT)seg_row<K>(uniq, segw[q] < 0 ? 0 : segw[q]);
        last_end = seg_offsets[(last_seg0 < 0 ? 0 : last_seg0) + 1];
        if (cmask) {
            bool any = false;
#pragma unroll
            for (int q = 0; q < AW; ++q) {
                const unsigned pos = (unsigned)posw[q];
                const unsigned fld = pos - (wa.F == 1 ? pos : __umulhi(pos, wa.magic)) * (unsigned)wa.F;
                const bool hot = ((unsigned)(cmask >> fld) & 1u) && (int)rowv[q] == s_hid[fld];
                const bool valid = segw[q] != -2 && !hot;
                any = any || valid;
                posw[q] = valid ? posw[q] : 0;
                segw[q] = valid ? segw[q] : -2;
            }
            if (!any) {                                  // the whole window lies inside hot runs: nothing to do, nothing carried
                if (sub == 0) owners[sw] = 0;
                continue;
            }
        }
    }
    if (!(s + AW < n)) segw[AW] = -2;
    const int first_seg = segw[0];
    const bool head_open = (s > 0) && (seg_before == first_seg);
    // the head entry is the second half of a pair the window before this one sums
    const bool head_pair = pairs_on && head_open && segw[1] != first_seg && (s < 2 || seg_before2 != first_seg);
    // the last entry is the first half of a pair whose second half lies right behind the window
    const bool tail_pair = pairs_on && s + AW < n && segw[AW] == segw[AW - 1] && segw[AW - 2] != segw[AW - 1] &&
                           (s + AW + 1 >= n || seg_after != segw[AW]);
    unsigned endm = 0u, openm = 0u;                       // bit q: entry q ends its run / belongs to the run open at the head
    // (row numbers held at the width of the ids: 32-bit ids name 32-bit rows -- registers, in the kernel that sits at its register budget)
    if constexpr (!HOT) {
#pragma unroll
        for (int q = 0; q < AW; ++q) rowv[q] = (RowT)seg_row<K>(uniq, segw[q] < 0 ? 0 : segw[q]);
    }
    rowv[AW] = tail_pair ? rowv[AW - 1] : (RowT)-1;
#pragma unroll
    for (int q = AW + 1; q < AW + GP; ++q) rowv[q] = (RowT)-1;
    // (the window's last group and the next window's first are index words it already holds; where the last run ends is
    // requested with the row numbers, not behind the window's stores: a load there waits for them -- vmcnt counts in order)
    int last_seg = segw[0];
#pragma unroll
    for (int q = 1; q < AW; ++q) last_seg = (s + q < e_end) ? segw[q] : last_seg;      // (HOT: -2 where the window ends inside a hot run)
    if constexpr (!HOT) last_end = seg_offsets[(last_seg < 0 ? 0 : last_seg) + 1];
#pragma unroll
    for (int q = 0; q < AW; ++q) {
        const bool valid = CONSTC ? segw[q] != -2 : s + q < e_end;     // (entries of constant columns are nobody's either)
        const bool is_end = valid && segw[q + 1] != segw[q];
        const bool open = head_open && segw[q] == first_seg;
        endm |= is_end ? (1u << q) : 0u;
        openm |= open ? (1u << q) : 0u;
        rowv[q] = (is_end && !open) ? rowv[q] : (RowT)-1;
    }
    endm |= tail_pair ? (1u << AW) : 0u;
    if (!tail_pair) posw[AW] = 0;
    if (head_pair) posw[0] = posw[1];                     // (skipped: its gradient row is not this window's -- request a line it fetches anyway)
    const int e_end2 = e_end + (tail_pair ? 1 : 0);       // the entries this window walks
    asm volatile("" : "+v"(last_end));      // landed with the row numbers (needed here: see vtouch)
    // (CHUNKS, round 5, built, parity-green at chunk widths 2 / 3 / 8 and measured: a lane-group walks W consecutive windows and
    // keeps a run's sum in its registers from one window into the next, so runs leave partial sums only where they cross a
    // chunk of W x 8 entries.  Zipf ids x 39 fields, one box: the finishing pass 44 -> 23-24 us at every W >= 2 (its floor inside
    // the dense Adam's launch), but this kernel 97 -> 104 us already at W = 1 -- the loop-carried sum and the chunk's state cost
    // 5-8 spilled registers at the 128 budget -- and 113 / 116 / 125 / 135 us at W = 2 / 3 / 4 / 6: neighbouring lane-groups no
    // longer share index lines and a slot's W windows are no longer balanced by the dispatcher.  Uniform ids 176.5 -> 179.5 (W = 1)
    // -> 182-201 us.  The step did not move (0.717-0.726 ms at every W): the finishing pass already hides behind the dense Adam.)
    // (A two-round-trip path for windows that own no run end -- all eight gradient rows at once, no row numbers; half of the
    // windows on Zipf ids x 39 fields -- was built and measured: 94.6-95.5 us against 95.6-96.1, not kept.)
    Vf<VEC> acc;
    vzero(acc);
#pragma unroll
    for (int jb = 0; jb < AW + GP; jb += GP) {
        if (s + jb >= e_end2) break;
        GBits<VEC, GT> gb[GP];
        float rsv[GP], gwq[GP];
#pragma unroll
        for (int q = 0; q < GP; ++q) {
            const int pos = posw[jb + q];
            gwq[q] = 0.0f;
            if (WIDE) gwq[q] = wa.gw[(wa.F == 1 ? (unsigned)pos : __umulhi((unsigned)pos, wa.magic)) * wa.gws];
            gload<NT>(gb[q], g + (int64_t)pos * ldg + (WIDE && wl ? 0 : col));       // (the wide lane's own gradient is gwq)
            rsv[q] = rscale ? rscale[pos] : 1.0f;
        }
#pragma unroll
        for (int sb = 0; sb < GP / AB; ++sb) {
            const int j0 = jb + sb * AB;
            if (s + j0 >= e_end2) break;
            bool upd_ok[AB];
            int64_t roff[AB];
            Vf<VEC> st[AB][Upd::NS];
#pragma unroll
            for (int k = 0; k < AB; ++k) {
                const int64_t row = (int64_t)rowv[j0 + k];
                upd_ok[k] = row >= 0 && row < V;
                roff[k] = upd_ok[k] ? row * ld + col : 0;
                if (upd_ok[k] && Upd::kLoad) {
#pragma unroll
                    for (int i = 0; i < Upd::NS; ++i)
                        if (!(WIDE && wl && i > 0)) vload<NT>(st[k][i], upd.s[i] + roff[k]);
                }
            }
            if (GP > AB && sb == 0) {
#pragma unroll
                for (int q = 0; q < GP; ++q) gtouch(gb[q]);
            }
            // the batch's ONE wait, in straight-line code and before any store: the gradients are needed here, and behind them
            // (the guarded state-row loads may or may not have been issued, so the wait is for everything) the rows have landed
            Vf<VEC> xs[AB];
#pragma unroll
            for (int k = 0; k < AB; ++k) {
                gwiden(xs[k], gb[sb * AB + k]);
                if (WIDE && wl) { vzero(xs[k]); vset_x(xs[k], gwq[sb * AB + k]); }
                if (rscale) vmul(xs[k], rsv[sb * AB + k]);
                vmul(xs[k], gscale);
                vtouch(xs[k]);
            }
#pragma unroll
            for (int k = 0; k < AB; ++k) {
                const int q = j0 + k;
                if (s + q >= e_end2 || (CONSTC && segw[q] == -2)) continue;
                Vf<VEC> x = xs[k];
                const bool is_start = q == 0 || segw[q] != segw[q - 1];
                if (is_start) acc = x; else vadd(acc, x);
                if ((endm >> q) & 1u) {
                    if ((openm >> q) & 1u) {
                        if (!head_pair) vstore<false>(carry_head + sw * gm.D + ccol, acc);
                    } else if (upd_ok[k]) {
                        if (WIDE && wl) wide_apply(st[k][0], acc, wa.h);
                        else upd_apply<Upd>(upd, st[k], acc);
#pragma unroll
                        for (int i = 0; i < Upd::NS; ++i)
                            if (!(WIDE && wl && i > 0)) vstore<NT>(upd.s[i] + roff[k], st[k][i]);
                    }
                }
            }
        }
    }
    // run continues past this window?  owners[sw]: 0 = no run of this window continues, 1 = this window owns
    // a run with few partials (finished by one lane-group of k_apply_long), 2 = a long run (a block's job).
    // Every window writes its flag, so the list needs neither clearing nor an atomic counter.
    int flag = 0;
    if (tail_pair) {
        // (the pair was summed and applied above: nothing is carried)
    } else if (e_end < n && segw[AW] == last_seg) {
        if (head_open && last_seg == first_seg) {
            vstore<false>(carry_head + sw * gm.D + ccol, acc);  // window lies wholly inside one run
        } else {
            vstore<false>(carry_tail + sw * gm.D + ccol, acc);
            const int k = (last_end - 1) / AW - (int)sw;   // head partials that follow
            flag = (k + 1 <= 4 * gm.G) ? 1 : 2;
        }
    }
    if (sub == 0) owners[sw] = flag;
    }
}

// Step scalars from device memory (ss != nullptr): the Adam step size of this step, and the kernel's own begin / end wall
// clock stamps (events recorded inside a captured graph cannot be timed on this stack; these can).
template <class Upd>
__device__ __forceinline__ void resolve_step(Upd&, const StepState*) {}
__device__ __forceinline__ void resolve_step(UpdAdam& u, const StepState* ss) { if (ss) u.h.lr_t = ss->lr_t; }

template <int VEC, class K, class Upd, class GT, bool WIDE = false, bool HOT = false>
__global__ __launch_bounds__(256, WIDE ? MREC_WPS4 : 1) void k_apply_main(Upd upd, int64_t V, int64_t ld, const K* __restrict__ uniq,
                                                    const int* __restrict__ spos, const int* __restrict__ sseg,
                                                    int n, const GT* __restrict__ g, int64_t ldg,
                                                    const float* __restrict__ rscale, float gscale, ApplyGeom gm,
                                                    float* __restrict__ carry_head, float* __restrict__ carry_tail,
                                                    int* __restrict__ owners, const int* __restrict__ seg_offsets,
                                                    WideArgs wa, StepState* ss, const int64_t* __restrict__ nv = nullptr,
                                                    const ConstCols cc = ConstCols{}) {
    // (Fetching all kernel arguments in one scalar round trip at the top -- left alone the compiler fetches them in pieces,
    // each in front of its first use -- was measured: uniform ids 172 -> 174 us, Zipf x 39 fields 96 -> 101; not kept.)
    resolve_step(upd, ss);
    if (nv) { const int64_t x = *nv; if (x < n) n = x < 0 ? 0 : (int)x; }      // entries of the index proper: known on the device only
    unsigned long long cmask = 0ull;
    __shared__ int s_hid[HOT ? 64 : 1];
    if constexpr (HOT) {
        cmask = const_mask(cc.mask);
        if (cmask) {                                   // (uniform over the launch)
            if (threadIdx.x < 64) s_hid[threadIdx.x] = (int)cc.hid[threadIdx.x];
            __syncthreads();
        }
        if (blockIdx.x < cc.cblocks) {                 // the hot columns' chunk sums: dispatched first, a chain of ~10 round trips
            if (cmask) const_part_body<K, GT>(cc, cmask, g, ldg, rscale, gscale, wa, gm);
            return;
        }
    }
    // Stamps: workgroup 0 (dispatched first) stores the begin; the last wave of every workgroup raises the end -- ONE global
    // atomic per workgroup (an atomic per wave on the one address serialised at ~6 ns each and made the kernel 140 us longer).
    __shared__ int waves_done;
    unsigned long long* stamp = ((MREC_STAMPS & 1) && ss && !ss->stamps_off) ? ss->stamps[(unsigned)ss->step % kStampRing] : nullptr;
    // (the end: only the LAST-dispatched 1024 workgroups -- the last round of residency at 4 per CU -- read the clock; every
    // workgroup doing it lengthened each of the four rounds by its realtime read: 4-6 us per step, profiles/r05_stamps_ab.txt)
    if (stamp && blockIdx.x != cc.cblocks && (int)blockIdx.x + 1024 < (int)gridDim.x) stamp = nullptr;
    if (stamp) {
        if (threadIdx.x == 0) {
            waves_done = 0;
            if (blockIdx.x == cc.cblocks) stamp[0] = (unsigned long long)wall_clock64();
        }
        __syncthreads();
    }
    apply_main_body<VEC, K, Upd, GT, WIDE, HOT>(upd, V, ld, uniq, spos, sseg, n, g, ldg, rscale, gscale, gm, carry_head, carry_tail, owners,
                                                seg_offsets, wa, cmask, HOT ? cc.cblocks : 0u, s_hid);
    if (stamp && (int)blockIdx.x + 1024 >= (int)gridDim.x && (threadIdx.x & 63) == 0 && atomicAdd(&waves_done, 1) == 3)
        ss->ends[(unsigned)ss->step % kStampRing][blockIdx.x & 63u] = (unsigned long long)wall_clock64();
}


// Finishes the runs that cross windows.  Partial 0 is the owner's tail, partials 1..k the heads of the
// following k windows.
//   pass A: runs with at most NG = 4*G partials -- one lane-group each, partials added in order, no barriers
//           (with heavy duplication, e.g. 640 K ids over a 200 K-row table, most windows own such a run);
//   pass B: longer runs -- one block each: lane-groups take partials round-robin (fixed assignment), then
//           group 0 adds the per-group sums in group order.
// For k + 1 <= NG both orders coincide (each group holds one partial), so the split does not change results.
// (bid, nblocks): the workgroup's number among those of this pass and their count -- the kernel below passes its own grid; the
// launch that carries this pass in front of the dense Adam (k_finish_dense_adam) passes the pass's share of its grid)
template <int VEC, class K, class Upd, bool WIDE = false>
__device__ __forceinline__ void apply_long_body(Upd upd, int64_t V, int64_t ld, const K* __restrict__ uniq,
                                                const int* __restrict__ sseg,
                                                const int* __restrict__ seg_offsets, int n, ApplyGeom gm,
                                                const float* __restrict__ carry_head,
                                                const float* __restrict__ carry_tail,
                                                const int* __restrict__ owners, int nsw, const WideArgs& wa,
                                                const StepState* ss, const int64_t* __restrict__ nv, int bid, int nblocks) {
    resolve_step(upd, ss);
    // the partials of a run are consecutive carry rows: stream them 16 deep per lane-group
    constexpr int AW = ACfg<VEC>::AW, AB = MREC_LONG_AB;
    if (nv) {
        const int64_t x = *nv;
        if (x < n) { n = x < 0 ? 0 : (int)x; nsw = (n + AW - 1) / AW; }
    }
    __shared__ float red[256 * 4];
    __shared__ int list[256];
    __shared__ int nlist;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane / gm.lpr, sub = lane - grp * gm.lpr;
    const bool active = grp < gm.G;
    const bool wl = WIDE && sub == gm.lpr - 1;
    const int col = sub * VEC;                          // column in the carry rows
    const int tcol = wl ? wa.wcol : col;                // column in the table rows
    const int NG = 4 * gm.G;
    const int gi = wave * gm.G + grp;

    // ---- this block's windows: 4 per lane-group, their flags read in ONE coalesced load and sorted into two lists in LDS (with
    // flags read window by window inside the passes, every iteration was a memory round trip to find, mostly, a zero)
    __shared__ int list_a[256], n_a;
    const int WPB = 4 * NG < 256 ? 4 * NG : 256;      // (the same expression sizes the grid in apply_cols; 16 * NG -- a quarter of the
                                                      // workgroups in front of the dense Adam's -- was measured: uniform ids 36.4 -> 34.4 us for the fused
                                                      // launch, but Zipf ids x 39 fields 46 us for the pass itself: a block walks its flagged windows serially)
    const int64_t w0 = (int64_t)bid * WPB;
    if (threadIdx.x == 0) { n_a = 0; nlist = 0; }
    __syncthreads();
    if ((int)threadIdx.x < WPB && w0 + threadIdx.x < nsw) {
        const int f = owners[w0 + threadIdx.x];
        if (f == 1) list_a[atomicAdd(&n_a, 1)] = (int)(w0 + threadIdx.x);
        else if (f == 2) list[atomicAdd(&nlist, 1)] = (int)(w0 + threadIdx.x);
    }
    __syncthreads();
    const int cnt_a = n_a, cnt = nlist;

    // ---- pass A.  A run's chain of dependent loads is kept short: its length and its row number are requested together (both
    // need only the group number), the state rows as soon as the row number is there, and the partials four at a time at
    // clamped addresses (added in order, the extra ones dropped) -- one partial per round trip before.
    if (active) {
        for (int ia = gi; ia < cnt_a; ia += NG) {
            const int sw = list_a[ia];
            const int last_e = ((sw + 1) * AW < n ? (sw + 1) * AW : n) - 1;
            const int u = sseg[last_e];
            const int seg_end = seg_offsets[u + 1];
            const int64_t row = seg_row<K>(uniq, u);
            const int k = (seg_end - 1) / AW - sw;
            Vf<VEC> acc;
            vload<false>(acc, carry_tail + (int64_t)sw * gm.D + col);
            const bool ok = row >= 0 && row < V;
            const int64_t roff = (ok ? row : 0) * ld + tcol;          // (row 0 for a row out of range: read, never written)
            Vf<VEC> st[Upd::NS];
            if (Upd::kLoad) {
#pragma unroll
                for (int i = 0; i < Upd::NS; ++i)
                    if (!(WIDE && wl && i > 0)) vload<false>(st[i], upd.s[i] + roff);
            }
            for (int t0 = 1; t0 <= k; t0 += 4) {
                Vf<VEC> x[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int t = t0 + q <= k ? t0 + q : k;
                    vload<false>(x[q], carry_head + (int64_t)(sw + t) * gm.D + col);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (t0 + q <= k) vadd(acc, x[q]);
            }
            if (ok) {
                if (WIDE && wl) {
                    wide_apply(st[0], acc, wa.h);
                    vstore<false>(upd.s[0] + roff, st[0]);
                } else {
                    upd_apply<Upd>(upd, st, acc);
#pragma unroll
                    for (int i = 0; i < Upd::NS; ++i) vstore<false>(upd.s[i] + roff, st[i]);
                }
            }
        }
    }

    // ---- pass B: the long runs among this block's windows, one after the other, the whole block on each
    // (Round 5, built, parity-green, measured and not kept: HOT runs -- Criteo's 13 constant dense-field ids, 2048 partials each --
    // finished COOPERATIVELY: k_apply_main listed their owner windows (a word whose high half carried the step number, so nothing
    // needed clearing), every workgroup of this pass summed chunks of 128 partials round-robin over the list, chunk sums travelled
    // write-through (sc1 stores; with agent-scope release fences instead the pass took 80 us: each fence writes back what the
    // dense Adam beside it has dirtied in the XCD's L2), the workgroup whose ticket was a run's last added the chunk sums in
    // chunk order and updated the row.  Zipf ids x 39 fields, same box, two A/B pairs: this pass 35.0-36.9 us against 29.6-30.8
    // as it is, k_apply_main 97.6-98.4 against 98.2-98.4, the step 0.7366 against 0.7321-0.7458 ms -- the hot runs are not what
    // the pass waits for; the thousands of runs of 9-1000 entries are, one lane-group or one workgroup each.
    // Nor is it their placement or the length of a run's chain: 12 or 24 windows per workgroup instead of 48 (the flagged windows
    // of Zipf ids sit at the head of the list), and a per-window record {row, partials} written by k_apply_main that takes three
    // dependent round trips off every run here, both left the pass at 42-47 us.  As a launch of its own the same pass took 13 us
    // (round 3): inside this launch every round trip waits in the memory system behind the dense Adam's 4.7 TB/s of streaming.)
    {
        for (int o = 0; o < cnt; ++o) {
            const int sw = list[o];
            const int last_e = ((sw + 1) * AW < n ? (sw + 1) * AW : n) - 1;
            const int u = sseg[last_e];
            const int seg_end = seg_offsets[u + 1];
            const int k = (seg_end - 1) / AW - sw;  // number of head partials (>= 1)
            Vf<VEC> acc;
            vzero(acc);
            bool has = false;
            if (active) {
                for (int t0 = gi; t0 <= k; t0 += NG * AB) {
                    Vf<VEC> x[AB];
#pragma unroll
                    for (int q = 0; q < AB; ++q) {
                        const int t = t0 + q * NG;
                        vzero(x[q]);
                        if (t <= k) {
                            const float* src = (t == 0) ? carry_tail + (int64_t)sw * gm.D : carry_head + (int64_t)(sw + t) * gm.D;
                            vload<false>(x[q], src + col);
                        }
                    }
#pragma unroll
                    for (int q = 0; q < AB; ++q) {
                        if (t0 + q * NG <= k) {
                            if (has) vadd(acc, x[q]); else { acc = x[q]; has = true; }
                        }
                    }
                }
                vstore<false>(red + gi * gm.D + col, acc);
            }
            __syncthreads();
            if (active && gi == 0) {
                const int ng = (k + 1 < NG) ? k + 1 : NG;
                for (int q = 1; q < ng; ++q) {
                    Vf<VEC> x;
                    vload<false>(x, red + q * gm.D + col);
                    vadd(acc, x);
                }
                const int64_t row = seg_row<K>(uniq, u);
                if (row >= 0 && row < V) {
                    const int64_t roff = row * ld + tcol;
                    Vf<VEC> st[Upd::NS];
                    if (WIDE && wl) {
                        vload<false>(st[0], upd.s[0] + roff);
                        wide_apply(st[0], acc, wa.h);
                        vstore<false>(upd.s[0] + roff, st[0]);
                    } else {
                        if (Upd::kLoad) {
#pragma unroll
                            for (int i = 0; i < Upd::NS; ++i) vload<false>(st[i], upd.s[i] + roff);
                        }
                        upd_apply<Upd>(upd, st, acc);
#pragma unroll
                        for (int i = 0; i < Upd::NS; ++i) vstore<false>(upd.s[i] + roff, st[i]);
                    }
                }
            }
            __syncthreads();
        }
    }
    // end stamp (measurement): the workgroups that had a run to finish raise it -- one atomic each.  A step whose batch left no
    // partial sums behind (uniform ids since round 5: straddling pairs are summed by k_apply_main) has no finishing work and
    // no finishing stamp: its apply ends with k_apply_main's own end stamp.
    if ((MREC_STAMPS & 4) && ss && !ss->stamps_off && threadIdx.x == 0 && cnt_a + cnt > 0)
        atomicMax((unsigned long long*)&ss->aux[(unsigned)ss->step % kStampRing][2], (unsigned long long)wall_clock64());
}

template <int VEC, class K, class Upd, bool WIDE = false>
__global__ __launch_bounds__(256) void k_apply_long(Upd upd, int64_t V, int64_t ld, const K* __restrict__ uniq,
                                                    const int* __restrict__ sseg,
                                                    const int* __restrict__ seg_offsets, int n, ApplyGeom gm,
                                                    const float* __restrict__ carry_head,
                                                    const float* __restrict__ carry_tail,
                                                    const int* __restrict__ owners, int nsw, WideArgs wa,
                                                    const StepState* ss, const int64_t* __restrict__ nv = nullptr) {
    apply_long_body<VEC, K, Upd, WIDE>(upd, V, ld, uniq, sseg, seg_offsets, n, gm, carry_head, carry_tail, owners, nsw, wa, ss, nv,
                                       (int)blockIdx.x, (int)gridDim.x);
}

// The finishing pass of the wide-folded apply (k_apply_long<4, K, UpdAdam, true>) and the dense net's Adam in ONE launch: the
// first `lblocks` workgroups finish the runs that cross windows, the rest run the dense Adam.  The two are independent -- one
// touches embedding rows, the other the dense net's flat buffers -- and the finishing pass is a chain of four or five
// dependent round trips with almost nothing to move (uniform ids: ~450 runs; 9-10 us as a launch of its own behind
// k_apply_main, 13 us on Zipf ids x 39 fields), which hides entirely inside the 32-us HBM-bound Adam pass beside it.  (As a graph
// side branch instead the cross-branch join costs more than the overlap returns)
// The constant columns' rows: workgroup j of this pass adds the lane-groups' partial sums of column j in lane-group order (twelve
// lane-groups take a twelfth each, eight rows in flight, then their sums in order) and updates the column's one row.
template <class K>
__device__ __forceinline__ void const_finish_body(UpdAdam upd, int64_t V, int64_t ld, ApplyGeom gm, const WideArgs& wa, const ConstCols& cc,
                                                  const StepState* ss, int j) {
    resolve_step(upd, ss);
    const unsigned long long cmask = const_mask(cc.mask);
    __shared__ float cred[1024];          // 4 G lane-groups x 4 lpr floats, G * lpr <= 64
    int f = -1;
    {
        unsigned long long m = cmask;
        for (int q = 0; q <= j && m; ++q) { if (q == j) f = (int)__ffsll((long long)m) - 1; m &= m - 1ull; }
    }
    if (f < 0) return;                                   // fewer constant columns than j + 1 (uniform over the workgroup)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane / gm.lpr, sub = lane - grp * gm.lpr;
    const bool active = grp < gm.G;
    const bool wl = sub == gm.lpr - 1;
    const int col = sub * 4, tcol = wl ? wa.wcol : col;
    const int NG = 4 * gm.G, gi = wave * gm.G + grp;
    const int per = (cc.nlg + NG - 1) / NG;
    // (the row's number and its state are requested BEFORE the chunk sums, which do not depend on them: two round trips less in a
    // pass that is nothing but its chain -- inside the dense Adam's launch a round trip is 3-4 us)
    const int64_t row = (int64_t)cc.hid[f];
    const bool rok = row >= 0 && row < V;
    const int64_t roff = (rok ? row : 0) * ld + tcol;
    Vf<4> st[3];
    if (active && gi == 0) {
        if (wl) {
            vload<false>(st[0], upd.s[0] + roff);
        } else {
#pragma unroll
            for (int i = 0; i < 3; ++i) vload<false>(st[i], upd.s[i] + roff);
        }
    }
    Vf<4> acc;
    vzero(acc);
    if (active) {
        const int l0 = gi * per, l1 = (l0 + per < cc.nlg) ? l0 + per : cc.nlg;
        for (int t0 = l0; t0 < l1; t0 += 11) {
            Vf<4> x[11];
#pragma unroll
            for (int q = 0; q < 11; ++q) {
                const int t = t0 + q < l1 ? t0 + q : l1 - 1;
                vload<false>(x[q], cc.part + ((int64_t)t * kConstMax + j) * gm.D + col);
            }
#pragma unroll
            for (int q = 0; q < 11; ++q)
                if (t0 + q < l1) { if (t0 + q == l0) acc = x[q]; else vadd(acc, x[q]); }
        }
        vstore<false>(cred + gi * gm.D + col, acc);
    }
    __syncthreads();
    if (active && gi == 0) {
        for (int q = 1; q < NG; ++q) {
            if (q * per >= cc.nlg) break;
            Vf<4> x;
            vload<false>(x, cred + q * gm.D + col);
            vadd(acc, x);
        }
        if (rok) {
            if (wl) {
                wide_apply(st[0], acc, wa.h);
                vstore<false>(upd.s[0] + roff, st[0]);
            } else {
                upd_apply<UpdAdam>(upd, st, acc);
#pragma unroll
                for (int i = 0; i < 3; ++i) vstore<false>(upd.s[i] + roff, st[i]);
            }
        }
    }
    if ((MREC_STAMPS & 4) && ss && !ss->stamps_off && threadIdx.x == 0)
        atomicMax((unsigned long long*)&ss->aux[(unsigned)ss->step % kStampRing][2], (unsigned long long)wall_clock64());
}

template <class K>
__global__ __launch_bounds__(256) void k_apply_const_finish(UpdAdam upd, int64_t V, int64_t ld, ApplyGeom gm, WideArgs wa, ConstCols cc,
                                                            const StepState* ss) {
    const_finish_body<K>(upd, V, ld, gm, wa, cc, ss, (int)blockIdx.x);
}

struct ApplyFinish {
    UpdAdam upd; int64_t V, ld; const void* uniq; int key_bytes; const int* sseg; const int* seg_offsets; int n; ApplyGeom gm;
    const float* carry_head; const float* carry_tail; const int* owners; int nsw; WideArgs wa; const StepState* ss;
    const int64_t* nv; unsigned lblocks; unsigned magic;
    ConstCols cc; unsigned cfin;            // cfin: workgroups of the constant columns' finishing pass (kConstMax, or 0: no such pass)
};
static_assert(sizeof(ApplyFinish) <= sizeof(mrec_apply_finish_t), "mrec_apply_finish_t too small");
constexpr unsigned kFinishMagic = 0x4D524543u;

template <class K, int SHK>
__global__ __launch_bounds__(256) void k_finish_dense_adam(ApplyFinish f, DenseAdamSlabArgs a, SlabSegs sg) {
    if (blockIdx.x < f.lblocks) {
        apply_long_body<4, K, UpdAdam, true>(f.upd, f.V, f.ld, (const K*)f.uniq, f.sseg, f.seg_offsets, f.n, f.gm, f.carry_head,
                                             f.carry_tail, f.owners, f.nsw, f.wa, f.ss, f.nv, (int)blockIdx.x, (int)f.lblocks);
        return;
    }
    if (blockIdx.x < f.lblocks + f.cfin) {
        const_finish_body<K>(f.upd, f.V, f.ld, f.gm, f.wa, f.cc, f.ss, (int)(blockIdx.x - f.lblocks));
        return;
    }
    const unsigned front = f.lblocks + f.cfin;
    dense_adam4_slabs_body<SHK>(a.p, a.m, a.v, a.g, a.n4, a.h, a.shadow, sg, a.ss, a.f1,
                                (int64_t)(blockIdx.x - front) * 256 + threadIdx.x, (int64_t)(gridDim.x - front) * 256);
}

inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

thread_local hipEvent_t t_prof_start = nullptr, t_prof_stop = nullptr;
thread_local ApplyFinish* t_defer = nullptr;        // set by mrec_sparse_lazy_adam_wide_defer: the finishing pass is handed back, not launched
thread_local ConstCols t_const = ConstCols{};       // set by mrec_sparse_apply_next_const_cols: the next wide apply takes constant columns out of its windows

struct ApplyWs { float* carry_head; float* carry_tail; int* owners; int* n_owners; float* dummy; float* cpart; };

size_t apply_ws_bytes(int64_t n, int32_t D) {
    const size_t nsw = (size_t)mrec_cdiv(n ? n : 1, AW_MIN);
    const int Dc = D > 256 ? 256 : D;
    const size_t nch = (size_t)mrec_cdiv(n ? n : 1, 64) < (size_t)kConstLG ? (size_t)mrec_cdiv(n ? n : 1, 64) : (size_t)kConstLG;
    return mrec_align_up(nsw * Dc * 4, 256) * 2 + mrec_align_up(nsw * 4, 256) + 256 + 2048 * 64 +
           mrec_align_up(nch * kConstMax * Dc * 4, 256);
}

// One launch pair over columns [c0, c0+Dc) of every array.
template <class K, class Upd, class GT>
int apply_cols(Upd upd, int64_t V, int64_t ld, const K* uniq, const int* spos, const int* sseg,
               const int* seg_offsets, int64_t n, const GT* g, int64_t ldg, const float* rscale, float gscale,
               int Dc, int vec, const ApplyWs& w, hipStream_t st, const WideArgs* wide = nullptr, StepState* ss = nullptr,
               const int64_t* nv = nullptr) {
    ApplyGeom gm;
    gm.D = Dc + (wide ? 4 : 0);
    gm.lpr = Dc / vec + (wide ? 1 : 0);
    gm.G = 64 / gm.lpr;
    WideArgs wa{};
    if (wide) {
        wa = *wide;
        wa.magic = (unsigned)(((uint64_t)1 << 32) / (uint64_t)wa.F + 1);       // pos / F = umulhi(pos, magic) while pos * F < 2^32
        wa.dummy = w.dummy;
        if ((uint64_t)n * (uint64_t)wa.F >= ((uint64_t)1 << 32) || (uint64_t)n * (uint64_t)wa.gws >= ((uint64_t)1 << 32)) return MREC_EUNSUPPORTED;
    }
    const int64_t nsw = mrec_cdiv(n, vec == 4 ? ACfg<4>::AW : ACfg<1>::AW);
    unsigned blocks = (unsigned)mrec_cdiv(nsw, (int64_t)4 * gm.G);
    if (blocks > MREC_APPLY_MAXB) blocks = MREC_APPLY_MAXB;
    const unsigned lblocks = (unsigned)mrec_cdiv(nsw, (int64_t)(16 * gm.G < 256 ? 16 * gm.G : 256));      // k_apply_long: 4 windows per lane-group, 4 G lane-groups, 256 at most
    const hipEvent_t ev0 = t_prof_start, ev1 = t_prof_stop;
    t_prof_start = t_prof_stop = nullptr;
    if (ev0) MREC_HIP_CHECK(hipEventRecord(ev0, st));
    if (nv && !(vec == 4 && wide)) return MREC_EUNSUPPORTED;
    // constant columns (armed for this call by mrec_sparse_apply_next_const_cols): the batch is B = n / F samples of F fields, ids of
    // the table's key width, LazyAdam on the fused rows -- anything else runs every column through the windows
    ConstCols cc = t_const;
    t_const = ConstCols{};
    if (!(vec == 4 && wide && std::is_same<Upd, UpdAdam>::value && sizeof(K) == 4 && !nv && cc.mask && cc.ids0 && wa.F >= 1 && wa.F <= 64 &&
          n % wa.F == 0 && cc.B == (int)(n / wa.F) && cc.B > 0 && cc.B <= 64 * kConstLG && wa.gws == 1 && cc.id_bytes == (int)sizeof(K)))
        cc = ConstCols{};
    if (cc.mask) {
        cc.part = w.cpart;
        cc.rr = 64;                                                                                  // (a chunk's samples are a 64-bit mask)
        cc.nlg = (int)mrec_cdiv((int64_t)cc.B, (int64_t)cc.rr);                                      // chunks
        cc.cblocks = (unsigned)mrec_cdiv((int64_t)cc.nlg * kConstMax, (int64_t)4 * gm.G);      // (room for kConstMax columns: how many there are is known on the device)
    }
    if (vec == 4 && wide) {
      if constexpr (!std::is_same<Upd, UpdAdam>::value) {
        return MREC_EUNSUPPORTED;                  // (the wide lane rides LazyAdam only: no such instantiation of the other updaters)
      } else {
        if constexpr (sizeof(K) == 4) {
            if (cc.mask)
                k_apply_main<4, K, Upd, GT, true, true><<<blocks + cc.cblocks, 256, 0, st>>>(upd, V, ld, uniq, spos, sseg, (int)n, g, ldg, rscale,
                                                             gscale, gm, w.carry_head, w.carry_tail, w.owners, seg_offsets, wa, ss, nv, cc);
        }
        if (!cc.mask)
            k_apply_main<4, K, Upd, GT, true><<<blocks, 256, 0, st>>>(upd, V, ld, uniq, spos, sseg, (int)n, g, ldg, rscale, gscale, gm,
                                                             w.carry_head, w.carry_tail, w.owners, seg_offsets, wa, ss, nv, cc);
        if (ev1) MREC_HIP_CHECK(hipEventRecord(ev1, st));
        if (t_defer) {
            if constexpr (std::is_same<Upd, UpdAdam>::value) {
                ApplyFinish* f = t_defer;
                f->upd = upd; f->V = V; f->ld = ld; f->uniq = uniq; f->key_bytes = (int)sizeof(K); f->sseg = sseg; f->seg_offsets = seg_offsets;
                f->n = (int)n; f->gm = gm; f->carry_head = w.carry_head; f->carry_tail = w.carry_tail; f->owners = w.owners; f->nsw = (int)nsw;
                f->wa = wa; f->ss = ss; f->nv = nv; f->lblocks = lblocks; f->magic = kFinishMagic;
                f->cc = cc; f->cfin = cc.mask ? (unsigned)kConstMax : 0u;
            } else {
                return MREC_EUNSUPPORTED;
            }
        } else {
            k_apply_long<4, K, Upd, true><<<lblocks, 256, 0, st>>>(upd, V, ld, uniq, sseg, seg_offsets, (int)n, gm,
                                                                  w.carry_head, w.carry_tail, w.owners, (int)nsw, wa, ss, nv);
            if constexpr (std::is_same<Upd, UpdAdam>::value) {
                if (cc.mask) k_apply_const_finish<K><<<kConstMax, 256, 0, st>>>(upd, V, ld, gm, wa, cc, ss);
            }
        }
      }
    } else if (vec == 4) {
        k_apply_main<4, K, Upd, GT><<<blocks, 256, 0, st>>>(upd, V, ld, uniq, spos, sseg, (int)n, g, ldg, rscale, gscale, gm,
                                                       w.carry_head, w.carry_tail, w.owners, seg_offsets, wa, ss);
        if (ev1) MREC_HIP_CHECK(hipEventRecord(ev1, st));
        k_apply_long<4, K, Upd><<<lblocks, 256, 0, st>>>(upd, V, ld, uniq, sseg, seg_offsets, (int)n, gm,
                                                        w.carry_head, w.carry_tail, w.owners, (int)nsw, wa, ss);
    } else if (vec == 2) {
        k_apply_main<2, K, Upd, GT><<<blocks, 256, 0, st>>>(upd, V, ld, uniq, spos, sseg, (int)n, g, ldg, rscale, gscale, gm,
                                                       w.carry_head, w.carry_tail, w.owners, seg_offsets, wa, ss);
        if (ev1) MREC_HIP_CHECK(hipEventRecord(ev1, st));
        k_apply_long<2, K, Upd><<<lblocks, 256, 0, st>>>(upd, V, ld, uniq, sseg, seg_offsets, (int)n, gm,
                                                        w.carry_head, w.carry_tail, w.owners, (int)nsw, wa, ss);
    } else {
        k_apply_main<1, K, Upd, GT><<<blocks, 256, 0, st>>>(upd, V, ld, uniq, spos, sseg, (int)n, g, ldg, rscale, gscale, gm,
                                                       w.carry_head, w.carry_tail, w.owners, seg_offsets, wa, ss);
        if (ev1) MREC_HIP_CHECK(hipEventRecord(ev1, st));
        k_apply_long<1, K, Upd><<<lblocks, 256, 0, st>>>(upd, V, ld, uniq, sseg, seg_offsets, (int)n, gm,
                                                        w.carry_head, w.carry_tail, w.owners, (int)nsw, wa, ss);
    }
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}

template <class K, class Upd, class GT = float>
int apply_impl(Upd upd, int64_t V, int64_t ld, int32_t D, const K* uniq, const int32_t* spos, const int32_t* sseg,
               const int32_t* seg_offsets, int64_t n, const GT* g, int64_t ldg, const float* rscale,
               float gscale, void* ws, size_t ws_bytes, void* stream, const WideArgs* wide = nullptr, StepState* ss = nullptr,
               const int64_t* nv = nullptr) {
    hipStream_t st = (hipStream_t)stream;
    if (n < 0 || D <= 0 || V < 0 || ld < D || ldg < D) return MREC_EINVAL;
    if (n == 0) return MREC_OK;
    if (V == 0) return MREC_EINVAL;      // rows are read unconditionally at clamped addresses: an empty table has no valid one
    if (!spos || !sseg || !seg_offsets || !g || !ws) return MREC_EINVAL;
    for (int i = 0; i < Upd::NS; ++i) if (!upd.s[i]) return MREC_EINVAL;
    if (n > (int64_t(1) << 30)) return MREC_EUNSUPPORTED;
    if (ws_bytes < apply_ws_bytes(n, D + (wide ? 4 : 0))) return MREC_EWORKSPACE;
    const size_t nsw = (size_t)mrec_cdiv(n, AW_MIN);
    const int Dc_max = (D > 256 ? 256 : D) + (wide ? 4 : 0);
    MrecArena a(ws, ws_bytes);
    ApplyWs w;
    w.carry_head = a.take<float>(nsw * Dc_max);
    w.carry_tail = a.take<float>(nsw * Dc_max);
    w.owners = a.take<int>(nsw);
    w.n_owners = a.take<int>(1);
    w.dummy = a.take<float>(2048 * 16);
    const size_t nch = (size_t)mrec_cdiv(n, 64) < (size_t)kConstLG ? (size_t)mrec_cdiv(n, 64) : (size_t)kConstLG;
    w.cpart = a.take<float>(nch * kConstMax * Dc_max);
    if (!a.ok) return MREC_EWORKSPACE;
    bool aligned = (ld % 4 == 0) && (ldg % 4 == 0) && ((((uintptr_t)g) & (4 * sizeof(GT) - 1)) == 0);
    for (int i = 0; i < Upd::NS; ++i) aligned = aligned && al16(upd.s[i]);
    bool aligned8 = (ld % 2 == 0) && (ldg % 2 == 0) && ((((uintptr_t)g) & (2 * sizeof(GT) - 1)) == 0);
    for (int i = 0; i < Upd::NS; ++i) aligned8 = aligned8 && ((((uintptr_t)upd.s[i]) & 7) == 0);
    // 16-byte lanes when rows allow it, else 8-byte lanes (D = 30 of the DCN table), else scalar
    const int vec = (aligned && D % 4 == 0) ? 4 : ((aligned8 && D % 2 == 0) ? 2 : 1);
    if (wide && (vec != 4 || D > 252 || wide->wcol != D || D + 4 > ld || wide->F <= 0 || !wide->gw)) return MREC_EUNSUPPORTED;
    const int CB = 64 * vec;  // columns per launch
    for (int c0 = 0; c0 < D; c0 += CB) {
        const int Dc = (D - c0 < CB) ? D - c0 : CB;
        Upd u2 = upd;
        for (int i = 0; i < Upd::NS; ++i) u2.s[i] = upd.s[i] + c0;
        int rc = apply_cols<K, Upd, GT>(u2, V, ld, uniq, spos, sseg, seg_offsets, n, g + c0, ldg, rscale, gscale, Dc, vec, w, st, wide, ss, nv);
        if (rc != MREC_OK) return rc;
    }
    return MREC_OK;
}

template <class K, class GT = float>
int lazy_adam_impl(float* p, float* m, float* v, int64_t V, int64_t ld, int32_t D, const K* uniq,
                   const int32_t* spos, const int32_t* sseg, const int32_t* seg_offsets, int64_t n, const GT* g,
                   int64_t ldg, const float* rscale, float lr, float b1, float b2, float eps, float b1_pow,
                   float b2_pow, float gscale, int nesterov, void* ws, size_t ws_bytes, void* stream,
                   const WideArgs* wide = nullptr, StepState* ss = nullptr, const int64_t* nv = nullptr) {
    if (!uniq && n > 0) return MREC_EINVAL;
    UpdAdam u;
    u.s[0] = p; u.s[1] = m; u.s[2] = v;
    u.h.lr_t = lr * sqrtf(1.0f - b2_pow) / (1.0f - b1_pow);
    u.h.b1 = b1; u.h.b2 = b2; u.h.omb1 = 1.0f - b1; u.h.omb2 = 1.0f - b2; u.h.eps = eps; u.h.gscale = gscale;
    u.h.nesterov = nesterov;
    return apply_impl<K, UpdAdam, GT>(u, V, ld, D, uniq, spos, sseg, seg_offsets, n, g, ldg, rscale, gscale, ws, ws_bytes,
                                  stream, wide, ss, nv);
}

template <class K>
int ftrl_impl(float* var, float* accum, float* linear, int64_t V, int64_t ld, int32_t D, const K* uniq,
              const int32_t* spos, const int32_t* sseg, const int32_t* seg_offsets, int64_t n, const float* g,
              int64_t ldg, const float* rscale, float lr, float l1, float l2, float lr_power, float gscale, void* ws,
              size_t ws_bytes, void* stream) {
    if (!uniq && n > 0) return MREC_EINVAL;
    UpdFtrl u;
    u.s[0] = var; u.s[1] = accum; u.s[2] = linear;
    u.h = FtrlH{lr, l1, l2, lr_power, gscale};
    return apply_impl<K, UpdFtrl>(u, V, ld, D, uniq, spos, sseg, seg_offsets, n, g, ldg, rscale, gscale, ws, ws_bytes,
                                  stream);
}

}  // namespace

namespace {
__global__ void k_step_init(StepState* s, float b1p, float b2p, long long step) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) { s->b1p = b1p; s->b2p = b2p; s->lr_t = 0.f; s->pad0 = 0.f; s->step = step; s->stamps_off = 0; }
    if (i < kStampRing) {
        s->stamps[i][0] = ~0ull; s->stamps[i][1] = 0ull;
        s->aux[i][0] = ~0ull; s->aux[i][1] = 0ull; s->aux[i][2] = 0ull; s->aux[i][3] = 0ull;
        for (int j = 0; j < 64; ++j) s->ends[i][j] = 0ull;
    }
}
__global__ void k_step_advance(StepState* s, float lr, float b1, float b2) {
    const float b1p = s->b1p * b1, b2p = s->b2p * b2;
    const long long step = s->step + 1;
    __syncthreads();                                                 // (64 threads: every one has read the old step)
    s->ends[(unsigned)step % kStampRing][threadIdx.x & 63u] = 0ull;  // this step's k_apply_main end slots
    if (threadIdx.x != 0) return;
    s->b1p = b1p; s->b2p = b2p; s->step = step;
    s->lr_t = lr * sqrtf(1.0f - b2p) / (1.0f - b1p);      // same fp32 operations as the host-side entries
    s->stamps[(unsigned)step % kStampRing][0] = ~0ull;
    s->stamps[(unsigned)step % kStampRing][1] = 0ull;
    s->aux[(unsigned)step % kStampRing][2] = 0ull;                   // this step's k_apply_long end
    s->aux[(unsigned)(step + 1) % kStampRing][0] = ~0ull;            // the NEXT step's lookup (it runs before that step's advance)
    s->aux[(unsigned)(step + 1) % kStampRing][1] = 0ull;
}
// Which fields of a [B, F] batch are HOT COLUMNS: field f qualifies iff ONE id -- its candidate: the most frequent id among the field's
// first 16 samples, in at least a quarter of them -- fills at least `min_count` of its B samples, lies in [0, V) and occurs in no other
// field.  min_count = B: constant columns (the 13 one-id dense fields of a Criteo batch); smaller: also a field's dominant id (the
// bucket that rare categories fall into).  Every workgroup works the candidates out itself (16 F loads); a batch without one costs
// those loads and nothing else: workgroup 0 stores an empty mask.  With candidates, every thread compares its ids with the candidates'
// (LDS): matches are counted (LDS, then one device atomic per workgroup and candidate), a candidate seen in another field is
// reported in ONE device word; both by returning agent-scope atomics that are waited for, THEN the workgroup takes a ticket (so
// the ticket's last holder sees every workgroup's counts: atomics at the memory side, no fence -- a release fence here writes back
// whatever the kernels beside this one have dirtied in L2, 17 us per step when it was tried), and that last workgroup turns the
// survivors into the mask and the ids the apply's kernels read, and clears counters, word and ticket for the next batch: no memset
// node, no second launch.
struct ConstState { unsigned long long badmask; unsigned ticket, pad; unsigned long long mask; unsigned long long pad2; long long hid[64]; unsigned cnt[8][64]; };
static_assert(sizeof(ConstState) == MREC_CONST_COLS_STATE_BYTES, "mrec.h: MREC_CONST_COLS_STATE_BYTES");
template <class K>
__global__ __launch_bounds__(256) void k_const_cols(const K* __restrict__ ids, int64_t n, int F, int64_t V, unsigned min_count, ConstState* __restrict__ st) {
    __shared__ long long c[64];
    __shared__ int clist[64];
    __shared__ unsigned lbad[2], lcnt[64];
    __shared__ unsigned long long candm;
    __shared__ int is_last;
    const int t = (int)threadIdx.x;
    const int64_t B = n / F;
    if (t < 64) {
        bool ok = false;
        long long best = 0;
        if (t < F) {
            const int S = B < 16 ? (int)B : 16;
            long long v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = (long long)ids[(int64_t)(i < S ? i : 0) * F + t];
            int bc = 0;
#pragma unroll
            for (int i = 0; i < 16; ++i) {                    // the most frequent of the first S samples (the earliest on a tie)
                int ci = 0;
#pragma unroll
                for (int j = 0; j < 16; ++j) ci += (j < S && v[j] == v[i]) ? 1 : 0;
                if (i < S && ci > bc) { bc = ci; best = v[i]; }
            }
            ok = 4 * bc >= S && best >= 0 && best < V;
            c[t] = best;
        }
        lcnt[t] = 0u;
        const unsigned long long m = __ballot(ok);           // (threads 0-63 are wave 0)
        if (t == 0) { candm = m; lbad[0] = lbad[1] = 0u; }
        if (ok) clist[__popcll(m & ((1ull << t) - 1ull))] = t;
    }
    __syncthreads();
    const unsigned long long m = candm;
    if (m == 0ull) {
        if (blockIdx.x == 0 && t == 0) st->mask = 0ull;
        return;
    }
    // A thread keeps ONE field (lane = field: a wave reads a sample's F ids as one contiguous piece) and counts its candidate's matches
    // in a register; whether an id is ANOTHER field's candidate is asked of a 256-bit filter first, so the candidate loop runs for one id
    // in seven.  (First form: a thread walked the flat id list and counted in LDS -- with the bench's Zipf ids every field has a
    // candidate and 40 % of all ids match theirs: 300 k LDS atomics on 39 words, 130 us beside the backward GEMMs.)
    const int nc = __popcll(m);
    // the candidates as a 128-slot hash set in LDS (id -> its field): "is this id ANOTHER field's candidate?" is one probe for almost
    // every id.  (A 256-bit filter in front of a loop over the candidates was the form before: some lane of a wave always hit, the
    // loop -- two dependent LDS reads per candidate -- ran for every id: 118 of that kernel's 133 us on the bench's Zipf ids.)
    __shared__ unsigned long long hkey[128];
    __shared__ int hfld[128];
    if (t < 128) hkey[t] = ~0ull;
    __syncthreads();
    if (t < nc) {
        const int f1 = clist[t];
        const unsigned long long key = (unsigned long long)c[f1];
        unsigned slot = (unsigned)(key * 0x9E3779B97F4A7C15ull >> 57);
        for (int it = 0; it < 128; ++it) {
            const unsigned long long old = atomicCAS(&hkey[slot], ~0ull, key);
            if (old == ~0ull) { hfld[slot] = f1; break; }
            if (old == key) break;                            // (two fields, one candidate: both fail below; one entry is enough)
            slot = (slot + 1u) & 127u;
        }
    }
    __syncthreads();
    // (two fields with the SAME candidate fail each other: settled here once, so that the loop below need not ask about an id that is
    // its own field's candidate -- with the bench's Zipf ids that is 40 % of all ids)
    if (t < nc) {
        const int f1 = clist[t];
        for (int q = 0; q < nc; ++q)
            if (q != t && c[clist[q]] == c[f1]) atomicOr(&lbad[f1 >> 5], 1u << (f1 & 31));
    }
    const int f = t & 63, sub = t >> 6;
    if (f < F) {
        const bool mycand = (m >> f) & 1ull;
        const long long cf = c[f];
        unsigned mine = 0u;
        const int64_t step = (int64_t)gridDim.x * 4;
        for (int64_t b0 = (int64_t)blockIdx.x * 4 + sub; b0 < B; b0 += 4 * step) {
            long long idv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {                     // four samples in flight
                const int64_t b = b0 + u * step;
                idv[u] = (long long)ids[(b < B ? b : b0) * F + f];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (b0 + u * step >= B) continue;
                const long long id = idv[u];
                if (mycand && id == cf) { ++mine; continue; }
                if (id < 0) continue;                         // (nobody's candidate: they are rows of the table)
                unsigned slot = (unsigned)((unsigned long long)id * 0x9E3779B97F4A7C15ull >> 57);
                for (int it = 0; it < 128; ++it) {
                    const unsigned long long k = hkey[slot];
                    if (k == (unsigned long long)id) {
                        const int f2 = hfld[slot];
                        if (f2 != f && !((lbad[f2 >> 5] >> (f2 & 31)) & 1u)) atomicOr(&lbad[f2 >> 5], 1u << (f2 & 31));
                        break;
                    }
                    if (k == ~0ull) break;
                    slot = (slot + 1u) & 127u;
                }
            }
        }
        if (mine) atomicAdd(&lcnt[f], mine);
    }
    __syncthreads();
    if (t < 64) {
        if (lcnt[t]) {
            // (eight copies of the counters, a workgroup adds to copy number % 8: atomics on ONE device word serialise at ~270 ns each,
            // and 256 workgroups adding to 39 words took 70 us of this kernel's 73)
            unsigned old = atomicAdd(&st->cnt[blockIdx.x & 7u][t], lcnt[t]);
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(old) : : "memory");        // performed before the ticket is taken
        }
        if (t == 0) {
            const unsigned long long mine = (unsigned long long)lbad[0] | ((unsigned long long)lbad[1] << 32);
            if (mine) {
                unsigned long long old = atomicOr(&st->badmask, mine);
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(old) : : "memory");
            }
        }
    }
    __syncthreads();
    if (t == 0) is_last = atomicAdd(&st->ticket, 1u) == gridDim.x - 1u;
    __syncthreads();
    if (is_last && t < 64) {
        const unsigned long long b = atomicOr(&st->badmask, 0ull);
        unsigned cn = 0u;
#pragma unroll
        for (int r = 0; r < 8; ++r) cn += atomicExch(&st->cnt[r][t], 0u);
        const bool hot = ((m >> t) & 1ull) && !((b >> t) & 1ull) && cn >= min_count;
        const unsigned long long keep = __ballot(hot);
        st->hid[t] = hot ? c[t] : -1ll;
        if (t == 0) {
            st->mask = keep;
            atomicExch(&st->badmask, 0ull);
            atomicExch(&st->ticket, 0u);
        }
    }
}
}  // namespace

MREC_API int mrec_const_cols_detect(const void* ids, int32_t id_bytes, int64_t B, int32_t F, int64_t V, int64_t min_count, void* state,
                                    void* stream) {
    if (!ids || !state || (id_bytes != 4 && id_bytes != 8) || B <= 0 || F <= 0 || V <= 0 || min_count < 1) return MREC_EINVAL;
    if (F > 64) return MREC_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = B * F;
    if (min_count > B) min_count = B + 1;              // (nothing qualifies)
    unsigned blocks = (unsigned)mrec_cdiv(B, (int64_t)4 * 8);          // (a workgroup takes four samples per trip)
    if (blocks > 256u) blocks = 256u;
    if (id_bytes == 4) k_const_cols<int32_t><<<blocks, 256, 0, st>>>((const int32_t*)ids, n, F, V, (unsigned)min_count, (ConstState*)state);
    else k_const_cols<int64_t><<<blocks, 256, 0, st>>>((const int64_t*)ids, n, F, V, (unsigned)min_count, (ConstState*)state);
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}
MREC_API int mrec_sparse_apply_next_const_cols(const void* state, const void* ids, int32_t id_bytes, int64_t B) {
    if ((state == nullptr) != (ids == nullptr) || B < 0 || B > (int64_t(1) << 30) || (ids && id_bytes != 4 && id_bytes != 8)) return MREC_EINVAL;
    t_const = ConstCols{};
    if (state) { t_const.mask = &((const ConstState*)state)->mask; t_const.hid = ((const ConstState*)state)->hid; t_const.ids0 = ids; t_const.B = (int)B; t_const.id_bytes = id_bytes; }
    return MREC_OK;
}

MREC_API int mrec_step_state_init(void* state, float beta1_power, float beta2_power, int64_t step, void* stream) {
    if (!state || step < 0) return MREC_EINVAL;
    k_step_init<<<1, 256, 0, (hipStream_t)stream>>>((StepState*)state, beta1_power, beta2_power, (long long)step);
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}
MREC_API int mrec_step_advance(void* state, float lr, float beta1, float beta2, void* stream) {
    if (!state) return MREC_EINVAL;
    k_step_advance<<<1, 64, 0, (hipStream_t)stream>>>((StepState*)state, lr, beta1, beta2);
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}
MREC_API int mrec_wall_clock_khz(int32_t* out) {
    if (!out) return MREC_EINVAL;
    int dev = 0, khz = 0;
    MREC_HIP_CHECK(hipGetDevice(&dev));
    MREC_HIP_CHECK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev));
    *out = khz;
    return MREC_OK;
}

MREC_API int mrec_event_create(void** ev_out) {
    if (!ev_out) return MREC_EINVAL;
    hipEvent_t e;
    MREC_HIP_CHECK(hipEventCreate(&e));
    *ev_out = (void*)e;
    return MREC_OK;
}
MREC_API int mrec_event_destroy(void* ev) {
    if (!ev) return MREC_EINVAL;
    MREC_HIP_CHECK(hipEventDestroy((hipEvent_t)ev));
    return MREC_OK;
}
MREC_API int mrec_event_elapsed_ms(void* start, void* stop, float* ms_out) {
    if (!start || !stop || !ms_out) return MREC_EINVAL;
    MREC_HIP_CHECK(hipEventSynchronize((hipEvent_t)stop));
    MREC_HIP_CHECK(hipEventElapsedTime(ms_out, (hipEvent_t)start, (hipEvent_t)stop));
    return MREC_OK;
}
MREC_API int mrec_profile_next_apply(void* start, void* stop) {
    t_prof_start = (hipEvent_t)start;
    t_prof_stop = (hipEvent_t)stop;
    return MREC_OK;
}

MREC_API int mrec_sparse_apply_window(int32_t D, int aligned16) {
    return (aligned16 && D % 4 == 0) ? ACfg<4>::AW : ACfg<1>::AW;
}

MREC_API int mrec_sparse_apply_workspace_bytes(int64_t n, int32_t D, size_t* out) {
    if (!out || n < 0 || D <= 0) return MREC_EINVAL;
    *out = apply_ws_bytes(n, D);
    return MREC_OK;
}

MREC_API int mrec_segment_sum_f32(const int32_t* sorted_pos, const int32_t* sorted_seg, const int32_t* seg_offsets,
                                  int64_t n, const float* g, int64_t ldg, const float* row_scale, float grad_scale,
                                  int32_t D, float* out, void* ws, size_t ws_bytes, void* stream) {
    UpdStore u;
    u.s[0] = out;
    // rows are group numbers; there are at most n groups
    return apply_impl<int32_t, UpdStore>(u, n, D, D, (const int32_t*)nullptr, sorted_pos, sorted_seg, seg_offsets, n, g,
                                         ldg, row_scale, grad_scale, ws, ws_bytes, stream);
}

/* ... over 16-bit row gradients (g_kind 1: bf16, 2: IEEE half), widened exactly and summed in fp32: what a rank of a row-sharded
 * step sends an owner when it exchanges UNIQUE ids -- one fp32 sum per unique id instead of one 16-bit row per position. */
MREC_API int mrec_segment_sum_g16(const int32_t* sorted_pos, const int32_t* sorted_seg, const int32_t* seg_offsets,
                                  int64_t n, const void* g, int32_t g_kind, int64_t ldg, const float* row_scale, float grad_scale,
                                  int32_t D, float* out, void* ws, size_t ws_bytes, void* stream) {
    UpdStore u;
    u.s[0] = out;
    if (g_kind == 1)
        return apply_impl<int32_t, UpdStore, bf16_t>(u, n, D, D, (const int32_t*)nullptr, sorted_pos, sorted_seg, seg_offsets, n, (const bf16_t*)g,
                                                     ldg, row_scale, grad_scale, ws, ws_bytes, stream);
    if (g_kind == 2)
        return apply_impl<int32_t, UpdStore, f16_t>(u, n, D, D, (const int32_t*)nullptr, sorted_pos, sorted_seg, seg_offsets, n, (const f16_t*)g,
                                                    ldg, row_scale, grad_scale, ws, ws_bytes, stream);
    return MREC_EINVAL;
}

MREC_API int mrec_sparse_lazy_adam_f32_i32(float* p, float* m, float* v, int64_t V, int64_t ld, int32_t D,
                                           const int32_t* uniq, const int32_t* sorted_pos, const int32_t* sorted_seg,
                                           const int32_t* seg_offsets, int64_t n, const float* g, int64_t ldg,
                                           const float* row_scale, float lr, float b1, float b2, float eps,
                                           float b1_pow, float b2_pow, float grad_scale, int nesterov, void* ws,
                                           size_t ws_bytes, void* stream) {
    return lazy_adam_impl<int32_t>(p, m, v, V, ld, D, uniq, sorted_pos, sorted_seg, seg_offsets, n, g, ldg, row_scale,
                                   lr, b1, b2, eps, b1_pow, b2_pow, grad_scale, nesterov, ws, ws_bytes, stream);
}
MREC_API int mrec_sparse_lazy_adam_f32_i64(float* p, float* m, float* v, int64_t V, int64_t ld, int32_t D,
                                           const int64_t* uniq, const int32_t* sorted_pos, const int32_t* sorted_seg,
                                           const int32_t* seg_offsets, int64_t n, const float* g, int64_t ldg,
                                           const float* row_scale, float lr, float b1, float b2, float eps,
                                           float b1_pow, float b2_pow, float grad_scale, int nesterov, void* ws,
                                           size_t ws_bytes, void* stream) {
    return lazy_adam_impl<int64_t>(p, m, v, V, ld, D, uniq, sorted_pos, sorted_seg, seg_offsets, n, g, ldg, row_scale,
                                   lr, b1, b2, eps, b1_pow, b2_pow, grad_scale, nesterov, ws, ws_bytes, stream);
}

MREC_API int mrec_sparse_lazy_adam_bf16g_i32(float* p, float* m, float* v, int64_t V, int64_t ld, int32_t D,
                                             const int32_t* uniq, const int32_t* sorted_pos, const int32_t* sorted_seg,
                                             const int32_t* seg_offsets, int64_t n, const uint16_t* g, int64_t ldg,
                                             const float* row_scale, float lr, float b1, float b2, float eps,
                                             float b1_pow, float b2_pow, float grad_scale, int nesterov, void* ws,
                                             size_t ws_bytes, void* stream) {
    return lazy_adam_impl<int32_t, bf16_t>(p, m, v, V, ld, D, uniq, sorted_pos, sorted_seg, seg_offsets, n,
                                           (const bf16_t*)g, ldg, row_scale, lr, b1, b2, eps, b1_pow, b2_pow, grad_scale,
                                           nesterov, ws, ws_bytes, stream);
}
MREC_API int mrec_sparse_lazy_adam_bf16g_i64(float* p, float* m, float* v, int64_t V, int64_t ld, int32_t D,
                                             const int64_t* uniq, const int32_t* sorted_pos, const int32_t* sorted_seg,
                                             const int32_t* seg_offsets, int64_t n, const uint16_t* g, int64_t ldg,
                                             const float* row_scale, float lr, float b1, float b2, float eps,
                                             float b1_pow, float b2_pow, float grad_scale, int nesterov, void* ws,
                                             size_t ws_bytes, void* stream) {
    return lazy_adam_impl<int64_t, bf16_t>(p, m, v, V, ld, D, uniq, sorted_pos, sorted_seg, seg_offsets, n,
                                           (const bf16_t*)g, ldg, row_scale, lr, b1, b2, eps, b1_pow, b2_pow, grad_scale,
                                           nesterov, ws, ws_bytes, stream);
}

MREC_API int mrec_sparse_lazy_adam_f16g_i32(float* p, float* m, float* v, int64_t V, int64_t ld, int32_t D,
                                            const int32_t* uniq, const int32_t* sorted_pos, const int32_t* sorted_seg,
                                            const int32_t* seg_offsets, int64_t n, const uint16_t* g, int64_t ldg,
                                            const float* row_scale, float lr, float b1, float b2, float eps,
                                            float b1_pow, float b2_pow, float grad_scale, int nesterov, void* ws,
                                            size_t ws_bytes, void* stream) {
    return lazy_adam_impl<int32_t, f16_t>(p, m, v, V, ld, D, uniq, sorted_pos, sorted_seg, seg_offsets, n,
                                          (const f16_t*)g, ldg, row_scale, lr, b1, b2, eps, b1_pow, b2_pow, grad_scale,
                                          nesterov, ws, ws_bytes, stream);
}
MREC_API int mrec_sparse_lazy_adam_f16g_i64(float* p, float* m, float* v, int64_t V, int64_t ld, int32_t D,
                                            const int64_t* uniq, const int32_t* sorted_pos, const int32_t* sorted_seg,
                                            const int32_t* seg_offsets, int64_t n, const uint16_t* g, int64_t ldg,
                                            const float* row_scale, float lr, float b1, float b2, float eps,
                                            float b1_pow, float b2_pow, float grad_scale, int nesterov, void* ws,
                                            size_t ws_bytes, void* stream) {
    return lazy_adam_impl<int64_t, f16_t>(p, m, v, V, ld, D, uniq, sorted_pos, sorted_seg, seg_offsets, n,
                                          (const f16_t*)g, ldg, row_scale, lr, b1, b2, eps, b1_pow, b2_pow, grad_scale,
                                          nesterov, ws, ws_bytes, stream);
}

/* LazyAdam on the deep columns + FTRL on the wide record of the same fused rows, one pass (see include/mrec.h) */
MREC_API int mrec_sparse_lazy_adam_wide(float* p, float* m, float* v, int64_t V, int64_t ld, int32_t D, const void* uniq,
                                        int32_t uniq_bytes, const int32_t* sorted_pos, const int32_t* sorted_seg,
                                        const int32_t* seg_offsets, int64_t n, const void* g, int32_t g_kind, int64_t ldg,
                                        const float* row_scale, float lr, float b1, float b2, float eps, float b1_pow,
                                        float b2_pow, float grad_scale, int nesterov, const float* gw, int64_t gw_stride, int32_t F,
                                        int32_t wide_col, float ftrl_lr, float l1, float l2, float lr_power, void* ws,
                                        size_t ws_bytes, void* step_state, const int64_t* n_valid_dev, void* stream) {
    if ((uniq_bytes != 4 && uniq_bytes != 8) || g_kind < 0 || g_kind > 2 || gw_stride < 1 || gw_stride > (1 << 20)) return MREC_EINVAL;
    WideArgs wa;
    wa.gw = gw; wa.F = F; wa.wcol = wide_col; wa.magic = 0; wa.dummy = nullptr; wa.gws = (unsigned)gw_stride;
    wa.h = FtrlH{ftrl_lr, l1, l2, lr_power, grad_scale};
#define MREC_WIDE_CALL(KT, GT)                                                                                          \
    return lazy_adam_impl<KT, GT>(p, m, v, V, ld, D, (const KT*)uniq, sorted_pos, sorted_seg, seg_offsets, n, (const GT*)g, ldg, \
                                  row_scale, lr, b1, b2, eps, b1_pow, b2_pow, grad_scale, nesterov, ws, ws_bytes, stream, &wa,      \
                                  (StepState*)step_state, n_valid_dev)
    if (uniq_bytes == 4) {
        if (g_kind == 0) { MREC_WIDE_CALL(int32_t, float); }
        if (g_kind == 1) { MREC_WIDE_CALL(int32_t, bf16_t); }
        MREC_WIDE_CALL(int32_t, f16_t);
    }
    if (g_kind == 0) { MREC_WIDE_CALL(int64_t, float); }
    if (g_kind == 1) { MREC_WIDE_CALL(int64_t, bf16_t); }
    MREC_WIDE_CALL(int64_t, f16_t);
#undef MREC_WIDE_CALL
}

/* The same with the finishing pass handed back instead of launched (see include/mrec.h). */
MREC_API int mrec_sparse_lazy_adam_wide_defer(float* p, float* m, float* v, int64_t V, int64_t ld, int32_t D, const void* uniq,
                                              int32_t uniq_bytes, const int32_t* sorted_pos, const int32_t* sorted_seg,
                                              const int32_t* seg_offsets, int64_t n, const void* g, int32_t g_kind, int64_t ldg,
                                              const float* row_scale, float lr, float b1, float b2, float eps, float b1_pow,
                                              float b2_pow, float grad_scale, int nesterov, const float* gw, int64_t gw_stride, int32_t F,
                                              int32_t wide_col, float ftrl_lr, float l1, float l2, float lr_power, void* ws,
                                              size_t ws_bytes, void* step_state, const int64_t* n_valid_dev,
                                              mrec_apply_finish_t* finish_out, void* stream) {
    if (!finish_out) return MREC_EINVAL;
    if (D > 252) return MREC_EUNSUPPORTED;            // (one column block: one finishing pass)
    ApplyFinish* f = (ApplyFinish*)finish_out;
    f->magic = 0u;
    t_defer = f;
    const int rc = mrec_sparse_lazy_adam_wide(p, m, v, V, ld, D, uniq, uniq_bytes, sorted_pos, sorted_seg, seg_offsets, n, g, g_kind, ldg,
                                              row_scale, lr, b1, b2, eps, b1_pow, b2_pow, grad_scale, nesterov, gw, gw_stride, F, wide_col,
                                              ftrl_lr, l1, l2, lr_power, ws, ws_bytes, step_state, n_valid_dev, stream);
    t_defer = nullptr;
    if (rc == MREC_OK && f->magic != kFinishMagic) { f->lblocks = 0; f->cfin = 0; f->magic = kFinishMagic; }      // (n == 0: nothing to finish)
    return rc;
}

/* mrec_dense_adam_slabs_one_ftrl_f32 + the finishing pass a deferred apply handed back, one launch. */
MREC_API int mrec_dense_adam_slabs_finish_f32(float* p, float* m, float* v, const float* g, void* shadow16, int shadow_kind, int64_t n,
                                              int32_t nseg, const float* const* slabs, const int64_t* starts, const int64_t* lens,
                                              const int32_t* splits, float lr, float b1, float b2, float eps, float b1_pow, float b2_pow,
                                              float grad_scale, int nesterov, void* step_state, const mrec_ftrl1_t* one,
                                              const mrec_apply_finish_t* finish, void* stream) {
    if (!finish || n < 0 || nseg < 0 || nseg > 16 || shadow_kind < 0 || shadow_kind > 2) return MREC_EINVAL;
    const ApplyFinish* fp = (const ApplyFinish*)finish;
    if (fp->magic != kFinishMagic) return MREC_EINVAL;
    if (n == 0) {            /* an empty dense buffer is a valid input (as for mrec_dense_adam_slabs_f32): only the finishing pass runs */
        if (nseg != 0) return MREC_EINVAL;
        nseg = 0; shadow_kind = 0;
    }
    if (n > 0 && (!p || !m || !v || !g || (nseg > 0 && (!slabs || !starts || !lens || !splits)) || (shadow_kind && !shadow16))) return MREC_EINVAL;
    if (n % 4 || !al16(p) || !al16(m) || !al16(v) || !al16(g) || (shadow16 && (((uintptr_t)shadow16) & 7))) return MREC_EUNSUPPORTED;
    SlabSegs sg;
    sg.n = nseg;
    for (int q = 0; q < nseg; ++q) {
        if (!slabs[q] || starts[q] < 0 || lens[q] <= 0 || starts[q] % 4 || lens[q] % 4 || starts[q] + lens[q] > n || splits[q] <= 0 ||
            !al16(slabs[q]))
            return MREC_EINVAL;
        sg.part[q] = (const float4*)slabs[q];
        sg.start4[q] = starts[q] / 4;
        sg.len4[q] = lens[q] / 4;
        sg.S[q] = splits[q];
    }
    DenseAdamSlabArgs a;
    a.p = (float4*)p; a.m = (float4*)m; a.v = (float4*)v; a.g = (const float4*)g; a.n4 = n / 4; a.shadow = (uint2*)shadow16;
    a.ss = (const StepState*)step_state;
    a.h.lr_t = lr * sqrtf(1.0f - b2_pow) / (1.0f - b1_pow);
    a.h.b1 = b1; a.h.b2 = b2; a.h.omb1 = 1.0f - b1; a.h.omb2 = 1.0f - b2; a.h.eps = eps; a.h.gscale = grad_scale; a.h.nesterov = nesterov;
    a.f1.idx = -1;
    a.f1.h = FtrlH{1.0f, 0.0f, 0.0f, -0.5f, grad_scale};
    if (one && one->index >= 0) {
        if (one->index >= n || !(one->lr > 0.0f) || one->l1 < 0.0f || one->l2 < 0.0f || one->lr_power > 0.0f) return MREC_EINVAL;
        a.f1.idx = one->index;
        a.f1.h = FtrlH{one->lr, one->l1, one->l2, one->lr_power, grad_scale};
    }
    int64_t ab = mrec_cdiv(a.n4, 256);
    if (ab > 256 * 16) ab = 256 * 16;
    const unsigned grid = fp->lblocks + fp->cfin + (unsigned)ab;
    if (grid == 0) return MREC_OK;
    hipStream_t st = (hipStream_t)stream;
#define MREC_FIN(KT)                                                                                      \
    do {                                                                                                   \
        if (shadow_kind == 1) k_finish_dense_adam<KT, 1><<<grid, 256, 0, st>>>(*fp, a, sg);                \
        else if (shadow_kind == 2) k_finish_dense_adam<KT, 2><<<grid, 256, 0, st>>>(*fp, a, sg);           \
        else k_finish_dense_adam<KT, 0><<<grid, 256, 0, st>>>(*fp, a, sg);                                 \
    } while (0)
    if (fp->key_bytes == 8) MREC_FIN(int64_t); else MREC_FIN(int32_t);
#undef MREC_FIN
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}

MREC_API int mrec_sparse_ftrl_f32_i32(float* var, float* accum, float* linear, int64_t V, int64_t ld, int32_t D,
                                      const int32_t* uniq, const int32_t* sorted_pos, const int32_t* sorted_seg,
                                      const int32_t* seg_offsets, int64_t n, const float* g, int64_t ldg,
                                      const float* row_scale, float lr, float l1, float l2, float lr_power,
                                      float grad_scale, void* ws, size_t ws_bytes, void* stream) {
    return ftrl_impl<int32_t>(var, accum, linear, V, ld, D, uniq, sorted_pos, sorted_seg, seg_offsets, n, g, ldg,
                              row_scale, lr, l1, l2, lr_power, grad_scale, ws, ws_bytes, stream);
}
MREC_API int mrec_sparse_ftrl_f32_i64(float* var, float* accum, float* linear, int64_t V, int64_t ld, int32_t D,
                                      const int64_t* uniq, const int32_t* sorted_pos, const int32_t* sorted_seg,
                                      const int32_t* seg_offsets, int64_t n, const float* g, int64_t ldg,
                                      const float* row_scale, float lr, float l1, float l2, float lr_power,
                                      float grad_scale, void* ws, size_t ws_bytes, void* stream) {
    return ftrl_impl<int64_t>(var, accum, linear, V, ld, D, uniq, sorted_pos, sorted_seg, seg_offsets, n, g, ldg,
                              row_scale, lr, l1, l2, lr_power, grad_scale, ws, ws_bytes, stream);
}
