// mrec_route.hip -- row-shard routing for the hybrid-parallel embedding (one process per GPU,
// RCCL all-to-all over xGMI), for gfx950.
//
// Reference: MindRec's hybrid mode row-slices the table across devices (README.md:140-144;
// nn.EmbeddingLookup(..., slice_mode=TABLE_ROW_SLICE) at models/wide_deep/src/wide_and_deep.py:232-249)
// and lets MindSpore auto-parallel all-reduce [N, D] partials [EXT].  xGMI is point-to-point, so
// here ids are bucketed by owner (id mod n_shards), exchanged with one all-to-all, gathered
// locally and returned with a second all-to-all.  This file is the bucketing:
// a one-pass stable radix split of positions by owner, reusing mrec_radix.h.
#include "mrec_common.h"
#include "mrec_radix.h"

namespace {

// HASH: the table is a hash table keyed by the raw id (MapParameter): owner = hash(key) mod n_shards, taken from the HIGH
// bits of the 64-bit mix -- the owner's key index probes from the low bits of the same mix, and keys that share their low
// bits must not all land on one owner -- and the raw key travels (the owner translates it to a row of its own index).
template <class K>
__device__ __forceinline__ int owner_of(K id, int S, bool hash) {
    if (hash) return (int)((mrec_mix64((uint64_t)(int64_t)id) >> 33) % (uint64_t)S);
    K o = id % (K)S;
    if (o < 0) o += (K)S;
    return (int)o;
}

template <class K>
__global__ __launch_bounds__(256) void k_owner(const K* __restrict__ ids, int64_t n, int S, int* __restrict__ owner, bool hash,
                                               int rot = 0, const int64_t* __restrict__ nv = nullptr) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int c = owner_of(ids[i], S, hash) - rot;      // the message chunk of the id's owner (rot = 0: chunk = owner)
    if (c < 0) c += S;
    if (nv && i >= *nv) c = S;                    // not an entry (the list's length lives on the device): nobody's, sorted behind every owner's
    owner[i] = c;
}

template <class K>
__global__ __launch_bounds__(256) void k_route_finish(const K* __restrict__ ids, int64_t n, int S,
                                                      const int* __restrict__ perm, const int* __restrict__ dbase,
                                                      K* __restrict__ send_local, int64_t* __restrict__ counts, bool hash) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k < S) counts[k] = (int64_t)((k + 1 < S) ? dbase[k + 1] : (int)n) - dbase[k];
    if (k >= n) return;
    const K id = ids[perm[k]];
    if (hash) { send_local[k] = id; return; }
    const K o = (K)owner_of(id, S, false);
    send_local[k] = (id - o) / (K)S;
}

__global__ void k_zero_counts(int64_t* counts, int S) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k < S) counts[k] = 0;
}

// dst[dst_row(k), :] = src[src_row(k), :] * scale[perm[k]]  (SCATTER: dst_row = perm[k], src_row = k;
// otherwise dst_row = k, src_row = perm[k]).  Lane-group of D/VEC lanes per row (float4 when aligned),
// PB rows per lane-group loaded before the first store.
constexpr int PB = 4;
template <bool SCATTER, int VEC>
__global__ __launch_bounds__(256) void k_perm_rows(const float* __restrict__ src, int64_t lds,
                                                   const int* __restrict__ perm, int64_t n, int D, int lpr, int G,
                                                   const float* __restrict__ row_scale, float* __restrict__ dst, int64_t ldd) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane / lpr, sub = lane - grp * lpr;
    if (grp >= G) return;
    const int64_t k0 = ((int64_t)blockIdx.x * 4 + wave) * (G * PB);
    float x[PB][VEC];
    int64_t p[PB];
    float s[PB];
    // (the PB permutation entries, then the PB scales and rows, each requested together at clamped indices, and "needed"
    // before the first store -- see k_unroute_slots)
    const int64_t nl = n > 0 ? n - 1 : 0;
#pragma unroll
    for (int q = 0; q < PB; ++q) {
        const int64_t k = k0 + (int64_t)q * G + grp;
        p[q] = perm[k < nl ? k : nl];
    }
#pragma unroll
    for (int q = 0; q < PB; ++q) {
        const int64_t k = k0 + (int64_t)q * G + grp;
        const int64_t kc = k < nl ? k : nl;
        s[q] = row_scale ? row_scale[p[q]] : 1.0f;
        const float* a = (SCATTER ? src + kc * lds : src + p[q] * lds) + sub * VEC;
        if (VEC == 4) {
            const float4 v = *(const float4*)a;
            x[q][0] = v.x; x[q][1 % VEC] = v.y; x[q][2 % VEC] = v.z; x[q][3 % VEC] = v.w;
        } else {
            x[q][0] = *a;
        }
    }
#pragma unroll
    for (int q = 0; q < PB; ++q) {
        const int64_t k = k0 + (int64_t)q * G + grp;
#pragma unroll
        for (int c = 0; c < VEC; ++c) asm volatile("" : "+v"(x[q][c]));
        asm volatile("" : "+v"(s[q]));
        if (k >= n) p[q] = -1;
    }
#pragma unroll
    for (int q = 0; q < PB; ++q) {
        const int64_t k = k0 + (int64_t)q * G + grp;
        if (p[q] >= 0) {
            float* o = (SCATTER ? dst + p[q] * ldd : dst + k * ldd) + sub * VEC;
            if (VEC == 4) {
                float4 v = make_float4(x[q][0], x[q][1 % VEC], x[q][2 % VEC], x[q][3 % VEC]);
                if (row_scale) { v.x *= s[q]; v.y *= s[q]; v.z *= s[q]; v.w *= s[q]; }
                *(float4*)o = v;
            } else {
                *o = row_scale ? x[q][0] * s[q] : x[q][0];
            }
        }
    }
}

// generic fallback: one wave per row, lanes stride the columns
template <bool SCATTER>
__global__ __launch_bounds__(256) void k_perm_rows_generic(const float* __restrict__ src, int64_t lds,
                                                           const int* __restrict__ perm, int64_t n, int D,
                                                           const float* __restrict__ row_scale,
                                                           float* __restrict__ dst, int64_t ldd) {
    const int lane = threadIdx.x & 63;
    const int64_t k = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= n) return;
    const int64_t p = perm[k];
    const float s = row_scale ? row_scale[p] : 1.0f;
    const float* a = SCATTER ? src + k * lds : src + p * lds;
    float* o = SCATTER ? dst + p * ldd : dst + k * ldd;
    for (int c = lane; c < D; c += 64) o[c] = row_scale ? a[c] * s : a[c];
}

template <bool SCATTER>
int perm_rows_launch(const float* src, int64_t lds, const int* perm, int64_t n, int D, const float* row_scale,
                     float* dst, hipStream_t st, int64_t ldd = 0) {
    if (ldd == 0) ldd = D;
    const bool al = ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0 && (lds % 4 == 0) && (ldd % 4 == 0);
    if (D % 4 == 0 && D <= 256 && al) {
        const int lpr = D / 4, G = 64 / lpr;
        k_perm_rows<SCATTER, 4><<<(unsigned)mrec_cdiv(n, (int64_t)4 * G * PB), 256, 0, st>>>(src, lds, perm, n, D, lpr, G,
                                                                                           row_scale, dst, ldd);
    } else if (D <= 64) {
        const int lpr = D, G = 64 / lpr;
        k_perm_rows<SCATTER, 1><<<(unsigned)mrec_cdiv(n, (int64_t)4 * G * PB), 256, 0, st>>>(src, lds, perm, n, D, lpr, G,
                                                                                           row_scale, dst, ldd);
    } else {
        k_perm_rows_generic<SCATTER><<<(unsigned)mrec_cdiv(n, 4), 256, 0, st>>>(src, lds, perm, n, D, row_scale, dst, ldd);
    }
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}

// (id, weight) pairs of one request message: out[k] = {send_local[k], bits(wts[perm[k]])}; and their split on arrival
__global__ __launch_bounds__(256) void k_pack_iw(const int* __restrict__ send_local, const float* __restrict__ wts,
                                                 const int* __restrict__ perm, int64_t n, int2* __restrict__ out) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k < n) out[k] = make_int2(send_local[k], __float_as_int(wts[perm[k]]));
}
__global__ __launch_bounds__(256) void k_unpack_iw(const int2* __restrict__ in, int64_t n, int* __restrict__ ids,
                                                   float* __restrict__ wts) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k < n) {
        const int2 v = in[k];
        ids[k] = v.x;
        wts[k] = __int_as_float(v.y);
    }
}

// ---- fixed-capacity routing (no host round trip: every message of the step has a static shape) ------------------------------
// A request entry: the id (dense tables: the owner's local row; hash tables: the raw key) and the position's weight, which
// travels so that the OWNER applies the mask in fp32 and rounds once.  int32 ids: 8 bytes; int64 ids: 16 bytes.
template <class K> struct ReqEntry;
template <> struct ReqEntry<int32_t> { int32_t id; float wt; };
template <> struct ReqEntry<int64_t> { int64_t id; float wt; int32_t pad; };

// Thread k < n: sorted position k (bucket order, stable) takes slot chunk * cap + (k - bucket start), chunk = (owner - rot)
// mod S (rot = 0: the owner itself; the engine rotates so that a rank's own chunk is the LAST one, see mrec.h); positions past a
// bucket's capacity are dropped and counted.  Thread t < S * cap: slots past a bucket's count carry id -1 (the owner's gather
// skips them, its plan sorts them behind everything else).
template <class K>
__global__ __launch_bounds__(256) void k_route_slots(const K* __restrict__ ids, const float* __restrict__ wts, int64_t n, int S,
                                                     int64_t cap, const int* __restrict__ perm, const int* __restrict__ dbase,
                                                     ReqEntry<K>* __restrict__ req, int* __restrict__ slot_of_pos,
                                                     int* __restrict__ pos_of_slot, unsigned long long* __restrict__ overflow,
                                                     bool hash, int rot, const int64_t* __restrict__ nv = nullptr) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t < n) {
        const int pos = perm[t];
        const K id = ids[pos];
        const int o = owner_of(id, S, hash);
        int c = o - rot;                                   // the owner's chunk of the message
        if (c < 0) c += S;
        const int64_t j = t - dbase[c];
        if (nv && pos >= *nv) {
            slot_of_pos[pos] = -1;                         // past the list's device-side length: no entry, no slot, no overflow
        } else if (j < cap) {
            const int64_t s = (int64_t)c * cap + j;
            ReqEntry<K> e{};
            e.id = hash ? id : (id - (K)o) / (K)S;
            e.wt = wts ? wts[pos] : 1.0f;
            req[s] = e;
            slot_of_pos[pos] = (int)s;
            pos_of_slot[s] = pos;
        } else {
            slot_of_pos[pos] = -1;
            atomicAdd(overflow, 1ull);
        }
    }
    if (t < (int64_t)S * cap) {
        const int o = (int)(t / cap);
        const int64_t j = t - (int64_t)o * cap;
        const int64_t cnt = (int64_t)((o + 1 < S || nv) ? dbase[o + 1] : (int)n) - dbase[o];      // (nv: bucket S -- nobody's -- starts at dbase[S])
        if (j >= cnt) {
            ReqEntry<K> e{};
            e.id = (K)-1;
            e.wt = 0.0f;
            req[t] = e;
            pos_of_slot[t] = -1;
        }
    }
}

template <class K>
__global__ __launch_bounds__(256) void k_unpack_req(const ReqEntry<K>* __restrict__ req, int64_t n, K* __restrict__ ids,
                                                    float* __restrict__ wts) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k < n) {
        const ReqEntry<K> e = req[k];
        ids[k] = e.id;
        wts[k] = e.wt;
    }
}

// The answers back in position order: position p reads message row slot_of_pos[p] ([Dw words of the looked-up row | wide
// product, 0 | pad], W words) and writes its row of the MLP input and its (product, 0) pair -- exactly what the one-GPU fused
// lookup hands out.  A lane per 16 bytes; the lane behind the row's last moves the pair.
__global__ __launch_bounds__(256) void k_unroute_slots(const float* __restrict__ back, int64_t W, const int* __restrict__ slot_of_pos,
                                                       int64_t n, int Dw, int lpr, int G, float* __restrict__ emb,
                                                       float* __restrict__ wprod) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane / lpr, sub = lane - grp * lpr;
    if (grp >= G) return;
    const int64_t p0 = ((int64_t)blockIdx.x * 4 + wave) * (G * PB);
    const bool wl = sub * 4 >= Dw;
    float4 x[PB];
    int s[PB];
    // (all loads at clamped / valid addresses and "needed" before the first store: guarded, every store was preceded by a wait
    // that also waited for the store before it -- vmcnt counts loads and stores in order; see k_gather_rows in mrec_gather.hip)
    const int64_t nl = n > 0 ? n - 1 : 0;
#pragma unroll
    for (int q = 0; q < PB; ++q) {
        const int64_t p = p0 + (int64_t)q * G + grp;
        s[q] = slot_of_pos[p < nl ? p : nl];
    }
#pragma unroll
    for (int q = 0; q < PB; ++q) {
        const int64_t p = p0 + (int64_t)q * G + grp;
        if (p >= n) s[q] = -1;
        x[q] = *(const float4*)(back + (int64_t)(s[q] >= 0 ? s[q] : 0) * W + sub * 4);
    }
#pragma unroll
    for (int q = 0; q < PB; ++q) {
        asm volatile("" : "+v"(x[q].x), "+v"(x[q].y), "+v"(x[q].z), "+v"(x[q].w));
        if (s[q] < 0) x[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int q = 0; q < PB; ++q) {
        const int64_t p = p0 + (int64_t)q * G + grp;
        if (p >= n) continue;
        if (wl) *(float2*)(wprod + 2 * p) = make_float2(x[q].x, 0.0f);
        else *(float4*)(emb + p * Dw + sub * 4) = x[q];
    }
}

// The gradient message in slot order: slot s of position p = pos_of_slot[s] carries [Dw words of p's row gradient | the wide
// branch's gradient of p's sample (the head's dlogit) | pad]; padding slots are left alone (nobody reads them).
__global__ __launch_bounds__(256) void k_route_grads(const float* __restrict__ g, int64_t ldg, const float* __restrict__ dlogit,
                                                     unsigned magic, int F, const int* __restrict__ pos_of_slot, int64_t n_slots,
                                                     int Dw, int lpr, int G, float* __restrict__ msg, int64_t W) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane / lpr, sub = lane - grp * lpr;
    if (grp >= G) return;
    const int64_t s0 = ((int64_t)blockIdx.x * 4 + wave) * (G * PB);
    const bool wl = sub * 4 >= Dw;
    float4 x[PB];
    float dl[PB];
    int p[PB];
    const int64_t sl = n_slots > 0 ? n_slots - 1 : 0;
#pragma unroll
    for (int q = 0; q < PB; ++q) {
        const int64_t s = s0 + (int64_t)q * G + grp;
        p[q] = pos_of_slot[s < sl ? s : sl];
    }
#pragma unroll
    for (int q = 0; q < PB; ++q) {
        const int64_t s = s0 + (int64_t)q * G + grp;
        if (s >= n_slots) p[q] = -1;
        const unsigned pp = p[q] >= 0 ? (unsigned)p[q] : 0u;       // (a padding slot reads position 0: dropped)
        dl[q] = dlogit[F == 1 ? pp : __umulhi(pp, magic)];
        x[q] = *(const float4*)(g + (int64_t)pp * ldg + (wl ? 0 : sub * 4));      // (the pair's lane: column 0, unused)
    }
#pragma unroll
    for (int q = 0; q < PB; ++q) {
        asm volatile("" : "+v"(x[q].x), "+v"(x[q].y), "+v"(x[q].z), "+v"(x[q].w), "+v"(dl[q]));
        if (wl) x[q] = make_float4(dl[q], 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int q = 0; q < PB; ++q) {
        const int64_t s = s0 + (int64_t)q * G + grp;
        if (p[q] >= 0) *(float4*)(msg + s * W + sub * 4) = x[q];
    }
}

template <class K>
int route_slots_impl(const K* ids, const float* wts, int64_t n, int32_t S, int64_t cap, int hashed, int32_t rot, void* req, int32_t* slot_of_pos,
                     int32_t* pos_of_slot, int64_t* overflow_dev, void* ws, size_t ws_bytes, void* stream, const int64_t* nv = nullptr) {
    hipStream_t st = (hipStream_t)stream;
    if (n <= 0 || S <= 0 || S + (nv ? 1 : 0) > RNB || cap <= 0 || !ids || !req || !slot_of_pos || !pos_of_slot || !overflow_dev || !ws) return MREC_EINVAL;
    if (rot < 0 || rot >= S) return MREC_EINVAL;
    if (n > (int64_t(1) << 30) || (int64_t)S * cap > (int64_t(1) << 30)) return MREC_EUNSUPPORTED;
    if (((uintptr_t)req) & (sizeof(ReqEntry<K>) - 1)) return MREC_EINVAL;
    const int nblk = (int)mrec_cdiv(n, RT);
    MrecArena a(ws, ws_bytes);
    int* hist = a.take<int>((size_t)nblk * RNB);
    int* hscan = a.take<int>((size_t)nblk * RNB);
    int* totals = a.take<int>(RNB);
    int* dbase = a.take<int>(RNB);
    int* owner = a.take<int>(n);
    int* okeys = a.take<int>(n);
    int* perm = a.take<int>(n);
    if (!a.ok) return MREC_EWORKSPACE;
    int nbits = 1;
    while ((1 << nbits) < S + (nv ? 1 : 0)) ++nbits;
    k_owner<K><<<(unsigned)mrec_cdiv(n, 256), 256, 0, st>>>(ids, n, S, owner, hashed != 0, rot, nv);
    radix_pass(owner, nullptr, (int)n, 0, nbits, hist, hscan, totals, dbase, okeys, perm, st);
    const int64_t m = n > (int64_t)S * cap ? n : (int64_t)S * cap;
    k_route_slots<K><<<(unsigned)mrec_cdiv(m, 256), 256, 0, st>>>(ids, wts, n, S, cap, perm, dbase, (ReqEntry<K>*)req, slot_of_pos,
                                                                 pos_of_slot, (unsigned long long*)overflow_dev, hashed != 0, rot, nv);
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}

template <class K>
int route_impl(const K* ids, int64_t n, int32_t S, K* send_local, int32_t* send_perm, int64_t* counts_dev, void* ws,
               size_t ws_bytes, void* stream, bool hash = false) {
    hipStream_t st = (hipStream_t)stream;
    if (n < 0 || S <= 0 || S > RNB || !counts_dev) return MREC_EINVAL;
    if (n == 0) {
        k_zero_counts<<<(unsigned)mrec_cdiv(S, 256), 256, 0, st>>>(counts_dev, S);
        MREC_LAUNCH_CHECK();
        return MREC_OK;
    }
    if (!ids || !send_local || !send_perm || !ws) return MREC_EINVAL;
    if (n > (int64_t(1) << 30)) return MREC_EUNSUPPORTED;
    const int nblk = (int)mrec_cdiv(n, RT);
    MrecArena a(ws, ws_bytes);
    int* hist = a.take<int>((size_t)nblk * RNB);
    int* hscan = a.take<int>((size_t)nblk * RNB);
    int* totals = a.take<int>(RNB);
    int* dbase = a.take<int>(RNB);
    int* owner = a.take<int>(n);
    int* okeys = a.take<int>(n);
    if (!a.ok) return MREC_EWORKSPACE;
    int nbits = 1;
    while ((1 << nbits) < S) ++nbits;
    k_owner<K><<<(unsigned)mrec_cdiv(n, 256), 256, 0, st>>>(ids, n, S, owner, hash);
    radix_pass(owner, nullptr, (int)n, 0, nbits, hist, hscan, totals, dbase, okeys, send_perm, st);
    const int64_t m = n > S ? n : S;
    k_route_finish<K><<<(unsigned)mrec_cdiv(m, 256), 256, 0, st>>>(ids, n, S, send_perm, dbase, send_local, counts_dev, hash);
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}

}  // namespace

MREC_API int mrec_shard_route_workspace_bytes(int64_t n, int32_t n_shards, size_t* out) {
    if (!out || n < 0 || n_shards <= 0) return MREC_EINVAL;
    const size_t nn = (size_t)(n ? n : 1);
    *out = mrec_align_up((size_t)mrec_cdiv(nn, RT) * RNB * 4, 256) * 2 + mrec_align_up((size_t)RNB * 4, 256) * 2 +
           mrec_align_up(nn * 4, 256) * 2;
    return MREC_OK;
}

MREC_API int mrec_shard_route_hash_i32(const int32_t* keys, int64_t n, int32_t n_shards, int32_t* send_keys,
                                       int32_t* send_perm, int64_t* counts_dev, void* ws, size_t ws_bytes, void* stream) {
    return route_impl<int32_t>(keys, n, n_shards, send_keys, send_perm, counts_dev, ws, ws_bytes, stream, true);
}
MREC_API int mrec_shard_route_hash_i64(const int64_t* keys, int64_t n, int32_t n_shards, int64_t* send_keys,
                                       int32_t* send_perm, int64_t* counts_dev, void* ws, size_t ws_bytes, void* stream) {
    return route_impl<int64_t>(keys, n, n_shards, send_keys, send_perm, counts_dev, ws, ws_bytes, stream, true);
}

MREC_API int mrec_shard_route_i32(const int32_t* ids, int64_t n, int32_t n_shards, int32_t* send_local,
                                  int32_t* send_perm, int64_t* counts_dev, void* ws, size_t ws_bytes, void* stream) {
    return route_impl<int32_t>(ids, n, n_shards, send_local, send_perm, counts_dev, ws, ws_bytes, stream);
}

MREC_API int mrec_shard_route_i64(const int64_t* ids, int64_t n, int32_t n_shards, int64_t* send_local,
                                  int32_t* send_perm, int64_t* counts_dev, void* ws, size_t ws_bytes, void* stream) {
    return route_impl<int64_t>(ids, n, n_shards, send_local, send_perm, counts_dev, ws, ws_bytes, stream);
}

MREC_API int mrec_shard_unroute_f32(const float* rows, const int32_t* send_perm, int64_t n, int32_t D,
                                    const float* row_scale, float* out, void* stream) {
    if (n < 0 || D <= 0) return MREC_EINVAL;
    if (n == 0) return MREC_OK;
    if (!rows || !send_perm || !out) return MREC_EINVAL;
    return perm_rows_launch<true>(rows, D, send_perm, n, D, row_scale, out, (hipStream_t)stream);
}

MREC_API int mrec_shard_route_rows_f32(const float* g, int64_t ldg, const int32_t* send_perm, int64_t n, int32_t D,
                                       const float* row_scale, float* rows_out, void* stream) {
    if (n < 0 || D <= 0 || ldg < D) return MREC_EINVAL;
    if (n == 0) return MREC_OK;
    if (!g || !send_perm || !rows_out) return MREC_EINVAL;
    return perm_rows_launch<false>(g, ldg, send_perm, n, D, row_scale, rows_out, (hipStream_t)stream);
}

/* The same two permutations with explicit row strides on both sides (messages that carry more than one thing per row:
 * [16-bit embedding row | wide value | pad], see mindrec_amd/wide_deep.py) */
MREC_API int mrec_shard_unroute_ld_f32(const float* rows, int64_t ldr, const int32_t* send_perm, int64_t n, int32_t D,
                                       const float* row_scale, float* out, int64_t ldo, void* stream) {
    if (n < 0 || D <= 0 || ldr < D || ldo < D) return MREC_EINVAL;
    if (n == 0) return MREC_OK;
    if (!rows || !send_perm || !out) return MREC_EINVAL;
    return perm_rows_launch<true>(rows, ldr, send_perm, n, D, row_scale, out, (hipStream_t)stream, ldo);
}
MREC_API int mrec_shard_route_rows_ld_f32(const float* g, int64_t ldg, const int32_t* send_perm, int64_t n, int32_t D,
                                          const float* row_scale, float* rows_out, int64_t ldo, void* stream) {
    if (n < 0 || D <= 0 || ldg < D || ldo < D) return MREC_EINVAL;
    if (n == 0) return MREC_OK;
    if (!g || !send_perm || !rows_out) return MREC_EINVAL;
    return perm_rows_launch<false>(g, ldg, send_perm, n, D, row_scale, rows_out, (hipStream_t)stream, ldo);
}
MREC_API int mrec_shard_pack_iw_i32(const int32_t* send_local, const float* wts, const int32_t* send_perm, int64_t n,
                                    int32_t* out_pairs, void* stream) {
    if (n < 0) return MREC_EINVAL;
    if (n == 0) return MREC_OK;
    if (!send_local || !wts || !send_perm || !out_pairs || (((uintptr_t)out_pairs) & 7)) return MREC_EINVAL;
    k_pack_iw<<<(unsigned)mrec_cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(send_local, wts, send_perm, n, (int2*)out_pairs);
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}
MREC_API int mrec_shard_unpack_iw_i32(const int32_t* pairs, int64_t n, int32_t* ids_out, float* wts_out, void* stream) {
    if (n < 0) return MREC_EINVAL;
    if (n == 0) return MREC_OK;
    if (!pairs || !ids_out || !wts_out || (((uintptr_t)pairs) & 7)) return MREC_EINVAL;
    k_unpack_iw<<<(unsigned)mrec_cdiv(n, 256), 256, 0, (hipStream_t)stream>>>((const int2*)pairs, n, ids_out, wts_out);
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}

/* ---- fixed-capacity routing: every message of a sharded step has a static shape (include/mrec.h) ---- */
MREC_API int mrec_shard_route_slots_workspace_bytes(int64_t n, int32_t n_shards, size_t* out) {
    if (!out || n < 0 || n_shards <= 0) return MREC_EINVAL;
    const size_t nn = (size_t)(n ? n : 1);
    *out = mrec_align_up((size_t)mrec_cdiv(nn, RT) * RNB * 4, 256) * 2 + mrec_align_up((size_t)RNB * 4, 256) * 2 +
           mrec_align_up(nn * 4, 256) * 3;
    return MREC_OK;
}
MREC_API int mrec_shard_route_slots_i32(const int32_t* ids, const float* wts, int64_t n, int32_t n_shards, int64_t cap, int hashed,
                                        int32_t chunk_rot, void* req, int32_t* slot_of_pos, int32_t* pos_of_slot, int64_t* overflow_dev,
                                        void* ws, size_t ws_bytes, void* stream) {
    return route_slots_impl<int32_t>(ids, wts, n, n_shards, cap, hashed, chunk_rot, req, slot_of_pos, pos_of_slot, overflow_dev, ws, ws_bytes, stream);
}
MREC_API int mrec_shard_route_slots_i64(const int64_t* ids, const float* wts, int64_t n, int32_t n_shards, int64_t cap, int hashed,
                                        int32_t chunk_rot, void* req, int32_t* slot_of_pos, int32_t* pos_of_slot, int64_t* overflow_dev,
                                        void* ws, size_t ws_bytes, void* stream) {
    return route_slots_impl<int64_t>(ids, wts, n, n_shards, cap, hashed, chunk_rot, req, slot_of_pos, pos_of_slot, overflow_dev, ws, ws_bytes, stream);
}
/* ... of a list whose length lives on the device (a step's UNIQUE ids: mrec_dedup_* writes n_uniq_dev): entries at and past
 * *n_valid_dev are nobody's -- no slot (slot_of_pos = -1), no overflow. */
MREC_API int mrec_shard_route_slots_nv_i32(const int32_t* ids, const float* wts, int64_t n, const int64_t* n_valid_dev, int32_t n_shards,
                                           int64_t cap, int hashed, int32_t chunk_rot, void* req, int32_t* slot_of_pos, int32_t* pos_of_slot,
                                           int64_t* overflow_dev, void* ws, size_t ws_bytes, void* stream) {
    if (!n_valid_dev) return MREC_EINVAL;
    return route_slots_impl<int32_t>(ids, wts, n, n_shards, cap, hashed, chunk_rot, req, slot_of_pos, pos_of_slot, overflow_dev, ws, ws_bytes, stream,
                                     n_valid_dev);
}
MREC_API int mrec_shard_route_slots_nv_i64(const int64_t* ids, const float* wts, int64_t n, const int64_t* n_valid_dev, int32_t n_shards,
                                           int64_t cap, int hashed, int32_t chunk_rot, void* req, int32_t* slot_of_pos, int32_t* pos_of_slot,
                                           int64_t* overflow_dev, void* ws, size_t ws_bytes, void* stream) {
    if (!n_valid_dev) return MREC_EINVAL;
    return route_slots_impl<int64_t>(ids, wts, n, n_shards, cap, hashed, chunk_rot, req, slot_of_pos, pos_of_slot, overflow_dev, ws, ws_bytes, stream,
                                     n_valid_dev);
}
MREC_API int mrec_shard_unpack_req(const void* req, int32_t id_bytes, int64_t n_slots, void* ids_out, float* wts_out, void* stream) {
    if (n_slots < 0 || (id_bytes != 4 && id_bytes != 8)) return MREC_EINVAL;
    if (n_slots == 0) return MREC_OK;
    if (!req || !ids_out || !wts_out) return MREC_EINVAL;
    const unsigned grid = (unsigned)mrec_cdiv(n_slots, 256);
    if (id_bytes == 4) k_unpack_req<int32_t><<<grid, 256, 0, (hipStream_t)stream>>>((const ReqEntry<int32_t>*)req, n_slots, (int32_t*)ids_out, wts_out);
    else k_unpack_req<int64_t><<<grid, 256, 0, (hipStream_t)stream>>>((const ReqEntry<int64_t>*)req, n_slots, (int64_t*)ids_out, wts_out);
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}
MREC_API int mrec_shard_unroute_slots(const float* back, int64_t W, const int32_t* slot_of_pos, int64_t n, int32_t Dw, float* emb_out,
                                      float* wprod_out, void* stream) {
    if (n < 0 || Dw <= 0 || Dw % 4 || Dw > 252 || W < Dw + 4 || W % 4) return MREC_EINVAL;
    if (n == 0) return MREC_OK;
    if (!back || !slot_of_pos || !emb_out || !wprod_out) return MREC_EINVAL;
    if ((((uintptr_t)back) | ((uintptr_t)emb_out)) & 15 || (((uintptr_t)wprod_out) & 7)) return MREC_EINVAL;
    const int lpr = Dw / 4 + 1, G = 64 / lpr;
    k_unroute_slots<<<(unsigned)mrec_cdiv(n, (int64_t)4 * G * PB), 256, 0, (hipStream_t)stream>>>(back, W, slot_of_pos, n, Dw, lpr, G,
                                                                                                emb_out, wprod_out);
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}
MREC_API int mrec_shard_route_grads(const float* g, int64_t ldg, const float* dlogit, int32_t F, const int32_t* pos_of_slot,
                                    int64_t n_slots, int32_t Dw, float* msg, int64_t W, void* stream) {
    if (n_slots < 0 || Dw <= 0 || Dw % 4 || Dw > 252 || W < Dw + 4 || W % 4 || ldg < Dw || ldg % 4 || F <= 0) return MREC_EINVAL;
    if (n_slots == 0) return MREC_OK;
    if (!g || !dlogit || !pos_of_slot || !msg) return MREC_EINVAL;
    if ((((uintptr_t)g) | ((uintptr_t)msg)) & 15) return MREC_EINVAL;
    if ((uint64_t)n_slots * (uint64_t)F >= ((uint64_t)1 << 32)) return MREC_EUNSUPPORTED;
    const unsigned magic = (unsigned)(((uint64_t)1 << 32) / (uint64_t)F + 1);       // pos / F = umulhi(pos, magic) while pos * F < 2^32
    const int lpr = Dw / 4 + 1, G = 64 / lpr;
    k_route_grads<<<(unsigned)mrec_cdiv(n_slots, (int64_t)4 * G * PB), 256, 0, (hipStream_t)stream>>>(g, ldg, dlogit, magic, F, pos_of_slot,
                                                                                                    n_slots, Dw, lpr, G, msg, W);
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}
