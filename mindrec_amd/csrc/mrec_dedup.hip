// mrec_dedup.hip -- ops.Unique (first-occurrence order) and its inverted index, for gfx950.
//
// Replaces the MindSpore `Unique` primitive invoked at mindspore_rec/ops/embedding.py:153,192 and
// models/wide_deep/src/wide_and_deep.py:212, plus the index half of the optimizer-side RowTensor
// dedup (Unique + UnsortedSegmentSum).
//
// Dedup: a scratch open-addressing table whose slots hold POSITIONS, not keys (slot value j means
// "key ids[j]"), claimed by CAS and lowered by atomicMin, so that after the insert kernel every
// slot holds the first occurrence of its key whatever the race order -- the result is a pure
// function of the input.  No key value needs to be reserved and int64 keys need only 32-bit
// atomics.  A flag + two-level scan then numbers the first occurrences in position order.
//
// Group-by: stable LSD radix sort of positions by their group number (11-bit digits; 2 passes up
// to 4 M ids).  In-tile ranks come from wave64 ballot matching: ten to eleven __ballot()s give every
// lane the mask of lanes sharing its digit, popcount of the lower lanes is its rank.
#include "mrec_common.h"
#include "mrec_radix.h"

namespace {

constexpr int kEmpty = 0x7f7f7f7f;
constexpr int DB = 256;        // threads per block
constexpr int DI = 8;          // ids per thread in the flag/scan kernels
constexpr int DT = DB * DI;    // ids per tile

constexpr int II = 4;            // ids per thread in the insert kernel
constexpr int IT = DB * II;      // ids per insert tile
constexpr int LT = 2 * IT;       // slots of the per-tile LDS table (load factor <= 0.5)

// Insert, two levels.  (1) The block deduplicates its tile of IT ids in an LDS table with the same
// position-valued CAS + min protocol, so each distinct key of the tile gets one leader: its first
// occurrence in the tile.  (2) Only leaders touch the global table.  Without (1) an id with c
// copies costs c same-word atomics, which the memory side serialises at ~10.6 ns each (measured:
// +174 us for one id with 16384 copies); with it the cost is one atomic per tile holding the id.
// The global minimum over tile-leaders is the global first occurrence, so the result is unchanged.
// 12 KB of LDS per block (int32 keys) keeps the latency-bound global phase at full occupancy.
template <class K>
__global__ __launch_bounds__(DB) void k_dedup_insert(const K* __restrict__ ids, int n, int* slots,
                                                     uint32_t mask, int* __restrict__ sidx, bool skipneg = false) {
    __shared__ int ltab[LT];       // phase 1: local position of the tile-first occurrence of the slot's key
                                   // phase 2: the global slot its leader found
    __shared__ K lkey[IT];
    const int base = blockIdx.x * IT;
    for (int j = threadIdx.x; j < LT; j += DB) ltab[j] = kEmpty;
    K key[II];
    uint32_t hsh[II];
    int ls[II];
    bool lead[II];
#pragma unroll
    for (int k = 0; k < II; ++k) {
        const int li = k * DB + threadIdx.x;          // coalesced; local order == global order
        const int i = base + li;
        key[k] = (i < n) ? ids[i] : K(0);
        hsh[k] = mrec_hash_key(key[k]);
        lkey[li] = key[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < II; ++k) {
        const int li = k * DB + threadIdx.x;
        ls[k] = -1;
        if (base + li >= n || (skipneg && key[k] < 0)) continue;       // (skipneg: negative ids are nobody's key -- padding)
        uint32_t s = (hsh[k] >> 11) & (LT - 1);       // different bits than the global probe start
        for (;;) {
            int cur = *(volatile int*)&ltab[s];
            if (cur == kEmpty) {
                const int old = atomicCAS(&ltab[s], kEmpty, li);
                if (old == kEmpty) break;
                cur = old;
            }
            if (lkey[cur] == key[k]) {
                if (cur > li) atomicMin(&ltab[s], li);
                break;
            }
            s = (s + 1) & (LT - 1);
        }
        ls[k] = (int)s;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < II; ++k) lead[k] = ls[k] >= 0 && ltab[ls[k]] == k * DB + (int)threadIdx.x;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < II; ++k) {
        if (!lead[k]) continue;                       // not the tile's first occurrence of this key
        const int i = base + k * DB + threadIdx.x;
        uint32_t s = hsh[k] & mask;
        for (;;) {
            // (Issuing a thread's II first claims together, ahead of the walks, was measured: plan alone 102.7 -> 108.3 us,
            // the step 0.630 -> 0.639 ms with the lookup beside it 2 us slower -- bursts of atomics; not kept.)
            // CAS first, no read-before-claim: tile leaders are mostly distinct keys, so the common case
            // is an empty slot and one memory operation.  A slot only moves EMPTY -> position -> smaller
            // position of the SAME key, so the returned value is always a valid position to compare.
            const int cur = atomicCAS(&slots[s], kEmpty, i);
            if (cur == kEmpty) break;
            if (ids[cur] == key[k]) {
                if (cur > i) atomicMin(&slots[s], i);
                break;
            }
            s = (s + 1) & mask;
        }
        ltab[ls[k]] = (int)s;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < II; ++k) {
        const int li = k * DB + threadIdx.x;
        if (ls[k] >= 0) sidx[base + li] = ltab[ls[k]];
        else if (skipneg && base + li < n) sidx[base + li] = -1;
    }
}

// (Round 4, measured and not kept: finding the first positions WITHOUT device-scope atomics -- a counting split of (key, position)
// pairs into ~n / 1536 buckets by hash (histogram, column scan, scatter; LDS atomics only), then one workgroup per bucket with an
// LDS table (slot = position << 32 | entry, 64-bit LDS min).  Bit-identical results, but the plan alone went from 103 to 165 us
// on uniform ids x 26 fields and from 109 to 344 us on Zipf ids x 39 fields: four launches of which the single-block scan and
// the scattered 4- and 8-byte pair writes cost more than the table's atomics, and a hot id puts all its copies in ONE bucket
// -- 16384 LDS atomics on one slot by one workgroup.)
//
// Decoupled look-back over the tiles' first-occurrence counts (one status word per tile: flag << 30 | value; flag 1 = the
// tile's own count, 2 = inclusive prefix).  Tiles are dispatched in index order and only ever wait for lower-numbered
// tiles, so the wait always ends.  The words are zero when the kernel starts: the NEXT kernel of the chain clears them again
// (k_radix_hist / k_dedup_inv), which is what keeps a primed workspace primed.
constexpr unsigned kFlagA = 1u << 30, kFlagP = 2u << 30, kFlagMask = 3u << 30;

__device__ __forceinline__ int lookback_excl(unsigned* status, int tile) {
    const int l = lane_id();
    int excl = 0;
    for (int base = tile - 1;; base -= 64) {
        const int idx = base - l;
        unsigned st;
        do {
            st = (idx >= 0) ? __hip_atomic_load(&status[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kFlagP;
        } while (__any((st & kFlagMask) == 0));
        const uint64_t pm = __ballot((st & kFlagMask) == kFlagP);
        const int firstp = pm ? __ffsll((long long)pm) - 1 : 63; // nearest predecessor with an inclusive prefix (the virtual
        int c = (l <= firstp) ? (int)(st & ~kFlagMask) : 0;      // tiles below 0 carry one); none in this round: add all 64
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
        excl += c;
        if (pm) break;
    }
    return excl;
}

// Numbers the first occurrences in position order in ONE pass (flag + count + scan + rank).  The scratch table is left as
// it is -- a slot keeps the first position of its key, which is how a duplicate finds its group afterwards
// (inv[slots[sidx[j]]]) -- and is cleared by a 4-MB memset at the head of the next call: handing the slots back one by
// one, and keeping a rank per slot for the duplicates, were 850 k random 4-byte writes per call, a third of this kernel.
template <class K>
__global__ __launch_bounds__(DB) void k_dedup_rank(const K* __restrict__ ids, const int* __restrict__ slots,
                                                   const int* __restrict__ sidx, int n, unsigned* __restrict__ status,
                                                   int nblk, K* __restrict__ uniq, int* __restrict__ inv,
                                                   int64_t* __restrict__ n_uniq_dev, int* __restrict__ first_pos = nullptr,
                                                   int* __restrict__ dup_pos = nullptr, int64_t* __restrict__ n_dup_dev = nullptr,
                                                   bool skipneg = false) {
    __shared__ int sm[8];
    __shared__ int s_excl;
    const int base = blockIdx.x * DT + threadIdx.x * DI;
    bool first[DI];
    int slot[DI];
    int c = 0;
    // (a thread's DI ids, scratch-slot numbers and slot contents are each requested together, at clamped indices: as guarded
    // loads they were DI dependent round trips per step -- see k_radix_hist)
    K idv[DI];
    int held[DI];
    const int nl = n > 0 ? n - 1 : 0;
#pragma unroll
    for (int k = 0; k < DI; ++k) {
        const int i = base + k;
        slot[k] = sidx[i < nl ? i : nl];
        idv[k] = ids[i < nl ? i : nl];
    }
#pragma unroll
    for (int k = 0; k < DI; ++k) held[k] = slots[slot[k] >= 0 ? slot[k] : 0];
#pragma unroll
    for (int k = 0; k < DI; ++k) {
        const int i = base + k;
        first[k] = (i < n) && slot[k] >= 0 && (held[k] == i);      // (slot < 0: a skipped negative id -- counted with the duplicates)
        c += first[k];
    }
    int tot;
    const int pre = block_excl_scan_256(c, sm, &tot);
    if (threadIdx.x < 64) {
        const int t = blockIdx.x;
        int excl = 0;
        if (t == 0) {
            if (threadIdx.x == 0) __hip_atomic_store(&status[0], kFlagP | (unsigned)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (threadIdx.x == 0) __hip_atomic_store(&status[t], kFlagA | (unsigned)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            excl = lookback_excl(status, t);
            if (threadIdx.x == 0) __hip_atomic_store(&status[t], kFlagP | (unsigned)(excl + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (threadIdx.x == 0) s_excl = excl;
    }
    __syncthreads();
    const int tile_base = s_excl;
    int r = tile_base + pre;
#pragma unroll
    for (int k = 0; k < DI; ++k) {
        const int i = base + k;
        if (first[k]) {
            uniq[r] = idv[k];
            inv[i] = r;                      // (a duplicate finds its group here, through the slot that still holds i)
            if (first_pos) first_pos[r] = i; // the step's plan: where the group starts
            ++r;
        } else if (dup_pos && i < n) {
            // every position before i is a first occurrence or a duplicate, so the duplicates need no scan of their own:
            // i is duplicate number i - (first occurrences before i), in position order
            dup_pos[i - r] = i;
        }
    }
    if ((int)blockIdx.x == nblk - 1 && threadIdx.x == 0) {
        *n_uniq_dev = (int64_t)tile_base + tot;
        if (n_dup_dev) *n_dup_dev = (int64_t)n - ((int64_t)tile_base + tot);
        if (skipneg && tile_base + tot < n) uniq[tile_base + tot] = (K)-1;     // the row of the skipped ids' pseudo-group: none
    }
}

// The inverted index from the sorted duplicates.  First occurrences arrive already in group order (group g starts at
// first_pos[g]), so only the duplicate positions were sorted by group (stable: ascending position inside a group); with
// lb(g) = number of duplicates of groups < g (a binary search in the sorted duplicate keys) group g's run starts at
// g + lb(g), holds its first occurrence and then its duplicates.  Thread t places group t and duplicate t.
__global__ __launch_bounds__(256) void k_plan_place(const int* __restrict__ first_pos, const int64_t* __restrict__ n_uniq_dev,
                                                    const int* __restrict__ dkey, const int* __restrict__ dpos,
                                                    const int64_t* __restrict__ n_dup_dev, int n, int* __restrict__ sorted_pos,
                                                    int* __restrict__ sorted_seg, int* __restrict__ seg_offsets, int gkey = -1,
                                                    int64_t* __restrict__ n_valid_dev = nullptr) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int U = (int)*n_uniq_dev, nD = (int)*n_dup_dev;
    auto lower = [&](int g) {            // first e in [0, nD) with dkey[e] >= g
        int lo = 0, hi = nD;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (dkey[mid] < g) lo = mid + 1; else hi = mid;
        }
        return lo;
    };
    // gkey >= 0: "duplicates" carrying this key are the skipped negative ids (key n - 1: above every real group, which are
    // < U <= n - 1 whenever a skipped id exists).  They sort behind the real duplicates and are placed behind every group's
    // run, as a pseudo-group U that nobody reads: the index proper is its first n_valid = U + (real duplicates) entries.
    const int nDr = (gkey >= 0) ? lower(gkey) : nD;
    if (t < U) {
        const int off = t + lower(t);
        seg_offsets[t] = off;
        sorted_pos[off] = first_pos[t];
        sorted_seg[off] = t;
        if (t == U - 1) {
            seg_offsets[U] = U + nDr;
            if (nDr < nD) seg_offsets[U + 1] = n;
        }
    }
    if (t < nD) {
        const int g = dkey[t];
        if (t >= nDr) {
            sorted_pos[U + t] = dpos[t];
            sorted_seg[U + t] = U;
        } else {
            const int lb = lower(g);
            const int dst = g + lb + 1 + (t - lb);
            sorted_pos[dst] = dpos[t];
            sorted_seg[dst] = g;
        }
    }
    if (t == 0) {
        if (U == 0) { seg_offsets[0] = 0; if (nDr < nD) seg_offsets[1] = n; }
        if (n_valid_dev) *n_valid_dev = (int64_t)U + nDr;
    }
}

__global__ __launch_bounds__(DB) void k_dedup_inv(const int* __restrict__ slots, const int* __restrict__ sidx,
                                                  int n, int* __restrict__ inv, unsigned* __restrict__ status, int nstatus) {
    const int i = blockIdx.x * DB + threadIdx.x;
    if (i < nstatus) status[i] = 0;                   // the look-back words of k_dedup_rank, for the next call
    if (i < n) {
        const int p = slots[sidx[i]];                 // first position of i's key (its inverse is final: written by k_dedup_rank)
        if (p != i) inv[i] = inv[p];
    }
}

// Few duplicates (uniform ids over a large vocabulary: ~n^2 / 2V of them): ONE workgroup resolves their groups and sorts them
// by (group, position) by counting, in LDS, and tells the radix passes behind it that there is nothing left to do
// (*n_radix = 0) -- a radix pass over one live tile still costs its three launches' single-workgroup latencies (~25 us).
constexpr int kSmallDups = 1024;
__global__ __launch_bounds__(1024) void k_plan_small(const int* __restrict__ slots, const int* __restrict__ sidx,
                                                     const int* __restrict__ dup_pos, const int64_t* __restrict__ n_dup_dev,
                                                     int* __restrict__ inv, int* __restrict__ dkey, int* __restrict__ dpos,
                                                     int64_t* __restrict__ n_radix, int gkey = -1) {
    __shared__ int skey[kSmallDups];
    const int64_t nD64 = *n_dup_dev;
    if (nD64 > kSmallDups) {
        if (threadIdx.x == 0) *n_radix = nD64;
        return;
    }
    const int nD = (int)nD64, t = threadIdx.x;
    if (t == 0) *n_radix = 0;
    int key = 0, pos = 0;
    if (t < nD) {
        pos = dup_pos[t];
        const int sx = sidx[pos];
        key = sx >= 0 ? inv[slots[sx]] : gkey;           // (a skipped negative id: the pseudo-group behind all others)
        inv[pos] = sx >= 0 ? key : -1;
        skey[t] = key;
    }
    __syncthreads();
    if (t < nD) {
        int rank = 0;                                  // duplicates with a smaller group, or the same group and an earlier position
        for (int e = 0; e < nD; ++e) {
            const int k = skey[e];
            rank += (k < key) || (k == key && e < t);
        }
        dkey[rank] = key;
        dpos[rank] = pos;
    }
}

__global__ void k_set_i64(int64_t* p, int64_t v) { *p = v; }

struct DedupScratch { int* slots; int* sidx; unsigned* status; int nstatus; int* first_pos; int* dup_pos; int64_t* n_dup; int64_t* n_radix; size_t slot_bytes; };

// fuse_inv_out != nullptr (the step's plan): the rank kernel also separates first occurrences (first_pos) from duplicates
// (dup_pos, their count on the device), the inverse of the duplicates is left to the caller's sort of the duplicates.
// primed: the caller vouches that the last thing that wrote this workspace was a completed call of this function with the
// same n: the look-back words are then already zero -- and, for the step's plan (fuse_inv_out), so is the scratch table, which
// plan_impl clears BEHIND its last kernel instead of in front of the first (a 4-MB memset: 10 us by which the insert kernel now
// starts earlier; the chain's end is far off the step's critical path, its head runs beside the lookup).
template <class K>
int dedup_impl(const K* ids, int64_t n, K* uniq, int32_t* inv, int64_t* n_uniq_dev, void* ws, size_t ws_bytes,
               void* stream_v, DedupScratch* fuse_inv_out = nullptr, bool primed = false, bool skipneg = false) {
    hipStream_t st = (hipStream_t)stream_v;
    if (n < 0 || !n_uniq_dev) return MREC_EINVAL;
    if (n == 0) {
        k_set_i64<<<1, 1, 0, st>>>(n_uniq_dev, 0);
        MREC_LAUNCH_CHECK();
        return MREC_OK;
    }
    if (!ids || !uniq || !inv || !ws) return MREC_EINVAL;
    if (n >= (int64_t(1) << 30)) return MREC_EUNSUPPORTED;
    uint64_t cap = 1024;
    while (cap < (uint64_t)n * 2) cap <<= 1;
    const int nblk = (int)mrec_cdiv(n, DT);
    MrecArena a(ws, ws_bytes);
    int* slots = a.take<int>(cap);
    int* sidx = a.take<int>(n);
    unsigned* status = (unsigned*)a.take<int>(nblk);
    int* first_pos = nullptr; int* dup_pos = nullptr; int64_t* n_dup = nullptr;
    if (fuse_inv_out) {                    // the step's plan: first occurrences and duplicates leave the rank kernel separated
        first_pos = a.take<int>(n);
        dup_pos = a.take<int>(n);
        n_dup = a.take<int64_t>(2);        // [0] duplicates, [1] duplicates left to the radix passes
    }
    if (!a.ok) return MREC_EWORKSPACE;
    if (!(primed && fuse_inv_out)) MREC_HIP_CHECK(hipMemsetAsync(slots, 0x7f, cap * sizeof(int), st));
    if (!primed) MREC_HIP_CHECK(hipMemsetAsync(status, 0, (size_t)nblk * sizeof(int), st));     // (a completed call leaves them zero)
    const int g256 = (int)mrec_cdiv(n, DB);
    k_dedup_insert<K><<<(int)mrec_cdiv(n, IT), DB, 0, st>>>(ids, (int)n, slots, (uint32_t)(cap - 1), sidx, skipneg);
    k_dedup_rank<K><<<nblk, DB, 0, st>>>(ids, slots, sidx, (int)n, status, nblk, uniq, inv, n_uniq_dev, first_pos, dup_pos, n_dup, skipneg);
    if (fuse_inv_out) {
        fuse_inv_out->slots = slots; fuse_inv_out->sidx = sidx; fuse_inv_out->status = status; fuse_inv_out->nstatus = nblk;
        fuse_inv_out->first_pos = first_pos; fuse_inv_out->dup_pos = dup_pos; fuse_inv_out->n_dup = n_dup;
        fuse_inv_out->n_radix = n_dup ? n_dup + 1 : nullptr;
        fuse_inv_out->slot_bytes = cap * sizeof(int);
    } else k_dedup_inv<<<g256, DB, 0, st>>>(slots, sidx, (int)n, inv, status, nblk);
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}

__global__ __launch_bounds__(256) void k_seg_offsets(const int* __restrict__ sseg, int n,
                                                     int* __restrict__ seg_offsets) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const int s = sseg[e];
    if (e == 0 || sseg[e - 1] != s) seg_offsets[s] = e;
    if (e == n - 1) seg_offsets[s + 1] = n;
}

}  // namespace

MREC_API int mrec_dedup_workspace_bytes(int64_t n, size_t* out) {
    if (!out || n < 0) return MREC_EINVAL;
    if (n >= (int64_t(1) << 30)) return MREC_EUNSUPPORTED;
    uint64_t cap = 1024;
    while (cap < (uint64_t)n * 2) cap <<= 1;
    size_t b = 0;
    b += mrec_align_up(cap * 4, 256) * 2;
    b += mrec_align_up((size_t)(n ? n : 1) * 4, 256);
    b += mrec_align_up((size_t)(mrec_cdiv(n ? n : 1, DT)) * 4, 256);
    *out = b;
    return MREC_OK;
}

MREC_API int mrec_dedup_i32(const int32_t* ids, int64_t n, int32_t* uniq, int32_t* inv, int64_t* n_uniq_dev,
                            void* ws, size_t ws_bytes, void* stream) {
    return dedup_impl<int32_t>(ids, n, uniq, inv, n_uniq_dev, ws, ws_bytes, stream);
}

MREC_API int mrec_dedup_i64(const int64_t* ids, int64_t n, int64_t* uniq, int32_t* inv, int64_t* n_uniq_dev,
                            void* ws, size_t ws_bytes, void* stream) {
    return dedup_impl<int64_t>(ids, n, uniq, inv, n_uniq_dev, ws, ws_bytes, stream);
}

MREC_API int mrec_group_workspace_bytes(int64_t n, size_t* out) {
    if (!out || n < 0) return MREC_EINVAL;
    if (n > (int64_t(1) << 30)) return MREC_EUNSUPPORTED;
    const size_t nn = (size_t)(n ? n : 1);
    size_t b = 0;
    b += mrec_align_up((size_t)mrec_cdiv(nn, RT) * RNB * 4, 256) * 2;
    b += mrec_align_up((size_t)RNB * 4, 256);
    b += mrec_align_up(nn * 4, 256) * 2;
    *out = b;
    return MREC_OK;
}

static int group_impl(const int32_t* inv, int64_t n, int32_t* sorted_pos, int32_t* sorted_seg, int32_t* seg_offsets,
                      void* ws, size_t ws_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (n < 0 || !seg_offsets) return MREC_EINVAL;
    if (n == 0) {
        MREC_HIP_CHECK(hipMemsetAsync(seg_offsets, 0, sizeof(int32_t), st));
        return MREC_OK;
    }
    if (!inv || !sorted_pos || !sorted_seg || !ws) return MREC_EINVAL;
    if (n > (int64_t(1) << 30)) return MREC_EUNSUPPORTED;
    const int nblk = (int)mrec_cdiv(n, RT);
    MrecArena a(ws, ws_bytes);
    int* hist = a.take<int>((size_t)nblk * RNB);
    int* hscan = a.take<int>((size_t)nblk * RNB);
    int* totals = a.take<int>(RNB);
    int* tk = a.take<int>(n);
    int* tv = a.take<int>(n);
    if (!a.ok) return MREC_EWORKSPACE;
    int bits = 1;
    while (((int64_t)1 << bits) < n) ++bits;  // group numbers are < n
    const int passes = (bits + RMAXB - 1) / RMAXB;
    const int pbits = (bits + passes - 1) / passes;
    const int* kin = inv;
    const int* vin = nullptr;
    for (int p = 0; p < passes; ++p) {
        const bool to_user = ((passes - 1 - p) % 2) == 0;
        int* kout = to_user ? sorted_seg : tk;
        int* vout = to_user ? sorted_pos : tv;
        radix_pass(kin, vin, (int)n, p * pbits, pbits, hist, hscan, totals, nullptr, kout, vout, st);
        kin = kout;
        vin = vout;
    }
    k_seg_offsets<<<(int)mrec_cdiv(n, 256), 256, 0, st>>>(sorted_seg, (int)n, seg_offsets);
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}

MREC_API int mrec_group_by_inverse(const int32_t* inv, int64_t n, int32_t* sorted_pos, int32_t* sorted_seg,
                                   int32_t* seg_offsets, void* ws, size_t ws_bytes, void* stream) {
    return group_impl(inv, n, sorted_pos, sorted_seg, seg_offsets, ws, ws_bytes, stream);
}

// Unique + inverted index in one call (what a training step needs): the inverse rides on the first
// radix histogram, saving a pass and a launch over mrec_dedup_* followed by mrec_group_by_inverse.
static size_t plan_extra_bytes(int64_t n) { return mrec_align_up((size_t)(n ? n : 1) * 4, 256) * 2 + 256; }

// Unique + inverted index in one call (what a training step needs).  First occurrences leave the rank kernel already in
// group order; only the DUPLICATE positions are sorted by group (their count lives on the device: the radix kernels run
// over the live tiles only -- a handful on uniform ids), and one placement kernel writes the inverted index.
template <class K>
static int plan_impl(const K* ids, int64_t n, K* uniq, int32_t* inv, int64_t* n_uniq_dev, int32_t* sorted_pos,
                     int32_t* sorted_seg, int32_t* seg_offsets, void* ws, size_t ws_bytes, void* stream, bool primed = false,
                     bool skipneg = false) {
    if (n < 0 || !n_uniq_dev || !seg_offsets) return MREC_EINVAL;
    // skipneg: negative ids are padding (a shard's fixed-capacity request message): no group, no entry of the index proper;
    // n_uniq_dev is then two words, [1] = the number of entries of the index proper (valid positions)
    const int gkey = skipneg ? (int)(n - 1) : -1;
    size_t db = 0, gb = 0;
    int rc = mrec_dedup_workspace_bytes(n, &db);
    if (rc != MREC_OK) return rc;
    db += plan_extra_bytes(n);
    rc = mrec_group_workspace_bytes(n, &gb);
    if (rc != MREC_OK) return rc;
    if (n > 0 && (!ws || ws_bytes < db + gb)) return MREC_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) {
        rc = dedup_impl<K>(ids, n, uniq, inv, n_uniq_dev, ws, db, stream, nullptr, primed);
        if (rc != MREC_OK) return rc;
        MREC_HIP_CHECK(hipMemsetAsync(seg_offsets, 0, sizeof(int32_t), st));
        if (skipneg) MREC_HIP_CHECK(hipMemsetAsync(n_uniq_dev + 1, 0, sizeof(int64_t), st));
        return MREC_OK;
    }
    if (!sorted_pos || !sorted_seg) return MREC_EINVAL;
    DedupScratch sc{};
    rc = dedup_impl<K>(ids, n, uniq, inv, n_uniq_dev, ws, db, stream, &sc, primed, skipneg);
    if (rc != MREC_OK) return rc;
    const int nblk = (int)mrec_cdiv(n, RT);
    MrecArena a((char*)ws + db, gb);
    int* hist = a.take<int>((size_t)nblk * RNB);
    int* hscan = a.take<int>((size_t)nblk * RNB);
    int* totals = a.take<int>(RNB);
    int* tk = a.take<int>(n);
    int* tv = a.take<int>(n);
    if (!a.ok) return MREC_EWORKSPACE;
    int bits = 1;
    while (((int64_t)1 << bits) < n) ++bits;  // group numbers are < n
    const int passes = (bits + RMAXB - 1) / RMAXB;
    const int pbits = (bits + passes - 1) / passes;
    // ping-pong between A = the caller's (sorted_seg, sorted_pos), free until the placement kernel writes them, and
    // B = (tk, tv); the last pass lands in B.  The generated keys of pass 0 live in the key array pass 0 does not write.
    const bool first_to_b = ((passes - 1) % 2) == 0;
    int* kgen = first_to_b ? sorted_seg : tk;
    // few duplicates: one workgroup sorts them straight into B and zeroes the count the radix passes run on
    k_plan_small<<<1, 1024, 0, st>>>(sc.slots, sc.sidx, sc.dup_pos, sc.n_dup, inv, tk, tv, sc.n_radix, gkey);
    const int* kin = kgen;
    const int* vin = sc.dup_pos;
    for (int p = 0; p < passes; ++p) {
        const bool to_b = ((passes - 1 - p) % 2) == 0;
        int* kout = to_b ? tk : sorted_seg;
        int* vout = to_b ? tv : sorted_pos;
        const int shift = p * pbits;
        if (p == 0)      // keys = group of every duplicate = the rank at its scratch slot; also its entry of `inv`
            radix_pass(kin, vin, (int)n, shift, pbits, hist, hscan, totals, nullptr, kout, vout, st, sc.slots, sc.sidx, kgen,
                       sc.status, sc.nstatus, sc.dup_pos, inv, sc.n_radix, gkey);
        else
            radix_pass(kin, vin, (int)n, shift, pbits, hist, hscan, totals, nullptr, kout, vout, st, nullptr, nullptr, nullptr,
                       nullptr, 0, nullptr, nullptr, sc.n_radix);
        kin = kout;
        vin = vout;
    }
    k_plan_place<<<(unsigned)mrec_cdiv(n, 256), 256, 0, st>>>(sc.first_pos, n_uniq_dev, tk, tv, sc.n_dup, (int)n, sorted_pos,
                                                             sorted_seg, seg_offsets, gkey, skipneg ? n_uniq_dev + 1 : nullptr);
    MREC_LAUNCH_CHECK();
    MREC_HIP_CHECK(hipMemsetAsync(sc.slots, 0x7f, sc.slot_bytes, st));       // hand the scratch table back clean (MREC_PLAN_WS_PRIMED)
    return MREC_OK;
}

MREC_API int mrec_sparse_plan_workspace_bytes(int64_t n, size_t* out) {
    size_t db = 0, gb = 0;
    int rc = mrec_dedup_workspace_bytes(n, &db);
    if (rc != MREC_OK) return rc;
    db += plan_extra_bytes(n);
    rc = mrec_group_workspace_bytes(n, &gb);
    if (rc != MREC_OK) return rc;
    *out = db + gb;
    return MREC_OK;
}
MREC_API int mrec_sparse_plan_i32(const int32_t* ids, int64_t n, int32_t* uniq, int32_t* inv, int64_t* n_uniq_dev,
                                  int32_t* sorted_pos, int32_t* sorted_seg, int32_t* seg_offsets, void* ws,
                                  size_t ws_bytes, void* stream) {
    return plan_impl<int32_t>(ids, n, uniq, inv, n_uniq_dev, sorted_pos, sorted_seg, seg_offsets, ws, ws_bytes, stream);
}
MREC_API int mrec_sparse_plan_i64(const int64_t* ids, int64_t n, int64_t* uniq, int32_t* inv, int64_t* n_uniq_dev,
                                  int32_t* sorted_pos, int32_t* sorted_seg, int32_t* seg_offsets, void* ws,
                                  size_t ws_bytes, void* stream) {
    return plan_impl<int64_t>(ids, n, uniq, inv, n_uniq_dev, sorted_pos, sorted_seg, seg_offsets, ws, ws_bytes, stream);
}

// flags & MREC_PLAN_WS_PRIMED: skip the two memsets (see include/mrec.h)
MREC_API int mrec_sparse_plan_ex_i32(const int32_t* ids, int64_t n, int32_t* uniq, int32_t* inv, int64_t* n_uniq_dev,
                                     int32_t* sorted_pos, int32_t* sorted_seg, int32_t* seg_offsets, void* ws,
                                     size_t ws_bytes, uint32_t flags, void* stream) {
    return plan_impl<int32_t>(ids, n, uniq, inv, n_uniq_dev, sorted_pos, sorted_seg, seg_offsets, ws, ws_bytes, stream,
                              (flags & MREC_PLAN_WS_PRIMED) != 0, (flags & MREC_PLAN_SKIP_NEGATIVE) != 0);
}
MREC_API int mrec_sparse_plan_ex_i64(const int64_t* ids, int64_t n, int64_t* uniq, int32_t* inv, int64_t* n_uniq_dev,
                                     int32_t* sorted_pos, int32_t* sorted_seg, int32_t* seg_offsets, void* ws,
                                     size_t ws_bytes, uint32_t flags, void* stream) {
    return plan_impl<int64_t>(ids, n, uniq, inv, n_uniq_dev, sorted_pos, sorted_seg, seg_offsets, ws, ws_bytes, stream,
                              (flags & MREC_PLAN_WS_PRIMED) != 0, (flags & MREC_PLAN_SKIP_NEGATIVE) != 0);
}
