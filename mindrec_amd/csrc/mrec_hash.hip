// mrec_hash.hip -- key index of MapParameter (mindspore.experimental), for gfx950.
//
// Reference: HashEmbeddingLookup builds MapParameter(key_dtype, float32, (D,), default_value,
// permit/evict) at mindspore_rec/ops/embedding.py:136-146 and reads it through
// MapTensorGet(insert_default_value=True) at :149,193,199; put/erase API by example at
// README.md:160-205.  MindSpore backs it with a GPU hash table holding keys AND values [EXT].
//
// MI355X layout instead: the map is a *key -> row number* index (open addressing, linear probing,
// int64 keys, 2x slots) over dense row storage [capacity, D] owned by the caller.  Values, Adam
// moments and FTRL accumulators are plain row-major tables addressed by row number, so lookups
// and updates run through the same coalesced gather / sparse-apply kernels as a dense table; only
// the index is probed per key.  Row numbers are handed out deterministically: misses are ranked
// by a prefix scan in the order they appear in `keys`, fresh rows first, then the free list.
//
// Calls take keys that are unique within the call (the Unique that precedes MapTensorGet in
// HashEmbeddingLookup.construct, embedding.py:192-193, guarantees it), so no two threads of one
// kernel ever insert the same key.
#include "mrec_common.h"
#include "mrec_rng.h"

namespace {

constexpr int HB = 256;
constexpr int HI = 8;
constexpr int HT = HB * HI;

// counters: [C_TOMB] = tombstones in the slot array, [C_REBUILD] = "rebuild the slot array now" (set by the erase
// commit, read by the two rebuild kernels behind it), [C_NREBUILD] = rebuilds so far
// [C_NLOG] = keys in the erased-keys log (incremental export)
enum { C_HWM = 0, C_LIVE = 1, C_DROPPED = 2, C_FREE = 3, C_TOMB = 4, C_REBUILD = 5, C_NREBUILD = 6, C_NLOG = 7, C_NCOUNTERS = 8 };

// one index slot: key and row side by side, so a probe step costs one 16-byte access (two arrays cost two cache lines)
struct MapSlot { int64_t key; int row; int pad; };   // row: -1 empty, -2 tombstone, >= 0 row

struct MapDev {
    MapSlot* slot;      // [S]
    int64_t* row_key;   // [C]
    uint8_t* row_live;  // [C]
    int* free_list;     // [C]
    int64_t* counters;  // [C_NCOUNTERS]
    int* hits;          // [C]  training lookups that touched the row's key since it was inserted (permit filter)
    int* last_step;     // [C]  step of the last training lookup (evict filter)
    uint8_t* dirty;     // [C]  inserted / looked up for training / put since the last incremental export
    int64_t* erased_log;  // [C]  keys erased or evicted since the last incremental export
    int64_t C;
    uint32_t mask;
};

__device__ __forceinline__ int64_t eff_n(int64_t n, const int64_t* n_dev) {
    if (!n_dev) return n;
    const int64_t nd = *n_dev;
    return nd < n ? (nd < 0 ? 0 : nd) : n;
}

__global__ __launch_bounds__(HB) void k_map_find(MapDev m, const int64_t* __restrict__ keys, int64_t n_max,
                                                 const int64_t* __restrict__ n_dev, int* __restrict__ rows_out,
                                                 int* __restrict__ slot_out, uint8_t* __restrict__ miss) {
    const int64_t i = (int64_t)blockIdx.x * HB + threadIdx.x;
    if (i >= n_max) return;
    if (i >= eff_n(n_max, n_dev)) { miss[i] = 0; rows_out[i] = -1; return; }
    const int64_t key = keys[i];
    uint32_t s = mrec_hash_key(key) & m.mask;
    int row = -1, slot = -1;
    // bounded: a slot array without an empty slot (it cannot arise while the rebuild below keeps
    // live + tombstones <= 0.7 S, but a probe must never depend on that) ends as a miss, not a hang
    for (uint32_t it = 0; it <= m.mask; ++it) {
        const int r = m.slot[s].row;
        if (r == -1) break;
        if (r >= 0 && m.slot[s].key == key) { row = r; slot = (int)s; break; }
        s = (s + 1) & m.mask;
    }
    rows_out[i] = row;
    if (slot_out) slot_out[i] = slot;
    miss[i] = (row < 0);
}

// flags -> per-tile counts
__global__ __launch_bounds__(HB) void k_flag_count(const uint8_t* __restrict__ flags, int64_t n, int invert,
                                                   int* __restrict__ blocksum) {
    __shared__ int sm[8];
    const int64_t base = (int64_t)blockIdx.x * HT + threadIdx.x * HI;
    int c = 0;
#pragma unroll
    for (int k = 0; k < HI; ++k) {
        const int64_t i = base + k;
        if (i < n) c += ((flags[i] != 0) != (invert != 0));
    }
    int tot;
    block_excl_scan_256(c, sm, &tot);
    if (threadIdx.x == 0) blocksum[blockIdx.x] = tot;
}

// exclusive rank of every flagged element (rank[i] undefined when not flagged); *total = count
__global__ __launch_bounds__(HB) void k_flag_rank(const uint8_t* __restrict__ flags, int64_t n, int invert,
                                                  const int* __restrict__ blocksum, int nblk,
                                                  int* __restrict__ rank, int64_t* __restrict__ total) {
    __shared__ int sm[8];
    int part = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += HB) part += blocksum[b];
    int tile_base;
    block_excl_scan_256(part, sm, &tile_base);
    const int64_t base = (int64_t)blockIdx.x * HT + threadIdx.x * HI;
    bool f[HI];
    int c = 0;
#pragma unroll
    for (int k = 0; k < HI; ++k) {
        const int64_t i = base + k;
        f[k] = (i < n) && ((flags[i] != 0) != (invert != 0));
        c += f[k];
    }
    int tot;
    int r = tile_base + block_excl_scan_256(c, sm, &tot);
#pragma unroll
    for (int k = 0; k < HI; ++k)
        if (f[k]) rank[base + k] = r++;
    if ((int)blockIdx.x == nblk - 1 && threadIdx.x == 0) *total = (int64_t)tile_base + tot;
}

__global__ __launch_bounds__(HB) void k_map_insert(MapDev m, const int64_t* __restrict__ keys, int64_t n,
                                                   const uint8_t* __restrict__ miss, const int* __restrict__ rank,
                                                   int* __restrict__ rows_out, uint8_t* __restrict__ is_new,
                                                   int64_t* __restrict__ n_dropped_call, int64_t* __restrict__ n_tomb_reused) {
    const int64_t i = (int64_t)blockIdx.x * HB + threadIdx.x;
    if (i >= n) return;
    if (!miss[i]) { if (is_new) is_new[i] = 0; return; }
    const int64_t hwm = m.counters[C_HWM];
    const int64_t nfree = m.counters[C_FREE];
    const int64_t fresh = m.C - hwm;
    const int64_t r = rank[i];
    int row;
    if (r < fresh) row = (int)(hwm + r);
    else if (r - fresh < nfree) row = m.free_list[nfree - 1 - (r - fresh)];
    else {
        rows_out[i] = -1;
        if (is_new) is_new[i] = 0;
        atomicAdd((unsigned long long*)n_dropped_call, 1ull);
        return;
    }
    const int64_t key = keys[i];
    uint32_t s = mrec_hash_key(key) & m.mask;
    bool placed = false;
    // live keys <= C <= S / 2, so a negative slot exists; bounded all the same
    for (uint32_t it = 0; it <= m.mask; ++it) {
        const int cur = __hip_atomic_load(&m.slot[s].row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur < 0) {
            const int old = atomicCAS(&m.slot[s].row, cur, row);
            if (old == cur) {
                if (cur == -2) atomicAdd((unsigned long long*)n_tomb_reused, 1ull);
                placed = true;
                break;
            }
        }
        s = (s + 1) & m.mask;
    }
    if (!placed) {
        rows_out[i] = -1;
        if (is_new) is_new[i] = 0;
        atomicAdd((unsigned long long*)n_dropped_call, 1ull);
        return;
    }
    m.slot[s].key = key;
    m.row_key[row] = key;
    m.row_live[row] = 1;
    rows_out[i] = row;
    if (is_new) is_new[i] = 1;
}

__global__ void k_map_commit_insert(MapDev m, const int64_t* n_miss, const int64_t* n_dropped_call,
                                    const int64_t* n_tomb_reused) {
    const int64_t M = *n_miss, dropped = *n_dropped_call;
    m.counters[C_TOMB] -= *n_tomb_reused;
    const int64_t hwm = m.counters[C_HWM], nfree = m.counters[C_FREE];
    const int64_t fresh = m.C - hwm;
    const int64_t use_fresh = M < fresh ? M : fresh;
    int64_t use_free = M - use_fresh - dropped;
    if (use_free < 0) use_free = 0;
    m.counters[C_HWM] = hwm + use_fresh;
    m.counters[C_FREE] = nfree - use_free;
    m.counters[C_LIVE] += use_fresh + use_free;
    m.counters[C_DROPPED] += dropped;
}

__global__ __launch_bounds__(HB) void k_map_erase(MapDev m, int64_t n, const int* __restrict__ rows,
                                                  const int* __restrict__ slots, const uint8_t* __restrict__ miss,
                                                  const int* __restrict__ rank) {
    const int64_t i = (int64_t)blockIdx.x * HB + threadIdx.x;
    if (i >= n || miss[i]) return;
    const int64_t nfree = m.counters[C_FREE], nlog = m.counters[C_NLOG];
    m.slot[slots[i]].row = -2;
    m.row_live[rows[i]] = 0;
    m.free_list[nfree + rank[i]] = rows[i];
    if (nlog + rank[i] < m.C) m.erased_log[nlog + rank[i]] = m.row_key[rows[i]];
}

// Erase leaves tombstones, and an insert only takes one back when its probe happens to pass it: under
// insert / erase churn the EMPTY slots (the only thing that ends a miss probe) would run out.  Once
// tombstones exceed S / 5 (live <= S / 2, so live + tombstones <= 0.7 S always holds) the slot array is
// rebuilt from the row side (row_key / row_live), which empties every tombstone.
__global__ void k_map_commit_erase(MapDev m, const int64_t* n_found, int64_t S) {
    m.counters[C_FREE] += *n_found;
    m.counters[C_LIVE] -= *n_found;
    const int64_t nlog = m.counters[C_NLOG] + *n_found;
    m.counters[C_NLOG] = nlog < m.C ? nlog : m.C;
    const int64_t tomb = m.counters[C_TOMB] + *n_found;
    const bool rebuild = tomb * 5 > S;
    m.counters[C_REBUILD] = rebuild ? 1 : 0;
    m.counters[C_TOMB] = rebuild ? 0 : tomb;
    if (rebuild) m.counters[C_NREBUILD] += 1;
}

__global__ __launch_bounds__(HB) void k_map_rebuild_clear(MapDev m, int64_t S) {
    if (!m.counters[C_REBUILD]) return;
    for (int64_t s = (int64_t)blockIdx.x * HB + threadIdx.x; s < S; s += (int64_t)gridDim.x * HB) m.slot[s].row = -1;
}

__global__ __launch_bounds__(HB) void k_map_rebuild_insert(MapDev m) {
    if (!m.counters[C_REBUILD]) return;
    const int64_t hwm = m.counters[C_HWM];
    for (int64_t r = (int64_t)blockIdx.x * HB + threadIdx.x; r < hwm; r += (int64_t)gridDim.x * HB) {
        if (!m.row_live[r]) continue;
        const int64_t key = m.row_key[r];
        uint32_t s = mrec_hash_key(key) & m.mask;
        for (uint32_t it = 0; it <= m.mask; ++it) {
            if (atomicCAS(&m.slot[s].row, -1, (int)r) == -1) { m.slot[s].key = key; break; }
            s = (s + 1) & m.mask;
        }
    }
}

__global__ __launch_bounds__(HB) void k_map_export(MapDev m, const int* __restrict__ rank,
                                                   int64_t* __restrict__ keys_out, int* __restrict__ rows_out) {
    const int64_t r = (int64_t)blockIdx.x * HB + threadIdx.x;
    if (r >= m.C || !m.row_live[r]) return;
    const int d = rank[r];
    keys_out[d] = m.row_key[r];
    rows_out[d] = (int)r;
}


// ===== MapTensorGet as one short chain ============================================================================
// mrec_map_lookup = k_map_probe -> k_map_place -> k_map_finish (3 launches; probe alone when not inserting):
//   probe : every key position probes the index; a hit counts (training lookups) and is done.  A miss that may insert
//           enters the call's scratch table by POSITION (claimed by CAS, lowered by atomicMin: the slot ends holding
//           the first position of its key, whatever the race order) unless the caller says the keys are unique.
//   place : one pass over the positions (decoupled look-back scan): the first occurrence of every missing key takes
//           the next row -- fresh rows in order, then the free list -- and enters the index.  Row numbers therefore
//           follow the order of first appearance in `keys`, as a sequential insert loop would hand them out.
//   finish: default values for the new rows of every table (values + optimizer slots), row numbers for the later
//           positions of new keys, the admission mask.
constexpr int kEmptyPos = 0x7f7f7f7f;
constexpr unsigned kMFlagA = 1u << 30, kMFlagP = 2u << 30, kMFlagMask = 3u << 30;
constexpr uint32_t F_INSERT = 1u, F_UNIQUE = 2u, F_TRAIN = 4u, F_PRIMED = 8u, F_SKIP_PAD = 16u;

struct MapTab { float* rows; int64_t ld; int D; float sigma; float fill; uint64_t seed; };
struct MapTabs { MapTab t[8]; int n; };

struct LookupWs {
    int* slots;        // [cap] position-valued scratch table (EMPTY between calls)
    int* srank;        // [cap] row of the slot's key
    int* sidx;         // [n]   scratch slot of a missing position
    unsigned* status;  // [tiles] look-back words (zero between calls)
    int* newrow;       // [n]   rows handed out by this call, in rank order
    int64_t* newkey;   // [n]
    int64_t* words;    // [0] = new keys of this call
    uint32_t smask;
    // mrec_map_lookup_out: the lookup's OUTPUT rows of the first occurrences of new keys are written by the kernel that generates
    // their default rows (one pass less over them than table -> gather -> output)
    int* newpos;       // [n]   position of the first occurrence, in rank order
    float* out;        // [n, ldo] (nullable)
    int64_t ldo;
    int out_tab;       // which table's rows `out` holds
    int* rows_gather;  // [n]   row per position for the gather BEHIND this call: -1 where `out` is written here
};

__device__ __forceinline__ int map_lookback(unsigned* status, int tile) {
    const int l = lane_id();
    int excl = 0;
    for (int base = tile - 1;; base -= 64) {
        const int idx = base - l;
        unsigned st;
        do {
            st = (idx >= 0) ? __hip_atomic_load(&status[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kMFlagP;
        } while (__any((st & kMFlagMask) == 0));
        const uint64_t pm = __ballot((st & kMFlagMask) == kMFlagP);
        const int firstp = pm ? __ffsll((long long)pm) - 1 : 63;
        int c = (l <= firstp) ? (int)(st & ~kMFlagMask) : 0;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
        excl += c;
        if (pm) break;
    }
    return excl;
}

// tile-exclusive rank of this thread's flagged items among all tiles; *total_out (last tile, thread 0 only) = grand total
__device__ __forceinline__ int map_tile_scan(int c, unsigned* status, int* sm, int* s_excl, int* tile_total) {
    int tot;
    const int pre = block_excl_scan_256(c, sm, &tot);
    if (threadIdx.x < 64) {
        const int t = blockIdx.x;
        int excl = 0;
        if (t == 0) {
            if (threadIdx.x == 0) __hip_atomic_store(&status[0], kMFlagP | (unsigned)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (threadIdx.x == 0) __hip_atomic_store(&status[t], kMFlagA | (unsigned)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            excl = map_lookback(status, t);
            if (threadIdx.x == 0) __hip_atomic_store(&status[t], kMFlagP | (unsigned)(excl + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (threadIdx.x == 0) *s_excl = excl;
    }
    __syncthreads();
    *tile_total = tot;
    return *s_excl + pre;
}

template <class K>
__global__ __launch_bounds__(HB) void k_map_probe(MapDev m, const K* __restrict__ keys, int64_t n_max,
                                                  const int64_t* __restrict__ n_dev, uint32_t flags, int step,
                                                  int* __restrict__ rows_out, LookupWs w) {
    const int64_t i = (int64_t)blockIdx.x * HB + threadIdx.x;
    if (i >= n_max) return;
    if (i >= eff_n(n_max, n_dev)) { rows_out[i] = -1; if (w.rows_gather) w.rows_gather[i] = -1; return; }
    const int64_t key = (int64_t)keys[i];
    if ((flags & F_SKIP_PAD) && key == -1) {      // a padding slot of a request message: nobody's key
        rows_out[i] = -1;
        w.sidx[i] = -1;
        if (w.rows_gather) w.rows_gather[i] = -1;
        return;
    }
    const uint32_t hsh = mrec_hash_key(key);
    uint32_t s = hsh & m.mask;
    int row = -1;
    for (uint32_t it = 0; it <= m.mask; ++it) {
        const int r = m.slot[s].row;
        if (r == -1) break;
        if (r >= 0 && m.slot[s].key == key) { row = r; break; }
        s = (s + 1) & m.mask;
    }
    rows_out[i] = row;
    if (w.rows_gather) w.rows_gather[i] = row;       // (a miss stays -1 where the finishing kernel writes the output row itself)
    if (row >= 0) {
        if (flags & F_TRAIN) {
            // one hit per key and step, however many positions carry the key: the position that moves last_step counts
            if (atomicExch(&m.last_step[row], step) != step) m.hits[row] += 1;
            m.dirty[row] = 1;
        }
        return;
    }
    if (!(flags & F_INSERT)) return;
    // (a lookup of resident keys -- the steady state -- ends here for every lane; the two kernels behind this one
    // return at once when this word stays zero)
    if (__hip_atomic_load(&w.words[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) atomicAdd((unsigned long long*)&w.words[1], 1ull);
    if (!(flags & F_UNIQUE)) {
        uint32_t q = (hsh >> 9) & w.smask;
        for (;;) {
            const int cur = atomicCAS(&w.slots[q], kEmptyPos, (int)i);
            if (cur == kEmptyPos) break;
            if ((int64_t)keys[cur] == key) {
                if (cur > (int)i) atomicMin(&w.slots[q], (int)i);
                break;
            }
            q = (q + 1) & w.smask;
        }
        w.sidx[i] = (int)q;
    }
}

template <class K>
__global__ __launch_bounds__(HB) void k_map_place(MapDev m, const K* __restrict__ keys, int64_t n_max,
                                                  const int64_t* __restrict__ n_dev, uint32_t flags, int step,
                                                  int* __restrict__ rows_out, LookupWs w, int ntiles) {
    __shared__ int sm[8];
    __shared__ int s_excl;
    if (w.words[1] == 0) return;                  // no key of this call was missing
    // the row counters as they stood before this call: read by every tile BEFORE it publishes its count, so the last
    // tile -- whose look-back ends only after all others published -- can commit the new values
    int64_t hwm0 = m.counters[C_HWM], nfree0 = m.counters[C_FREE];
    // (the two loads have RETURNED before anything below is issued -- what the ordering above needs; a release fence in their place
    // also wrote back whatever the workgroup's CU had dirtied in L2: MREC_MAP_FENCE=1 keeps it for A/B builds)
#if defined(MREC_MAP_FENCE) && MREC_MAP_FENCE
    __threadfence();
#else
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(hwm0), "+v"(nfree0) : : "memory");
#endif
    const int64_t eff = eff_n(n_max, n_dev);
    const int64_t base = (int64_t)blockIdx.x * HT + threadIdx.x * HI;
    bool first[HI];
    int c = 0;
#pragma unroll
    for (int k = 0; k < HI; ++k) {
        const int64_t i = base + k;
        bool f = (i < eff) && rows_out[i] < 0;
        if (f && (flags & F_SKIP_PAD) && (int64_t)keys[i] == -1) f = false;
        if (f && !(flags & F_UNIQUE)) f = (w.slots[w.sidx[i]] == (int)i);
        first[k] = f;
        c += f;
    }
    int tot;
    int r = map_tile_scan(c, w.status, sm, &s_excl, &tot);
    const int64_t fresh = m.C - hwm0;
#pragma unroll
    for (int k = 0; k < HI; ++k) {
        if (!first[k]) continue;
        const int64_t i = base + k;
        const int64_t key = (int64_t)keys[i];
        int row = -1;
        if (r < fresh) row = (int)(hwm0 + r);
        else if (r - fresh < nfree0) row = m.free_list[nfree0 - 1 - (r - fresh)];
        if (row >= 0) {
            uint32_t s = mrec_hash_key(key) & m.mask;
            bool placed = false;
            for (uint32_t it = 0; it <= m.mask; ++it) {
                const int cur = __hip_atomic_load(&m.slot[s].row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (cur < 0) {
                    if (atomicCAS(&m.slot[s].row, cur, row) == cur) {
                        if (cur == -2) atomicAdd((unsigned long long*)&m.counters[C_TOMB], ~0ull);      // -1
                        placed = true;
                        break;
                    }
                    continue;                      // somebody else took it: look at the same slot again
                }
                s = (s + 1) & m.mask;
            }
            if (placed) {
                m.slot[s].key = key;
                m.row_key[row] = key;
                m.row_live[row] = 1;
                m.hits[row] = (flags & F_TRAIN) ? 1 : 0;
                m.last_step[row] = step;
                m.dirty[row] = 1;
            } else {
                row = -1;                          // cannot happen while live <= C <= S / 2 (bounded all the same)
            }
        }
        rows_out[i] = row;
        w.newrow[r] = row;
        w.newkey[r] = key;
        if (w.newpos) w.newpos[r] = (int)i;
        if (!(flags & F_UNIQUE)) {
            w.srank[w.sidx[i]] = row;
            w.slots[w.sidx[i]] = kEmptyPos;        // the scratch table leaves the call as it entered it
        }
        ++r;
    }
    if ((int)blockIdx.x == ntiles - 1 && threadIdx.x == 0) {
        const int64_t M = (int64_t)s_excl + tot;
        const int64_t use_fresh = M < fresh ? M : fresh;
        const int64_t use_free = (M - use_fresh) < nfree0 ? (M - use_fresh) : nfree0;
        m.counters[C_HWM] = hwm0 + use_fresh;
        m.counters[C_FREE] = nfree0 - use_free;
        m.counters[C_LIVE] += use_fresh + use_free;
        m.counters[C_DROPPED] += M - use_fresh - use_free;
        w.words[0] = M;
    }
}

template <class K>
__global__ __launch_bounds__(HB) void k_map_finish(MapDev m, MapTabs tabs, int64_t n_max, const int64_t* __restrict__ n_dev,
                                                   uint32_t flags, int permit, int* __restrict__ rows_out,
                                                   int* __restrict__ rows_adm, LookupWs w, int ntiles) {
    const int64_t tid = (int64_t)blockIdx.x * HB + threadIdx.x, nthreads = (int64_t)gridDim.x * HB;
    const bool inserting = (flags & F_INSERT) != 0;
    const bool placed = inserting && w.words[1] != 0;
    const int64_t n_new = placed ? w.words[0] : 0;
    if (placed)
        for (int64_t j = tid; j < ntiles; j += nthreads) w.status[j] = 0;          // look-back words, for the next call
    if (n_new > 0) {
        // default rows: per table, a lane owns 4 consecutive columns (one 16-byte store) and a wave as many rows at a time
        // as fit its 64 lanes; the table loop is outermost so its parameters stay in scalar registers
        const int lane = threadIdx.x & 63;
        const int64_t wave = tid >> 6, nwaves = nthreads >> 6;
        for (int t = 0; t < tabs.n; ++t) {
            const MapTab tb = tabs.t[t];
            const bool vec = (tb.D % 4 == 0) && (tb.ld % 4 == 0) && ((((uintptr_t)tb.rows) & 15) == 0);
            const int lpr = vec ? (tb.D / 4 < 64 ? tb.D / 4 : 64) : 1;          // lanes per row
            const int rpw = 64 / lpr;                                            // rows per wave and round
            const int sub = lane % lpr, rin = lane / lpr;
            for (int64_t r0 = wave * rpw; r0 < n_new; r0 += nwaves * rpw) {
                const int64_t r = r0 + rin;
                if (rin >= rpw || r >= n_new) continue;
                const int row = w.newrow[r];
                if (row < 0) continue;
                const int64_t key = w.newkey[r];
                float* dst = tb.rows + (int64_t)row * tb.ld;
                float* dst2 = (w.out && t == w.out_tab) ? w.out + (int64_t)w.newpos[r] * w.ldo : nullptr;       // the lookup's output row of the key's first position
                if (vec) {
                    for (int c = sub * 4; c < tb.D; c += lpr * 4) {
                        float4 v;
                        if (tb.sigma >= 0.0f) {
                            mrec_det_normal2(tb.seed, key, c >> 1, v.x, v.y);          // (c is a multiple of 4)
                            mrec_det_normal2(tb.seed, key, (c >> 1) + 1, v.z, v.w);
                            v.x *= tb.sigma; v.y *= tb.sigma; v.z *= tb.sigma; v.w *= tb.sigma;
                        } else {
                            v = make_float4(tb.fill, tb.fill, tb.fill, tb.fill);
                        }
                        *(float4*)(dst + c) = v;
                        if (dst2) *(float4*)(dst2 + c) = v;
                    }
                } else {
                    for (int c = 0; c < tb.D; ++c) {
                        const float x = tb.sigma >= 0.0f ? tb.sigma * mrec_det_normal(tb.seed, key, c) : tb.fill;
                        dst[c] = x;
                        if (dst2) dst2[c] = x;
                    }
                }
            }
        }
    }
    const int64_t eff = eff_n(n_max, n_dev);
    const bool fix = n_new > 0 && !(flags & F_UNIQUE);
    if (!fix && !rows_adm) return;
    for (int64_t i = tid; i < n_max; i += nthreads) {
        int row = rows_out[i];
        if (fix && i < eff && row < 0 && !((flags & F_SKIP_PAD) && w.sidx[i] < 0)) {
            row = w.srank[w.sidx[i]];
            rows_out[i] = row;
            if (w.rows_gather) w.rows_gather[i] = row;      // a LATER position of a new key: the gather behind this call reads the new row
        }
        if (rows_adm) rows_adm[i] = (row >= 0 && m.hits[row] >= permit) ? row : -1;
    }
}

// out[i, :] = the default row of keys[i] wherever rows[i] < 0 (MapTensorGet with insert_default_value=False: a missing key
// reads as its default value without entering the table)
template <class K>
__global__ __launch_bounds__(HB) void k_map_fill_missing(const K* __restrict__ keys, const int* __restrict__ rows, int64_t n,
                                                         float* __restrict__ out, int64_t ldo, MapTab tb) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * HB + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * HB) >> 6;
    for (int64_t base = wave * 64; base < n; base += nwaves * 64) {
        const int64_t i = base + lane;
        const bool missing = i < n && rows[i] < 0;
        const int64_t key = missing ? (int64_t)keys[i] : 0;
        uint64_t mask = __ballot(missing);
        while (mask) {
            const int src = __ffsll((unsigned long long)mask) - 1;
            mask &= mask - 1;
            const int64_t kk = ((int64_t)__shfl((int)(key >> 32), src, 64) << 32) | (uint32_t)__shfl((int)key, src, 64);
            float* dst = out + (base + src) * ldo;
            for (int c = 2 * lane; c < tb.D; c += 128) {          // a lane fills a pair of columns: one transform
                float z0 = tb.fill, z1 = tb.fill;
                if (tb.sigma >= 0.0f) {
                    mrec_det_normal2(tb.seed, kk, c >> 1, z0, z1);
                    z0 *= tb.sigma; z1 *= tb.sigma;
                }
                dst[c] = z0;
                if (c + 1 < tb.D) dst[c + 1] = z1;
            }
        }
    }
}

// ===== eviction on the device (README.md:182-183: evict_filter_value in training steps) ============================
// One pass over the rows: a live row whose key was last looked up for training more than `threshold` steps ago leaves the
// index (tombstone), joins the free list in row order and its key the erased-keys log.
__global__ __launch_bounds__(HB) void k_map_evict(MapDev m, int step, int64_t threshold, unsigned* status, int ntiles,
                                                  int64_t* __restrict__ n_evicted, int64_t S) {
    __shared__ int sm[8];
    __shared__ int s_excl;
    int64_t nfree0 = m.counters[C_FREE], nlog0 = m.counters[C_NLOG];
#if defined(MREC_MAP_FENCE) && MREC_MAP_FENCE
    __threadfence();
#else
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(nfree0), "+v"(nlog0) : : "memory");      // (as k_map_place: returned before anything below is issued)
#endif
    const int64_t base = (int64_t)blockIdx.x * HT + threadIdx.x * HI;
    bool dead[HI];
    int c = 0;
#pragma unroll
    for (int k = 0; k < HI; ++k) {
        const int64_t r = base + k;
        dead[k] = r < m.C && m.row_live[r] && ((int64_t)step - (int64_t)m.last_step[r] > threshold);
        c += dead[k];
    }
    int tot;
    int rk = map_tile_scan(c, status, sm, &s_excl, &tot);
#pragma unroll
    for (int k = 0; k < HI; ++k) {
        if (!dead[k]) continue;
        const int64_t r = base + k;
        const int64_t key = m.row_key[r];
        uint32_t s = mrec_hash_key(key) & m.mask;
        for (uint32_t it = 0; it <= m.mask; ++it) {
            const int cur = m.slot[s].row;
            if (cur == -1) break;
            if (cur == (int)r) { m.slot[s].row = -2; break; }
            s = (s + 1) & m.mask;
        }
        m.row_live[r] = 0;
        m.free_list[nfree0 + rk] = (int)r;
        if (nlog0 + rk < m.C) m.erased_log[nlog0 + rk] = key;
        ++rk;
    }
    if ((int)blockIdx.x == ntiles - 1 && threadIdx.x == 0) {
        const int64_t M = (int64_t)s_excl + tot;
        *n_evicted = M;
        m.counters[C_FREE] = nfree0 + M;
        m.counters[C_LIVE] -= M;
        m.counters[C_NLOG] = (nlog0 + M < m.C) ? nlog0 + M : m.C;
        const int64_t tomb = m.counters[C_TOMB] + M;
        const bool rebuild = tomb * 5 > S;
        m.counters[C_REBUILD] = rebuild ? 1 : 0;
        m.counters[C_TOMB] = rebuild ? 0 : tomb;
        if (rebuild) m.counters[C_NREBUILD] += 1;
    }
}

// ===== incremental export ==============================================================================================
__global__ __launch_bounds__(HB) void k_map_dirty_flags(MapDev m, uint8_t* __restrict__ f) {
    const int64_t r = (int64_t)blockIdx.x * HB + threadIdx.x;
    if (r < m.C) f[r] = m.row_live[r] && m.dirty[r];
}
__global__ __launch_bounds__(HB) void k_map_export_flagged(MapDev m, const uint8_t* __restrict__ f, const int* __restrict__ rank,
                                                           int64_t* __restrict__ keys_out, int* __restrict__ rows_out,
                                                           int* __restrict__ status_out) {
    const int64_t r = (int64_t)blockIdx.x * HB + threadIdx.x;
    if (r >= m.C || !f[r]) return;
    const int d = rank[r];
    keys_out[d] = m.row_key[r];
    rows_out[d] = (int)r;
    status_out[d] = 1;                 // modified (inserted, trained on or put since the last incremental export)
}
// log entries whose key is not live now (a key erased and inserted again is reported as modified only)
__global__ __launch_bounds__(HB) void k_map_log_flags(MapDev m, uint8_t* __restrict__ f) {
    const int64_t j = (int64_t)blockIdx.x * HB + threadIdx.x;
    if (j >= m.C) return;
    bool gone = false;
    if (j < m.counters[C_NLOG]) {
        const int64_t key = m.erased_log[j];
        uint32_t s = mrec_hash_key(key) & m.mask;
        gone = true;
        for (uint32_t it = 0; it <= m.mask; ++it) {
            const int r = m.slot[s].row;
            if (r == -1) break;
            if (r >= 0 && m.slot[s].key == key) { gone = false; break; }
            s = (s + 1) & m.mask;
        }
    }
    f[j] = gone;
}
__global__ __launch_bounds__(HB) void k_map_export_log(MapDev m, const uint8_t* __restrict__ f, const int* __restrict__ rank,
                                                       const int64_t* __restrict__ n_mod, const int64_t* __restrict__ n_gone,
                                                       int64_t* __restrict__ keys_out, int* __restrict__ rows_out,
                                                       int* __restrict__ status_out, int64_t* __restrict__ n_out) {
    const int64_t j = (int64_t)blockIdx.x * HB + threadIdx.x;
    if (j == 0) *n_out = *n_mod + *n_gone;
    if (j >= m.C || !f[j]) return;
    const int64_t d = *n_mod + rank[j];
    keys_out[d] = m.erased_log[j];
    rows_out[d] = -1;
    status_out[d] = 2;                 // erased
}
__global__ __launch_bounds__(HB) void k_map_clear_dirty(MapDev m) {
    const int64_t r = (int64_t)blockIdx.x * HB + threadIdx.x;
    if (r < m.C) m.dirty[r] = 0;
    if (r == 0) m.counters[C_NLOG] = 0;
}
__global__ __launch_bounds__(HB) void k_map_mark_dirty(MapDev m, const int* __restrict__ rows, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * HB + threadIdx.x;
    if (i < n && rows[i] >= 0 && rows[i] < m.C) m.dirty[rows[i]] = 1;
}

// ===== MapTensorPut with duplicates: the LAST position of a key wins (a sequential upsert loop) ==================
__global__ __launch_bounds__(HB) void k_put_winner(const int* __restrict__ rows, int64_t n, int* __restrict__ winner) {
    const int64_t i = (int64_t)blockIdx.x * HB + threadIdx.x;
    if (i < n && rows[i] >= 0) atomicMax(&winner[rows[i]], (int)i);
}
__global__ __launch_bounds__(HB) void k_put_rows(float* __restrict__ table, int64_t ld, int D, const int* __restrict__ rows,
                                                 int64_t n, const float* __restrict__ vals, int* __restrict__ winner) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const int r = rows[i];
    if (r < 0 || winner[r] != (int)i) return;
    for (int c = lane; c < D; c += 64) table[(int64_t)r * ld + c] = vals[i * D + c];
}
__global__ __launch_bounds__(HB) void k_put_reset(const int* __restrict__ rows, int64_t n, int* __restrict__ winner) {
    const int64_t i = (int64_t)blockIdx.x * HB + threadIdx.x;
    if (i < n && rows[i] >= 0) winner[rows[i]] = -1;
}

size_t map_ws_bytes(int64_t n) {
    const size_t nn = (size_t)(n ? n : 1);
    return mrec_align_up(nn * 4, 256) * 3 + mrec_align_up(nn, 256) + mrec_align_up((size_t)mrec_cdiv(nn, HT) * 4, 256) +
           512;
}

}  // namespace

struct mrec_map {
    MapDev d;
    uint64_t S;
};

static void map_layout(int64_t C, uint64_t* S_out, size_t off[10], size_t* total) {
    uint64_t S = 1024;
    while (S < (uint64_t)C * 2) S <<= 1;
    size_t o = 0;
    off[0] = o; o += mrec_align_up(S * sizeof(MapSlot), 256);
    off[1] = o;                                   // (unused)
    off[2] = o; o += mrec_align_up((size_t)C * 8, 256);
    off[3] = o; o += mrec_align_up((size_t)C, 256);
    off[4] = o; o += mrec_align_up((size_t)C * 4, 256);
    off[5] = o; o += mrec_align_up(C_NCOUNTERS * 8, 256);
    off[6] = o; o += mrec_align_up((size_t)C * 4, 256);
    off[7] = o; o += mrec_align_up((size_t)C * 4, 256);
    off[8] = o; o += mrec_align_up((size_t)C, 256);
    off[9] = o; o += mrec_align_up((size_t)C * 8, 256);
    *S_out = S;
    *total = o;
}

MREC_API int mrec_map_bytes(int64_t capacity_rows, size_t* out) {
    if (!out || capacity_rows <= 0 || capacity_rows > (int64_t(1) << 30)) return MREC_EINVAL;
    uint64_t S; size_t off[10];
    map_layout(capacity_rows, &S, off, out);
    return MREC_OK;
}

MREC_API int mrec_map_create(mrec_map_t** out, void* mem, size_t mem_bytes, int64_t capacity_rows, void* stream) {
    if (!out || !mem || capacity_rows <= 0 || capacity_rows > (int64_t(1) << 30)) return MREC_EINVAL;
    if (((uintptr_t)mem) & 255) return MREC_EINVAL;
    uint64_t S; size_t off[10], total;
    map_layout(capacity_rows, &S, off, &total);
    if (mem_bytes < total) return MREC_EWORKSPACE;
    char* b = (char*)mem;
    mrec_map* h = new mrec_map;
    h->S = S;
    h->d.slot = (MapSlot*)(b + off[0]);
    h->d.row_key = (int64_t*)(b + off[2]);
    h->d.row_live = (uint8_t*)(b + off[3]);
    h->d.free_list = (int*)(b + off[4]);
    h->d.counters = (int64_t*)(b + off[5]);
    h->d.hits = (int*)(b + off[6]);
    h->d.last_step = (int*)(b + off[7]);
    h->d.dirty = (uint8_t*)(b + off[8]);
    h->d.erased_log = (int64_t*)(b + off[9]);
    h->d.C = capacity_rows;
    h->d.mask = (uint32_t)(S - 1);
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(h->d.slot, 0xFF, S * sizeof(MapSlot), st);        // row = -1: empty
    if (e == hipSuccess) e = hipMemsetAsync(h->d.row_live, 0, (size_t)capacity_rows, st);
    if (e == hipSuccess) e = hipMemsetAsync(h->d.counters, 0, C_NCOUNTERS * 8, st);
    if (e == hipSuccess) e = hipMemsetAsync(h->d.hits, 0, off[9] - off[6], st);        // hits, last_step, dirty
    if (e != hipSuccess) { g_mrec_last_hip_error = (int)e; delete h; return MREC_EHIP; }
    *out = h;
    return MREC_OK;
}

MREC_API int mrec_map_destroy(mrec_map_t* h) {
    delete h;
    return MREC_OK;
}

MREC_API const int64_t* mrec_map_counters_dev(const mrec_map_t* h) { return h ? h->d.counters : nullptr; }
MREC_API const int64_t* mrec_map_row_keys_dev(const mrec_map_t* h) { return h ? h->d.row_key : nullptr; }
MREC_API int mrec_map_tracking_dev(const mrec_map_t* h, int32_t** hits, int32_t** last_step, uint8_t** dirty) {
    if (!h) return MREC_EINVAL;
    if (hits) *hits = h->d.hits;
    if (last_step) *last_step = h->d.last_step;
    if (dirty) *dirty = h->d.dirty;
    return MREC_OK;
}

MREC_API int mrec_map_workspace_bytes(int64_t n, size_t* out) {
    if (!out || n < 0) return MREC_EINVAL;
    *out = map_ws_bytes(n);
    return MREC_OK;
}

MREC_API int mrec_map_find_or_insert(mrec_map_t* h, const int64_t* keys, int64_t n, const int64_t* n_dev, int insert,
                                     int32_t* rows_out, uint8_t* is_new_out, void* ws, size_t ws_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!h || n < 0) return MREC_EINVAL;
    if (n == 0) return MREC_OK;
    if (!keys || !rows_out || !ws) return MREC_EINVAL;
    if (n > (int64_t(1) << 30)) return MREC_EUNSUPPORTED;
    MrecArena a(ws, ws_bytes);
    int* rank = a.take<int>(n);
    int* slots = a.take<int>(n);
    uint8_t* miss = a.take<uint8_t>(n);
    const int nblk = (int)mrec_cdiv(n, HT);
    int* blocksum = a.take<int>(nblk);
    int64_t* words = a.take<int64_t>(3);  // [0] = misses, [1] = dropped in this call, [2] = tombstones taken back
    if (!a.ok) return MREC_EWORKSPACE;
    (void)slots;
    const unsigned g = (unsigned)mrec_cdiv(n, HB);
    k_map_find<<<g, HB, 0, st>>>(h->d, keys, n, n_dev, rows_out, nullptr, miss);
    if (insert) {
        MREC_HIP_CHECK(hipMemsetAsync(words, 0, 24, st));
        k_flag_count<<<nblk, HB, 0, st>>>(miss, n, 0, blocksum);
        k_flag_rank<<<nblk, HB, 0, st>>>(miss, n, 0, blocksum, nblk, rank, words);
        k_map_insert<<<g, HB, 0, st>>>(h->d, keys, n, miss, rank, rows_out, is_new_out, words + 1, words + 2);
        k_map_commit_insert<<<1, 1, 0, st>>>(h->d, words, words + 1, words + 2);
    } else if (is_new_out) {
        MREC_HIP_CHECK(hipMemsetAsync(is_new_out, 0, (size_t)n, st));
    }
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}

MREC_API int mrec_map_erase(mrec_map_t* h, const int64_t* keys, int64_t n, void* ws, size_t ws_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!h || n < 0) return MREC_EINVAL;
    if (n == 0) return MREC_OK;
    if (!keys || !ws) return MREC_EINVAL;
    if (n > (int64_t(1) << 30)) return MREC_EUNSUPPORTED;
    MrecArena a(ws, ws_bytes);
    int* rank = a.take<int>(n);
    int* slots = a.take<int>(n);
    uint8_t* miss = a.take<uint8_t>(n);
    const int nblk = (int)mrec_cdiv(n, HT);
    int* blocksum = a.take<int>(nblk);
    int64_t* words = a.take<int64_t>(2);  // [0] = keys found
    int* rows = a.take<int>(n);
    if (!a.ok) return MREC_EWORKSPACE;
    const unsigned g = (unsigned)mrec_cdiv(n, HB);
    k_map_find<<<g, HB, 0, st>>>(h->d, keys, n, nullptr, rows, slots, miss);
    k_flag_count<<<nblk, HB, 0, st>>>(miss, n, 1, blocksum);
    k_flag_rank<<<nblk, HB, 0, st>>>(miss, n, 1, blocksum, nblk, rank, words);
    k_map_erase<<<g, HB, 0, st>>>(h->d, n, rows, slots, miss, rank);
    k_map_commit_erase<<<1, 1, 0, st>>>(h->d, words, (int64_t)h->S);
    // both return at once unless the commit asked for a rebuild (grid-stride, so the idle launch is small)
    const unsigned gr = (unsigned)(mrec_cdiv((int64_t)h->S, HB) < 2048 ? mrec_cdiv((int64_t)h->S, HB) : 2048);
    k_map_rebuild_clear<<<gr, HB, 0, st>>>(h->d, (int64_t)h->S);
    k_map_rebuild_insert<<<gr, HB, 0, st>>>(h->d);
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}

MREC_API int mrec_map_export(mrec_map_t* h, int64_t* keys_out, int32_t* rows_out, int64_t* n_out_dev, void* ws,
                             size_t ws_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!h || !keys_out || !rows_out || !n_out_dev || !ws) return MREC_EINVAL;
    const int64_t C = h->d.C;
    MrecArena a(ws, ws_bytes);
    int* rank = a.take<int>(C);
    const int nblk = (int)mrec_cdiv(C, HT);
    int* blocksum = a.take<int>(nblk);
    if (!a.ok) return MREC_EWORKSPACE;
    k_flag_count<<<nblk, HB, 0, st>>>(h->d.row_live, C, 0, blocksum);
    k_flag_rank<<<nblk, HB, 0, st>>>(h->d.row_live, C, 0, blocksum, nblk, rank, n_out_dev);
    k_map_export<<<(unsigned)mrec_cdiv(C, HB), HB, 0, st>>>(h->d, rank, keys_out, rows_out);
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}

// ---- MapTensorGet chain -----------------------------------------------------------------------------------------
static uint64_t lookup_cap(int64_t n) {
    uint64_t cap = 1024;
    while (cap < (uint64_t)(n ? n : 1) * 2) cap <<= 1;
    return cap;
}
MREC_API int mrec_map_lookup_workspace_bytes(int64_t n, size_t* out) {
    if (!out || n < 0) return MREC_EINVAL;
    if (n >= (int64_t(1) << 30)) return MREC_EUNSUPPORTED;
    const size_t nn = (size_t)(n ? n : 1);
    const uint64_t cap = lookup_cap(n);
    *out = mrec_align_up(cap * 4, 256) * 2 + mrec_align_up(nn * 4, 256) * 3 + mrec_align_up(nn * 8, 256) +
           mrec_align_up((size_t)mrec_cdiv(nn, HT) * 4, 256) + 256;
    return MREC_OK;
}

template <class K>
static int map_lookup_impl(mrec_map* h, const K* keys, int64_t n, const int64_t* n_dev, uint32_t flags, int64_t step,
                           int32_t permit, const mrec_map_table_t* tables, int32_t n_tables, int32_t* rows_out,
                           int32_t* rows_adm, void* ws, size_t ws_bytes, hipStream_t st, float* out = nullptr, int64_t ldo = 0,
                           int32_t out_table = 0, int32_t* rows_gather = nullptr) {
    MapTabs tabs;
    tabs.n = n_tables;
    for (int t = 0; t < n_tables; ++t) {
        if (!tables[t].rows || tables[t].D <= 0 || tables[t].ld < tables[t].D) return MREC_EINVAL;
        tabs.t[t] = MapTab{tables[t].rows, tables[t].ld, tables[t].D, tables[t].sigma, tables[t].fill, tables[t].seed};
    }
    const int ntiles = (int)mrec_cdiv(n, HT);
    const uint64_t cap = lookup_cap(n);
    MrecArena a(ws, ws_bytes);
    LookupWs w;
    w.slots = a.take<int>(cap);
    w.srank = a.take<int>(cap);
    w.sidx = a.take<int>(n);
    w.newrow = a.take<int>(n);
    w.newkey = a.take<int64_t>(n);
    w.status = (unsigned*)a.take<int>(ntiles);
    w.words = a.take<int64_t>(2);
    w.smask = (uint32_t)(cap - 1);
    w.newpos = a.take<int>(n);
    w.out = out; w.ldo = ldo; w.out_tab = out_table; w.rows_gather = rows_gather;
    if (!a.ok) return MREC_EWORKSPACE;
    const bool inserting = (flags & F_INSERT) != 0;
    if (inserting && !(flags & F_PRIMED)) {
        MREC_HIP_CHECK(hipMemsetAsync(w.slots, 0x7f, cap * sizeof(int), st));      // (also for unique keys: "primed" must hold for any later call)
        MREC_HIP_CHECK(hipMemsetAsync(w.status, 0, (size_t)ntiles * sizeof(int), st));
    }
    if (inserting) MREC_HIP_CHECK(hipMemsetAsync(w.words, 0, 16, st));     // [0] new keys, [1] "some key was missing"
    const unsigned g = (unsigned)mrec_cdiv(n, HB);
    k_map_probe<K><<<g, HB, 0, st>>>(h->d, keys, n, n_dev, flags, (int)step, rows_out, w);
    if (inserting) k_map_place<K><<<ntiles, HB, 0, st>>>(h->d, keys, n, n_dev, flags, (int)step, rows_out, w, ntiles);
    if (inserting || rows_adm) {
        const unsigned gf = g < 2048 ? g : 2048;
        k_map_finish<K><<<gf, HB, 0, st>>>(h->d, tabs, n, n_dev, flags, permit, rows_out, rows_adm, w, ntiles);
    }
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}

MREC_API int mrec_map_lookup(mrec_map_t* h, const void* keys, int32_t key_bytes, int64_t n, const int64_t* n_dev, uint32_t flags,
                             int64_t step, int32_t permit, const mrec_map_table_t* tables, int32_t n_tables,
                             int32_t* rows_out, int32_t* rows_admitted_out, void* ws, size_t ws_bytes, void* stream) {
    if (!h || n < 0 || (key_bytes != 4 && key_bytes != 8) || n_tables < 0 || n_tables > 8 || (n_tables > 0 && !tables) ||
        step < 0 || step > 0x7fffffff)
        return MREC_EINVAL;
    if (n == 0) return MREC_OK;
    if (!keys || !rows_out || !ws) return MREC_EINVAL;
    if (n >= (int64_t(1) << 30)) return MREC_EUNSUPPORTED;
    if (key_bytes == 4)
        return map_lookup_impl<int32_t>(h, (const int32_t*)keys, n, n_dev, flags, step, permit, tables, n_tables, rows_out,
                                        rows_admitted_out, ws, ws_bytes, (hipStream_t)stream);
    return map_lookup_impl<int64_t>(h, (const int64_t*)keys, n, n_dev, flags, step, permit, tables, n_tables, rows_out,
                                    rows_admitted_out, ws, ws_bytes, (hipStream_t)stream);
}

/* MapTensorGet with insertion, the lookup's OUTPUT included for new keys: as mrec_map_lookup, and the kernel that generates the
 * default rows also writes them to out[position of the key's first occurrence, :] (table `out_table` of `tables`);
 * rows_gather[i] = the row the gather behind this call must read for position i, -1 where `out` is already written (run
 * mrec_gather_rows_f32_skip_i32(table, ..., rows_gather, n, out) next: one pass over the new rows less than lookup + gather). */
MREC_API int mrec_map_lookup_out(mrec_map_t* h, const void* keys, int32_t key_bytes, int64_t n, const int64_t* n_dev, uint32_t flags,
                                 int64_t step, int32_t permit, const mrec_map_table_t* tables, int32_t n_tables,
                                 int32_t* rows_out, int32_t* rows_admitted_out, float* out, int64_t ldo, int32_t out_table,
                                 int32_t* rows_gather, void* ws, size_t ws_bytes, void* stream) {
    if (!h || n < 0 || (key_bytes != 4 && key_bytes != 8) || n_tables <= 0 || n_tables > 8 || !tables || step < 0 || step > 0x7fffffff ||
        out_table < 0 || out_table >= n_tables)
        return MREC_EINVAL;
    if (n == 0) return MREC_OK;
    if (!keys || !rows_out || !ws || !out || !rows_gather || ldo < tables[out_table].D) return MREC_EINVAL;
    if (n >= (int64_t(1) << 30)) return MREC_EUNSUPPORTED;
    if ((ldo % 4) || (((uintptr_t)out) & 15)) return MREC_EUNSUPPORTED;
    if (key_bytes == 4)
        return map_lookup_impl<int32_t>(h, (const int32_t*)keys, n, n_dev, flags, step, permit, tables, n_tables, rows_out,
                                        rows_admitted_out, ws, ws_bytes, (hipStream_t)stream, out, ldo, out_table, rows_gather);
    return map_lookup_impl<int64_t>(h, (const int64_t*)keys, n, n_dev, flags, step, permit, tables, n_tables, rows_out,
                                    rows_admitted_out, ws, ws_bytes, (hipStream_t)stream, out, ldo, out_table, rows_gather);
}

MREC_API int mrec_map_fill_missing(const void* keys, int32_t key_bytes, const int32_t* rows, int64_t n, float* out, int64_t ldo,
                                   const mrec_map_table_t* table, void* stream) {
    if (n < 0 || (key_bytes != 4 && key_bytes != 8) || !table || table->D <= 0 || ldo < table->D) return MREC_EINVAL;
    if (n == 0) return MREC_OK;
    if (!keys || !rows || !out) return MREC_EINVAL;
    MapTab tb{nullptr, 0, table->D, table->sigma, table->fill, table->seed};
    const unsigned g = (unsigned)(mrec_cdiv(n, HB) < 2048 ? mrec_cdiv(n, HB) : 2048);
    if (key_bytes == 4) k_map_fill_missing<int32_t><<<g, HB, 0, (hipStream_t)stream>>>((const int32_t*)keys, rows, n, out, ldo, tb);
    else k_map_fill_missing<int64_t><<<g, HB, 0, (hipStream_t)stream>>>((const int64_t*)keys, rows, n, out, ldo, tb);
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}

MREC_API int mrec_map_evict(mrec_map_t* h, int64_t step, int64_t threshold, int64_t* n_evicted_dev, void* ws, size_t ws_bytes,
                            void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!h || !n_evicted_dev || !ws || step < 0 || step > 0x7fffffff || threshold < 0) return MREC_EINVAL;
    const int ntiles = (int)mrec_cdiv(h->d.C, HT);
    if (ws_bytes < (size_t)ntiles * 4) return MREC_EWORKSPACE;
    MREC_HIP_CHECK(hipMemsetAsync(ws, 0, (size_t)ntiles * 4, st));
    k_map_evict<<<ntiles, HB, 0, st>>>(h->d, (int)step, threshold, (unsigned*)ws, ntiles, n_evicted_dev, (int64_t)h->S);
    const unsigned gr = (unsigned)(mrec_cdiv((int64_t)h->S, HB) < 2048 ? mrec_cdiv((int64_t)h->S, HB) : 2048);
    k_map_rebuild_clear<<<gr, HB, 0, st>>>(h->d, (int64_t)h->S);
    k_map_rebuild_insert<<<gr, HB, 0, st>>>(h->d);
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}

MREC_API int mrec_map_export_dirty(mrec_map_t* h, int64_t* keys_out, int32_t* rows_out, int32_t* status_out,
                                   int64_t* n_out_dev, int clear, void* ws, size_t ws_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!h || !keys_out || !rows_out || !status_out || !n_out_dev || !ws) return MREC_EINVAL;
    const int64_t C = h->d.C;
    MrecArena a(ws, ws_bytes);
    int* rank = a.take<int>(C);
    const int nblk = (int)mrec_cdiv(C, HT);
    int* blocksum = a.take<int>(nblk);
    uint8_t* f = a.take<uint8_t>(C);
    int64_t* words = a.take<int64_t>(2);
    if (!a.ok) return MREC_EWORKSPACE;
    const unsigned g = (unsigned)mrec_cdiv(C, HB);
    k_map_dirty_flags<<<g, HB, 0, st>>>(h->d, f);
    k_flag_count<<<nblk, HB, 0, st>>>(f, C, 0, blocksum);
    k_flag_rank<<<nblk, HB, 0, st>>>(f, C, 0, blocksum, nblk, rank, words);
    k_map_export_flagged<<<g, HB, 0, st>>>(h->d, f, rank, keys_out, rows_out, status_out);
    k_map_log_flags<<<g, HB, 0, st>>>(h->d, f);
    k_flag_count<<<nblk, HB, 0, st>>>(f, C, 0, blocksum);
    k_flag_rank<<<nblk, HB, 0, st>>>(f, C, 0, blocksum, nblk, rank, words + 1);
    k_map_export_log<<<g, HB, 0, st>>>(h->d, f, rank, words, words + 1, keys_out, rows_out, status_out, n_out_dev);
    if (clear) k_map_clear_dirty<<<g, HB, 0, st>>>(h->d);
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}

MREC_API int mrec_map_mark_dirty(mrec_map_t* h, const int32_t* rows, int64_t n, void* stream) {
    if (!h || n < 0) return MREC_EINVAL;
    if (n == 0) return MREC_OK;
    if (!rows) return MREC_EINVAL;
    k_map_mark_dirty<<<(unsigned)mrec_cdiv(n, HB), HB, 0, (hipStream_t)stream>>>(h->d, rows, n);
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}

// winner: int32 scratch of one word per table row, all -1 between calls (the call restores it)
MREC_API int mrec_put_rows_last_f32(float* table, int64_t ld, int32_t D, const int32_t* rows, int64_t n, const float* vals,
                                    int32_t* winner, void* stream) {
    if (n < 0 || D <= 0 || ld < D) return MREC_EINVAL;
    if (n == 0) return MREC_OK;
    if (!table || !rows || !vals || !winner) return MREC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const unsigned g = (unsigned)mrec_cdiv(n, HB);
    k_put_winner<<<g, HB, 0, st>>>(rows, n, winner);
    k_put_rows<<<(unsigned)mrec_cdiv(n, 4), HB, 0, st>>>(table, ld, D, rows, n, vals, winner);
    k_put_reset<<<g, HB, 0, st>>>(rows, n, winner);
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}
