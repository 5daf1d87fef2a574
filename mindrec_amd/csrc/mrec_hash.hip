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

namespace {

constexpr int HB = 256;
constexpr int HI = 8;
constexpr int HT = HB * HI;

// counters: [C_TOMB] = tombstones in the slot array, [C_REBUILD] = "rebuild the slot array now" (set by the erase
// commit, read by the two rebuild kernels behind it), [C_NREBUILD] = rebuilds so far
enum { C_HWM = 0, C_LIVE = 1, C_DROPPED = 2, C_FREE = 3, C_TOMB = 4, C_REBUILD = 5, C_NREBUILD = 6, C_NCOUNTERS = 8 };

struct MapDev {
    int64_t* skey;      // [S]
    int* srow;          // [S]  -1 empty, -2 tombstone, >= 0 row
    int64_t* row_key;   // [C]
    uint8_t* row_live;  // [C]
    int* free_list;     // [C]
    int64_t* counters;  // [C_NCOUNTERS]
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
        const int r = m.srow[s];
        if (r == -1) break;
        if (r >= 0 && m.skey[s] == key) { row = r; slot = (int)s; break; }
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
        const int cur = __hip_atomic_load(&m.srow[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur < 0) {
            const int old = atomicCAS(&m.srow[s], cur, row);
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
    m.skey[s] = key;
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
    const int64_t nfree = m.counters[C_FREE];
    m.srow[slots[i]] = -2;
    m.row_live[rows[i]] = 0;
    m.free_list[nfree + rank[i]] = rows[i];
}

// Erase leaves tombstones, and an insert only takes one back when its probe happens to pass it: under
// insert / erase churn the EMPTY slots (the only thing that ends a miss probe) would run out.  Once
// tombstones exceed S / 5 (live <= S / 2, so live + tombstones <= 0.7 S always holds) the slot array is
// rebuilt from the row side (row_key / row_live), which empties every tombstone.
__global__ void k_map_commit_erase(MapDev m, const int64_t* n_found, int64_t S) {
    m.counters[C_FREE] += *n_found;
    m.counters[C_LIVE] -= *n_found;
    const int64_t tomb = m.counters[C_TOMB] + *n_found;
    const bool rebuild = tomb * 5 > S;
    m.counters[C_REBUILD] = rebuild ? 1 : 0;
    m.counters[C_TOMB] = rebuild ? 0 : tomb;
    if (rebuild) m.counters[C_NREBUILD] += 1;
}

__global__ __launch_bounds__(HB) void k_map_rebuild_clear(MapDev m, int64_t S) {
    if (!m.counters[C_REBUILD]) return;
    for (int64_t s = (int64_t)blockIdx.x * HB + threadIdx.x; s < S; s += (int64_t)gridDim.x * HB) m.srow[s] = -1;
}

__global__ __launch_bounds__(HB) void k_map_rebuild_insert(MapDev m) {
    if (!m.counters[C_REBUILD]) return;
    const int64_t hwm = m.counters[C_HWM];
    for (int64_t r = (int64_t)blockIdx.x * HB + threadIdx.x; r < hwm; r += (int64_t)gridDim.x * HB) {
        if (!m.row_live[r]) continue;
        const int64_t key = m.row_key[r];
        uint32_t s = mrec_hash_key(key) & m.mask;
        for (uint32_t it = 0; it <= m.mask; ++it) {
            if (atomicCAS(&m.srow[s], -1, (int)r) == -1) { m.skey[s] = key; break; }
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

static void map_layout(int64_t C, uint64_t* S_out, size_t off[6], size_t* total) {
    uint64_t S = 1024;
    while (S < (uint64_t)C * 2) S <<= 1;
    size_t o = 0;
    off[0] = o; o += mrec_align_up(S * 8, 256);
    off[1] = o; o += mrec_align_up(S * 4, 256);
    off[2] = o; o += mrec_align_up((size_t)C * 8, 256);
    off[3] = o; o += mrec_align_up((size_t)C, 256);
    off[4] = o; o += mrec_align_up((size_t)C * 4, 256);
    off[5] = o; o += mrec_align_up(C_NCOUNTERS * 8, 256);
    *S_out = S;
    *total = o;
}

MREC_API int mrec_map_bytes(int64_t capacity_rows, size_t* out) {
    if (!out || capacity_rows <= 0 || capacity_rows > (int64_t(1) << 30)) return MREC_EINVAL;
    uint64_t S; size_t off[6];
    map_layout(capacity_rows, &S, off, out);
    return MREC_OK;
}

MREC_API int mrec_map_create(mrec_map_t** out, void* mem, size_t mem_bytes, int64_t capacity_rows, void* stream) {
    if (!out || !mem || capacity_rows <= 0 || capacity_rows > (int64_t(1) << 30)) return MREC_EINVAL;
    if (((uintptr_t)mem) & 255) return MREC_EINVAL;
    uint64_t S; size_t off[6], total;
    map_layout(capacity_rows, &S, off, &total);
    if (mem_bytes < total) return MREC_EWORKSPACE;
    char* b = (char*)mem;
    mrec_map* h = new mrec_map;
    h->S = S;
    h->d.skey = (int64_t*)(b + off[0]);
    h->d.srow = (int*)(b + off[1]);
    h->d.row_key = (int64_t*)(b + off[2]);
    h->d.row_live = (uint8_t*)(b + off[3]);
    h->d.free_list = (int*)(b + off[4]);
    h->d.counters = (int64_t*)(b + off[5]);
    h->d.C = capacity_rows;
    h->d.mask = (uint32_t)(S - 1);
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(h->d.srow, 0xFF, S * 4, st);
    if (e == hipSuccess) e = hipMemsetAsync(h->d.row_live, 0, (size_t)capacity_rows, st);
    if (e == hipSuccess) e = hipMemsetAsync(h->d.counters, 0, C_NCOUNTERS * 8, st);
    if (e != hipSuccess) { g_mrec_last_hip_error = (int)e; delete h; return MREC_EHIP; }
    *out = h;
    return MREC_OK;
}

MREC_API int mrec_map_destroy(mrec_map_t* h) {
    delete h;
    return MREC_OK;
}

MREC_API const int64_t* mrec_map_counters_dev(const mrec_map_t* h) { return h ? h->d.counters : nullptr; }

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
