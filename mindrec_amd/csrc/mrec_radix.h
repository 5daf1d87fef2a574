// mrec_radix.h -- one stable LSD radix pass over int32 keys (11-bit digits max), shared by the
// group-by of mrec_dedup.hip and the shard routing of mrec_route.hip.
//
// In-tile ranks come from wave64 ballot matching: nbits __ballot()s give every lane the mask of
// lanes sharing its digit; popcount of the lower lanes is its rank inside the wave-round; per-wave
// LDS counters carry the rank across the 8 rounds of a wave's 512-key sub-tile, and a cross-wave
// prefix plus the scanned global histogram turn it into the destination.
#pragma once
#include "mrec_common.h"

namespace {

constexpr int RMAXB = 11;             // max digit bits
constexpr int RNB = 1 << RMAXB;       // max bins
constexpr int RT = 2048;              // keys per tile: 4 waves x 8 rounds x 64 lanes
constexpr int RROUNDS = 8;

// When (slots, sidx, pos, inv) are given, the keys are produced here -- the group of the i-th duplicate, the
// inverse of the dedup -- instead of being read: the Unique's last pass rides on the first histogram.
__global__ __launch_bounds__(256) void k_radix_hist(const int* __restrict__ keys, int n, int shift, int nbits,
                                                    int* __restrict__ hist, const int* __restrict__ slots,
                                                    const int* __restrict__ sidx, int* __restrict__ keys_out,
                                                    unsigned* __restrict__ clear, int nclear,
                                                    const int* __restrict__ pos = nullptr, int* __restrict__ inv_out = nullptr,
                                                    const int64_t* __restrict__ n_dev = nullptr, int gkey = -1) {
    __shared__ int h[RNB];
    // (the look-back words the Unique's rank kernel left behind: zeroed here for the next call, no launch of their own)
    for (int j = blockIdx.x * 256 + threadIdx.x; j < nclear; j += gridDim.x * 256) clear[j] = 0;
    if (n_dev) { const int64_t nd = *n_dev; n = nd < n ? (int)(nd < 0 ? 0 : nd) : n; }      // element count known on the device only
    const int base = blockIdx.x * RT;
    if (base >= n) return;                 // (the column scan reads the rows of the live tiles only)
    const int NB = 1 << nbits;
    for (int d = threadIdx.x; d < NB; d += 256) h[d] = 0;
    __syncthreads();
    // A thread's 8 keys are fetched TOGETHER, step by step of the chain (clamped indices, values past the end dropped): as one
    // guarded block per key -- a four-deep chain of dependent loads each, stores in between -- this kernel was 32 dependent
    // round trips long (35 us in the step).  The loads of inv_out may pass the stores to it: they read first positions of keys,
    // the stores write positions of duplicates.
    constexpr int RK = RT / 256;
    int key[RK];
    const int nl = n - 1;
    if (slots) {
        // keys produced here: the group of the i-th duplicate (position pos[i]) = the inverse of the first position of
        // its key, which its scratch slot still holds; also written as the duplicate's own inverse
        int p_[RK], sx[RK], fp[RK];
#pragma unroll
        for (int k = 0; k < RK; ++k) {
            const int i = base + k * 256 + threadIdx.x;
            p_[k] = pos[i < nl ? i : nl];
        }
#pragma unroll
        for (int k = 0; k < RK; ++k) sx[k] = sidx[p_[k]];
#pragma unroll
        for (int k = 0; k < RK; ++k) fp[k] = slots[sx[k] >= 0 ? sx[k] : 0];
#pragma unroll
        for (int k = 0; k < RK; ++k) key[k] = inv_out[sx[k] >= 0 ? fp[k] : 0];      // (slot 0 may be empty: not a position)
#pragma unroll
        for (int k = 0; k < RK; ++k) {
            const int i = base + k * 256 + threadIdx.x;
            key[k] = sx[k] >= 0 ? key[k] : gkey;        // (sx < 0: a skipped negative id -- the pseudo-group behind all others)
            if (i < n) {
                inv_out[p_[k]] = sx[k] >= 0 ? key[k] : -1;
                keys_out[i] = key[k];
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < RK; ++k) {
            const int i = base + k * 256 + threadIdx.x;
            key[k] = keys[i < nl ? i : nl];
        }
    }
#pragma unroll
    for (int k = 0; k < RK; ++k) {
        const int i = base + k * 256 + threadIdx.x;
        if (i < n) atomicAdd(&h[(key[k] >> shift) & (NB - 1)], 1);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < NB; d += 256) hist[(int64_t)blockIdx.x * NB + d] = h[d];
}

// Per digit column: exclusive prefix over tiles (hist -> hscan) and the column total.  A block owns
// 32 adjacent columns (one 128-B segment per tile row) and splits the tile rows over 8 row-groups:
// each thread sums its chunk of rows, an 8-way exclusive scan over the row-groups goes through LDS,
// then each thread rewrites its chunk as running prefixes.  2^nbits / 32 blocks instead of
// 2^nbits / 256 keeps this latency-bound pass off the critical path (11.8 -> ~4 us at 208 tiles).
__global__ __launch_bounds__(256) void k_radix_colscan(const int* __restrict__ hist, int nblk, int nbits,
                                                       int* __restrict__ hscan, int* __restrict__ totals,
                                                       const int64_t* __restrict__ n_dev = nullptr) {
    __shared__ int part[8][32];
    if (n_dev) {                           // live tiles only
        const int64_t nd = *n_dev < 0 ? 0 : *n_dev;
        const int64_t live = (nd + RT - 1) / RT;
        if (live < nblk) nblk = (int)live;
    }
    const int NB = 1 << nbits;
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int d = blockIdx.x * 32 + cx;
    const int chunk = (nblk + 7) / 8;
    const int r0 = ry * chunk, r1 = (r0 + chunk < nblk) ? r0 + chunk : nblk;
    int sum = 0;
    if (d < NB) {
        int b = r0;
        for (; b + 4 <= r1; b += 4) {
            int t[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) t[k] = hist[(int64_t)(b + k) * NB + d];
            sum += t[0] + t[1] + t[2] + t[3];
        }
        for (; b < r1; ++b) sum += hist[(int64_t)b * NB + d];
    }
    part[ry][cx] = sum;
    __syncthreads();
    int run = 0;
    for (int k = 0; k < ry; ++k) run += part[k][cx];
    if (d < NB) {
        if (ry == 7) totals[d] = run + sum;
        int b = r0;
        for (; b + 4 <= r1; b += 4) {
            int t[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) t[k] = hist[(int64_t)(b + k) * NB + d];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                hscan[(int64_t)(b + k) * NB + d] = run;
                run += t[k];
            }
        }
        for (; b < r1; ++b) {
            const int t = hist[(int64_t)b * NB + d];
            hscan[(int64_t)b * NB + d] = run;
            run += t;
        }
    }
}

__global__ __launch_bounds__(256) void k_radix_scatter(const int* __restrict__ keys_in,
                                                       const int* __restrict__ vals_in, int n, int shift,
                                                       int nbits, const int* __restrict__ hist,
                                                       const int* __restrict__ totals, int* __restrict__ keys_out,
                                                       int* __restrict__ vals_out, int* __restrict__ dbase_out,
                                                       const int64_t* __restrict__ n_dev = nullptr) {
    __shared__ int cnt[4][RNB];
    if (n_dev) { const int64_t nd = *n_dev; n = nd < n ? (int)(nd < 0 ? 0 : nd) : n; }
    if ((int)blockIdx.x * RT >= n && !(dbase_out && blockIdx.x == 0)) return;
    __shared__ int dbase[RNB];
    __shared__ int sm[8];
    const int NB = 1 << nbits;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    for (int d = threadIdx.x; d < 4 * RNB; d += 256) (&cnt[0][0])[d] = 0;
    {   // exclusive scan of the digit totals (every block recomputes it: NB <= 2048 ints from L2)
        const int per = (NB + 255) / 256;      // <= 8 consecutive digits per thread
        const int d0 = threadIdx.x * per;
        int t[8], s = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = totals[d0 + k < NB ? d0 + k : NB - 1];      // (requested together; see k_radix_hist)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            t[k] = (k < per && d0 + k < NB) ? t[k] : 0;
            s += t[k];
        }
        int tot;
        int run = block_excl_scan_256(s, sm, &tot);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (k < per && d0 + k < NB) {
                dbase[d0 + k] = run;
                if (dbase_out && blockIdx.x == 0) dbase_out[d0 + k] = run;
                run += t[k];
            }
        }
    }
    __syncthreads();
    volatile int* my = cnt[w];
    const int base = blockIdx.x * RT + w * (RROUNDS * 64);
    int key[RROUNDS], pre[RROUNDS];
    const uint64_t lt = (l == 0) ? 0ull : (~0ull >> (64 - l));
    const int nl = n > 0 ? n - 1 : 0;
    int val[RROUNDS];
#pragma unroll
    for (int r = 0; r < RROUNDS; ++r) {            // the wave's 8 rounds of keys (and values) requested together
        const int i = base + r * 64 + l;
        key[r] = keys_in[i < nl ? i : nl];
        val[r] = vals_in ? vals_in[i < nl ? i : nl] : i;
    }
#pragma unroll
    for (int r = 0; r < RROUNDS; ++r) {
        const int i = base + r * 64 + l;
        const bool valid = i < n;
        key[r] = valid ? key[r] : 0;
        const int d = (key[r] >> shift) & (NB - 1);
        uint64_t m = __ballot(valid);
        for (int b = 0; b < nbits; ++b) {
            const bool bit = (d >> b) & 1;
            const uint64_t bal = __ballot(bit);
            m &= bit ? bal : ~bal;
        }
        // m: valid lanes sharing this lane's digit (garbage-free for valid lanes, which include self)
        const int rank_in = __popcll(m & lt);
        const int leader = valid ? (__ffsll((long long)m) - 1) : l;
        int old = 0;
        if (valid && l == leader) {
            old = my[d];
            my[d] = old + __popcll(m);
        }
        old = __shfl(old, leader, 64);
        pre[r] = old + rank_in;
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    for (int d = threadIdx.x; d < NB; d += 256) {
        const int c0 = cnt[0][d], c1 = cnt[1][d], c2 = cnt[2][d];
        const int b0 = dbase[d] + hist[(int64_t)blockIdx.x * NB + d];
        cnt[0][d] = b0;
        cnt[1][d] = b0 + c0;
        cnt[2][d] = b0 + c0 + c1;
        cnt[3][d] = b0 + c0 + c1 + c2;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RROUNDS; ++r) {
        const int i = base + r * 64 + l;
        if (i < n) {
            const int d = (key[r] >> shift) & (NB - 1);
            const int dst = cnt[w][d] + pre[r];
            keys_out[dst] = key[r];
            vals_out[dst] = val[r];
        }
    }
}


// Runs pass (shift, nbits) of the sort: keys_in/vals_in -> keys_out/vals_out (vals_in == nullptr
// means "identity").  hist, hscan: [nblk * 2^nbits] ints each; totals: [2^nbits]; dbase (nullable):
// [2^nbits] ints, receives the exclusive digit offsets.
inline void radix_pass(const int* kin, const int* vin, int n, int shift, int nbits, int* hist, int* hscan, int* totals,
                       int* dbase, int* kout, int* vout, hipStream_t st, const int* slots = nullptr,
                       const int* sidx = nullptr, int* keys_gen = nullptr, unsigned* clear = nullptr, int nclear = 0,
                       const int* pos = nullptr, int* inv_out = nullptr, const int64_t* n_dev = nullptr, int gkey = -1) {
    const int nblk = (int)mrec_cdiv(n, RT);
    const int NB = 1 << nbits;
    k_radix_hist<<<nblk, 256, 0, st>>>(kin, n, shift, nbits, hist, slots, sidx, keys_gen, clear, nclear, pos, inv_out, n_dev, gkey);
    k_radix_colscan<<<(NB + 31) / 32, 256, 0, st>>>(hist, nblk, nbits, hscan, totals, n_dev);
    k_radix_scatter<<<nblk, 256, 0, st>>>(kin, vin, n, shift, nbits, hscan, totals, kout, vout, dbase, n_dev);
}

}  // namespace
