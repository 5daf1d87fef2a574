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

__global__ __launch_bounds__(256) void k_radix_hist(const int* __restrict__ keys, int n, int shift, int nbits,
                                                    int* __restrict__ hist) {
    __shared__ int h[RNB];
    const int NB = 1 << nbits;
    for (int d = threadIdx.x; d < NB; d += 256) h[d] = 0;
    __syncthreads();
    const int base = blockIdx.x * RT;
#pragma unroll
    for (int k = 0; k < RT / 256; ++k) {
        const int i = base + k * 256 + threadIdx.x;
        if (i < n) atomicAdd(&h[(keys[i] >> shift) & (NB - 1)], 1);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < NB; d += 256) hist[(int64_t)blockIdx.x * NB + d] = h[d];
}

// One block of 1024 threads: per digit, exclusive prefix over tiles (in place), then exclusive
// prefix of digit totals into dbase.
__global__ __launch_bounds__(1024) void k_radix_scan(int* __restrict__ hist, int nblk, int nbits,
                                                     int* __restrict__ dbase) {
    __shared__ int wsum[16];
    const int NB = 1 << nbits;
    const int per = (NB + 1023) / 1024;  // 1 or 2 digits per thread
    const int d0 = threadIdx.x * per;
    int run0 = 0, run1 = 0;
    if (d0 < NB) {
        if (per == 1) {
            for (int b = 0; b < nblk; ++b) {
                int* p = hist + (int64_t)b * NB + d0;
                const int t = *p; *p = run0; run0 += t;
            }
        } else {
            for (int b = 0; b < nblk; ++b) {
                int2* p = (int2*)(hist + (int64_t)b * NB + d0);
                const int2 t = *p;
                *p = make_int2(run0, run1);
                run0 += t.x; run1 += t.y;
            }
        }
    }
    const int mine = run0 + run1;
    int incl = wave_incl_scan(mine);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (l == 63) wsum[w] = incl;
    __syncthreads();
    int base = 0;
    for (int i = 0; i < w; ++i) base += wsum[i];
    const int excl = base + incl - mine;
    if (d0 < NB) {
        dbase[d0] = excl;
        if (per == 2) dbase[d0 + 1] = excl + run0;
    }
}

__global__ __launch_bounds__(256) void k_radix_scatter(const int* __restrict__ keys_in,
                                                       const int* __restrict__ vals_in, int n, int shift,
                                                       int nbits, const int* __restrict__ hist,
                                                       const int* __restrict__ dbase, int* __restrict__ keys_out,
                                                       int* __restrict__ vals_out) {
    __shared__ int cnt[4][RNB];
    const int NB = 1 << nbits;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    for (int d = threadIdx.x; d < 4 * RNB; d += 256) (&cnt[0][0])[d] = 0;
    __syncthreads();
    volatile int* my = cnt[w];
    const int base = blockIdx.x * RT + w * (RROUNDS * 64);
    int key[RROUNDS], pre[RROUNDS];
    const uint64_t lt = (l == 0) ? 0ull : (~0ull >> (64 - l));
#pragma unroll
    for (int r = 0; r < RROUNDS; ++r) {
        const int i = base + r * 64 + l;
        const bool valid = i < n;
        key[r] = valid ? keys_in[i] : 0;
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
            vals_out[dst] = vals_in ? vals_in[i] : i;
        }
    }
}


// Runs pass (shift, nbits) of the sort: keys_in/vals_in -> keys_out/vals_out (vals_in == nullptr
// means "identity").  hist: [nblk * 2^nbits] ints, dbase: [2^nbits] ints (exclusive digit offsets
// on return).
inline void radix_pass(const int* kin, const int* vin, int n, int shift, int nbits, int* hist, int* dbase, int* kout,
                       int* vout, hipStream_t st) {
    const int nblk = (int)mrec_cdiv(n, RT);
    k_radix_hist<<<nblk, 256, 0, st>>>(kin, n, shift, nbits, hist);
    k_radix_scan<<<1, 1024, 0, st>>>(hist, nblk, nbits, dbase);
    k_radix_scatter<<<nblk, 256, 0, st>>>(kin, vin, n, shift, nbits, hist, dbase, kout, vout);
}

}  // namespace
