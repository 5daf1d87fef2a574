// mrec_gemm.h -- the MFMA GEMM body behind DenseLayer (models/wide_deep/src/wide_and_deep.py:113-133:
// MatMul + BiasAdd + ReLU in fp16/bf16) and its two bprops, hand-written for gfx950.
//
// One workgroup = 8 waves (512 threads) owns a 256 x 256 output tile; wave (wr, wc) = (w >> 2, w & 3) owns
// 128 (P side) x 64 (Q side) of it as 8 x 4 accumulators of v_mfma_f32_16x16x32_{bf16,f16}.  The reduction
// dimension is cut in K-tiles of 64; a K-tile is staged global -> LDS by LDS-DMA (buffer_load ... lds, 16 B per
// lane) in four 16-KB pieces (P rows 0-63 / 64-127 of every wave row, Q columns 0-31 / 32-63 of every wave
// column), two LDS buffers = 128 KB.  A K-tile is computed in four phases, one 64 x 32 quadrant of the wave's
// tile each (16 MFMAs); every phase issues the LDS reads of the fragments it needs, stages ONE piece that will
// be needed five phases later, waits with a COUNTED vmcnt (4 pieces stay in flight across the barriers) and
// runs its MFMAs between two raw s_barriers.  The two wave rows run staggered by one barrier, so that on every
// SIMD one wave issues MFMAs while its partner issues LDS reads and DMA (MI355X guide, "256^2 8-phase").
//
// LDS image: 1-KB subtiles of 16 rows x 64 B, byte ^= ((byte >> 9) & 1) << 5 (conflict-free for ds_read_b128
// fragments and for ds_read_b64_tr_b16 blocks alike).  LDS-DMA writes lane-linear, so the swizzle is applied to
// the per-lane SOURCE address and again on the read.
//
// Each operand is either K-contiguous (P[p, k] / Q[q, k]: fragments by ds_read_b128) or reduction-strided
// (P[k, p] / Q[k, q]: fragments through ds_read_b64_tr_b16), chosen per operand by PT / QT:
//   forward   y  = x . W      P = x  [M, K]  (PT = 0)   Q = W  [K, N] (QT = 1)   -- W is used as stored
//   dgrad     dx = dy . W^T   P = dy [M, N]  (PT = 0)   Q = W  [K, N] (QT = 0: its rows ARE the outputs)
//   wgrad     dW = x^T . dy   P = x  [M, K]  (PT = 1)   Q = dy [M, N] (QT = 1), reduction over M, split in slabs
// A trailing partial K-tile is zero-filled by dropping the staging loads of k >= K (buffer range check); when it
// holds at most 32 k its second k-step is skipped altogether.  K-contiguous operands need K % 8 == 0.
// Accumulators hold C transposed (MFMA "A" = Q fragment, "B" = P fragment), so a lane owns 4 consecutive q of
// one p: 8-byte (16-bit output) / 16-byte (fp32 output) stores.
//
// Tile configurations (MR = 16-row repeats of the P side per wave; 8 waves as 2 x 4, Q tile always 256):
//   MR = 8   256 x 256 tile, 2 LDS buffers (128 KB), 4 phases per K-tile         -- the layers that fill the chip with it
//   MR = 4   128 x 256 tile, 3 LDS buffers (144 KB), 2 phases per K-tile         -- the narrow layers: twice the workgroups,
//            half the time per K-tile; a phase is again 4 P reps x 2 Q reps x 2 k-steps = 16 MFMAs
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mrec_dropout.h"

#ifndef MREC_GEMM_READS_8484
#define MREC_GEMM_READS_8484 0       // 1: the 8 / 4 / 8 / 4 fragment-read schedule of the 256 x 256 body (round 5: built, bit-identical, measured, not the default)
#endif

namespace mgemm {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));

#define MGEMM_LDS __attribute__((address_space(3)))

enum { EPI_FWD = 0,     // out = relu?(acc + bias[q])                      -> 16-bit
       EPI_DGRAD = 1,   // out = H[p, q] > 0 ? acc : 0 (H nullable), optional column sums over p -> 16-bit
       EPI_F32 = 2,     // out = acc                                        -> fp32 (split slabs)
       EPI_X3 = 3 };    // fp32 DenseLayer on three-part operands (mrec_gemm_x3.hip): x3_mode 1: out = relu?(acc + bias[q]);
                        // 2: out = (Hf[p, q] > 0 ? acc : 0) * x3_scale, column sums per 64 rows -> fp32, AND out's own three bf16 parts

struct Args {
    const void* P;
    const void* Q;
    void* C;
    const float* bias;      // EPI_FWD: [Qext] fp32 (nullable)
    const void* H;          // EPI_DGRAD: [Pext, ldc] 16-bit activations of the layer below (nullable: no mask)
    float* colsum_ws;       // EPI_DGRAD: [nTp, Qext] per-tile-row column sums (nullable)
    int64_t ldp, ldq, ldc;  // row strides in elements
    int Pext, Qext, K;      // output extents (P side, Q side) and the reduction extent
    int nTp, nTq;           // 256-tiles per side
    int kt_per_slab;        // K-tiles per split slab (grid has nTp * nTq * S workgroups; no split: all of them)
    int64_t slab_stride;    // elements between consecutive slabs of C
    int relu;               // EPI_FWD
    // VAR & 4 -- a SEGMENTED reduction of six segments: segment s reads P from part (seg_codeP >> 2 s) & 3 of its buffer (parts are
    // seg_partP bytes apart) and Q from part (seg_codeQ >> 2 s) & 3 (+ the tile's offset inside the part).  What an fp32 GEMM needs when
    // each fp32 operand is held as THREE bf16 parts (x = x1 + x2 + x3, 8 mantissa bits each): the six products a1 b1, a1 b2,
    // a2 b1, a1 b3, a2 b2, a3 b1 are six segments of one reduction into the same fp32 accumulators (mrec_gemm_x3.hip).
    // (part numbers packed in a word, not arrays: a kernel that selects between two Args keeps them in SGPRs)
    int seg_tiles;          // K-tiles per segment (K = 6 * 64 * seg_tiles); 0: an ordinary reduction
    uint32_t seg_codeP, seg_codeQ, seg_partP, seg_partQ;
    int seg_inter;          // 1: K-tile t of the launch is tile t / 6 of segment t % 6 (the six products of one stretch of the reduction
                            // back to back: every operand tile is fetched once from HBM and found in L2 the other times); 0: segment
                            // after segment
    // EPI_X3 (H is then fp32 [Pext, ldh]; colsum_ws [ceil(Pext / 64), Qext])
    int x3_mode;
    float x3_scale;
    int64_t ldh;
    uint16_t* parts;        // nullable: [3][parts_stride] bf16 images of out, rows of parts_ld elements (zero padding is the caller's)
    int64_t parts_ld, parts_stride;
    int64_t rangeP, rangeQ; // VAR & 4: bytes the P / Q buffer resources span (all parts)
    DropArgs drop;          // thresh != 0: Dropout on the layer input this launch produces (EPI_FWD: C is the next layer's input,
                            // masked and scaled after the rounding; EPI_DGRAD: C is the gradient of this layer's dropped-out
                            // input -- scaled, and masked by the hash when there is no H whose zeros already carry the mask)
};

template <bool F16> struct Elem;
template <> struct Elem<false> {
    typedef bf16x8_t v8;
    static __device__ __forceinline__ f32x4_t mfma(u32x4_t a, u32x4_t b, f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
        typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
        bf16x2_t v = {(__bf16)lo, (__bf16)hi};
        return __builtin_bit_cast(uint32_t, v);
    }
    static __device__ __forceinline__ float widen(uint32_t bits16) { return __uint_as_float(bits16 << 16); }
};
template <> struct Elem<true> {
    typedef f16x8_t v8;
    static __device__ __forceinline__ f32x4_t mfma(u32x4_t a, u32x4_t b, f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
        typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
        f16x2_t v = {(_Float16)lo, (_Float16)hi};
        return __builtin_bit_cast(uint32_t, v);
    }
    static __device__ __forceinline__ float widen(uint32_t bits16) {
        return (float)__builtin_bit_cast(_Float16, (uint16_t)bits16);
    }
};

constexpr int kThreads = 512;
template <int MR> struct Lds { static constexpr int bytes = MR == 8 ? 131072 : 147456; };
constexpr uint32_t kOob = 0x80000000u;      // voffset beyond any buffer: the load is dropped (zero fill), nothing is fetched

__device__ __forceinline__ constexpr int slot_off(int buf, int type) { return (buf * 4 + type) * 16384; }

// The body: one workgroup computes tile `bid` of the `nblk` tiles of problem `a`; smem = the kernel's LDS (Lds<MR>::bytes).
template <int MR, bool PT, bool QT, int EPI, bool F16, int VAR = 0>
__device__ __forceinline__ void gemm256_body(const Args& a, const int bid, const int nblk, MGEMM_LDS char* const smem) {
    typedef Elem<F16> E;
    static_assert(MR == 8 || MR == 4, "tile configurations: MR = 8 (256 x 256) or MR = 4 (128 x 256)");
    constexpr int WP = MR * 16;          // P rows per wave
    constexpr int BP = 2 * WP;           // P rows per workgroup
    // These waves outrank whatever shares the CU with them (the step's plan kernels on the side branch): base priority 2,
    // 3 inside the MFMA clusters.
    if (!(VAR & 2)) __builtin_amdgcn_s_setprio(2);
    const int tid = threadIdx.x, l = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 2, wc = w & 3;

    // ---- workgroup -> (tq, tp, z): neighbours in the remapped order share an XCD (its L2 then serves the shared
    // operand panel); the remap is bijective for any grid size.
    int tq, tp, z;
    {
        const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
        const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
        tq = swz % a.nTq;
        const int rest = swz / a.nTq;
        tp = rest % a.nTp;
        z = rest / a.nTp;
    }
    const int Ttot = (a.K + 63) >> 6;
    const int kt0 = z * a.kt_per_slab;
    int T = min(a.kt_per_slab, Ttot - kt0);
    if (T < 0) T = 0;
    const int krem = a.K & 63;                        // k in the trailing partial K-tile (0: none)
    const bool ktail = krem != 0;

    constexpr bool SEG = (VAR & 4) != 0;
    const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(a.P), 0, (int)(SEG ? a.rangeP : (PT ? (int64_t)a.K : (int64_t)a.Pext) * a.ldp * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rQ = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(a.Q), 0, (int)(SEG ? a.rangeQ : (QT ? (int64_t)a.K : (int64_t)a.Qext) * a.ldq * 2), 0x00020000);

    // ---- per-lane source offsets of the staging loads (bytes); [half][e]
    uint32_t voffP[2][2], voffQ[2][2];        // ... and for the trailing partial K-tile (loads of k >= K dropped)
    uint32_t voffPt[2][2], voffQt[2][2];
    {
        const int srow = l >> 2, chunk = (l & 3) ^ ((l >> 5) << 1);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                if (!PT) {
                    const int rowp = tp * BP + (w >> 2) * WP + h * 64 + (w & 3) * 16 + srow;
                    voffP[h][e] = rowp < a.Pext ? (uint32_t)(((int64_t)rowp * a.ldp + e * 32 + chunk * 8) * 2) : kOob;
                    voffPt[h][e] = (e * 32 + chunk * 8 < krem) ? voffP[h][e] : kOob;
                } else {
                    const int ip = tp * BP + (w >> 2) * WP + h * 64 + e * 32 + chunk * 8;
                    const int mp = (w & 3) * 16 + srow;
                    voffP[h][e] = ip < a.Pext ? (uint32_t)(((int64_t)mp * a.ldp + ip) * 2) : kOob;
                    voffPt[h][e] = mp < krem ? voffP[h][e] : kOob;
                }
                if (!QT) {
                    const int rowq = tq * 256 + (w >> 1) * 64 + h * 32 + (w & 1) * 16 + srow;
                    voffQ[h][e] = rowq < a.Qext ? (uint32_t)(((int64_t)rowq * a.ldq + e * 32 + chunk * 8) * 2) : kOob;
                    voffQt[h][e] = (e * 32 + chunk * 8 < krem) ? voffQ[h][e] : kOob;
                } else {
                    const int jq = tq * 256 + (w >> 1) * 64 + h * 32 + chunk * 8;
                    const int mq = (2 * (w & 1) + e) * 16 + srow;
                    voffQ[h][e] = jq < a.Qext ? (uint32_t)(((int64_t)mq * a.ldq + jq) * 2) : kOob;
                    voffQt[h][e] = mq < krem ? voffQ[h][e] : kOob;
                }
            }
    }
    const uint32_t ktP = PT ? (uint32_t)(64 * a.ldp * 2) : 128u;     // soffset step per K-tile
    const uint32_t ktQ = QT ? (uint32_t)(64 * a.ldq * 2) : 128u;
    // byte offset of K-tile tg of the launch (SEG: inside its segment's part of the operand)
    auto soffOf = [&](int tg, bool isP) -> uint32_t {
        if constexpr (SEG) {
            int sg, ti;
            if (a.seg_inter) {
                ti = tg / 6; sg = tg - 6 * ti;
                if (ti >= a.seg_tiles) { ti = a.seg_tiles - 1; sg = 5; }      // (prefetch past the end: its loads are dropped anyway)
            } else {
                sg = tg / a.seg_tiles;
                sg = sg > 5 ? 5 : sg;
                ti = tg - sg * a.seg_tiles;
            }
            return (((isP ? a.seg_codeP : a.seg_codeQ) >> (2 * sg)) & 3u) * (isP ? a.seg_partP : a.seg_partQ) + (uint32_t)ti * (isP ? ktP : ktQ);
        } else {
            return (uint32_t)tg * (isP ? ktP : ktQ);
        }
    };
    // piece types: 0 = P rows 0-63 of each wave row, 1 = Q cols 0-31 of each wave column, 2 = Q cols 32-63, 3 = P rows 64-127
#define MG_STAGE(TYPE, BUF, tt)                                                                                   \
    do {                                                                                                          \
        const int tt_ = (tt);                                                                                     \
        const bool live_ = tt_ < T;                                                                               \
        constexpr bool isP_ = (TYPE) == 0 || (TYPE) == 3;                                                         \
        constexpr int h_ = ((TYPE) >= 2) ? 1 : 0;                                                                 \
        const uint32_t soff_ = soffOf(kt0 + tt_, isP_);                                                           \
        MGEMM_LDS char* dst_ = (MGEMM_LDS char*)smem + slot_off(BUF, TYPE) + w * 2048;                           \
        const bool last_ = ktail && (kt0 + tt_ == Ttot - 1);                                                      \
        const uint32_t v0_ = !live_ ? kOob : last_ ? (isP_ ? voffPt[h_][0] : voffQt[h_][0]) : (isP_ ? voffP[h_][0] : voffQ[h_][0]); \
        const uint32_t v1_ = !live_ ? kOob : last_ ? (isP_ ? voffPt[h_][1] : voffQt[h_][1]) : (isP_ ? voffP[h_][1] : voffQ[h_][1]); \
        if (!((VAR & 8) && tt_ >= 2)) {                                                                           \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(isP_ ? rP : rQ, (MGEMM_LDS void*)dst_, 16, v0_, (VAR & 32) ? 0u : soff_, 0, 0);    \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(isP_ ? rP : rQ, (MGEMM_LDS void*)(dst_ + 1024), 16, v1_, (VAR & 32) ? 0u : soff_, 0, 0); \
        }                                                                                                         \
    } while (0)

    // ---- per-lane LDS read offsets
    uint32_t rdP0, rdP1, rdQ0, rdQ1;      // K-contiguous operands use *0 only; strided: index = 16-column half of the subtile row
    {
        const int r = l & 15, c = l >> 4;
        const uint32_t lane = (uint32_t)((64 * r + 16 * c) ^ ((r >> 3) << 5));
        const int lg = l >> 4, rr = 8 * (lg & 1) + ((l & 15) >> 2), pp = l & 3;
        const uint32_t b = (uint32_t)(64 * rr + 8 * pp);
        const uint32_t x0 = rr >= 8 ? 32u : 0u, x1 = rr >= 8 ? 0u : 32u;
        if (!PT) {
            rdP0 = lane + wr * 8192;          // subtile ((wr*4 + mi)*2 + ks)
            rdP1 = 0;
        } else {
            rdP0 = b + x0 + (lg >> 1) * 2048 + wr * 8192;     // subtile (wr*4 + 2ks + (lg>>1))*2 + (mi>>1)
            rdP1 = b + x1 + (lg >> 1) * 2048 + wr * 8192;
        }
        if (!QT) {
            rdQ0 = lane + wc * 4096;          // subtile ((wc*2 + nj)*2 + ks)
            rdQ1 = 0;
        } else {
            rdQ0 = b + x0 + (lg >> 1) * 1024 + wc * 4096;     // subtile wc*4 + 2ks + (lg>>1)
            rdQ1 = b + x1 + (lg >> 1) * 1024 + wc * 4096;
        }
    }

    // (Skipping the MFMAs of the wave quadrants that lie outside the output in the tiles that hang over its edge -- 2080 =
    // 8 x 256 + 32: the ninth column tile of layer 0's input gradient -- was measured: the launch takes as long; such a
    // workgroup's K-tile takes what its staging and barriers take.)
    // Harness-only variants (tools/probes/dense_gemm_test.hip; no product kernel sets these bits): VAR & 8 stages the first two
    // K-tiles only, VAR & 16 reads the fragments of the first K-tile only, VAR & 32 stages every K-tile from the first one's
    // addresses (served by the caches), VAR & 64 adds up where a phase's cycles go (s_memtime per segment; sums to colsum_ws).
    bool rd_done = false;
    uint32_t tseg[8][4] = {}, tlast = 0;          // [phase + 4 * buffer][segment: issue + waits, first barrier, MFMAs, second barrier]
    u32x4_t fP[4][2], fQ0[2][2], fQ1[2][2];       // fragments [rep][ks]
    u32x4_t fPb[4];                               // MR = 8: the second k-step's P fragments of a K-tile's FIRST half (read a phase early)
    f32x4_t acc[MR][4];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    auto ld128 = [&](uint32_t off) -> u32x4_t { return *(const MGEMM_LDS u32x4_t*)((MGEMM_LDS char*)smem + off); };
    auto ldtr = [&](uint32_t off) -> u32x4_t {
        s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((MGEMM_LDS s16x4_t*)((MGEMM_LDS char*)smem + off));
        s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((MGEMM_LDS s16x4_t*)((MGEMM_LDS char*)smem + off + 256));
        u32x2_t a2 = __builtin_bit_cast(u32x2_t, lo), b2 = __builtin_bit_cast(u32x2_t, hi);
        return u32x4_t{a2[0], a2[1], b2[0], b2[1]};
    };

#define MG_READ_P(BUF, H)                                                                              \
    do {                                                                                               \
        if ((VAR & 16) && rd_done) break;                                                              \
        _Pragma("unroll") for (int mi_ = 0; mi_ < 4; ++mi_) _Pragma("unroll") for (int ks_ = 0; ks_ < 2; ++ks_) { \
            if (!PT) fP[mi_][ks_] = ld128(slot_off(BUF, (H) ? 3 : 0) + (mi_ * 2 + ks_) * 1024 + rdP0); \
            else fP[mi_][ks_] = ldtr(slot_off(BUF, (H) ? 3 : 0) + (4 * ks_ + (mi_ >> 1)) * 1024 + ((mi_ & 1) ? rdP1 : rdP0)); \
        }                                                                                              \
    } while (0)
    // the P fragments of ONE k-step of half H (MR = 8 schedule below): ks 0 into fP[.][0], ks 1 of the first half into fPb
#define MG_READ_P_KS0(BUF)                                                                             \
    do {                                                                                               \
        if ((VAR & 16) && rd_done) break;                                                              \
        _Pragma("unroll") for (int mi_ = 0; mi_ < 4; ++mi_) {                                          \
            if (!PT) fP[mi_][0] = ld128(slot_off(BUF, 0) + (mi_ * 2 + 0) * 1024 + rdP0);               \
            else fP[mi_][0] = ldtr(slot_off(BUF, 0) + (4 * 0 + (mi_ >> 1)) * 1024 + ((mi_ & 1) ? rdP1 : rdP0)); \
        }                                                                                              \
    } while (0)
#define MG_READ_P_KS1B(BUF)                                                                            \
    do {                                                                                               \
        if ((VAR & 16) && rd_done) break;                                                              \
        _Pragma("unroll") for (int mi_ = 0; mi_ < 4; ++mi_) {                                          \
            if (!PT) fPb[mi_] = ld128(slot_off(BUF, 0) + (mi_ * 2 + 1) * 1024 + rdP0);                 \
            else fPb[mi_] = ldtr(slot_off(BUF, 0) + (4 * 1 + (mi_ >> 1)) * 1024 + ((mi_ & 1) ? rdP1 : rdP0)); \
        }                                                                                              \
    } while (0)
#define MG_READ_Q(BUF, H, F)                                                                           \
    do {                                                                                               \
        if ((VAR & 16) && rd_done) break;                                                              \
        _Pragma("unroll") for (int nj_ = 0; nj_ < 2; ++nj_) _Pragma("unroll") for (int ks_ = 0; ks_ < 2; ++ks_) { \
            if (!QT) F[nj_][ks_] = ld128(slot_off(BUF, (H) ? 2 : 1) + (nj_ * 2 + ks_) * 1024 + rdQ0);  \
            else F[nj_][ks_] = ldtr(slot_off(BUF, (H) ? 2 : 1) + (2 * ks_) * 1024 + (nj_ ? rdQ1 : rdQ0)); \
        }                                                                                              \
    } while (0)
    // one quadrant: 4 P reps x 2 Q reps x 2 k-steps (the second k-step is skipped in a half tail tile)
#define MG_MFMA(MH, NH, F, HALF)                                                                       \
    do {                                                                                               \
        if (!(VAR & 2)) __builtin_amdgcn_s_setprio(3);                                                 \
        _Pragma("unroll") for (int mi_ = 0; mi_ < 4; ++mi_) _Pragma("unroll") for (int nj_ = 0; nj_ < 2; ++nj_) \
            acc[(MH) * 4 + mi_][(NH) * 2 + nj_] = E::mfma(F[nj_][0], fP[mi_][0], acc[(MH) * 4 + mi_][(NH) * 2 + nj_]); \
        if (!(HALF)) {                                                                                 \
            _Pragma("unroll") for (int mi_ = 0; mi_ < 4; ++mi_) _Pragma("unroll") for (int nj_ = 0; nj_ < 2; ++nj_) \
                acc[(MH) * 4 + mi_][(NH) * 2 + nj_] = E::mfma(F[nj_][1], fP[mi_][1], acc[(MH) * 4 + mi_][(NH) * 2 + nj_]); \
        }                                                                                              \
        if (!(VAR & 2)) __builtin_amdgcn_s_setprio(2);                                                 \
    } while (0)
    // ... of the first half (MH = 0) with the second k-step's P fragments in fPb
#define MG_MFMA0(NH, F, HALF)                                                                          \
    do {                                                                                               \
        if (!(VAR & 2)) __builtin_amdgcn_s_setprio(3);                                                 \
        _Pragma("unroll") for (int mi_ = 0; mi_ < 4; ++mi_) _Pragma("unroll") for (int nj_ = 0; nj_ < 2; ++nj_) \
            acc[mi_][(NH) * 2 + nj_] = E::mfma(F[nj_][0], fP[mi_][0], acc[mi_][(NH) * 2 + nj_]);      \
        if (!(HALF)) {                                                                                 \
            _Pragma("unroll") for (int mi_ = 0; mi_ < 4; ++mi_) _Pragma("unroll") for (int nj_ = 0; nj_ < 2; ++nj_) \
                acc[mi_][(NH) * 2 + nj_] = E::mfma(F[nj_][1], fPb[mi_], acc[mi_][(NH) * 2 + nj_]);    \
        }                                                                                              \
        if (!(VAR & 2)) __builtin_amdgcn_s_setprio(2);                                                 \
    } while (0)
#define MG_SYNC_PRE()                                     \
    do {                                                  \
        if constexpr ((VAR & 4096) != 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); \
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); \
        if constexpr ((VAR & 2048) != 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        __builtin_amdgcn_sched_barrier(0);                \
        __builtin_amdgcn_s_barrier();                     \
        __builtin_amdgcn_sched_barrier(0);                \
    } while (0)
#define MG_SYNC_POST()                     \
    do {                                   \
        __builtin_amdgcn_sched_barrier(0); \
        __builtin_amdgcn_s_barrier();      \
        __builtin_amdgcn_sched_barrier(0); \
    } while (0)
#define MG_T(PH, SG)                                                              \
    do {                                                                              \
        if constexpr ((VAR & 64) != 0) {                                              \
            __builtin_amdgcn_sched_barrier(0);                                        \
            const uint32_t now_ = (uint32_t)__builtin_amdgcn_s_memtime();             \
            tseg[PH][SG] += now_ - tlast;                                             \
            tlast = now_;                                                             \
            __builtin_amdgcn_sched_barrier(0);                                        \
        }                                                                             \
    } while (0)
#define MG_PHASE_SYNC_MFMA(PH, MH, NH, F, half_)     \
    do {                                             \
        if constexpr ((VAR & 64) != 0) {             \
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); \
            MG_T(PH, 0);                             \
            __builtin_amdgcn_s_barrier();            \
            MG_T(PH, 1);                             \
            MG_MFMA(MH, NH, F, half_);               \
            MG_T(PH, 2);                             \
            __builtin_amdgcn_s_barrier();            \
            MG_T(PH, 3);                             \
        } else {                                     \
            MG_SYNC_PRE();                           \
            MG_MFMA(MH, NH, F, half_);               \
            MG_SYNC_POST();                          \
        }                                            \
    } while (0)
    // (VAR & 128, harness only, WRONG results: the same 24 fragment reads dealt out 6 / 6 / 6 / 6 over the four phases instead of
    // 12 / 4 / 8 / 0 -- what balancing the load segments would be worth, before building it)
#define MG_RD1P(BUF, H, mi_, ks_) fP[mi_][ks_] = ld128(slot_off(BUF, (H) ? 3 : 0) + ((mi_) * 2 + (ks_)) * 1024 + rdP0)
#define MG_KTILE_BAL(BUF, t)                         \
    do {                                             \
        const bool half_ = false;                    \
        MG_READ_Q(BUF, 0, fQ0);                      \
        MG_RD1P(BUF, 0, 0, 0); MG_RD1P(BUF, 0, 0, 1); \
        MG_STAGE(2, (BUF) ^ 1, (t) + 1);             \
        MG_PHASE_SYNC_MFMA(0 + 4 * (BUF), 0, 0, fQ0, half_); \
        MG_READ_Q(BUF, 1, fQ1);                      \
        MG_RD1P(BUF, 0, 1, 0); MG_RD1P(BUF, 0, 1, 1); \
        MG_STAGE(3, (BUF) ^ 1, (t) + 1);             \
        MG_PHASE_SYNC_MFMA(1 + 4 * (BUF), 0, 1, fQ1, half_); \
        MG_RD1P(BUF, 1, 0, 0); MG_RD1P(BUF, 1, 0, 1); MG_RD1P(BUF, 1, 1, 0); MG_RD1P(BUF, 1, 1, 1); MG_RD1P(BUF, 1, 2, 0); MG_RD1P(BUF, 1, 2, 1); \
        MG_STAGE(0, BUF, (t) + 2);                   \
        MG_PHASE_SYNC_MFMA(2 + 4 * (BUF), 1, 1, fQ1, half_); \
        MG_RD1P(BUF, 1, 3, 0); MG_RD1P(BUF, 1, 3, 1); MG_RD1P(BUF, 0, 2, 0); MG_RD1P(BUF, 0, 2, 1); MG_RD1P(BUF, 0, 3, 0); MG_RD1P(BUF, 0, 3, 1); \
        MG_STAGE(1, BUF, (t) + 2);                   \
        MG_PHASE_SYNC_MFMA(3 + 4 * (BUF), 1, 0, fQ0, half_); \
    } while (0)
    // (VAR & 16384, harness only, WRONG results: 8 / 4 / 8 / 4 -- what moving the second k-step's P fragments of phase 0 into the
    // reads of the phase before it would be worth; that form needs no change of the staging order or of the waits)
#define MG_KTILE_B8(BUF, t)                          \
    do {                                             \
        const bool half_ = false;                    \
        MG_READ_Q(BUF, 0, fQ0);                      \
        MG_RD1P(BUF, 0, 0, 0); MG_RD1P(BUF, 0, 1, 0); MG_RD1P(BUF, 0, 2, 0); MG_RD1P(BUF, 0, 3, 0); \
        MG_STAGE(2, (BUF) ^ 1, (t) + 1);             \
        MG_PHASE_SYNC_MFMA(0, 0, 0, fQ0, half_);     \
        MG_READ_Q(BUF, 1, fQ1);                      \
        MG_STAGE(3, (BUF) ^ 1, (t) + 1);             \
        MG_PHASE_SYNC_MFMA(1, 0, 1, fQ1, half_);     \
        MG_READ_P(BUF, 1);                           \
        MG_STAGE(0, BUF, (t) + 2);                   \
        MG_PHASE_SYNC_MFMA(2, 1, 1, fQ1, half_);     \
        MG_RD1P(BUF, 0, 0, 1); MG_RD1P(BUF, 0, 1, 1); MG_RD1P(BUF, 0, 2, 1); MG_RD1P(BUF, 0, 3, 1); \
        MG_STAGE(1, BUF, (t) + 2);                   \
        MG_PHASE_SYNC_MFMA(3, 1, 0, fQ0, half_);     \
    } while (0)
#define MG_PHASE_SYNC_MFMA0(PH, NH, F, half_)        \
    do {                                             \
        if constexpr ((VAR & 64) != 0) {             \
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); \
            MG_T(PH, 0);                             \
            __builtin_amdgcn_s_barrier();            \
            MG_T(PH, 1);                             \
            MG_MFMA0(NH, F, half_);                  \
            MG_T(PH, 2);                             \
            __builtin_amdgcn_s_barrier();            \
            MG_T(PH, 3);                             \
        } else {                                     \
            MG_SYNC_PRE();                           \
            MG_MFMA0(NH, F, half_);                  \
            MG_SYNC_POST();                          \
        }                                            \
    } while (0)
    // An 8 / 4 / 8 / 4 read schedule (round 5; built, measured, NOT the default: -DMREC_GEMM_READS_8484=1, or VAR & 65536 in the harness).
    // The shipped schedule reads 12 / 4 / 8 / 0 fragments in a K-tile's four phases, and the phase with twelve is the long pole of the
    // ping-pong (profiles/r05_gemm_ablation.txt).  Here the first half's P fragments of the SECOND k-step are read a phase early -- in
    // phase 3 of the K-tile before, from the other buffer: that piece was staged six phases earlier, the wait that retires it sits in
    // front of phase 2's first barrier, and the piece is restaged in the next tile's phase 2, behind its last read in phase 0 as
    // before: no wait, no staging order and no barrier changes -- into four registers of their own (fPb; 216 -> 230-237 VGPRs).
    // The same MFMAs in the same order: every output bit equal to the shipped schedule's over forward / input gradient / weight
    // gradient, ragged shapes and K tails, six launches each (`dense_gemm_test compare`).  Alone: layer 0's forward 62.3 -> 61.6 us,
    // K = 1024 37.1 -> 36.3, K = 8320 212.5 -> 209 (half of what the emulation with misplaced reads promised).  In the step: 0.629-0.633
    // -> 0.636-0.638 ms, three A/B pairs on one box -- slower; the launches beside the GEMMs find 20 registers less per lane on every
    // SIMD.  Not kept.
#define MG_KTILE_8484(BUF, t)                        \
    do {                                             \
        const bool half_ = !(VAR & 1024) && ktail && krem <= 32 && (kt0 + (t)) == Ttot - 1;    \
        MG_READ_Q(BUF, 0, fQ0);                      \
        MG_READ_P_KS0(BUF);                          \
        MG_STAGE(2, (BUF) ^ 1, (t) + 1);             \
        MG_PHASE_SYNC_MFMA0(0 + 4 * (BUF), 0, fQ0, half_); \
        MG_READ_Q(BUF, 1, fQ1);                      \
        MG_STAGE(3, (BUF) ^ 1, (t) + 1);             \
        MG_PHASE_SYNC_MFMA0(1 + 4 * (BUF), 1, fQ1, half_); \
        MG_READ_P(BUF, 1);                           \
        MG_STAGE(0, BUF, (t) + 2);                   \
        MG_PHASE_SYNC_MFMA(2 + 4 * (BUF), 1, 1, fQ1, half_); \
        MG_READ_P_KS1B((BUF) ^ 1);                   \
        MG_STAGE(1, BUF, (t) + 2);                   \
        MG_PHASE_SYNC_MFMA(3 + 4 * (BUF), 1, 0, fQ0, half_); \
        if (VAR & 16) rd_done = true;                \
    } while (0)
    // the four phases of K-tile t living in LDS buffer BUF (the shipped schedule)
#define MG_KTILE(BUF, t)                         \
    do {                                             \
        const bool half_ = !(VAR & 1024) && ktail && krem <= 32 && (kt0 + (t)) == Ttot - 1;    \
        MG_READ_Q(BUF, 0, fQ0);                      \
        MG_READ_P(BUF, 0);                           \
        MG_STAGE(2, (BUF) ^ 1, (t) + 1);             \
        MG_PHASE_SYNC_MFMA(0 + 4 * (BUF), 0, 0, fQ0, half_); \
        MG_READ_Q(BUF, 1, fQ1);                      \
        MG_STAGE(3, (BUF) ^ 1, (t) + 1);             \
        MG_PHASE_SYNC_MFMA(1 + 4 * (BUF), 0, 1, fQ1, half_); \
        MG_READ_P(BUF, 1);                           \
        MG_STAGE(0, BUF, (t) + 2);                   \
        MG_PHASE_SYNC_MFMA(2 + 4 * (BUF), 1, 1, fQ1, half_); \
        MG_STAGE(1, BUF, (t) + 2);                   \
        MG_PHASE_SYNC_MFMA(3 + 4 * (BUF), 1, 0, fQ0, half_); \
        if (VAR & 16) rd_done = true;                \
    } while (0)

    if constexpr (MR == 8) {
        // ---- prologue: six pieces in flight, the first two landed
        MG_STAGE(0, 0, 0);
        MG_STAGE(1, 0, 0);
        MG_STAGE(2, 0, 0);
        MG_STAGE(3, 0, 0);
        MG_STAGE(0, 1, 1);
        MG_STAGE(1, 1, 1);
        MG_SYNC_PRE();
        if (!(VAR & 1) && wr == 1) __builtin_amdgcn_s_barrier();       // stagger: wave row 1 runs one barrier behind wave row 0

        [[maybe_unused]] uint64_t tc0 = 0, tr0 = 0;
        if constexpr ((VAR & 64) != 0) {
            tc0 = __builtin_amdgcn_s_memtime();
            tr0 = __builtin_amdgcn_s_memrealtime();
            tlast = (uint32_t)tc0;
        }
        if constexpr ((VAR & 128) != 0 && !PT && !QT) {
            MG_READ_P(0, 0);
            for (int t = 0; t < T; t += 2) {
                MG_KTILE_BAL(0, t);
                if (t + 1 < T) MG_KTILE_BAL(1, t + 1);
            }
        } else if constexpr ((VAR & 16384) != 0 && !PT && !QT) {
            MG_READ_P(0, 0);
            for (int t = 0; t < T; t += 2) {
                MG_KTILE_B8(0, t);
                if (t + 1 < T) MG_KTILE_B8(1, t + 1);
            }
        } else if constexpr ((VAR & 256) != 0) {
            for (int t = 0; t < T; t += 4) {
                MG_KTILE(0, t);
                if (t + 1 < T) MG_KTILE(1, t + 1);
                if (t + 2 < T) MG_KTILE(0, t + 2);
                if (t + 3 < T) MG_KTILE(1, t + 3);
            }
        } else if constexpr ((VAR & 512) != 0) {
            for (int t = 0; t < T; t += 8) {
                MG_KTILE(0, t);
                if (t + 1 < T) MG_KTILE(1, t + 1);
                if (t + 2 < T) MG_KTILE(0, t + 2);
                if (t + 3 < T) MG_KTILE(1, t + 3);
                if (t + 4 < T) MG_KTILE(0, t + 4);
                if (t + 5 < T) MG_KTILE(1, t + 5);
                if (t + 6 < T) MG_KTILE(0, t + 6);
                if (t + 7 < T) MG_KTILE(1, t + 7);
            }
        } else if constexpr ((VAR & 65536) != 0 || MREC_GEMM_READS_8484) {
            MG_READ_P_KS1B(0);                     // K-tile 0's (landed with the prologue's first two pieces)
            for (int t = 0; t < T; t += 2) {
                MG_KTILE_8484(0, t);
                if (t + 1 < T) MG_KTILE_8484(1, t + 1);
            }
        } else {
            for (int t = 0; t < T; t += 2) {
                MG_KTILE(0, t);
                if (t + 1 < T) MG_KTILE(1, t + 1);
            }
        }
        if constexpr ((VAR & 64) != 0) {
            if (l == 0 && a.colsum_ws != nullptr) {          // [workgroup][wave][phase][segment] cycle sums
#pragma unroll
                for (int ph = 0; ph < 8; ++ph)
#pragma unroll
                    for (int sg = 0; sg < 4; ++sg) a.colsum_ws[((int64_t)bid * 8 + w) * 36 + ph * 4 + sg] = (float)tseg[ph][sg];
                a.colsum_ws[((int64_t)bid * 8 + w) * 36 + 32] = (float)(__builtin_amdgcn_s_memtime() - tc0);         // shader cycles
                a.colsum_ws[((int64_t)bid * 8 + w) * 36 + 33] = (float)(__builtin_amdgcn_s_memrealtime() - tr0);     // 100 MHz ticks
            }
        }
    } else {
        // ---- 128 x 256 tile: a K-tile is three 16-KB pieces (P, Q cols 0-31, Q cols 32-63 of every wave column) in one of
        // three LDS buffers, and two phases: phase 0 reads P and Q_n0 and stages P, Q_n0 of K-tile t + 2 (into the buffer
        // K-tile t - 1 was read from two phases ago), phase 1 reads Q_n1 and stages Q_n1 of K-tile t + 2.  Loads issued
        // after the piece the NEXT phase needs: 10 at the wait of phase 0, 8 at the wait of phase 1.
        constexpr uint32_t KTB = 49152;
#define MG4_STAGE(TYPE, bo_, tt)                                                                                  \
    do {                                                                                                          \
        const int tt_ = (tt);                                                                                     \
        const bool live_ = tt_ < T;                                                                               \
        constexpr bool isP_ = (TYPE) == 0;                                                                        \
        constexpr int h_ = (TYPE) == 2 ? 1 : 0;                                                                   \
        const uint32_t soff_ = soffOf(kt0 + tt_, isP_);                                                           \
        MGEMM_LDS char* dst_ = smem + (bo_) + (TYPE) * 16384 + w * 2048;                                         \
        const bool last_ = ktail && (kt0 + tt_ == Ttot - 1);                                                      \
        const uint32_t v0_ = !live_ ? kOob : last_ ? (isP_ ? voffPt[0][0] : voffQt[h_][0]) : (isP_ ? voffP[0][0] : voffQ[h_][0]); \
        const uint32_t v1_ = !live_ ? kOob : last_ ? (isP_ ? voffPt[0][1] : voffQt[h_][1]) : (isP_ ? voffP[0][1] : voffQ[h_][1]); \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(isP_ ? rP : rQ, (MGEMM_LDS void*)dst_, 16, v0_, soff_, 0, 0);    \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(isP_ ? rP : rQ, (MGEMM_LDS void*)(dst_ + 1024), 16, v1_, soff_, 0, 0); \
    } while (0)
#define MG4_READ_P(bo_)                                                                                \
    do {                                                                                               \
        _Pragma("unroll") for (int mi_ = 0; mi_ < 4; ++mi_) _Pragma("unroll") for (int ks_ = 0; ks_ < 2; ++ks_) { \
            if (!PT) fP[mi_][ks_] = ld128((bo_) + (mi_ * 2 + ks_) * 1024 + rdP0);                      \
            else fP[mi_][ks_] = ldtr((bo_) + (4 * ks_ + (mi_ >> 1)) * 1024 + ((mi_ & 1) ? rdP1 : rdP0)); \
        }                                                                                              \
    } while (0)
#define MG4_READ_Q(bo_, H, F)                                                                          \
    do {                                                                                               \
        _Pragma("unroll") for (int nj_ = 0; nj_ < 2; ++nj_) _Pragma("unroll") for (int ks_ = 0; ks_ < 2; ++ks_) { \
            if (!QT) F[nj_][ks_] = ld128((bo_) + ((H) ? 2 : 1) * 16384 + (nj_ * 2 + ks_) * 1024 + rdQ0); \
            else F[nj_][ks_] = ldtr((bo_) + ((H) ? 2 : 1) * 16384 + (2 * ks_) * 1024 + (nj_ ? rdQ1 : rdQ0)); \
        }                                                                                              \
    } while (0)
#define MG4_SYNC_PRE(N)                                          \
    do {                                                         \
        asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory");    \
        __builtin_amdgcn_sched_barrier(0);                       \
        __builtin_amdgcn_s_barrier();                            \
        __builtin_amdgcn_sched_barrier(0);                       \
    } while (0)
        MG4_STAGE(0, 0u, 0);
        MG4_STAGE(1, 0u, 0);
        MG4_STAGE(2, 0u, 0);
        MG4_STAGE(0, KTB, 1);
        MG4_STAGE(1, KTB, 1);
        MG4_STAGE(2, KTB, 1);
        MG4_SYNC_PRE(8);                                               // P, Q_n0 of K-tile 0 have landed
        if (!(VAR & 1) && wr == 1) __builtin_amdgcn_s_barrier();       // stagger
        uint32_t bo = 0, bs = 2 * KTB;                                 // buffer of K-tile t / of K-tile t + 2
        for (int t = 0; t < T; ++t) {
            const bool half_ = ktail && krem <= 32 && (kt0 + t) == Ttot - 1;
            MG4_READ_Q(bo, 0, fQ0);
            MG4_READ_P(bo);
            MG4_STAGE(0, bs, t + 2);
            MG4_STAGE(1, bs, t + 2);
            MG4_SYNC_PRE(10);
            MG_MFMA(0, 0, fQ0, half_);
            MG_SYNC_POST();
            MG4_READ_Q(bo, 1, fQ1);
            MG4_STAGE(2, bs, t + 2);
            MG4_SYNC_PRE(8);
            MG_MFMA(0, 1, fQ1, half_);
            MG_SYNC_POST();
            bo = bo == 2 * KTB ? 0u : bo + KTB;
            bs = bs == 2 * KTB ? 0u : bs + KTB;
        }
#undef MG4_STAGE
#undef MG4_READ_P
#undef MG4_READ_Q
#undef MG4_SYNC_PRE
    }
    if (!(VAR & 1) && wr == 0) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- epilogue: lane owns p = p0 + mi*16 + (l & 15), q = q0 + ni*16 + 4*(l >> 4) + {0..3}
    const int p0 = tp * BP + wr * WP + (l & 15);
    const int q0 = tq * 256 + wc * 64 + 4 * (l >> 4);
    if (EPI == EPI_F32) {
        float* C = (float*)a.C + (int64_t)z * a.slab_stride;
#pragma unroll
        for (int mi = 0; mi < MR; ++mi) {
            const int p = p0 + mi * 16;
            if (p < a.Pext) {
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    const int q = q0 + ni * 16;
                    if (q < a.Qext) {
                        float* dst = C + (int64_t)p * a.ldc + q;
                        if ((a.ldc & 3) == 0) {
                            *(f32x4_t*)dst = acc[mi][ni];
                        } else {            // rows that are only 8-byte aligned (Deep&Cross's 1170-column input gradient)
                            typedef float f32x2_t __attribute__((ext_vector_type(2)));
                            *(f32x2_t*)dst = f32x2_t{acc[mi][ni][0], acc[mi][ni][1]};
                            if (q + 2 < a.Qext) *(f32x2_t*)(dst + 2) = f32x2_t{acc[mi][ni][2], acc[mi][ni][3]};
                        }
                    }
                }
            }
        }
    } else if (EPI == EPI_X3) {
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        float* C = (float*)a.C;
        const float* Hf = (const float*)a.H;
        const bool fwd = a.x3_mode == 1;
        float bq[4][4];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = q0 + ni * 16 + r;
                bq[ni][r] = (fwd && a.bias != nullptr && q < a.Qext) ? a.bias[q] : 0.f;
            }
        // a wave's rows are one (MR = 4) or two (MR = 8) of the 64-row groups the column sums are kept for
        float cs[MR / 4][4][4];
#pragma unroll
        for (int g = 0; g < MR / 4; ++g)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r) cs[g][ni][r] = 0.f;
        const bool v16 = (a.ldc & 3) == 0, h16 = (a.ldh & 3) == 0;
        const bool xdrop = fwd && a.drop.thresh != 0;          // Dropout on the NEXT layer's input = this output (fp32: x * (1 / keep) or 0)
        const uint64_t xkey = xdrop ? drop_key(a.drop) : 0ull;
#pragma unroll
        for (int mi = 0; mi < MR; ++mi) {
            const int p = p0 + mi * 16;
            const bool pv = p < a.Pext;
            f32x4_t hv[4];
            if (!fwd && Hf != nullptr) {
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    const int q = q0 + ni * 16;
                    const bool ok = pv && q < a.Qext;
                    const float* src = Hf + (ok ? (int64_t)p * a.ldh + q : (int64_t)0);
                    if (h16) {
                        hv[ni] = *(const f32x4_t*)src;
                    } else {
                        const f32x2_t lo = *(const f32x2_t*)src;
                        const f32x2_t hi = (ok && q + 2 < a.Qext) ? *(const f32x2_t*)(src + 2) : f32x2_t{0.f, 0.f};
                        hv[ni] = f32x4_t{lo[0], lo[1], hi[0], hi[1]};
                    }
                }
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int q = q0 + ni * 16;
                const bool ok = pv && q < a.Qext;
                f32x4_t v = acc[mi][ni];
                const uint64_t qd = xdrop ? drop_quad(xkey, a.drop.row0 + p, a.Qext, q) : 0ull;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (fwd) {
                        v[r] += bq[ni][r];
                        if (a.relu) v[r] = v[r] > 0.f ? v[r] : 0.f;
                        if (xdrop) v[r] = drop_keep(qd, r, a.drop.thresh) ? v[r] * a.drop.scale : 0.0f;
                    } else {
                        if (Hf != nullptr && !(hv[ni][r] > 0.f)) v[r] = 0.f;
                        v[r] *= a.x3_scale;
                    }
                    if (q + r >= a.Qext) v[r] = 0.f;           // (the parts image is zero past the last column)
                }
                if (ok) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) cs[mi / 4][ni][r] += v[r];
                    float* dst = C + (int64_t)p * a.ldc + q;
                    if (v16) {
                        *(f32x4_t*)dst = v;
                    } else {
                        *(f32x2_t*)dst = f32x2_t{v[0], v[1]};
                        if (q + 2 < a.Qext) *(f32x2_t*)(dst + 2) = f32x2_t{v[2], v[3]};
                    }
                    if (a.parts != nullptr) {
                        // x = x1 + x2 + x3, each part the bf16 rounding of what the parts before it leave (residuals exact in fp32)
                        uint32_t pk[3][2];
#pragma unroll
                        for (int hh = 0; hh < 2; ++hh) {
                            float x0 = v[2 * hh], x1 = v[2 * hh + 1];
                            uint32_t w0[3], w1[3];
#pragma unroll
                            for (int t = 0; t < 3; ++t) {
                                const __bf16 b0 = (__bf16)x0, b1 = (__bf16)x1;
                                w0[t] = __builtin_bit_cast(uint16_t, b0); w1[t] = __builtin_bit_cast(uint16_t, b1);
                                x0 -= (float)b0; x1 -= (float)b1;
                            }
#pragma unroll
                            for (int t = 0; t < 3; ++t) pk[t][hh] = w0[t] | (w1[t] << 16);
                        }
                        uint16_t* pd = a.parts + (int64_t)p * a.parts_ld + q;
#pragma unroll
                        for (int t = 0; t < 3; ++t) *(u32x2_t*)(pd + t * a.parts_stride) = u32x2_t{pk[t][0], pk[t][1]};
                    }
                }
            }
        }
        if (!fwd && a.colsum_ws != nullptr) {
            // over the 16 lanes that share l >> 4 (fixed xor tree): a wave owns its 64-row groups x 64 columns outright
#pragma unroll
            for (int g = 0; g < MR / 4; ++g) {
                const int prow = tp * BP + wr * WP + g * 64;
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float s_ = cs[g][ni][r];
                        s_ += __shfl_xor(s_, 1, 64);
                        s_ += __shfl_xor(s_, 2, 64);
                        s_ += __shfl_xor(s_, 4, 64);
                        s_ += __shfl_xor(s_, 8, 64);
                        const int q = q0 + ni * 16 + r;
                        if ((l & 15) == 0 && prow < a.Pext && q < a.Qext) a.colsum_ws[(int64_t)(prow >> 6) * a.Qext + q] = s_;
                    }
            }
        }
    } else {
        uint16_t* C = (uint16_t*)a.C;
        const bool drop = a.drop.thresh != 0;
        const uint64_t dkey = drop ? drop_key(a.drop) : 0ull;
        float bq[4][4];
        if (EPI == EPI_FWD) {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int q = q0 + ni * 16 + r;
                    bq[ni][r] = (a.bias != nullptr && q < a.Qext) ? a.bias[q] : 0.f;
                }
        }
        float cs[4][4];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) cs[ni][r] = 0.f;
        // EPI_DGRAD: the activations the ReLU mask is read from, ALL requested before the first is used (an out-of-range element
        // reads H[0] and is never stored).  Loaded one by one in front of their stores, each load was followed by a wait for
        // everything in flight -- the store before it included: 32 memory round trips in a row per lane.
        u32x2_t hmask[MR][4];
        if (EPI == EPI_DGRAD && a.H != nullptr) {
#pragma unroll
            for (int mi = 0; mi < MR; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    const int p = p0 + mi * 16, q = q0 + ni * 16;
                    const bool ok = p < a.Pext && q < a.Qext;
                    hmask[mi][ni] = *(const u32x2_t*)((const uint16_t*)a.H + (ok ? (int64_t)p * a.ldc + q : (int64_t)0));
                }
        }
#pragma unroll
        for (int mi = 0; mi < MR; ++mi) {
            const int p = p0 + mi * 16;
            const bool pv = p < a.Pext;
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int q = q0 + ni * 16;
                const bool ok = pv && q < a.Qext;
                f32x4_t v = acc[mi][ni];
                if (EPI == EPI_FWD) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[r] += bq[ni][r];
                        if (a.relu) v[r] = v[r] > 0.f ? v[r] : 0.f;
                    }
                }
                if (EPI == EPI_DGRAD && drop) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] *= a.drop.scale;
                }
                u32x2_t o = {E::pack2(v[0], v[1]), E::pack2(v[2], v[3])};
                if (drop && (EPI == EPI_FWD || a.H == nullptr)) {
                    const uint64_t qd = drop_quad(dkey, a.drop.row0 + p, a.Qext, q);
                    if (EPI == EPI_FWD) {        // Dropout acts on the stored (rounded) activation: x * (1 / keep), rounded again
                        o[0] = E::pack2(E::widen(o[0] & 0xFFFFu) * a.drop.scale, E::widen(o[0] >> 16) * a.drop.scale);
                        o[1] = E::pack2(E::widen(o[1] & 0xFFFFu) * a.drop.scale, E::widen(o[1] >> 16) * a.drop.scale);
                    }
                    if (!drop_keep(qd, 0, a.drop.thresh)) o[0] &= 0xFFFF0000u;
                    if (!drop_keep(qd, 1, a.drop.thresh)) o[0] &= 0x0000FFFFu;
                    if (!drop_keep(qd, 2, a.drop.thresh)) o[1] &= 0xFFFF0000u;
                    if (!drop_keep(qd, 3, a.drop.thresh)) o[1] &= 0x0000FFFFu;
                }
                if (EPI == EPI_DGRAD) {
                    if (a.H != nullptr) {
                        const u32x2_t hb = hmask[mi][ni];
                        // activation > 0 <=> its 16-bit pattern is a positive number (sign clear, not zero)
                        if (!((hb[0] & 0xFFFFu) - 1u < 0x7FFFu)) o[0] &= 0xFFFF0000u;
                        if (!((hb[0] >> 16) - 1u < 0x7FFFu)) o[0] &= 0x0000FFFFu;
                        if (!((hb[1] & 0xFFFFu) - 1u < 0x7FFFu)) o[1] &= 0xFFFF0000u;
                        if (!((hb[1] >> 16) - 1u < 0x7FFFu)) o[1] &= 0x0000FFFFu;
                    }
                    if (a.colsum_ws != nullptr && ok) {      // sums of the ROUNDED gradients, fp32
                        cs[ni][0] += E::widen(o[0] & 0xFFFFu);
                        cs[ni][1] += E::widen(o[0] >> 16);
                        cs[ni][2] += E::widen(o[1] & 0xFFFFu);
                        cs[ni][3] += E::widen(o[1] >> 16);
                    }
                }
                if constexpr ((VAR & 8192) != 0) asm volatile("" ::"v"(o));      // (harness: no output stores; the values stay live)
                else if (ok) *(u32x2_t*)(C + (int64_t)p * a.ldc + q) = o;
            }
        }
        if (EPI == EPI_DGRAD && a.colsum_ws != nullptr) {
            // over the 16 lanes that share l >> 4 (fixed xor tree), then over the two wave rows through LDS
            MGEMM_LDS float* red = (MGEMM_LDS float*)smem;      // [2][256]
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float s = cs[ni][r];
                    s += __shfl_xor(s, 1, 64);
                    s += __shfl_xor(s, 2, 64);
                    s += __shfl_xor(s, 4, 64);
                    s += __shfl_xor(s, 8, 64);
                    if ((l & 15) == 0) red[wr * 256 + wc * 64 + ni * 16 + 4 * (l >> 4) + r] = s;
                }
            __syncthreads();
            if (tid < 256) {
                const int q = tq * 256 + tid;
                if (q < a.Qext) a.colsum_ws[(int64_t)tp * a.Qext + q] = red[tid] + red[256 + tid];
            }
        }
    }
#undef MG_STAGE
#undef MG_READ_P
#undef MG_READ_Q
#undef MG_MFMA
#undef MG_SYNC_PRE
#undef MG_SYNC_POST
#undef MG_KTILE
#undef MG_T
#undef MG_RD1P
#undef MG_KTILE_BAL
#undef MG_KTILE_8484
#undef MG_READ_P_KS0
#undef MG_READ_P_KS1B
#undef MG_MFMA0
#undef MG_PHASE_SYNC_MFMA0
#undef MG_KTILE_B8
#undef MG_PHASE_SYNC_MFMA
}

template <bool PT, bool QT, int EPI, bool F16, int VAR = 0, int MR = 8>
__global__ __launch_bounds__(kThreads, 2) void k_gemm256(const Args a) {
    __shared__ __attribute__((aligned(1024))) char smem[Lds<MR>::bytes];
    gemm256_body<MR, PT, QT, EPI, F16, VAR>(a, blockIdx.x, gridDim.x, (MGEMM_LDS char*)smem);
}

// Both bprops of one DenseLayer in ONE launch: the two problems are independent (both read dy) and, for the narrow
// layers, neither fills the chip alone.  Workgroups [0, n1) run the problem whose workgroups take longer (more K-tiles
// each; dispatched first so that the short ones pack behind them), the rest the other one.
// ... plus, behind them, the workgroups of up to two MORE weight-gradient problems (other layers' -- the tail launch of
// mrec_tail.hip leaves its two layers' weight gradients to be computed, and alone each is a latency-bound 10-16 us launch).
struct ExtraW { Args a[2]; int n[2]; };
template <bool F16, int MRD = 8, int MRW = 8>
__global__ __launch_bounds__(kThreads, 2) void k_gemm256_bwd(const Args ad, const Args aw, const int n1, const int wfirst, const ExtraW ex) {
    __shared__ __attribute__((aligned(1024))) char smem[Lds<(MRD < MRW ? MRD : MRW)>::bytes];      // the 128 x 256 config needs more
    const int nx = ex.n[0] + ex.n[1];
    int b = blockIdx.x;
    const int n2 = (int)gridDim.x - nx - n1;
    const bool second = b >= n1;
    if (b < n1 + n2 && second == (wfirst != 0)) {
        gemm256_body<MRD, false, false, EPI_DGRAD, F16>(ad, second ? b - n1 : b, second ? n2 : n1, (MGEMM_LDS char*)smem);
        return;
    }
    // a weight-gradient workgroup: of this layer, or of one of the extra problems (uniform selects: the arguments stay in SGPRs)
    Args sel = aw;
    int nb = second ? n2 : n1;
    if (b >= n1 + n2) {
        b -= n1 + n2;
        if (b < ex.n[0]) { sel = ex.a[0]; nb = ex.n[0]; }
        else { b -= ex.n[0]; sel = ex.a[1]; nb = ex.n[1]; }
    } else if (second) b -= n1;
    gemm256_body<MRW, true, true, EPI_F32, F16>(sel, b, nb, (MGEMM_LDS char*)smem);
}

}  // namespace mgemm
