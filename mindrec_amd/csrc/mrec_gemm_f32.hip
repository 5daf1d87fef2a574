// mrec_gemm_f32.hip -- DenseLayer in fp32 (convert_dtype=False: Deep&Cross, models/deep_and_cross/src/deep_and_cross.py:94-114,
// 293-309) and its two bprops on the fp32-input matrix instruction of gfx950, v_mfma_f32_32x32x2_f32.
//
// The fp32 MFMA runs at the fp32 vector rate (157 TFLOP/s, 1/16 of the 16-bit forms) and is bit-for-bit a k-ordered chain of
// fmaf's: one rounding per product, no wider accumulation.  At that rate nothing else on the CU is near its limit -- what
// costs is an idle matrix pipe: rounds of workgroups that do not fill the chip, and the waits around the staging.  So:
//   * a 128 x 128 output tile per 256-thread workgroup, two workgroups per CU (one computes across the other's barrier);
//     four waves as 2 x 2, each 64 x 64 = 2 x 2 accumulators of 32 x 32;
//   * a 32-deep K-tile in LDS in REDUCTION-MAJOR order (rows = k, 132 floats apart; the instruction wants
//     A[i = lane & 31][k = lane >> 5], so a fragment is one conflict-free ds_read_b32), two buffers, and a staging
//     pipeline whose every instruction sits in a gap between MFMAs: while K-tile j is multiplied, K-tile j + 1 is requested
//     from memory (first half of the MFMAs) and written to the other buffer (second half) -- ONE barrier per K-tile;
//   * staging by buffer loads: per-lane byte offsets computed once, the K-tile's advance in the scalar offset, rows and k
//     outside the operand switched off through the hardware range check (rows need only 8-byte alignment -- the DCN input is
//     39 x 30 = 1170 floats wide -- so VEC = 4, 2 or 1 floats per load is picked per operand such that no vector straddles an
//     edge);
//   * a tile that hangs over the output's edge skips the MFMAs of its 32 x 32 blocks that lie wholly outside.
//
// Operands are either reduction-contiguous (X[r, k]: a row of the operand is a row of the matrix: transposed on its way into
// LDS, VEC ds_write_b32 per load) or reduction-strided (Y[k, c]: written as loaded):
//   forward   y  = x . W        A = x  [M, K] contiguous      B = W  [K, N] strided
//   dgrad     dx = dy . W^T     A = dy [M, N] contiguous      B = W  [K, N] contiguous (its rows ARE the outputs)
//   wgrad     dW = x^T . dy     A = x  [M, K] strided         B = dy [M, N] strided, reduction over the batch, split in slabs
// Epilogues: bias + ReLU; mask by the activation below > 0 + column sums per 64 output rows (the layer below's BiasAdd
// bprop); plain fp32 slabs.
//
// Measured (MI355X, Deep&Cross layer 1: 39.3 GFLOP per launch; tools/dcn_bench.py, tools/probes/mfma_f32_shape_probe.hip): a bare
// loop of this instruction with its fragments re-read from LDS sustains 154 TFLOP/s (98 %, either shape -- 32x32x2 or 16x16x4).
// This kernel: forward 373 us (105 TFLOP/s), weight gradient 353 (111), input gradient 400 (98: 1280 tiles = 2.5 rounds of the
// chip).  How it got there: plain structure (loads, barrier, stores, barrier, MFMAs) 506 / 474 / 681 us; two LDS buffers, one
// barrier, branch-free loads 437 / 433 / 428; staging spread between the MFMAs 423 / 401 / 395; epilogue loads up front 408 / 397 /
// 397; buffer loads (2.7 -> 0.75 VALU instructions per MFMA in the K-loop) 373 / 400 / 353.  Ablations before the last step: no
// global loads in the K-loop 352 us, no LDS stores / barrier either 340, no epilogue stores 329; the chip held 2.07 GHz.
#include <type_traits>
#include "mrec_common.h"

namespace gf32 {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32, LD = 132;      // tile, K-tile; LDS row stride in floats (528 B: 16-byte aligned rows)
constexpr int TILE_F = BK * LD;                 // floats per operand tile in LDS
constexpr int LDS_BYTES = 4 * TILE_F * (int)sizeof(float);
constexpr int SLOTS = 512;                      // workgroups the chip holds at 2 per CU
enum { EPI_FWD = 0, EPI_DGRAD = 1, EPI_PLAIN = 2 };

struct Args {
    const float* A; const float* B; float* C;
    int64_t lda, ldb, ldc;
    int M, N, K;               // output extents and the reduction extent
    int tiles_m, tiles_n;
    int k_per_slab;            // reduction elements per split slab (multiple of BK; >= K: no split)
    int64_t slab_stride;       // floats between consecutive slabs of C
    const float* bias;         // EPI_FWD: [N] (nullable)
    int relu;
    const float* H; int64_t ldh;   // EPI_DGRAD: [M, N] activations of the layer below (nullable: no mask)
    float* colsum;             // EPI_DGRAD: [ceil(M / 64), N] column sums per 64 output rows (nullable)
};

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// one K-tile of an operand, global -> registers -> LDS, one load at a time (the K-loop spreads them between its MFMAs).
// KC: stored [R, K] (reduction contiguous), R = the tile's ROWS output rows / columns; else stored [K, R].  VEC floats per load;
// the host guarantees that no vector straddles an edge.  Loads are BUFFER loads: the lane's byte offset inside K-tile 0 is
// computed once (a row outside the operand gets an offset beyond any buffer: the hardware range check returns zeros and
// fetches nothing), the K-tile's advance is the instruction's scalar offset, and the one K-tile that can hold k beyond the
// reduction extent has a second offset set with those loads switched off -- so a staged load costs one select between the two
// sets (a wave-uniform condition) and no address arithmetic, and nothing needs zeroing on the way into LDS.
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t kOob = 0x80000000u;

template <bool KC, int VEC, int ROWS>
struct Stage {
    static constexpr int NV = ROWS * BK / 256 / VEC;       // loads per thread and K-tile
    static constexpr int QK = BK / VEC, QR = ROWS / VEC;   // loads per row (KC) / per k-row
    float v[NV][VEC];
    uint32_t off[NV], offl[NV];      // byte offset of load `it` in K-tile 0; the same with the k >= k_end loads of the LAST K-tile off
    __device__ __forceinline__ void init(int64_t ldx, int r0, int R, int kbeg, int k_last_valid, int t) {
        const int kq = KC ? (t % QK) * VEC : t / QR;       // k of this thread's loads inside a K-tile: kq + (256 / QR) * it (strided)
#pragma unroll
        for (int it = 0; it < NV; ++it) {
            const int r = KC ? r0 + t / QK + (256 / QK) * it : r0 + (t % QR) * VEC;
            const int kl = kq + (KC ? 0 : (256 / QR) * it);
            const int64_t e = KC ? (int64_t)r * ldx + kbeg + kl : (int64_t)(kbeg + kl) * ldx + r;
            off[it] = r < R ? (uint32_t)(e * 4) : kOob;
            offl[it] = kl < k_last_valid ? off[it] : kOob;
        }
    }
    // soff: byte offset of the K-tile against K-tile 0 (wave-uniform); last: it is the workgroup's last K-tile (wave-uniform)
    template <int IT>
    __device__ __forceinline__ void load_one(__amdgpu_buffer_rsrc_t rs, uint32_t soff, bool last) {
        const uint32_t vo = last ? offl[IT] : off[IT];
        if (VEC == 4) {
            const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, soff, 0);
            v[IT][0] = __uint_as_float(x[0]); v[IT][1 % VEC] = __uint_as_float(x[1]); v[IT][2 % VEC] = __uint_as_float(x[2]); v[IT][3 % VEC] = __uint_as_float(x[3]);
        } else if (VEC == 2) {
            const u32x2 x = __builtin_amdgcn_raw_buffer_load_b64(rs, vo, soff, 0);
            v[IT][0] = __uint_as_float(x[0]); v[IT][1 % VEC] = __uint_as_float(x[1]);
        } else {
            v[IT][0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, vo, soff, 0));
        }
    }
    template <int IT>
    __device__ __forceinline__ void store_one(float* __restrict__ S, int t) const {      // S: [BK][LD] reduction-major
        if (KC) {
            const int q = t % QK, rr = t / QK + (256 / QK) * IT;
#pragma unroll
            for (int c = 0; c < VEC; ++c) S[(q * VEC + c) * LD + rr] = v[IT][c];
        } else {
            const int q = t % QR, kk = t / QR + (256 / QR) * IT;
            float* d = S + kk * LD + q * VEC;
            if (VEC == 4) *(float4*)d = make_float4(v[IT][0], v[IT][1 % VEC], v[IT][2 % VEC], v[IT][3 % VEC]);
            else if (VEC == 2) *(float2*)d = make_float2(v[IT][0], v[IT][1 % VEC]);
            else *d = v[IT][0];
        }
    }
};

// workgroup barrier that waits for this wave's LDS traffic only: the global loads of the K-tile after next stay in flight
// across it (__syncthreads waits for vmcnt(0) as well)
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <bool AKC, bool BKC, int EPI, int VA, int VB>
__global__ __launch_bounds__(256, 2) void k_gemm_f32(const Args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];      // As[2] | Bs[2]
    constexpr int MI = BM / 64;
    float* const As = smem;
    float* const Bs = smem + 2 * TILE_F;
    const int t = threadIdx.x, l = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = w >> 1, wc = w & 1;
    // workgroup -> (tn, tm, z); column tiles fastest: neighbours share the A panel
    int bid = blockIdx.x;
    const int tn = bid % a.tiles_n; bid /= a.tiles_n;
    const int tm = bid % a.tiles_m;
    const int z = bid / a.tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = z * a.k_per_slab;
    const int kend = min(a.K, kbeg + a.k_per_slab);
    const int T = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;

    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // 32 x 32 blocks of this wave that hold any output at all (wave-uniform)
    bool mv[MI], nv[2];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) mv[mi] = m0 + wr * (BM / 2) + mi * 32 < a.M;
#pragma unroll
    for (int nj = 0; nj < 2; ++nj) nv[nj] = n0 + wc * 64 + nj * 32 < a.N;
    const bool full = (m0 + BM <= a.M) && (n0 + BN <= a.N);

    typedef Stage<AKC, VA, BM> SA;
    typedef Stage<BKC, VB, BN> SB;
    constexpr int NA = SA::NV, NT = SA::NV + SB::NV;       // staging items per thread and K-tile: A's loads, then B's
    SA sa;
    SB sb;
    const int k_last_valid = kend - (kbeg + (T - 1) * BK);       // k inside the last K-tile (1 .. BK)
    sa.init(a.lda, m0, a.M, kbeg, k_last_valid, t);
    sb.init(a.ldb, n0, a.N, kbeg, k_last_valid, t);
    // the operands as buffers of exactly their extent (rows x stride, the last row without its padding)
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.A), 0, (int)((((int64_t)(AKC ? a.M : a.K) - 1) * a.lda + (AKC ? a.K : a.M)) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.B), 0, (int)((((int64_t)(BKC ? a.N : a.K) - 1) * a.ldb + (BKC ? a.K : a.N)) * 4), 0x00020000);
    const uint32_t advA = (AKC ? (uint32_t)BK : (uint32_t)(BK * a.lda)) * 4u, advB = (BKC ? (uint32_t)BK : (uint32_t)(BK * a.ldb)) * 4u;
    auto load_item = [&](auto I, int tile) {
        constexpr int i = decltype(I)::value;
        if constexpr (i < NA) sa.template load_one<i>(rA, advA * (uint32_t)tile, tile == T - 1);
        else sb.template load_one<i - NA>(rB, advB * (uint32_t)tile, tile == T - 1);
    };
    auto store_item = [&](auto I, int buf) {
        constexpr int i = decltype(I)::value;
        if constexpr (i < NA) sa.template store_one<i>(As + buf * TILE_F, t);
        else sb.template store_one<i - NA>(Bs + buf * TILE_F, t);
    };
    // ---- prologue: K-tile 0 in LDS
    if (T > 0) {
        static_for<0, NT>([&](auto I) { load_item(I, 0); });
        static_for<0, NT>([&](auto I) { store_item(I, 0); });
    }
    __syncthreads();

    const int ai = wr * (BM / 2) + (l & 31), bj = wc * 64 + (l & 31), kh = l >> 5;
    // ---- K-loop.  K-tile j is multiplied from buffer j & 1 in 16 steps of 2 k; the loads of K-tile j + 1 are issued in the
    // gaps between the MFMAs of steps 0-7, one staged load per gap, and written to the other buffer in the gaps of steps 8-15
    // (a load has eight steps -- ~4000 cycles of MFMAs -- to land): everything but the MFMAs runs in their shadow, and ONE
    // barrier per K-tile closes it.
    auto kloop = [&](auto FULLC) {
    constexpr bool FULL = decltype(FULLC)::value;      // the tile lies wholly inside the output: no per-block tests
    for (int j = 0; j < T; ++j) {
        const float* ar = As + (j & 1) * TILE_F + kh * LD + ai;
        const float* br = Bs + (j & 1) * TILE_F + kh * LD + bj;
        float fa[2][MI], fb[2][2];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) fa[0][mi] = ar[mi * 32];
#pragma unroll
        for (int nj = 0; nj < 2; ++nj) fb[0][nj] = br[nj * 32];
        static_for<0, BK / 2>([&](auto Sx) {
            constexpr int s = decltype(Sx)::value, cur = s & 1, nxt = cur ^ 1;
            if constexpr (s + 1 < BK / 2) {      // fragments of step s + 1 are requested before the MFMAs of step s
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) fa[nxt][mi] = ar[(2 * s + 2) * LD + mi * 32];
#pragma unroll
                for (int nj = 0; nj < 2; ++nj) fb[nxt][nj] = br[(2 * s + 2) * LD + nj * 32];
            }
            __builtin_amdgcn_sched_barrier(0);      // (left alone the scheduler sinks the reads to just in front of their MFMAs)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int nj = 0; nj < 2; ++nj)
                    if (FULL || (mv[mi] && nv[nj]))
                        acc[mi][nj] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][mi], fb[cur][nj], acc[mi][nj], 0, 0, 0);
            // the staging work of this gap
            constexpr int h = BK / 4;                                                   // steps per half
            constexpr int lo = (s % h) * NT / h, hi = (s % h + 1) * NT / h;
            // (unconditional: what is loaded behind the last K-tile -- zeros, or the next slab's rows -- goes into a buffer
            // nobody reads, and the loop body stays one basic block, which is what lets the waits be counted exactly)
            if constexpr (s < h) static_for<lo, hi>([&](auto I) { load_item(I, j + 1); });
            else static_for<lo, hi>([&](auto I) { store_item(I, (j + 1) & 1); });
            __builtin_amdgcn_sched_barrier(0);
        });
        lds_barrier();
    }
    };
    if (full) kloop(std::true_type{});
    else kloop(std::false_type{});

    // ---- epilogue: lane owns column j = n0 + wc*64 + nj*32 + (l & 31), rows i = m0 + wr*64 + mi*32 + (r & 3) + 8 (r >> 2) + 4 kh.
    // Everything the stores depend on -- the bias, the activations the ReLU mask is read from -- is requested up front and
    // consumed in straight-line code (an element outside the output reads element 0 and is never stored): a load first used
    // inside a conditional store's block is waited for THERE, with a wait that also covers every store issued before it -- 64
    // memory round trips in a row per lane (the forward kernel lost 45 us per workgroup to it).
    // (The mask values of an input-gradient tile are requested 16 at a time, in front of the 16 stores they gate.  Because those
    // stores are guarded, the compiler's counted waits assume none of them was issued, so the later waits of a group also wait
    // for a store eight back.  All 64 requested up front and pinned in front of the first store -- no wait between stores --
    // was measured: 417-420 us against 391-398 for Deep&Cross layer 1; the loads' latency is then exposed once per tile.)
    float* C = a.C + (EPI == EPI_PLAIN ? (int64_t)z * a.slab_stride : 0);
    float cs[MI][2];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) cs[mi][0] = cs[mi][1] = 0.0f;
    float bvs[2] = {0.0f, 0.0f};
    if (EPI == EPI_FWD && a.bias != nullptr) {
#pragma unroll
        for (int nj = 0; nj < 2; ++nj) {
            const int j = n0 + wc * 64 + nj * 32 + (l & 31);
            bvs[nj] = a.bias[j < a.N ? j : 0];
        }
        asm volatile("" : "+v"(bvs[0]), "+v"(bvs[1]));      // consumed HERE: the one wait for them sits in front of the stores
    }
#pragma unroll
    for (int nj = 0; nj < 2; ++nj) {
        const int j = n0 + wc * 64 + nj * 32 + (l & 31);
        const bool jok = j < a.N;
        const float bv = bvs[nj];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            float hv[16];
            if (EPI == EPI_DGRAD && a.H != nullptr) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int i = m0 + wr * (BM / 2) + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    hv[r] = a.H[(jok && i < a.M) ? (int64_t)i * a.ldh + j : (int64_t)0];
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = m0 + wr * (BM / 2) + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                float v = acc[mi][nj][r];
                if (EPI == EPI_FWD) {
                    v = v + bv;
                    if (a.relu) v = v > 0.0f ? v : 0.0f;
                }
                const bool ok = jok && i < a.M;
                if (EPI == EPI_DGRAD) {
                    if (a.H != nullptr && !(hv[r] > 0.0f)) v = 0.0f;
                    if (ok) cs[mi][nj] += v;     // rows in register order, then the fixed tree below
                }
                if (full || ok) C[(int64_t)i * a.ldc + j] = v;
            }
        }
    }
    if (EPI == EPI_DGRAD && a.colsum != nullptr) {
        // per 32-row block: the two lane halves (rows 4 kh); per 64 rows (a wave's): block 0 + block 1
#pragma unroll
        for (int nj = 0; nj < 2; ++nj) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) cs[mi][nj] += __shfl_xor(cs[mi][nj], 32, 64);
            const int jl = wc * 64 + nj * 32 + (l & 31);
            const int row = 2 * tm + wr;
            if (kh == 0 && n0 + jl < a.N && (int64_t)row * 64 < a.M) a.colsum[(int64_t)row * a.N + n0 + jl] = cs[0][nj] + cs[1][nj];
        }
    }
}

// widest aligned load for rows `ld` apart whose contiguous extent is ext_c: no vector may straddle the extent's end
inline int vec_of(const float* p, int64_t ld, int ext_c) {
    const uintptr_t u = (uintptr_t)p;
    if ((u & 15) == 0 && ld % 4 == 0 && ext_c % 4 == 0) return 4;
    if ((u & 7) == 0 && ld % 2 == 0 && ext_c % 2 == 0) return 2;
    return 1;
}

template <bool AKC, bool BKC, int EPI, int VA, int VB>
int launch_one(const Args& a, unsigned grid, hipStream_t st) {
    static bool attr_done = false;      // (per instantiation; racing threads set the same value)
    if (!attr_done) {
        MREC_HIP_CHECK(hipFuncSetAttribute((const void*)k_gemm_f32<AKC, BKC, EPI, VA, VB>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr_done = true;
    }
    k_gemm_f32<AKC, BKC, EPI, VA, VB><<<grid, 256, LDS_BYTES, st>>>(a);
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}

// a: M, N, K, operands and epilogue set; tiles and the grid are filled in here
template <bool AKC, bool BKC, int EPI>
int launch(Args a, int va, int vb, int S, hipStream_t st) {
    // staged loads are buffer loads: 31-bit byte offsets
    const int64_t ea = (AKC ? (int64_t)a.M : (int64_t)a.K) * a.lda, eb = (BKC ? (int64_t)a.N : (int64_t)a.K) * a.ldb;
    if (ea * 4 >= (int64_t(1) << 31) || eb * 4 >= (int64_t(1) << 31)) return MREC_EUNSUPPORTED;
    a.tiles_m = (int)mrec_cdiv(a.M, BM);
    a.tiles_n = (int)mrec_cdiv(a.N, BN);
    const unsigned grid = (unsigned)((int64_t)a.tiles_m * a.tiles_n * S);
#define GF_GO(VA, VB) return launch_one<AKC, BKC, EPI, VA, VB>(a, grid, st)
    if (va == 4 && vb == 4) GF_GO(4, 4);
    else if (va == 4 && vb == 2) GF_GO(4, 2);
    else if (va == 2 && vb == 4) GF_GO(2, 4);
    else if (va == 2 && vb == 2) GF_GO(2, 2);
    else if (va == 1 && vb == 1) GF_GO(1, 1);
    else if (va == 1) { if (vb == 4) GF_GO(1, 4); else GF_GO(1, 2); }
    else { if (va == 4) GF_GO(4, 1); else GF_GO(2, 1); }
#undef GF_GO
}

}  // namespace gf32

using gf32::Args;

/* y = act(x . w + bias): x [M, K] (ldx), w [K, N] contiguous rows (ldw), bias [N] (nullable), y [M, N] (ldy) */
MREC_API int mrec_dense32_fwd(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, int64_t M, int32_t K,
                              int32_t N, int relu, float* y, int64_t ldy, void* stream) {
    if (M < 0 || K <= 0 || N <= 0 || ldx < K || ldw < N || ldy < N) return MREC_EINVAL;
    if (M == 0) return MREC_OK;
    if (!x || !w || !y) return MREC_EINVAL;
    if (M > (int64_t(1) << 30)) return MREC_EUNSUPPORTED;
    Args a{};
    a.A = x; a.lda = ldx; a.B = w; a.ldb = ldw; a.C = y; a.ldc = ldy;
    a.M = (int)M; a.N = N; a.K = K;
    a.k_per_slab = (int)mrec_align_up((size_t)K, gf32::BK);
    a.bias = bias; a.relu = relu;
    return gf32::launch<true, false, gf32::EPI_FWD>(a, gf32::vec_of(x, ldx, K), gf32::vec_of(w, ldw, N), 1, (hipStream_t)stream);
}

/* dx = (dy . w^T) * (h > 0): dy [M, N] (lddy), w [K, N] (ldw), h [M, K] (ldh, nullable), dx [M, K] (lddx);
 * colsum_ws (nullable): [ceil(M / 64), K] column sums of dx per 64 rows = the partials of the layer below's bias gradient */
MREC_API int mrec_dense32_bwd_input(const float* dy, int64_t lddy, const float* w, int64_t ldw, const float* h, int64_t ldh, int64_t M,
                                    int32_t K, int32_t N, float* dx, int64_t lddx, float* colsum_ws, void* stream) {
    if (M < 0 || K <= 0 || N <= 0 || lddy < N || ldw < N || lddx < K || (h && ldh < K)) return MREC_EINVAL;
    if (M == 0) return MREC_OK;
    if (!dy || !w || !dx) return MREC_EINVAL;
    if (M > (int64_t(1) << 30)) return MREC_EUNSUPPORTED;
    Args a{};
    a.A = dy; a.lda = lddy; a.B = w; a.ldb = ldw; a.C = dx; a.ldc = lddx;
    a.M = (int)M; a.N = K; a.K = N;                       // outputs [M, K], reduction over N
    a.k_per_slab = (int)mrec_align_up((size_t)N, gf32::BK);
    a.H = h; a.ldh = ldh; a.colsum = colsum_ws;
    return gf32::launch<true, true, gf32::EPI_DGRAD>(a, gf32::vec_of(dy, lddy, N), gf32::vec_of(w, ldw, N), 1, (hipStream_t)stream);
}

/* dw_slabs[s] = x[slab s]^T . dy[slab s]: x [M, K] (ldx), dy [M, N] (lddy), dw_slabs [S, K, N] contiguous; the batch is cut
 * into S slabs of ceil(M / S / 32) * 32 rows (mrec_dense32_bwd_weight_slabs proposes S) */
MREC_API int mrec_dense32_bwd_weight(const float* x, int64_t ldx, const float* dy, int64_t lddy, int64_t M, int32_t K, int32_t N,
                                     int32_t S, float* dw_slabs, void* stream) {
    if (M < 0 || K <= 0 || N <= 0 || S <= 0 || ldx < K || lddy < N) return MREC_EINVAL;
    if (!dw_slabs) return MREC_EINVAL;
    if (M == 0) {
        MREC_HIP_CHECK(hipMemsetAsync(dw_slabs, 0, (size_t)S * K * N * sizeof(float), (hipStream_t)stream));
        return MREC_OK;
    }
    if (!x || !dy) return MREC_EINVAL;
    if (M > (int64_t(1) << 30)) return MREC_EUNSUPPORTED;
    Args a{};
    a.A = x; a.lda = ldx; a.B = dy; a.ldb = lddy; a.C = dw_slabs; a.ldc = N;
    a.M = K; a.N = N; a.K = (int)M;                       // outputs [K, N], reduction over the batch
    a.k_per_slab = (int)mrec_align_up((size_t)mrec_cdiv(M, S), gf32::BK);
    a.slab_stride = (int64_t)K * N;
    if ((int64_t)a.k_per_slab * (S - 1) >= M && S > 1) return MREC_EINVAL;      // an empty slab: S too large for this batch
    return gf32::launch<false, false, gf32::EPI_PLAIN>(a, gf32::vec_of(x, ldx, K), gf32::vec_of(dy, lddy, N), S, (hipStream_t)stream);
}

/* Batch slabs: the output has only ceil(K / 128) * ceil(N / 128) tiles.  The count
 * that takes the fewest rounds of the chip's 512 workgroup slots x rows per workgroup; among equals the smallest (the slabs are
 * the optimizer's to read). */
MREC_API int mrec_dense32_bwd_weight_slabs(int64_t M, int32_t K, int32_t N, int32_t* out) {
    if (!out || M < 0 || K <= 0 || N <= 0) return MREC_EINVAL;
    int64_t smax = M / 256 > 0 ? M / 256 : 1;             // at least 256 batch rows per slab
    if (smax > 64) smax = 64;
    int64_t best = 1, best_cost = -1;
    for (int64_t S = 1; S <= smax; ++S) {
        const int64_t rows = (int64_t)mrec_align_up((size_t)mrec_cdiv(M, S), gf32::BK);
        if (S > 1 && rows * (S - 1) >= M) continue;       // an empty slab
        const int64_t rounds = mrec_cdiv(mrec_cdiv(K, gf32::BM) * mrec_cdiv(N, gf32::BN) * S, gf32::SLOTS);
        // in units of one tile row x one batch row (~1.2 ns of a workgroup's MFMAs): the rounds, a workgroup's fill and drain,
        // and the slab's trip to HBM and back into the optimizer (8 bytes per element at ~4 TB/s)
        const int64_t cost = rounds * (gf32::BM * rows + 8192) + S * ((int64_t)K * N / 600);
        if (best_cost < 0 || cost < best_cost) { best = S; best_cost = cost; }
    }
    *out = (int32_t)best;
    return MREC_OK;
}
