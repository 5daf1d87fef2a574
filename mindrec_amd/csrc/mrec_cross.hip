// mrec_cross.hip -- DCN-v1 cross layers, all L layers in one HBM pass, for gfx950.
//
// Reference: CrossLayer.construct, models/deep_and_cross/src/deep_and_cross.py:139-149
//     y = x0 * (x_l . w) + b + x_l          (w, b in R^D: a GEMV + rank-1 update, HBM-bound)
// applied six times in DeepCrossModel.construct (:300-306).  MindSpore runs 6 x (tensor_dot,
// MatMul, two adds), re-reading the [B, D] activations every layer; here one wave64 owns a row,
// keeps x0 and x_l in registers (D = 1170 -> 19 floats per lane) and streams x0 in / y out once:
// 2*B*D*4 bytes for the whole stack.
//
// Backward.  The stack is affine in x0 with per-row scalar coefficients:
//     x_l = a_l * x0 + beta_l,   a_l = 1 + sum_{l'<l} s_l',   beta_l = sum_{l'<l} b_l',  s_l = x_l . w_l
// so with t_l = gy_{l+1} . x0 (a scalar recursion t_l = dy.x0 + sum_{l'>l} t_l' (w_l'.x0)) and
// u_l = t_l * a_l:
//     dx0  = a_L * dy + sum_l u_l w_l
//     dw_l = sum_rows u_l x0 + beta_l * sum_rows t_l
//     db_l = colsum(dy) + sum_{l'>l} w_l' * sum_rows t_l'
// One pass reads x0 and dy, writes dx0, and keeps the [L, D] batch sums sum_rows u_l x0 in
// registers per wave; per-wave slabs are then added in wave order (bitwise reproducible).
#include "mrec_common.h"

namespace {

constexpr int LMAX = 8;

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d, 64);
    return x;
}

// The same sum on the DPP network (__shfl_xor compiles to ds_bpermute: an LDS round trip per step, six dependent ones per
// sum): two quad permutes and the two row mirrors leave every lane of a 16-lane row with the row's sum, row_bcast15 /
// row_bcast31 carry the sums down the rows, lane 63 ends with the total.  A different (but fixed) order of additions.
template <int CTRL, int ROWS>
__device__ __forceinline__ float dpp_take(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROWS, 0xF, false));
}
__device__ __forceinline__ float wave_sum_dpp(float x) {
    x += dpp_take<0xB1, 0xF>(x);      // quad_perm [1,0,3,2]
    x += dpp_take<0x4E, 0xF>(x);      // quad_perm [2,3,0,1]
    x += dpp_take<0x141, 0xF>(x);     // row_half_mirror
    x += dpp_take<0x140, 0xF>(x);     // row_mirror
    x += dpp_take<0x142, 0xA>(x);     // row_bcast15 into rows 1, 3
    x += dpp_take<0x143, 0xC>(x);     // row_bcast31 into rows 2, 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));
}

// One activation row as a buffer resource of D*4 bytes (0 bytes = a row that does not exist): the hardware
// range check returns 0 for the padded columns c >= D and drops stores to them, so the row loops carry no
// guards, and every access is base + lane*4 + an immediate (one address register, not one per load).
// The descriptor must be provably wave-uniform, hence the readfirstlane of the row pointer.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_rsrc(const float* p, int bytes) {
    const uint64_t a = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0,
                                             __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
// voff = the lane's byte offset (one VGPR for the whole row), soff = the wave-uniform part (scalar operand)
__device__ __forceinline__ float row_load(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void row_store(__amdgpu_buffer_rsrc_t r, int voff, int soff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 0);
}

// w and b of all layers ([L, D] each) are staged once per block in LDS, each layer padded with zeros to
// DP = 64 * NPL columns (2*L*DP*4 bytes: 60 KB for the reference's L = 6, D = 1170).  Two reasons:
// re-reading them from L2 for every row made the kernel L2-bound at 9 % of the HBM roofline (56 KB of
// weights per 9 KB row), and with the padding the inner loops need no `c < D` guards -- a guard per LDS
// read compiled to a branch per read and serialised the reads (~1100 branches in the backward kernel).
// WLDS = false reads w / b from global memory through a per-layer buffer resource (also guard-free) for stacks
// too large for LDS.
template <bool WLDS>
__device__ __forceinline__ float wload(const float* __restrict__ g, const float* s, int l, int lane, int j, int D, int DP) {
    if (WLDS) return s[l * DP + lane + 64 * j];
    return row_load(row_rsrc(g + (int64_t)l * D, D * 4), lane * 4, j * 256);
}

template <int NT, int DP, bool WITH_B = true>
__device__ __forceinline__ void stage_wb(const float* __restrict__ w, const float* __restrict__ b, int L, int D,
                                         float* sw, float* sb) {
    const int total = L * DP;
    for (int base = 0; base < total; base += NT * 4) {
        float tw[4], tb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = base + u * NT + (int)threadIdx.x;
            const int l = i / DP, cc = i - l * DP;
            const bool ok = i < total && cc < D;
            const int src = ok ? l * D + cc : 0;
            const float a = w[src], bb = WITH_B ? b[src] : 0.0f;
            tw[u] = ok ? a : 0.0f;
            tb[u] = ok ? bb : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = base + u * NT + (int)threadIdx.x;
            if (i < total) {
                sw[i] = tw[u];
                if (WITH_B) sb[i] = tb[u];
            }
        }
    }
    __syncthreads();
}

// A lane owns FOUR consecutive columns per 256-column block (columns 256 q + 4 lane + {0..3}): x0, dy and dx0 move as
// 16-byte buffer accesses (the range check zero-fills / drops the dwords past column D one by one), w comes out of LDS as
// ds_read_b128, and the arithmetic runs on 2-wide packed fp32 (v_pk_fma_f32) -- a quarter of the memory and LDS instructions
// and half of the FMAs of a one-column-per-lane layout, which was what bound this kernel (issue slots, two waves per SIMD).
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f4 row_load4(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ void row_store4(__amdgpu_buffer_rsrc_t r, int voff, int soff, f4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), r, voff, soff, 0);
}
// the activation rows: touched once per launch
#ifndef MREC_CROSS_AUX_LD
#define MREC_CROSS_AUX_LD 0
#endif
#ifndef MREC_CROSS_AUX_ST
#define MREC_CROSS_AUX_ST 2      // results: nontemporal (B = 32768: backward 94 -> 83 us, forward 64 -> 56; B = 16384: the same
#endif                           // to 1 % better); nontemporal LOADS of the rows cost 2-5 us at B = 16384 (r03_cross_bwd_steps.txt)
__device__ __forceinline__ f4 act_load4(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, MREC_CROSS_AUX_LD));
}
__device__ __forceinline__ void act_store4(__amdgpu_buffer_rsrc_t r, int voff, int soff, f4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), r, voff, soff, MREC_CROSS_AUX_ST);
}
__device__ __forceinline__ f4 fma4(f4 a, f4 b, f4 c) {
    const f2 lo = __builtin_elementwise_fma(a.lo, b.lo, c.lo), hi = __builtin_elementwise_fma(a.hi, b.hi, c.hi);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}
__device__ __forceinline__ f4 bc4(float u) { return f4{u, u, u, u}; }
__device__ __forceinline__ float hsum4(f4 v) { return (v.x + v.y) + (v.z + v.w); }


// w (and b) of all layers into LDS in ONE memory round trip: the (array, layer, 256-column block) pieces are dealt out to the
// waves, every wave requests its pieces as 16-byte buffer loads (the range check pads the rows with zeros) before it stores
// the first.  (stage_wb above takes total / (4 NT) dependent rounds: two in the forward -- 4 of its 32 us.)
template <int NT, int NPL, bool WITH_B>
__device__ __forceinline__ void stage_rows4(const float* __restrict__ w, const float* __restrict__ b, int L, int D,
                                            float* sw, float* sb) {
    constexpr int DP = NPL * 64, NQ = NPL / 4, WPB = NT / 64;
    constexpr int NSTG = ((WITH_B ? 2 : 1) * LMAX * NQ + WPB - 1) / WPB;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int per = L * NQ, total = (WITH_B ? 2 : 1) * per;
    f4 stg[NSTG];
#pragma unroll
    for (int i = 0; i < NSTG; ++i) {
        const int pc = wave + i * WPB;
        const bool ok = pc < total;
        const int a = pc >= per ? 1 : 0, r = pc - a * per, l = r / NQ, q = r - l * NQ;
        const float* src = (WITH_B && a) ? b : w;
        stg[i] = row_load4(row_rsrc(src + (int64_t)(ok ? l : 0) * D, ok ? D * 4 : 0), lane * 16, 1024 * q);
    }
#pragma unroll
    for (int i = 0; i < NSTG; ++i) {
        const int pc = wave + i * WPB;
        const int a = pc >= per ? 1 : 0, r = pc - a * per, l = r / NQ, q = r - l * NQ;
        if (pc < total) *(f4*)(((WITH_B && a) ? sb : sw) + l * DP + 256 * q + 4 * lane) = stg[i];
    }
    __syncthreads();
}

// forward, 4 columns per lane (NPL a multiple of 4: every D > 128): 16-byte row accesses, ds_read_b128 for w and b, packed
// fp32 arithmetic; the update keeps the oracle's operation order (x * s + b, then + x_l; no fused multiply-add)
template <bool WLDS>
__device__ __forceinline__ f4 wload4(const float* __restrict__ g, const float* s, int l, int lane, int q, int D, int DP) {
    if (WLDS) return *(const f4*)(s + l * DP + 256 * q + 4 * lane);
    return row_load4(row_rsrc(g + (int64_t)l * D, D * 4), lane * 16, q * 1024);
}
constexpr int FWD_NT = 1024;   // 16 waves: one block per CU, w / b staged once per CU
constexpr int BWD_NT = 512;   // 8 waves: one block per CU at two waves per SIMD -> one slab per CU for the reduction kernel

template <int NPL, bool WLDS>
__global__ __launch_bounds__(FWD_NT) void k_cross_fwd(const float* __restrict__ x0, const float* __restrict__ w,
                                                      const float* __restrict__ b, int L, int64_t B, int D,
                                                      float* __restrict__ out) {
    constexpr int DP = NPL * 64;
    extern __shared__ float smem[];
    float* sw = smem;
    float* sb = smem + (WLDS ? L * DP : 0);
    if (WLDS) stage_wb<FWD_NT, DP>(w, b, L, D, sw, sb);
    const int lane = threadIdx.x & 63;
    const int voff = lane * 4;
    constexpr int WPB = FWD_NT / 64;
    const int64_t nw = (int64_t)gridDim.x * WPB;
    // two rows per wave per iteration: one LDS read of w / b serves both
    for (int64_t rowA = (int64_t)blockIdx.x * WPB + (threadIdx.x >> 6); rowA < B; rowA += 2 * nw) {
        const int64_t rowB = rowA + nw;
        const bool vB = rowB < B;
        float xA[NPL], lA[NPL], xB[NPL], lB[NPL];
        const __amdgpu_buffer_rsrc_t rA = row_rsrc(x0 + rowA * D, D * 4);
        const __amdgpu_buffer_rsrc_t rB = row_rsrc(x0 + (vB ? rowB : rowA) * D, vB ? D * 4 : 0);
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            xA[j] = row_load(rA, voff, 256 * j);
            xB[j] = row_load(rB, voff, 256 * j);
            lA[j] = xA[j];
            lB[j] = xB[j];
        }
        for (int l = 0; l < L; ++l) {
            float pA = 0.0f, pB = 0.0f;
#pragma unroll
            for (int j = 0; j < NPL; ++j) {
                const float wv = wload<WLDS>(w, sw, l, lane, j, D, DP);
                pA += lA[j] * wv;
                pB += lB[j] * wv;
            }
            const float sA = wave_sum(pA), sB = wave_sum(pB);
#pragma unroll
            for (int j = 0; j < NPL; ++j) {
                const float bv = wload<WLDS>(b, sb, l, lane, j, D, DP);
                lA[j] = (xA[j] * sA + bv) + lA[j];
                lB[j] = (xB[j] * sB + bv) + lB[j];
            }
        }
        const __amdgpu_buffer_rsrc_t oA = row_rsrc(out + rowA * D, D * 4);
        const __amdgpu_buffer_rsrc_t oB = row_rsrc(out + (vB ? rowB : rowA) * D, vB ? D * 4 : 0);
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            row_store(oA, voff, 256 * j, lA[j]);
            row_store(oB, voff, 256 * j, lB[j]);
        }
    }
}

template <int NPL, bool WLDS>
__global__ __launch_bounds__(FWD_NT) void k_cross_fwd4(const float* __restrict__ x0, const float* __restrict__ w,
                                                       const float* __restrict__ b, int L, int64_t B, int D,
                                                       float* __restrict__ out) {
    static_assert(NPL % 4 == 0, "256-column blocks");
    constexpr int DP = NPL * 64, NQ = NPL / 4;
    extern __shared__ float smem[];
    float* sw = smem;
    float* sb = smem + (WLDS ? L * DP : 0);
    if (WLDS) stage_rows4<FWD_NT, NPL, true>(w, b, L, D, sw, sb);
    const int lane = threadIdx.x & 63;
    const int voff = lane * 16;
    constexpr int WPB = FWD_NT / 64;
    const int64_t nw = (int64_t)gridDim.x * WPB;
    // two rows per wave per iteration: one LDS read of w / b serves both.  (Requesting a wave's next pair right behind the
    // stores of the one before, the first pair in front of the staging: 31.1 us against 28.4 at B = 16384 -- not kept.)
    for (int64_t rowA = (int64_t)blockIdx.x * WPB + (threadIdx.x >> 6); rowA < B; rowA += 2 * nw) {
        const int64_t rowB = rowA + nw;
        const bool vB = rowB < B;
        f4 xA[NQ], lA[NQ], xB[NQ], lB[NQ];
        const __amdgpu_buffer_rsrc_t rA = row_rsrc(x0 + rowA * D, D * 4);
        const __amdgpu_buffer_rsrc_t rB = row_rsrc(x0 + (vB ? rowB : rowA) * D, vB ? D * 4 : 0);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            xA[q] = act_load4(rA, voff, 1024 * q);
            xB[q] = act_load4(rB, voff, 1024 * q);
            lA[q] = xA[q];
            lB[q] = xB[q];
        }
        for (int l = 0; l < L; ++l) {
            f4 pA = bc4(0.0f), pB = bc4(0.0f);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const f4 wv = wload4<WLDS>(w, sw, l, lane, q, D, DP);
                pA += lA[q] * wv;
                pB += lB[q] * wv;
            }
            const f4 sA = bc4(wave_sum_dpp(hsum4(pA))), sB = bc4(wave_sum_dpp(hsum4(pB)));
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const f4 bv = wload4<WLDS>(b, sb, l, lane, q, D, DP);
                lA[q] = (xA[q] * sA + bv) + lA[q];
                lB[q] = (xB[q] * sB + bv) + lB[q];
            }
        }
        const __amdgpu_buffer_rsrc_t oA = row_rsrc(out + rowA * D, D * 4);
        const __amdgpu_buffer_rsrc_t oB = row_rsrc(out + (vB ? rowB : rowA) * D, vB ? D * 4 : 0);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            act_store4(oA, voff, 1024 * q, lA[q]);
            act_store4(oB, voff, 1024 * q, lB[q]);
        }
    }
}

// slab layout per block: [LMAX][D] sums of u_l*x0, then [D] colsum(dy), then [LMAX] sums of t_l.
__host__ __device__ inline int64_t slab_floats(int D) { return (int64_t)(LMAX + 1) * D + LMAX; }

// two waves per SIMD where the [LT][NPL] accumulators + x, dy, colsum rows leave room under 256 registers
constexpr int bwd_occ(int npl, int lt) { return (lt + 3) * npl <= 180 ? 2 : 1; }

// LDS of the backward: w of all layers while the rows are walked; afterwards the same area holds up to four REGIONS of
// (LT + 1) * DP + 64 floats through which the waves' batch sums are added up pairwise.
constexpr int bwd_region_floats(int npl, int lt) { return (lt + 1) * npl * 64 + 64; }
constexpr int bwd_regions(int npl, int lt) {
    const int fit = (int)(160 * 1024 / (bwd_region_floats(npl, lt) * sizeof(float)));
    return fit >= 4 ? 4 : (fit >= 2 ? 2 : 1);
}

// LT = compile-time bound on the layer count (2, 4, 6 or 8): the [LT][NPL] batch-sum accumulators live in
// registers, so they must be sized statically.
//
// Per row the kernel needs only the L dots P_l = x0 . w_l (independent of each other, so their wave
// reductions overlap): with x_l = a_l x0 + beta_l,  s_l = x_l . w_l = a_l P_l + c_l,  a_{l+1} = a_l + s_l.
// x_l itself is never rebuilt; fmaf is used freely -- this kernel is checked against the oracle's
// double-precision backward to a tolerance, not bit for bit.
//
// What a launch costs besides the rows (measured with one block on an idle chip, profiles/r03_cross_bwd_steps.txt: 16 us of
// the 53 us at B = 16384 were prologue and epilogue) is kept to one memory round trip in front and three LDS exchanges behind:
//  * prologue: every wave requests its share of w's rows (on their way to LDS; the range check pads them with zeros) and,
//    wave l = 1 .. L-1, the rows b_0 .. b_{l-1} and w_l for c_l = (b_0 + .. + b_{l-1}) . w_l, all before the first wait
//    (before: four dependent staging rounds, then 15 pair dots b_l' . w_l four or five deep per wave, each a load + a
//    six-step shuffle chain);
//  * epilogue: the eight waves' sums are added pairwise through LDS -- (w0+w4), (w1+w5), .. then (.. + ..) -- three
//    exchanges instead of eight turns on one LDS slab; wave 0 stores the block's slab straight from its registers.
template <int NPL, int LT, bool WLDS>
__global__ __launch_bounds__(BWD_NT) __attribute__((amdgpu_waves_per_eu(bwd_occ(NPL, LT), bwd_occ(NPL, LT)))) void k_cross_bwd(const float* __restrict__ x0, const float* __restrict__ w,
                                                      const float* __restrict__ b, int L, int64_t B, int D,
                                                      const float* __restrict__ dy, float* __restrict__ dx0,
                                                      float* __restrict__ slabs, int accumulate) {
    static_assert(NPL % 4 == 0 && WLDS, "the backward is instantiated for 256-column blocks with w staged in LDS");
    constexpr int DP = NPL * 64;
    constexpr int NQ = NPL / 4;             // 256-column blocks
    extern __shared__ float smem[];
    __shared__ float s_cl[LMAX];
    float* sw = smem;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int voff = lane * 16;
    constexpr int WPB = BWD_NT / 64;
    const int64_t wid = (int64_t)blockIdx.x * WPB + wave;
    const int64_t nw = (int64_t)gridDim.x * WPB;
    {
        constexpr int NSTG = (LT * NQ + WPB - 1) / WPB;       // (layer, 256-column block) pieces of w per wave
        f4 stg[NSTG];
#pragma unroll
        for (int i = 0; i < NSTG; ++i) {
            const int pc = wave + i * WPB, l = pc / NQ, q = pc - l * NQ;
            const bool ok = l < L;
            stg[i] = row_load4(row_rsrc(w + (int64_t)(ok ? l : 0) * D, ok ? D * 4 : 0), voff, 1024 * q);
        }
        const bool dot = wave >= 1 && wave < L;                 // c_wave; wave-uniform
        f4 wl[NQ], beta[NQ];
        {
            const __amdgpu_buffer_rsrc_t rw = row_rsrc(w + (int64_t)(dot ? wave : 0) * D, dot ? D * 4 : 0);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                wl[q] = row_load4(rw, voff, 1024 * q);
                beta[q] = bc4(0.0f);
            }
        }
#pragma unroll
        for (int lp = 0; lp < LT - 1; ++lp) {
            const bool ok = dot && lp < wave;
            const __amdgpu_buffer_rsrc_t rb = row_rsrc(b + (int64_t)(ok ? lp : 0) * D, ok ? D * 4 : 0);
            f4 bv[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) bv[q] = row_load4(rb, voff, 1024 * q);
#pragma unroll
            for (int q = 0; q < NQ; ++q) beta[q] += bv[q];      // rows past the wave's own come back as zeros
        }
#pragma unroll
        for (int i = 0; i < NSTG; ++i) {
            const int pc = wave + i * WPB, l = pc / NQ, q = pc - l * NQ;
            if (l < L) *(f4*)(sw + l * DP + 256 * q + 4 * lane) = stg[i];
        }
        f4 acc = bc4(0.0f);
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc = fma4(beta[q], wl[q], acc);
        const float c = wave_sum_dpp(hsum4(acc));
        if (lane == 0 && wave < LMAX) s_cl[wave] = dot ? c : 0.0f;
    }
    __syncthreads();
    float cl[LT];
#pragma unroll
    for (int l = 0; l < LT; ++l) cl[l] = s_cl[l];
    f4 accw[LT][NQ];
    f4 accd[NQ];
    float accT[LT];
#pragma unroll
    for (int l = 0; l < LT; ++l) {
        accT[l] = 0.0f;
#pragma unroll
        for (int q = 0; q < NQ; ++q) accw[l][q] = bc4(0.0f);
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) accd[q] = bc4(0.0f);

    // (Requesting the next row ahead of time -- into a per-wave LDS area with buffer_load ... lds, the registers being full,
    // waited for with a counted vmcnt so that the row's own stores stay in flight -- was built and measured twice, the second
    // time on this 4-column layout: 57.2 us against 53.4 at B = 16384, 39.0 against 34.9 for one block on an idle chip.  A
    // row's turn is a chain of dependent LDS and cross-lane steps, 2.4 us long even with nothing else on the chip, and the
    // load is the smaller part of it.)
    for (int64_t row = wid; row < B; row += nw) {
        f4 x[NQ], g[NQ];
        const __amdgpu_buffer_rsrc_t rx = row_rsrc(x0 + row * D, D * 4);
        const __amdgpu_buffer_rsrc_t rg = row_rsrc(dy + row * D, D * 4);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            x[q] = act_load4(rx, voff, 1024 * q);
            g[q] = act_load4(rg, voff, 1024 * q);
        }
        float P[LT];
        f4 qa = bc4(0.0f);
#pragma unroll
        for (int q = 0; q < NQ; ++q) qa = fma4(g[q], x[q], qa);
#pragma unroll
        for (int l = 0; l < LT; ++l) {
            f4 pa = bc4(0.0f);
            if (l < L) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) pa = fma4(x[q], *(const f4*)(sw + l * DP + 256 * q + 4 * lane), pa);
            }
            P[l] = hsum4(pa);
            __builtin_amdgcn_sched_barrier(0);   // keep one layer's LDS reads in flight, not all of them (registers)
        }
        const float qd = wave_sum_dpp(hsum4(qa));
#pragma unroll
        for (int l = 0; l < LT; ++l) P[l] = wave_sum_dpp(P[l]);      // P[l] == 0 for l >= L
        float a[LT + 1], t[LT];
        a[0] = 1.0f;
#pragma unroll
        for (int l = 0; l < LT; ++l) a[l + 1] = l < L ? a[l] + fmaf(a[l], P[l], cl[l]) : a[l];
        // t_l = q + sum_{l' > l} t_l' P_l'
        float run = 0.0f;
#pragma unroll
        for (int l = LT - 1; l >= 0; --l) {
            t[l] = l < L ? qd + run : 0.0f;
            run = fmaf(t[l], P[l], run);
        }
        // dx0 = a_L * dy + sum_l u_l w_l, built in dy's registers; batch sums accumulate
        const f4 aL = bc4(a[LT]);                // a[LT] == a[L]: layers past L add nothing
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            accd[q] += g[q];
            g[q] = aL * g[q];
        }
#pragma unroll
        for (int l = 0; l < LT; ++l) {
            if (l < L) {
                const f4 u = bc4(t[l] * a[l]);
                accT[l] += t[l];
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    g[q] = fma4(u, *(const f4*)(sw + l * DP + 256 * q + 4 * lane), g[q]);
                    accw[l][q] = fma4(u, x[q], accw[l][q]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        const __amdgpu_buffer_rsrc_t ro = row_rsrc(dx0 + row * D, D * 4);
        if (accumulate) {              // dx0 += ...: onto the gradient another branch left there (Deep&Cross: the deep net's)
#pragma unroll
            for (int q = 0; q < NQ; ++q) g[q] += act_load4(ro, voff, 1024 * q);
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) act_store4(ro, voff, 1024 * q, g[q]);
    }
    // Block slab: pairwise through LDS (the w staging area is dead by now; rows padded to DP, so no guards), the same
    // order every run; wave 0 ends with the block's sums and stores them (the range check drops the padding).
    __syncthreads();
    constexpr int NR = bwd_regions(NPL, LT);
    constexpr int RF = bwd_region_floats(NPL, LT);
#pragma unroll
    for (int h = WPB / 2; h >= 1; h >>= 1) {
#pragma unroll
        for (int c0 = 0; c0 < h; c0 += NR) {
            if (wave >= h + c0 && wave < h + c0 + NR && wave < 2 * h) {
                float* rg = smem + (wave - h - c0) * RF;
#pragma unroll
                for (int l = 0; l < LT; ++l)
#pragma unroll
                    for (int q = 0; q < NQ; ++q) *(f4*)(rg + l * DP + 256 * q + 4 * lane) = accw[l][q];
#pragma unroll
                for (int q = 0; q < NQ; ++q) *(f4*)(rg + LT * DP + 256 * q + 4 * lane) = accd[q];
                if (lane == 0) {
#pragma unroll
                    for (int l = 0; l < LT; ++l) rg[(LT + 1) * DP + l] = accT[l];
                }
            }
            __syncthreads();
            if (wave >= c0 && wave < c0 + NR && wave < h) {
                const float* rg = smem + (wave - c0) * RF;
#pragma unroll
                for (int l = 0; l < LT; ++l)
#pragma unroll
                    for (int q = 0; q < NQ; ++q) accw[l][q] += *(const f4*)(rg + l * DP + 256 * q + 4 * lane);
#pragma unroll
                for (int q = 0; q < NQ; ++q) accd[q] += *(const f4*)(rg + LT * DP + 256 * q + 4 * lane);
#pragma unroll
                for (int l = 0; l < LT; ++l) accT[l] += rg[(LT + 1) * DP + l];
            }
            if (h > 1 || c0 + NR < h) __syncthreads();
        }
    }
    if (wave != 0) return;
    float* sl = slabs + (int64_t)blockIdx.x * slab_floats(D);
#pragma unroll
    for (int l = 0; l < LT; ++l) {
        if (l < L) {
            const __amdgpu_buffer_rsrc_t ro = row_rsrc(sl + (int64_t)l * D, D * 4);
#pragma unroll
            for (int q = 0; q < NQ; ++q) row_store4(ro, voff, 1024 * q, accw[l][q]);
        }
    }
    {
        const __amdgpu_buffer_rsrc_t ro = row_rsrc(sl + (int64_t)LMAX * D, D * 4);
#pragma unroll
        for (int q = 0; q < NQ; ++q) row_store4(ro, voff, 1024 * q, accd[q]);
    }
    if (lane == 0) {
#pragma unroll
        for (int l = 0; l < LT; ++l)
            if (l < L) sl[(int64_t)(LMAX + 1) * D + l] = accT[l];
    }
}

// Slab reduction and the composition of dw / db in ONE launch with no exchange between its blocks.  blockIdx.y = quantity q
// (q < L: sum u_l x0 of layer q -> dw_q; q == L: colsum(dy) -> db of every layer).  A block owns 32 columns; its 32 thread
// groups each add a strided subset of the slabs, then the 32 partial sums are added in group order: a fixed order,
// independent of timing.  The L scalars T_l = sum over rows of t_l that the compositions need are added up by EVERY block from
// the same slabs in the same order (same bits; 6 KB out of L2) -- the version before this one reduced them in blocks of their
// own and had the block finishing a column tile last compose from a scratch slab: a release fence, a counter and a dozen
// agent-scope loads per tile, 12.7 us for the launch even with a single slab to add; this one: see profiles/r03_cross_bwd_steps.txt.
__global__ __launch_bounds__(1024) void k_cross_bwd_finish(const float* __restrict__ slabs, int nslabs, int L, int D,
                                                           const float* __restrict__ w, const float* __restrict__ b,
                                                           float* __restrict__ dw, float* __restrict__ db) {
    __shared__ float part[32][33];
    __shared__ float partT[32][LMAX];
    __shared__ float sT[LMAX];
    const int q = blockIdx.y;
    const int cl = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int64_t sf = slab_floats(D);
    const int c = blockIdx.x * 32 + cl;
    const bool live = c < D;
    const int64_t off = (int64_t)(q < L ? q : LMAX) * D + (live ? c : 0);
    const int64_t offT = (int64_t)(LMAX + 1) * D + (cl < L ? cl : 0);
    // what the composition reads of b / w is requested now, with the slabs, not after them
    float bw[LMAX];
#pragma unroll
    for (int l = 0; l < LMAX; ++l) {
        const bool need = grp == 0 && live && (q < L ? l < q : l < L);
        bw[l] = need ? (q < L ? b : w)[l * D + c] : 0.0f;
    }
    float s = 0.0f, st = 0.0f;
    const bool tcol = cl < L;
#pragma unroll 8
    for (int k = grp; k < nslabs; k += 32) {
        s += slabs[k * sf + off];
        if (tcol) st += slabs[k * sf + offT];
    }
    part[grp][cl] = s;
    if (cl < LMAX) partT[grp][cl] = st;
    __syncthreads();
    if ((int)threadIdx.x < L) {
        float t = partT[0][threadIdx.x];
#pragma unroll
        for (int g2 = 1; g2 < 32; ++g2) t += partT[g2][threadIdx.x];
        sT[threadIdx.x] = t;
    }
    __syncthreads();
    if (grp != 0 || !live) return;
    float t = part[0][cl];
#pragma unroll
    for (int g2 = 1; g2 < 32; ++g2) t += part[g2][cl];
    if (q < L) {
        float beta = 0.0f;                                  // beta_q[c] = b_0[c] + .. + b_{q-1}[c]
#pragma unroll
        for (int l = 0; l < LMAX; ++l) beta += bw[l];      // (zeros from l = q on)
        dw[q * D + c] = t + beta * sT[q];
    } else {
        float tail = 0.0f;                                  // sum_{l' > l} w_l'[c] * T_l'
#pragma unroll
        for (int l = LMAX - 1; l >= 0; --l) {
            if (l < L) {
                db[l * D + c] = t + tail;
                tail += bw[l] * sT[l];
            }
        }
    }
}

inline int npl_bucket(int D) {
    const int need = (D + 63) / 64;
    const int buckets[] = {1, 2, 4, 8, 12, 16, 20, 24, 32};
    for (int v : buckets) if (need <= v) return v;
    return -1;
}

inline unsigned bwd_blocks(int64_t B) {
    int64_t blocks = mrec_cdiv(B, (BWD_NT / 64) * 8);  // >= 8 rows per wave
    if (blocks > 256) blocks = 256;       // one block per CU (8 waves x 256 VGPRs)
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

}  // namespace

#define MREC_NPL_DISPATCH(NPLV, CALL)                                                              \
    switch (NPLV) {                                                                                \
        case 1: { constexpr int N_ = 1; CALL; } break;                                             \
        case 2: { constexpr int N_ = 2; CALL; } break;                                             \
        case 4: { constexpr int N_ = 4; CALL; } break;                                             \
        case 8: { constexpr int N_ = 8; CALL; } break;                                             \
        case 12: { constexpr int N_ = 12; CALL; } break;                                           \
        case 16: { constexpr int N_ = 16; CALL; } break;                                           \
        case 20: { constexpr int N_ = 20; CALL; } break;                                           \
        case 24: { constexpr int N_ = 24; CALL; } break;                                           \
        default: { constexpr int N_ = 32; CALL; } break;                                           \
    }

namespace {
constexpr size_t kMaxLds = 160 * 1024;

template <class Kern>
int set_lds(Kern kern, size_t bytes) {
    if (bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) { g_mrec_last_hip_error = (int)e; return MREC_EHIP; }
    }
    return MREC_OK;
}

template <int NPL>
int launch_fwd(const float* x0, const float* w, const float* b, int L, int64_t B, int D, float* out, hipStream_t st) {
    const size_t lds = (size_t)2 * L * NPL * 64 * sizeof(float);
    int64_t blocks = mrec_cdiv(B, 2 * (FWD_NT / 64));            // 16 waves x 2 rows per pass
    if (blocks > 256) blocks = 256;
    if constexpr (NPL % 4 == 0) {
        if (lds > 0 && lds <= kMaxLds) {
            int rc = set_lds(k_cross_fwd4<NPL, true>, lds);
            if (rc != MREC_OK) return rc;
            k_cross_fwd4<NPL, true><<<(unsigned)blocks, FWD_NT, lds, st>>>(x0, w, b, L, B, D, out);
        } else {
            k_cross_fwd4<NPL, false><<<(unsigned)blocks, FWD_NT, 0, st>>>(x0, w, b, L, B, D, out);
        }
        return MREC_OK;
    }
    if (lds > 0 && lds <= kMaxLds) {
        int rc = set_lds(k_cross_fwd<NPL, true>, lds);
        if (rc != MREC_OK) return rc;
        k_cross_fwd<NPL, true><<<(unsigned)blocks, FWD_NT, lds, st>>>(x0, w, b, L, B, D, out);
    } else {
        k_cross_fwd<NPL, false><<<(unsigned)blocks, FWD_NT, 0, st>>>(x0, w, b, L, B, D, out);
    }
    return MREC_OK;
}

template <int NPL, int LT>
int launch_bwd_lt(const float* x0, const float* w, const float* b, int L, int64_t B, int D, const float* dy, float* dx0,
                  float* slabs, unsigned blocks, hipStream_t st, int acc) {
    constexpr int DP = NPL * 64;
    // w of all layers (<= 8 x 2048 floats = 64 KB) always fits in LDS; the exchange regions of the epilogue reuse the area
    const size_t xlds = (size_t)bwd_regions(NPL, LT) * bwd_region_floats(NPL, LT) * sizeof(float);
    const size_t wlds = (size_t)L * DP * sizeof(float);
    const size_t lds = wlds > xlds ? wlds : xlds;
    static_assert((size_t)bwd_regions(NPL, LT) * bwd_region_floats(NPL, LT) * sizeof(float) <= 160 * 1024, "LDS");
    int rc = set_lds(k_cross_bwd<NPL, LT, true>, lds);
    if (rc != MREC_OK) return rc;
    k_cross_bwd<NPL, LT, true><<<blocks, BWD_NT, lds, st>>>(x0, w, b, L, B, D, dy, dx0, slabs, acc);
    return MREC_OK;
}

template <int NPL>
int launch_bwd(const float* x0, const float* w, const float* b, int L, int64_t B, int D, const float* dy, float* dx0,
               float* slabs, unsigned blocks, hipStream_t st, int acc) {
    if (L <= 2) return launch_bwd_lt<NPL, 2>(x0, w, b, L, B, D, dy, dx0, slabs, blocks, st, acc);
    if (L <= 4) return launch_bwd_lt<NPL, 4>(x0, w, b, L, B, D, dy, dx0, slabs, blocks, st, acc);
    if (L <= 6) return launch_bwd_lt<NPL, 6>(x0, w, b, L, B, D, dy, dx0, slabs, blocks, st, acc);
    return launch_bwd_lt<NPL, 8>(x0, w, b, L, B, D, dy, dx0, slabs, blocks, st, acc);
}
}  // namespace

MREC_API int mrec_cross_layers_f32(const float* x0, const float* w, const float* b, int32_t L, int64_t B, int32_t D,
                                   float* out, void* stream) {
    if (B < 0 || D <= 0 || L < 0) return MREC_EINVAL;
    if (B == 0) return MREC_OK;
    if (!x0 || !out || (L > 0 && (!w || !b))) return MREC_EINVAL;
    const int npl = npl_bucket(D);
    if (npl < 0) return MREC_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    int rc = MREC_OK;
    MREC_NPL_DISPATCH(npl, (rc = launch_fwd<N_>(x0, w, b, L, B, D, out, st)));
    if (rc != MREC_OK) return rc;
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}

MREC_API int mrec_cross_layers_bwd_workspace_bytes(int32_t L, int64_t B, int32_t D, size_t* out) {
    if (!out || B < 0 || D <= 0 || L < 0) return MREC_EINVAL;
    *out = (size_t)bwd_blocks(B) * slab_floats(D) * sizeof(float) + 256;      // one slab of batch sums per block
    return MREC_OK;
}

namespace {
int cross_bwd(const float* x0, const float* w, const float* b, int32_t L, int64_t B, int32_t D, const float* dy, float* dx0, float* dw,
              float* db, void* ws, size_t ws_bytes, void* stream, int acc);
}

MREC_API int mrec_cross_layers_bwd_f32(const float* x0, const float* w, const float* b, int32_t L, int64_t B,
                                       int32_t D, const float* dy, float* dx0, float* dw, float* db, void* ws,
                                       size_t ws_bytes, void* stream) {
    return cross_bwd(x0, w, b, L, B, D, dy, dx0, dw, db, ws, ws_bytes, stream, 0);
}

MREC_API int mrec_cross_layers_bwd_acc_f32(const float* x0, const float* w, const float* b, int32_t L, int64_t B,
                                           int32_t D, const float* dy, float* dx0, float* dw, float* db, void* ws,
                                           size_t ws_bytes, void* stream) {
    return cross_bwd(x0, w, b, L, B, D, dy, dx0, dw, db, ws, ws_bytes, stream, 1);
}

namespace {
int cross_bwd(const float* x0, const float* w, const float* b, int32_t L, int64_t B, int32_t D, const float* dy, float* dx0, float* dw,
              float* db, void* ws, size_t ws_bytes, void* stream, int acc) {
    if (B < 0 || D <= 0 || L < 0) return MREC_EINVAL;
    if (L > LMAX) return MREC_EUNSUPPORTED;
    if (!x0 || !dy || !dx0 || !ws || (L > 0 && (!w || !b || !dw || !db))) return MREC_EINVAL;
    const int npl = npl_bucket(D);
    if (npl < 0) return MREC_EUNSUPPORTED;
    const unsigned blocks = bwd_blocks(B);
    const int nslabs = (int)blocks;
    const int ntiles = (int)mrec_cdiv(D, 32);
    if (ws_bytes < (size_t)nslabs * slab_floats(D) * sizeof(float)) return MREC_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* slabs = (float*)ws;
    int rc = MREC_OK;
    // the backward's register-heavy instantiations use coarser column buckets (compile time)
    const int nb = npl <= 4 ? 4 : (npl <= 8 ? 8 : (npl <= 16 ? 16 : (npl <= 20 ? 20 : 32)));
    switch (nb) {
        case 4: rc = launch_bwd<4>(x0, w, b, L, B, D, dy, dx0, slabs, blocks, st, acc); break;
        case 8: rc = launch_bwd<8>(x0, w, b, L, B, D, dy, dx0, slabs, blocks, st, acc); break;
        case 16: rc = launch_bwd<16>(x0, w, b, L, B, D, dy, dx0, slabs, blocks, st, acc); break;
        case 20: rc = launch_bwd<20>(x0, w, b, L, B, D, dy, dx0, slabs, blocks, st, acc); break;
        default: rc = launch_bwd<32>(x0, w, b, L, B, D, dy, dx0, slabs, blocks, st, acc); break;
    }
    if (rc != MREC_OK) return rc;
    if (L > 0)
        k_cross_bwd_finish<<<dim3((unsigned)ntiles, (unsigned)(L + 1)), 1024, 0, st>>>(slabs, nslabs, L, D, w, b, dw, db);
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}
}  // namespace
