// mrec_tail.hip -- the TAIL of the dense net as one launch, and the derived copies of the 16-bit weights the GEMM kernels read.
//
// The last two hidden DenseLayers (512 -> 256 -> 128 in the reference's net, wide_and_deep.py:176-199), the output head
// (k_head_fwd_bwd's arithmetic, operation for operation) and the input-gradient bprops back through both layers, for a tile of
// 64 samples per workgroup with every intermediate activation in LDS.  As separate launches these five steps are latency-bound
// (5 x launch + prologue + epilogue for 16 GFLOP): 98 us of the 746-us step; here the activations never leave the CU between
// them.  The weight gradients of the two layers (batch reductions) ride the next backward launch (ExtraW, mrec_gemm.h).
//
// Work split: 8 waves; in every GEMM phase a wave owns ALL 64 rows x 1/8 of the output columns, so its "Q" fragments (the
// weights) are read by nobody else in the workgroup and go global/L2 -> registers directly (no LDS staging, no barriers inside
// a phase); the "P" fragments (activations / gradients, rows padded by 16 B: conflict-free ds_read_b128) come from LDS.
// The weights are read from FRAGMENT-ORDERED copies (k_tail_pack, refreshed with the operand shadow): [k-step][16-column
// tile][lane][8 elements], so a wave instruction reads 1 KB contiguous and the 8 waves of all 256 workgroups, which walk the
// k-steps in lockstep, sweep a contiguous window.  (Read from the row-major matrices -- 16 rows x 64 B per instruction, the same
// column offset in every wave -- the stream ran at 26 GB/s per CU, a third of what the XCD's L2 delivers: 10 us per 256-KB pass.)
// Accumulators hold C transposed as in mrec_gemm.h: a lane owns 4 consecutive output columns of one row.
//
// (Round 5, built, parity-green and measured, not kept: the tail THREE layers deep -- the 1024 -> 512 layer's forward as a phase in
// front and its input gradient as a phase behind, its 1-MB weight copies streamed through a two-chunk register ring, its three
// weight gradients riding the first layer's backward launch.  The launch took 90 us against 34 + the 26-us forward GEMM it
// replaces, the backward launch that absorbed the weight gradients 194 us against 129 + 63, the step 0.626 -> 0.657 ms.  Every
// workgroup pulls the whole weight matrix through its XCD's L2 for 64 rows of reuse: 32 CUs x 2 MB = 64 MB per XCD for the two
// phases, 56 us at the ~1.1 TB/s an XCD's L2 delivered to this access pattern -- the 256 x 256 GEMM tiles move a third of that.)
#include "mrec_common.h"
#include "mrec_dropout.h"
#include "mrec_mlp.h"
#include "mrec_gemm.h"

namespace tail {
using namespace mgemm;
constexpr int R = 64, K2 = 512, N2 = 256, N3 = 128, TT = 512;
constexpr int SX = K2 * 2 + 16, S2 = N2 * 2 + 16, S3 = N3 * 2 + 16;           // LDS row strides in bytes
constexpr int OX = 0, O2 = OX + R * SX, O3 = O2 + R * S2, ORED = O3 + R * S3;   // X | Y2 -> dz3 | Y3 -> dz4 | reduction scratch
constexpr int RED_BYTES = (8 * 2 * N3 + 8 * 2) * 4;
constexpr int PW = 2 * N3 + 4 + N2 + K2, P_DB3 = 2 * N3 + 4, P_DB2 = P_DB3 + N2;      // layout of a partial row
constexpr int LDS_BYTES = ORED + RED_BYTES;
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");

struct TailArgs {
    const uint16_t* x; int64_t ldx;                    // [B, K2] input of the first tail layer
    const uint16_t *w2f, *w3f, *w3b, *w2b;             // fragment-ordered weights: forward / backward operand of either layer
    const float *b2, *b3, *w5, *b5;
    const float *wide, *wprod, *wide_bias; int F;      // wide branch: per sample, or per-field products + bias
    const float* label;
    int64_t B; float dscale;
    uint16_t *y2, *dz4, *dz3, *dz2;                    // [B, N2] input of the second tail layer; [B, N3], [B, N2], [B, K2] gradients
    float *logit, *dlogit, *partial;                   // per-workgroup partial sums [B / 64][PW]: dW5 | db4 | db5, loss, 0, 0 | column
                                                       // sums of dz3 [N2] | of dz2 [K2] (k_tail_finish adds the rows up)
    DropArgs drop;                                     // .layer = index of the first tail layer; thresh 0: none
};

// This wave's weight fragments of one GEMM phase: Q is fragment-ordered, [KS k-steps][QT tiles][64 lanes] x 16 B; tiles
// t0 .. t0 + NT - 1.  Requested long before they are used (the phases of a workgroup run one after the other: whatever a phase
// waits for is exposed), so the loads are pinned where they are written (left alone the scheduler sinks them to four in flight).
template <int NT, int KS, int QT, int KS0 = 0, int KS1 = KS>
__device__ __forceinline__ void load_q(u32x4_t (&qf)[KS][NT], const uint16_t* __restrict__ Q, const int t0, const int l) {
    const u32x4_t* qp = (const u32x4_t*)Q + t0 * 64 + l;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = KS0; ks < KS1; ++ks)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) qf[ks][ni] = qp[(ks * QT + ni) * 64];
    __builtin_amdgcn_sched_barrier(0);
}
// acc[mi][ni] += A(rows mi*16.., K-contiguous in LDS, stride SA) . Q fragments
template <bool F16, int NT, int KS, int SA>
__device__ __forceinline__ void mma(f32x4_t (&acc)[4][NT], const MGEMM_LDS char* A, const u32x4_t (&qf)[KS][NT], const int l) {
    typedef Elem<F16> E;
    const uint32_t aoff = (uint32_t)((l & 15) * SA + 16 * (l >> 4));
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        u32x4_t af[4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) af[mi] = *(const MGEMM_LDS u32x4_t*)(A + aoff + mi * 16 * SA + ks * 64);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < NT; ++ni) acc[mi][ni] = E::mfma(qf[ks][ni], af[mi], acc[mi][ni]);
    }
}
// workgroup barrier that waits for this wave's LDS traffic only: global loads requested for a LATER phase stay in flight
// (__syncthreads waits for vmcnt(0) as well)
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int NT>
__device__ __forceinline__ void zero_acc(f32x4_t (&acc)[4][NT]) {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) acc[mi][ni] = f32x4_t{0.f, 0.f, 0.f, 0.f};
}

__device__ __forceinline__ uint64_t key_of(const DropArgs& d, int layer) {
    DropArgs t = d;
    t.layer = layer;
    return drop_key(t);
}
__device__ __forceinline__ bool pos16(uint32_t h) { return h - 1u < 0x7FFFu; }      // a positive 16-bit float pattern

// forward epilogue: y = [Dropout](round16(relu(acc + bias))) -> LDS (stride S) and, when yg, global [B, W]
template <bool F16, int NT, int S>
__device__ __forceinline__ void epi_fwd(const f32x4_t (&acc)[4][NT], const float4 (&bias)[NT], const int q0, const int l,
                                        MGEMM_LDS char* Y, uint16_t* __restrict__ yg, const int W, const int64_t row0,
                                        const DropArgs& d, const uint64_t dkey) {
    typedef Elem<F16> E;
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
        const int q = q0 + ni * 16 + 4 * (l >> 4);
        const float4 bq = bias[ni];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int p = mi * 16 + (l & 15);
            f32x4_t v = acc[mi][ni];
            v[0] += bq.x; v[1] += bq.y; v[2] += bq.z; v[3] += bq.w;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
            u32x2_t o = {E::pack2(v[0], v[1]), E::pack2(v[2], v[3])};
            if (d.thresh) {
                const uint64_t qd = drop_quad(dkey, d.row0 + row0 + p, W, q);
                o[0] = E::pack2(E::widen(o[0] & 0xFFFFu) * d.scale, E::widen(o[0] >> 16) * d.scale);
                o[1] = E::pack2(E::widen(o[1] & 0xFFFFu) * d.scale, E::widen(o[1] >> 16) * d.scale);
                if (!drop_keep(qd, 0, d.thresh)) o[0] &= 0xFFFF0000u;
                if (!drop_keep(qd, 1, d.thresh)) o[0] &= 0x0000FFFFu;
                if (!drop_keep(qd, 2, d.thresh)) o[1] &= 0xFFFF0000u;
                if (!drop_keep(qd, 3, d.thresh)) o[1] &= 0x0000FFFFu;
            }
            *(MGEMM_LDS u32x2_t*)(Y + p * S + q * 2) = o;
            if (yg) *(u32x2_t*)(yg + (row0 + p) * W + q) = o;
        }
    }
}

// input-gradient epilogue: dx = round16(acc [* 1 / keep]) where the activation H (LDS, stride S) is positive, else 0 -> global
// [B, W] and, when INPLACE, over H itself; column sums of the rounded values over the 64 rows -> slab[q]
template <bool F16, int NT, int S, bool INPLACE>
__device__ __forceinline__ void epi_dgrad(const f32x4_t (&acc)[4][NT], const int q0, const int l, MGEMM_LDS char* H,
                                          uint16_t* __restrict__ dxg, const int W, const int64_t row0, float* __restrict__ slab,
                                          const DropArgs& d) {
    typedef Elem<F16> E;
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
        const int q = q0 + ni * 16 + 4 * (l >> 4);
        float cs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int p = mi * 16 + (l & 15);
            f32x4_t v = acc[mi][ni];
            if (d.thresh) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] *= d.scale;
            }
            u32x2_t o = {E::pack2(v[0], v[1]), E::pack2(v[2], v[3])};
            const u32x2_t hb = *(const MGEMM_LDS u32x2_t*)(H + p * S + q * 2);
            if (!pos16(hb[0] & 0xFFFFu)) o[0] &= 0xFFFF0000u;
            if (!pos16(hb[0] >> 16)) o[0] &= 0x0000FFFFu;
            if (!pos16(hb[1] & 0xFFFFu)) o[1] &= 0xFFFF0000u;
            if (!pos16(hb[1] >> 16)) o[1] &= 0x0000FFFFu;
            cs[0] += E::widen(o[0] & 0xFFFFu);
            cs[1] += E::widen(o[0] >> 16);
            cs[2] += E::widen(o[1] & 0xFFFFu);
            cs[3] += E::widen(o[1] >> 16);
            if (INPLACE) *(MGEMM_LDS u32x2_t*)(H + p * S + q * 2) = o;
            *(u32x2_t*)(dxg + (row0 + p) * W + q) = o;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float s = cs[r];
            s += __shfl_xor(s, 1, 64);
            s += __shfl_xor(s, 2, 64);
            s += __shfl_xor(s, 4, 64);
            s += __shfl_xor(s, 8, 64);
            cs[r] = s;
        }
        if ((l & 15) == 0) *(float4*)(slab + q) = make_float4(cs[0], cs[1], cs[2], cs[3]);
    }
}

#ifdef MREC_TAIL_STAMPS
__device__ unsigned long long g_tail_stamps[512][8];
#define TAIL_STAMP(i) do { if (threadIdx.x == 0) g_tail_stamps[blockIdx.x][i] = wall_clock64(); } while (0)
#else
#define TAIL_STAMP(i) do { } while (0)
#endif

template <bool F16>
__global__ __launch_bounds__(TT, 2) void k_tail(const TailArgs a) {
    extern __shared__ __attribute__((aligned(1024))) char smem_raw[];
    MGEMM_LDS char* const sm = (MGEMM_LDS char*)smem_raw;
    const int tid = threadIdx.x, l = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t row0 = (int64_t)blockIdx.x * R;
    const DropArgs& d = a.drop;
    constexpr int CG = N3 / 8, RP = TT / CG, NPASS = R / RP, WR = 4;      // head: 16 lanes per row, 32 rows per pass; <= 64 wide fields
    const int cg = tid % CG, rl = tid / CG;
    TAIL_STAMP(0);

    // ---- requests, in the order they are needed: the X tile (one 1-KB row per wave instruction), the weights of both forward
    // layers, the head's per-sample inputs
    u32x4_t xv[R / 8];
    {
        const uint16_t* xs = a.x + row0 * a.ldx;
#pragma unroll
        for (int i = 0; i < R / 8; ++i) xv[i] = *(const u32x4_t*)(xs + (int64_t)(w * (R / 8) + i) * a.ldx + l * 8);
    }
    float4 bias2[2], bias3[1];             // (before the weights: whatever is requested later is waited for later, in order)
    bias2[0] = *(const float4*)(a.b2 + w * 32 + 4 * (l >> 4));
    bias2[1] = *(const float4*)(a.b2 + w * 32 + 16 + 4 * (l >> 4));
    bias3[0] = *(const float4*)(a.b3 + w * 16 + 4 * (l >> 4));
    const uint64_t dkey1 = d.thresh ? key_of(d, d.layer + 1) : 0ull, dkey2 = d.thresh ? key_of(d, d.layer + 2) : 0ull;
    const float4 w5a = *(const float4*)(a.w5 + cg * 8), w5b = *(const float4*)(a.w5 + cg * 8 + 4);
    const float bias = *a.b5;
    const float wbias = a.wprod ? *a.wide_bias : 0.0f;
    u32x4_t qA[K2 / 32][2], qB[N2 / 32][1];
    load_q<2, K2 / 32, N2 / 16>(qA, a.w2f, w * 2, l);
    load_q<1, N2 / 32, N3 / 16>(qB, a.w3f, w, l);
    // (unconditional loads from clamped addresses into registers of their own: a conditional load makes the compiler wait for
    // EVERYTHING requested so far before it reuses the register)
    float hy[NPASS], hw[NPASS][WR];
    {
        const float* wsrc = a.wprod ? a.wprod : a.wide;
        const int64_t wstride = a.wprod ? 2 * (int64_t)a.F : 1;      // floats per sample
        const int wstep = a.wprod ? 2 : 0;                            // floats per field
        const int fmax = a.wprod ? a.F - 1 : 0;
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            const int64_t r = row0 + pass * RP + rl;
            hy[pass] = a.label[r];
#pragma unroll
            for (int q = 0; q < WR; ++q) {
                const int f = min(q * CG + cg, fmax);
                hw[pass][q] = wsrc[r * wstride + f * wstep];
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < R / 8; ++i) *(MGEMM_LDS u32x4_t*)(sm + OX + (w * (R / 8) + i) * SX + l * 16) = xv[i];
    lds_barrier();
    TAIL_STAMP(1);

    // ---- forward, first tail layer: Y2 = relu(X . W2 + b2), this wave's 32 columns
    {
        f32x4_t acc[4][2];
        zero_acc<2>(acc);
        mma<F16, 2, K2 / 32, SX>(acc, sm + OX, qA, l);
        epi_fwd<F16, 2, S2>(acc, bias2, w * 32, l, sm + O2, a.y2, N2, row0, d, dkey1);
    }
    lds_barrier();
    TAIL_STAMP(2);
    // ---- forward, second tail layer: Y3 = relu(Y2 . W3 + b3), this wave's 16 columns (stays in LDS: only the head reads it);
    // behind its MFMAs the weights of the backward phases are requested (the second half of the last phase's behind the MFMAs
    // of the phase before it: 256 VGPRs do not hold everything at once): they arrive while the head runs
    u32x4_t qC[N3 / 32][2], qD[N2 / 32][4];
    {
        f32x4_t acc[4][1];
        zero_acc<1>(acc);
        mma<F16, 1, N2 / 32, S2>(acc, sm + O2, qB, l);
        load_q<2, N3 / 32, N2 / 16>(qC, a.w3b, w * 2, l);
        load_q<4, N2 / 32, K2 / 16, 0, N2 / 64>(qD, a.w2b, w * 4, l);
        epi_fwd<F16, 1, S3>(acc, bias3, w * 16, l, sm + O3, nullptr, N3, row0, d, dkey2);
    }
    lds_barrier();
    TAIL_STAMP(3);

    // ---- output head on Y3 (k_head_fwd_bwd's per-sample arithmetic, operation for operation); dz4 over Y3 in place
    {
        float* const redw = (float*)(smem_raw + ORED);                  // [8 waves][2][N3]: dW5 and column sums of dh4 per wave
        float* const reds = redw + 8 * 2 * N3;                          // [8 waves][2]: db5, loss per wave
        const float wv5[8] = {w5a.x, w5a.y, w5a.z, w5a.w, w5b.x, w5b.y, w5b.z, w5b.w};
        const float dhs = d.thresh ? d.scale : 1.0f;
        float accw[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float accd[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float accb = 0.0f, accl = 0.0f;
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            const int p = pass * RP + rl;
            const int64_t r = row0 + p;
            MGEMM_LDS u32x4_t* hp = (MGEMM_LDS u32x4_t*)(sm + O3 + p * S3 + cg * 16);
            float fh[8];
            const u32x4_t hv = *hp;
            unpack8t<F16>(make_uint4(hv[0], hv[1], hv[2], hv[3]), fh);
            float part = 0.0f;
#pragma unroll
            for (int k = 0; k < 8; ++k) part += fh[k] * wv5[k];
            for (int dd = CG >> 1; dd >= 1; dd >>= 1) part += __shfl_xor(part, dd, 64);
            float wv;
            if (a.wprod) {                 // ReduceSum over the fields in FIELD ORDER, then + Wide_b (mrec_wide_sum's adds)
                const int lane_row0 = l - cg;
                float acc_w = 0.0f;
#pragma unroll
                for (int q = 0; q < WR; ++q) {
                    const int f0 = q * CG;
                    if (f0 < a.F) {            // (uniform) all 16 lane reads issued together, then the adds in field order
                        float pv[CG];
#pragma unroll
                        for (int c = 0; c < CG; ++c) pv[c] = __shfl(hw[pass][q], lane_row0 + c, 64);
#pragma unroll
                        for (int c = 0; c < CG; ++c)
                            if (f0 + c < a.F) acc_w = acc_w + pv[c];
                    }
                }
                wv = acc_w + wbias;
            } else {
                wv = hw[pass][0];
            }
            const float z = part + bias + wv;
            const float y = hy[pass];
            const float loss = fmaxf(z, 0.0f) - z * y + log1pf(expf(-fabsf(z)));
            const float sg = 1.0f / (1.0f + expf(-z));
            const float dl = (sg - y) * a.dscale;
            if (cg == 0) {
                a.logit[r] = z;
                a.dlogit[r] = dl;
                accl += loss;
                accb += dl;
            }
            float o[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                o[k] = fh[k] > 0.0f ? (dl * wv5[k]) * dhs : 0.0f;
                accw[k] += fh[k] * dl;
                accd[k] += o[k];
            }
            const uint4 ob = pack8t<F16>(o);
            *hp = u32x4_t{ob.x, ob.y, ob.z, ob.w};
            *(uint4*)(a.dz4 + r * N3 + cg * 8) = ob;
        }
        // per-workgroup partials [N3] dW5 | [N3] column sums of dh4 | db5 | loss: over the wave's 4 row groups by shuffles, over the 8
        // waves through LDS -- a fixed order
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            accw[k] += __shfl_xor(accw[k], 16, 64); accw[k] += __shfl_xor(accw[k], 32, 64);
            accd[k] += __shfl_xor(accd[k], 16, 64); accd[k] += __shfl_xor(accd[k], 32, 64);
        }
        accb += __shfl_xor(accb, 16, 64); accb += __shfl_xor(accb, 32, 64);
        accl += __shfl_xor(accl, 16, 64); accl += __shfl_xor(accl, 32, 64);
        if (l < CG) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                redw[(w * 2 + 0) * N3 + cg * 8 + k] = accw[k];
                redw[(w * 2 + 1) * N3 + cg * 8 + k] = accd[k];
            }
            if (l == 0) { reds[w * 2] = accb; reds[w * 2 + 1] = accl; }
        }
        lds_barrier();
        float* pp = a.partial + (int64_t)blockIdx.x * PW;
        if (tid < 2 * N3) {
            const int which = tid / N3, c = tid - which * N3;
            float sacc = redw[which * N3 + c];
#pragma unroll
            for (int q = 1; q < 8; ++q) sacc += redw[(q * 2 + which) * N3 + c];
            pp[which * N3 + c] = sacc;
        } else if (tid < 2 * N3 + 2) {
            const int which = tid - 2 * N3;
            float sacc = reds[which];
#pragma unroll
            for (int q = 1; q < 8; ++q) sacc += reds[q * 2 + which];
            pp[2 * N3 + which] = sacc;
        } else if (tid < 2 * N3 + 4) {
            pp[tid] = 0.0f;
        }
    }
    TAIL_STAMP(4);

    // ---- backward through the second tail layer: dz3 = (dz4 . W3^T) masked by Y2 > 0, over Y2 in place (this wave's 32 columns)
    // (the barrier inside the head stands between the dz4 writes and these reads)
    {
        f32x4_t acc[4][2];
        zero_acc<2>(acc);
        mma<F16, 2, N3 / 32, S3>(acc, sm + O3, qC, l);
        load_q<4, N2 / 32, K2 / 16, N2 / 64, N2 / 32>(qD, a.w2b, w * 4, l);
        epi_dgrad<F16, 2, S2, true>(acc, w * 32, l, sm + O2, a.dz3, N2, row0, a.partial + (int64_t)blockIdx.x * PW + P_DB3, d);
    }
    lds_barrier();
    TAIL_STAMP(5);
    // ---- backward through the first tail layer: dz2 = (dz3 . W2^T) masked by X > 0 (this wave's 64 columns)
    {
        f32x4_t acc[4][4];
        zero_acc<4>(acc);
        mma<F16, 4, N2 / 32, S2>(acc, sm + O2, qD, l);
        epi_dgrad<F16, 4, SX, false>(acc, w * 64, l, sm + OX, a.dz2, K2, row0, a.partial + (int64_t)blockIdx.x * PW + P_DB2, d);
    }
#ifdef MREC_TAIL_STAMPS
    __syncthreads();
    TAIL_STAMP(6);
#endif
}

// the partial rows of all workgroups added up in workgroup order (finish_column: 8 row groups x 8 loads in flight per column)
__global__ __launch_bounds__(MB) void k_tail_finish(const float* __restrict__ partial, int nblk, float inv_B, float* __restrict__ dw5,
                                                    float* __restrict__ db4, float* __restrict__ db5, float* __restrict__ loss,
                                                    float* __restrict__ dwide_b, float* __restrict__ db3, float* __restrict__ db2) {
    __shared__ float sm[8][32];
    const int c = blockIdx.x * 32 + (threadIdx.x & 31);
    const float sacc = finish_column(partial, nblk, PW, c, sm);
    if ((threadIdx.x >> 5) != 0 || c >= PW) return;
    if (c < N3) dw5[c] = sacc;
    else if (c < 2 * N3) db4[c - N3] = sacc;
    else if (c == 2 * N3) {
        *db5 = sacc;
        if (dwide_b) *dwide_b = sacc;
    } else if (c == 2 * N3 + 1) *loss = sacc * inv_B;
    else if (c >= P_DB3 && c < P_DB2) db3[c - P_DB3] = sacc;
    else if (c >= P_DB2) db2[c - P_DB2] = sacc;
}

// Fragment-ordered copies of one weight W [K, N] (row-major, as the reference stores it): one thread per 16-byte chunk.
//   forward operand  F[ks][nt][l][j] = W[ks*32 + 8*(l>>4) + j][nt*16 + (l&15)]     (output column q = n, reduction over k)
//   backward operand G[ks][kt][l][j] = W[kt*16 + (l&15)][ks*32 + 8*(l>>4) + j]     (output column q = k, reduction over n)
__device__ __forceinline__ void tail_pack_chunk(const uint16_t* __restrict__ w2, const uint16_t* __restrict__ w3, uint4* __restrict__ out, int i) {
    constexpr int C2 = K2 * N2 / 8, C3 = N2 * N3 / 8;            // chunks per copy
    if (i >= 2 * (C2 + C3)) return;
    const uint4* dst = out + i;
    // order of the copies in `out`: W2 forward | W3 forward | W3 backward | W2 backward
    const uint16_t* W; int K, N; bool fwd;
    if (i < C2) { W = w2; K = K2; N = N2; fwd = true; }
    else if (i < C2 + C3) { i -= C2; W = w3; K = N2; N = N3; fwd = true; }
    else if (i < C2 + 2 * C3) { i -= C2 + C3; W = w3; K = N2; N = N3; fwd = false; }
    else { i -= C2 + 2 * C3; W = w2; K = K2; N = N2; fwd = false; }
    const int l = i & 63, t = i >> 6;
    uint4 v;
    if (fwd) {
        const int QT = N / 16, ks = t / QT, nt = t - ks * QT;
        const uint16_t* p = W + (int64_t)(ks * 32 + 8 * (l >> 4)) * N + nt * 16 + (l & 15);
        uint32_t e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = p[(int64_t)j * N];
        v = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
    } else {
        const int QT = K / 16, ks = t / QT, kt = t - ks * QT;
        v = *(const uint4*)(W + (int64_t)(kt * 16 + (l & 15)) * N + ks * 32 + 8 * (l >> 4));
    }
    *(uint4*)dst = v;
}
__global__ __launch_bounds__(256) void k_tail_pack(const uint16_t* __restrict__ w2, const uint16_t* __restrict__ w3, uint4* __restrict__ out) {
    tail_pack_chunk(w2, w3, out, blockIdx.x * 256 + threadIdx.x);
}

// Every derived copy of the 16-bit weights in ONE launch (each launch costs ~4 us of a step that is one dependent chain): the
// tail launch's fragment-ordered weights, and the TRANSPOSES [out, in] of the other hidden layers' weights -- with them the
// forward GEMM reads both operands K-contiguous (ds_read_b128), 15 % faster than through ds_read_b64_tr_b16 on W as stored.
struct TransDesc { const uint16_t* src; uint16_t* dst; int rows, cols, tiles_c, first_block; };
struct CopyArgs { TransDesc t[4]; int n_t, pack_blocks; const uint16_t *w2, *w3; uint4* packed; };
__global__ __launch_bounds__(256) void k_operand_copies(const CopyArgs a) {
    int b = blockIdx.x;
    if (b < a.pack_blocks) { tail_pack_chunk(a.w2, a.w3, a.packed, b * 256 + threadIdx.x); return; }
    b -= a.pack_blocks;
    int q = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k)
        if (k < a.n_t && b >= a.t[k].first_block) q = k;
    const TransDesc d = a.t[q];
    b -= d.first_block;
    // 64 x 64 tile through LDS: 16-byte loads along the rows, 16-byte stores along the columns
    __shared__ uint16_t tile[64][72];
    const int r0 = (b / d.tiles_c) * 64, c0 = (b % d.tiles_c) * 64;
    const int tr = threadIdx.x >> 3, tc = (threadIdx.x & 7) * 8;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = r0 + tr + 32 * i, c = c0 + tc;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (r < d.rows && c < d.cols) v = *(const uint4*)(d.src + (int64_t)r * d.cols + c);       // cols % 8 == 0
        *(uint4*)&tile[tr + 32 * i][tc] = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = c0 + tr + 32 * i, r = r0 + tc;            // output row c, 8 consecutive source rows r .. r + 7
        if (c < d.cols && r < d.rows) {
            uint32_t e[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) e[j] = tile[tc + j][tr + 32 * i];
            *(uint4*)(d.dst + (int64_t)c * d.rows + r) = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
        }
    }
}
}  // namespace tail

MREC_API int mrec_tail_packed_elems(int32_t K2, int32_t N2, int32_t N3, int64_t* out) {
    if (!out) return MREC_EINVAL;
    if (K2 != tail::K2 || N2 != tail::N2 || N3 != tail::N3) return MREC_EUNSUPPORTED;
    *out = 2 * ((int64_t)K2 * N2 + (int64_t)N2 * N3);
    return MREC_OK;
}
MREC_API int mrec_tail_pack_weights(const uint16_t* w2, const uint16_t* w3, int32_t K2, int32_t N2, int32_t N3, uint16_t* packed,
                                    void* stream) {
    if (K2 != tail::K2 || N2 != tail::N2 || N3 != tail::N3) return MREC_EUNSUPPORTED;
    if (!w2 || !w3 || !packed) return MREC_EINVAL;
    if (((uintptr_t)w2 | (uintptr_t)w3 | (uintptr_t)packed) & 15) return MREC_EUNSUPPORTED;
    const int n = 2 * (K2 * N2 + N2 * N3) / 8;
    tail::k_tail_pack<<<(unsigned)mrec_cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(w2, w3, (uint4*)packed);
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}

MREC_API int mrec_dense_operand_copies(int32_t n_t, const mrec_transpose_t* t, const uint16_t* tail_w2, const uint16_t* tail_w3,
                                       int32_t K2, int32_t N2, int32_t N3, uint16_t* tail_packed, void* stream) {
    if (n_t < 0 || n_t > 4 || (n_t && !t)) return MREC_EINVAL;
    tail::CopyArgs a{};
    int blocks = 0;
    if (tail_packed) {
        if (K2 != tail::K2 || N2 != tail::N2 || N3 != tail::N3) return MREC_EUNSUPPORTED;
        if (!tail_w2 || !tail_w3) return MREC_EINVAL;
        if (((uintptr_t)tail_w2 | (uintptr_t)tail_w3 | (uintptr_t)tail_packed) & 15) return MREC_EUNSUPPORTED;
        a.w2 = tail_w2; a.w3 = tail_w3; a.packed = (uint4*)tail_packed;
        a.pack_blocks = (int)mrec_cdiv(2 * (K2 * N2 + N2 * N3) / 8, 256);
        blocks = a.pack_blocks;
    }
    int tb = 0;
    for (int k = 0; k < n_t; ++k) {
        if (!t[k].src || !t[k].dst || t[k].rows <= 0 || t[k].cols <= 0 || t[k].rows > (1 << 20) || t[k].cols > (1 << 20)) return MREC_EINVAL;
        if (t[k].rows % 8 || t[k].cols % 8 || (((uintptr_t)t[k].src | (uintptr_t)t[k].dst) & 15)) return MREC_EUNSUPPORTED;
        a.t[k].src = t[k].src; a.t[k].dst = t[k].dst; a.t[k].rows = (int)t[k].rows; a.t[k].cols = (int)t[k].cols;
        a.t[k].tiles_c = (int)mrec_cdiv(t[k].cols, 64);
        a.t[k].first_block = tb;
        tb += (int)mrec_cdiv(t[k].rows, 64) * a.t[k].tiles_c;
    }
    a.n_t = n_t;
    blocks += tb;
    if (blocks == 0) return MREC_OK;
    tail::k_operand_copies<<<blocks, 256, 0, (hipStream_t)stream>>>(a);
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}

#ifdef MREC_TAIL_STAMPS
MREC_API int mrec_tail_debug_stamps(unsigned long long* host_out) {
    MREC_HIP_CHECK(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(tail::g_tail_stamps), sizeof(unsigned long long) * 512 * 8));
    return MREC_OK;
}
#endif

MREC_API int mrec_tail_supported(int64_t B, int32_t K2, int32_t N2, int32_t N3) {
    return B > 0 && B % tail::R == 0 && B / tail::R <= 512 && K2 == tail::K2 && N2 == tail::N2 && N3 == tail::N3;
}

MREC_API int mrec_tail_fwd_bwd(int32_t f16, const uint16_t* x, int64_t ldx, const uint16_t* packed, const float* b2,
                               const float* b3, const float* w5, const float* b5,
                               const float* wide, const float* wide_prod, int32_t F, const float* wide_bias, const float* label,
                               int64_t B, int32_t K2, int32_t N2, int32_t N3, float dscale, uint16_t* y2, uint16_t* dz4,
                               uint16_t* dz3, uint16_t* dz2, float* logit, float* dlogit, float* dw5, float* db4, float* db5,
                               float* dwide_bias, float* loss, float* db3, float* db2, void* ws, size_t ws_bytes,
                               const mrec_dropout_t* drop_in, void* stream) {
    if (B <= 0 || ldx < K2) return MREC_EINVAL;
    if (!mrec_tail_supported(B, K2, N2, N3)) return MREC_EUNSUPPORTED;
    if (!x || !packed || !b2 || !b3 || !w5 || !b5 || (!wide && !wide_prod) || !label || !y2 || !dz4 || !dz3 || !dz2 ||
        !logit || !dlogit || !dw5 || !db4 || !db5 || !loss || !db3 || !db2 || !ws)
        return MREC_EINVAL;
    if (((uintptr_t)w5) & 15) return MREC_EUNSUPPORTED;
    if (wide_prod && (F <= 0 || !wide_bias)) return MREC_EINVAL;
    if (wide_prod && F > 64) return MREC_EUNSUPPORTED;
    if (ldx % 8 || (((uintptr_t)x | (uintptr_t)packed | (uintptr_t)y2 | (uintptr_t)dz4 |
                     (uintptr_t)dz3 | (uintptr_t)dz2 | (uintptr_t)b2 | (uintptr_t)b3 | (uintptr_t)ws) & 15))
        return MREC_EUNSUPPORTED;
    const int nb = (int)(B / tail::R);
    if (ws_bytes < (size_t)nb * tail::PW * sizeof(float)) return MREC_EWORKSPACE;
    tail::TailArgs a{};
    if (!drop_from(drop_in, 4, &a.drop)) return MREC_EINVAL;
    if (a.drop.thresh && a.drop.layer + 2 > 15) return MREC_EINVAL;
    a.x = x; a.ldx = ldx;
    a.w2f = packed; a.w3f = a.w2f + (int64_t)K2 * N2; a.w3b = a.w3f + (int64_t)N2 * N3; a.w2b = a.w3b + (int64_t)N2 * N3;
    a.b2 = b2; a.b3 = b3; a.w5 = w5; a.b5 = b5;
    a.wide = wide_prod ? nullptr : wide; a.wprod = wide_prod; a.wide_bias = wide_bias; a.F = F; a.label = label;
    a.B = B; a.dscale = dscale; a.y2 = y2; a.dz4 = dz4; a.dz3 = dz3; a.dz2 = dz2; a.logit = logit; a.dlogit = dlogit;
    a.partial = (float*)ws;
    hipStream_t st = (hipStream_t)stream;
    static bool attr_set = false;
    if (!attr_set) {
        MREC_HIP_CHECK(hipFuncSetAttribute((const void*)tail::k_tail<false>, hipFuncAttributeMaxDynamicSharedMemorySize, tail::LDS_BYTES));
        MREC_HIP_CHECK(hipFuncSetAttribute((const void*)tail::k_tail<true>, hipFuncAttributeMaxDynamicSharedMemorySize, tail::LDS_BYTES));
        attr_set = true;
    }
    if (f16) tail::k_tail<true><<<nb, tail::TT, tail::LDS_BYTES, st>>>(a);
    else tail::k_tail<false><<<nb, tail::TT, tail::LDS_BYTES, st>>>(a);
    tail::k_tail_finish<<<(unsigned)mrec_cdiv(tail::PW, 32), MB, 0, st>>>((const float*)ws, nb, 1.0f / (float)B, dw5, db4, db5, loss, dwide_bias,
                                                                        db3, db2);
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}
MREC_API int mrec_tail_workspace_bytes(int64_t B, size_t* out) {
    if (!out || B < 0) return MREC_EINVAL;
    *out = (size_t)mrec_cdiv(B > 0 ? B : 1, tail::R) * tail::PW * sizeof(float);
    return MREC_OK;
}
