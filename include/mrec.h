/*
 * mrec.h -- C-ABI of libmrec_hip.so, the MI355X (gfx950) replacement for the MindSpore
 * primitives that mindspore-lab/mindrec invokes on its embedding hot path.
 *
 * The reference's boundary for this path is a *Python operator* boundary (Primitive.__call__ on
 * framework-owned tensors); there is no FFI in /root/reference to mirror, so every entry point
 * below cites the reference call site whose MindSpore primitive it replaces.  INTEGRATION.md
 * shows the ctypes binding a maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (e.g. a torch tensor's data_ptr())
 *     unless the name ends in `_host`;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*), never synchronises,
 *     never allocates: scratch comes from the caller through (ws, ws_bytes), sized by the
 *     matching *_workspace_bytes query, so calls are hipGraph-capturable;
 *   - return value: 0 = MREC_OK, <0 = error code below; nothing throws;
 *   - ids/keys are int32 (`_i32`) or int64 (`_i64`); rows are fp32; `ld` = row stride in floats;
 *   - counts that are only known on the device (number of unique ids) live in device int64 words.
 */
#ifndef MREC_H_
#define MREC_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MREC_OK 0
#define MREC_EINVAL (-1)       /* bad argument (null pointer, negative size, misaligned row) */
#define MREC_EWORKSPACE (-2)   /* workspace too small */
#define MREC_EUNSUPPORTED (-3) /* shape outside what the kernels cover */
#define MREC_EHIP (-4)         /* a HIP runtime call failed (see mrec_last_hip_error) */
#define MREC_ENODEVICE (-5)    /* no gfx950 device / kernel image not loadable */

const char* mrec_strerror(int code);
int mrec_last_hip_error(void);
int mrec_version(void);
/* 0 when a HIP device is visible and the gfx950 code object loads; MREC_ENODEVICE otherwise. */
int mrec_device_ok(void);
/* CRC-32C of a HOST buffer (TFRecord checksums: the reference's second data format, models/wide_deep/src/datasets.py:226-271; host code,
 * data preparation). */
int mrec_crc32c_host(const void* data, size_t n, uint32_t* out);

/* ---- table initialisation --------------------------------------------------------------
 * initializer('normal') of nn.EmbeddingLookup / MapParameter default_value
 * (models/wide_deep/default_config.yaml:41; mindspore_rec/ops/embedding.py:88,141).
 * out[r, c] = sigma * N01(seed, row0 + r * row_stride, c) from the counter-based generator of
 * csrc/mrec_rng.h (bit-identical to the oracle's), written straight into HBM.  A row shard
 * (owner = id mod n) passes row0 = rank, row_stride = n so values depend on the GLOBAL row only. */
int mrec_fill_normal_f32(float* out, int64_t nrows, int32_t D, int64_t ld, uint64_t seed,
                         int64_t row0, int64_t row_stride, float sigma, void* stream);

/* ---- ops.Unique --------------------------------------------------------------------------
 * mindspore_rec/ops/embedding.py:153,192; models/wide_deep/src/wide_and_deep.py:212.
 * uniq keeps first-occurrence order (MindSpore CPU kernel), uniq[inv[i]] == ids[i].
 * uniq has room for n entries; *n_uniq_dev receives U. */
int mrec_dedup_workspace_bytes(int64_t n, size_t* out);
int mrec_dedup_i32(const int32_t* ids, int64_t n, int32_t* uniq, int32_t* inv, int64_t* n_uniq_dev,
                   void* ws, size_t ws_bytes, void* stream);
int mrec_dedup_i64(const int64_t* ids, int64_t n, int64_t* uniq, int32_t* inv, int64_t* n_uniq_dev,
                   void* ws, size_t ws_bytes, void* stream);

/* ---- inverted index of a Unique result ---------------------------------------------------
 * Groups positions 0..n-1 by inv[] (stable: ascending position inside each group).  This is the
 * index the optimizer-side RowTensor dedup (Unique + UnsortedSegmentSum) needs:
 * sorted_seg[e] = inv[sorted_pos[e]] non-decreasing; seg_offsets[u] = first e of group u,
 * seg_offsets[U] = n  (seg_offsets has room for n+1 entries). */
int mrec_group_workspace_bytes(int64_t n, size_t* out);
int mrec_group_by_inverse(const int32_t* inv, int64_t n, int32_t* sorted_pos, int32_t* sorted_seg,
                          int32_t* seg_offsets, void* ws, size_t ws_bytes, void* stream);

/* Unique + inverted index in one call -- what one training step needs from its id tensor (the forward's
 * Unique, embedding.py:192, and the optimizer-side RowTensor dedup share it).  Same outputs as
 * mrec_dedup_* followed by mrec_group_by_inverse, one pass and one launch fewer. */
int mrec_sparse_plan_workspace_bytes(int64_t n, size_t* out);
int mrec_sparse_plan_i32(const int32_t* ids, int64_t n, int32_t* uniq, int32_t* inv, int64_t* n_uniq_dev,
                         int32_t* sorted_pos, int32_t* sorted_seg, int32_t* seg_offsets, void* ws,
                         size_t ws_bytes, void* stream);
int mrec_sparse_plan_i64(const int64_t* ids, int64_t n, int64_t* uniq, int32_t* inv, int64_t* n_uniq_dev,
                         int32_t* sorted_pos, int32_t* sorted_seg, int32_t* seg_offsets, void* ws,
                         size_t ws_bytes, void* stream);
/* Same with flags.  MREC_PLAN_WS_PRIMED: the caller vouches that the last thing that wrote `ws` was a COMPLETED
 * mrec_sparse_plan_* call with the same n (any flags) and the same ws pointer: the scratch hash table and the scan's
 * look-back words are then already clean -- every call hands them back clean (the table by a memset BEHIND its last kernel) --
 * and the two memsets at the head of the chain are skipped: the first kernel starts 10 us earlier.  (A training step replays the same plan on the same workspace every step.) */
#define MREC_PLAN_WS_PRIMED 1u
/* MREC_PLAN_SKIP_NEGATIVE: negative ids are padding, not keys (the unused slots of a shard's fixed-capacity request
 * message, mrec_shard_route_slots_*): they get no group (inv = -1) and no entry of the index proper, which is then the first
 * n_valid = (number of non-negative ids) entries of sorted_pos / sorted_seg, seg_offsets[U] = n_valid; the padding positions
 * follow as a pseudo-group U whose row uniq[U] is -1.  n_uniq_dev must then point at TWO words: [0] = U, [1] = n_valid --
 * the count the apply kernels clamp their n to (n_valid_dev of mrec_sparse_lazy_adam_wide). */
#define MREC_PLAN_SKIP_NEGATIVE 2u
int mrec_sparse_plan_ex_i32(const int32_t* ids, int64_t n, int32_t* uniq, int32_t* inv, int64_t* n_uniq_dev,
                            int32_t* sorted_pos, int32_t* sorted_seg, int32_t* seg_offsets, void* ws, size_t ws_bytes,
                            uint32_t flags, void* stream);
int mrec_sparse_plan_ex_i64(const int64_t* ids, int64_t n, int64_t* uniq, int32_t* inv, int64_t* n_uniq_dev,
                            int32_t* sorted_pos, int32_t* sorted_seg, int32_t* seg_offsets, void* ws, size_t ws_bytes,
                            uint32_t flags, void* stream);

/* ---- ops.Gather / SparseGatherV2 / EmbeddingLookup ----------------------------------------
 * mindspore_rec/ops/embedding.py:150,194; models/deep_and_cross/src/deep_and_cross.py:199;
 * nn.EmbeddingLookup at models/wide_deep/src/wide_and_deep.py:277-290.
 * out[i, :] = table[ids[i], :] * (row_scale ? row_scale[i] : 1); ids outside [0,V) give zeros.
 * row_scale fuses the mask multiply of wide_and_deep.py:303,308 (nullable).  out is [n, D]
 * contiguous. */
int mrec_gather_rows_f32_i32(const float* table, int64_t V, int64_t ld, int32_t D, const int32_t* ids,
                             int64_t n, const float* row_scale, float* out, void* stream);
int mrec_gather_rows_f32_i64(const float* table, int64_t V, int64_t ld, int32_t D, const int64_t* ids,
                             int64_t n, const float* row_scale, float* out, void* stream);

/* Same gather with bf16 output rows (round-to-nearest-even): fuses the Cast(x, float16) that
 * DenseLayer.construct applies to the masked embeddings (wide_and_deep.py:122) -- bf16 on MI355X.
 * D % 4 == 0 and D <= 256, or D <= 64. */
int mrec_gather_rows_bf16_i32(const float* table, int64_t V, int64_t ld, int32_t D, const int32_t* ids,
                              int64_t n, const float* row_scale, uint16_t* out, void* stream);
int mrec_gather_rows_bf16_i64(const float* table, int64_t V, int64_t ld, int32_t D, const int64_t* ids,
                              int64_t n, const float* row_scale, uint16_t* out, void* stream);

/* ... and with IEEE half output rows: exactly the Cast(x, float16) of wide_and_deep.py:122. */
int mrec_gather_rows_f16_i32(const float* table, int64_t V, int64_t ld, int32_t D, const int32_t* ids,
                             int64_t n, const float* row_scale, uint16_t* out, void* stream);
int mrec_gather_rows_f16_i64(const float* table, int64_t V, int64_t ld, int32_t D, const int64_t* ids,
                             int64_t n, const float* row_scale, uint16_t* out, void* stream);

/* Deep lookup + the wide branch's products in one pass over FUSED rows [p(D) | w accum linear pad | ...] (both lookups of
 * WideDeepModel.construct read the same id tensor, wide_and_deep.py:300-302): out as mrec_gather_rows_{bf16,f16} (out_kind 1 /
 * 2), and wide_prod[2 i] = table[ids[i], wide_col] * row_scale[i] (0 for ids outside [0, V); wide_prod[2 i + 1] = 0: the
 * products are stored as 8-byte pairs).  wide_col must equal D (the wide word right behind the deep columns), D % 4 == 0,
 * D <= 252.  The per-sample sum over the fields + bias is taken by mrec_head_fwd_bwd_wide in field order: same adds, same
 * order as mrec_wide_sum.  ldo: row stride of `out` in 16-bit elements (D for a plain [n, D] result),
 * ldw: stride of wide_prod in floats (2 for the plain [n, 2] result) -- a shard's answer message packs both into one row
 * [D 16-bit values | product, 0 | pad] by pointing wide_prod at column D / 2 of the same rows.
 * drop (nullable; mrec_dropout_t, declared with the DenseLayer entries below) + fields: the ids are [n / fields, fields] and the
 * looked-up rows the [fields * D] input of DenseLayer drop->layer: Dropout is applied to the rounded rows on their way out
 * (x * (1 / keep_prob), rounded again: what mrec_dropout would do in a pass of its own). */
struct mrec_dropout;
int mrec_gather_rows_wide(const float* table, int64_t V, int64_t ld, int32_t D, const void* ids, int32_t id_bytes, int64_t n,
                          const float* row_scale, void* out, int32_t out_kind, int64_t ldo, int32_t wide_col, float* wide_prod,
                          int64_t ldw, const struct mrec_dropout* drop, int32_t fields, void* stream);
/* The same pass serving a shard's request message in place (the owner's side of nn.EmbeddingLookup(..., slice_mode=
 * TABLE_ROW_SLICE), wide_and_deep.py:232-249): ids[i * id_stride] and row_scale[i * scale_stride] let ids and weights be read
 * straight out of the received {id, weight} entries; out_kind 0 adds fp32 rows (ldo then in floats; the fp32 wire format of
 * an fp32 net); MREC_GATHER_SKIP_INVALID leaves the rows of ids outside [0, V) alone instead of writing zeros (the padding
 * slots of a fixed-capacity message: nobody reads their answers).  step_state (nullable, an mrec_step_state_t): the kernel
 * leaves its begin / end wall-clock stamps in stamps_aux[(step + 1) % ring] (measurement only). */
#define MREC_GATHER_SKIP_INVALID 1u
int mrec_gather_rows_wide_ex(const float* table, int64_t V, int64_t ld, int32_t D, const void* ids, int32_t id_bytes,
                             int64_t id_stride, int64_t n, const float* row_scale, int64_t scale_stride, void* out, int32_t out_kind,
                             int64_t ldo, int32_t wide_col, float* wide_prod, int64_t ldw, const struct mrec_dropout* drop,
                             int32_t fields, uint32_t flags, void* step_state, void* stream);

/* Wide branch of WideDeepModel.construct (wide_and_deep.py:300,303-306) in one pass:
 * out[b] = sum_f w[ids[b,f] * ldw] * wts[b,f] + *bias_dev   (w is the [V,1] wide table, row
 * stride ldw floats: 1 for a dense column, 4 when it lives in a fused w|accum|linear|pad record). */
int mrec_wide_sum_f32_i32(const float* w, int64_t V, int64_t ldw, const int32_t* ids, const float* wts, int64_t B,
                          int32_t F, const float* bias_dev, float* out, void* stream);
int mrec_wide_sum_f32_i64(const float* w, int64_t V, int64_t ldw, const int64_t* ids, const float* wts, int64_t B,
                          int32_t F, const float* bias_dev, float* out, void* stream);

/* ---- sparse gradient apply ----------------------------------------------------------------
 * All three take the inverted index (sorted_pos, sorted_seg, seg_offsets) of the step's ids and
 * the per-position row gradients g[n, D] (row stride ldg).  Contribution i is
 * (g[i,:] * row_scale[i]) * grad_scale, summed per unique id in ascending position order for
 * groups that fit one window (mrec_sparse_apply_window), and as a fixed-order tree of window partials otherwise
 * (bitwise reproducible run to run either way).  uniq maps group -> table row. */
int mrec_sparse_apply_workspace_bytes(int64_t n, int32_t D, size_t* out);
/* Entries per window of the sorted index for a table of width D (8 on every path in this build: the float4 path --
 * D % 4 == 0, rows and row strides 16-byte aligned: aligned16 != 0 -- and the 8-byte / scalar paths).  A group whose run
 * of sorted entries stays inside one window is summed in ascending position order (= the CPU
 * reference order); longer runs are a fixed tree of window partials. */
int mrec_sparse_apply_window(int32_t D, int aligned16);

/* ops.UnsortedSegmentSum (bprop of Gather; models/wide_deep/op_precision.ini:2-3):
 * out[u, :] = sum of contributions of group u, u < U; out is [>=U, D] contiguous. */
int mrec_segment_sum_f32(const int32_t* sorted_pos, const int32_t* sorted_seg, const int32_t* seg_offsets,
                         int64_t n, const float* g, int64_t ldg, const float* row_scale, float grad_scale,
                         int32_t D, float* out, void* ws, size_t ws_bytes, void* stream);
/* The same over 16-bit row gradients (g_kind 1: bf16, 2: IEEE half; ldg in 16-bit elements), widened exactly, summed in fp32. */
int mrec_segment_sum_g16(const int32_t* sorted_pos, const int32_t* sorted_seg, const int32_t* seg_offsets,
                         int64_t n, const void* g, int32_t g_kind, int64_t ldg, const float* row_scale, float grad_scale,
                         int32_t D, float* out, void* ws, size_t ws_bytes, void* stream);

/* nn.LazyAdam on a RowTensor gradient (wide_and_deep.py:420-422).  b1_pow/b2_pow are
 * beta^t AFTER this step's multiply.  Rows outside [0,V) are skipped. */
int mrec_sparse_lazy_adam_f32_i32(float* p, float* m, float* v, int64_t V, int64_t ld, int32_t D,
                                  const int32_t* uniq, const int32_t* sorted_pos, const int32_t* sorted_seg,
                                  const int32_t* seg_offsets, int64_t n, const float* g, int64_t ldg,
                                  const float* row_scale, float lr, float b1, float b2, float eps,
                                  float b1_pow, float b2_pow, float grad_scale, int nesterov, void* ws,
                                  size_t ws_bytes, void* stream);
int mrec_sparse_lazy_adam_f32_i64(float* p, float* m, float* v, int64_t V, int64_t ld, int32_t D,
                                  const int64_t* uniq, const int32_t* sorted_pos, const int32_t* sorted_seg,
                                  const int32_t* seg_offsets, int64_t n, const float* g, int64_t ldg,
                                  const float* row_scale, float lr, float b1, float b2, float eps,
                                  float b1_pow, float b2_pow, float grad_scale, int nesterov, void* ws,
                                  size_t ws_bytes, void* stream);

/* LazyAdam with bf16 row gradients g[n, D] (what the mixed-precision MLP's backward hands back;
 * the Cast bprop would widen them to fp32 first -- the kernel widens on load, bit-identically). */
int mrec_sparse_lazy_adam_bf16g_i32(float* p, float* m, float* v, int64_t V, int64_t ld, int32_t D,
                                    const int32_t* uniq, const int32_t* sorted_pos, const int32_t* sorted_seg,
                                    const int32_t* seg_offsets, int64_t n, const uint16_t* g, int64_t ldg,
                                    const float* row_scale, float lr, float b1, float b2, float eps,
                                    float b1_pow, float b2_pow, float grad_scale, int nesterov, void* ws,
                                    size_t ws_bytes, void* stream);
int mrec_sparse_lazy_adam_bf16g_i64(float* p, float* m, float* v, int64_t V, int64_t ld, int32_t D,
                                    const int64_t* uniq, const int32_t* sorted_pos, const int32_t* sorted_seg,
                                    const int32_t* seg_offsets, int64_t n, const uint16_t* g, int64_t ldg,
                                    const float* row_scale, float lr, float b1, float b2, float eps,
                                    float b1_pow, float b2_pow, float grad_scale, int nesterov, void* ws,
                                    size_t ws_bytes, void* stream);

/* LazyAdam with IEEE half row gradients (widened exactly on load). */
int mrec_sparse_lazy_adam_f16g_i32(float* p, float* m, float* v, int64_t V, int64_t ld, int32_t D,
                                   const int32_t* uniq, const int32_t* sorted_pos, const int32_t* sorted_seg,
                                   const int32_t* seg_offsets, int64_t n, const uint16_t* g, int64_t ldg,
                                   const float* row_scale, float lr, float b1, float b2, float eps,
                                   float b1_pow, float b2_pow, float grad_scale, int nesterov, void* ws,
                                   size_t ws_bytes, void* stream);
int mrec_sparse_lazy_adam_f16g_i64(float* p, float* m, float* v, int64_t V, int64_t ld, int32_t D,
                                   const int64_t* uniq, const int32_t* sorted_pos, const int32_t* sorted_seg,
                                   const int32_t* seg_offsets, int64_t n, const uint16_t* g, int64_t ldg,
                                   const float* row_scale, float lr, float b1, float b2, float eps,
                                   float b1_pow, float b2_pow, float grad_scale, int nesterov, void* ws,
                                   size_t ws_bytes, void* stream);

/* LazyAdam on the deep columns AND FTRL on the wide record of the same fused rows, one pass: the wide table's
 * Unique + UnsortedSegmentSum + FusedSparseFtrl (wide_and_deep.py:423-430) ride the deep table's row visit.  Row layout:
 * p at column 0 (pointer p), the wide record [w, accum, linear, pad] as one float4 at column wide_col of the same rows
 * (relative to p), m and v wherever their pointers say; all with row stride ld.  The wide gradient of position i is
 * gw[i / F] (the head's dlogit of sample i / F: Mul bprop of wide_and_deep.py:304), scaled by row_scale[i] * grad_scale and
 * summed per id in the same window / tree order as the deep columns.  uniq_bytes 4 / 8; g_kind 0 f32, 1 bf16, 2 f16;
 * D % 4 == 0, D <= 252, wide_col == D (the record right behind p: it is loaded and stored by the instruction that moves p),
 * 16-byte aligned rows (128-byte aligned rows avoid a second line per record), n * F < 2^32.  gw_stride: distance in floats
 * between consecutive gradients (1 for a plain array); F == 1 gives every position its own gradient gw[i * gw_stride] (a
 * shard's owner: the received positions are not grouped by sample, and the gradient is a column of the gradient message).
 * ws: mrec_sparse_apply_workspace_bytes(n, D + 4).
 * step_state (nullable): an mrec_step_state_t in device memory; when given, the Adam step size comes from it instead of
 * b1_pow / b2_pow, and the main kernel leaves its begin / end wall clock stamps there.
 * n_valid_dev (nullable): a device word holding the number of entries of the index proper (MREC_PLAN_SKIP_NEGATIVE); the
 * kernels walk min(n, *n_valid_dev) entries -- the size of a shard's received batch is known on the device only. */
int mrec_sparse_lazy_adam_wide(float* p, float* m, float* v, int64_t V, int64_t ld, int32_t D, const void* uniq,
                               int32_t uniq_bytes, const int32_t* sorted_pos, const int32_t* sorted_seg,
                               const int32_t* seg_offsets, int64_t n, const void* g, int32_t g_kind, int64_t ldg,
                               const float* row_scale, float lr, float b1, float b2, float eps, float b1_pow, float b2_pow,
                               float grad_scale, int nesterov, const float* gw, int64_t gw_stride, int32_t F, int32_t wide_col, float ftrl_lr,
                               float l1, float l2, float lr_power, void* ws, size_t ws_bytes, void* step_state,
                               const int64_t* n_valid_dev, void* stream);

/* ---- step scalars in device memory ------------------------------------------------------------------------------
 * nn.Adam / nn.LazyAdam keep beta1_power and beta2_power as Parameters that the optimizer's own graph multiplies by
 * beta1 / beta2 every step (mindspore.nn.Adam: beta1_power = beta1_power * beta1; wide_and_deep.py:420-422 builds the
 * optimizers).  The same here: the powers live in device memory and a one-thread kernel advances them, so every kernel
 * argument of a training step is constant and the whole step replays as one HIP graph.  Layout (little endian): */
#define MREC_STAMP_RING 256
typedef struct mrec_step_state {
    float beta1_power, beta2_power;   /* after `step` advances */
    float lr_t;                       /* lr * sqrt(1 - beta2_power) / (1 - beta1_power), fp32 operations in this order */
    float reserved0;
    int64_t step;
    uint64_t stamps_off;              /* nonzero: the kernels leave no stamps (mrec_step_state_init sets 0; the host may store to this word
                                       * between steps: it is read by the kernels of the steps that follow) */
    /* [step % MREC_STAMP_RING][0] = begin of the first workgroup of the main sparse-apply kernel that ran with this state, in ticks of
     * the device wall clock (mrec_wall_clock_khz); its end: stamps_end below */
    uint64_t stamps[MREC_STAMP_RING][2];
    /* [step % MREC_STAMP_RING] = {begin, end of the step's fused lookup kernel (mrec_gather_rows_wide_ex with step_state: it runs
     * BEFORE the step's mrec_step_advance and stamps the slot of step + 1), end of the finishing kernel of the sparse apply
     * (k_apply_long), 0}: what bench.py reads the in-graph times of EmbeddingLookup + sparse apply from */
    uint64_t stamps_aux[MREC_STAMP_RING][4];
    /* [step % MREC_STAMP_RING][workgroup % 64] = end of that workgroup's last wave in the main sparse-apply kernel: the kernel's end is
     * the maximum over the 64 slots (stamps[.][1] is no longer written: one atomicMax per workgroup on it cost the step 4 us) */
    uint64_t stamps_end[MREC_STAMP_RING][64];
} mrec_step_state_t;
int mrec_step_state_init(void* state, float beta1_power, float beta2_power, int64_t step, void* stream);
/* powers *= betas, step += 1, lr_t recomputed, the stamp slot of the new step reset */
int mrec_step_advance(void* state, float lr, float beta1, float beta2, void* stream);
int mrec_wall_clock_khz(int32_t* out);

/* nn.FTRL sparse apply (FusedSparseFtrl; wide_and_deep.py:423-430). */
int mrec_sparse_ftrl_f32_i32(float* var, float* accum, float* linear, int64_t V, int64_t ld, int32_t D,
                             const int32_t* uniq, const int32_t* sorted_pos, const int32_t* sorted_seg,
                             const int32_t* seg_offsets, int64_t n, const float* g, int64_t ldg,
                             const float* row_scale, float lr, float l1, float l2, float lr_power,
                             float grad_scale, void* ws, size_t ws_bytes, void* stream);
int mrec_sparse_ftrl_f32_i64(float* var, float* accum, float* linear, int64_t V, int64_t ld, int32_t D,
                             const int64_t* uniq, const int32_t* sorted_pos, const int32_t* sorted_seg,
                             const int32_t* seg_offsets, int64_t n, const float* g, int64_t ldg,
                             const float* row_scale, float lr, float l1, float l2, float lr_power,
                             float grad_scale, void* ws, size_t ws_bytes, void* stream);

/* ---- dense optimizers (whole tensor) ------------------------------------------------------
 * nn.Adam / nn.FTRL on dense gradients (wide_and_deep.py:435-445; deep_and_cross.py:342-344). */
int mrec_dense_adam_f32(float* p, float* m, float* v, const float* g, int64_t n, float lr, float b1,
                        float b2, float eps, float b1_pow, float b2_pow, float grad_scale, int nesterov,
                        void* stream);
/* Same update for the parameters of a mixed-precision MLP: g may be bf16 (g_is_bf16 != 0: the weight
 * gradient as the bf16 GEMM wrote it; widened on load, exactly), and shadow_bf16 (nullable) receives
 * the updated parameters rounded to bf16 (round-to-nearest-even) -- the operand copy the next forward
 * reads, replacing the Cast(weight, float16) of DenseLayer.construct (wide_and_deep.py:123-124). */
int mrec_dense_adam_ex_f32(float* p, float* m, float* v, const void* g, int g_is_bf16, uint16_t* shadow_bf16,
                           int64_t n, float lr, float b1, float b2, float eps, float b1_pow, float b2_pow,
                           float grad_scale, int nesterov, void* stream);
/* Dense Adam over a flat buffer whose gradient is, for up to 16 segments, the sum of S fp32 slabs
 * slabs[q][s*lens[q] + e], s < splits[q] (what mrec_dense_bwd_weight_* leaves behind), added in slab order; every other
 * element reads g.  shadow_kind: 0 none, 1 bf16, 2 fp16 -- the 16-bit operand copy of the updated parameters
 * (Cast(weight, float16) of DenseLayer.construct, wide_and_deep.py:123-124).  slabs / starts / lens / splits are HOST arrays. */
int mrec_dense_adam_slabs_f32(float* p, float* m, float* v, const float* g, void* shadow16, int shadow_kind, int64_t n,
                              int32_t nseg, const float* const* slabs, const int64_t* starts, const int64_t* lens,
                              const int32_t* splits, float lr, float b1, float b2, float eps, float b1_pow, float b2_pow,
                              float grad_scale, int nesterov, void* step_state /* nullable mrec_step_state_t: lr_t */,
                              void* stream);
/* nn.Adam over a whole TABLE whose loss carries the L2 term l2_coef * sum(p^2) / 2 (NetWithLossClass.construct with sparse=False,
 * models/wide_deep/src/wide_and_deep.py:356-360; both tables of models/deepfm/src/deepfm.py:252-259): g holds the scattered
 * row-gradient sums only, the kernel adds l2_scaled * p (l2_scaled = l2_coef * sens; product rounded, then added) and, when
 * sumsq != NULL, leaves sum(p^2) of the values BEFORE the update in *sumsq (fp64, fixed summation order; sumsq_accumulate != 0:
 * added onto what *sumsq holds -- the second table of DeepFM).  n % 4 == 0, 16-byte aligned; ws from
 * mrec_dense_adam_l2_workspace_bytes. */
int mrec_dense_adam_l2_workspace_bytes(int64_t n, size_t* out);
int mrec_dense_adam_l2_f32(float* p, float* m, float* v, const float* g, int64_t n, float lr, float b1, float b2, float eps,
                           float b1_pow, float b2_pow, float grad_scale, int nesterov, float l2_scaled, double* sumsq,
                           int sumsq_accumulate, void* ws, size_t ws_bytes, void* stream);
/* The same over a table [V, D] (contiguous) whose gradient is nonzero on the step's touched rows only -- the bprop of a dense Gather
 * (deep_and_cross.py:199,342-344; deepfm.py:198; wide_and_deep.py:434-437 with sparse False): `sums` [U, D] are the row-gradient sums in
 * group order (mrec_segment_sum_f32), `uniq_rows` [U] the groups' table rows (n_uniq_dev: device count, nullable), every other row has
 * gradient zero.  One pass over p, m, v: no [V, D] gradient is zeroed, scattered into or read.  Element for element the arithmetic of
 * mrec_dense_adam_l2_f32 on the scattered gradient.  step_state (nullable): the step size from an mrec_step_state_t. */
int mrec_dense_adam_rows_l2_workspace_bytes(int64_t V, int32_t D, size_t* out);
int mrec_dense_adam_rows_l2_f32(float* p, float* m, float* v, int64_t V, int32_t D, const int32_t* uniq_rows, int64_t U,
                                const int64_t* n_uniq_dev, const float* sums, float lr, float b1, float b2, float eps, float b1_pow,
                                float b2_pow, float grad_scale, int nesterov, float l2_scaled, double* sumsq, int sumsq_accumulate,
                                const void* step_state, void* ws, size_t ws_bytes, void* stream);
/* Both of the above with ONE element of the buffer under FTRL instead of Adam: Wide&Deep's `wide_b` (models/wide_deep/src/
 * wide_and_deep.py:161-163) is a member of the FTRL optimizer's parameter list -- TrainStepWrap sorts by `"wide" in params.name`
 * (:407-411) and MindSpore names the Parameter held in the attribute `wide_b` "<prefix>.wide_b" [EXT: Cell.update_parameters_name
 * uses the attribute path] -- while it lives in the dense net's flat buffer here (its gradient is written by the output-head
 * kernel).  For element one_ftrl->index, m[index] is FTRL's accum and v[index] its linear (nn.FTRL, :423-430,438-445);
 * one_ftrl == NULL or index < 0: plain Adam everywhere.  fp32 gradients only. */
typedef struct mrec_ftrl1_t { int64_t index; float lr, l1, l2, lr_power; } mrec_ftrl1_t;
int mrec_dense_adam_one_ftrl_f32(float* p, float* m, float* v, const float* g, int64_t n, float lr, float b1, float b2, float eps,
                                 float b1_pow, float b2_pow, float grad_scale, int nesterov, const mrec_ftrl1_t* one_ftrl,
                                 void* stream);
int mrec_dense_adam_slabs_one_ftrl_f32(float* p, float* m, float* v, const float* g, void* shadow16, int shadow_kind, int64_t n,
                                       int32_t nseg, const float* const* slabs, const int64_t* starts, const int64_t* lens,
                                       const int32_t* splits, float lr, float b1, float b2, float eps, float b1_pow, float b2_pow,
                                       float grad_scale, int nesterov, void* step_state, const mrec_ftrl1_t* one_ftrl,
                                       void* stream);
/* The finishing pass of mrec_sparse_lazy_adam_wide -- the runs of duplicates that cross windows of the sorted index: a chain of
 * dependent round trips with almost nothing to move -- handed back instead of launched, and run as the first workgroups of the
 * dense net's Adam launch (independent work: embedding rows vs the dense flat buffers): the step loses a latency-bound launch.
 * mrec_sparse_lazy_adam_wide_defer fills *finish_out (an opaque record of launch arguments, valid until the workspace `ws`, the
 * index arrays and the tables are reused); mrec_dense_adam_slabs_finish_f32 = mrec_dense_adam_slabs_one_ftrl_f32 + that pass.
 * Results are bit-identical to the two separate launches.  Replaces the optimizer half of TrainStepWrap.construct,
 * models/wide_deep/src/wide_and_deep.py:490-492. */
typedef struct mrec_apply_finish_t { unsigned char opaque[448]; } mrec_apply_finish_t;
int mrec_sparse_lazy_adam_wide_defer(float* p, float* m, float* v, int64_t V, int64_t ld, int32_t D, const void* uniq,
                                     int32_t uniq_bytes, const int32_t* sorted_pos, const int32_t* sorted_seg,
                                     const int32_t* seg_offsets, int64_t n, const void* g, int32_t g_kind, int64_t ldg,
                                     const float* row_scale, float lr, float b1, float b2, float eps, float b1_pow, float b2_pow,
                                     float grad_scale, int nesterov, const float* gw, int64_t gw_stride, int32_t F, int32_t wide_col,
                                     float ftrl_lr, float l1, float l2, float lr_power, void* ws, size_t ws_bytes, void* step_state,
                                     const int64_t* n_valid_dev, mrec_apply_finish_t* finish_out, void* stream);
/* Hot columns.  The reference's Criteo pipeline gives each of the 13 dense features ONE id (datasets/criteo_1tb/process_data.py:138-147:
 * the feature's value travels as the weight), so a 39-field batch holds 13 fields whose id is the same in every sample: a third of the
 * positions, in 13 runs of B entries of the inverted index.  mrec_const_cols_detect finds the fields of ids[B, F] (F <= 64) in which ONE
 * id -- the most frequent of the field's first 16 samples, in at least a quarter of them -- fills at least min_count of the B samples,
 * lies in [0, V) and occurs in no other field (min_count = B: constant columns; smaller: also a field's dominant id), and leaves them
 * as a bit mask and their ids in `state`: MREC_CONST_COLS_STATE_BYTES of device memory, ZEROED ONCE by the caller and owned by these
 * calls from then on (counters, a word of failed candidates and a ticket that the launch's last workgroup clears again: one launch, no
 * memset; a batch without a candidate costs 16 F loads per workgroup).  mrec_sparse_apply_next_const_cols arms the NEXT
 * mrec_sparse_lazy_adam_wide(_defer) call of this host thread (ids: the same [B, F] batch the plan was built from, 32-bit, B <= 65536;
 * otherwise the call runs as if not armed): it launches the HOT variant of the apply kernel, whose windows skip the entries that hold a
 * field's hot id, whose first workgroups add those entries' gradient rows up sample by sample (no index: a field sits at a fixed offset
 * of every gradient row, neighbouring hot fields adjacent), and whose finishing pass (own launch, or the workgroups handed back in
 * mrec_apply_finish_t) updates the hot rows: same products, a fixed order of additions (another one than the windows': the sums agree to
 * rounding).  Not armed, the call launches the plain kernel: not an instruction of this path.  (nullptr, nullptr, 0, 0) disarms.
 * Replaces the optimizer-side Unique + UnsortedSegmentSum + LazyAdam / FTRL of those rows (wide_and_deep.py:420-430, 490-492). */
#define MREC_CONST_COLS_STATE_BYTES 2592
int mrec_const_cols_detect(const void* ids, int32_t id_bytes, int64_t B, int32_t F, int64_t V, int64_t min_count, void* state, void* stream);
int mrec_sparse_apply_next_const_cols(const void* state, const void* ids, int32_t id_bytes, int64_t B);
int mrec_dense_adam_slabs_finish_f32(float* p, float* m, float* v, const float* g, void* shadow16, int shadow_kind, int64_t n,
                                     int32_t nseg, const float* const* slabs, const int64_t* starts, const int64_t* lens,
                                     const int32_t* splits, float lr, float b1, float b2, float eps, float b1_pow, float b2_pow,
                                     float grad_scale, int nesterov, void* step_state, const mrec_ftrl1_t* one_ftrl,
                                     const mrec_apply_finish_t* finish, void* stream);
/* The same slab sums WITHOUT the optimizer, all segments in one launch: g[starts[q] + e] = sum_s slabs[q][s*lens[q] + e] in slab
 * order -- what a data-parallel rank needs before the all-reduce of the dense gradients (train_and_eval_distribute.py:135-138). */
int mrec_dense_sum_slab_segments_f32(float* g, int64_t n, int32_t nseg, const float* const* slabs, const int64_t* starts,
                                     const int64_t* lens, const int32_t* splits, void* stream);
int mrec_dense_ftrl_f32(float* var, float* accum, float* linear, const float* g, int64_t n, float lr,
                        float l1, float l2, float lr_power, float grad_scale, void* stream);

/* ---- DenseLayer on the matrix cores (hand-written MFMA GEMM, csrc/mrec_gemm.h) ------------------------
 * DenseLayer.construct, models/wide_deep/src/wide_and_deep.py:113-133 (MatMul + BiasAdd + ReLU with fp16 casts under
 * use_mixed_precision; deep_and_cross.py:94-114 is the same cell) and its bprops.  16-bit operands (`_bf16`: bfloat16,
 * `_f16`: IEEE half, the reference's dtype), fp32 accumulation, one rounding of the result.  All matrices row-major with
 * row strides in elements; x / dy / w 16-byte aligned, strides multiples of 8.
 *
 * Dropout (wide_and_deep.py:98,117-118: `x = self.dropout(x)` on the INPUT of every DenseLayer while training,
 * Dropout(p = 1 - keep_prob)): the mask is a pure function of (seed, step, layer, sample row, column) -- spec in
 * csrc/mrec_dropout.h, restated by the oracle -- so nothing is stored between forward and backward and a captured step
 * replays with a moving step.  A descriptor names the DenseLayer whose [M, W] input is dropped out:
 *   keep(r, c)  <=>  bits16(seed, step, layer, row0 + r, c; W) < round(keep_prob * 65536);  kept: x * (1 / keep_prob), else 0.
 * step_state (nullable): an mrec_step_state_t in device memory whose `step` is used instead of `step`. keep_prob in (0, 1]
 * (1: identity); layer in [0, 16); W % 4 == 0. */
typedef struct mrec_dropout {
    const void* step_state;
    uint64_t seed;
    int64_t step;
    int64_t row0;      /* global sample index of row 0 (a data-parallel rank: rank * local batch) */
    int32_t layer;
    float keep_prob;
} mrec_dropout_t;
/* y = Dropout(x) for a [M, W] matrix, kind 0: fp32, 1: bfloat16, 2: IEEE half (the product is rounded once to the kind);
 * y == x allowed.  Being its own bprop, the same call maps dy to dx.  Used for the first layer's input (the looked-up
 * rows) and by the fp32 net; the hidden layers of the 16-bit net get theirs in the GEMM epilogues below. */
int mrec_dropout(const void* x, int64_t ldx, void* y, int64_t ldy, int32_t kind, int64_t M, int32_t W,
                 const mrec_dropout_t* drop, void* stream);
/* the mask alone, as fp32 {0, 1 / keep_prob} (tests; the autograd fp32 net multiplies by it) */
int mrec_dropout_mask_f32(float* mask, int64_t ld, int64_t M, int32_t W, const mrec_dropout_t* drop, void* stream);
/*
 * forward:  y[M, N] = act(x[M, K] . w[K, N] + bias[N])   bias fp32 (nullable), relu != 0 applies max(., 0);
 *           K % 8 == 0, N % 8 == 0.  w is the weight as the reference stores it ([in, out], :100-103).
 *           drop_next (nullable): y is the input of DenseLayer drop_next->layer -- y = Dropout(round16(act(.))), W = N. */
int mrec_dense_fwd_bf16(const uint16_t* x, int64_t ldx, const uint16_t* w, const float* bias, int64_t M, int32_t K,
                        int32_t N, int relu, uint16_t* y, int64_t ldy, const mrec_dropout_t* drop_next, void* stream);
int mrec_dense_fwd_f16(const uint16_t* x, int64_t ldx, const uint16_t* w, const float* bias, int64_t M, int32_t K,
                       int32_t N, int relu, uint16_t* y, int64_t ldy, const mrec_dropout_t* drop_next, void* stream);
/* The same forward with the weight given TRANSPOSED, wt[N, K] (mrec_dense_operand_copies keeps such copies current): both
 * operands are then read along the reduction dimension, which the kernel does 15 % faster (layer 0: 72 -> 62 us).  Same products
 * in the same order: identical results. */
int mrec_dense_fwd_wt_bf16(const uint16_t* x, int64_t ldx, const uint16_t* wt, const float* bias, int64_t M, int32_t K,
                           int32_t N, int relu, uint16_t* y, int64_t ldy, const mrec_dropout_t* drop_next, void* stream);
int mrec_dense_fwd_wt_f16(const uint16_t* x, int64_t ldx, const uint16_t* wt, const float* bias, int64_t M, int32_t K,
                          int32_t N, int relu, uint16_t* y, int64_t ldy, const mrec_dropout_t* drop_next, void* stream);
/* Every derived copy of the 16-bit weights in one launch: up to 4 transposes dst[cols, rows] = src[rows, cols]^T (rows, cols
 * multiples of 8) and, when tail_packed != NULL, mrec_tail_pack_weights(tail_w2, tail_w3) (declared below). */
typedef struct mrec_transpose {
    const uint16_t* src;
    int64_t rows, cols;
    uint16_t* dst;
} mrec_transpose_t;
int mrec_dense_operand_copies(int32_t n_t, const mrec_transpose_t* t, const uint16_t* tail_w2, const uint16_t* tail_w3, int32_t K2,
                              int32_t N2, int32_t N3, uint16_t* tail_packed, void* stream);
/* bprop with respect to the layer's input, fused with the ReLU and BiasAdd bprops of the layer BELOW:
 *   dx[m, k] = h[m, k] > 0 ? sum_n dy[m, n] * w[k, n] : 0      (h nullable: no mask -- the first layer's input)
 *   db[k]    = sum_m dx[m, k]                                  (db nullable; sums of the rounded dx, fp32, fixed order)
 * N % 8 == 0, K % 4 == 0.  ws: mrec_dense_bwd_input_workspace_bytes(M, K) bytes when db != NULL.  With db == NULL and
 * ws != NULL the per-tile-row column sums are left in ws as T fp32 slabs of K (T from mrec_dense_bwd_bias_slabs);
 * mrec_dense_adam_slabs_f32 / mrec_dense_sum_slabs_f32 add them up in slab order.
 * drop_in (nullable): this layer's input went through Dropout (W = K): dx is the gradient of the value BEFORE it --
 * (sum) * (1 / keep_prob), then masked: by h > 0 when h is given (h is the dropped-out activation: its zeros are the ReLU's
 * and the mask's), by the mask function itself otherwise. */
int mrec_dense_bwd_input_workspace_bytes(int64_t M, int32_t K, size_t* out);
/* Rows T of the bias-gradient slabs [T, K] that mrec_dense_bwd_* (fused != 0) or mrec_dense_bwd_input_* (fused == 0) with
 * db == NULL leave in ws for this shape on this device: one per row tile of the configuration the library picks
 * (256 or 128 batch rows per tile). */
int mrec_dense_bwd_bias_slabs(int64_t M, int32_t K, int32_t N, int fused, int32_t* rows_out);
int mrec_dense_bwd_input_bf16(const uint16_t* dy, int64_t lddy, const uint16_t* w, const uint16_t* h, int64_t M, int32_t K,
                              int32_t N, uint16_t* dx, int64_t lddx, float* db, void* ws, size_t ws_bytes,
                              const mrec_dropout_t* drop_in, void* stream);
int mrec_dense_bwd_input_f16(const uint16_t* dy, int64_t lddy, const uint16_t* w, const uint16_t* h, int64_t M, int32_t K,
                             int32_t N, uint16_t* dx, int64_t lddx, float* db, void* ws, size_t ws_bytes,
                             const mrec_dropout_t* drop_in, void* stream);
/* bprop with respect to the weight: dw[K, N] = x[M, K]^T . dy[M, N], the batch cut in S slabs whose fp32 partial sums
 * are written to dw_slabs[S, K, N] (slab s = rows [s*c, (s+1)*c) of the batch, c = 64*ceil(ceil(M/64)/S)); the
 * optimizer adds them up (mrec_dense_adam_slabs_f32).  mrec_dense_bwd_weight_slabs proposes S for this device.
 * K % 8 == 0, N % 8 == 0 (any M). */
int mrec_dense_bwd_weight_slabs(int64_t M, int32_t K, int32_t N, int32_t* S_out);
int mrec_dense_bwd_weight_bf16(const uint16_t* x, int64_t ldx, const uint16_t* dy, int64_t lddy, int64_t M, int32_t K,
                               int32_t N, int32_t S, float* dw_slabs, void* stream);
int mrec_dense_bwd_weight_f16(const uint16_t* x, int64_t ldx, const uint16_t* dy, int64_t lddy, int64_t M, int32_t K,
                              int32_t N, int32_t S, float* dw_slabs, void* stream);

/* Both bprops of one DenseLayer in ONE launch (the two are independent and, for the narrow layers, neither fills the chip
 * alone): dx / bias-gradient slabs exactly as mrec_dense_bwd_input_* with db == NULL (db_slabs nullable: no bias gradient),
 * dw_slabs exactly as mrec_dense_bwd_weight_*.  x [M, K] is the layer's input, h (nullable) the same tensor when the layer
 * below has a ReLU to back-propagate through.  M > 0.
 * extra / n_extra (<= 2): weight-gradient problems of OTHER layers over the same batch (dw_slabs[S, K, N] = x[M, K]^T . dy[M, N]
 * exactly as mrec_dense_bwd_weight_*) whose workgroups ride this launch behind its own -- mrec_tail_fwd_bwd leaves its two
 * layers' weight gradients to be computed, and alone each is a latency-bound launch. */
typedef struct mrec_wgrad {
    const uint16_t* x; int64_t ldx;
    const uint16_t* dy; int64_t lddy;
    int32_t K, N, S, reserved;
    float* dw_slabs;
} mrec_wgrad_t;
int mrec_dense_bwd_bf16(const uint16_t* dy, int64_t lddy, const uint16_t* w, const uint16_t* h, const uint16_t* x, int64_t ldx,
                        int64_t M, int32_t K, int32_t N, uint16_t* dx, int64_t lddx, void* db_slabs, size_t db_slabs_bytes,
                        int32_t S, float* dw_slabs, const mrec_dropout_t* drop_in, const mrec_wgrad_t* extra, int32_t n_extra, void* stream);
int mrec_dense_bwd_f16(const uint16_t* dy, int64_t lddy, const uint16_t* w, const uint16_t* h, const uint16_t* x, int64_t ldx,
                       int64_t M, int32_t K, int32_t N, uint16_t* dx, int64_t lddx, void* db_slabs, size_t db_slabs_bytes,
                       int32_t S, float* dw_slabs, const mrec_dropout_t* drop_in, const mrec_wgrad_t* extra, int32_t n_extra, void* stream);
/* out[e] = sum over s < S of slabs[s*len + e], in slab order: the weight gradient as one tensor, for consumers other than
 * mrec_dense_adam_slabs_f32 (the data-parallel all-reduce).  len % 4 == 0, 16-byte aligned. */
int mrec_dense_sum_slabs_f32(const float* slabs, int32_t S, int64_t len, float* out, void* stream);

/* Output head of Wide&Deep, forward and backward in one pass over the last hidden activations h4
 * [B, K5] bf16: dense_layer_5 (K5 -> 1, fp32 weights w5[K5], b5), out = wide + deep (:315),
 * SigmoidCrossEntropyWithLogits + ReduceMean (:352-354), and their bprops seeded with dscale
 * (= sens / B):   logit[b] = h4[b,:].w5 + b5 + wide[b];  loss = mean_b BCE(logit, label);
 *   dlogit[b] = (sigmoid(logit[b]) - label[b]) * dscale;   dh4[b,k] = h4[b,k] > 0 ? dlogit[b] * w5[k] : 0;
 *   dw5[k] = sum_b h4[b,k] * dlogit[b];  db5 = sum_b dlogit[b];  db4[k] = sum_b dh4[b,k].
 * dh_scale: 1, or 1 / keep_prob when h4 went through Dropout (its zeros then carry the mask): dh4 = (dlogit * w5) * dh_scale.
 * K5 / 8 must be a power of two <= 64. */
int mrec_head_workspace_bytes(int64_t B, int32_t K5, size_t* out);
int mrec_head_fwd_bwd_bf16(const uint16_t* h4, const float* w5, const float* b5, const float* wide,
                           const float* label, int64_t B, int32_t K5, float dscale, float dh_scale, float* logit, float* dlogit,
                           uint16_t* dh4, float* dw5, float* db4, float* db5, float* loss, void* ws,
                           size_t ws_bytes, void* stream);
/* the same with IEEE half activations (the reference's mixed-precision dtype, wide_and_deep.py:119-128) */
int mrec_head_fwd_bwd_f16(const uint16_t* h4, const float* w5, const float* b5, const float* wide,
                          const float* label, int64_t B, int32_t K5, float dscale, float dh_scale, float* logit, float* dlogit,
                          uint16_t* dh4, float* dw5, float* db4, float* db5, float* loss, void* ws,
                          size_t ws_bytes, void* stream);
/* the same with fp32 activations (h4, dh4 float32 [B, K5]): the output end of the fp32 net -- DenseLayer without casts, as
 * models/deepfm with convert_dtype False or wide_deep without use_mixed_precision; the hidden layers: mrec_dense32_* */
int mrec_head_fwd_bwd_f32(const float* h4, const float* w5, const float* b5, const float* wide,
                          const float* label, int64_t B, int32_t K5, float dscale, float dh_scale, float* logit, float* dlogit,
                          float* dh4, float* dw5, float* db4, float* db5, float* loss, void* ws,
                          size_t ws_bytes, void* stream);

/* The same head (f16 != 0: IEEE half activations) with the wide branch given as the per-field products of
 * mrec_gather_rows_wide ([B, F, 2] floats: product, pad): wide[b] = (((0 + prod[b,0]) + prod[b,1]) + ...) + *wide_bias.
 * dwide_bias (nullable): receives d loss / d wide_bias = sum of dlogit (the value db5 gets), so that "Wide_b", which the
 * reference's deep optimizer owns (wide_and_deep.py:407-411), has its gradient in place without a copy. */
int mrec_head_fwd_bwd_wide(int32_t f16, const uint16_t* h4, const float* w5, const float* b5, const float* wide_prod, int32_t F,
                           const float* wide_bias, const float* label, int64_t B, int32_t K5, float dscale, float dh_scale, float* logit,
                           float* dlogit, uint16_t* dh4, float* dw5, float* db4, float* db5, float* dwide_bias, float* loss,
                           void* ws, size_t ws_bytes, void* stream);

/* ---- the tail of the dense net in one launch -------------------------------------------------
 * dense_layer_3, dense_layer_4 (512 -> 256 -> 128, wide_and_deep.py:176-199), the output head above and the input-gradient
 * bprops back through both layers, for 64 samples per workgroup with every intermediate in LDS -- five latency-bound launches
 * (mrec_dense_fwd_* x2, mrec_head_fwd_bwd_*, mrec_dense_bwd_input_* x2) as one; results identical to theirs (same products in
 * the same order; the bias-gradient partial sums come per 64 rows).  The two weight gradients stay mrec_dense_bwd_weight_*.
 *   x [B, K2] (row stride ldx) input of the first tail layer; packed: the two weights w2 [K2, N2], w3 [N2, N3] in the kernel's
 *   operand order (mrec_tail_pack_weights: mrec_tail_packed_elems 16-bit elements, to be refreshed whenever the weights
 *   change); b2, b3, w5 [N3], b5 fp32; the wide branch and label as for
 *   mrec_head_fwd_bwd_wide (wide_prod != NULL) or mrec_head_fwd_bwd_* (wide).
 * out: y2 [B, N2] (the second tail layer's input, needed by its weight gradient), dz4 [B, N3], dz3 [B, N2], dz2 [B, K2] the
 *   gradients at the outputs of the second / first tail layer and at x's layer output (masked by the ReLUs: the `dy` of the
 *   weight-gradient calls and of the next mrec_dense_bwd_*), logit, dlogit [B], dw5 [N3], db4 [N3], db5, dwide_bias (nullable),
 *   loss as the head's; db3 [N2], db2 [K2]: the column sums of dz3 / dz2 = the bias gradients of the first tail layer and of the
 *   layer below it (per-workgroup partial sums added up in workgroup order by a second small launch, like the head's).
 * drop_in (nullable): Dropout descriptor of the FIRST tail layer's input (layers +1, +2 are derived).
 * Supported (mrec_tail_supported != 0): K2 = 512, N2 = 256, N3 = 128, B % 64 == 0, B <= 32768.  ws: mrec_tail_workspace_bytes(B). */
int mrec_tail_supported(int64_t B, int32_t K2, int32_t N2, int32_t N3);
int mrec_tail_workspace_bytes(int64_t B, size_t* out);
int mrec_tail_packed_elems(int32_t K2, int32_t N2, int32_t N3, int64_t* out);
int mrec_tail_pack_weights(const uint16_t* w2, const uint16_t* w3, int32_t K2, int32_t N2, int32_t N3, uint16_t* packed, void* stream);
int mrec_tail_fwd_bwd(int32_t f16, const uint16_t* x, int64_t ldx, const uint16_t* packed, const float* b2,
                      const float* b3, const float* w5, const float* b5,
                      const float* wide, const float* wide_prod, int32_t F, const float* wide_bias, const float* label,
                      int64_t B, int32_t K2, int32_t N2, int32_t N3, float dscale, uint16_t* y2, uint16_t* dz4,
                      uint16_t* dz3, uint16_t* dz2, float* logit, float* dlogit, float* dw5, float* db4, float* db5,
                      float* dwide_bias, float* loss, float* db3, float* db2, void* ws, size_t ws_bytes,
                      const mrec_dropout_t* drop_in, void* stream);

/* ---- MapParameter key index ---------------------------------------------------------------
 * mindspore.experimental.MapParameter as built by HashEmbeddingLookup
 * (mindspore_rec/ops/embedding.py:136-146) and driven by MapTensorGet/Put/Erase
 * (embedding.py:149,193,199; README.md:160-205).  The table is split MI355X-style into a key
 * index (open addressing, this API) and dense row storage owned by the caller: get/put/apply on
 * the rows reuse the dense gather / sparse-apply kernels above with the row numbers returned here.
 *
 * The handle is a host object; its slot arrays live in caller-provided device memory
 * (`mem`, mrec_map_bytes(capacity) bytes, 256-B aligned).  Keys may be any int64 (int32 keys are
 * widened by the caller); MindRec reserves -1 and -2 (embedding.py:55-56) and so do we not: no
 * key value is reserved here.  Rows are handed out in order 0,1,2,... of first insertion
 * (deterministic: misses are numbered in the order they appear in `keys`); erased rows go to a
 * free list and are reused in erase order once fresh rows run out. */
typedef struct mrec_map mrec_map_t;
int mrec_map_bytes(int64_t capacity_rows, size_t* out);
int mrec_map_create(mrec_map_t** out, void* mem, size_t mem_bytes, int64_t capacity_rows, void* stream);
int mrec_map_destroy(mrec_map_t* h);
/* Device words: [0] = rows handed out so far (high-water mark), [1] = live keys,
 * [2] = keys dropped because the table was full (sticky error counter), [3] = rows on the free list,
 * [4] = tombstones in the slot array, [6] = slot-array rebuilds so far.  Erase leaves tombstones; once they
 * exceed a fifth of the slots the erase call rebuilds the slot array from the row side, so a probe for a
 * missing key always meets an empty slot (every probe loop is bounded by the slot count besides). */
const int64_t* mrec_map_counters_dev(const mrec_map_t* h);
/* Device arrays inside `mem`, one entry per row: the key a row holds (valid while the row is live); training-lookup hits,
 * last training step, dirty mark (see mrec_map_lookup). */
const int64_t* mrec_map_row_keys_dev(const mrec_map_t* h);
int mrec_map_tracking_dev(const mrec_map_t* h, int32_t** hits, int32_t** last_step, uint8_t** dirty);
int mrec_map_workspace_bytes(int64_t n, size_t* out);
/* keys must be unique within the call (run mrec_dedup first).  rows_out[i] = row of keys[i];
 * a missing key gets a new row when insert != 0 (is_new_out[i] = 1, caller initialises the row),
 * else rows_out[i] = -1.  n_dev (nullable, device int64): only the first min(n, *n_dev) keys are
 * processed -- pass mrec_dedup's n_uniq_dev so the Unique -> MapTensorGet chain of
 * HashEmbeddingLookup.construct (embedding.py:192-193) needs no host synchronisation. */
int mrec_map_find_or_insert(mrec_map_t* h, const int64_t* keys, int64_t n, const int64_t* n_dev, int insert,
                            int32_t* rows_out, uint8_t* is_new_out, void* ws, size_t ws_bytes, void* stream);
/* Removes keys (unique within the call); missing keys are ignored. */
int mrec_map_erase(mrec_map_t* h, const int64_t* keys, int64_t n, void* ws, size_t ws_bytes, void* stream);
/* Writes the live (key,row) pairs in row order; *n_out_dev receives the count. */
int mrec_map_export(mrec_map_t* h, int64_t* keys_out, int32_t* rows_out, int64_t* n_out_dev, void* ws,
                    size_t ws_bytes, void* stream);
/* ---- MapTensorGet as one short chain (embedding.py:149,193,199: MapTensorGet(insert_default_value=True); README.md:160-205) ----
 * mrec_map_lookup resolves n key positions to table rows in 3 launches (probe; rank + place the missing keys; default
 * rows + admission), 1 launch when not inserting:
 *   - keys need NOT be unique (MREC_MAP_UNIQUE says they are and skips the in-call dedup of the missing ones); int32 or
 *     int64 (key_bytes); n_dev as in mrec_map_find_or_insert;
 *   - a missing key gets the next row in order of first appearance in `keys` (MREC_MAP_INSERT), and the default value
 *     of every table in `tables` (values first, then the optimizer slots that share the row numbering) is written to it;
 *     without MREC_MAP_INSERT rows_out[i] = -1 (mrec_map_fill_missing then gives such positions their default value);
 *   - MREC_MAP_TRAIN marks a training lookup at step `step` (1, 2, ...): every distinct key found counts one hit and has
 *     its last-seen step stamped (the permit / evict filters of embedding.py:141-146, README.md:176-183: thresholds are
 *     counted in training steps), touched rows are marked for the next incremental export;
 *   - rows_admitted_out (nullable): rows_out with the rows of keys seen in fewer than `permit` training lookups replaced
 *     by -1 -- the row list the sparse-apply kernels take, which skip negative rows (an un-admitted key reads its
 *     default row and is not updated);
 *   - ws: mrec_map_lookup_workspace_bytes(n).  MREC_MAP_WS_PRIMED: as MREC_PLAN_WS_PRIMED -- the last writer of ws was a
 *     completed mrec_map_lookup with the same n and pointer.
 * At most 8 tables. */
typedef struct mrec_map_table {
    float* rows;      /* [capacity, D] row-major, row stride ld */
    int64_t ld;
    int32_t D;
    float sigma;      /* >= 0: default value sigma * N(0,1) keyed by (seed, key, column); < 0: the constant `fill` */
    float fill;
    uint64_t seed;
} mrec_map_table_t;
#define MREC_MAP_INSERT 1u
#define MREC_MAP_UNIQUE 2u
#define MREC_MAP_TRAIN 4u
#define MREC_MAP_WS_PRIMED 8u
#define MREC_MAP_SKIP_PAD 16u      /* key -1 (reserved by the reference: "any integers except -1, -2", embedding.py:53) is a padding slot
                                    * of a shard's request message: rows_out = -1, never inserted */
int mrec_map_lookup_workspace_bytes(int64_t n, size_t* out);
int mrec_map_lookup(mrec_map_t* h, const void* keys, int32_t key_bytes, int64_t n, const int64_t* n_dev, uint32_t flags,
                    int64_t step, int32_t permit, const mrec_map_table_t* tables, int32_t n_tables, int32_t* rows_out,
                    int32_t* rows_admitted_out, void* ws, size_t ws_bytes, void* stream);
/* out[i, :] = default value of keys[i] wherever rows[i] < 0 (insert_default_value=False: a missing key reads as its
 * default without entering the table); only D / sigma / fill / seed of `table` are used. */
/* MapTensorGet with insertion AND its output for new keys (mindspore_rec/ops/embedding.py:193; README.md:160-205): as
 * mrec_map_lookup, and the kernel that generates the default rows of new keys also writes them to out[p, :] for the position p of
 * the key's first occurrence (`out_table`: which of `tables` is the one being read; out rows 16-byte aligned, ldo % 4 == 0).
 * rows_gather[i] = the row a gather behind this call reads for position i, -1 where `out` holds the row already -- follow with
 * mrec_gather_rows_f32_skip_i32(table, V, ld, D, rows_gather, n, out, stream).  A lookup of all-new keys moves the new rows
 * once (generator -> table and output) instead of three times (generator -> table -> gather -> output). */
int mrec_map_lookup_out(mrec_map_t* h, const void* keys, int32_t key_bytes, int64_t n, const int64_t* n_dev, uint32_t flags,
                        int64_t step, int32_t permit, const mrec_map_table_t* tables, int32_t n_tables, int32_t* rows_out,
                        int32_t* rows_admitted_out, float* out, int64_t ldo, int32_t out_table, int32_t* rows_gather, void* ws,
                        size_t ws_bytes, void* stream);
int mrec_gather_rows_f32_skip_i32(const float* table, int64_t V, int64_t ld, int32_t D, const int32_t* ids, int64_t n, float* out,
                                  void* stream);
int mrec_map_fill_missing(const void* keys, int32_t key_bytes, const int32_t* rows, int64_t n, float* out, int64_t ldo,
                          const mrec_map_table_t* table, void* stream);
/* Eviction on the device: every live key whose last training lookup is more than `threshold` steps before `step` leaves
 * the table (rows to the free list in row order, keys to the erased-keys log); *n_evicted_dev receives the count.
 * ws: 4 * ceil(capacity / 2048) bytes. */
int mrec_map_evict(mrec_map_t* h, int64_t step, int64_t threshold, int64_t* n_evicted_dev, void* ws, size_t ws_bytes,
                   void* stream);
/* Incremental export (RELEASE.md:18 "MapParameter supports incremental export"): the live rows inserted, trained on or
 * put since the last call with clear != 0 as (key, row, status 1 = modified), in row order, followed by the keys erased or
 * evicted since then that are not live now as (key, -1, status 2 = erased).  Outputs sized for 2 * capacity entries;
 * ws: mrec_map_workspace_bytes(capacity) + capacity bytes. */
int mrec_map_export_dirty(mrec_map_t* h, int64_t* keys_out, int32_t* rows_out, int32_t* status_out, int64_t* n_out_dev,
                          int clear, void* ws, size_t ws_bytes, void* stream);
int mrec_map_mark_dirty(mrec_map_t* h, const int32_t* rows, int64_t n, void* stream);
/* MapTensorPut with duplicate keys in one call: table[rows[i], :] = vals[i, :] for the LAST position i of every row (what a
 * sequential upsert loop leaves, README.md:188-190).  winner: one int32 per table row, all -1 on entry and on return. */
int mrec_put_rows_last_f32(float* table, int64_t ld, int32_t D, const int32_t* rows, int64_t n, const float* vals,
                           int32_t* winner, void* stream);
/* Default-value rows for newly inserted keys: table[rows[i], :] = sigma * N01(seed, keys[i], c)
 * where is_new[i] (all i when is_new is null); sigma < 0 selects the constant `fill`. */
int mrec_init_rows_f32(float* table, int64_t ld, int32_t D, const int32_t* rows, const int64_t* keys,
                       const uint8_t* is_new, int64_t n, const int64_t* n_dev, uint64_t seed, float sigma,
                       float fill, void* stream);
/* Three device-to-device copies in ONE launch: a step's ids / weights / labels (the dataset contract of
 * models/wide_deep/src/datasets.py:212-216) into the static input buffers of the step's HIP graph.  Sizes are multiples
 * of 16 bytes, pointers 16-byte aligned. */
int mrec_copy3(void* dst0, const void* src0, int64_t bytes0, void* dst1, const void* src1, int64_t bytes1, void* dst2,
               const void* src2, int64_t bytes2, void* stream);
/* the same for up to 24 tensors */
int mrec_copy_many(int32_t n, void* const* dst, const void* const* src, const int64_t* bytes, void* stream);
/* out[i] = table[idx[i]] for int32 arrays (idx[i] < 0 gives -1): rows_of_position = rows_of_unique[inv]. */
int mrec_compose_i32(const int32_t* table, const int32_t* idx, int64_t n, int32_t* out, void* stream);
/* int32 -> int64 widening of key arrays (MapParameter key_dtype int32). */
int mrec_widen_i32_i64(const int32_t* in, int64_t n, int64_t* out, void* stream);
/* MapTensorPut: table[rows[i], :] = vals[i, :] (rows < 0 skipped). */
int mrec_scatter_rows_f32(float* table, int64_t ld, int32_t D, const int32_t* rows, int64_t n,
                          const float* vals, void* stream);
/* dst[dst_rows[i], 0:W] = src[src_rows[i], 0:W] for i < min(n, *n_dev) (n_dev: device word, NULL = n); a pair with a negative
 * row is skipped.  Either table may be pinned host memory (the device reads / writes it over PCIe): the row traffic of the
 * embedding cache (vocab_cache_size: mindspore_rec/ops/embedding.py:164-182 -- evicted rows to their host home, missing rows
 * back), driven by victim / miss lists that were compacted on the device and whose lengths the host never learns. */
int mrec_move_rows_f32(const float* src, int64_t ld_src, const int64_t* src_rows, float* dst, int64_t ld_dst,
                       const int64_t* dst_rows, int64_t n, const int64_t* n_dev, int32_t W, void* stream);

/* ---- DCN-v1 cross layers -----------------------------------------------------------------
 * CrossLayer.construct, models/deep_and_cross/src/deep_and_cross.py:139-149, all L layers of
 * DeepCrossModel.construct (:300-306) in one HBM pass: y = x0*(x_l . w_l) + b_l + x_l.
 * w, b are [L, D]; x0, out, dy, dx0 are [B, D] contiguous.  D <= 2048; the backward supports L <= 8 and
 * uses the stack's affine closed form (x_l = a_l*x0 + beta_l): it needs one dot x0 . w_l per layer and row,
 * never rebuilds x_l.  Both passes are reproducible run to run (fixed-order sums) and agree with the
 * layer-by-layer restatement to fp32 rounding: the row dots are summed lane-strided + tree, not sequentially
 * (tests: 1e-5 relative forward, 1e-4 backward). */
int mrec_cross_layers_f32(const float* x0, const float* w, const float* b, int32_t L, int64_t B, int32_t D,
                          float* out, void* stream);
int mrec_cross_layers_bwd_workspace_bytes(int32_t L, int64_t B, int32_t D, size_t* out);
int mrec_cross_layers_bwd_f32(const float* x0, const float* w, const float* b, int32_t L, int64_t B,
                              int32_t D, const float* dy, float* dx0, float* dw, float* db, void* ws,
                              size_t ws_bytes, void* stream);
/* the same with dx0 += ... : onto the gradient of x0 another branch of the net left in dx0 (DeepCrossModel.construct feeds the
 * embeddings to the deep net and to the cross stack, deep_and_cross.py:300-306: the two input gradients add) */
int mrec_cross_layers_bwd_acc_f32(const float* x0, const float* w, const float* b, int32_t L, int64_t B,
                                  int32_t D, const float* dy, float* dx0, float* dw, float* db, void* ws,
                                  size_t ws_bytes, void* stream);

/* ---- DeepFM second-order term ----------------------------------------------------------------
 * models/deepfm/src/deepfm.py:221-228 on the gathered, masked embeddings vx [B, F, D] (fp32):
 *   fm_out[b] = 0.5 * sum_d ( (sum_f vx[b,f,d])^2 - sum_f vx[b,f,d]^2 ),   colsum[b,d] = sum_f vx[b,f,d]
 * (colsum is what the backward needs).  Backward: g[b,f,d] += dout[b] * (colsum[b,d] - vx[b,f,d]).  D <= 256. */
int mrec_fm_fwd_f32(const float* vx, int64_t B, int32_t F, int32_t D, float* fm_out, float* colsum, void* stream);
int mrec_fm_bwd_f32(const float* vx, const float* colsum, const float* dout, int64_t B, int32_t F, int32_t D,
                    float* g, void* stream);
/* The same two with the model's other terms riding along (DeepFMModel.construct, deepfm.py:229-237: out = linear + fm + deep):
 * fm_out[b] = addend[b] + fm(b) (addend = the linear term: what the MLP's output head takes as its per-sample addend);
 * g_out[i] = widen(g16[i]) + dout[b] * (colsum - vx[i]): the FM gradient ADDED to the 16-bit input gradient of the
 * mixed-precision MLP (g16_kind 1 bf16, 2 f16; DenseLayer with convert_dtype, :135-145), written as the fp32 row gradient the
 * lookup's bprop hands the optimizer.  D % 4 == 0. */
int mrec_fm_fwd_add_f32(const float* vx, int64_t B, int32_t F, int32_t D, const float* addend, float* fm_out, float* colsum, void* stream);
/* ... and a 16-bit copy of vx (kind16 1: bf16, 2: f16; x16 [B, F, D], D % 4 == 0) written in the same pass: the input of DeepFM's fp16
 * DenseLayers (deepfm.py:135-137 casts the very tensor the FM term reads). */
int mrec_fm_fwd_add16_f32(const float* vx, int64_t B, int32_t F, int32_t D, const float* addend, float* fm_out, float* colsum, void* x16,
                          int32_t kind16, void* stream);
int mrec_fm_bwd_mix_f32(const float* vx, const float* colsum, const float* dout, const void* g16, int32_t g16_kind, int64_t B, int32_t F,
                        int32_t D, float* g_out, void* stream);
/* table[rows[i], :] += vals[i, :] for distinct rows (rows < 0 skipped): adds a segment-sum into a dense
 * [V, D] gradient that already holds the L2 term (deepfm.py:252-259). */
int mrec_scatter_add_rows_f32(float* table, int64_t ld, int32_t D, const int32_t* rows, int64_t n,
                              const float* vals, void* stream);

/* ---- DenseLayer in fp32 (convert_dtype=False: Deep&Cross, models/deep_and_cross/src/deep_and_cross.py:94-114,293-309) --------
 * MatMul + BiasAdd + ReLU and the two MatMul bprops on v_mfma_f32_32x32x2_f32: exact fp32 (a k-ordered chain of fmaf's per
 * output element), 157 TFLOP/s peak.  Row strides in floats; rows need 4-byte alignment only (16 / 8-byte aligned rows load
 * wider).
 *   mrec_dense32_fwd:        y = relu?(x . w + bias): x [M, K], w [K, N], bias [N] (nullable), y [M, N]
 *   mrec_dense32_bwd_input:  dx = (dy . w^T) masked by h > 0 (h [M, K]: the activation of the layer below, nullable); colsum_ws
 *                            (nullable): [ceil(M / 64), K] column sums of dx per 64 rows = partial sums of the layer
 *                            below's bias gradient, to be added up in tile order (mrec_dense_adam_slabs_f32 does)
 *   mrec_dense32_bwd_weight: dw_slabs[s] = x[rows of slab s]^T . dy[rows of slab s], fp32 [S, K, N]: the batch is cut into S
 *                            slabs of ceil(M / S / 32) * 32 rows (mrec_dense32_bwd_weight_slabs proposes an S that fills the chip) */
int mrec_dense32_fwd(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, int64_t M, int32_t K, int32_t N,
                     int relu, float* y, int64_t ldy, void* stream);
int mrec_dense32_bwd_input(const float* dy, int64_t lddy, const float* w, int64_t ldw, const float* h, int64_t ldh, int64_t M, int32_t K,
                           int32_t N, float* dx, int64_t lddx, float* colsum_ws, void* stream);
int mrec_dense32_bwd_weight(const float* x, int64_t ldx, const float* dy, int64_t lddy, int64_t M, int32_t K, int32_t N, int32_t S,
                            float* dw_slabs, void* stream);
int mrec_dense32_bwd_weight_slabs(int64_t M, int32_t K, int32_t N, int32_t* out);

/* The output end of Deep&Cross in one pass (deep_and_cross.py:306-309,326-331): logit = [d2 | c] . w3 + b3 (the concat is never
 * materialised), loss = mean sigmoid cross-entropy, dlogit = (sigmoid(logit) - label) * dscale, and the bprops that hang off it:
 * dd2 = dlogit * w3[:H] where d2 > 0 (the gradient at dense_layer_2's pre-activation), dc = dlogit * w3[H:], dw3 = [d2 | c]^T .
 * dlogit, db3 = sum dlogit, db2 = column sums of dd2.  d2 [B, H] / c [B, X] fp32 with row strides ldd / ldc; H % 4 == 0,
 * H <= 1024, X % 2 == 0, X <= 1280; batch sums are added in a fixed order (reproducible).  ws: mrec_dcn_head_workspace_bytes. */
int mrec_dcn_head_workspace_bytes(int64_t B, int32_t H, int32_t X, size_t* out);
int mrec_dcn_head_fwd_bwd(const float* d2, int64_t ldd, const float* c, int64_t ldc, const float* w3, const float* b3,
                          const float* label, int64_t B, int32_t H, int32_t X, float dscale, float* logit_out, float* dd2, int64_t lddd,
                          float* dc, int64_t lddc, float* dw3_out, float* db2_out, float* db3_out, float* loss_out, void* ws,
                          size_t ws_bytes, void* stream);

/* ---- row-shard routing (hybrid-parallel embedding, README.md:140-144) --------
 * owner(id) = id mod n_shards, local row = id div n_shards.  Stable bucketing of ids by owner so
 * one RCCL all-to-all can ship them: send_local[k] = local row of the k-th id in bucket order,
 * send_perm[k] = its original position, counts_dev[s] = bucket sizes (int64). */
int mrec_shard_route_workspace_bytes(int64_t n, int32_t n_shards, size_t* out);
int mrec_shard_route_i32(const int32_t* ids, int64_t n, int32_t n_shards, int32_t* send_local,
                         int32_t* send_perm, int64_t* counts_dev, void* ws, size_t ws_bytes, void* stream);
int mrec_shard_route_i64(const int64_t* ids, int64_t n, int32_t n_shards, int64_t* send_local,
                         int32_t* send_perm, int64_t* counts_dev, void* ws, size_t ws_bytes, void* stream);
/* Same for hash tables keyed by the raw id (MapParameter under row sharding, BASELINE configs[4]): owner =
 * (mix64(key) >> 33) mod n_shards and send_keys carries the raw keys, which the owner translates to rows of its own
 * key index (mrec_map_lookup). */
int mrec_shard_route_hash_i32(const int32_t* keys, int64_t n, int32_t n_shards, int32_t* send_keys, int32_t* send_perm,
                              int64_t* counts_dev, void* ws, size_t ws_bytes, void* stream);
int mrec_shard_route_hash_i64(const int64_t* keys, int64_t n, int32_t n_shards, int64_t* send_keys, int32_t* send_perm,
                              int64_t* counts_dev, void* ws, size_t ws_bytes, void* stream);
/* out[send_perm[k], :] = rows[k, :] * (row_scale ? row_scale[send_perm[k]] : 1): undoes the
 * bucketing on the returned embedding rows. */
int mrec_shard_unroute_f32(const float* rows, const int32_t* send_perm, int64_t n, int32_t D,
                           const float* row_scale, float* out, void* stream);
/* rows_out[k, :] = g[send_perm[k], :] * (row_scale ? row_scale[send_perm[k]] : 1): buckets the
 * row gradients for the backward all-to-all. */
int mrec_shard_route_rows_f32(const float* g, int64_t ldg, const int32_t* send_perm, int64_t n, int32_t D,
                              const float* row_scale, float* rows_out, void* stream);
/* The same two permutations with explicit row strides on both sides (rows of a message that carries more than one thing),
 * and the (id, weight) pairs of a request message: out_pairs[k] = {send_local[k], bits(wts[send_perm[k]])}. */
int mrec_shard_unroute_ld_f32(const float* rows, int64_t ldr, const int32_t* send_perm, int64_t n, int32_t D,
                              const float* row_scale, float* out, int64_t ldo, void* stream);
int mrec_shard_route_rows_ld_f32(const float* g, int64_t ldg, const int32_t* send_perm, int64_t n, int32_t D,
                                 const float* row_scale, float* rows_out, int64_t ldo, void* stream);
int mrec_shard_pack_iw_i32(const int32_t* send_local, const float* wts, const int32_t* send_perm, int64_t n, int32_t* out_pairs,
                           void* stream);
int mrec_shard_unpack_iw_i32(const int32_t* pairs, int64_t n, int32_t* ids_out, float* wts_out, void* stream);

/* ---- fixed-capacity routing: a sharded step whose every message has a static shape ------------------------------------
 * (hybrid parallel, README.md:140-144; models/wide_deep/train_and_eval_distribute.py:135-138; the reference's own mechanism,
 * replicated ids + masked local Gather + AllReduce of [N, D] partials, wide_and_deep.py:232-249, has static shapes too.)
 * Every rank hands every owner exactly `cap` request slots, so the three all-to-alls of a step (requests, answers, row
 * gradients) have equal, host-known splits: no bucket sizes cross the host, and the step can be captured as one HIP graph.
 *   mrec_shard_route_slots_*: position i (owner o = id mod n_shards, or (mix64(key) >> 33) mod n_shards when `hashed`) takes
 *     slot o * cap + j, j = its rank among the positions of owner o in ascending position order (stable: the owner's
 *     summation order is a function of the batch alone).  req[slot] = {id' , wts[i]} with id' = id div n_shards (dense
 *     tables) or the raw key (hashed); entries are 8 bytes {int32, float} for int32 ids, 16 bytes {int64, float, 0} for int64.
 *     Unused slots carry id' = -1, weight 0.  slot_of_pos[i] = the slot (or -1: the bucket was full), pos_of_slot[slot] = i
 *     (or -1).  *overflow_dev += the number of positions that found their bucket full (sticky; the caller checks it once per
 *     sink and raises: the step that dropped positions is not a valid step).  wts nullable (all 1).
 *     chunk_rot (0 .. n_shards - 1): owner o's slots are chunk (o - chunk_rot) mod n_shards of the message (0: chunk = owner).
 *     A rank that passes its own rank + 1 finds its OWN chunk last; with the receive buffer laid out so that the sender's
 *     chunk comes first (chunk (s - rank) mod n for sender s), send buffer = X[0 : n] and receive buffer = X[n - 1 : 2n - 1] of
 *     one allocation share exactly the rank's own chunk, which then needs no copy at all (mindrec_amd/wide_deep_shard.py).
 *   mrec_shard_unpack_req: the received entries as two plain arrays (the plan and the apply's row_scale want them so).
 *   mrec_shard_unroute_slots: back at the requester, position i reads row slot_of_pos[i] of the returned message
 *     ([Dw words of the looked-up row | wide product, 0 | pad], W words per row) and writes emb_out[i, 0:Dw] and
 *     wprod_out[2 i] = (product, 0): what mrec_gather_rows_wide hands out on one GPU.  Dw = D / 2 (16-bit rows) or D (fp32).
 *   mrec_shard_route_grads: the gradient message, msg[slot] = [Dw words of g[pos_of_slot[slot]] | dlogit[pos / F] | pad];
 *     padding slots are left alone. */
int mrec_shard_route_slots_workspace_bytes(int64_t n, int32_t n_shards, size_t* out);
int mrec_shard_route_slots_i32(const int32_t* ids, const float* wts, int64_t n, int32_t n_shards, int64_t cap, int hashed,
                               int32_t chunk_rot, void* req, int32_t* slot_of_pos, int32_t* pos_of_slot, int64_t* overflow_dev, void* ws,
                               size_t ws_bytes, void* stream);
int mrec_shard_route_slots_i64(const int64_t* ids, const float* wts, int64_t n, int32_t n_shards, int64_t cap, int hashed,
                               int32_t chunk_rot, void* req, int32_t* slot_of_pos, int32_t* pos_of_slot, int64_t* overflow_dev, void* ws,
                               size_t ws_bytes, void* stream);
/* mrec_shard_route_slots_* over a list whose LENGTH lives on the device (*n_valid_dev <= n; entries past it are nobody's: no slot,
 * slot_of_pos = -1, no overflow): a step's UNIQUE ids straight out of mrec_dedup_* (uniq, n_uniq_dev) -- the reference dedups in
 * front of the sharded lookup too (Unique().shard(((1,),)), models/wide_deep/src/wide_and_deep.py:212; embedding.py:189-195).
 * An owner then answers one row per unique id and receives one summed gradient row per unique id (mindrec_amd/wide_deep_shard.py). */
int mrec_shard_route_slots_nv_i32(const int32_t* ids, const float* wts, int64_t n, const int64_t* n_valid_dev, int32_t n_shards,
                                  int64_t cap, int hashed, int32_t chunk_rot, void* req, int32_t* slot_of_pos, int32_t* pos_of_slot,
                                  int64_t* overflow_dev, void* ws, size_t ws_bytes, void* stream);
int mrec_shard_route_slots_nv_i64(const int64_t* ids, const float* wts, int64_t n, const int64_t* n_valid_dev, int32_t n_shards,
                                  int64_t cap, int hashed, int32_t chunk_rot, void* req, int32_t* slot_of_pos, int32_t* pos_of_slot,
                                  int64_t* overflow_dev, void* ws, size_t ws_bytes, void* stream);
int mrec_shard_unpack_req(const void* req, int32_t id_bytes, int64_t n_slots, void* ids_out, float* wts_out, void* stream);
int mrec_shard_unroute_slots(const float* back, int64_t W, const int32_t* slot_of_pos, int64_t n, int32_t Dw, float* emb_out,
                             float* wprod_out, void* stream);
int mrec_shard_route_grads(const float* g, int64_t ldg, const float* dlogit, int32_t F, const int32_t* pos_of_slot, int64_t n_slots,
                           int32_t Dw, float* msg, int64_t W, void* stream);

/* ---- measurement hooks (used by bench.py; no effect on results) --------------------------
 * HIP events owned by the library, and a one-shot hook: the NEXT sparse-apply call (segment_sum /
 * lazy_adam / ftrl) on this host thread records `start` immediately before its main kernel
 * (k_apply_main) and `stop` immediately after it, on the stream the kernel is launched on, so the
 * pair times exactly the kernel rocprofv3 lists under that name. */
int mrec_event_create(void** ev_out);
int mrec_event_destroy(void* ev);
int mrec_event_elapsed_ms(void* start, void* stop, float* ms_out); /* waits for `stop` */
int mrec_profile_next_apply(void* start, void* stop);

/* ---- fp32 DenseLayers at the 16-bit matrix rate: three-part bf16 operands (csrc/mrec_gemm_x3.hip) ---------------------------------
 * DenseLayer.construct with convert_dtype=False (models/deep_and_cross/src/deep_and_cross.py:94-114; the reference's benchmark
 * net, benchmarks/wide_deep/default_config.yaml:16) and its bprops.  An fp32 operand x is held as x1 + x2 + x3, bf16 each
 * (x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2): 24 mantissa bits); a . b is the six bf16 products a1 b1, a1 b2, a2 b1,
 * a1 b3, a2 b2, a3 b1 accumulated in fp32 by the MFMA, smallest first -- an fp32-class result (<= ~2^-22 of sum |a b|) in 6/16 of
 * the time of the fp32-input matrix instruction.  mrec_dense32_* (exact fp32 products, csrc/mrec_gemm_f32.hip) stay as they are.
 *   parts image of an fp32 [R, C] tensor: bf16 [3][Rp][Cp], Rp / Cp = R / C rounded up to 64, zero padded
 *   (mrec_x3_parts_elems gives 3 * Rp * Cp); one image serves every GEMM the tensor is an operand of.
 *   mrec_x3_gemm form 0: C[M, N] = P[M, K] . Q[K, N]         P = parts of x [M, K],  Q = parts of w [K, N]
 *                form 1: C[M, K] = P[M, N] . Q[K, N]^T       P = parts of dy [M, N], Q = parts of w [K, N]
 *                form 2: C[S][K, N] = P[M, K]^T . Q[M, N]    P = parts of x [M, K],  Q = parts of dy [M, N]; S batch slabs of
 *                        fp32 partial sums (slab s at C + s * K * ldc), added up by mrec_dense_adam_slabs_* / mrec_dense_sum_slabs_f32
 *   C is fp32 with row stride ldc (even; 16-byte stores when ldc % 4 == 0).
 *   mrec_x3_bias_relu: y = relu?(acc + bias) in place (BiasAdd + ReLU) and, parts_out != NULL, y's parts image in the same pass.
 *   mrec_x3_mask_colsum: dx = (h > 0 ? acc : 0) * scale in place (scale: 1 / keep of a Dropout on the layer's input, else 1) (the ReLU bprop of the layer below; h nullable), colsum [ceil(M / 64), K]
 *   (nullable) = column sums of dx per 64 rows (that layer's BiasAdd bprop), and dx's parts image. */
int mrec_x3_parts_elems(int64_t rows, int64_t cols, int64_t* out);
int mrec_x3_split(const float* x, int64_t ldx, int64_t R, int32_t C, uint16_t* parts, void* stream);
int mrec_x3_gemm(int form, const uint16_t* Pparts, const uint16_t* Qparts, int64_t M, int32_t K, int32_t N, float* C, int64_t ldc,
                 int32_t S, void* stream);
int mrec_x3_bias_relu(float* acc, int64_t ld, int64_t M, int32_t N, const float* bias, int relu, uint16_t* parts_out, void* stream);
/* the slab count mrec_x3_gemm's form 2 runs fastest with (a 256 x 256 tile x slab per CU); any S <= 6 * ceil(M / 64) whose slabs are
 * all non-empty is accepted */
int mrec_x3_wgrad_slabs(int64_t M, int32_t K, int32_t N, int32_t* out);
int mrec_x3_mask_colsum(float* acc, int64_t ld, int64_t M, int32_t K, const float* h, int64_t ldh, float scale, float* colsum,
                        uint16_t* parts_out, void* stream);
/* the same output ends inside the GEMM's epilogue (no second pass over the layer's output):
 *   mrec_x3_gemm_fwd:   y = dropout?(relu?(x . w + bias)) [M, N] AND (parts_out != NULL) y's parts image; drop_next (nullable): the
 *                       Dropout on the NEXT DenseLayer's input (wide_and_deep.py:117-118), as mrec_dropout would apply it to y
 *   mrec_x3_gemm_dgrad: dx = (h > 0 ? dy . w^T : 0) * scale [M, K], colsum [ceil(M / 64), K] (nullable) AND (parts_out != NULL) dx's parts
 * ws (nullable; mrec_x3_gemm_dgrad_workspace_bytes, 0 for most shapes): lets the PLAIN input gradient (no h, colsum, parts; scale 1) of a
 * width with a narrow last 256-column tile run that tile as slabs of the reduction behind one round of full tiles.
 * parts_out: the padding of the image (rows M.., columns past the width) is NOT written -- the caller zeroes the image once. */
int mrec_x3_gemm_fwd(const uint16_t* xparts, const uint16_t* wparts, int64_t M, int32_t K, int32_t N, float* y, int64_t ldy,
                     const float* bias, int relu, const mrec_dropout_t* drop_next, uint16_t* parts_out, void* stream);
int mrec_x3_gemm_dgrad_workspace_bytes(int64_t M, int32_t K, int32_t N, size_t* out);
int mrec_x3_gemm_dgrad(const uint16_t* dyparts, const uint16_t* wparts, int64_t M, int32_t K, int32_t N, float* dx, int64_t lddx,
                       const float* h, int64_t ldh, float scale, float* colsum, uint16_t* parts_out, void* ws, size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MREC_H_ */
