// mrec_route.hip -- row-shard routing for the hybrid-parallel embedding (one process per GPU,
// RCCL all-to-all over xGMI), for gfx950.
//
// Reference: MindRec's hybrid mode row-slices the table across devices (README.md:140-144;
// nn.EmbeddingLookup(..., slice_mode=TABLE_ROW_SLICE) at models/wide_deep/src/wide_and_deep.py:232-249)
// and lets MindSpore auto-parallel all-reduce [N, D] partials [EXT].  xGMI is point-to-point, so
// here ids are bucketed by owner (id mod n_shards), exchanged with one all-to-all, gathered
// locally and returned with a second all-to-all.  This file is the bucketing:
// a one-pass stable radix split of positions by owner, reusing mrec_radix.h.
#include "mrec_common.h"
#include "mrec_radix.h"

namespace {

template <class K>
__global__ __launch_bounds__(256) void k_owner(const K* __restrict__ ids, int64_t n, int S, int* __restrict__ owner) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    K o = ids[i] % (K)S;
    if (o < 0) o += (K)S;
    owner[i] = (int)o;
}

template <class K>
__global__ __launch_bounds__(256) void k_route_finish(const K* __restrict__ ids, int64_t n, int S,
                                                      const int* __restrict__ perm, const int* __restrict__ dbase,
                                                      K* __restrict__ send_local, int64_t* __restrict__ counts) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k < S) counts[k] = (int64_t)((k + 1 < S) ? dbase[k + 1] : (int)n) - dbase[k];
    if (k >= n) return;
    const K id = ids[perm[k]];
    K o = id % (K)S;
    if (o < 0) o += (K)S;
    send_local[k] = (id - o) / (K)S;
}

__global__ void k_zero_counts(int64_t* counts, int S) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k < S) counts[k] = 0;
}

// One wave per row: dst[dst_row(k), :] = src[src_row(k), :] * scale.
template <bool SCATTER>
__global__ __launch_bounds__(256) void k_perm_rows(const float* __restrict__ src, int64_t lds,
                                                   const int* __restrict__ perm, int64_t n, int D,
                                                   const float* __restrict__ row_scale, float* __restrict__ dst) {
    const int lane = threadIdx.x & 63;
    const int64_t k = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= n) return;
    const int64_t p = perm[k];
    const float s = row_scale ? row_scale[p] : 1.0f;
    const float* a = SCATTER ? src + k * lds : src + p * lds;
    float* o = SCATTER ? dst + p * (int64_t)D : dst + k * (int64_t)D;
    for (int c = lane; c < D; c += 64) o[c] = row_scale ? a[c] * s : a[c];
}

template <class K>
int route_impl(const K* ids, int64_t n, int32_t S, K* send_local, int32_t* send_perm, int64_t* counts_dev, void* ws,
               size_t ws_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (n < 0 || S <= 0 || S > RNB || !counts_dev) return MREC_EINVAL;
    if (n == 0) {
        k_zero_counts<<<(unsigned)mrec_cdiv(S, 256), 256, 0, st>>>(counts_dev, S);
        MREC_LAUNCH_CHECK();
        return MREC_OK;
    }
    if (!ids || !send_local || !send_perm || !ws) return MREC_EINVAL;
    if (n > (int64_t(1) << 30)) return MREC_EUNSUPPORTED;
    const int nblk = (int)mrec_cdiv(n, RT);
    MrecArena a(ws, ws_bytes);
    int* hist = a.take<int>((size_t)nblk * RNB);
    int* hscan = a.take<int>((size_t)nblk * RNB);
    int* totals = a.take<int>(RNB);
    int* dbase = a.take<int>(RNB);
    int* owner = a.take<int>(n);
    int* okeys = a.take<int>(n);
    if (!a.ok) return MREC_EWORKSPACE;
    int nbits = 1;
    while ((1 << nbits) < S) ++nbits;
    k_owner<K><<<(unsigned)mrec_cdiv(n, 256), 256, 0, st>>>(ids, n, S, owner);
    radix_pass(owner, nullptr, (int)n, 0, nbits, hist, hscan, totals, dbase, okeys, send_perm, st);
    const int64_t m = n > S ? n : S;
    k_route_finish<K><<<(unsigned)mrec_cdiv(m, 256), 256, 0, st>>>(ids, n, S, send_perm, dbase, send_local, counts_dev);
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}

}  // namespace

MREC_API int mrec_shard_route_workspace_bytes(int64_t n, int32_t n_shards, size_t* out) {
    if (!out || n < 0 || n_shards <= 0) return MREC_EINVAL;
    const size_t nn = (size_t)(n ? n : 1);
    *out = mrec_align_up((size_t)mrec_cdiv(nn, RT) * RNB * 4, 256) * 2 + mrec_align_up((size_t)RNB * 4, 256) * 2 +
           mrec_align_up(nn * 4, 256) * 2;
    return MREC_OK;
}

MREC_API int mrec_shard_route_i32(const int32_t* ids, int64_t n, int32_t n_shards, int32_t* send_local,
                                  int32_t* send_perm, int64_t* counts_dev, void* ws, size_t ws_bytes, void* stream) {
    return route_impl<int32_t>(ids, n, n_shards, send_local, send_perm, counts_dev, ws, ws_bytes, stream);
}

MREC_API int mrec_shard_route_i64(const int64_t* ids, int64_t n, int32_t n_shards, int64_t* send_local,
                                  int32_t* send_perm, int64_t* counts_dev, void* ws, size_t ws_bytes, void* stream) {
    return route_impl<int64_t>(ids, n, n_shards, send_local, send_perm, counts_dev, ws, ws_bytes, stream);
}

MREC_API int mrec_shard_unroute_f32(const float* rows, const int32_t* send_perm, int64_t n, int32_t D,
                                    const float* row_scale, float* out, void* stream) {
    if (n < 0 || D <= 0) return MREC_EINVAL;
    if (n == 0) return MREC_OK;
    if (!rows || !send_perm || !out) return MREC_EINVAL;
    k_perm_rows<true><<<(unsigned)mrec_cdiv(n, 4), 256, 0, (hipStream_t)stream>>>(rows, D, send_perm, n, D, row_scale, out);
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}

MREC_API int mrec_shard_route_rows_f32(const float* g, int64_t ldg, const int32_t* send_perm, int64_t n, int32_t D,
                                       const float* row_scale, float* rows_out, void* stream) {
    if (n < 0 || D <= 0 || ldg < D) return MREC_EINVAL;
    if (n == 0) return MREC_OK;
    if (!g || !send_perm || !rows_out) return MREC_EINVAL;
    k_perm_rows<false><<<(unsigned)mrec_cdiv(n, 4), 256, 0, (hipStream_t)stream>>>(g, ldg, send_perm, n, D, row_scale,
                                                                                 rows_out);
    MREC_LAUNCH_CHECK();
    return MREC_OK;
}
