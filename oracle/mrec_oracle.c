/*
 * mrec_oracle.c -- CPU restatement of the MindRec hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity checker for the HIP kernels in mindrec_amd/csrc/.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product path never
 * does (it fails loudly when libmrec_hip.so is missing).
 *
 * PARITY UNPINNED at the MindSpore boundary: the reference (/root/reference) delegates every
 * primitive below to the un-vendored, un-pinned PyPI dependency `mindspore`
 * (requirements/cpu_requirements.txt:3), which is not installed here, and the reference's own
 * tests hold no golden vectors for this path (tests/ut/test_example.py:6,
 * tests/st/test_example.py:6 are empty).  Each function restates the *published* semantics of
 * the MindSpore primitive that the cited reference call site invokes and
 * is pinned by hand-computed known-answer tests in tests/test_oracle.py.
 *
 * Plain C99, scalar, single-threaded unless a *_mt entry is used.  Build with
 *   gcc -O2 -ffp-contract=off -fPIC -shared   (see oracle/Makefile)
 * -ffp-contract=off + explicit fmaf() keeps the float arithmetic bit-reproducible against the
 * HIP build, which is compiled with the same flag.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MREC_O_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------
 * Counter-based N(0,1) generator shared (bit-for-bit) with mindrec_amd/csrc/mrec_rng.h.
 * Stands in for MindSpore's initializer('normal') [EXT: N(0, 0.01)], reference use:
 * models/wide_deep/default_config.yaml:41 (emb_init: 'normal'), mindspore_rec/ops/embedding.py:88
 * (param_init="normal" -> MapParameter default_value).  The distribution matches; the stream is
 * ours (keyed by seed,row,col) so CPU and GPU agree without ever staging a table on the host.
 * ---------------------------------------------------------------------------------------- */
static inline uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* ln(u) for u in (0,1], cephes-style polynomial, every multiply-add an explicit fmaf. */
static inline float det_logf(float x) {
    /* x is a normal float in [2^-24, 1]: split as m * 2^e, m in [0.5,1), by bit surgery */
    uint32_t xb; memcpy(&xb, &x, 4);
    int e = (int)((xb >> 23) & 0xFFu) - 126;
    uint32_t mb = (xb & 0x007FFFFFu) | 0x3F000000u;
    float m; memcpy(&m, &mb, 4);
    if (m < 0.70710678118654752440f) {
        e -= 1;
        m = m + m - 1.0f;
    } else {
        m = m - 1.0f;
    }
    float z = m * m;
    float y = 7.0376836292E-2f;
    y = fmaf(y, m, -1.1514610310E-1f);
    y = fmaf(y, m, 1.1676998740E-1f);
    y = fmaf(y, m, -1.2420140846E-1f);
    y = fmaf(y, m, 1.4249322787E-1f);
    y = fmaf(y, m, -1.6668057665E-1f);
    y = fmaf(y, m, 2.0000714765E-1f);
    y = fmaf(y, m, -2.4999993993E-1f);
    y = fmaf(y, m, 3.3333331174E-1f);
    y = y * m;
    y = y * z;
    float fe = (float)e;
    y = fmaf(-2.12194440e-4f, fe, y);
    y = fmaf(-0.5f, z, y);
    float r = m + y;
    r = fmaf(0.693359375f, fe, r);
    return r;
}

/* (cos, sin)(2*pi*k/2^24) for integer k in [0,2^24): integer quadrant reduction + fmaf polynomials on [0, pi/4]. */
static inline void det_sincos2pi_u24(uint32_t k, float* cos_out, float* sin_out) {
    uint32_t q = k >> 22;            /* quadrant */
    uint32_t r = k & 0x3FFFFFu;      /* phi = (pi/2) * r / 2^22 */
    int swap = 0;
    if (r > 0x200000u) { r = 0x400000u - r; swap = 1; } /* phi' = pi/2 - phi in [0,pi/4] */
    float t = (float)r * 3.7450703e-07f; /* (pi/2)/2^22 */
    float z = t * t;
    float s = -1.9515295891E-4f;
    s = fmaf(s, z, 8.3321608736E-3f);
    s = fmaf(s, z, -1.6666654611E-1f);
    s = s * z;
    s = fmaf(s, t, t);
    float c = 2.443315711809948E-005f;
    c = fmaf(c, z, -1.388731625493765E-003f);
    c = fmaf(c, z, 4.166664568298827E-002f);
    c = c * z;
    c = c * z;
    c = fmaf(-0.5f, z, c);
    c = c + 1.0f;
    float cs = swap ? s : c; /* cos(phi) */
    float sn = swap ? c : s; /* sin(phi) */
    switch (q) {             /* theta = q * pi/2 + phi */
        case 0: *cos_out = cs; *sin_out = sn; break;
        case 1: *cos_out = -sn; *sin_out = cs; break;
        case 2: *cos_out = -cs; *sin_out = -sn; break;
        default: *cos_out = sn; *sin_out = -cs; break;
    }
}

/* Columns come in pairs: both outputs of one Box-Muller transform keyed by (row, col >> 1). */
static inline float det_normal(uint64_t seed, int64_t row, int32_t col) {
    uint64_t h = mix64(seed ^ mix64((uint64_t)row * 0xD1342543DE82EF95ull + (uint64_t)(uint32_t)(col >> 1)));
    uint32_t a = (uint32_t)(h >> 40);            /* 24 bits */
    uint32_t b = (uint32_t)(h >> 8) & 0xFFFFFFu; /* 24 bits */
    float u1 = ((float)a + 1.0f) * 5.9604644775390625e-08f; /* (0,1] */
    float rad = sqrtf(-2.0f * det_logf(u1));
    float c, s;
    det_sincos2pi_u24(b, &c, &s);
    return rad * ((col & 1) ? s : c);
}

MREC_O_API void mrec_o_normal_rows_f32(uint64_t seed, const int64_t* rows, int64_t n, int32_t D,
                                       float sigma, float* out) {
    for (int64_t i = 0; i < n; ++i)
        for (int32_t c = 0; c < D; ++c) out[i * D + c] = sigma * det_normal(seed, rows[i], c);
}

MREC_O_API void mrec_o_fill_normal_f32(uint64_t seed, int64_t row0, int64_t nrows, int32_t D,
                                       int64_t ld, float sigma, float* out) {
    for (int64_t r = 0; r < nrows; ++r)
        for (int32_t c = 0; c < D; ++c) out[r * ld + c] = sigma * det_normal(seed, row0 + r, c);
}

/* ------------------------------------------------------------------------------------------
 * ops.Unique  -- reference call sites mindspore_rec/ops/embedding.py:153,192 and
 * models/wide_deep/src/wide_and_deep.py:212.  CPU kernel semantics [EXT, SURVEY A.1]:
 * y keeps first-occurrence order, y[idx[i]] == x[i].
 * ---------------------------------------------------------------------------------------- */
typedef struct { int64_t key; int32_t val; int32_t used; } oslot_t;

static inline uint64_t ohash(int64_t k) { return mix64((uint64_t)k); }

static int64_t unique_i64_impl(const int64_t* x, int64_t n, int64_t* uniq, int32_t* inv) {
    uint64_t cap = 16;
    while (cap < (uint64_t)n * 2) cap <<= 1;
    oslot_t* tab = (oslot_t*)calloc(cap, sizeof(oslot_t));
    int64_t U = 0;
    for (int64_t i = 0; i < n; ++i) {
        uint64_t s = ohash(x[i]) & (cap - 1);
        while (tab[s].used && tab[s].key != x[i]) s = (s + 1) & (cap - 1);
        if (!tab[s].used) {
            tab[s].used = 1; tab[s].key = x[i]; tab[s].val = (int32_t)U;
            uniq[U++] = x[i];
        }
        inv[i] = tab[s].val;
    }
    free(tab);
    return U;
}

MREC_O_API int64_t mrec_o_unique_i64(const int64_t* x, int64_t n, int64_t* uniq, int32_t* inv) {
    return unique_i64_impl(x, n, uniq, inv);
}

MREC_O_API int64_t mrec_o_unique_i32(const int32_t* x, int64_t n, int32_t* uniq, int32_t* inv) {
    int64_t* xl = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n ? n : 1));
    int64_t* ul = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n ? n : 1));
    for (int64_t i = 0; i < n; ++i) xl[i] = x[i];
    int64_t U = unique_i64_impl(xl, n, ul, inv);
    for (int64_t i = 0; i < U; ++i) uniq[i] = (int32_t)ul[i];
    free(xl); free(ul);
    return U;
}

/* ------------------------------------------------------------------------------------------
 * ops.Gather(params, idx, 0) / SparseGatherV2 / EmbeddingLookup [EXT, SURVEY A.2]:
 * out[i,:] = params[idx[i],:]; out-of-range -> zeros (EmbeddingLookup semantics).
 * Call sites: embedding.py:150,194; deep_and_cross.py:199; wide_and_deep.py:277-290.
 * row_scale fuses the mask multiply of wide_and_deep.py:303-309 (vx = emb * mask).
 * ---------------------------------------------------------------------------------------- */
MREC_O_API void mrec_o_gather_rows_f32(const float* table, int64_t V, int64_t ld, int32_t D,
                                       const int64_t* ids, int64_t n, const float* row_scale,
                                       float* out) {
    for (int64_t i = 0; i < n; ++i) {
        int64_t r = ids[i];
        float* o = out + i * D;
        if (r < 0 || r >= V) { memset(o, 0, sizeof(float) * (size_t)D); continue; }
        const float* t = table + r * ld;
        if (row_scale) { float s = row_scale[i]; for (int32_t c = 0; c < D; ++c) o[c] = t[c] * s; }
        else for (int32_t c = 0; c < D; ++c) o[c] = t[c];
    }
}

/* wide branch of WideDeepModel.construct, wide_and_deep.py:300,303-306:
 * wide_out[b] = sum_f w[id[b,f]] * wt[b,f] + wide_b  (ReduceSum over axis 1, sequential in f). */
MREC_O_API void mrec_o_wide_sum_f32(const float* w, int64_t V, const int64_t* ids, const float* wts,
                                    int64_t B, int32_t F, float bias, float* out) {
    for (int64_t b = 0; b < B; ++b) {
        float acc = 0.0f;
        for (int32_t f = 0; f < F; ++f) {
            int64_t r = ids[b * F + f];
            float x = (r < 0 || r >= V) ? 0.0f : w[r];
            acc = acc + x * wts[b * F + f];
        }
        out[b] = acc + bias;
    }
}

/* ops.UnsortedSegmentSum [EXT, SURVEY 2.2]: out[seg[i],:] += vals[i,:], i ascending. */
MREC_O_API void mrec_o_segment_sum_f32(const float* vals, int64_t ldv, const int32_t* seg, int64_t n,
                                       int32_t D, float* out, int64_t U) {
    memset(out, 0, sizeof(float) * (size_t)(U * D));
    for (int64_t i = 0; i < n; ++i) {
        if (seg[i] < 0 || seg[i] >= U) continue;
        float* o = out + (int64_t)seg[i] * D;
        const float* v = vals + i * ldv;
        for (int32_t c = 0; c < D; ++c) o[c] = o[c] + v[c];
    }
}

/* Row-gradient as the optimizer sees it: ((g * mask) * grad_scale) summed per unique id in
 * ascending position order (Mul bprop, then optimizer loss-scale, then RowTensor dedup =
 * Unique + UnsortedSegmentSum).  The first contribution initialises the sum
 * (so -0.0 survives exactly as in `0 + x` only when x is +-0; harmless). */
static void dedup_grads(const int64_t* ids, int64_t n, int32_t D, const float* g, int64_t ldg,
                        const float* row_scale, float gscale, int64_t** uniq_out, float** sum_out,
                        int64_t* U_out) {
    int64_t* uniq = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n ? n : 1));
    int32_t* inv = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n ? n : 1));
    int64_t U = unique_i64_impl(ids, n, uniq, inv);
    float* sum = (float*)calloc((size_t)((U ? U : 1) * D), sizeof(float));
    char* seen = (char*)calloc((size_t)(U ? U : 1), 1);
    for (int64_t i = 0; i < n; ++i) {
        float* o = sum + (int64_t)inv[i] * D;
        const float* gi = g + i * ldg;
        float s = row_scale ? row_scale[i] : 1.0f;
        if (!seen[inv[i]]) {
            seen[inv[i]] = 1;
            for (int32_t c = 0; c < D; ++c) {
                float x = row_scale ? gi[c] * s : gi[c];
                o[c] = x * gscale;
            }
        } else {
            for (int32_t c = 0; c < D; ++c) {
                float x = row_scale ? gi[c] * s : gi[c];
                o[c] = o[c] + x * gscale;
            }
        }
    }
    free(seen); free(inv);
    *uniq_out = uniq; *sum_out = sum; *U_out = U;
}

/* ------------------------------------------------------------------------------------------
 * nn.LazyAdam on a RowTensor gradient [EXT, SURVEY A.4]; reference construction
 * wide_and_deep.py:420-422 (lr 3.5e-4, eps 1e-8, loss_scale=sens).  Only touched rows move.
 *   lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t)
 *   m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g
 *   p = p - lr_t * (nesterov ? b1*m + (1-b1)*g : m) / (sqrt(v) + eps)
 * Rows with id outside [0,V) are ignored.
 * ---------------------------------------------------------------------------------------- */
MREC_O_API void mrec_o_sparse_lazy_adam_f32(float* p, float* m, float* v, int64_t V, int64_t ld,
                                            int32_t D, const int64_t* ids, int64_t n, const float* g,
                                            int64_t ldg, const float* row_scale, float lr, float b1,
                                            float b2, float eps, float b1_pow, float b2_pow,
                                            float grad_scale, int nesterov) {
    int64_t* uniq; float* sum; int64_t U;
    dedup_grads(ids, n, D, g, ldg, row_scale, grad_scale, &uniq, &sum, &U);
    float lr_t = lr * sqrtf(1.0f - b2_pow) / (1.0f - b1_pow);
    float omb1 = 1.0f - b1, omb2 = 1.0f - b2;
    for (int64_t u = 0; u < U; ++u) {
        int64_t r = uniq[u];
        if (r < 0 || r >= V) continue;
        float *pp = p + r * ld, *mm = m + r * ld, *vv = v + r * ld;
        const float* gg = sum + u * D;
        for (int32_t c = 0; c < D; ++c) {
            float gc = gg[c];
            float mn = b1 * mm[c] + omb1 * gc;
            float vn = b2 * vv[c] + omb2 * (gc * gc);
            float num = nesterov ? (b1 * mn + omb1 * gc) : mn;
            pp[c] = pp[c] - (lr_t * num) / (sqrtf(vn) + eps);
            mm[c] = mn; vv[c] = vn;
        }
    }
    free(uniq); free(sum);
}

/* ------------------------------------------------------------------------------------------
 * nn.FTRL sparse apply (FusedSparseFtrl / SparseApplyFtrl) [EXT, SURVEY A.5]; reference
 * construction wide_and_deep.py:423-430 (lr 5e-2, l1 1e-8, l2 1e-8, initial_accum 1.0).
 *   a' = a + g*g
 *   y  = (lr_power == -0.5) ? sqrt(a') : pow(a', -lr_power); y0 likewise from a
 *   linear += g - (y - y0)/lr * w
 *   w = (clip(linear,-l1,l1) - linear) / (y/lr + 2*l2) ;  a = a'
 * ---------------------------------------------------------------------------------------- */
static inline void ftrl_elem(float* w, float* a, float* lin, float g, float lr, float l1, float l2,
                             float lr_power) {
    float an = *a + g * g;
    float y, y0;
    if (lr_power == -0.5f) { y = sqrtf(an); y0 = sqrtf(*a); }
    else { y = powf(an, -lr_power); y0 = powf(*a, -lr_power); }
    float sigma = (y - y0) / lr;
    float ln = *lin + (g - sigma * (*w));
    float cl = ln < -l1 ? -l1 : (ln > l1 ? l1 : ln);
    float x = cl - ln;
    float q = y / lr + 2.0f * l2;
    *w = x / q; *lin = ln; *a = an;
}

MREC_O_API void mrec_o_sparse_ftrl_f32(float* var, float* accum, float* linear, int64_t V,
                                       int64_t ld, int32_t D, const int64_t* ids, int64_t n,
                                       const float* g, int64_t ldg, const float* row_scale, float lr,
                                       float l1, float l2, float lr_power, float grad_scale) {
    int64_t* uniq; float* sum; int64_t U;
    dedup_grads(ids, n, D, g, ldg, row_scale, grad_scale, &uniq, &sum, &U);
    for (int64_t u = 0; u < U; ++u) {
        int64_t r = uniq[u];
        if (r < 0 || r >= V) continue;
        for (int32_t c = 0; c < D; ++c)
            ftrl_elem(var + r * ld + c, accum + r * ld + c, linear + r * ld + c, sum[u * D + c], lr,
                      l1, l2, lr_power);
    }
    free(uniq); free(sum);
}

/* ------------------------------------------------------------------------------------------
 * *_mt entries: the same restatements run on T host threads (bench.py's all-core cpu_baseline leg).
 * Results are bit-identical to the single-thread entries: rows of the output (gather, wide sum) or
 * unique ids (sparse applies) are partitioned over the threads, and every unique id's contributions
 * are still added in ascending position order (a stable counting sort of the positions by group).
 * The Unique itself stays sequential (first-occurrence numbering is inherently ordered).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int kind; int64_t lo, hi;
    /* gather / wide */
    const float* table; int64_t V, ld; int32_t D; const int64_t* ids; const float* rs; float* out; int32_t F; float bias;
    /* apply */
    float *p, *m, *v; const int64_t* uniq; const int32_t *order, *start; const float* g; int64_t ldg;
    float gscale, lr, b1, b2, eps, lr_t, l1, l2, lr_power; int nesterov;
} mt_job_t;

static void* mt_worker(void* arg) {
    mt_job_t* j = (mt_job_t*)arg;
    if (j->kind == 0) {
        mrec_o_gather_rows_f32(j->table, j->V, j->ld, j->D, j->ids + j->lo, j->hi - j->lo, j->rs ? j->rs + j->lo : NULL,
                               j->out + j->lo * j->D);
    } else if (j->kind == 1) {
        mrec_o_wide_sum_f32(j->table, j->V, j->ids + j->lo * j->F, j->rs + j->lo * j->F, j->hi - j->lo, j->F, j->bias,
                            j->out + j->lo);
    } else {
        const int32_t D = j->D;
        float* sum = (float*)malloc(sizeof(float) * (size_t)D);
        float omb1 = 1.0f - j->b1, omb2 = 1.0f - j->b2;
        for (int64_t u = j->lo; u < j->hi; ++u) {
            for (int32_t e = j->start[u]; e < j->start[u + 1]; ++e) {          /* ascending position order */
                const int64_t i = j->order[e];
                const float* gi = j->g + i * j->ldg;
                const float s = j->rs ? j->rs[i] : 1.0f;
                for (int32_t c = 0; c < D; ++c) {
                    float x = j->rs ? gi[c] * s : gi[c];
                    if (e == j->start[u]) sum[c] = x * j->gscale; else sum[c] = sum[c] + x * j->gscale;
                }
            }
            const int64_t r = j->uniq[u];
            if (r < 0 || r >= j->V) continue;
            if (j->kind == 2) {
                float *pp = j->p + r * j->ld, *mm = j->m + r * j->ld, *vv = j->v + r * j->ld;
                for (int32_t c = 0; c < D; ++c) {
                    float gc = sum[c];
                    float mn = j->b1 * mm[c] + omb1 * gc;
                    float vn = j->b2 * vv[c] + omb2 * (gc * gc);
                    float num = j->nesterov ? (j->b1 * mn + omb1 * gc) : mn;
                    pp[c] = pp[c] - (j->lr_t * num) / (sqrtf(vn) + j->eps);
                    mm[c] = mn; vv[c] = vn;
                }
            } else {
                for (int32_t c = 0; c < D; ++c)
                    ftrl_elem(j->p + r * j->ld + c, j->m + r * j->ld + c, j->v + r * j->ld + c, sum[c], j->lr, j->l1, j->l2,
                              j->lr_power);
            }
        }
        free(sum);
    }
    return NULL;
}

static void mt_run(mt_job_t* proto, int64_t n, int T) {
    if (T < 1) T = 1;
    if (T > 256) T = 256;
    pthread_t th[256]; mt_job_t jobs[256];
    for (int t = 0; t < T; ++t) {
        jobs[t] = *proto;
        jobs[t].lo = n * t / T; jobs[t].hi = n * (t + 1) / T;
        pthread_create(&th[t], NULL, mt_worker, &jobs[t]);
    }
    for (int t = 0; t < T; ++t) pthread_join(th[t], NULL);
}

MREC_O_API void mrec_o_gather_rows_f32_mt(const float* table, int64_t V, int64_t ld, int32_t D, const int64_t* ids,
                                          int64_t n, const float* row_scale, float* out, int threads) {
    mt_job_t j; memset(&j, 0, sizeof j);
    j.kind = 0; j.table = table; j.V = V; j.ld = ld; j.D = D; j.ids = ids; j.rs = row_scale; j.out = out;
    mt_run(&j, n, threads);
}

MREC_O_API void mrec_o_wide_sum_f32_mt(const float* w, int64_t V, const int64_t* ids, const float* wts, int64_t B,
                                       int32_t F, float bias, float* out, int threads) {
    mt_job_t j; memset(&j, 0, sizeof j);
    j.kind = 1; j.table = w; j.V = V; j.ids = ids; j.rs = wts; j.F = F; j.bias = bias; j.out = out;
    mt_run(&j, B, threads);
}

static void mt_apply(int kind, float* p, float* m, float* v, int64_t V, int64_t ld, int32_t D, const int64_t* ids,
                     int64_t n, const float* g, int64_t ldg, const float* row_scale, mt_job_t* hp, int threads) {
    int64_t* uniq = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n ? n : 1));
    int32_t* inv = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n ? n : 1));
    int64_t U = unique_i64_impl(ids, n, uniq, inv);
    int32_t* start = (int32_t*)calloc((size_t)(U + 2), sizeof(int32_t));
    int32_t* order = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n ? n : 1));
    for (int64_t i = 0; i < n; ++i) start[inv[i] + 1]++;
    for (int64_t u = 0; u < U; ++u) start[u + 1] += start[u];
    int32_t* fill = (int32_t*)malloc(sizeof(int32_t) * (size_t)(U ? U : 1));
    for (int64_t u = 0; u < U; ++u) fill[u] = start[u];
    for (int64_t i = 0; i < n; ++i) order[fill[inv[i]]++] = (int32_t)i;       /* stable: ascending i per group */
    mt_job_t j = *hp;
    j.kind = kind; j.p = p; j.m = m; j.v = v; j.V = V; j.ld = ld; j.D = D; j.uniq = uniq; j.order = order; j.start = start;
    j.g = g; j.ldg = ldg; j.rs = row_scale;
    mt_run(&j, U, threads);
    free(uniq); free(inv); free(start); free(order); free(fill);
}

MREC_O_API void mrec_o_sparse_lazy_adam_f32_mt(float* p, float* m, float* v, int64_t V, int64_t ld, int32_t D,
                                               const int64_t* ids, int64_t n, const float* g, int64_t ldg,
                                               const float* row_scale, float lr, float b1, float b2, float eps,
                                               float b1_pow, float b2_pow, float grad_scale, int nesterov, int threads) {
    mt_job_t h; memset(&h, 0, sizeof h);
    h.gscale = grad_scale; h.b1 = b1; h.b2 = b2; h.eps = eps; h.nesterov = nesterov;
    h.lr_t = lr * sqrtf(1.0f - b2_pow) / (1.0f - b1_pow);
    mt_apply(2, p, m, v, V, ld, D, ids, n, g, ldg, row_scale, &h, threads);
}

MREC_O_API void mrec_o_sparse_ftrl_f32_mt(float* var, float* accum, float* linear, int64_t V, int64_t ld, int32_t D,
                                          const int64_t* ids, int64_t n, const float* g, int64_t ldg,
                                          const float* row_scale, float lr, float l1, float l2, float lr_power,
                                          float grad_scale, int threads) {
    mt_job_t h; memset(&h, 0, sizeof h);
    h.gscale = grad_scale; h.lr = lr; h.l1 = l1; h.l2 = l2; h.lr_power = lr_power;
    mt_apply(3, var, accum, linear, V, ld, D, ids, n, g, ldg, row_scale, &h, threads);
}

/* Dense nn.Adam / nn.FTRL over a whole tensor [EXT A.4/A.5]; wide_and_deep.py:435-445,
 * deep_and_cross.py:342-344.  g is scaled by grad_scale first (optimizer loss_scale). */
MREC_O_API void mrec_o_dense_adam_f32(float* p, float* m, float* v, const float* g, int64_t n,
                                      float lr, float b1, float b2, float eps, float b1_pow,
                                      float b2_pow, float grad_scale, int nesterov) {
    float lr_t = lr * sqrtf(1.0f - b2_pow) / (1.0f - b1_pow);
    float omb1 = 1.0f - b1, omb2 = 1.0f - b2;
    for (int64_t i = 0; i < n; ++i) {
        float gc = g[i] * grad_scale;
        float mn = b1 * m[i] + omb1 * gc;
        float vn = b2 * v[i] + omb2 * (gc * gc);
        float num = nesterov ? (b1 * mn + omb1 * gc) : mn;
        p[i] = p[i] - (lr_t * num) / (sqrtf(vn) + eps);
        m[i] = mn; v[i] = vn;
    }
}

MREC_O_API void mrec_o_dense_ftrl_f32(float* var, float* accum, float* linear, const float* g,
                                      int64_t n, float lr, float l1, float l2, float lr_power,
                                      float grad_scale) {
    for (int64_t i = 0; i < n; ++i)
        ftrl_elem(var + i, accum + i, linear + i, g[i] * grad_scale, lr, l1, l2, lr_power);
}

/* ------------------------------------------------------------------------------------------
 * MapParameter (mindspore.experimental) [EXT, SURVEY A.6]; API by example README.md:160-205,
 * built at mindspore_rec/ops/embedding.py:136-146, read through MapTensorGet(insert_default=True)
 * at embedding.py:149,193,199.  CPU backing store restated as: open-addressing key index +
 * append-only row store; a missed key is inserted in call order with a row drawn from
 * default_value (here: sigma * det_normal(seed, key, col), or a constant when sigma < 0).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int32_t D; int64_t cap_rows; int64_t n_rows; uint64_t nslots;
    int64_t* skey; int32_t* srow; /* srow: -1 empty, -2 tombstone */
    float* rows; int64_t* row_key; char* live;
    uint64_t seed; float sigma; float fill;
} omap_t;

MREC_O_API void* mrec_o_map_create(int32_t D, int64_t cap_rows, uint64_t seed, float sigma, float fill) {
    omap_t* h = (omap_t*)calloc(1, sizeof(omap_t));
    h->D = D; h->cap_rows = cap_rows; h->seed = seed; h->sigma = sigma; h->fill = fill;
    h->nslots = 16; while (h->nslots < (uint64_t)cap_rows * 2) h->nslots <<= 1;
    h->skey = (int64_t*)calloc(h->nslots, sizeof(int64_t));
    h->srow = (int32_t*)malloc(h->nslots * sizeof(int32_t));
    for (uint64_t i = 0; i < h->nslots; ++i) h->srow[i] = -1;
    h->rows = (float*)calloc((size_t)(cap_rows * D), sizeof(float));
    h->row_key = (int64_t*)calloc((size_t)cap_rows, sizeof(int64_t));
    h->live = (char*)calloc((size_t)cap_rows, 1);
    return h;
}

MREC_O_API void mrec_o_map_destroy(void* hp) {
    omap_t* h = (omap_t*)hp;
    free(h->skey); free(h->srow); free(h->rows); free(h->row_key); free(h->live); free(h);
}

static int32_t omap_find(omap_t* h, int64_t key) {
    uint64_t s = ohash(key) & (h->nslots - 1);
    while (h->srow[s] != -1) {
        if (h->srow[s] >= 0 && h->skey[s] == key) return h->srow[s];
        s = (s + 1) & (h->nslots - 1);
    }
    return -1;
}

static int32_t omap_insert(omap_t* h, int64_t key) {
    if (h->n_rows >= h->cap_rows) return -1;
    uint64_t s = ohash(key) & (h->nslots - 1);
    while (h->srow[s] >= 0) s = (s + 1) & (h->nslots - 1);
    int32_t r = (int32_t)h->n_rows++;
    h->srow[s] = r; h->skey[s] = key; h->row_key[r] = key; h->live[r] = 1;
    return r;
}

/* returns row indices (or -1 when missing and !insert_default, or table full) */
MREC_O_API void mrec_o_map_find_or_insert(void* hp, const int64_t* keys, int64_t n, int insert_default,
                                          int32_t* rows_out) {
    omap_t* h = (omap_t*)hp;
    for (int64_t i = 0; i < n; ++i) {
        int32_t r = omap_find(h, keys[i]);
        if (r < 0 && insert_default) {
            r = omap_insert(h, keys[i]);
            if (r >= 0) {
                float* o = h->rows + (int64_t)r * h->D;
                for (int32_t c = 0; c < h->D; ++c)
                    o[c] = h->sigma >= 0.0f ? h->sigma * det_normal(h->seed, keys[i], c) : h->fill;
            }
        }
        rows_out[i] = r;
    }
}

/* MapTensorGet: values for keys; missing (not inserted) -> default row without insertion. */
MREC_O_API void mrec_o_map_get(void* hp, const int64_t* keys, int64_t n, int insert_default, float* out) {
    omap_t* h = (omap_t*)hp;
    for (int64_t i = 0; i < n; ++i) {
        int32_t r;
        mrec_o_map_find_or_insert(hp, keys + i, 1, insert_default, &r);
        float* o = out + i * h->D;
        if (r >= 0) memcpy(o, h->rows + (int64_t)r * h->D, sizeof(float) * (size_t)h->D);
        else for (int32_t c = 0; c < h->D; ++c)
            o[c] = h->sigma >= 0.0f ? h->sigma * det_normal(h->seed, keys[i], c) : h->fill;
    }
}

/* MapTensorPut: upsert (README.md:188-190, `m[keys] = values`); later duplicates win. */
MREC_O_API void mrec_o_map_put(void* hp, const int64_t* keys, int64_t n, const float* vals) {
    omap_t* h = (omap_t*)hp;
    for (int64_t i = 0; i < n; ++i) {
        int32_t r = omap_find(h, keys[i]);
        if (r < 0) r = omap_insert(h, keys[i]);
        if (r >= 0) memcpy(h->rows + (int64_t)r * h->D, vals + i * h->D, sizeof(float) * (size_t)h->D);
    }
}

/* MapTensorErase (README.md:193-195). */
MREC_O_API void mrec_o_map_erase(void* hp, const int64_t* keys, int64_t n) {
    omap_t* h = (omap_t*)hp;
    for (int64_t i = 0; i < n; ++i) {
        uint64_t s = ohash(keys[i]) & (h->nslots - 1);
        while (h->srow[s] != -1) {
            if (h->srow[s] >= 0 && h->skey[s] == keys[i]) {
                h->live[h->srow[s]] = 0; h->srow[s] = -2; break;
            }
            s = (s + 1) & (h->nslots - 1);
        }
    }
}

MREC_O_API int64_t mrec_o_map_size(void* hp) {
    omap_t* h = (omap_t*)hp; int64_t c = 0;
    for (int64_t r = 0; r < h->n_rows; ++r) c += h->live[r];
    return c;
}

/* get_data(): live (key,row) pairs in row order. */
MREC_O_API int64_t mrec_o_map_export(void* hp, int64_t* keys, float* vals) {
    omap_t* h = (omap_t*)hp; int64_t c = 0;
    for (int64_t r = 0; r < h->n_rows; ++r) if (h->live[r]) {
        keys[c] = h->row_key[r];
        memcpy(vals + c * h->D, h->rows + r * h->D, sizeof(float) * (size_t)h->D);
        ++c;
    }
    return c;
}

MREC_O_API float* mrec_o_map_rows_ptr(void* hp) { return ((omap_t*)hp)->rows; }

/* ------------------------------------------------------------------------------------------
 * CrossLayer (DCN-v1), models/deep_and_cross/src/deep_and_cross.py:139-149:
 *   y = x0 * (x_l . w) + b + x_l        w,b in R^D;  applied L times with x_l := y.
 * Dot product accumulated sequentially in column order.
 * ---------------------------------------------------------------------------------------- */
MREC_O_API void mrec_o_cross_layers_f32(const float* x0, const float* w, const float* b, int32_t L,
                                        int64_t B, int32_t D, float* out, float* xl_save /*[L,B,D] or NULL*/) {
    float* cur = (float*)malloc(sizeof(float) * (size_t)D);
    for (int64_t r = 0; r < B; ++r) {
        const float* x = x0 + r * D;
        memcpy(cur, x, sizeof(float) * (size_t)D);
        for (int32_t l = 0; l < L; ++l) {
            if (xl_save) memcpy(xl_save + ((int64_t)l * B + r) * D, cur, sizeof(float) * (size_t)D);
            float s = 0.0f;
            for (int32_t c = 0; c < D; ++c) s = s + cur[c] * w[l * D + c];
            for (int32_t c = 0; c < D; ++c) cur[c] = (x[c] * s + b[l * D + c]) + cur[c];
        }
        memcpy(out + r * D, cur, sizeof(float) * (size_t)D);
    }
    free(cur);
}

/* Backward of the L-layer stack given dy=[B,D]; returns dx0, dw[L,D], db[L,D].
 * (bprop of deep_and_cross.py:143-149 derived by hand; double accumulators for dw/db so the
 * checker is tighter than the thing it checks.) */
MREC_O_API void mrec_o_cross_layers_bwd_f32(const float* x0, const float* w, const float* b, int32_t L,
                                            int64_t B, int32_t D, const float* dy, float* dx0,
                                            float* dw, float* db) {
    (void)b;
    double* dwa = (double*)calloc((size_t)(L * D), sizeof(double));
    double* dba = (double*)calloc((size_t)(L * D), sizeof(double));
    float* xs = (float*)malloc(sizeof(float) * (size_t)((L + 1) * D));
    double* gy = (double*)malloc(sizeof(double) * (size_t)D);
    double* gx0 = (double*)malloc(sizeof(double) * (size_t)D);
    for (int64_t r = 0; r < B; ++r) {
        const float* x = x0 + r * D;
        memcpy(xs, x, sizeof(float) * (size_t)D);
        for (int32_t l = 0; l < L; ++l) {
            float s = 0.0f; const float* c0 = xs + l * D; float* c1 = xs + (l + 1) * D;
            for (int32_t c = 0; c < D; ++c) s = s + c0[c] * w[l * D + c];
            for (int32_t c = 0; c < D; ++c) c1[c] = (x[c] * s + b[l * D + c]) + c0[c];
        }
        for (int32_t c = 0; c < D; ++c) { gy[c] = dy[r * D + c]; gx0[c] = 0.0; }
        for (int32_t l = L - 1; l >= 0; --l) {
            const float* xl = xs + l * D;
            double s = 0.0, t = 0.0;
            for (int32_t c = 0; c < D; ++c) { s += (double)xl[c] * w[l * D + c]; t += gy[c] * x[c]; }
            for (int32_t c = 0; c < D; ++c) {
                dba[l * D + c] += gy[c];
                dwa[l * D + c] += t * xl[c];
                gx0[c] += gy[c] * s;
                gy[c] = gy[c] + t * w[l * D + c];
            }
        }
        for (int32_t c = 0; c < D; ++c) dx0[r * D + c] = (float)(gx0[c] + gy[c]);
    }
    for (int32_t i = 0; i < L * D; ++i) { dw[i] = (float)dwa[i]; db[i] = (float)dba[i]; }
    free(dwa); free(dba); free(xs); free(gy); free(gx0);
}

/* ------------------------------------------------------------------------------------------
 * Row-shard routing for the hybrid-parallel mode (README.md:140-144):
 * owner(id) = id mod n_shards, local row = id div n_shards.  Stable bucketing by owner.
 * ---------------------------------------------------------------------------------------- */
MREC_O_API void mrec_o_shard_route_i64(const int64_t* ids, int64_t n, int32_t n_shards,
                                       int64_t* send_local /*[n]*/, int32_t* send_perm /*[n]*/,
                                       int64_t* counts /*[n_shards]*/) {
    memset(counts, 0, sizeof(int64_t) * (size_t)n_shards);
    for (int64_t i = 0; i < n; ++i) {
        int64_t o = ids[i] % n_shards; if (o < 0) o += n_shards;
        counts[o]++;
    }
    int64_t* off = (int64_t*)calloc((size_t)n_shards + 1, sizeof(int64_t));
    for (int32_t s = 0; s < n_shards; ++s) off[s + 1] = off[s] + counts[s];
    for (int64_t i = 0; i < n; ++i) {
        int64_t o = ids[i] % n_shards; if (o < 0) o += n_shards;
        int64_t loc = (ids[i] - o) / n_shards;
        int64_t d = off[o]++;
        send_local[d] = loc; send_perm[d] = (int32_t)i;
    }
    free(off);
}
