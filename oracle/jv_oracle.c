/*
 * jv_oracle.c — CPU restatement (ORACLE) of the jVector GraphSearcher hot path.  See jv_oracle.h:
 * test infrastructure only; "parity unpinned" beyond the reference's analytic known-answer tests.
 *
 * What each part follows:
 *   heap keys / NodeQueue ........ jvector NodeQueue + BoundedLongHeap/GrowableLongHeap
 *   search loop .................. jvector GraphSearcher.search/searchOneLayer,
 *                                  called from J/JVectorReader.java:165-173
 *   accept predicate ............. J/JVectorReader.java:157-163 + J/GraphNodeIdToDocMap.java:159-161
 *   score-provider choice ........ J/JVectorReader.java:352-365 (PQ approx + exact reranker | exact (x2 if MIP))
 *   rerank ....................... jvector NodeQueue.rerank
 *   similarity mappings .......... jvector VectorSimilarityFunction
 *   PQ LUT / ADC ................. jvector PQVectors.precomputedScoreFunctionFor / PQDecoder (App. A.4)
 *   exact scorer ................. J/JVectorVectorScorer.java:36-53
 *
 * Compile with -ffp-contract=off: every fused multiply-add below is an explicit fmaf(), every other
 * operation rounds separately, exactly like the HIP kernels.
 */
#include "jv_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------ */
/* NodeQueue keys                                                                             */
/* ------------------------------------------------------------------------------------------ */

int32_t jvo_float_to_sortable_int(float f) {
    int32_t bits;
    memcpy(&bits, &f, 4);
    return bits ^ ((bits >> 31) & 0x7fffffff);
}

float jvo_sortable_int_to_float(int32_t s) {
    int32_t bits = s ^ ((s >> 31) & 0x7fffffff);
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

/* key = (sortable(score) << 32) | (0xFFFFFFFF & ~node): higher score = larger key; among equal
 * scores the LOWER node id has the larger key. */
int64_t jvo_encode_key(int32_t node, float score) {
    return (int64_t)(((uint64_t)(uint32_t)jvo_float_to_sortable_int(score) << 32) | (uint64_t)(uint32_t)(~node));
}
static inline int32_t key_node(int64_t key) { return ~(int32_t)(uint32_t)(key & 0xFFFFFFFFll); }
static inline float key_score(int64_t key) { return jvo_sortable_int_to_float((int32_t)(key >> 32)); }

/* ------------------------------------------------------------------------------------------ */
/* Canonical fp32 accumulation                                                                */
/* ------------------------------------------------------------------------------------------ */

/* adjacent-pair binary tree over 64 partials: ((P0+P1)+(P2+P3)) + ... */
static inline float tree64(float* p) {
    for (int w = 32; w >= 1; w >>= 1)
        for (int i = 0; i < w; i++) p[i] = p[2 * i] + p[2 * i + 1];
    return p[0];
}

/* Elements are consumed in float4 groups (one GPU lane loads 16 B): indices in [d, roundup(d,4))
 * are zero-padding and still pass through the fma (the HBM rows are zero-padded to 4 floats);
 * groups that start at or beyond roundup(d,4) do not exist. */
float jvo_raw_dot(const float* a, const float* b, int d) {
    float p[64];
    for (int m = 0; m < 64; m++) p[m] = 0.0f;
    int d4 = (d + 3) & ~3;
    int full = d & ~63;
    for (int base = 0; base < full; base += 64)
        for (int m = 0; m < 64; m++) p[m] = fmaf(a[base + m], b[base + m], p[m]);
    for (int i = full; i < d4; i++) {
        float x = i < d ? a[i] : 0.0f, y = i < d ? b[i] : 0.0f;
        p[i & 63] = fmaf(x, y, p[i & 63]);
    }
    return tree64(p);
}

float jvo_raw_l2(const float* a, const float* b, int d) {
    float p[64];
    for (int m = 0; m < 64; m++) p[m] = 0.0f;
    int d4 = (d + 3) & ~3;
    int full = d & ~63;
    for (int base = 0; base < full; base += 64)
        for (int m = 0; m < 64; m++) {
            float t = a[base + m] - b[base + m];
            p[m] = fmaf(t, t, p[m]);
        }
    for (int i = full; i < d4; i++) {
        float t = i < d ? a[i] - b[i] : 0.0f;
        p[i & 63] = fmaf(t, t, p[i & 63]);
    }
    return tree64(p);
}

static inline float map_score(int sim, float raw) {
    if (sim == JV_SIM_EUCLIDEAN) return 1.0f / (1.0f + raw);
    return (1.0f + raw) / 2.0f; /* DOT_PRODUCT and (raw = cosine) COSINE */
}

static inline float cosine_from(float dot, float na, float nb) { return dot / sqrtf(na * nb); }

/* ------------------------------------------------------------------------------------------ */
/* NVQ-inline vectors: dequantisation (J/JVectorIndexQuantization.java:319-361, restated)        */
/* ------------------------------------------------------------------------------------------ */

/* Java Math.round(float): closest int, ties towards positive infinity */
static inline int java_round(float x) {
    float r = floorf(x);
    return (int)r + ((x - r) >= 0.5f ? 1 : 0);
}
/* logisticNQT (:344-350) */
static inline float nvq_logistic(float value, float alpha, float x0) {
    float temp = fmaf(value, alpha, -alpha * x0);
    int p = java_round(temp + 0.5f);
    float f = fmaf(temp - (float)p, 0.5f, 1.0f);
    int32_t m;
    memcpy(&m, &f, 4);
    m = (int32_t)((uint32_t)m + ((uint32_t)p << 23));
    memcpy(&temp, &m, 4);
    return temp / (temp + 1.0f);
}
/* logitNQT (:353-360) */
static inline float nvq_logit(float scaled, float inverse_alpha, float x0) {
    float z = scaled / (1.0f - scaled);
    int32_t temp;
    memcpy(&temp, &z, 4);
    int32_t e = temp & 0x7f800000;
    float p = (float)((e >> 23) - 128);
    int32_t mb = (temp & 0x007fffff) + 0x3f800000;
    float m;
    memcpy(&m, &mb, 4);
    return (m + p) * inverse_alpha + x0;
}
/* nvqDequantize (:319-341) of node's record into out[d] */
void jvo_nvq_dequantize(const jv_index_desc* ix, int node, float* out) {
    int d = ix->d, M = ix->nvq_M, off = 0;
    const uint8_t* bytes = ix->nvq_bytes + (size_t)node * d;
    for (int s = 0; s < M; s++) {
        int size = ix->nvq_sub_sizes ? ix->nvq_sub_sizes[s] : d / M + (s < d % M ? 1 : 0);
        const float* pr = ix->nvq_params + ((size_t)node * M + s) * 4;
        float growth = pr[0], midpoint = pr[1], minv = pr[2], maxv = pr[3];
        float delta = maxv - minv;
        float scaled_growth = growth / delta;
        float scaled_mid = midpoint * delta;
        float bias = nvq_logistic(minv, scaled_growth, scaled_mid);
        float scale = (nvq_logistic(maxv, scaled_growth, scaled_mid) - bias) / 255.0f;
        float inv = 1.0f / scaled_growth;
        for (int i = 0; i < size; i++) {
            float sv = fmaf((float)bytes[off + i], scale, bias);
            out[off + i] = nvq_logit(sv, inv, scaled_mid);
        }
        off += size;
    }
    for (int i = 0; i < d; i++) out[i] = out[i] + ix->nvq_global_mean[i];
}

static _Thread_local float* tls_nvq_row;
static _Thread_local int tls_nvq_cap;
/* the row every "exact" score is taken against: the full-precision vector, or the dequantised NVQ record */
static const float* exact_row(const jv_index_desc* ix, int node) {
    if (ix->nvq_M <= 0) return ix->vectors + (size_t)node * ix->d;
    if (tls_nvq_cap < ix->d) {
        free(tls_nvq_row);
        tls_nvq_row = (float*)malloc(sizeof(float) * (size_t)ix->d);
        tls_nvq_cap = ix->d;
    }
    jvo_nvq_dequantize(ix, node, tls_nvq_row);
    return tls_nvq_row;
}

static float exact_unscaled(int sim, const float* q, const float* v, int d, float qnorm2) {
    if (sim == JV_SIM_EUCLIDEAN) return map_score(sim, jvo_raw_l2(q, v, d));
    if (sim == JV_SIM_DOT_PRODUCT) return map_score(sim, jvo_raw_dot(q, v, d));
    float dot = jvo_raw_dot(q, v, d);
    float nv = jvo_raw_dot(v, v, d);
    return map_score(sim, cosine_from(dot, qnorm2, nv));
}

float jvo_exact_score(int sim, float score_scale, const float* q, const float* v, int d) {
    float qn = sim == JV_SIM_COSINE ? jvo_raw_dot(q, q, d) : 0.0f;
    float s = exact_unscaled(sim, q, v, d, qn);
    return score_scale != 1.0f ? s * score_scale : s;
}

/* ------------------------------------------------------------------------------------------ */
/* PQ                                                                                         */
/* ------------------------------------------------------------------------------------------ */

/* jvector ProductQuantization.getSubvectorSizesAndOffsets: size = d/M, first d%M subspaces +1 */
void jvo_pq_sub_layout(int d, int M, const int32_t* sub_sizes, int32_t* sizes, int32_t* offsets) {
    int off = 0;
    for (int m = 0; m < M; m++) {
        int s = sub_sizes ? sub_sizes[m] : d / M + (m < d % M ? 1 : 0);
        sizes[m] = s;
        offsets[m] = off;
        off += s;
    }
}

/* lut[m*256 + c] = dot(q'_m, codebook[m][c]) (DOT/COSINE) or sum (q'_m - codebook[m][c])^2 (L2),
 * a sequential fmaf chain over the subspace; q' = q - globalCentroid when a centroid is present. */
static void pq_build_lut_with(const jv_index_desc* ix, const float* q, float* lut, int32_t* sizes, float* qc) {
    int M = ix->pq_M, K = ix->pq_K, d = ix->d;
    int32_t* offs = sizes + M;
    jvo_pq_sub_layout(d, M, ix->pq_sub_sizes, sizes, offs);
    for (int i = 0; i < d; i++) qc[i] = ix->pq_centroid ? q[i] - ix->pq_centroid[i] : q[i];
    const float* cb = ix->pq_codebooks;
    for (int m = 0; m < M; m++) {
        int s = sizes[m];
        const float* qs = qc + offs[m];
        for (int c = 0; c < 256; c++) {
            float acc = 0.0f;
            if (c < K) {
                const float* cv = cb + (size_t)c * s;
                if (ix->similarity == JV_SIM_EUCLIDEAN) {
                    for (int i = 0; i < s; i++) {
                        float t = qs[i] - cv[i];
                        acc = fmaf(t, t, acc);
                    }
                } else {
                    for (int i = 0; i < s; i++) acc = fmaf(qs[i], cv[i], acc);
                }
            }
            lut[m * 256 + c] = acc;
        }
        cb += (size_t)K * s;
    }
}

void jvo_pq_build_lut(const jv_index_desc* ix, const float* q, float* lut) {
    int32_t* sizes = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)ix->pq_M);
    float* qc = (float*)malloc(sizeof(float) * (size_t)ix->d);
    pq_build_lut_with(ix, q, lut, sizes, qc);
    free(qc);
    free(sizes);
}

/* cosine only: norm_lut[m*256+c] = |codebook[m][c]|^2 (sequential fmaf chain) */
void jvo_pq_build_norm_lut(const jv_index_desc* ix, float* norm_lut) {
    int M = ix->pq_M, K = ix->pq_K, d = ix->d;
    int32_t* sizes = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)M);
    int32_t* offs = sizes + M;
    jvo_pq_sub_layout(d, M, ix->pq_sub_sizes, sizes, offs);
    const float* cb = ix->pq_codebooks;
    for (int m = 0; m < M; m++) {
        int s = sizes[m];
        for (int c = 0; c < 256; c++) {
            float acc = 0.0f;
            if (c < K) {
                const float* cv = cb + (size_t)c * s;
                for (int i = 0; i < s; i++) acc = fmaf(cv[i], cv[i], acc);
            }
            norm_lut[m * 256 + c] = acc;
        }
        cb += (size_t)K * s;
    }
    free(sizes);
}

/* SIMD mode of the CPU baseline (bench.py cpu_baseline reports both): 0 = the plain C loops (gcc -O3 -mavx2 -mfma
 * auto-vectorises the 64-partial exact scorer; the ADC is scalar loads), 1 = explicit AVX2: look-up-table reads as
 * 8-wide gathers, the neighbours' code rows and the rerank rows software-prefetched before they are scored.
 * The ORDER of every floating-point operation is the same in both modes, so ids, score bits and counters are identical
 * (tests/test_oracle_units.py::test_simd_mode_changes_nothing). */
#include <immintrin.h>
static int g_simd_mode = 0;
void jvo_set_simd(int mode) { g_simd_mode = mode; }
int jvo_get_simd(void) { return g_simd_mode; }

static inline float pq_chunk_avx2(const float* lut, const uint8_t* code, int m0) { /* 16 subspaces m0 .. m0 + 15, left to right */
    float t[16];
    const __m256i step = _mm256_setr_epi32(0, 256, 512, 768, 1024, 1280, 1536, 1792);
    for (int h = 0; h < 2; h++) {
        const __m256i c8 = _mm256_cvtepu8_epi32(_mm_loadl_epi64((const __m128i*)(code + m0 + 8 * h)));
        const __m256i idx = _mm256_add_epi32(_mm256_add_epi32(c8, step), _mm256_set1_epi32((m0 + 8 * h) * 256));
        _mm256_storeu_ps(t + 8 * h, _mm256_i32gather_ps(lut, idx, 4));
    }
    float s = t[0];
    for (int i = 1; i < 16; i++) s = s + t[i];
    return s;
}

/* raw(n) = sum_m lut[m][code[m]]: 16-subspace chunks summed left to right (one GPU lane each),
 * chunk sums combined by an adjacent-pair tree over next_pow2(#chunks) (missing chunks = +0). */
float jvo_pq_raw(const float* lut, const uint8_t* code, int M) {
    float cs[64];
    int nch = (M + 15) / 16;
    int np = 1;
    while (np < nch) np <<= 1;
    for (int c = 0; c < np; c++) {
        float s = 0.0f;
        if (c < nch) {
            int m0 = c * 16, m1 = m0 + 16 < M ? m0 + 16 : M;
            if (g_simd_mode && m1 - m0 == 16) {
                s = pq_chunk_avx2(lut, code, m0);
            } else {
                s = lut[m0 * 256 + code[m0]];
                for (int m = m0 + 1; m < m1; m++) s = s + lut[m * 256 + code[m]];
            }
        }
        cs[c] = s;
    }
    for (int w = np >> 1; w >= 1; w >>= 1)
        for (int i = 0; i < w; i++) cs[i] = cs[2 * i] + cs[2 * i + 1];
    return cs[0];
}

float jvo_pq_score(const jv_index_desc* ix, const float* lut, const float* norm_lut, float qnorm2, int node) {
    const uint8_t* code = ix->pq_codes + (size_t)node * ix->pq_M;
    float raw = jvo_pq_raw(lut, code, ix->pq_M);
    if (ix->similarity == JV_SIM_COSINE) {
        float na = jvo_pq_raw(norm_lut, code, ix->pq_M);
        return map_score(JV_SIM_COSINE, cosine_from(raw, qnorm2, na));
    }
    return map_score(ix->similarity, raw);
}

/* ------------------------------------------------------------------------------------------ */
/* Heaps (Lucene/jvector LongHeap shape: 1-based implicit binary MIN heap over int64)           */
/* ------------------------------------------------------------------------------------------ */

typedef struct {
    int64_t* h; /* h[1..size] */
    int size, cap;
} lheap;

static void lh_init(lheap* q, int cap) {
    q->cap = cap < 4 ? 4 : cap;
    q->size = 0;
    q->h = (int64_t*)malloc(sizeof(int64_t) * (size_t)(q->cap + 1));
}
static void lh_free(lheap* q) { free(q->h); }
/* re-use a heap's storage for the next query (searcher scratch, below) */
static void lh_reset(lheap* q, int cap) {
    if (!q->h) {
        lh_init(q, cap);
        return;
    }
    q->size = 0;
    if (q->cap < cap) {
        q->cap = cap;
        q->h = (int64_t*)realloc(q->h, sizeof(int64_t) * (size_t)(q->cap + 1));
    }
}
static void lh_up(lheap* q, int i) {
    int64_t v = q->h[i];
    int j = i >> 1;
    while (j > 0 && v < q->h[j]) {
        q->h[i] = q->h[j];
        i = j;
        j = j >> 1;
    }
    q->h[i] = v;
}
static void lh_down(lheap* q, int i) {
    int64_t v = q->h[i];
    int j = i << 1, k = j + 1;
    if (k <= q->size && q->h[k] < q->h[j]) j = k;
    while (j <= q->size && q->h[j] < v) {
        q->h[i] = q->h[j];
        i = j;
        j = i << 1;
        k = j + 1;
        if (k <= q->size && q->h[k] < q->h[j]) j = k;
    }
    q->h[i] = v;
}
static void lh_push(lheap* q, int64_t v) {
    if (q->size == q->cap) {
        q->cap = q->cap * 2;
        q->h = (int64_t*)realloc(q->h, sizeof(int64_t) * (size_t)(q->cap + 1));
    }
    q->h[++q->size] = v;
    lh_up(q, q->size);
}
static int64_t lh_pop(lheap* q) {
    int64_t r = q->h[1];
    q->h[1] = q->h[q->size--];
    if (q->size > 0) lh_down(q, 1);
    return r;
}
/* BoundedLongHeap.push: when full reject if v < top else replace top */
static int lh_push_bounded(lheap* q, int64_t v, int max) {
    if (q->size >= max) {
        if (v < q->h[1]) return 0;
        q->h[1] = v;
        lh_down(q, 1);
        return 1;
    }
    lh_push(q, v);
    return 1;
}

/* ------------------------------------------------------------------------------------------ */
/* ScoreTracker.TwoPhaseTracker (threshold queries only; jvector graph/ScoreTracker.java)      */
/* restated from the published algorithm: window of the 500 most recent scores, bounded heap of */
/* the 100 best; evaluated when observationCount is a multiple of 100 (>= 500):                */
/* stop iff percentile99(window) < threshold && worst-of-best < threshold.                     */
/* percentile = commons-math3 StatUtils.percentile (LEGACY estimation: pos = p(n+1)/100).       */
/* ------------------------------------------------------------------------------------------ */
#define TRK_RECENT 500
#define TRK_BEST 100
typedef struct {
    double recent[TRK_RECENT];
    int recent_idx;
    lheap best; /* min-heap of sortable ints, bounded TRK_BEST */
    int observations;
    double threshold;
} tracker;

static int cmp_double(const void* a, const void* b) {
    double x = *(const double*)a, y = *(const double*)b;
    return x < y ? -1 : x > y ? 1 : 0;
}
static void trk_track(tracker* t, float score) {
    lh_push_bounded(&t->best, (int64_t)jvo_float_to_sortable_int(score), TRK_BEST);
    t->recent[t->recent_idx] = (double)score;
    t->recent_idx = (t->recent_idx + 1) % TRK_RECENT;
    t->observations++;
}
static int trk_should_stop(tracker* t) {
    if (t->observations < TRK_RECENT) return 0;
    if (t->observations % 100 != 0) return 0;
    double sorted[TRK_RECENT];
    memcpy(sorted, t->recent, sizeof(sorted));
    qsort(sorted, TRK_RECENT, sizeof(double), cmp_double);
    double pos = 99.0 * (TRK_RECENT + 1) / 100.0;
    double fpos = floor(pos);
    int ip = (int)fpos;
    double dd = pos - fpos;
    double pct;
    if (pos < 1) pct = sorted[0];
    else if (pos >= TRK_RECENT) pct = sorted[TRK_RECENT - 1];
    else pct = sorted[ip - 1] + dd * (sorted[ip] - sorted[ip - 1]);
    double worst_best = (double)jvo_sortable_int_to_float((int32_t)t->best.h[1]);
    return pct < t->threshold && worst_best < t->threshold;
}

/* ------------------------------------------------------------------------------------------ */
/* The search                                                                                 */
/* ------------------------------------------------------------------------------------------ */

typedef struct {
    const jv_index_desc* ix;
    const float* q;
    float qnorm2;          /* cosine */
    float* lut;            /* PQ approx (NULL => exact provider) */
    float* norm_lut;       /* PQ cosine */
    const uint64_t* accept;
    int64_t accept_docs;
    uint64_t* visited;     /* n bits (thread scratch, all zero between queries) */
    int32_t* touched;      /* indices of the visited words this query made non-zero */
    int n_touched, cap_touched;
    lheap candidates;      /* MAX heap: stores -key */
    lheap results;         /* bounded MIN heap (approximateResults) */
    int64_t* evicted;      /* evictedResults keys */
    int n_evicted, cap_evicted;
    int visited_count, expanded, expanded_base, reranked;
} searcher;

static inline float score_fn(searcher* s, int node) {
    const jv_index_desc* ix = s->ix;
    if (s->lut) return jvo_pq_score(ix, s->lut, s->norm_lut, s->qnorm2, node);
    float v = exact_unscaled(ix->similarity, s->q, exact_row(ix, node), ix->d, s->qnorm2);
    /* wrapExactScoreFunction: x2 for Lucene MIP, exact-provider path only (J/JVectorReader.java:220-239,359-364) */
    return ix->score_scale != 1.0f ? v * ix->score_scale : v;
}
/* view.rerankerFor(q, sim): NOT wrapped (J/JVectorReader.java:355) */
static inline float rerank_fn(searcher* s, int node) {
    const jv_index_desc* ix = s->ix;
    return exact_unscaled(ix->similarity, s->q, exact_row(ix, node), ix->d, s->qnorm2);
}

/* J/JVectorReader.java:157-163 */
static inline int accept_ord(const searcher* s, int ord) {
    if (!s->accept) return 1;
    int doc = s->ix->ord2doc ? s->ix->ord2doc[ord] : ord;
    if (doc < 0 || doc >= s->accept_docs) return 0;
    return (int)((s->accept[doc >> 6] >> (doc & 63)) & 1);
}

static inline int visited_add(searcher* s, int node) {
    uint64_t bit = 1ull << (node & 63);
    uint64_t* w = &s->visited[node >> 6];
    if (*w & bit) return 0;
    if (*w == 0) { /* remember the word so that the set is cleared in O(visited), not O(n) */
        if (s->n_touched == s->cap_touched) {
            s->cap_touched = s->cap_touched ? s->cap_touched * 2 : 4096;
            s->touched = (int32_t*)realloc(s->touched, sizeof(int32_t) * (size_t)s->cap_touched);
        }
        s->touched[s->n_touched++] = node >> 6;
    }
    *w |= bit;
    return 1;
}

static void evict_add(searcher* s, int64_t key) {
    if (s->n_evicted == s->cap_evicted) {
        s->cap_evicted = s->cap_evicted ? s->cap_evicted * 2 : 64;
        s->evicted = (int64_t*)realloc(s->evicted, sizeof(int64_t) * (size_t)s->cap_evicted);
    }
    s->evicted[s->n_evicted++] = key;
}

/* neighbours of `node` at `level`; returns row pointer and degree stride */
static const int32_t* neighbours(const jv_index_desc* ix, int level, int node, int* stride) {
    if (level == 0) {
        *stride = ix->R;
        return ix->adj + (size_t)node * ix->R;
    }
    const jv_layer_desc* L = &ix->upper_layers[level - 1];
    int lo = 0, hi = L->count - 1;
    while (lo <= hi) {
        int mid = (lo + hi) >> 1;
        if (L->nodes[mid] == node) {
            *stride = L->degree;
            return L->adj + (size_t)mid * L->degree;
        }
        if (L->nodes[mid] < node) lo = mid + 1;
        else hi = mid - 1;
    }
    *stride = 0;
    return NULL;
}

void jvo_parallel_copy(void* dst, const void* src, size_t bytes) {
    const size_t chunk = (size_t)2 << 20;
    const long nchunks = (long)((bytes + chunk - 1) / chunk);
#pragma omp parallel for schedule(static)
    for (long c = 0; c < nchunks; c++) {
        size_t o = (size_t)c * chunk;
        size_t len = bytes - o < chunk ? bytes - o : chunk;
        memcpy((char*)dst + o, (const char*)src + o, len);
    }
}

/* GraphSearcher.addTopCandidate: when full, only a STRICTLY better score replaces the worst
 * result; an equal score is treated as evicted. */
static void add_top_candidate(searcher* s, int node, float score, int rk) {
    int64_t key = jvo_encode_key(node, score);
    if (s->results.size < rk) {
        lh_push_bounded(&s->results, key, rk);
    } else if (score > key_score(s->results.h[1])) {
        evict_add(s, s->results.h[1]);
        lh_push_bounded(&s->results, key, rk);
    } else {
        evict_add(s, key);
    }
}

static void search_one_layer(searcher* s, int rk, float thr, int level, int accept_all) {
    tracker* trk = NULL;
    if (thr > 0) {
        trk = (tracker*)calloc(1, sizeof(tracker));
        lh_init(&trk->best, TRK_BEST);
        trk->threshold = (double)thr;
    }
    while (s->candidates.size > 0) {
        int64_t top = -s->candidates.h[1];
        float sc = key_score(top);
        if (s->results.size >= rk && sc < key_score(s->results.h[1])) break;
        if (thr > 0 && trk_should_stop(trk)) break;
        lh_pop(&s->candidates);
        int c = key_node(top);
        if ((accept_all || accept_ord(s, c)) && sc >= thr) add_top_candidate(s, c, sc, rk);
        int stride;
        const int32_t* nb = neighbours(s->ix, level, c, &stride);
        if (g_simd_mode) { /* the rows the loop below is about to read */
            for (int i = 0; i < stride; i++) {
                if (nb[i] < 0) continue;
                if (s->lut) __builtin_prefetch(s->ix->pq_codes + (size_t)nb[i] * s->ix->pq_M);
                else __builtin_prefetch(exact_row(s->ix, nb[i]));
            }
        }
        for (int i = 0; i < stride; i++) {
            int nn = nb[i];
            if (nn < 0) continue; /* -1 padding */
            if (!visited_add(s, nn)) continue;
            s->visited_count++;
            float fs = score_fn(s, nn);
            if (trk) trk_track(trk, fs);
            lh_push(&s->candidates, -jvo_encode_key(nn, fs));
        }
        s->expanded++;
        if (level == 0) s->expanded_base++;
    }
    if (trk) {
        lh_free(&trk->best);
        free(trk);
    }
}

static int cmp_key_desc(const void* a, const void* b) {
    int64_t x = *(const int64_t*)a, y = *(const int64_t*)b;
    return x > y ? -1 : x < y ? 1 : 0;
}

/* Per-thread reusable searcher state, as jvector keeps inside a GraphSearcher (one per Lucene search
 * thread, re-used across queries): visited bitset cleared through the list of touched words, both
 * heaps, the evicted list, the PQ tables and the result buffer.  No allocation per query once warm. */
typedef struct {
    uint64_t* visited;
    size_t visited_words;
    int32_t* touched;
    int cap_touched;
    lheap candidates, results;
    int64_t* evicted;
    int cap_evicted;
    float* lut;
    size_t lut_floats;
    float* norm_lut;
    size_t norm_floats;
    int32_t* sizes;
    int sizes_cap;
    float* qc;
    int qc_cap;
    int64_t* fin;
    int fin_cap;
} jvo_scratch;
static _Thread_local jvo_scratch tls_scratch;

int jvo_search(const jv_index_desc* ix, const float* query, int32_t topK, int32_t rerankK,
               float threshold, float rerankFloor, const uint64_t* accept_doc_words,
               int64_t accept_num_docs, int32_t* out_nodes, int32_t* out_docs, float* out_scores,
               int32_t* out_count, int32_t* out_stats) {
    if (!ix || !query || topK < 0 || rerankK < topK) return JV_EINVAL;
    for (int i = 0; i < topK; i++) {
        if (out_nodes) out_nodes[i] = -1;
        if (out_docs) out_docs[i] = -1;
        if (out_scores) out_scores[i] = 0.0f;
    }
    if (out_count) *out_count = 0;
    if (out_stats) memset(out_stats, 0, sizeof(int32_t) * JV_NUM_STATS);
    if (ix->n <= 0 || ix->entry_node < 0 || topK == 0) return JV_OK;

    jvo_scratch* sc = &tls_scratch;
    searcher s;
    memset(&s, 0, sizeof(s));
    s.ix = ix;
    s.q = query;
    s.accept = accept_doc_words;
    s.accept_docs = accept_num_docs;
    s.qnorm2 = ix->similarity == JV_SIM_COSINE ? jvo_raw_dot(query, query, ix->d) : 0.0f;
    if (ix->pq_M > 0) {
        const size_t lf = 256 * (size_t)ix->pq_M;
        if (sc->lut_floats < lf) {
            free(sc->lut);
            sc->lut = (float*)malloc(sizeof(float) * lf);
            sc->lut_floats = lf;
        }
        if (sc->sizes_cap < 2 * ix->pq_M) {
            free(sc->sizes);
            sc->sizes = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)ix->pq_M);
            sc->sizes_cap = 2 * ix->pq_M;
        }
        if (sc->qc_cap < ix->d) {
            free(sc->qc);
            sc->qc = (float*)malloc(sizeof(float) * (size_t)ix->d);
            sc->qc_cap = ix->d;
        }
        s.lut = sc->lut;
        pq_build_lut_with(ix, query, s.lut, sc->sizes, sc->qc);
        if (ix->similarity == JV_SIM_COSINE) {
            if (sc->norm_floats < lf) {
                free(sc->norm_lut);
                sc->norm_lut = (float*)malloc(sizeof(float) * lf);
                sc->norm_floats = lf;
            }
            /* rebuilt per query (the cost of one LUT): a cache keyed by the codebook ADDRESS went stale when a later
             * index's arrays were allocated where an earlier one's had been */
            jvo_pq_build_norm_lut(ix, sc->norm_lut);
            s.norm_lut = sc->norm_lut;
        }
    }
    {
        const size_t words = ((size_t)ix->n + 63) / 64;
        if (sc->visited_words < words) {
            free(sc->visited);
            sc->visited = (uint64_t*)calloc(words, sizeof(uint64_t));
            sc->visited_words = words;
        }
    }
    s.visited = sc->visited;
    s.touched = sc->touched;
    s.cap_touched = sc->cap_touched;
    s.evicted = sc->evicted;
    s.cap_evicted = sc->cap_evicted;
    lh_reset(&sc->candidates, 1024);
    lh_reset(&sc->results, rerankK > 0 ? rerankK : 1);
    s.candidates = sc->candidates;
    s.results = sc->results;

    /* initializeInternal: score the entry point, mark visited (not counted), push */
    int ep = ix->entry_node;
    float eps = score_fn(&s, ep);
    visited_add(&s, ep);
    lh_push(&s.candidates, -jvo_encode_key(ep, eps));

    /* greedy descent through the upper layers; setEntryPointsFromPreviousLayer hands every
     * popped node (results + evicted) back to the candidate queue */
    for (int lvl = ix->num_upper_layers; lvl > 0; lvl--) {
        search_one_layer(&s, 1, 0.0f, lvl, 1);
        for (int i = 1; i <= s.results.size; i++) lh_push(&s.candidates, -s.results.h[i]);
        for (int i = 0; i < s.n_evicted; i++) lh_push(&s.candidates, -s.evicted[i]);
        s.n_evicted = 0;
        s.results.size = 0;
    }
    search_one_layer(&s, rerankK, threshold, 0, 0);

    /* result assembly (App. A.3) */
    if (sc->fin_cap < s.results.size + 1) {
        free(sc->fin);
        sc->fin_cap = s.results.size + 1 + 256;
        sc->fin = (int64_t*)malloc(sizeof(int64_t) * (size_t)sc->fin_cap);
    }
    int64_t* fin = sc->fin;
    int nfin = 0;
    if (!s.lut) {
        /* exact provider: pop down to topK (worst first) => keep the topK largest keys */
        for (int i = 1; i <= s.results.size; i++) fin[nfin++] = s.results.h[i];
        qsort(fin, (size_t)nfin, sizeof(int64_t), cmp_key_desc);
        if (nfin > topK) nfin = topK;
        s.reranked = 0;
    } else {
        /* NodeQueue.rerank: rescore entries with approx >= rerankFloor (or only the best one if
         * none qualifies); keep the best topK exact scores.  Boundary ties (equal exact score at
         * the topK edge) resolve by the key order (score desc, node asc): DESIGN.md "Tie rule". */
        int above = 0, best_i = -1;
        float best = -INFINITY;
        for (int i = 1; i <= s.results.size; i++) {
            float a = key_score(s.results.h[i]);
            if (a > best) { best = a; best_i = i; }
            if (a >= rerankFloor) above++;
        }
        for (int i = 1; i <= s.results.size; i++) {
            float a = key_score(s.results.h[i]);
            int take = above > 0 ? (a >= rerankFloor) : (i == best_i);
            if (g_simd_mode && i + 4 <= s.results.size) { /* the row four entries ahead: its first cache lines */
                const char* nx = (const char*)exact_row(ix, key_node(s.results.h[i + 4]));
                for (int o = 0; o < ix->d * 4; o += 256) __builtin_prefetch(nx + o);
            }
            if (!take) continue;
            int node = key_node(s.results.h[i]);
            fin[nfin++] = jvo_encode_key(node, rerank_fn(&s, node));
            s.reranked++;
        }
        qsort(fin, (size_t)nfin, sizeof(int64_t), cmp_key_desc);
        if (nfin > topK) nfin = topK;
    }
    for (int i = 0; i < nfin; i++) {
        int node = key_node(fin[i]);
        if (out_nodes) out_nodes[i] = node;
        if (out_docs) out_docs[i] = ix->ord2doc ? ix->ord2doc[node] : node;
        if (out_scores) out_scores[i] = key_score(fin[i]);
    }
    if (out_count) *out_count = nfin;
    if (out_stats) {
        out_stats[JV_STAT_VISITED] = s.visited_count;
        out_stats[JV_STAT_RERANKED] = s.reranked;
        out_stats[JV_STAT_EXPANDED] = s.expanded;
        out_stats[JV_STAT_EXPANDED_BASE] = s.expanded_base;
    }
    /* hand the (possibly re-allocated) buffers back to the thread's scratch; clear the visited set */
    for (int i = 0; i < s.n_touched; i++) s.visited[s.touched[i]] = 0;
    sc->touched = s.touched;
    sc->cap_touched = s.cap_touched;
    sc->evicted = s.evicted;
    sc->cap_evicted = s.cap_evicted;
    sc->candidates = s.candidates;
    sc->results = s.results;
    return JV_OK;
}

int jvo_search_batch(const jv_index_desc* ix, const float* queries, int32_t nq, int32_t topK,
                     int32_t rerankK, float threshold, float rerankFloor,
                     const uint64_t* accept_doc_words, int64_t accept_num_docs, int32_t* out_nodes,
                     int32_t* out_docs, float* out_scores, int32_t* out_count, int32_t* out_stats,
                     int threads) {
    int used = 1;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
    used = threads;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
#endif
    for (int i = 0; i < nq; i++) {
        jvo_search(ix, queries + (size_t)i * ix->d, topK, rerankK, threshold, rerankFloor,
                   accept_doc_words, accept_num_docs, out_nodes ? out_nodes + (size_t)i * topK : NULL,
                   out_docs ? out_docs + (size_t)i * topK : NULL,
                   out_scores ? out_scores + (size_t)i * topK : NULL, out_count ? out_count + i : NULL,
                   out_stats ? out_stats + (size_t)i * JV_NUM_STATS : NULL);
    }
    return used;
}

void jvo_score_ordinals(const jv_index_desc* ix, const float* query, const int32_t* ordinals,
                        int32_t count, float* out_scores) {
    float qn = ix->similarity == JV_SIM_COSINE ? jvo_raw_dot(query, query, ix->d) : 0.0f;
    for (int i = 0; i < count; i++) {
        int o = ordinals[i];
        if (o < 0 || o >= ix->n) { /* NO_VECTOR_OR_DELETED_DOC -> 0.0f (J/JVectorVectorScorer.java:38-40) */
            out_scores[i] = 0.0f;
            continue;
        }
        float s = exact_unscaled(ix->similarity, query, exact_row(ix, o), ix->d, qn);
        out_scores[i] = ix->score_scale != 1.0f ? s * ix->score_scale : s;
    }
}

void jvo_brute_force(const jv_index_desc* ix, const float* queries, int32_t nq, int32_t k,
                     const uint64_t* accept_doc_words, int32_t* out_nodes, float* out_scores,
                     int threads) {
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
#endif
    for (int qi = 0; qi < nq; qi++) {
        const float* q = queries + (size_t)qi * ix->d;
        float qn = ix->similarity == JV_SIM_COSINE ? jvo_raw_dot(q, q, ix->d) : 0.0f;
        lheap h;
        lh_init(&h, k);
        for (int o = 0; o < ix->n; o++) {
            if (accept_doc_words) {
                int doc = ix->ord2doc ? ix->ord2doc[o] : o;
                if (doc < 0 || !((accept_doc_words[doc >> 6] >> (doc & 63)) & 1)) continue;
            }
            float s = exact_unscaled(ix->similarity, q, exact_row(ix, o), ix->d, qn);
            if (ix->score_scale != 1.0f) s = s * ix->score_scale;
            lh_push_bounded(&h, jvo_encode_key(o, s), k);
        }
        int cnt = h.size;
        for (int i = cnt - 1; i >= 0; i--) {
            int64_t key = lh_pop(&h);
            out_nodes[(size_t)qi * k + i] = key_node(key);
            if (out_scores) out_scores[(size_t)qi * k + i] = key_score(key);
        }
        for (int i = cnt; i < k; i++) {
            out_nodes[(size_t)qi * k + i] = -1;
            if (out_scores) out_scores[(size_t)qi * k + i] = 0.0f;
        }
        lh_free(&h);
    }
}

void jvo_merge_topk(const int32_t* docs, const float* scores, int32_t nq, int32_t lists, int32_t k,
                    int32_t* out_docs, float* out_scores) {
    int total = lists * k;
    int64_t* keys = (int64_t*)malloc(sizeof(int64_t) * (size_t)(total > 0 ? total : 1));
    for (int qi = 0; qi < nq; qi++) {
        int nk = 0;
        for (int i = 0; i < total; i++) {
            int doc = docs[(size_t)qi * total + i];
            if (doc < 0) continue;
            keys[nk++] = jvo_encode_key(doc, scores[(size_t)qi * total + i]);
        }
        qsort(keys, (size_t)nk, sizeof(int64_t), cmp_key_desc);
        for (int i = 0; i < k; i++) {
            out_docs[(size_t)qi * k + i] = i < nk ? key_node(keys[i]) : -1;
            out_scores[(size_t)qi * k + i] = i < nk ? key_score(keys[i]) : 0.0f;
        }
    }
    free(keys);
}
