/*
 * jv_oracle.h — CPU restatement (ORACLE) of the jVector GraphSearcher hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing shipped may import, link or call this: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as the checker /
 * the reported CPU baseline.  The product path is the HIP library behind include/jvgpu.h.
 *
 * PARITY STATUS: "parity unpinned" for traversal order, visited/expanded counts, tie-breaks and
 * PQ values: the arithmetic lives in the third-party library io.github.jbellis:jvector:4.0.0-rc.9
 * (reference build.gradle:362, gradle.properties:9), which is neither vendored in the reference
 * nor runnable here (no JVM).  This file restates that library's published algorithm (SURVEY.md
 * Appendix A) and is pinned against every known-answer case the reference's own tests hold for
 * this path (tests/golden/ka_cases.json, from KNNJVectorTests.java et al.).
 */
#ifndef JV_ORACLE_H
#define JV_ORACLE_H

#include "../include/jvgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* --- NodeQueue key (jvector NodeQueue.encode) --- */
int32_t jvo_float_to_sortable_int(float f);
float jvo_sortable_int_to_float(int32_t s);
int64_t jvo_encode_key(int32_t node, float score);

/* --- canonical fp32 accumulations (DESIGN.md "Canonical arithmetic") ---
 * 64 strided partial sums P[m] = sum_j fma(a[64j+m], b[64j+m]) followed by an adjacent-pair
 * binary tree; the HIP kernels use the identical order so scores are bit-equal. */
float jvo_raw_dot(const float* a, const float* b, int d);
float jvo_raw_l2(const float* a, const float* b, int d);
/* exact similarity score incl. jVector's mapping and score_scale
 * (VectorSimilarityFunction.compare; J/JVectorReader.java:220-239, J/JVectorVectorScorer.java:36-53) */
float jvo_exact_score(int sim, float score_scale, const float* q, const float* v, int d);

/* --- PQ (jvector PQVectors.precomputedScoreFunctionFor / PQDecoder) --- */
/* lut: [M][256] floats; norm_lut (cosine only, may be NULL otherwise): [M][256] */
void jvo_pq_sub_layout(int d, int M, const int32_t* sub_sizes, int32_t* sizes, int32_t* offsets);
void jvo_pq_build_lut(const jv_index_desc* ix, const float* q, float* lut);
void jvo_pq_build_norm_lut(const jv_index_desc* ix, float* norm_lut);
float jvo_pq_raw(const float* lut, const uint8_t* code, int M);
/* 0 = plain C loops, 1 = explicit AVX2 gathers + software prefetch; results are identical in both modes */
void jvo_set_simd(int mode);
int jvo_get_simd(void);
float jvo_pq_score(const jv_index_desc* ix, const float* lut, const float* norm_lut, float qnorm2, int node);

/* --- NVQ-inline vectors: nvqDequantize restated (J/JVectorIndexQuantization.java:319-361); out has d floats --- */
void jvo_nvq_dequantize(const jv_index_desc* ix, int node, float* out);

/* --- the search (GraphSearcher.search; call site J/JVectorReader.java:165-173) ---
 * Same contract as jv_search in include/jvgpu.h. Returns JV_OK / JV_EINVAL. */
int jvo_search(const jv_index_desc* ix, const float* query, int32_t topK, int32_t rerankK,
               float threshold, float rerankFloor, const uint64_t* accept_doc_words,
               int64_t accept_num_docs, int32_t* out_nodes, int32_t* out_docs, float* out_scores,
               int32_t* out_count, int32_t* out_stats);

/* nq searches, OpenMP over queries with `threads` threads (<=0: all cores). One query per
 * thread, like Lucene's one-thread-per-leaf-search model. Returns threads actually used. */
int jvo_search_batch(const jv_index_desc* ix, const float* queries, int32_t nq, int32_t topK,
                     int32_t rerankK, float threshold, float rerankFloor,
                     const uint64_t* accept_doc_words, int64_t accept_num_docs, int32_t* out_nodes,
                     int32_t* out_docs, float* out_scores, int32_t* out_count, int32_t* out_stats,
                     int threads);

/* bytes from src to dst with all OpenMP threads, 2 MiB per chunk: used by bench.py's cpu_baseline so that the host copy
 * of the index is FIRST TOUCHED by threads on every NUMA node (a single-threaded copy puts the whole index behind one
 * socket's memory controllers and starves the other socket's searcher threads). */
void jvo_parallel_copy(void* dst, const void* src, size_t bytes);

/* exact scorer over an ordinal list (JVectorVectorScorer.score, J/JVectorVectorScorer.java:36-53) */
void jvo_score_ordinals(const jv_index_desc* ix, const float* query, const int32_t* ordinals,
                        int32_t count, float* out_scores);

/* brute-force top-k by exact score over all (accepted) ordinals: ground truth for recall
 * (same definition as the reference harness: F/TestUtils.java:185-200,
 *  scripts/jvector_index_and_search/jvector_utils/recall_measurement.py:48-108). */
void jvo_brute_force(const jv_index_desc* ix, const float* queries, int32_t nq, int32_t k,
                     const uint64_t* accept_doc_words, int32_t* out_nodes, float* out_scores,
                     int threads);

/* k-way merge of per-shard top-k lists by (score desc, doc asc) — TopDocs.merge semantic. */
void jvo_merge_topk(const int32_t* docs, const float* scores, int32_t nq, int32_t lists, int32_t k,
                    int32_t* out_docs, float* out_scores);

#ifdef __cplusplus
}
#endif
#endif
