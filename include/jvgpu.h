/*
 * jvgpu.h — C ABI of the MI355X-native jVector graph-search engine.
 *
 * This is the drop-in boundary for ONE path of opensearch-project/opensearch-jvector:
 *   JVectorKnnFloatVectorQuery.approximateSearch -> JVectorReader.search -> jvector GraphSearcher.search
 * Every entry point below states the reference call site it replaces (paths relative to the
 * reference repo root; J/ = src/main/java/org/opensearch/knn/index/codec/jvector/).
 *
 * Plain C: pointers + sizes only, no C++/torch types.  All functions return 0 (JV_OK) or a
 * negative jv_status; the message for the calling thread's last failure is jv_last_error().
 * No function aborts the process.
 *
 * Threading: jv_search / jv_search_batch* are re-entrant on one handle (the reference calls
 * JVectorReader.search concurrently from Lucene's per-leaf search threads:
 * src/test/java/org/opensearch/knn/index/codec/jvector/KNNJVectorTests.java:982-1027).
 * jv_index_create / jv_index_destroy are single-threaded per handle, like the reference's
 * FieldEntry constructor / close (J/JVectorReader.java:284-337, :367-378).
 */
#ifndef JVGPU_H
#define JVGPU_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JVGPU_ABI_VERSION 4

/* ---- status codes (the Java shim maps them to the reference's exception types,
 *      SURVEY §8(b) "Errors": EINVAL -> IllegalArgumentException, EUNSUPPORTED ->
 *      UnsupportedOperationException, EDEVICE/ENOMEM -> IOException) ---- */
typedef enum jv_status {
    JV_OK = 0,
    JV_EINVAL = -1,
    JV_ENOMEM = -2,
    JV_EDEVICE = -3,
    JV_EUNSUPPORTED = -4,
    JV_EINTERNAL = -5
} jv_status;

/* ---- similarity: jvector VectorSimilarityFunction ordinals as mapped by
 *      VectorSimilarityMapper (J/JVectorReader.java:384-432): Lucene EUCLIDEAN->0,
 *      DOT_PRODUCT->1, COSINE->2, MAXIMUM_INNER_PRODUCT->1 (with score_scale = 2 when the
 *      field has no PQ, J/JVectorReader.java:220-239,359-364). ---- */
typedef enum jv_similarity {
    JV_SIM_EUCLIDEAN = 0,   /* score = 1 / (1 + sum (a-b)^2)            */
    JV_SIM_DOT_PRODUCT = 1, /* score = (1 + sum a*b) / 2                */
    JV_SIM_COSINE = 2       /* score = (1 + dot / sqrt(|a|^2 |b|^2)) / 2 */
} jv_similarity;

/* ---- descriptor flags ---- */
#define JV_DESC_DEVICE_POINTERS 0x1u /* vectors/adj/pq_codes/ord2doc already live in HBM on `device` */
#define JV_DESC_BORROW          0x2u /* with DEVICE_POINTERS: do not copy, caller keeps them alive   */
#define JV_DESC_FUSED_ADC       0x4u /* also build the fused layout: per node, its neighbours' PQ codes
                                        stored next to its adjacency row (a layout choice; scores are
                                        bit-identical to the plain layout)                            */

#define JV_DESC_BUILD_CLIENT    0x8u /* handle used by an index builder: identical search, launched under a
                                        separate kernel name so profiles keep build and query launches apart */

/* One upper layer of a hierarchical graph (hierarchy_enabled; default false:
 * K/common/KNNConstants.java:109).  Layer 0 is the dense adj[n][R] array below. */
typedef struct jv_layer_desc {
    int32_t count;        /* nodes present in this layer                                   */
    int32_t degree;       /* row stride of `adj`                                           */
    const int32_t* nodes; /* [count] ordinals, strictly ascending                          */
    const int32_t* adj;   /* [count][degree] neighbour ordinals, -1 padded                 */
} jv_layer_desc;

/* Everything FieldEntry holds after OnDiskGraphIndex.load + PQVectors.load
 * (J/JVectorReader.java:284-337; J/JVectorIndexQuantization.java:68-89,166-186),
 * flattened.  The library copies what it needs to HBM before jv_index_create returns
 * (unless JV_DESC_BORROW); the caller may free its arrays immediately. */
typedef struct jv_index_desc {
    uint32_t struct_size; /* sizeof(jv_index_desc), for ABI evolution                      */
    uint32_t flags;       /* JV_DESC_*                                                     */
    int32_t device;       /* HIP device ordinal                                            */
    int32_t n;            /* graph nodes (ordinals 0..n-1)                                 */
    int32_t d;            /* vector dimension                                              */
    int32_t R;            /* layer-0 max degree = row stride of adj                        */
    int32_t similarity;   /* jv_similarity                                                 */
    float score_scale;    /* 1, or 2 for Lucene MAXIMUM_INNER_PRODUCT without PQ           */
    int32_t entry_node;   /* view.entryNode().node; -1 => empty index                      */
    int32_t num_upper_layers;          /* 0 when hierarchy is off                          */
    const jv_layer_desc* upper_layers; /* [num_upper_layers], index 0 = layer 1            */
    const float* vectors;   /* [n][d] inline full-precision vectors                        */
    const int32_t* adj;     /* [n][R] neighbour ordinals in stored order, -1 padded        */
    /* product quantisation (pq_M == 0 => exact-only index) */
    int32_t pq_M;                 /* subspaces                                             */
    int32_t pq_K;                 /* clusters per subspace, <= 256                         */
    const int32_t* pq_sub_sizes;  /* [pq_M] or NULL => jvector's even split (d/M, first d%M get +1) */
    const float* pq_codebooks;    /* concat over m of [pq_K][sub_size[m]]                  */
    const float* pq_centroid;     /* [d] global centroid or NULL                           */
    const uint8_t* pq_codes;      /* [n][pq_M]                                             */
    /* ordinal -> Lucene doc id (GraphNodeIdToDocMap, J/GraphNodeIdToDocMap.java:147-161) */
    const int32_t* ord2doc;       /* [n] or NULL => identity                               */
    int32_t max_doc;              /* doc-id space size (for accept bitsets)                */
    int32_t reserved;
    /* NVQ-inline vectors (quantType 2, J/JVectorReader.java:357-358; decode = J/JVectorIndexQuantization.java:306-361):
     * when nvq_M > 0 every "exact" score (the rerank, the exact provider of an NVQ-only field, jv_score_ordinals) is
     * taken against the DEQUANTISED vector
     *     x[i] = logitNQT(fma(byte[i], logisticScale_s, logisticBias_s), 1 / scaledGrowthRate_s, scaledMidpoint_s) + globalMean[i]
     * of the node's 8-bit NVQ record instead of a full-precision row, and `vectors` may be NULL (4x fewer rerank bytes). */
    int32_t nvq_M;                /* NVQ subvectors per vector (0 => no NVQ)                                  */
    int32_t reserved2;
    const int32_t* nvq_sub_sizes; /* [nvq_M] or NULL => jvector's even split (d/M, first d%M get +1)          */
    const float* nvq_params;      /* [n][nvq_M][4]: growthRate, midpoint, minValue, maxValue per subvector    */
    const uint8_t* nvq_bytes;     /* [n][d] quantised components                                              */
    const float* nvq_global_mean; /* [d] subtracted before encoding, added back after decoding                */
} jv_index_desc;

typedef struct jv_index jv_index; /* opaque handle */

/* Per-query counters, in the order the reference reads them from SearchResult
 * (J/JVectorReader.java:183-187) and adds to KNNCounter (:189-192). */
enum { JV_STAT_VISITED = 0, JV_STAT_RERANKED = 1, JV_STAT_EXPANDED = 2, JV_STAT_EXPANDED_BASE = 3, JV_NUM_STATS = 4 };

/* Extended per-query status written to out_flags by the batch calls. */
#define JV_QFLAG_RETRIED_BIG 0x1 /* on-chip scratch overflowed; query was re-run on the HBM-scratch variant */
#define JV_QFLAG_EARLY_TERMINATED 0x2 /* visit_limit reached: the search stopped and returned nothing; out_stats holds the
                                         counters so far (Lucene then runs the exact scan, see jv_search_params.visit_limit) */

/* Replaces: FieldEntry constructor (J/JVectorReader.java:284-337) — once per segment x field. */
int jv_index_create(const jv_index_desc* desc, jv_index** out);

/* Replaces: FieldEntry.close (J/JVectorReader.java:367-378).  NULL is a no-op. */
void jv_index_destroy(jv_index* index);

/* Replaces: the body of JVectorReader.search, J/JVectorReader.java:147-177, i.e.
 *   buildScoreFunctionProvider(q, view)  (:152, :352-365)
 *   acceptOrds lambda                    (:157-163)  -> accept_doc_words (doc-space bitset, bit = doc id;
 *                                                       NULL => accept all, like `acceptDocs == null`)
 *   GraphSearcher.search(ssp, topK, rerankK, threshold, rerankFloor, acceptOrds)  (:165-173)
 * Outputs: up to topK (ordinal, score) pairs in descending score order, ties by ascending
 * ordinal; out_docs (optional) = ord2doc[ordinal] as the reference's collect loop does (:175-177);
 * out_stats[JV_NUM_STATS] = visited, reranked, expanded, expandedBaseLayer (:183-187).
 * rerankK < topK -> JV_EINVAL (jvector throws IllegalArgumentException).
 *
 * Concurrency: the reference issues ONE query per call from many searcher threads
 * (T/index/engine/JVectorConcurrentQueryTests.java:78-138).  Calls that are in flight at the same time on
 * one handle are combined inside the library into batch launches (group commit: the caller that finds a
 * free leader slot runs every queued call with the same topK / rerankK / threshold / rerankFloor — its own
 * included — as one launch, each query with its OWN filter, and hands the answers back; later arrivals form
 * the next batch).  Semantics per call are unchanged (same ids, scores and counters as a lone call); a lone
 * caller pays no extra latency.  Options "combine" (1), "combine_leaders" (2 batches in flight),
 * "combine_max_batch" (2048). */
int jv_search(jv_index* index, const float* query, int32_t topK, int32_t rerankK, float threshold,
              float rerankFloor, const uint64_t* accept_doc_words, int64_t accept_num_docs,
              int32_t* out_nodes, int32_t* out_docs, float* out_scores, int32_t* out_count,
              int32_t* out_stats);

/* Optional search parameters beyond the reference's call (jv_search_ex / jv_search_batch_ex). */
typedef struct jv_search_params {
    uint32_t struct_size;  /* sizeof(jv_search_params) */
    int32_t topK, rerankK;
    float threshold, rerankFloor;
    const uint64_t* accept_doc_words; /* doc-space bitset or NULL */
    int64_t accept_num_docs;
    /* Lucene's KnnCollector.visitLimit() (= the filter's cardinality for a filtered query).  The reference never checks it
     * while searching (J/JVectorReader.java:202-207 only reports visited + expanded afterwards) and
     * AbstractKnnVectorQuery then DISCARDS the approximate result when that sum reached the limit and runs the exact
     * scan (-> jv_score_ordinals).  With visit_limit > 0 the engine stops such a search as soon as visited + expanded
     * reaches the limit and sets JV_QFLAG_EARLY_TERMINATED instead of finishing work that is going to be thrown away;
     * searches that stay below the limit are unchanged (same ids, scores, counters).  0 = never stop early.
     * The fused-PQ pool kernels only know `expanded` while searching (they count `visited` afterwards): they stop once the
     * expansions alone reach the limit, and a search that finished but whose visited + expanded reaches it is flagged
     * afterwards — either way every flagged row is empty and every unflagged row has visited + expanded < visit_limit. */
    int64_t visit_limit;
    /* Identity of the filter's CONTENTS for the device-side filter cache (option "filter_cache", per index): a bitset
     * that was uploaded before is served from HBM instead of crossing PCIe again.  0 = the library hashes the words.
     * The key (or hash) only FINDS a cached copy: it is served after a byte-for-byte comparison with the caller's words,
     * so a colliding hash or a re-used key can never apply another filter's bits (they carry deletes and doc-level security). */
    uint64_t accept_key;
} jv_search_params;

/* jv_search with jv_search_params; out_flags (optional) receives the query's JV_QFLAG_* word. */
int jv_search_ex(jv_index* index, const float* query, const jv_search_params* params, int32_t* out_nodes,
                 int32_t* out_docs, float* out_scores, int32_t* out_count, int32_t* out_stats, int32_t* out_flags);

/* jv_search_batch with jv_search_params and PER-QUERY status: out_status[i] = JV_OK or the reason query i has no answer
 * (JV_ENOMEM: it outgrew even the HBM scratch); the call returns JV_OK only if every query succeeded, but the rows of
 * the successful queries are valid either way — one pathological query does not fail its batch. */
int jv_search_batch_ex(jv_index* index, const float* queries, int32_t nq, const jv_search_params* params,
                       int32_t* out_nodes, int32_t* out_docs, float* out_scores, int32_t* out_count,
                       int32_t* out_stats, int32_t* out_status, int32_t* out_flags);

/* nq independent searches with shared parameters (the GPU's natural unit of work; the reference
 * has no batch call — Lucene issues one search per leaf per thread).  Row-major outputs:
 * out_nodes/out_docs/out_scores [nq][topK], out_count [nq], out_stats [nq][JV_NUM_STATS].
 * Unused tail entries of a row are ordinal/doc -1 and score 0.  Host pointers.
 * Returns JV_ENOMEM if any query could not be answered (see jv_search_batch_ex for per-query status); the rows of the
 * other queries are valid. */
int jv_search_batch(jv_index* index, const float* queries, int32_t nq, int32_t topK, int32_t rerankK,
                    float threshold, float rerankFloor, const uint64_t* accept_doc_words,
                    int64_t accept_num_docs, int32_t* out_nodes, int32_t* out_docs, float* out_scores,
                    int32_t* out_count, int32_t* out_stats);

/* Same, but every pointer is a DEVICE pointer on the index's device and the work is enqueued on
 * `hip_stream` (a hipStream_t passed as void*; NULL = the library's own stream, synchronous, ordered behind what the legacy
 * default stream — handle 0 — has in flight at call time, NOT behind other streams' work).
 * With a caller stream the call returns after enqueueing; results are valid after the caller
 * synchronises that stream.  The whole ladder (on-chip kernels, retry, HBM-scratch rung) is enqueued without host
 * round trips; d_out_flags (strongly recommended) receives per query JV_QFLAG_* in the low bits and 0x40000000 if the
 * query exhausted even the HBM scratch (its row is then empty) — without d_out_flags such a failure is invisible.
 * Calls on DIFFERENT streams run side by side, each in a launch context (scratch, counters) of its own, up to the index's
 * "async_contexts" option (default 4); further streams share a context and are ordered behind its previous use. */
int jv_search_batch_device(jv_index* index, const float* d_queries, int32_t nq, int32_t topK,
                           int32_t rerankK, float threshold, float rerankFloor,
                           const uint64_t* d_accept_doc_words, int64_t accept_num_docs,
                           int32_t* d_out_nodes, int32_t* d_out_docs, float* d_out_scores,
                           int32_t* d_out_count, int32_t* d_out_stats, int32_t* d_out_flags,
                           void* hip_stream);

/* Replaces (next-scope row f1): JVectorVectorScorer.score (J/JVectorVectorScorer.java:36-53) — the
 * exact scorer Lucene falls back to for selective filters: score `count` ordinals against one query.
 * Scores carry score_scale exactly like the search path. Host pointers. */
int jv_score_ordinals(jv_index* index, const float* query, const int32_t* ordinals, int32_t count,
                      float* out_scores);

/* ---- the exact scorer for a BATCH of queries that share one candidate set (ABI v4) ----
 * Replaces: Lucene's exact fallback as the reference wires it — AbstractKnnVectorQuery.exactSearch iterating the accepted
 * docs through JVectorFloatVectorValues.scorer(target) (J/JVectorFloatVectorValues.java:189-191) and
 * JVectorVectorScorer.score (J/JVectorVectorScorer.java:36-53) into a HitQueue of k — for nq queries that run under the SAME
 * acceptDocs (one tenant / ACL / facet filter, the case a filtered k-NN query takes whenever its filter is selective:
 * J/JVectorReader.java:202-207 reports visited + expanded and Lucene discards a graph search that reached the filter's
 * cardinality).  The candidate set is, in this order of precedence:
 *   accept_doc_words  a doc-space bitset (bit = Lucene doc id): every ordinal with ord2doc[ord] >= 0 and its bit set;
 *   ordinals / count  an explicit ordinal list (entries < 0 or >= n, and ordinals whose doc is deleted, are skipped);
 *   neither           every live ordinal of the index (brute force).
 * Per query the call returns the topK best candidates by (score desc, doc asc) — Lucene's HitQueue order — with EXACTLY
 * the scores jv_score_ordinals returns (canonical fp32 accumulation, score_scale applied): out_nodes / out_docs / out_scores
 * [nq][topK], unused tail = -1 / -1 / 0, out_count [nq].  On the way a bf16 matrix-core pass over a bf16 mirror of the
 * vectors (built by the first call, n * (2 * roundup(d, 64) + 4) bytes of HBM; when that does not fit, or the field is
 * NVQ-only, the call scans the list in fp32 instead) only DISCARDS candidates that provably cannot reach the top k; the
 * survivors are re-scored in fp32.  Results never depend on that pass (flag JV_XB_NO_PREFILTER runs without it).
 * out_info (optional, host, JV_XB_INFO_WORDS words): [0] candidates, [1] sample rows of the bound pass (0 = no pre-filter
 * ran), [2] rows re-scored in fp32 summed over the queries, [3] queries whose survivor list overflowed (they scanned the
 * whole list).  topK <= JV_XB_TOPK_MAX (JV_EUNSUPPORTED beyond: score the ordinals with jv_score_ordinals). */
#define JV_XB_NO_PREFILTER 0x1u
#define JV_XB_FORCE_PREFILTER 0x2u /* diagnostics: run the matrix-core pass for lists shorter than 2 048 entries as well (tests, fuzz) */
#define JV_XB_TOPK_MAX 1024
#define JV_XB_INFO_WORDS 4
typedef struct jv_exact_batch_params {
    uint32_t struct_size;              /* sizeof(jv_exact_batch_params) */
    int32_t topK;
    const uint64_t* accept_doc_words;  /* shared doc filter or NULL */
    int64_t accept_num_docs;
    const int32_t* ordinals;           /* shared ordinal list (used when accept_doc_words is NULL) or NULL */
    int32_t count;
    uint32_t flags;                    /* JV_XB_* */
    uint64_t accept_key;               /* filter-cache key of accept_doc_words, 0 = hash (see jv_search_params) */
} jv_exact_batch_params;
/* Host pointers (queries, the params' arrays, outputs); synchronous. */
int jv_score_ordinals_batch(jv_index* index, const float* queries, int32_t nq, const jv_exact_batch_params* params,
                            int32_t* out_nodes, int32_t* out_docs, float* out_scores, int32_t* out_count, int64_t* out_info);
/* DEVICE pointers on the index's device (queries, the params' arrays, outputs; out_info stays a host pointer).  The work is
 * ordered behind what `hip_stream` has enqueued at call time and that stream continues behind it; the call itself returns
 * after enqueueing unless a doc filter (its cardinality sizes the launch) or out_info asks for a host round trip.  NULL
 * stream = synchronous. */
int jv_score_ordinals_batch_device(jv_index* index, const float* d_queries, int32_t nq, const jv_exact_batch_params* params,
                                   int32_t* d_out_nodes, int32_t* d_out_docs, float* d_out_scores, int32_t* d_out_count,
                                   int64_t* out_info, void* hip_stream);

/* The same for ONE query — what Lucene's exactSearch is per leaf and query, issued the way the reference issues it: one call per
 * searcher thread.  Calls in flight at the same time on one handle that carry the SAME doc filter (same accept_key or content hash,
 * same length, bits compared) and the same topK are combined inside the library into one jv_score_ordinals_batch call (group
 * commit, as jv_search does); per-call semantics are unchanged and a lone caller pays no delay.  Counters "exact_calls" /
 * "exact_batches" (jv_index_get_counter) show the combining.  An explicit ordinal list (accept_doc_words == NULL) is answered alone. */
int jv_exact_search(jv_index* index, const float* query, const jv_exact_batch_params* params, int32_t* out_nodes,
                    int32_t* out_docs, float* out_scores, int32_t* out_count);

/* Merge per-shard top-k lists (the step Lucene's TopDocs.merge performs over leaves, and the
 * exchange step of the doc-range sharded multi-GPU layout): `lists` rows of `k` (doc, score)
 * pairs each (doc < 0 = empty slot) -> best k by (score desc, doc asc).  DEVICE pointers,
 * nq independent merges. */
int jv_merge_topk_device(int32_t device, const int32_t* d_docs, const float* d_scores, int32_t nq,
                         int32_t lists, int32_t k, int32_t* d_out_docs, float* d_out_scores,
                         void* hip_stream);

/* Introspection used by the benchmark's roofline accounting. */
typedef struct jv_index_info {
    int32_t n, d, R, similarity, pq_M, pq_K, num_upper_layers, device;
    int64_t hbm_bytes;          /* bytes resident in HBM for this index          */
    int32_t row_stride_floats;  /* padded vector row stride                       */
    int32_t fused_adc;          /* 1 if the fused layout is present               */
    int64_t scratch_bytes;      /* HBM scratch currently held on behalf of this index: the batched exact scorer's bf16 mirror (once built), launch contexts (staging, spill
                                   tables, expansion logs), cached filters, and the device's SHARED HBM-scratch rung    */
    int64_t filter_cache_hits, filter_cache_misses;
} jv_index_info;
int jv_index_get_info(const jv_index* index, jv_index_info* out);
/* Diagnostics: launches per kernel family since the handle was created — "launches_pqw" (several-waves-per-query pool
 * kernel), "launches_pqp" (one-wave pool kernel), "launches_pqf" (round-1 fused-PQ kernel), "launches_lds" (generic LDS
 * kernel).  Lets a test or an operator see which rung served a workload.  JV_EINVAL for unknown names.
 * "retry_rungs_skipped": redo launches that host-pointer batch calls left out because their workgroups would not have fitted beside a
 * live query-server grid (the flagged rows of such a call take the HBM-scratch rung instead).
 * "launches_serve" / "served_queries": starts of the device-resident query-server grid and one-query calls it answered;
 * "serve_alive": how many of this handle's resident grids are running right now (0 once the grid has idled out after
 * option "serve_idle_ms" without a query — a test can wait for that instead of sleeping).
 * Measurement: with option "time_search_kernel" = 1 the library records HIP events around the first (main) search launch of every
 * batch call on the stream it launches on; "search_kernel_ns" / "search_kernel_timed" are their sum and number (reading them waits for
 * the timed launches still in flight).  bench.py reports that duration beside the whole call's (its roofline fraction is on the whole call). */
int jv_index_get_counter(const jv_index* index, const char* name, int64_t* out);

/* Tunables are PER INDEX: jv_index_set_option changes one handle; jv_set_option only changes the defaults that indexes
 * created afterwards start from (nothing process-wide is read at call time).  Names: "lds_visited_slots",
 * "lds_candidates", "force_big_path", "force_general_path", "big_blocks", "big_cand_cap", "big_budget_mb",
 * "spill_tables", "spill_slots", "combine", "combine_leaders", "combine_max_batch", "max_contexts", "async_contexts", "filter_cache",
 * "serve", "serve_wgs_per_cu", "serve_idle_ms", "direct_completion", "lazy_big_rung"
 * (+ diagnostics: "no_escalation", "no_pqf", "no_pqp", "no_pqw", "pqw_min_queries", "no_lutr", "lutr_min_queries",
 * "pqf_only", "dbg_ptr").
 * JV_EINVAL for unknown names. */
int jv_set_option(const char* name, int64_t value);
int jv_index_set_option(jv_index* index, const char* name, int64_t value);

/* ---- doc-ID-range sharding inside ONE process (the reference's host is one JVM per node) ----
 * A shard group bundles the per-shard handles of one logical field: shard g owns a contiguous doc-id range, with its own
 * graph and an ord2doc map of GLOBAL doc ids, on any device of the node.  jv_search_sharded_batch runs every query on every
 * shard (concurrently, one stream per shard), gathers the per-shard top-k lists of (doc, score) onto the first shard's
 * device with peer-to-peer copies over xGMI (one 8-byte-pair buffer per shard), and merges them there
 * (jv_merge_topk kernel: score desc, doc asc) — Lucene's per-leaf search + TopDocs.merge in one call.  out_docs are
 * global doc ids; out_stats are summed over the shards.  The group borrows the handles (destroy the group first). */
typedef struct jv_shard_group jv_shard_group;
int jv_shard_group_create(jv_index* const* shards, int32_t num_shards, jv_shard_group** out);
void jv_shard_group_destroy(jv_shard_group* group);
int jv_search_sharded_batch(jv_shard_group* group, const float* queries, int32_t nq, int32_t topK, int32_t rerankK,
                            float threshold, float rerankFloor, int32_t* out_docs, float* out_scores,
                            int32_t* out_count, int32_t* out_stats);

/* Group options: "gather" = 0 (default) moves every shard's (doc, score) lists to the first shard's device with one
 * peer-to-peer copy per shard; = 1 gathers them with ONE RCCL all-gather over xGMI (communicators owned by the group,
 * librccl opened at run time; every shard must sit on its own device).  The merged answers are identical. */
int jv_shard_group_set_option(jv_shard_group* group, const char* name, int64_t value);

/* The same with jv_search_params: the doc filter is a bitset over the GLOBAL doc-id space of the group (every shard's
 * ord2doc maps into it; the reference hands every leaf search its acceptDocs, J/JVectorReader.java:157-163), visit_limit
 * applies to every shard's search on its own, out_status[i] (optional) is JV_OK / JV_ENOMEM per query, out_flags[i] (optional)
 * the OR of the shards' JV_QFLAG_* words (EARLY_TERMINATED: at least one shard stopped at the visit limit and contributed
 * nothing).  On any error every stream of the group is drained before the call returns. */
int jv_search_sharded_batch_ex(jv_shard_group* group, const float* queries, int32_t nq, const jv_search_params* params,
                               int32_t* out_docs, float* out_scores, int32_t* out_count, int32_t* out_stats,
                               int32_t* out_status, int32_t* out_flags);

/* Thread-local message of the calling thread's most recent failing call ("" if none). */
const char* jv_last_error(void);

int jv_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* JVGPU_H */
