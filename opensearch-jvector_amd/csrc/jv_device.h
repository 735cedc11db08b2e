// jv_device.h — structures shared between the C-ABI host code (jv_abi.cpp) and the HIP kernels
// (jv_kernels.hip).  gfx950 only.
#pragma once
#include <stdint.h>

#define JV_WAVE 64
#define JV_MAX_UPPER_LAYERS 16
#define JV_TODO 64 /* neighbours scored per adjacency chunk */
#define JV_TRACKER_LDS (512 * 4 + 2 * (100 + 64 + 4) * 4) /* threshold-query tracker state (bytes) */

struct JvLayerDev {
    int32_t count;
    int32_t degree;
    const int32_t* nodes;
    const int32_t* adj;
};

// The HBM-resident index: flat arrays, nothing else (DESIGN.md "Data layout in HBM").
struct JvIndexDev {
    int32_t n, d, R;
    int32_t stride;       // floats per vector row, = roundup(d, 4): rows are 16-B aligned, zero padded
    int32_t nch;          // number of 64-float chunks per row = ceil(stride / 64)
    int32_t sim;          // jv_similarity
    float score_scale;    // Lucene MIP fix-up (exact-provider search path and exact scorer only)
    int32_t entry;
    int32_t num_upper;
    JvLayerDev upper[JV_MAX_UPPER_LAYERS];
    const float* vectors;    // [n][stride]
    const int32_t* adj;      // [n][R]
    const int32_t* ord2doc;  // [n] or nullptr (identity)
    // PQ
    int32_t pq_M, pq_K;
    int32_t pq_lanes;        // lanes per node in ADC scoring = next_pow2(ceil(M/16))
    const float* pq_cbT;     // [d][256]: codebook transposed, entry (dim, c) = codebook[m(dim)][c][dim - off(m)]
    const int32_t* pq_sub_off;  // [M+1] first dimension of each subspace
    int32_t pq_code_stride;     // bytes per code row = roundup(M, 16)
    const float* pq_centroid;   // [d] or nullptr
    const uint8_t* pq_codes;    // [n][M]
    const float* pq_norm_lut;   // [M][256] |centroid|^2 (cosine) or nullptr
    const uint8_t* pq_fused;    // fused layout: [n][R][M] neighbours' codes next to the adjacency order, or nullptr
    // cosine on the fused layout (round 6): |decoded code vector|^2 = the canonical sum over the norm table's M entries of a node's
    // code row — a property of the NODE, not of the query, so it is summed once at index creation (same chunks, same pair tree:
    // bit-equal to jvo_pq_raw over the norm table) and read by the pool wave next to the neighbour's ordinal
    const float* pq_node_norm;  // [n] or nullptr
    const float* pq_fused_norm; // [n][R]: pq_node_norm[adj[u][j]] (0 for -1) or nullptr
    // NVQ-inline vectors (exact scores against the dequantised record; `vectors` may be nullptr then)
    int32_t nvq_M;
    int32_t nvq_stride;         // bytes per row = roundup(d, 4)
    const int32_t* nvq_sub_off; // [nvq_M + 1]
    const float* nvq_params;    // [n][nvq_M][4]: growthRate, midpoint, minValue, maxValue
    const uint8_t* nvq_bytes;   // [n][nvq_stride], zero padded
    const float* nvq_mean;      // [nch * 64] global mean, zero padded
};
#define JV_NVQ_MAX_M 8

struct JvSearchArgs {
    const float* queries;  // [nq][d]
    const int32_t* qlist;  // optional indirection (big-path retry list) or nullptr
    int32_t nq;
    int32_t topK, rk;
    float threshold, rerank_floor;
    const uint64_t* accept;  // doc-space bitset or nullptr
    int64_t accept_docs;
    int64_t accept_stride;   // 64-bit words between consecutive queries' bitsets; 0 = one bitset for the whole batch
    const uint64_t* accept_ord;  // optional: the SHARED filter translated to ordinal space ([ceil(n/64)] words; jvk_launch_accept_to_ord) — one load per neighbour instead of ord -> doc -> word
    int32_t* out_nodes;
    int32_t* out_docs;
    float* out_scores;
    int32_t* out_count;
    int32_t* out_stats;
    int32_t* out_flags;
    // on-chip scratch geometry (bytes offsets into dynamic LDS)
    int32_t hash_slots;  // power of two (LDS path)
    int32_t cand_cap;
    int32_t res_cap;     // >= rk
    // HBM scratch (big path): per resident block
    uint32_t* big_visited;   // [blocks][ceil(n/32)]
    int64_t* big_cand;       // [blocks][big_cand_cap]
    int32_t big_cand_cap;
    int32_t* work_counter;   // big path: dynamic query dequeue
    float* lut_scratch;      // big path, PQ tables too large for LDS (the reference's default 192 subspaces at d >= 768): [blocks][pq_M][256] in HBM
    int32_t visit_limit;     // > 0: stop (flag EARLY) once visited + expanded reaches it (Lucene KnnCollector.visitLimit)
    int32_t retry_only;      // later launches: 1 = walk the flag array and redo pool/log overflows only; 2 = the same, last on-chip rung (filtered: never skips on the selectivity estimate)
    int32_t no_prescore;     // diagnostics (option "no_prescore"): the latency variant's helper wave does not pre-score the pair requested ahead (jv_pqw_body.h PRE)
    int32_t* retry_counter;  // escalation launches: flag-chunk dequeue counter (zeroed per call, one per rung)
    // two-level visited set: pool of per-query spill tables in HBM (taken with spill_counter)
    uint32_t* spill;
    int32_t spill_slots;     // entries per table (power of two)
    int32_t spill_tables;
    int32_t* spill_counter;  // zeroed per call
    int64_t* dbg;            // diagnostic build (-DJV_STAMPS) only: 8 cycle accumulators; nullptr in the product
    // persistent pool kernel (jv_pqp_body.h)
    int32_t* pqp_log;        // [blocks][pqp_log_cap] expansion log scratch
    int32_t pqp_log_cap;
    int32_t pqp_qc_off;      // LDS byte offset of the centred query during the LUT build
    int32_t pqp_pool_off;    // LDS byte offset of the pool (jv_kernels_pqp.hip)
    int32_t pqp_scratch_off; // LDS byte offset of the 768-byte merge scratch (kept keys + ranks of one expansion)
    int32_t pqp_lds_bytes;   // dynamic LDS bytes of the launch (register-LUT variant: the visited-count hash set uses all of it)
    int32_t* pqp_counter;    // query dequeue counter (zeroed per call)
    // device-resident query server (jv_pqw_body.h, SERVE instances): single queries arrive through a ring of slots in pinned
    // host memory; the workgroups claim tickets [head, published) and answer into the slot
    unsigned char* serve_ring;   // [serve_slots][serve_slot_bytes], host-visible
    int32_t serve_slots;         // power of two
    int32_t serve_slot_bytes;
    int32_t* serve_dev;          // device words: JV_SV_* below
    int32_t* serve_host;         // pinned host words: JV_SH_* below
    int32_t serve_idle_ticks;    // s_memrealtime ticks (100 MHz) without a claim after which the kernel exits
    int32_t done_all;            // completion words are also set for rows that come back flagged (no later rung in server mode)
    int32_t* done;           // optional completion words in host-visible memory: 1 once query i's row is final (see jv_pqw_body.h)
    int32_t pqw_lut_off;     // several-waves kernel: LDS byte offset of the table rows kept in LDS ([W][NL][256] floats)
    // visited counts taken AFTER the search launch by jv_visited_kernel (jv_kernels_vis.hip, round 5): a query without a visit
    // limit copies its expansion log into the arena instead of counting inside the search kernel
    int32_t* vis_arena;      // [vis_cap_units * 4] ints, or nullptr = count inside the search kernel
    uint32_t vis_cap_units;  // arena size in 16-byte units
    uint32_t* vis_cursor;    // units handed out so far (zeroed per call)
    uint32_t* vis_off;       // [nq] first unit of query i's log
    int32_t* vis_n;          // [nq] its length; 0 = counted by the search kernel (zeroed per call)
};

// one launch of jv_visited_kernel: jvector's visitedCount of every query whose log sits in the arena
struct JvVisArgs {
    const int32_t* adj;      // [n][R] adjacency of the base layer
    int32_t R;
    int32_t entry;           // entry point: in the set before the counted inserts, never counted
    const int32_t* arena;
    const uint32_t* vis_off;
    const int32_t* vis_n;
    int32_t nq;
    int32_t slots;           // hash slots in LDS (power of two)
    int32_t* out_stats;      // [nq][4]: word 0 = visited
    // Lucene's visit limit (> 0): a row whose visited + expanded reaches it becomes an early-terminated row (flag, no results), as
    // the search kernel makes it when it counts itself (J/JVectorReader.java:202-207, AbstractKnnVectorQuery)
    int32_t visit_limit, topK;
    int32_t* out_nodes;      // [nq][topK]
    int32_t* out_docs;       // [nq][topK] or nullptr
    float* out_scores;       // [nq][topK]
    int32_t* out_count;      // [nq]
    int32_t* out_flags;      // [nq]
    unsigned long long* dbg; // diagnostic build (-DJV_STAMPS) only: cycle accumulators of jv_visited_fast_kernel; nullptr in the product
};

// query-server words and slot layout (shared by jv_abi.cpp and the SERVE kernel instances)
enum { JV_SV_HEAD = 0, JV_SV_PUBLISHED = 1, JV_SV_LOCK = 2, JV_SV_EXITED = 3, JV_SV_LAST_CLAIM = 4, JV_SV_STOP_SEEN = 5 };
enum { JV_SH_TAIL = 0, JV_SH_STOP = 1, JV_SH_ALIVE = 2 };
#define JV_SERVE_TOPK_MAX 128  /* (round 6: 64 -> 128: the reference benchmark's own shape is K = 100, README.md:90-95) */
struct JvServeSlot {           // header of one ring slot; the query (float[d], zero padded to 16 B) follows at JV_SERVE_QUERY_OFF
    int32_t topK, rk, visit_limit;
    float rerank_floor;
    int32_t done;              // 0 while pending; 1 once the row below is final
    int32_t count, flags;
    int32_t stats[4];
    int32_t ticket;            // the sequence number this slot's content belongs to (a grid skips a slot of another generation)
    uint64_t accept;           // filtered server only: the query's doc filter, a DEVICE pointer (the index's filter cache owns it) ...
    int64_t accept_docs;       // ... and its doc-id space
    int32_t nodes[JV_SERVE_TOPK_MAX];
    int32_t docs[JV_SERVE_TOPK_MAX];
    float scores[JV_SERVE_TOPK_MAX];
};
#define JV_SERVE_QUERY_OFF ((int)sizeof(JvServeSlot))

#define JV_FLAG_OVERFLOW 0x80000000u /* on-chip scratch overflow: query must be re-run on the big path */
#define JV_FLAG_FAILED   0x40000000u /* big path overflowed as well */
#define JV_FLAG_BIG      0x1u
#define JV_FLAG_EARLY    0x2u /* visit_limit reached: the search stopped, no results (the caller falls back to the exact scan) */
