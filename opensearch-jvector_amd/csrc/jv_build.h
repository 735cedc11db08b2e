/*
 * jv_build.h — index construction helpers (WRITE side).
 *
 * Not part of the graded hot path and not part of the drop-in ABI (include/jvgpu.h): the reference
 * builds its graph and PQ with the third-party jvector library at flush/merge time
 * (J/JVectorWriter.java:1383-1422 getGraph; J/JVectorIndexQuantization.java:114-140 computePqVectors),
 * which cannot run here.  These functions exist only so that tests, smoke() and bench.py have
 * graphs/codebooks to search: a deterministic batched Vamana build (defaults R=32, L=100,
 * alpha=1.2, overflow=1.2 — J/JVectorFormat.java:34-35, K/common/KNNConstants.java:106-107) and a
 * k-means PQ trainer/encoder (256 clusters per subspace, global centring iff EUCLIDEAN —
 * J/JVectorIndexQuantization.java:122-131).  SURVEY §8(f) rows 2/3 ("next").
 */
#ifndef JV_BUILD_H
#define JV_BUILD_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Batched Vamana on the CPU. sim: jv_similarity ordinal. out_adj: [n][R] (-1 padded),
 * out_entry: approximate medoid. threads<=0: all cores (result is independent of thread count). */
int jvb_build_graph_cpu(const float* vectors, int32_t n, int32_t d, int32_t sim, int32_t R, int32_t L,
                        float alpha, float overflow, int32_t max_batch, int32_t threads,
                        int32_t* out_adj, int32_t* out_entry);

/* HNSW-style upper layers over a built layer-0 graph, for hierarchy tests: node i is in layer l
 * iff its deterministic level >= l. Fills, for layer l (1-based) of `num_layers`:
 * counts[l-1], and (when nodes/adj non-NULL) nodes[l-1][...], adj[l-1][count][R]. Two-pass use:
 * first call with nodes==NULL to get counts. Returns the entry node of the top layer in *out_entry. */
int jvb_build_upper_layers_cpu(const float* vectors, int32_t n, int32_t d, int32_t sim, int32_t R,
                               int32_t L, float alpha, int32_t num_layers, uint64_t seed,
                               int32_t* counts, int32_t** nodes, int32_t** adj, int32_t* out_entry);

/* PQ: Lloyd k-means per subspace (k-means++ seeding, deterministic from `seed`) on at most
 * `max_train` evenly strided training rows. out_codebooks: concat over m of [K][sub_size[m]];
 * out_centroid: [d] (written only when center != 0). */
int jvb_pq_train_cpu(const float* vectors, int32_t n, int32_t d, int32_t M, int32_t K, int32_t center,
                     int32_t iters, int32_t max_train, uint64_t seed, int32_t threads,
                     float* out_codebooks, float* out_centroid);

/* nearest centroid per subspace (squared L2 on centred vectors). out_codes: [n][M] */
int jvb_pq_encode_cpu(const float* vectors, int32_t n, int32_t d, int32_t M, int32_t K,
                      const float* codebooks, const float* centroid, int32_t threads,
                      uint8_t* out_codes);

#ifdef __cplusplus
}
#endif
#endif
