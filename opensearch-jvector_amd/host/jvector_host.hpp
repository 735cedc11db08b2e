// jvector_host.hpp — C++ mirror of the reference's HOST side of the hot path, above the C ABI.
//
// The reference's host code is Java (no JVM in this image), so the classes the path runs through are
// restated here in C++ with the same names, argument meaning and error behaviour, each calling the
// MI355X engine through include/jvgpu.h exactly where the Java shim of INTEGRATION.md would:
//
//   JVectorKnnFloatVectorQuery::approximateSearch   J/JVectorKnnFloatVectorQuery.java:50-70
//   JVectorKnnCollector                             J/JVectorKnnCollector.java:15-66
//   JVectorReader::search                           J/JVectorReader.java:129-210   (one jv_search call)
//   JVectorReader::FieldEntry                       J/JVectorReader.java:284-337   (jv_index_create/destroy)
//   VectorSimilarityMapper                          J/JVectorReader.java:384-432
//   GraphNodeIdToDocMap                             J/GraphNodeIdToDocMap.java:25-177
//   JVectorVectorScorer (exact fallback scorer)     J/JVectorVectorScorer.java:36-53 (jv_score_ordinals)
//   KNNCounter (the five search counters)           K/plugin/stats/KNNCounter.java:30-34
//
// plus the thin slice of Lucene that wraps the path (TopKnnCollector, AcceptDocs, the per-leaf logic
// of AbstractKnnVectorQuery: filter cost <= k -> exact search; approximate search with
// visitLimit = cost; exact fallback when the collector early-terminated).
// (J/ = src/main/java/org/opensearch/knn/index/codec/jvector/, K/ = src/main/java/org/opensearch/knn/)
#pragma once

#include <atomic>
#include <cstdint>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/jvgpu.h"

namespace jvector_amd {

// ---- Java exception types the reference throws on this path ----
struct IOException : std::runtime_error { using std::runtime_error::runtime_error; };
struct IllegalArgumentException : std::invalid_argument { using std::invalid_argument::invalid_argument; };
struct UnsupportedOperationException : std::logic_error { using std::logic_error::logic_error; };
void throwForStatus(int status);  // jv_status + jv_last_error() -> the exception the Java shim would raise

// org.apache.lucene.index.VectorSimilarityFunction ordinals
enum class LuceneSimilarity : int { EUCLIDEAN = 0, DOT_PRODUCT = 1, COSINE = 2, MAXIMUM_INNER_PRODUCT = 3 };

// J/JVectorReader.java:384-432
struct VectorSimilarityMapper {
    static jv_similarity ordToDistFunc(int luceneOrdinal);  // throws IllegalArgumentException
    static int distFuncToOrd(LuceneSimilarity s);
};

// org.apache.lucene.util.FixedBitSet (the part the path uses)
class FixedBitSet {
public:
    explicit FixedBitSet(int numBits) : numBits_(numBits), words_((size_t)(numBits + 63) / 64, 0ull) {}
    void set(int i) { words_[(size_t)i >> 6] |= 1ull << (i & 63); }
    void clear(int i) { words_[(size_t)i >> 6] &= ~(1ull << (i & 63)); }
    bool get(int i) const { return (words_[(size_t)i >> 6] >> (i & 63)) & 1ull; }
    int length() const { return numBits_; }
    int cardinality() const;
    const uint64_t* getBits() const { return words_.data(); }
private:
    int numBits_;
    std::vector<uint64_t> words_;
};

// org.apache.lucene.search.AcceptDocs: bits() == nullptr means "accept all"
struct AcceptDocs {
    const FixedBitSet* bitsPtr = nullptr;
    const FixedBitSet* bits() const { return bitsPtr; }
    int cost(int maxDoc) const { return bitsPtr ? bitsPtr->cardinality() : maxDoc; }
};

struct ScoreDoc { int doc; float score; };
struct TopDocs { int64_t totalHits = 0; bool totalHitsIsLowerBound = false; std::vector<ScoreDoc> scoreDocs; };

// org.apache.lucene.search.KnnCollector
class KnnCollector {
public:
    virtual ~KnnCollector() = default;
    virtual bool earlyTerminated() const = 0;
    virtual void incVisitedCount(int count) = 0;
    virtual int64_t visitedCount() const = 0;
    virtual int64_t visitLimit() const = 0;
    virtual int k() const = 0;
    virtual bool collect(int docId, float similarity) = 0;
    virtual float minCompetitiveSimilarity() const = 0;
    virtual TopDocs topDocs() = 0;
};

// org.apache.lucene.search.TopKnnCollector: bounded min-heap; equal scores prefer the lower doc id
class TopKnnCollector : public KnnCollector {
public:
    TopKnnCollector(int k, int64_t visitLimit) : k_(k), visitLimit_(visitLimit) {}
    bool earlyTerminated() const override { return visited_ >= visitLimit_; }
    void incVisitedCount(int count) override { visited_ += count; }
    int64_t visitedCount() const override { return visited_; }
    int64_t visitLimit() const override { return visitLimit_; }
    int k() const override { return k_; }
    bool collect(int docId, float similarity) override;
    float minCompetitiveSimilarity() const override;
    TopDocs topDocs() override;
private:
    int k_;
    int64_t visitLimit_;
    int64_t visited_ = 0;
    std::vector<int64_t> heap_;  // NeighborQueue encoding: sortable(score) << 32 | ~doc
};

// J/JVectorKnnCollector.java:15-66 — carries threshold / rerankFloor / overQueryFactor to the reader
class JVectorKnnCollector : public KnnCollector {
public:
    JVectorKnnCollector(KnnCollector& delegate, float threshold, float rerankFloor, int overQueryFactor)
        : delegate_(delegate), threshold_(threshold), rerankFloor_(rerankFloor), overQueryFactor_(overQueryFactor) {}
    bool earlyTerminated() const override { return delegate_.earlyTerminated(); }
    void incVisitedCount(int count) override { delegate_.incVisitedCount(count); }
    int64_t visitedCount() const override { return delegate_.visitedCount(); }
    int64_t visitLimit() const override { return delegate_.visitLimit(); }
    int k() const override { return delegate_.k(); }
    bool collect(int docId, float similarity) override { return delegate_.collect(docId, similarity); }
    float minCompetitiveSimilarity() const override { return delegate_.minCompetitiveSimilarity(); }
    TopDocs topDocs() override { return delegate_.topDocs(); }
    float getThreshold() const { return threshold_; }
    float getRerankFloor() const { return rerankFloor_; }
    int getOverQueryFactor() const { return overQueryFactor_; }
private:
    KnnCollector& delegate_;
    float threshold_, rerankFloor_;
    int overQueryFactor_;
};

// K/common/KNNConstants.java:86-91
struct KNNConstants {
    static constexpr int DEFAULT_OVER_QUERY_FACTOR = 5;
    static constexpr float DEFAULT_QUERY_SIMILARITY_THRESHOLD = 0.0f;
    static constexpr float DEFAULT_QUERY_RERANK_FLOOR = 0.0f;
    static constexpr int DEFAULT_MINIMUM_BATCH_SIZE_FOR_QUANTIZATION = 1024;
};

// K/plugin/stats/KNNCounter.java:30-34 (only the counters JVectorReader.search feeds, :189-193)
struct KNNCounter {
    static std::atomic<int64_t> KNN_QUERY_VISITED_NODES, KNN_QUERY_RERANKED_COUNT, KNN_QUERY_EXPANDED_NODES,
        KNN_QUERY_EXPANDED_BASE_LAYER_NODES, KNN_QUERY_GRAPH_SEARCH_TIME;
};

// J/GraphNodeIdToDocMap.java:25-177
class GraphNodeIdToDocMap {
public:
    static constexpr int NO_VECTOR_OR_DELETED_DOC = -1;
    GraphNodeIdToDocMap() = default;
    GraphNodeIdToDocMap(const std::vector<int>& graphNodeIdsToDocIds, int maxDoc);  // :61-95 shape
    static GraphNodeIdToDocMap fromBytes(const std::vector<uint8_t>& in);            // :39-59 (int32 LE version, VInts)
    std::vector<uint8_t> toOutput() const;                                            // :169-176
    int getJVectorNodeId(int luceneDocId) const { return docIdsToGraphNodeIds_.at((size_t)luceneDocId); }
    int getLuceneDocId(int graphNodeId) const { return graphNodeIdsToDocIds_.at((size_t)graphNodeId); }
    // index-sort remap (:104-139): newDocId = old2new[oldDocId]
    void update(const std::vector<int>& oldToNew);
    int size() const { return (int)graphNodeIdsToDocIds_.size(); }
    int maxDoc() const { return (int)docIdsToGraphNodeIds_.size(); }
    const std::vector<int>& ordToDoc() const { return graphNodeIdsToDocIds_; }
private:
    std::vector<int> graphNodeIdsToDocIds_, docIdsToGraphNodeIds_;
};

// ---- the plugin-owned segment metadata file "<segment>_<suffix>.meta-jvector"  ----
// J/JVectorWriter.java:512-563 (VectorIndexFieldMetadata.toOutput / the IndexInput constructor), :299 (field number written
// in front of every record), :573-577 (end marker -1 + footer); J/JVectorReader.java:52-81,255-262 (checkIndexHeader /
// readFields / checkFooter); J/JVectorFormat.java:23,31-33 (codec name, versions 0 and 1).
struct VectorIndexFieldMetadata {
    int32_t fieldNumber = 0;
    int32_t vectorEncoding = 1;            // Lucene VectorEncoding ordinal: 0 BYTE, 1 FLOAT32
    int32_t similarityOrd = 0;             // VectorSimilarityMapper.distFuncToOrd: 0 L2, 1 DOT (also MIP), 2 COSINE
    int32_t vectorDimension = 0;
    int64_t vectorIndexOffset = 0, vectorIndexLength = 0, compressedVectorsOffset = 0, compressedVectorsLength = 0;
    int8_t quantizationType = 0;           // 0 none, 1 PQ, 2 NVQ-inline (absent in version 0: inferred from the PQ length)
    float degreeOverflow = 0.0f;
    GraphNodeIdToDocMap graphNodeIdToDocMap;
};
namespace JVectorMeta {
constexpr const char* META_CODEC_NAME = "JVectorVectorsFormatMeta";
constexpr int VERSION_START = 0, VERSION_WITH_QUANTIZATION_TYPE = 1, VERSION_CURRENT = 1;
// the whole file: CodecUtil index header (big-endian magic / version, codec name, 16-byte segment id, suffix), the
// records (little-endian ints, vints), the end marker and the CodecUtil footer (magic, algorithm 0, CRC32 of all before it)
std::vector<uint8_t> write(const uint8_t segmentId[16], const std::string& segmentSuffix, int version,
                           const std::vector<VectorIndexFieldMetadata>& fields);
// throws IOException for a truncated / corrupt file (bad magic, codec, version range, segment id, suffix, checksum)
std::vector<VectorIndexFieldMetadata> read(const std::vector<uint8_t>& file, const uint8_t expectedSegmentId[16],
                                           const std::string& expectedSuffix, int* versionOut);
}  // namespace JVectorMeta

// J/JVectorReader.java — the codec reader; one FieldEntry per (segment, field) owns one HBM index
class JVectorReader {
public:
    struct FieldEntry {
        jv_index* index = nullptr;  // OnDiskGraphIndex + PQVectors, resident in HBM
        GraphNodeIdToDocMap graphNodeIdToDocMap;
        LuceneSimilarity luceneSimilarity = LuceneSimilarity::EUCLIDEAN;
        jv_similarity similarityFunction = JV_SIM_EUCLIDEAN;
        int dimension = 0;
        int size = 0;
        bool hasPQ = false;
        ~FieldEntry();  // FieldEntry.close (:367-378) -> jv_index_destroy
    };
    // stand-in for the segment-open path (:52-81, :255-337): the caller supplies what
    // OnDiskGraphIndex.load / PQVectors.load would have produced
    void addField(const std::string& field, const jv_index_desc& flattened, LuceneSimilarity sim,
                  GraphNodeIdToDocMap map);
    // :129-210
    void search(const std::string& field, const float* target, KnnCollector& knnCollector, const AcceptDocs* acceptDocs);
    // :241-245 — always throws UnsupportedOperationException
    void search(const std::string& field, const int8_t* target, KnnCollector& knnCollector, const AcceptDocs* acceptDocs);
    // exact scorer used by Lucene's fallback: JVectorFloatVectorValues.scorer -> JVectorVectorScorer.score
    std::vector<float> scoreDocs(const std::string& field, const float* target, const std::vector<int>& docIds);
    // ---- extensions of this engine (not in the reference: Lucene runs one leaf search per query and thread) ----
    // nq searches with the SAME collector parameters and the SAME acceptDocs in one engine call: what :147-207 does per
    // query — k, k * overQueryFactor, threshold, rerankFloor, the acceptOrds lambda, the counters, incVisitedCount — for
    // every target.  collectors[i] receives query i's results.
    void searchBatch(const std::string& field, const float* targets, int nq, std::vector<JVectorKnnCollector*>& collectors,
                     const AcceptDocs* acceptDocs);
    // the exact fallback for nq targets under one acceptDocs: top k of JVectorVectorScorer.score over the accepted docs
    // that have a vector, by (score desc, doc asc) — HitQueue order (jv_score_ordinals_batch)
    std::vector<TopDocs> exactSearchBatch(const std::string& field, const float* targets, int nq, int k, const FixedBitSet& accept);
    const FieldEntry* fieldEntry(const std::string& field) const;
    void close() { fieldEntryMap_.clear(); }
private:
    std::map<std::string, std::unique_ptr<FieldEntry>> fieldEntryMap_;
};

// J/JVectorKnnFloatVectorQuery.java:20-70 + the per-leaf part of Lucene's AbstractKnnVectorQuery
class JVectorKnnFloatVectorQuery {
public:
    JVectorKnnFloatVectorQuery(std::string field, std::vector<float> target, int k, int overQueryFactor,
                               float threshold, float rerankFloor)
        : field_(std::move(field)), target_(std::move(target)), k_(k), overQueryFactor_(overQueryFactor),
          threshold_(threshold), rerankFloor_(rerankFloor) {}
    // :50-70
    TopDocs approximateSearch(JVectorReader& reader, const AcceptDocs* acceptDocs, int64_t visitedLimit) const;
    // AbstractKnnVectorQuery.getLeafResults: filter (may be null) AND liveDocs (may be null)
    TopDocs searchLeaf(JVectorReader& reader, const FixedBitSet* filter, const FixedBitSet* liveDocs, int maxDoc,
                       bool* usedExactSearch = nullptr) const;
    int k() const { return k_; }
    // The same per-leaf logic for MANY queries of one field that share filter, liveDocs and parameters (extension: the
    // reference issues them one by one from Lucene's searcher threads).  Every query gets exactly the answer searchLeaf
    // gives it: cost <= k -> exact; approximate search with visitLimit = cost; queries whose collector early-terminated ->
    // exact.  Both steps are ONE engine call each (jv_search_batch_ex, jv_score_ordinals_batch).
    // exactWhenCheaper (default false — a DEVIATION from Lucene's rule, opt-in): when the filter is selective enough that the
    // batched exact scan is estimated cheaper than the graph search (selectivity <= crossoverSelectivity), skip the graph
    // search and answer with the exact top k — a different (exact, recall 1) answer than the graph's approximate one.
    static std::vector<TopDocs> searchLeafBatch(JVectorReader& reader, const std::string& field, const float* targets, int nq,
                                                int dim, int k, int overQueryFactor, float threshold, float rerankFloor,
                                                const FixedBitSet* filter, const FixedBitSet* liveDocs, int maxDoc,
                                                std::vector<uint8_t>* usedExactSearch = nullptr, bool exactWhenCheaper = false,
                                                double crossoverSelectivity = 0.25);
private:
    TopDocs exactSearch(JVectorReader& reader, const FixedBitSet& accept) const;
    std::string field_;
    std::vector<float> target_;
    int k_, overQueryFactor_;
    float threshold_, rerankFloor_;
};

}  // namespace jvector_amd
