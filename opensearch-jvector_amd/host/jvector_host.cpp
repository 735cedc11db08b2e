// jvector_host.cpp — see jvector_host.hpp. C++ mirror of the reference's Java host side of the hot
// path; every search goes through the C ABI (include/jvgpu.h) to the HIP engine.
#include "jvector_host.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>

namespace jvector_amd {

std::atomic<int64_t> KNNCounter::KNN_QUERY_VISITED_NODES{0};
std::atomic<int64_t> KNNCounter::KNN_QUERY_RERANKED_COUNT{0};
std::atomic<int64_t> KNNCounter::KNN_QUERY_EXPANDED_NODES{0};
std::atomic<int64_t> KNNCounter::KNN_QUERY_EXPANDED_BASE_LAYER_NODES{0};
std::atomic<int64_t> KNNCounter::KNN_QUERY_GRAPH_SEARCH_TIME{0};

void throwForStatus(int status) {
    if (status == JV_OK) return;
    std::string msg = jv_last_error();
    switch (status) {
        case JV_EINVAL: throw IllegalArgumentException(msg);
        case JV_EUNSUPPORTED: throw UnsupportedOperationException(msg);
        default: throw IOException(msg);  // JV_ENOMEM / JV_EDEVICE / JV_EINTERNAL
    }
}

// JVECTOR_SUPPORTED_SIMILARITY_FUNCTIONS = [EUCLIDEAN, DOT_PRODUCT, COSINE, DOT_PRODUCT] (J/JVectorReader.java:389-394)
jv_similarity VectorSimilarityMapper::ordToDistFunc(int ord) {
    static const jv_similarity table[4] = {JV_SIM_EUCLIDEAN, JV_SIM_DOT_PRODUCT, JV_SIM_COSINE, JV_SIM_DOT_PRODUCT};
    if (ord < 0 || ord >= 4) throw IllegalArgumentException("Invalid ordinal: " + std::to_string(ord));
    return table[ord];
}
// indexOf(LUCENE_TO_JVECTOR_MAP.get(func)): MAXIMUM_INNER_PRODUCT -> DOT_PRODUCT -> 1
int VectorSimilarityMapper::distFuncToOrd(LuceneSimilarity s) {
    switch (s) {
        case LuceneSimilarity::EUCLIDEAN: return 0;
        case LuceneSimilarity::DOT_PRODUCT: return 1;
        case LuceneSimilarity::COSINE: return 2;
        case LuceneSimilarity::MAXIMUM_INNER_PRODUCT: return 1;
    }
    throw IllegalArgumentException("invalid distance function");
}

int FixedBitSet::cardinality() const {
    int c = 0;
    for (uint64_t w : words_) c += __builtin_popcountll(w);
    return c;
}

// ---- TopKnnCollector (Lucene NeighborQueue min-heap of encoded longs) ----
namespace {
inline int32_t sortableInt(float f) {
    int32_t b;
    std::memcpy(&b, &f, 4);
    return b ^ ((b >> 31) & 0x7fffffff);
}
inline float fromSortable(int32_t s) {
    int32_t b = s ^ ((s >> 31) & 0x7fffffff);
    float f;
    std::memcpy(&f, &b, 4);
    return f;
}
inline int64_t encode(int doc, float score) {
    return (int64_t)(((uint64_t)(uint32_t)sortableInt(score) << 32) | (uint64_t)(uint32_t)(~doc));
}
struct MinCmp {
    bool operator()(int64_t a, int64_t b) const { return a > b; }
};
}  // namespace

bool TopKnnCollector::collect(int docId, float similarity) {
    int64_t e = encode(docId, similarity);
    if ((int)heap_.size() < k_) {
        heap_.push_back(e);
        std::push_heap(heap_.begin(), heap_.end(), MinCmp());
        return true;
    }
    if (k_ == 0 || e <= heap_.front()) return false;
    std::pop_heap(heap_.begin(), heap_.end(), MinCmp());
    heap_.back() = e;
    std::push_heap(heap_.begin(), heap_.end(), MinCmp());
    return true;
}
float TopKnnCollector::minCompetitiveSimilarity() const {
    return (int)heap_.size() >= k_ && k_ > 0 ? fromSortable((int32_t)(heap_.front() >> 32)) : -INFINITY;
}
TopDocs TopKnnCollector::topDocs() {
    std::vector<int64_t> h = heap_;
    std::sort(h.begin(), h.end(), [](int64_t a, int64_t b) { return a > b; });
    TopDocs t;
    for (int64_t e : h) t.scoreDocs.push_back({~(int32_t)(uint32_t)(e & 0xFFFFFFFFll), fromSortable((int32_t)(e >> 32))});
    t.totalHits = visited_;
    t.totalHitsIsLowerBound = earlyTerminated();
    return t;
}

// ---- GraphNodeIdToDocMap ----
GraphNodeIdToDocMap::GraphNodeIdToDocMap(const std::vector<int>& ord2doc, int maxDocId) {
    if (ord2doc.empty()) return;
    graphNodeIdsToDocIds_ = ord2doc;
    int observed = *std::max_element(ord2doc.begin(), ord2doc.end());
    if (maxDocId < observed)
        throw IllegalArgumentException("The maxDocId is incorrect, provided " + std::to_string(maxDocId) +
                                       ", expected at least " + std::to_string(observed));
    docIdsToGraphNodeIds_.assign((size_t)maxDocId + 1, NO_VECTOR_OR_DELETED_DOC);
    for (size_t ord = 0; ord < ord2doc.size(); ord++)
        if (ord2doc[ord] != NO_VECTOR_OR_DELETED_DOC) docIdsToGraphNodeIds_[(size_t)ord2doc[ord]] = (int)ord;
}

namespace {
void writeVInt(std::vector<uint8_t>& out, int32_t v) {  // Lucene DataOutput.writeVInt
    uint32_t i = (uint32_t)v;
    while ((i & ~0x7Fu) != 0) {
        out.push_back((uint8_t)((i & 0x7F) | 0x80));
        i >>= 7;
    }
    out.push_back((uint8_t)i);
}
int32_t readVInt(const std::vector<uint8_t>& in, size_t& pos) {
    uint32_t v = 0;
    for (int shift = 0; shift < 35; shift += 7) {
        if (pos >= in.size()) throw IOException("read past EOF");
        uint8_t b = in[pos++];
        v |= (uint32_t)(b & 0x7F) << shift;
        if (!(b & 0x80)) return (int32_t)v;
    }
    throw IOException("Invalid vInt detected (too many bits)");
}
}  // namespace

std::vector<uint8_t> GraphNodeIdToDocMap::toOutput() const {
    std::vector<uint8_t> out;
    const int32_t version = 1;  // writeInt: little-endian since Lucene 9
    for (int i = 0; i < 4; i++) out.push_back((uint8_t)((uint32_t)version >> (8 * i)));
    writeVInt(out, (int32_t)graphNodeIdsToDocIds_.size());
    writeVInt(out, (int32_t)docIdsToGraphNodeIds_.size());
    for (int doc : graphNodeIdsToDocIds_) writeVInt(out, doc);  // -1 is written as a 5-byte vint, as in Java
    return out;
}

GraphNodeIdToDocMap GraphNodeIdToDocMap::fromBytes(const std::vector<uint8_t>& in) {
    if (in.size() < 4) throw IOException("read past EOF");
    int32_t version = (int32_t)((uint32_t)in[0] | (uint32_t)in[1] << 8 | (uint32_t)in[2] << 16 | (uint32_t)in[3] << 24);
    if (version != 1) throw IOException("Unsupported version: " + std::to_string(version));
    size_t pos = 4;
    int size = readVInt(in, pos);
    int maxDocId = readVInt(in, pos);
    GraphNodeIdToDocMap m;
    m.graphNodeIdsToDocIds_.assign((size_t)size, NO_VECTOR_OR_DELETED_DOC);
    m.docIdsToGraphNodeIds_.assign((size_t)maxDocId, NO_VECTOR_OR_DELETED_DOC);
    for (int ord = 0; ord < size; ord++) {
        int doc = readVInt(in, pos);
        if (doc != NO_VECTOR_OR_DELETED_DOC) {
            m.graphNodeIdsToDocIds_[(size_t)ord] = doc;
            m.docIdsToGraphNodeIds_.at((size_t)doc) = ord;
        }
    }
    return m;
}

void GraphNodeIdToDocMap::update(const std::vector<int>& oldToNew) {
    int maxNew = -1;
    for (int doc : graphNodeIdsToDocIds_) maxNew = std::max(maxNew, oldToNew.at((size_t)doc));
    int maxDocs = maxNew + 1;
    if (maxDocs < (int)graphNodeIdsToDocIds_.size())
        throw std::logic_error("Max docs " + std::to_string(maxDocs) + " is less than the number of ordinals " +
                               std::to_string(graphNodeIdsToDocIds_.size()));
    std::vector<int> newDoc2Ord((size_t)maxDocs, -1), newOrd2Doc(graphNodeIdsToDocIds_.size(), 0);
    for (size_t oldDoc = 0; oldDoc < docIdsToGraphNodeIds_.size(); oldDoc++) {
        int oldOrd = docIdsToGraphNodeIds_[oldDoc];
        if (oldOrd == -1) continue;
        int newDoc = oldToNew.at(oldDoc);
        newDoc2Ord[(size_t)newDoc] = oldOrd;
        newOrd2Doc[(size_t)oldOrd] = newDoc;
    }
    docIdsToGraphNodeIds_.swap(newDoc2Ord);
    graphNodeIdsToDocIds_.swap(newOrd2Doc);
}

// ---- .meta-jvector ----
namespace {
struct ByteWriter {
    std::vector<uint8_t> b;
    void u8(uint8_t v) { b.push_back(v); }
    void beInt(uint32_t v) { for (int i = 3; i >= 0; i--) b.push_back((uint8_t)(v >> (8 * i))); }
    void beLong(uint64_t v) { for (int i = 7; i >= 0; i--) b.push_back((uint8_t)(v >> (8 * i))); }
    void leInt(uint32_t v) { for (int i = 0; i < 4; i++) b.push_back((uint8_t)(v >> (8 * i))); }   // DataOutput.writeInt, Lucene >= 9
    void vInt(int32_t v) { writeVInt(b, v); }
    void vLong(int64_t v) {  // DataOutput.writeVLong (non-negative)
        uint64_t i = (uint64_t)v;
        while ((i & ~0x7Full) != 0) {
            b.push_back((uint8_t)((i & 0x7F) | 0x80));
            i >>= 7;
        }
        b.push_back((uint8_t)i);
    }
    void str(const std::string& s) {
        vInt((int32_t)s.size());
        b.insert(b.end(), s.begin(), s.end());
    }
};
struct ByteReader {
    const std::vector<uint8_t>& in;
    size_t pos = 0;
    uint8_t u8() {
        if (pos >= in.size()) throw IOException("read past EOF");
        return in[pos++];
    }
    uint32_t beInt() { uint32_t v = 0; for (int i = 0; i < 4; i++) v = (v << 8) | u8(); return v; }
    uint64_t beLong() { uint64_t v = 0; for (int i = 0; i < 8; i++) v = (v << 8) | u8(); return v; }
    uint32_t leInt() { uint32_t v = 0; for (int i = 0; i < 4; i++) v |= (uint32_t)u8() << (8 * i); return v; }
    int32_t vInt() { return readVInt(in, pos); }
    int64_t vLong() {
        uint64_t v = 0;
        for (int shift = 0; shift < 63; shift += 7) {
            uint8_t b = u8();
            v |= (uint64_t)(b & 0x7F) << shift;
            if (!(b & 0x80)) return (int64_t)v;
        }
        throw IOException("Invalid vLong detected (negative values disallowed)");
    }
};
uint32_t crc32(const uint8_t* p, size_t n) {  // java.util.zip.CRC32 (reflected 0xEDB88320), as BufferedChecksum uses
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        init = true;
    }
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; i++) c = table[(c ^ p[i]) & 0xFF] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}
constexpr uint32_t CODEC_MAGIC = 0x3fd76c17u, FOOTER_MAGIC = ~CODEC_MAGIC;
}  // namespace

std::vector<uint8_t> JVectorMeta::write(const uint8_t segmentId[16], const std::string& segmentSuffix, int version,
                                        const std::vector<VectorIndexFieldMetadata>& fields) {
    if (version < VERSION_START || version > VERSION_CURRENT) throw IllegalArgumentException("unsupported meta version " + std::to_string(version));
    if (segmentSuffix.size() > 255) throw IllegalArgumentException("suffix too long");
    ByteWriter w;
    // CodecUtil.writeIndexHeader
    w.beInt(CODEC_MAGIC);
    w.str(META_CODEC_NAME);
    w.beInt((uint32_t)version);
    for (int i = 0; i < 16; i++) w.u8(segmentId[i]);
    w.u8((uint8_t)segmentSuffix.size());
    w.b.insert(w.b.end(), segmentSuffix.begin(), segmentSuffix.end());
    for (const VectorIndexFieldMetadata& f : fields) {
        w.leInt((uint32_t)f.fieldNumber);  // J/JVectorWriter.java:299 (consumed by readFields' loop header)
        // VectorIndexFieldMetadata.toOutput (:528-540)
        w.leInt((uint32_t)f.fieldNumber);
        w.leInt((uint32_t)f.vectorEncoding);
        w.leInt((uint32_t)f.similarityOrd);
        w.vInt(f.vectorDimension);
        w.vLong(f.vectorIndexOffset);
        w.vLong(f.vectorIndexLength);
        w.vLong(f.compressedVectorsOffset);
        w.vLong(f.compressedVectorsLength);
        if (version >= VERSION_WITH_QUANTIZATION_TYPE) w.u8((uint8_t)f.quantizationType);
        uint32_t bits;
        memcpy(&bits, &f.degreeOverflow, 4);
        w.leInt(bits);
        const std::vector<uint8_t> m = f.graphNodeIdToDocMap.toOutput();
        w.b.insert(w.b.end(), m.begin(), m.end());
    }
    w.leInt(0xFFFFFFFFu);  // end-of-fields marker (:575)
    // CodecUtil.writeFooter
    w.beInt(FOOTER_MAGIC);
    w.beInt(0);
    w.beLong((uint64_t)crc32(w.b.data(), w.b.size()));
    return w.b;
}

std::vector<VectorIndexFieldMetadata> JVectorMeta::read(const std::vector<uint8_t>& file, const uint8_t expectedSegmentId[16],
                                                        const std::string& expectedSuffix, int* versionOut) {
    ByteReader r{file};
    // CodecUtil.checkIndexHeader
    if (r.beInt() != CODEC_MAGIC) throw IOException("codec header mismatch: bad magic");
    const int32_t nameLen = r.vInt();
    std::string name;
    for (int i = 0; i < nameLen; i++) name.push_back((char)r.u8());
    if (name != META_CODEC_NAME) throw IOException("codec mismatch: actual codec=" + name + " vs expected codec=" + META_CODEC_NAME);
    const int version = (int)r.beInt();
    if (version < VERSION_START) throw IOException("Format version is not supported (too old): " + std::to_string(version));
    if (version > VERSION_CURRENT) throw IOException("Format version is not supported (too new): " + std::to_string(version));
    for (int i = 0; i < 16; i++)
        if (r.u8() != expectedSegmentId[i]) throw IOException("file mismatch, expected id differs");
    const int suffixLen = r.u8();
    std::string suffix;
    for (int i = 0; i < suffixLen; i++) suffix.push_back((char)r.u8());
    if (suffix != expectedSuffix) throw IOException("file mismatch, expected suffix=" + expectedSuffix + ", got=" + suffix);
    std::vector<VectorIndexFieldMetadata> fields;
    // readFields (J/JVectorReader.java:255-262)
    for (int32_t fieldNumber = (int32_t)r.leInt(); fieldNumber != -1; fieldNumber = (int32_t)r.leInt()) {
        VectorIndexFieldMetadata f;
        f.fieldNumber = (int32_t)r.leInt();
        if (f.fieldNumber != fieldNumber) throw IOException("field number mismatch in meta record");
        f.vectorEncoding = (int32_t)r.leInt();
        if (f.vectorEncoding < 0 || f.vectorEncoding > 1) throw IOException("Invalid vector encoding id: " + std::to_string(f.vectorEncoding));
        f.similarityOrd = (int32_t)r.leInt();
        if (f.similarityOrd < 0 || f.similarityOrd > 2) throw IllegalArgumentException("Invalid ordinal: " + std::to_string(f.similarityOrd));
        f.vectorDimension = r.vInt();
        f.vectorIndexOffset = r.vLong();
        f.vectorIndexLength = r.vLong();
        f.compressedVectorsOffset = r.vLong();
        f.compressedVectorsLength = r.vLong();
        if (version >= VERSION_WITH_QUANTIZATION_TYPE) f.quantizationType = (int8_t)r.u8();
        else f.quantizationType = f.compressedVectorsLength > 0 ? 1 : 0;  // v0: PQ iff compressed vectors are present
        const uint32_t bits = r.leInt();
        memcpy(&f.degreeOverflow, &bits, 4);
        // GraphNodeIdToDocMap(IndexInput): parse in place to learn its length
        const size_t start = r.pos;
        const int32_t mapVersion = (int32_t)r.leInt();
        if (mapVersion != 1) throw IOException("Unsupported version: " + std::to_string(mapVersion));
        const int32_t size = r.vInt();
        (void)r.vInt();
        for (int i = 0; i < size; i++) (void)r.vInt();
        f.graphNodeIdToDocMap = GraphNodeIdToDocMap::fromBytes(std::vector<uint8_t>(file.begin() + (long)start, file.begin() + (long)r.pos));
        fields.push_back(std::move(f));
    }
    // CodecUtil.checkFooter
    const size_t footerStart = r.pos;
    if (file.size() - footerStart != 16) throw IOException("misplaced codec footer (file truncated or extended?)");
    if (r.beInt() != FOOTER_MAGIC) throw IOException("codec footer mismatch (file truncated?)");
    if (r.beInt() != 0) throw IOException("codec footer mismatch: unknown algorithmID");
    const uint64_t stored = r.beLong();
    const uint32_t actual = crc32(file.data(), footerStart + 8);
    if ((stored & 0xFFFFFFFF00000000ull) != 0 || (uint32_t)stored != actual) throw IOException("checksum failed (hardware problem?)");
    if (versionOut) *versionOut = version;
    return fields;
}

// ---- JVectorReader ----
JVectorReader::FieldEntry::~FieldEntry() { jv_index_destroy(index); }

void JVectorReader::addField(const std::string& field, const jv_index_desc& flattened, LuceneSimilarity sim,
                             GraphNodeIdToDocMap map) {
    auto fe = std::make_unique<FieldEntry>();
    jv_index_desc desc = flattened;
    // similarityFunction = VectorSimilarityMapper.ordToDistFunc(distFuncToOrd(luceneSim))  (:286-288)
    fe->luceneSimilarity = sim;
    fe->similarityFunction = VectorSimilarityMapper::ordToDistFunc(VectorSimilarityMapper::distFuncToOrd(sim));
    desc.similarity = fe->similarityFunction;
    // the MIP fix-up of wrapExactScoreFunction / JVectorVectorScorer (:220-239; JVectorVectorScorer.java:46-50)
    desc.score_scale = sim == LuceneSimilarity::MAXIMUM_INNER_PRODUCT ? 2.0f : 1.0f;
    fe->graphNodeIdToDocMap = std::move(map);
    desc.ord2doc = fe->graphNodeIdToDocMap.size() ? fe->graphNodeIdToDocMap.ordToDoc().data() : nullptr;
    desc.max_doc = fe->graphNodeIdToDocMap.maxDoc();
    fe->dimension = desc.d;
    fe->size = desc.n;
    fe->hasPQ = desc.pq_M > 0;
    throwForStatus(jv_index_create(&desc, &fe->index));
    fieldEntryMap_[field] = std::move(fe);
}

const JVectorReader::FieldEntry* JVectorReader::fieldEntry(const std::string& field) const {
    auto it = fieldEntryMap_.find(field);
    return it == fieldEntryMap_.end() ? nullptr : it->second.get();
}

void JVectorReader::search(const std::string& field, const float* target, KnnCollector& knnCollector,
                           const AcceptDocs* acceptDocs) {
    const FieldEntry* fieldEntry = this->fieldEntry(field);
    if (!fieldEntry) throw IllegalArgumentException("field not found: " + field);
    // :132-144 — foreign collectors are re-wrapped with the defaults
    JVectorKnnCollector* jvectorKnnCollector = dynamic_cast<JVectorKnnCollector*>(&knnCollector);
    std::unique_ptr<JVectorKnnCollector> rewrapped;
    if (!jvectorKnnCollector) {
        rewrapped = std::make_unique<JVectorKnnCollector>(knnCollector, KNNConstants::DEFAULT_QUERY_SIMILARITY_THRESHOLD,
                                                          KNNConstants::DEFAULT_QUERY_RERANK_FLOOR,
                                                          KNNConstants::DEFAULT_OVER_QUERY_FACTOR);
        jvectorKnnCollector = rewrapped.get();
    }
    const auto graphSearchStart = std::chrono::steady_clock::now();
    // :157-163 — acceptDocs == null or bits == null accepts every ordinal; otherwise the engine tests
    // ord2doc[ord] != -1 && bits.get(ord2doc[ord]) on the device
    const FixedBitSet* b = acceptDocs ? acceptDocs->bits() : nullptr;
    const int topK = jvectorKnnCollector->k();
    const int rerankK = topK * jvectorKnnCollector->getOverQueryFactor();
    std::vector<int32_t> nodes((size_t)std::max(topK, 1)), docs((size_t)std::max(topK, 1));
    std::vector<float> scores((size_t)std::max(topK, 1));
    int32_t count = 0, stats[JV_NUM_STATS] = {0, 0, 0, 0};
    // :165-173 — graphSearcher.search(ssp, k, k * overQueryFactor, threshold, rerankFloor, compatibleBits)
    // (the engine also receives Lucene's visit limit: a search that reaches it is going to be discarded by
    //  AbstractKnnVectorQuery for the exact scan, so the engine stops it instead of finishing it — include/jvgpu.h)
    jv_search_params sp;
    memset(&sp, 0, sizeof(sp));
    sp.struct_size = sizeof(sp);
    sp.topK = topK;
    sp.rerankK = rerankK;
    sp.threshold = jvectorKnnCollector->getThreshold();
    sp.rerankFloor = jvectorKnnCollector->getRerankFloor();
    sp.accept_doc_words = b ? b->getBits() : nullptr;
    sp.accept_num_docs = b ? b->length() : 0;
    const int64_t limit = jvectorKnnCollector->visitLimit();
    sp.visit_limit = (limit > 0 && limit < INT32_MAX) ? limit : 0;
    int32_t qflags = 0;
    throwForStatus(jv_search_ex(fieldEntry->index, target, &sp, nodes.data(), docs.data(), scores.data(), &count, stats, &qflags));
    // :175-177
    for (int i = 0; i < count; i++) jvectorKnnCollector->collect(docs[(size_t)i], scores[(size_t)i]);
    const auto searchTime = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - graphSearchStart).count();
    // :183-193
    KNNCounter::KNN_QUERY_VISITED_NODES += stats[JV_STAT_VISITED];
    KNNCounter::KNN_QUERY_RERANKED_COUNT += stats[JV_STAT_RERANKED];
    KNNCounter::KNN_QUERY_EXPANDED_NODES += stats[JV_STAT_EXPANDED];
    KNNCounter::KNN_QUERY_EXPANDED_BASE_LAYER_NODES += stats[JV_STAT_EXPANDED_BASE];
    KNNCounter::KNN_QUERY_GRAPH_SEARCH_TIME += searchTime;
    // :202-207
    const int visitedCount = stats[JV_STAT_VISITED] + stats[JV_STAT_EXPANDED];
    if (visitedCount > 0) jvectorKnnCollector->incVisitedCount(visitedCount);
}

void JVectorReader::search(const std::string&, const int8_t*, KnnCollector&, const AcceptDocs*) {
    throw UnsupportedOperationException("Byte vector search is not supported yet with jVector");
}

std::vector<float> JVectorReader::scoreDocs(const std::string& field, const float* target, const std::vector<int>& docIds) {
    const FieldEntry* fe = fieldEntry(field);
    if (!fe) throw IllegalArgumentException("field not found: " + field);
    std::vector<int32_t> ords(docIds.size());
    for (size_t i = 0; i < docIds.size(); i++) {
        int doc = docIds[i];
        ords[i] = doc >= 0 && doc < fe->graphNodeIdToDocMap.maxDoc() ? fe->graphNodeIdToDocMap.getJVectorNodeId(doc)
                                                                        : GraphNodeIdToDocMap::NO_VECTOR_OR_DELETED_DOC;
    }
    std::vector<float> out(docIds.size(), 0.0f);
    if (!docIds.empty()) throwForStatus(jv_score_ordinals(fe->index, target, ords.data(), (int32_t)ords.size(), out.data()));
    return out;
}

void JVectorReader::searchBatch(const std::string& field, const float* targets, int nq,
                                std::vector<JVectorKnnCollector*>& collectors, const AcceptDocs* acceptDocs) {
    const FieldEntry* fieldEntry = this->fieldEntry(field);
    if (!fieldEntry) throw IllegalArgumentException("field not found: " + field);
    if (nq <= 0) return;
    if ((int)collectors.size() != nq) throw IllegalArgumentException("one collector per target");
    JVectorKnnCollector* first = collectors[0];
    for (JVectorKnnCollector* c : collectors)
        if (c->k() != first->k() || c->getOverQueryFactor() != first->getOverQueryFactor() || c->getThreshold() != first->getThreshold() ||
            c->getRerankFloor() != first->getRerankFloor() || c->visitLimit() != first->visitLimit())
            throw IllegalArgumentException("searchBatch: the collectors must share k, overQueryFactor, threshold, rerankFloor and visitLimit");
    const auto graphSearchStart = std::chrono::steady_clock::now();
    const FixedBitSet* b = acceptDocs ? acceptDocs->bits() : nullptr;
    const int topK = first->k();
    const int rerankK = topK * first->getOverQueryFactor();
    const size_t rows = (size_t)nq * (size_t)std::max(topK, 1);
    std::vector<int32_t> nodes(rows), docs(rows), count((size_t)nq), stats((size_t)nq * JV_NUM_STATS), status((size_t)nq), flags((size_t)nq);
    std::vector<float> scores(rows);
    jv_search_params sp;
    memset(&sp, 0, sizeof(sp));
    sp.struct_size = sizeof(sp);
    sp.topK = topK;
    sp.rerankK = rerankK;
    sp.threshold = first->getThreshold();
    sp.rerankFloor = first->getRerankFloor();
    sp.accept_doc_words = b ? b->getBits() : nullptr;
    sp.accept_num_docs = b ? b->length() : 0;
    const int64_t limit = first->visitLimit();
    sp.visit_limit = (limit > 0 && limit < INT32_MAX) ? limit : 0;
    throwForStatus(jv_search_batch_ex(fieldEntry->index, targets, nq, &sp, nodes.data(), docs.data(), scores.data(), count.data(),
                                      stats.data(), status.data(), flags.data()));
    const auto searchTime = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - graphSearchStart).count();
    for (int i = 0; i < nq; i++) {
        JVectorKnnCollector* c = collectors[(size_t)i];
        const int32_t* st = stats.data() + (size_t)i * JV_NUM_STATS;
        for (int j = 0; j < count[(size_t)i]; j++) c->collect(docs[(size_t)i * topK + j], scores[(size_t)i * topK + j]);  // :175-177
        KNNCounter::KNN_QUERY_VISITED_NODES += st[JV_STAT_VISITED];                                                      // :183-193
        KNNCounter::KNN_QUERY_RERANKED_COUNT += st[JV_STAT_RERANKED];
        KNNCounter::KNN_QUERY_EXPANDED_NODES += st[JV_STAT_EXPANDED];
        KNNCounter::KNN_QUERY_EXPANDED_BASE_LAYER_NODES += st[JV_STAT_EXPANDED_BASE];
        int visitedCount = st[JV_STAT_VISITED] + st[JV_STAT_EXPANDED];                                                    // :202-207
        // (a search the engine stopped AT the limit has nothing to collect and must read as early-terminated)
        if ((flags[(size_t)i] & JV_QFLAG_EARLY_TERMINATED) && sp.visit_limit > 0 && visitedCount < sp.visit_limit) visitedCount = (int)sp.visit_limit;
        if (visitedCount > 0) c->incVisitedCount(visitedCount);
    }
    KNNCounter::KNN_QUERY_GRAPH_SEARCH_TIME += searchTime;
}

std::vector<TopDocs> JVectorReader::exactSearchBatch(const std::string& field, const float* targets, int nq, int k,
                                                     const FixedBitSet& accept) {
    const FieldEntry* fe = fieldEntry(field);
    if (!fe) throw IllegalArgumentException("field not found: " + field);
    std::vector<TopDocs> out((size_t)std::max(nq, 0));
    if (nq <= 0 || k <= 0) return out;
    if (k > JV_XB_TOPK_MAX) {
        // beyond the batch scorer's top-k: the per-query scorer + a host-side queue, as a single exactSearch does
        std::vector<int> docs;
        for (int doc = 0; doc < accept.length(); doc++)
            if (accept.get(doc) && doc < fe->graphNodeIdToDocMap.maxDoc() &&
                fe->graphNodeIdToDocMap.getJVectorNodeId(doc) != GraphNodeIdToDocMap::NO_VECTOR_OR_DELETED_DOC)
                docs.push_back(doc);
        for (int i = 0; i < nq; i++) {
            std::vector<float> sc = scoreDocs(field, targets + (size_t)i * fe->dimension, docs);
            TopKnnCollector top(k, INT64_MAX);
            for (size_t j = 0; j < docs.size(); j++) top.collect(docs[j], sc[j]);
            out[(size_t)i] = top.topDocs();
            out[(size_t)i].totalHits = (int64_t)out[(size_t)i].scoreDocs.size();
        }
        return out;
    }
    const size_t rows = (size_t)nq * (size_t)k;
    std::vector<int32_t> docs(rows), count((size_t)nq);
    std::vector<float> scores(rows);
    jv_exact_batch_params p;
    memset(&p, 0, sizeof(p));
    p.struct_size = sizeof(p);
    p.topK = k;
    p.accept_doc_words = accept.getBits();
    p.accept_num_docs = accept.length();
    throwForStatus(jv_score_ordinals_batch(fe->index, targets, nq, &p, nullptr, docs.data(), scores.data(), count.data(), nullptr));
    for (int i = 0; i < nq; i++) {
        TopDocs& t = out[(size_t)i];
        for (int j = 0; j < count[(size_t)i]; j++) t.scoreDocs.push_back({docs[(size_t)i * k + j], scores[(size_t)i * k + j]});
        t.totalHits = (int64_t)t.scoreDocs.size();
    }
    return out;
}

// ---- JVectorKnnFloatVectorQuery ----
TopDocs JVectorKnnFloatVectorQuery::approximateSearch(JVectorReader& reader, const AcceptDocs* acceptDocs,
                                                      int64_t visitedLimit) const {
    TopKnnCollector delegateCollector(k_, visitedLimit);
    JVectorKnnCollector knnCollector(delegateCollector, threshold_, rerankFloor_, overQueryFactor_);
    const JVectorReader::FieldEntry* fe = reader.fieldEntry(field_);
    if (!fe) return TopDocs();                                     // :58-61 NO_RESULTS
    if (std::min(knnCollector.k(), fe->size) == 0) return TopDocs();  // :62-64
    std::vector<float> targetCopy = target_;                       // getTargetCopy()
    reader.search(field_, targetCopy.data(), knnCollector, acceptDocs);
    return knnCollector.topDocs();
}

TopDocs JVectorKnnFloatVectorQuery::exactSearch(JVectorReader& reader, const FixedBitSet& accept) const {
    const JVectorReader::FieldEntry* fe = reader.fieldEntry(field_);
    if (k_ >= 1 && k_ <= JV_XB_TOPK_MAX && accept.length() > 0) {
        // Round 5: one engine call — the accepted docs that have a vector are found, scored (JVectorVectorScorer.score's exact
        // scores) and cut to the best k by (score desc, doc asc) on the device; searcher threads that fall back at the same time
        // under the same filter are answered as one batch inside the library (jv_exact_search's group commit)
        std::vector<int32_t> docs((size_t)k_);
        std::vector<float> scores((size_t)k_);
        int32_t count = 0;
        jv_exact_batch_params p;
        memset(&p, 0, sizeof(p));
        p.struct_size = sizeof(p);
        p.topK = k_;
        p.accept_doc_words = accept.getBits();
        p.accept_num_docs = accept.length();
        throwForStatus(jv_exact_search(fe->index, target_.data(), &p, nullptr, docs.data(), scores.data(), &count));
        TopDocs t;
        for (int i = 0; i < count; i++) t.scoreDocs.push_back({docs[(size_t)i], scores[(size_t)i]});
        t.totalHits = (int64_t)t.scoreDocs.size();
        return t;
    }
    std::vector<int> docs;
    for (int doc = 0; doc < accept.length(); doc++)
        if (accept.get(doc) && doc < fe->graphNodeIdToDocMap.maxDoc() &&
            fe->graphNodeIdToDocMap.getJVectorNodeId(doc) != GraphNodeIdToDocMap::NO_VECTOR_OR_DELETED_DOC)
            docs.push_back(doc);
    std::vector<float> sc = reader.scoreDocs(field_, target_.data(), docs);
    TopKnnCollector top(k_, INT64_MAX);
    for (size_t i = 0; i < docs.size(); i++) top.collect(docs[i], sc[i]);
    TopDocs t = top.topDocs();
    t.totalHits = (int64_t)t.scoreDocs.size();
    return t;
}

TopDocs JVectorKnnFloatVectorQuery::searchLeaf(JVectorReader& reader, const FixedBitSet* filter,
                                               const FixedBitSet* liveDocs, int maxDoc, bool* usedExact) const {
    if (usedExact) *usedExact = false;
    const JVectorReader::FieldEntry* fe = reader.fieldEntry(field_);
    if (!fe) return TopDocs();
    // AbstractKnnVectorQuery.getLeafResults: no filter -> approximateSearch(liveDocs, Integer.MAX_VALUE)
    if (!filter) {
        AcceptDocs ad{liveDocs};
        TopDocs t = approximateSearch(reader, liveDocs ? &ad : nullptr, INT32_MAX);
        t.totalHits = (int64_t)t.scoreDocs.size();
        return t;
    }
    FixedBitSet accept(maxDoc);
    for (int doc = 0; doc < maxDoc; doc++)
        if (filter->get(doc) && (!liveDocs || liveDocs->get(doc))) accept.set(doc);
    const int cost = accept.cardinality();
    if (cost <= k_) {  // fewer matches than k: exact search
        if (usedExact) *usedExact = true;
        return exactSearch(reader, accept);
    }
    AcceptDocs ad{&accept};
    TopKnnCollector probe(k_, cost);
    TopDocs t = approximateSearch(reader, &ad, cost);
    if (!t.totalHitsIsLowerBound) {  // collector did not early-terminate
        t.totalHits = (int64_t)t.scoreDocs.size();
        return t;
    }
    if (usedExact) *usedExact = true;
    return exactSearch(reader, accept);  // visited limit reached -> exact fallback
}

std::vector<TopDocs> JVectorKnnFloatVectorQuery::searchLeafBatch(JVectorReader& reader, const std::string& field, const float* targets,
                                                                 int nq, int dim, int k, int overQueryFactor, float threshold,
                                                                 float rerankFloor, const FixedBitSet* filter, const FixedBitSet* liveDocs,
                                                                 int maxDoc, std::vector<uint8_t>* usedExactSearch, bool exactWhenCheaper,
                                                                 double crossoverSelectivity) {
    std::vector<TopDocs> out((size_t)std::max(nq, 0));
    if (usedExactSearch) usedExactSearch->assign((size_t)std::max(nq, 0), 0);
    const JVectorReader::FieldEntry* fe = reader.fieldEntry(field);
    if (!fe || nq <= 0) return out;
    if (dim != fe->dimension) throw IllegalArgumentException("vector dimension differs from the field's");
    if (std::min(k, fe->size) == 0) return out;  // J/JVectorKnnFloatVectorQuery.java:62-64
    auto approximate = [&](const AcceptDocs* ad, int64_t visitedLimit, std::vector<uint8_t>& early) {
        std::vector<std::unique_ptr<TopKnnCollector>> tops;
        std::vector<std::unique_ptr<JVectorKnnCollector>> wraps;
        std::vector<JVectorKnnCollector*> ptrs;
        for (int i = 0; i < nq; i++) {
            tops.push_back(std::make_unique<TopKnnCollector>(k, visitedLimit));
            wraps.push_back(std::make_unique<JVectorKnnCollector>(*tops.back(), threshold, rerankFloor, overQueryFactor));
            ptrs.push_back(wraps.back().get());
        }
        reader.searchBatch(field, targets, nq, ptrs, ad);
        early.assign((size_t)nq, 0);
        for (int i = 0; i < nq; i++) {
            out[(size_t)i] = wraps[(size_t)i]->topDocs();
            early[(size_t)i] = out[(size_t)i].totalHitsIsLowerBound ? 1 : 0;
            out[(size_t)i].totalHits = (int64_t)out[(size_t)i].scoreDocs.size();
        }
    };
    std::vector<uint8_t> early;
    if (!filter) {  // AbstractKnnVectorQuery.getLeafResults: no filter -> approximateSearch(liveDocs, Integer.MAX_VALUE)
        AcceptDocs ad{liveDocs};
        approximate(liveDocs ? &ad : nullptr, INT32_MAX, early);
        return out;
    }
    FixedBitSet accept(maxDoc);
    for (int doc = 0; doc < maxDoc; doc++)
        if (filter->get(doc) && (!liveDocs || liveDocs->get(doc))) accept.set(doc);
    const int cost = accept.cardinality();
    const bool allExact = cost <= k || (exactWhenCheaper && (double)cost <= crossoverSelectivity * (double)std::max(fe->size, 1));
    if (allExact) {
        out = reader.exactSearchBatch(field, targets, nq, k, accept);
        if (usedExactSearch) usedExactSearch->assign((size_t)nq, 1);
        return out;
    }
    AcceptDocs ad{&accept};
    approximate(&ad, cost, early);
    std::vector<int> redo;
    for (int i = 0; i < nq; i++)
        if (early[(size_t)i]) redo.push_back(i);
    if (!redo.empty()) {  // visited limit reached -> exact fallback, all such queries in one call
        std::vector<float> sub(redo.size() * (size_t)dim);
        for (size_t j = 0; j < redo.size(); j++) memcpy(sub.data() + j * (size_t)dim, targets + (size_t)redo[j] * (size_t)dim, (size_t)dim * sizeof(float));
        std::vector<TopDocs> ex = reader.exactSearchBatch(field, sub.data(), (int)redo.size(), k, accept);
        for (size_t j = 0; j < redo.size(); j++) {
            out[(size_t)redo[j]] = std::move(ex[j]);
            if (usedExactSearch) (*usedExactSearch)[(size_t)redo[j]] = 1;
        }
    }
    return out;
}

}  // namespace jvector_amd
