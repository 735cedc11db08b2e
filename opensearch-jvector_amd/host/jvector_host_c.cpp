// jvector_host_c.cpp — flat C entry points over the C++ host mirror, so the Python parity tests can
// drive JVectorReader / JVectorKnnFloatVectorQuery the way the reference's own tests do
// (KNNJVectorTests.java).  Exceptions become negative codes: -1 IllegalArgument, -4 Unsupported, -3 IO.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "jvector_host.hpp"

using namespace jvector_amd;

namespace {
thread_local std::string g_err;
template <typename F>
int guard(F&& f) {
    try {
        f();
        return 0;
    } catch (const IllegalArgumentException& e) {
        g_err = e.what();
        return -1;
    } catch (const UnsupportedOperationException& e) {
        g_err = e.what();
        return -4;
    } catch (const IOException& e) {
        g_err = e.what();
        return -3;
    } catch (const std::exception& e) {
        g_err = e.what();
        return -5;
    }
}
const char* kField = "test_field";
FixedBitSet* bitsFrom(const uint64_t* words, int maxDoc) {
    if (!words) return nullptr;
    FixedBitSet* b = new FixedBitSet(maxDoc);
    for (int i = 0; i < maxDoc; i++)
        if ((words[i >> 6] >> (i & 63)) & 1ull) b->set(i);
    return b;
}
}  // namespace

extern "C" {

const char* jvh_last_error() { return g_err.c_str(); }

// opens a reader with one field ("test_field") over a flattened index
int jvh_reader_open(const jv_index_desc* desc, int lucene_similarity, const int32_t* ord2doc, int n_ord, int max_doc_id,
                    void** out) {
    return guard([&] {
        std::vector<int> o2d(ord2doc, ord2doc + n_ord);
        GraphNodeIdToDocMap map(o2d, max_doc_id);
        auto* r = new JVectorReader();
        try {
            r->addField(kField, *desc, (LuceneSimilarity)lucene_similarity, std::move(map));
        } catch (...) {
            delete r;
            throw;
        }
        *out = r;
    });
}

void jvh_reader_close(void* reader) { delete (JVectorReader*)reader; }

// JVectorKnnFloatVectorQuery through the per-leaf logic of AbstractKnnVectorQuery
int jvh_query_search_leaf(void* reader, const float* target, int dim, int k, int over_query_factor, float threshold,
                          float rerank_floor, const uint64_t* filter_words, const uint64_t* live_words, int max_doc,
                          int32_t* out_docs, float* out_scores, int32_t* out_count, int64_t* out_total_hits,
                          int32_t* out_used_exact) {
    return guard([&] {
        std::unique_ptr<FixedBitSet> filter(bitsFrom(filter_words, max_doc)), live(bitsFrom(live_words, max_doc));
        JVectorKnnFloatVectorQuery q(kField, std::vector<float>(target, target + dim), k, over_query_factor, threshold, rerank_floor);
        bool exact = false;
        TopDocs t = q.searchLeaf(*(JVectorReader*)reader, filter.get(), live.get(), max_doc, &exact);
        *out_count = (int32_t)t.scoreDocs.size();
        for (size_t i = 0; i < t.scoreDocs.size(); i++) {
            out_docs[i] = t.scoreDocs[i].doc;
            out_scores[i] = t.scoreDocs[i].score;
        }
        *out_total_hits = t.totalHits;
        *out_used_exact = exact ? 1 : 0;
    });
}

// the same for nq queries that share filter, liveDocs and parameters (JVectorKnnFloatVectorQuery::searchLeafBatch): out_docs /
// out_scores [nq][k], out_count / out_used_exact [nq]
int jvh_query_search_leaf_batch(void* reader, const float* targets, int nq, int dim, int k, int over_query_factor, float threshold,
                                float rerank_floor, const uint64_t* filter_words, const uint64_t* live_words, int max_doc,
                                int exact_when_cheaper, double crossover_selectivity, int32_t* out_docs, float* out_scores,
                                int32_t* out_count, int32_t* out_used_exact) {
    return guard([&] {
        std::unique_ptr<FixedBitSet> filter(bitsFrom(filter_words, max_doc)), live(bitsFrom(live_words, max_doc));
        std::vector<uint8_t> exact;
        std::vector<TopDocs> res = JVectorKnnFloatVectorQuery::searchLeafBatch(*(JVectorReader*)reader, kField, targets, nq, dim, k,
                                                                              over_query_factor, threshold, rerank_floor, filter.get(),
                                                                              live.get(), max_doc, &exact, exact_when_cheaper != 0,
                                                                              crossover_selectivity);
        for (int i = 0; i < nq; i++) {
            const TopDocs& t = res[(size_t)i];
            out_count[i] = (int32_t)t.scoreDocs.size();
            for (size_t j = 0; j < (size_t)k; j++) {
                out_docs[(size_t)i * k + j] = j < t.scoreDocs.size() ? t.scoreDocs[j].doc : -1;
                out_scores[(size_t)i * k + j] = j < t.scoreDocs.size() ? t.scoreDocs[j].score : 0.0f;
            }
            if (out_used_exact) out_used_exact[i] = exact[(size_t)i];
        }
    });
}

// plain Lucene KnnFloatVectorQuery path: a foreign TopKnnCollector, re-wrapped by the reader with the
// defaults (J/JVectorReader.java:133-144) — what KNNJVectorTests.java:982-1027 exercises concurrently
int jvh_reader_search_plain_collector(void* reader, const float* target, int k, const uint64_t* accept_words,
                                      int max_doc, int32_t* out_docs, float* out_scores, int32_t* out_count,
                                      int64_t* out_visited) {
    return guard([&] {
        std::unique_ptr<FixedBitSet> acc(bitsFrom(accept_words, max_doc));
        AcceptDocs ad{acc.get()};
        TopKnnCollector c(k, INT32_MAX);
        ((JVectorReader*)reader)->search(kField, target, c, acc ? &ad : nullptr);
        TopDocs t = c.topDocs();
        *out_count = (int32_t)t.scoreDocs.size();
        for (size_t i = 0; i < t.scoreDocs.size(); i++) {
            out_docs[i] = t.scoreDocs[i].doc;
            out_scores[i] = t.scoreDocs[i].score;
        }
        *out_visited = c.visitedCount();
    });
}

int jvh_reader_search_bytes(void* reader) {
    return guard([&] {
        int8_t t[4] = {0, 0, 0, 0};
        TopKnnCollector c(1, 10);
        ((JVectorReader*)reader)->search(kField, t, c, nullptr);
    });
}

void jvh_counters(int64_t out[5]) {
    out[0] = KNNCounter::KNN_QUERY_VISITED_NODES;
    out[1] = KNNCounter::KNN_QUERY_RERANKED_COUNT;
    out[2] = KNNCounter::KNN_QUERY_EXPANDED_NODES;
    out[3] = KNNCounter::KNN_QUERY_EXPANDED_BASE_LAYER_NODES;
    out[4] = KNNCounter::KNN_QUERY_GRAPH_SEARCH_TIME;
}

// GraphNodeIdToDocMap: serialise -> parse -> (optional sort remap) -> lookups (no GPU involved)
int jvh_docmap_roundtrip(const int32_t* ord2doc, int n_ord, int max_doc_id, const int32_t* old_to_new /*nullable*/,
                         uint8_t* out_bytes, int32_t* inout_nbytes, int32_t* out_ord2doc, int32_t* out_doc2ord,
                         int32_t* out_max_doc) {
    return guard([&] {
        GraphNodeIdToDocMap m(std::vector<int>(ord2doc, ord2doc + n_ord), max_doc_id);
        if (old_to_new) m.update(std::vector<int>(old_to_new, old_to_new + m.maxDoc()));
        std::vector<uint8_t> bytes = m.toOutput();
        if ((int)bytes.size() > *inout_nbytes) throw IOException("buffer too small");
        std::memcpy(out_bytes, bytes.data(), bytes.size());
        *inout_nbytes = (int32_t)bytes.size();
        GraphNodeIdToDocMap r = GraphNodeIdToDocMap::fromBytes(bytes);
        for (int i = 0; i < r.size(); i++) out_ord2doc[i] = r.getLuceneDocId(i);
        for (int i = 0; i < r.maxDoc(); i++) out_doc2ord[i] = r.getJVectorNodeId(i);
        *out_max_doc = r.maxDoc();
    });
}

int jvh_similarity_ord_to_dist_func(int ord, int* out) {
    return guard([&] { *out = (int)VectorSimilarityMapper::ordToDistFunc(ord); });
}
int jvh_similarity_dist_func_to_ord(int lucene_similarity) {
    return VectorSimilarityMapper::distFuncToOrd((LuceneSimilarity)lucene_similarity);
}

// The reference's calling pattern at the boundary: many searcher threads, each issuing ONE query at a time on the
// same reader handle (T/index/engine/JVectorConcurrentQueryTests.java:78-138, TJ/KNNJVectorTests.java:982-1027).
// Drives jv_search from `threads` native threads for `seconds`; thread t walks queries t, t+threads, ... round robin.
// out[0] = completed queries/s, out[1] = p50 ms, out[2] = p99 ms, out[3] = completed queries.
// When `check_nodes` is given ([nq][topK], e.g. from a batch call) every answer is compared with it and the
// number of mismatching queries is returned in out[4].
// `accept_words` (optional): one doc filter handed to every call, as a filtered k-NN query's leaf searches do
// (J/JVectorReader.java:157-163); `accept_key` != 0 names it for the library's filter cache (jv_search_ex).
int jvh_concurrent_search_bench_filtered(jv_index* index, const float* queries, int nq, int dim, int topK, int rerankK,
                                         int threads, double seconds, const int32_t* check_nodes, const uint64_t* accept_words,
                                         int64_t accept_num_docs, uint64_t accept_key, double out[5]) {
    if (!index || !queries || nq <= 0 || threads <= 0 || topK <= 0) {
        g_err = "bad argument";
        return -1;
    }
    std::atomic<bool> stop{false};
    std::atomic<int> failed{0};
    std::atomic<long long> mismatches{0};
    std::vector<std::vector<float>> lat((size_t)threads);
    std::vector<std::thread> pool;
    const auto t0 = std::chrono::steady_clock::now();
    for (int t = 0; t < threads; t++) {
        pool.emplace_back([&, t]() {
            std::vector<int32_t> nodes((size_t)topK), docs((size_t)topK);
            std::vector<float> scores((size_t)topK);
            int32_t count = 0, stats[4];
            auto& l = lat[(size_t)t];
            l.reserve(1 << 16);
            for (int qi = t % nq; !stop.load(std::memory_order_relaxed); qi = (qi + threads) % nq) {
                const auto a = std::chrono::steady_clock::now();
                int rc;
                if (accept_words && accept_key) {
                    jv_search_params p{};
                    p.struct_size = sizeof(p);
                    p.topK = topK;
                    p.rerankK = rerankK;
                    p.accept_doc_words = accept_words;
                    p.accept_num_docs = accept_num_docs;
                    p.accept_key = accept_key;
                    int32_t qflags = 0;
                    rc = jv_search_ex(index, queries + (size_t)qi * dim, &p, nodes.data(), docs.data(), scores.data(), &count, stats, &qflags);
                } else {
                    rc = jv_search(index, queries + (size_t)qi * dim, topK, rerankK, 0.0f, 0.0f, accept_words, accept_words ? accept_num_docs : 0,
                                   nodes.data(), docs.data(), scores.data(), &count, stats);
                }
                const auto b = std::chrono::steady_clock::now();
                if (rc != 0) {
                    failed.store(rc);
                    break;
                }
                if (check_nodes && memcmp(check_nodes + (size_t)qi * topK, nodes.data(), sizeof(int32_t) * topK) != 0)
                    mismatches.fetch_add(1);
                l.push_back(std::chrono::duration<float, std::milli>(b - a).count());
            }
        });
    }
    std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
    stop.store(true);
    for (auto& th : pool) th.join();
    const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (failed.load() != 0) {
        g_err = std::string("jv_search failed: ") + jv_last_error();
        return -5;
    }
    std::vector<float> all;
    for (auto& l : lat) all.insert(all.end(), l.begin(), l.end());
    std::sort(all.begin(), all.end());
    out[0] = all.size() / el;
    out[1] = all.empty() ? 0.0 : all[all.size() / 2];
    out[2] = all.empty() ? 0.0 : all[(size_t)(all.size() * 0.99)];
    out[3] = (double)all.size();
    out[4] = (double)mismatches.load();
    return 0;
}

int jvh_concurrent_search_bench(jv_index* index, const float* queries, int nq, int dim, int topK, int rerankK,
                                int threads, double seconds, const int32_t* check_nodes, double out[5]) {
    return jvh_concurrent_search_bench_filtered(index, queries, nq, dim, topK, rerankK, threads, seconds, check_nodes, nullptr, 0, 0, out);
}


// ---- .meta-jvector (host/jvector_host.hpp JVectorMeta): flat view for the byte-level round-trip test ----
struct jvh_meta_field {
    int32_t fieldNumber, vectorEncoding, similarityOrd, vectorDimension;
    int64_t vectorIndexOffset, vectorIndexLength, compressedVectorsOffset, compressedVectorsLength;
    int32_t quantizationType;
    float degreeOverflow;
    int32_t mapSize, mapMaxDoc;       // GraphNodeIdToDocMap: ordinals, doc-id space (maxDoc = highest doc id + 1)
    const int32_t* ord2doc;           // write: [mapSize]; read: filled into the caller's buffer
};

int jvh_meta_write(const uint8_t* segment_id, const char* suffix, int version, const jvh_meta_field* fields, int nfields,
                   uint8_t* out, int64_t cap, int64_t* out_len) {
    return guard([&] {
        std::vector<VectorIndexFieldMetadata> fs;
        for (int i = 0; i < nfields; i++) {
            VectorIndexFieldMetadata f;
            f.fieldNumber = fields[i].fieldNumber;
            f.vectorEncoding = fields[i].vectorEncoding;
            f.similarityOrd = fields[i].similarityOrd;
            f.vectorDimension = fields[i].vectorDimension;
            f.vectorIndexOffset = fields[i].vectorIndexOffset;
            f.vectorIndexLength = fields[i].vectorIndexLength;
            f.compressedVectorsOffset = fields[i].compressedVectorsOffset;
            f.compressedVectorsLength = fields[i].compressedVectorsLength;
            f.quantizationType = (int8_t)fields[i].quantizationType;
            f.degreeOverflow = fields[i].degreeOverflow;
            std::vector<int> o2d(fields[i].ord2doc, fields[i].ord2doc + fields[i].mapSize);
            f.graphNodeIdToDocMap = GraphNodeIdToDocMap(o2d, fields[i].mapMaxDoc - 1);
            fs.push_back(std::move(f));
        }
        std::vector<uint8_t> b = JVectorMeta::write(segment_id, suffix ? suffix : "", version, fs);
        *out_len = (int64_t)b.size();
        if ((int64_t)b.size() > cap) throw IOException("output buffer too small");
        memcpy(out, b.data(), b.size());
    });
}

int jvh_meta_read(const uint8_t* file, int64_t len, const uint8_t* segment_id, const char* suffix, jvh_meta_field* fields,
                  int max_fields, int* nfields, int* version, int32_t* ord2doc_buf, int64_t ord_cap) {
    return guard([&] {
        std::vector<uint8_t> in(file, file + len);
        std::vector<VectorIndexFieldMetadata> fs = JVectorMeta::read(in, segment_id, suffix ? suffix : "", version);
        *nfields = (int)fs.size();
        if ((int)fs.size() > max_fields) throw IOException("too many fields for the caller's buffer");
        int64_t used = 0;
        for (size_t i = 0; i < fs.size(); i++) {
            const VectorIndexFieldMetadata& f = fs[i];
            jvh_meta_field& o = fields[i];
            o.fieldNumber = f.fieldNumber;
            o.vectorEncoding = f.vectorEncoding;
            o.similarityOrd = f.similarityOrd;
            o.vectorDimension = f.vectorDimension;
            o.vectorIndexOffset = f.vectorIndexOffset;
            o.vectorIndexLength = f.vectorIndexLength;
            o.compressedVectorsOffset = f.compressedVectorsOffset;
            o.compressedVectorsLength = f.compressedVectorsLength;
            o.quantizationType = f.quantizationType;
            o.degreeOverflow = f.degreeOverflow;
            o.mapSize = f.graphNodeIdToDocMap.size();
            o.mapMaxDoc = f.graphNodeIdToDocMap.maxDoc();
            if (used + o.mapSize > ord_cap) throw IOException("ord2doc buffer too small");
            for (int k = 0; k < o.mapSize; k++) ord2doc_buf[used + k] = f.graphNodeIdToDocMap.ordToDoc()[(size_t)k];
            o.ord2doc = ord2doc_buf + used;
            used += o.mapSize;
        }
    });
}

}

