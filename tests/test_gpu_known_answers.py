"""The reference's known-answer tests (KNNJVectorTests.java) replayed on the GPU through the C++ host
mirror: JVectorKnnFloatVectorQuery -> JVectorReader.search -> C ABI -> HIP kernels."""
import importlib
import threading

import numpy as np
import pytest

from conftest import recall_at_k
from ka_support import case_index, leaf_search, load_cases

pytestmark = pytest.mark.gpu
CASES = load_cases()


@pytest.fixture(scope="module")
def host(pkg):
    return importlib.import_module("opensearch_jvector_amd.host")


@pytest.mark.parametrize("case", CASES["cases"], ids=[c["name"] for c in CASES["cases"]])
def test_known_answer_cases_through_host_mirror(pkg, host, case):
    ix = case_index(pkg, case)
    reader = host.JVectorReader(ix, case["lucene_similarity"])
    try:
        docs, scores, total, used_exact = reader.search_leaf(case["query"], case["k"], case["over_query_factor"],
                                                            filter_docs=case["filter_docs"], deleted_docs=case["deleted_docs"])
        assert total == case["k"]                      # assertEquals(k, topDocs.totalHits.value())
        assert docs == case["expected_docs"], case["cite"]
        np.testing.assert_allclose(scores, case["expected_scores"], atol=case["tol"], rtol=0)
    finally:
        reader.close()


@pytest.mark.parametrize("case", CASES["cases"], ids=[c["name"] for c in CASES["cases"]])
def test_known_answer_cases_raw_abi_equals_oracle(pkg, pyoracle, case):
    ix = case_index(pkg, case)
    gpu = pkg.binding.GpuIndex(ix)
    d1 = leaf_search(pkg, gpu, ix, case)
    d2 = leaf_search(pkg, pyoracle.Oracle(pkg.binding, ix), ix, case)
    assert d1[0] == d2[0] == case["expected_docs"]
    assert np.array_equal(np.asarray(d1[1], np.float32).view(np.uint32), np.asarray(d2[1], np.float32).view(np.uint32))
    gpu.close()


def test_ka8_pq_recall_gpu(pkg, host):
    """KNNJVectorTests.java:1358-1403 on the GPU engine."""
    base = pkg.datagen.java_random_vectors(1, 1024, 16)
    ix = pkg.builder.build_index_cpu(base, 0, R=32, L=100, pq_M=8)
    reader = host.JVectorReader(ix, "EUCLIDEAN")
    q = np.zeros(16, np.float32)
    docs, scores, total, _ = reader.search_leaf(q, 50, 5)
    d2 = ((base - q) ** 2).sum(1)
    truth = np.argsort(d2, kind="stable")[:50]
    assert total == 50
    assert recall_at_k(np.asarray([docs]), np.asarray([truth])) >= 0.95
    c = host.counters()
    assert c["KNN_QUERY_RERANKED_COUNT"] > 0 and c["KNN_QUERY_VISITED_NODES"] > 0
    reader.close()


def test_foreign_collector_is_rewrapped_and_concurrent_search_is_safe(pkg, host, pyoracle):
    """KNNJVectorTests.java:982-1027 + JVectorConcurrentQueryTests.java:78-196: plain KnnFloatVectorQuery
    (re-wrap path: over-query 5) from 10 threads on one reader; every thread must get the brute-force answer."""
    n, d, k = 2000, 32, 5
    base = pkg.datagen.java_random_vectors(7, n, d)
    ix = pkg.builder.build_index_cpu(base, 0, R=32, L=100)
    reader = host.JVectorReader(ix, "EUCLIDEAN")
    queries = pkg.datagen.java_random_vectors(8, 40, d)
    orc = pyoracle.Oracle(pkg.binding, ix)
    want = orc.search_batch(queries, k, k * 5)
    errors = []

    def worker(tid):
        try:
            for it in range(50):
                qi = (tid * 7 + it) % len(queries)
                docs, scores, visited = reader.search_plain_collector(queries[qi], k)
                if docs != want.docs[qi].tolist():
                    errors.append((tid, qi, docs, want.docs[qi].tolist()))
                if visited != want.stats[qi][0] + want.stats[qi][2]:
                    errors.append((tid, qi, "visited", visited))
        except Exception as e:  # pragma: no cover
            errors.append((tid, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(10)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errors, errors[:3]
    reader.close()


def test_byte_vector_search_is_unsupported(pkg, host):
    """J/JVectorReader.java:241-245 throws UnsupportedOperationException."""
    ix = case_index(pkg, CASES["cases"][0])
    reader = host.JVectorReader(ix, "EUCLIDEAN")
    with pytest.raises(host.HostError) as ei:
        reader.search_bytes()
    assert ei.value.code == -4 and "Byte vector search is not supported" in str(ei.value)
    reader.close()


def test_rerankk_smaller_than_topk_is_illegal_argument(pkg):
    b = pkg.binding
    ix = case_index(pkg, CASES["cases"][0])
    gpu = b.GpuIndex(ix)
    with pytest.raises(b.JvError) as ei:
        gpu.search(np.zeros(2, np.float32), 5, 3)
    assert ei.value.code == b.JV_EINVAL and "rerankK 3 must be >= topK 5" in str(ei.value)
    gpu.close()


def test_empty_index_and_k_zero(pkg, host):
    b = pkg.binding
    empty = b.IndexData(vectors=np.zeros((0, 8), np.float32), adj=np.zeros((0, 4), np.int32), entry_node=-1)
    gpu = b.GpuIndex(empty)
    r = gpu.search_batch(np.ones((3, 8), np.float32), 5, 25)
    assert r.count.tolist() == [0, 0, 0] and (r.nodes == -1).all()
    gpu.close()
    one = b.IndexData(vectors=np.ones((1, 8), np.float32), adj=np.full((1, 4), -1, np.int32), entry_node=0)
    gpu = b.GpuIndex(one)
    r = gpu.search_batch(np.ones((2, 8), np.float32), 5, 25)
    assert r.count.tolist() == [1, 1] and r.nodes[:, 0].tolist() == [0, 0] and r.stats[0].tolist() == [0, 0, 1, 1]
    r = gpu.search_batch(np.ones((2, 8), np.float32), 0, 0)
    assert r.count.tolist() == [0, 0]
    gpu.close()


def test_baseline_config_c1_10k_128_l2_k10(pkg, host, pyoracle):
    """BASELINE.json configs[0]: 10k random float32 vectors, d=128, L2, k=10 through the query/reader surface
    (JVectorKnnFloatVectorQuery defaults: over-query 5) — GPU engine vs oracle vs brute force."""
    n, d, k = 10_000, 128, 10
    base = pkg.datagen.java_random_vectors(42, n, d)          # the JMH generator (Random(42) uniform [0,1))
    queries = pkg.datagen.java_random_vectors(43, 50, d)
    ix = pkg.builder.build_index_cpu(base, 0, R=32, L=100)
    reader = host.JVectorReader(ix, "EUCLIDEAN")
    orc = pyoracle.Oracle(pkg.binding, ix)
    want = orc.search_batch(queries, k, k * 5)
    truth, _ = orc.brute_force(queries, k)
    found = []
    for i in range(len(queries)):
        docs, scores, total, used_exact = reader.search_leaf(queries[i], k, 5)
        assert total == k and not used_exact
        assert docs == want.docs[i].tolist()
        assert np.array_equal(np.asarray(scores, np.float32).view(np.uint32), want.scores[i].view(np.uint32))
        found.append(docs)
    # i.i.d. uniform 128-d is the reference's JMH input; it asserts no recall on it (graph quality is the
    # builder's, not the search path's) — sanity floor only
    assert recall_at_k(np.asarray(found), truth) >= 0.6
    reader.close()
