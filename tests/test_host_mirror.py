"""Host-side mirror (C++ restatement of the reference's Java classes): the parts that need no GPU."""
import importlib

import numpy as np
import pytest


@pytest.fixture(scope="module")
def host(pkg):
    return importlib.import_module("opensearch_jvector_amd.host")


def test_graph_node_id_to_doc_map_roundtrip_ka14(host):
    """GraphNodeIdToDocMapTests.java:73-154: serialise/parse round trip and lookups, incl. deleted (-1)."""
    ord2doc = np.array([5, 0, 3, -1, 9, 7], dtype=np.int32)
    data, o2d, d2o = host.docmap_roundtrip(ord2doc, 9)
    assert o2d.tolist() == ord2doc.tolist()
    assert len(d2o) == 10
    assert d2o.tolist() == [1, -1, -1, 2, -1, 0, -1, 5, -1, 4]
    # header: int32 LE version 1, vint size, vint maxDoc(len), then one vint per ordinal (-1 takes 5 bytes)
    assert data[:4] == bytes([1, 0, 0, 0]) and data[4] == 6 and data[5] == 10
    assert len(data) == 4 + 1 + 1 + 5 + 5  # five 1-byte docs + one 5-byte -1


def test_graph_node_id_to_doc_map_sort_remap(host):
    """GraphNodeIdToDocMap.update (index sort, J/GraphNodeIdToDocMap.java:104-139): reverse order."""
    n = 10
    ord2doc = np.arange(n, dtype=np.int32)
    old_to_new = np.arange(n - 1, -1, -1, dtype=np.int32)
    _, o2d, d2o = host.docmap_roundtrip(ord2doc, n - 1, old_to_new)
    assert o2d.tolist() == list(range(9, -1, -1))
    assert d2o.tolist() == list(range(9, -1, -1))


def test_docmap_rejects_bad_max_doc(host):
    with pytest.raises(host.HostError) as ei:
        host.docmap_roundtrip(np.array([0, 5], dtype=np.int32), 3)
    assert ei.value.code == -1 and "maxDocId is incorrect" in str(ei.value)


def test_vector_similarity_mapper(host):
    """J/JVectorReader.java:384-432: [EUCLIDEAN, DOT_PRODUCT, COSINE, DOT_PRODUCT]; MIP maps to ordinal 1."""
    assert [host.ord_to_dist_func(i) for i in range(4)] == [0, 1, 2, 1]
    assert [host.dist_func_to_ord(i) for i in range(4)] == [0, 1, 2, 1]
    with pytest.raises(host.HostError) as ei:
        host.ord_to_dist_func(9)
    assert ei.value.code == -1  # IllegalArgumentException


def test_host_library_exports_the_mirror_entry_points(host):
    """libjvhost.so (the C++ mirror of the reference's Java host classes) loads and exports every entry the Python
    plumbing binds, incl. the concurrent-caller driver (no compute without a GPU)."""
    lib = host.load_library()
    for name in ("jvh_reader_open", "jvh_reader_close", "jvh_query_search_leaf", "jvh_reader_search_plain_collector",
                 "jvh_reader_search_bytes", "jvh_counters", "jvh_docmap_roundtrip", "jvh_similarity_ord_to_dist_func",
                 "jvh_similarity_dist_func_to_ord", "jvh_concurrent_search_bench", "jvh_last_error"):
        assert hasattr(lib, name), name
