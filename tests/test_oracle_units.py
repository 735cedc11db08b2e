"""Unit tests of the oracle's building blocks (CPU only)."""
import numpy as np
import pytest


def test_sortable_int_and_key_order(pkg, pyoracle):
    lib = pyoracle.load(pkg.binding.JvIndexDesc)
    vals = np.array([-np.inf, -3.5, -1e-30, -0.0, 0.0, 1e-30, 0.25, 1.0, 7.5, np.inf], dtype=np.float32)
    s = [lib.jvo_float_to_sortable_int(float(v)) for v in vals]
    assert s == sorted(s), "floatToSortableInt must be monotone"
    # higher score -> larger key; equal score -> LOWER node id has the larger key
    assert lib.jvo_encode_key(5, 0.5) > lib.jvo_encode_key(5, 0.25)
    assert lib.jvo_encode_key(3, 0.5) > lib.jvo_encode_key(4, 0.5)
    assert lib.jvo_encode_key(0, 0.5) > lib.jvo_encode_key(2**31 - 1, 0.5)
    assert lib.jvo_encode_key(7, -1.0) < lib.jvo_encode_key(7, 0.0)


@pytest.mark.parametrize("d", [1, 2, 3, 4, 5, 63, 64, 65, 100, 127, 128, 768, 1000])
def test_canonical_sums_close_to_float64(pkg, pyoracle, d):
    lib = pyoracle.load(pkg.binding.JvIndexDesc)
    rng = np.random.default_rng(d)
    a = rng.standard_normal(d).astype(np.float32)
    b = rng.standard_normal(d).astype(np.float32)
    dot = lib.jvo_raw_dot(a.ctypes.data, b.ctypes.data, d)
    l2 = lib.jvo_raw_l2(a.ctypes.data, b.ctypes.data, d)
    a64, b64 = a.astype(np.float64), b.astype(np.float64)
    assert dot == pytest.approx(a64.dot(b64), rel=1e-5, abs=1e-5)
    assert l2 == pytest.approx(((a64 - b64) ** 2).sum(), rel=1e-5)


def test_canonical_dot_is_the_documented_order(pkg, pyoracle):
    """64 strided fmaf partials + adjacent-pair tree, restated independently in numpy/float32."""
    import math
    lib = pyoracle.load(pkg.binding.JvIndexDesc)
    rng = np.random.default_rng(3)
    for d in (7, 64, 200, 768):
        a = rng.standard_normal(d).astype(np.float32)
        b = rng.standard_normal(d).astype(np.float32)
        p = [np.float32(0)] * 64
        d4 = (d + 3) & ~3
        for i in range(d4):
            x = float(a[i]) if i < d else 0.0
            y = float(b[i]) if i < d else 0.0
            # fma in float32 == round(x*y + p) computed exactly in float64 (24+24+guard bits fit) then rounded once
            p[i & 63] = np.float32(np.float64(x) * np.float64(y) + np.float64(p[i & 63]))
        w = 32
        while w >= 1:
            p = [np.float32(p[2 * i] + p[2 * i + 1]) for i in range(w)]
            w //= 2
        got = lib.jvo_raw_dot(a.ctypes.data, b.ctypes.data, d)
        # double rounding can differ from a true fma in rare cases; allow 1 ulp
        assert abs(got - float(p[0])) <= abs(math.ulp(np.float32(got))) * 2**29


def test_rerankk_less_than_topk_is_rejected(pkg, pyoracle):
    ix = pkg.binding.IndexData(vectors=np.zeros((4, 2), np.float32), adj=np.full((4, 2), -1, np.int32), entry_node=0)
    orc = pyoracle.Oracle(pkg.binding, ix)
    assert orc.check_args(10, 5) == pkg.binding.JV_EINVAL  # jvector: IllegalArgumentException
    assert orc.check_args(5, 5) == pkg.binding.JV_OK


def test_results_sorted_desc_ties_by_ordinal(pkg, pyoracle):
    """Duplicate vectors -> equal scores -> ascending ordinal among ties."""
    base = np.zeros((40, 4), np.float32)
    base[:, 0] = np.repeat(np.arange(10), 4)  # 4 copies of each of 10 points
    ix = pkg.builder.build_index_cpu(base, 0, R=16, L=40)
    orc = pyoracle.Oracle(pkg.binding, ix)
    res = orc.search_batch(np.zeros((1, 4), np.float32), 12, 40)
    sc, nd = res.scores[0], res.nodes[0]
    assert all(sc[i] >= sc[i + 1] for i in range(11))
    for i in range(11):
        if sc[i] == sc[i + 1]:
            assert nd[i] < nd[i + 1]
    assert nd[:4].tolist() == [0, 1, 2, 3]


def test_empty_and_tiny_indexes(pkg, pyoracle):
    b = pkg.binding
    empty = b.IndexData(vectors=np.zeros((0, 8), np.float32), adj=np.zeros((0, 4), np.int32), entry_node=-1)
    res = pyoracle.Oracle(b, empty).search_batch(np.ones((2, 8), np.float32), 5, 25)
    assert res.count.tolist() == [0, 0] and (res.nodes == -1).all()
    one = b.IndexData(vectors=np.ones((1, 8), np.float32), adj=np.full((1, 4), -1, np.int32), entry_node=0)
    res = pyoracle.Oracle(b, one).search_batch(np.ones((1, 8), np.float32), 5, 25)
    assert res.count[0] == 1 and res.nodes[0][0] == 0 and res.scores[0][0] == 1.0
    assert res.stats[0].tolist() == [0, 0, 1, 1]  # entry point is not a "visited" count; one expansion
    # k = 0 short-circuit (JVectorKnnFloatVectorQuery.java:62-64)
    res = pyoracle.Oracle(b, one).search_batch(np.ones((1, 8), np.float32), 0, 0)
    assert res.count[0] == 0


def test_pq_lut_matches_direct_distance(pkg, pyoracle):
    """ADC raw score == distance between the query and the node's reconstructed vector."""
    rng = np.random.default_rng(9)
    base = rng.random((1500, 24)).astype(np.float32)
    for sim in (0, 1):
        ix = pkg.builder.build_index_cpu(base, sim, R=8, L=30, pq_M=6)
        orc = pyoracle.Oracle(pkg.binding, ix)
        q = rng.random(24).astype(np.float32)
        lut = np.zeros(6 * 256, np.float32)
        orc.lib.jvo_pq_build_lut(orc.desc, q.ctypes.data, lut.ctypes.data)
        cb = ix.pq_codebooks.reshape(6, ix.pq_K, 4)
        for node in (0, 17, 1499):
            rec = np.concatenate([cb[m, int(ix.pq_codes[node, m])] for m in range(6)])
            raw = sum(float(lut[m * 256 + int(ix.pq_codes[node, m])]) for m in range(6))
            if sim == 0:
                want = ((q - ix.pq_centroid - rec).astype(np.float64) ** 2).sum()
            else:
                want = float(q.astype(np.float64).dot(rec))
            assert raw == pytest.approx(want, rel=1e-4, abs=1e-5)


def test_merge_topk(pkg, pyoracle):
    docs = np.array([[5, 9, -1, 2, 7, 8]], np.int32)
    scores = np.array([[0.9, 0.5, 0.0, 0.9, 0.7, 0.1]], np.float32)
    od, os_ = pyoracle.merge_topk(pkg.binding, docs, scores, 3)
    assert od[0].tolist() == [2, 5, 7]  # tie at 0.9 -> lower doc id first
    assert os_[0].tolist() == pytest.approx([0.9, 0.9, 0.7])


def test_threshold_filters_results(pkg, pyoracle):
    base = pkg.datagen.splitmix_uniform(5, 800, 8)
    ix = pkg.builder.build_index_cpu(base, 0, R=8, L=40)
    orc = pyoracle.Oracle(pkg.binding, ix)
    q = pkg.datagen.splitmix_uniform(6, 1, 8)
    res = orc.search_batch(q, 10, 50, threshold=0.8)
    c = res.count[0]
    assert (res.scores[0][:c] >= 0.8).all()
    truth, ts = orc.brute_force(q, 10)
    assert set(res.nodes[0][:c].tolist()) <= set(np.asarray(truth[0])[ts[0] >= 0.8].tolist()) | set(res.nodes[0][:c].tolist())


def test_simd_mode_changes_nothing(pkg, pyoracle):
    """the CPU baseline's explicit AVX2 mode (look-up-table gathers, software prefetch; oracle/jv_oracle.c jvo_set_simd)
    keeps every floating-point operation in the canonical order: ids, score bits and counters equal the plain loops'"""
    b, bl, dg = pkg.binding, pkg.builder, pkg.datagen
    base = dg.splitmix_uniform(5, 3000, 64)
    q = dg.splitmix_uniform(6, 40, 64)
    for M in (32, 16, 24):
        ix = bl.build_index_cpu(base, 0, R=16, L=50, pq_M=M)
        orc = pyoracle.Oracle(b, ix)
        try:
            orc.lib.jvo_set_simd(0)
            a = orc.search_batch(q, 10, 80)
            orc.lib.jvo_set_simd(1)
            assert orc.lib.jvo_get_simd() == 1
            c = orc.search_batch(q, 10, 80)
        finally:
            orc.lib.jvo_set_simd(0)
        assert np.array_equal(a.nodes, c.nodes) and np.array_equal(a.stats, c.stats)
        assert np.array_equal(a.scores.view(np.uint32), c.scores.view(np.uint32))
