"""The batched exact scorer (jv_score_ordinals_batch, ABI v4): B queries against ONE shared candidate set.

Replaces the batched form of Lucene's exact fallback (AbstractKnnVectorQuery.exactSearch over JVectorVectorScorer.score,
J/JVectorVectorScorer.java:36-53, reached through J/JVectorReader.java:202-207).  Bar: ids, order (score desc, doc asc —
Lucene's HitQueue) and score BITS equal the oracle's jvo_score_ordinals + a sort, whatever the bf16 matrix-core pass
discarded on the way; and the interval that pass computes must contain the exact value for every pair."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _index(b, base, sim, scale=1.0, ord2doc=None, max_doc=0):
    n = base.shape[0]
    ix = b.IndexData(vectors=base, adj=np.full((n, 4), -1, dtype=np.int32), entry_node=0, similarity=sim, score_scale=scale)
    if ord2doc is not None:
        ix.ord2doc = ord2doc.astype(np.int32)
        ix.max_doc = int(max_doc)
    return ix


def _want(orc, q, ords, docs_of, k):
    """the oracle's exact scan: jvo_score_ordinals + (score desc, doc asc)"""
    sc = orc.score_ordinals(q, ords)
    docs = docs_of[ords]
    order = np.lexsort((docs, -sc.astype(np.float64)))[:k]
    # -score as float64 is exact for float32 inputs; equal float32 scores stay equal
    return ords[order], docs[order], sc[order]


def _check(got, orc, queries, ords, docs_of, k, what):
    nodes, docs, scores, count, info = got
    for i in range(queries.shape[0]):
        wn, wd, ws = _want(orc, queries[i], ords, docs_of, k)
        m = len(wn)
        assert count[i] == m, (what, i, count[i], m)
        assert np.array_equal(docs[i, :m], wd), f"{what}: docs of query {i} differ from the oracle's exact scan"
        assert np.array_equal(nodes[i, :m], wn), f"{what}: ordinals of query {i} differ"
        assert np.array_equal(scores[i, :m].view(np.uint32), ws.view(np.uint32)), f"{what}: score bits of query {i} differ"
        assert (nodes[i, m:] == -1).all() and (docs[i, m:] == -1).all() and (scores[i, m:] == 0).all()


@pytest.mark.parametrize("d,sim,scale", [(64, 0, 1.0), (128, 1, 1.0), (128, 1, 2.0), (100, 2, 1.0), (768, 0, 1.0), (70, 1, 1.0)])
def test_batched_exact_scorer_equals_the_oracle_scan(pkg, pyoracle, d, sim, scale):
    """doc filters of several selectivities over a permuted, sparse doc-id space with deleted ordinals; explicit ordinal lists
    (with invalid entries); no filter at all; 1 ... 300 queries; topK 1 ... 100; with and without the matrix-core pass."""
    b = pkg.binding
    rng = np.random.default_rng(d * 10 + sim)
    n = 30000
    base = pkg.datagen.splitmix_uniform(100 + d, n, d)
    if sim != 0:
        base = base - np.float32(0.5)
    max_doc = n + n // 2
    perm = rng.permutation(max_doc)[:n].astype(np.int32)
    dead = rng.random(n) < 0.03
    ord2doc = np.where(dead, -1, perm).astype(np.int32)
    ix = _index(b, base, sim, scale, ord2doc, max_doc)
    gpu = b.GpuIndex(ix)
    orc = pyoracle.Oracle(b, ix)
    docs_of = ord2doc
    queries = pkg.datagen.splitmix_uniform(7, 300, d)
    if sim != 0:
        queries = queries - np.float32(0.5)
    for sel, nq, k in [(0.5, 37, 10), (0.1, 300, 10), (0.1, 5, 100), (0.01, 130, 1), (0.003, 64, 100)]:
        acc_docs = np.nonzero(rng.random(max_doc) < sel)[0]
        words = b.accept_words(acc_docs, max_doc)
        accepted = np.zeros(max_doc, dtype=bool)
        accepted[acc_docs] = True
        ords = np.nonzero((ord2doc >= 0) & accepted[np.maximum(ord2doc, 0)])[0].astype(np.int32)
        got = gpu.score_ordinals_batch(queries[:nq], k, accept=words, accept_num_docs=max_doc)
        assert got[4][0] == len(ords), "candidate count = accepted live ordinals"
        if len(ords) >= 2048:
            assert got[4][1] > 0, "the matrix-core pass must have run"
        _check(got, orc, queries[:nq], ords, docs_of, k, f"filter {sel}")
        plain = gpu.score_ordinals_batch(queries[:nq], k, accept=words, accept_num_docs=max_doc, flags=b.XB_NO_PREFILTER)
        assert plain[4][1] == 0
        for a, c in zip(got[:4], plain[:4]):
            assert np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, c.view(np.uint32) if c.dtype == np.float32 else c)
    # explicit list with invalid entries and a deleted ordinal
    lst = np.concatenate([rng.choice(n, 5000, replace=False), [-1, n + 5, int(np.nonzero(dead)[0][0])]]).astype(np.int32)
    valid = lst[(lst >= 0) & (lst < n)]
    valid = valid[ord2doc[valid] >= 0]
    got = gpu.score_ordinals_batch(queries[:20], 10, ordinals=lst)
    _check(got, orc, queries[:20], valid, docs_of, 10, "explicit list")
    # neither: every live ordinal
    live = np.nonzero(ord2doc >= 0)[0].astype(np.int32)
    got = gpu.score_ordinals_batch(queries[:9], 10)
    _check(got, orc, queries[:9], live, docs_of, 10, "all ordinals")
    gpu.close()


def test_the_bound_contains_the_exact_value(pkg):
    """[lower, upper] of the bf16 pass against float64 for every pair: uniform, centred, badly scaled and near-duplicate rows —
    the property the pre-filter's correctness rests on — and not vacuously (width <= 2.2 kappa |q||c| (x2 for L2) + slack)."""
    b = pkg.binding
    lib = b.load_library()
    lib.jv_xb_debug_bounds.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.jv_xb_debug_bounds.restype = C.c_int
    rng = np.random.default_rng(5)
    for d in (64, 128, 200, 500, 768, 1536):
        n, nq = 4096, 130
        base = rng.standard_normal((n, d)).astype(np.float32)
        base[:512] = np.abs(base[:512]) * 3 + 1                      # far from the origin: worst case for the L2 cancellation
        base[512:1024] *= np.float32(1e-3)
        base[1024:1536] *= np.float32(1e3)
        base[1536:1600] = base[1536] + rng.standard_normal((64, d)).astype(np.float32) * np.float32(1e-4)   # near-duplicates
        q = rng.standard_normal((nq, d)).astype(np.float32)
        q[:16] = base[1536:1552] + np.float32(1e-5)
        q[16:32] = np.abs(q[16:32]) * 3 + 1
        ords = rng.permutation(n)[:3000].astype(np.int32)
        for sim in (0, 1, 2):
            gpu = b.GpuIndex(_index(b, base, sim))
            lo = np.zeros((nq, len(ords)), dtype=np.float32)
            hi = np.zeros_like(lo)
            kappa = C.c_float(0)
            rc = lib.jv_xb_debug_bounds(gpu.handle, q.ctypes.data, nq, ords.ctypes.data, len(ords), lo.ctypes.data, hi.ctypes.data, C.byref(kappa))
            assert rc == 0, lib.jv_last_error()
            q64, c64 = q.astype(np.float64), base[ords].astype(np.float64)
            dots = q64 @ c64.T
            qn, cn = np.linalg.norm(q64, axis=1)[:, None], np.linalg.norm(c64, axis=1)[None, :]
            if sim == 0:
                true = -(qn ** 2 + cn ** 2 - 2 * dots)
                width = 2 * 2.2 * kappa.value * qn * cn + 3e-5 * (qn ** 2 + cn ** 2) + 1e-5
            elif sim == 1:
                true = dots
                width = 2.2 * kappa.value * qn * cn + 1e-5
            else:
                true = dots / (qn * cn)
                width = 2.2 * kappa.value + 1e-4
            assert (lo <= true).all(), (d, sim, float((lo - true).max()))
            assert (hi >= true).all(), (d, sim, float((true - hi).max()))
            assert ((hi.astype(np.float64) - lo) <= width * 1.05 + 1e-6 * np.abs(true)).all(), (d, sim)
            # how much of the interval the bf16 error actually uses (printed with -s; the bound must hold, not be tight)
            mid = (hi.astype(np.float64) + lo) / 2
            half = (hi.astype(np.float64) - lo) / 2
            used = float((np.abs(mid - true) / np.maximum(half, 1e-30)).max())
            print(f"d={d} sim={sim}: worst |mid - true| / half-width = {used:.3f}")
            assert used <= 1.0
            gpu.close()


def test_ties_and_duplicates_keep_lucenes_order(pkg, pyoracle):
    """grid-valued vectors: hundreds of candidates share a score exactly; among equal scores the LOWER DOC wins (HitQueue),
    not the lower ordinal — the doc ids are a permutation — and the k-th / (k+1)-th tie must survive the pre-filter."""
    b = pkg.binding
    rng = np.random.default_rng(3)
    n, d = 20000, 32
    base = rng.integers(0, 3, size=(n, d)).astype(np.float32)
    base[5000:5200] = base[5000]                       # 200 exact duplicates
    ord2doc = rng.permutation(n).astype(np.int32)
    queries = np.concatenate([base[5000:5004], rng.integers(0, 3, size=(60, d)).astype(np.float32)])
    for sim in (0, 1, 2):
        ix = _index(b, base, sim, 1.0, ord2doc, n)
        gpu = b.GpuIndex(ix)
        orc = pyoracle.Oracle(b, ix)
        words = b.accept_words(np.arange(0, n, 2), n)
        ords = np.nonzero(ord2doc % 2 == 0)[0].astype(np.int32)
        for k in (1, 10, 150):
            got = gpu.score_ordinals_batch(queries, k, accept=words, accept_num_docs=n)
            _check(got, orc, queries, ords, ord2doc, k, f"ties sim={sim} k={k}")
        gpu.close()


def test_a_survivor_overflow_falls_back_to_the_full_scan(pkg, pyoracle):
    """every candidate at (almost) the same distance: the bound cannot separate them, every query's survivor list
    overflows and the canonical scan of the whole list answers — identically."""
    b = pkg.binding
    rng = np.random.default_rng(9)
    n, d = 24000, 64
    base = np.ones((n, d), dtype=np.float32) + rng.standard_normal((n, d)).astype(np.float32) * np.float32(1e-4)
    ix = _index(b, base, 0)
    gpu = b.GpuIndex(ix)
    orc = pyoracle.Oracle(b, ix)
    queries = np.ones((6, d), dtype=np.float32) * np.float32(1.5)
    ords = np.arange(n, dtype=np.int32)
    got = gpu.score_ordinals_batch(queries, 10)
    assert got[4][1] > 0 and got[4][3] == 6, got[4]
    _check(got, orc, queries, ords, ords, 10, "overflow")
    gpu.close()


def test_device_pointer_form_and_argument_errors(pkg, pyoracle):
    import torch
    b = pkg.binding
    dev = torch.device("cuda", 0)
    n, d, nq, k = 50000, 128, 256, 10
    base = pkg.datagen.splitmix_uniform(21, n, d)
    ix = _index(b, base, 0)
    gpu = b.GpuIndex(ix)
    orc = pyoracle.Oracle(b, ix)
    queries = pkg.datagen.splitmix_uniform(22, nq, d)
    acc = np.arange(0, n, 7)
    words = b.accept_words(acc, n)
    tq = torch.from_numpy(queries).to(dev)
    tw = torch.from_numpy(words.view(np.int64)).to(dev)
    o_nodes = torch.empty((nq, k), dtype=torch.int32, device=dev)
    o_docs = torch.empty((nq, k), dtype=torch.int32, device=dev)
    o_scores = torch.empty((nq, k), dtype=torch.float32, device=dev)
    o_count = torch.empty((nq,), dtype=torch.int32, device=dev)
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        tq2 = tq * 1.0   # work on the caller's stream the call must be ordered behind
        info = gpu.score_ordinals_batch_device(tq2.data_ptr(), nq, k, o_nodes.data_ptr(), o_docs.data_ptr(), o_scores.data_ptr(),
                                               o_count.data_ptr(), d_accept=tw.data_ptr(), accept_num_docs=n, stream=s.cuda_stream,
                                               want_info=True)
        nodes = o_nodes.cpu().numpy()   # (on the same stream: ordered behind the library's work)
        docs, scores, count = o_docs.cpu().numpy(), o_scores.cpu().numpy(), o_count.cpu().numpy()
    assert info[0] == len(acc) and info[1] > 0
    _check((nodes, docs, scores, count, info), orc, queries[:40], acc.astype(np.int32), np.arange(n, dtype=np.int32), k, "device form")
    host = gpu.score_ordinals_batch(queries, k, accept=words, accept_num_docs=n)
    assert np.array_equal(host[0], nodes) and np.array_equal(host[2].view(np.uint32), scores.view(np.uint32))
    # the explicit-list device form does not wait for the host
    tl = torch.from_numpy(acc.astype(np.int32)).to(dev)
    gpu.score_ordinals_batch_device(tq.data_ptr(), nq, k, o_nodes.data_ptr(), o_docs.data_ptr(), o_scores.data_ptr(), o_count.data_ptr(),
                                    d_ordinals=tl.data_ptr(), count=len(acc), stream=torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(o_nodes.cpu().numpy(), nodes)
    for kk, code in ((0, b.JV_EINVAL), (b.XB_TOPK_MAX + 1, b.JV_EUNSUPPORTED)):
        with pytest.raises(b.JvError) as e:
            gpu.score_ordinals_batch(queries[:2], kk)
        assert e.value.code == code
    # fewer candidates than topK: short rows
    got = gpu.score_ordinals_batch(queries[:3], 50, ordinals=np.arange(20, dtype=np.int32))
    assert (got[3] == 20).all() and (got[0][:, 20:] == -1).all()
    got = gpu.score_ordinals_batch(queries[:3], 5, ordinals=np.zeros(0, dtype=np.int32))
    assert (got[3] == 0).all()
    gpu.close()


def test_two_million_docs_three_selectivities(pkg, pyoracle):
    """>= 2M docs, selectivity 0.001 / 0.01 / 0.1, L2 / dot / cosine incl. score_scale 2 (Lucene MIP), 256
    queries under one filter.  The oracle's scan checks a sample of the queries at every point; all 256 are checked against
    the engine's own fp32 scan of the whole list (the arithmetic the oracle sample pins)."""
    import torch
    b = pkg.binding
    dev = torch.device("cuda", 0)
    n, d, nq, k = 2_000_000, 768, 256, 10
    g = torch.Generator(device=dev)
    g.manual_seed(42)
    cen = torch.randn((2048, d), generator=g, device=dev)
    base_t = cen[torch.randint(0, 2048, (n,), generator=g, device=dev)] + 0.35 * torch.randn((n, d), generator=g, device=dev)
    q_t = cen[torch.randint(0, 2048, (nq,), generator=g, device=dev)] + 0.35 * torch.randn((nq, d), generator=g, device=dev)
    base = base_t.cpu().numpy()
    queries = q_t.cpu().numpy()
    del base_t, q_t, cen
    torch.cuda.empty_cache()
    rng = np.random.default_rng(1)
    ident = np.arange(n, dtype=np.int32)
    for sim, scale in ((0, 1.0), (1, 1.0), (1, 2.0), (2, 1.0)):
        ix = _index(b, base, sim, scale)
        gpu = b.GpuIndex(ix)
        orc = pyoracle.Oracle(b, ix)
        for sel, sample in ((0.001, 24), (0.01, 12), (0.1, 4)):
            acc = np.nonzero(rng.random(n) < sel)[0].astype(np.int32)
            words = b.accept_words(acc, n)
            got = gpu.score_ordinals_batch(queries, k, accept=words, accept_num_docs=n)
            assert got[4][0] == len(acc)
            if len(acc) >= 2048:
                assert got[4][1] > 0 and got[4][3] == 0, got[4]
                assert got[4][2] < 0.05 * nq * len(acc), f"the pre-filter must discard most candidates: {got[4]}"
            _check(tuple(x[:sample] if x.ndim else x for x in got[:4]) + (got[4],), orc, queries[:sample], acc, ident, k,
                   f"2M sim={sim} scale={scale} sel={sel}")
            plain = gpu.score_ordinals_batch(queries, k, accept=words, accept_num_docs=n, flags=b.XB_NO_PREFILTER)
            assert np.array_equal(got[0], plain[0]) and np.array_equal(got[2].view(np.uint32), plain[2].view(np.uint32))
            print(f"sim={sim} scale={scale} sel={sel}: candidates {got[4][0]}, sample {got[4][1]}, re-scored per query {got[4][2] / nq:.0f}")
        gpu.close()


def test_batched_leaf_search_equals_the_single_query_leaf_search_and_the_oracle(pkg, pyoracle):
    """JVectorKnnFloatVectorQuery::searchLeafBatch (host mirror): Lucene's per-leaf rule — cost <= k -> exact; graph search with
    visitLimit = cost; early-terminated -> exact — for 96 queries under ONE filter in two engine calls.  Every query must get
    what the one-query searchLeaf gives it AND what the oracle's leaf search gives it (tests/ka_support.py), at selectivities
    where all, some and none of the graph searches reach the limit; with deletes; and for a field with PQ."""
    import importlib
    import ka_support
    host = importlib.import_module("opensearch_jvector_amd.host")
    b, bl = pkg.binding, pkg.builder
    rng = np.random.default_rng(17)
    n, d, k, oqf, nq = 20000, 64, 10, 5, 96
    base = pkg.datagen.splitmix_uniform(31, n, d)
    queries = pkg.datagen.splitmix_uniform(32, nq, d)
    max_doc = n + 500
    ord2doc = np.sort(rng.permutation(max_doc)[:n]).astype(np.int32)   # docs without a vector in between
    mixed = 0
    for lucene_sim, pq_M in (("EUCLIDEAN", 0), ("MAXIMUM_INNER_PRODUCT", 0), ("EUCLIDEAN", 16)):
        sim, scale = ka_support.LUCENE_SIM[lucene_sim]
        ix = bl.build_index_cpu(base, sim, R=16, L=60, score_scale=(scale if pq_M == 0 else 1.0), ord2doc=ord2doc, max_doc=max_doc, pq_M=pq_M)
        reader = host.JVectorReader(ix, lucene_sim)
        orc = pyoracle.Oracle(b, ix)
        deleted = rng.choice(max_doc, 300, replace=False).tolist()
        for sel, dels in ((0.0004, ()), (0.01, ()), (0.04, deleted), (0.08, ()), (0.15, ()), (0.17, deleted), (0.18, ()), (0.3, deleted),
                          (None, deleted), (None, ())):
            fdocs = None if sel is None else np.nonzero(rng.random(max_doc) < sel)[0]
            docs, scores, count, exact = reader.search_leaf_batch(queries, k, oqf, filter_docs=fdocs, deleted_docs=dels)
            if 0 < exact.sum() < nq:
                mixed += 1
            for i in range(nq):
                one = reader.search_leaf(queries[i], k, oqf, filter_docs=fdocs, deleted_docs=dels)
                c = int(count[i])
                assert docs[i, :c].tolist() == one[0] and bool(exact[i]) == one[3], (lucene_sim, pq_M, sel, i)
                assert np.array_equal(scores[i, :c].view(np.uint32), np.asarray(one[1], dtype=np.float32).view(np.uint32))
                if i < 24:
                    case = dict(k=k, over_query_factor=oqf, deleted_docs=list(dels), filter_docs=None if fdocs is None else fdocs.tolist(),
                                query=queries[i].tolist())
                    wd, ws, wex = ka_support.leaf_search(pkg, orc, ix, case)
                    assert docs[i, :c].tolist() == wd and bool(exact[i]) == wex, (lucene_sim, pq_M, sel, i)
                    assert np.array_equal(scores[i, :c].view(np.uint32), np.asarray(ws, dtype=np.float32).view(np.uint32))
            # the opt-in crossover answers with the exact top k whenever the filter is selective enough
            if sel is not None and sel <= 0.08:
                xd, xs, xc, xe = reader.search_leaf_batch(queries, k, oqf, filter_docs=fdocs, deleted_docs=dels, exact_when_cheaper=True,
                                                          crossover_selectivity=0.1)
                assert xe.all()
                for i in np.nonzero(exact)[0]:
                    assert np.array_equal(xd[i], docs[i]) and np.array_equal(xs[i].view(np.uint32), scores[i].view(np.uint32))
        reader.close()
    assert mixed >= 1, "no selectivity exercised the mixed case (some queries exact, some approximate)"


def test_concurrent_one_query_exact_searches_are_combined(pkg, pyoracle):
    """jv_exact_search: Lucene's exactSearch for ONE query, the way the reference issues it — one call per searcher thread
    (T/index/engine/JVectorConcurrentQueryTests.java:78-138).  48 threads under two different filters and two topK values: every
    answer equals the oracle's scan, and the library answered them in fewer engine calls than there were callers (group commit)."""
    import threading
    b = pkg.binding
    rng = np.random.default_rng(23)
    n, d = 60000, 96
    base = pkg.datagen.splitmix_uniform(51, n, d)
    ident = np.arange(n, dtype=np.int32)
    ix = _index(b, base, 0)
    gpu = b.GpuIndex(ix)
    orc = pyoracle.Oracle(b, ix)
    queries = pkg.datagen.splitmix_uniform(52, 48 * 6, d)
    filters = []
    for sel in (0.2, 0.05):
        docs = np.nonzero(rng.random(n) < sel)[0].astype(np.int32)
        filters.append((docs, b.accept_words(docs, n)))
    bad = []

    def caller(t):
        for it in range(6):
            qi = t * 6 + it
            docs, words = filters[(t + it) % 2]
            k = 10 if t % 3 else 25
            nodes, dcs, sc, cnt = gpu.exact_search(queries[qi], k, words, n)
            wn, wd, ws = _want(orc, queries[qi], docs, ident, k)
            if cnt != len(wn) or not np.array_equal(dcs[:cnt], wd) or not np.array_equal(sc[:cnt].view(np.uint32), ws.view(np.uint32)):
                bad.append(qi)

    ts = [threading.Thread(target=caller, args=(t,)) for t in range(48)]
    [t.start() for t in ts]
    [t.join(timeout=120) for t in ts]
    assert not any(t.is_alive() for t in ts), "a caller is stuck"
    assert not bad, bad[:5]
    calls, batches = gpu.counter("exact_calls"), gpu.counter("exact_batches")
    assert calls == 48 * 6 and batches < calls, (calls, batches)
    print(f"{calls} one-query exact searches answered in {batches} engine calls")
    gpu.close()


@pytest.mark.parametrize("sim", [0, 1, 2])
def test_half_the_ordinals_deleted_short_lists_with_the_pre_filter_forced(pkg, pyoracle, sim):
    """the matrix-core pre-filter took every in-range ordinal for a candidate — deleted ones (ord2doc < 0,
    never returned: J/JVectorReader.java:157-163) too.  With a list of 2048 ... 4096 entries the sample IS the list, so the bar
    was the k-th best bound over live AND dead rows while the re-score dropped the dead ones: live true neighbours were lost and
    count < k.  Half the ordinals are deleted here, and the dead ones are the BEST rows for every query (copies of the queries):
    forced pre-filter, no pre-filter and the oracle's scan must agree, on explicit lists and on the identity path."""
    b = pkg.binding
    rng = np.random.default_rng(77 + sim)
    n, d, nq = 9000, 96, 40
    base = pkg.datagen.splitmix_uniform(300 + sim, n, d) - np.float32(0.5)
    queries = pkg.datagen.splitmix_uniform(301 + sim, nq, d) - np.float32(0.5)
    dead = rng.random(n) < 0.5
    dead_idx = np.nonzero(dead)[0]
    # every query has 30 deleted near-duplicates: the dead rows would fill the whole top of any sample that admits them
    for i in range(nq):
        rows = dead_idx[i * 30:(i + 1) * 30]
        base[rows] = queries[i] + np.float32(1e-3) * rng.standard_normal((30, d)).astype(np.float32)
    ord2doc = np.where(dead, -1, np.arange(n)).astype(np.int32)
    ix = _index(b, base, sim, 1.0, ord2doc, n)
    gpu = b.GpuIndex(ix)
    orc = pyoracle.Oracle(b, ix)
    for C in (2048, 3000, 4096):
        lst = rng.choice(n, C, replace=False).astype(np.int32)
        lst[:nq * 30] = dead_idx[:nq * 30]              # (the planted dead rows are on every list)
        valid = lst[ord2doc[lst] >= 0]
        for k in (10, 100):
            forced = gpu.score_ordinals_batch(queries, k, ordinals=lst, flags=b.XB_FORCE_PREFILTER)
            plain = gpu.score_ordinals_batch(queries, k, ordinals=lst, flags=b.XB_NO_PREFILTER)
            assert forced[4][1] > 0 and plain[4][1] == 0
            _check(plain, orc, queries, valid, ord2doc, k, f"no pre-filter C={C} k={k}")
            _check(forced, orc, queries, valid, ord2doc, k, f"forced pre-filter C={C} k={k}")
    live = np.nonzero(~dead)[0].astype(np.int32)
    got = gpu.score_ordinals_batch(queries, 10, flags=b.XB_FORCE_PREFILTER)   # identity path: every ordinal, dead ones included
    assert got[4][1] > 0
    _check(got, orc, queries, live, ord2doc, 10, "identity path")
    gpu.close()


def test_exact_calls_next_to_one_query_traffic(pkg, pyoracle):
    """the batched exact scorer's kernels take a whole CU's LDS, and a live query-server grid holds LDS on every
    CU until it has been idle for serve_idle_ms — under steady one-query jv_search traffic it never is, so an exact call (the
    host mirror's default exactSearch comes through jv_exact_search) waited without a bound while it held the scorer's lock.
    Four searcher threads keep the grid busy (no pauses); two others issue exact searches and batched exact calls: every answer
    is right and no exact call takes longer than a second (they take milliseconds; the old code never returned)."""
    import threading, time
    b, bl = pkg.binding, pkg.builder
    rng = np.random.default_rng(5)
    n, d = 20000, 64
    base = pkg.datagen.splitmix_uniform(61, n, d)
    q = pkg.datagen.splitmix_uniform(62, 256, d)
    ix = bl.build_index_cpu(base, 0, R=32, L=60, pq_M=32)
    gpu = b.GpuIndex(ix, flags=b.DESC_FUSED_ADC)
    orc = pyoracle.Oracle(b, ix)
    ident = np.arange(n, dtype=np.int32)
    docs = np.nonzero(rng.random(n) < 0.3)[0].astype(np.int32)
    words = b.accept_words(docs, n)
    want = gpu.search_batch(q, 10, 120)
    stop, bad, lat = threading.Event(), [], []

    def searcher(t):
        i = t
        while not stop.is_set():
            j = i % len(q)
            i += 5
            r = gpu.search(q[j], 10, 120)
            if not np.array_equal(r.nodes[0], want.nodes[j]):
                bad.append(("search", j))

    def exact(t):
        for it in range(40):
            qi = (t * 40 + it) % len(q)
            t0 = time.perf_counter()
            if it % 2:
                nodes, dcs, sc, cnt = gpu.exact_search(q[qi], 10, words, n)
            else:
                got = gpu.score_ordinals_batch(q[qi:qi + 1], 10, accept=words, accept_num_docs=n, flags=b.XB_FORCE_PREFILTER)
                dcs, sc, cnt = got[1][0], got[2][0], int(got[3][0])
            lat.append((time.perf_counter() - t0) * 1e3)
            wn, wd, ws = _want(orc, q[qi], docs, ident, 10)
            if cnt != len(wn) or not np.array_equal(dcs[:cnt], wd) or not np.array_equal(sc[:cnt].view(np.uint32), ws.view(np.uint32)):
                bad.append(("exact", qi))

    ss = [threading.Thread(target=searcher, args=(t,)) for t in range(4)]
    [t.start() for t in ss]
    t_end = time.time() + 20.0   # (the first one-query call creates the server: ring, logs, launch — not a fixed number of milliseconds)
    while gpu.counter("serve_alive") < 1 and time.time() < t_end:
        time.sleep(0.01)
    assert gpu.counter("serve_alive") >= 1, "the one-query traffic is not served by a resident grid: the fixture tests nothing"
    es = [threading.Thread(target=exact, args=(t,)) for t in range(2)]
    [t.start() for t in es]
    [t.join(timeout=120) for t in es]
    stuck = any(t.is_alive() for t in es)
    stop.set()
    [t.join(timeout=60) for t in ss]
    assert not stuck and not any(t.is_alive() for t in ss), "a caller is stuck"
    assert not bad, bad[:4]
    assert len(lat) == 80 and max(lat) < 1000.0, f"exact calls beside a live grid: max {max(lat):.1f} ms"
    assert gpu.counter("served_queries") > 100
    print(f"exact calls beside one-query traffic: p50 {np.percentile(lat, 50):.2f} ms, max {max(lat):.2f} ms")
    gpu.close()
