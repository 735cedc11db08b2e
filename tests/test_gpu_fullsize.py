"""Parity at BASELINE.json's full sizes, through size-independent properties and an oracle sample.

C2 (1M x 768 dot, fp32), C3 (10M x 768 L2, PQ-32 + rerank), C5 (batch = 256 on the C3 index) and one C4 shard
(12.5M x 1536 L2, PQ-64 + rerank: one of the 8 doc-range shards of 100M) are built on the GPU exactly as bench.py
does (same generators, default distribution).  Checked: every result row is sorted (score desc, ordinal asc),
idempotence (same batch twice), batch == single query, batch-size independence, fused layout == plain layout ==
generic kernel, recall against a GPU brute force, and ids / score bits / counters equal to the CPU oracle on a query
sample.

The oracle sample at these sizes does not copy the 30-77 GB of vectors to the host: a PQ search reads full-precision
rows only in the rerank, and WHICH rows it reranks is decided by the approximate phase alone (codes + graph).  So the
oracle runs twice on a lazily zero-filled host array: pass 1 (topK = rerankK) yields every query's rerank set, those
rows are fetched from HBM into the host array, pass 2 is the real search."""
import importlib
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _setup():
    import torch
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import bench
    gb = importlib.import_module("opensearch_jvector_amd.builder_gpu")
    return torch, bench, gb


def _search(torch, gpu, q, k, rk):
    dev = q.device
    nq = q.shape[0]
    o = dict(nodes=torch.empty((nq, k), dtype=torch.int32, device=dev), docs=torch.empty((nq, k), dtype=torch.int32, device=dev),
             scores=torch.empty((nq, k), dtype=torch.float32, device=dev), count=torch.empty((nq,), dtype=torch.int32, device=dev),
             stats=torch.empty((nq, 4), dtype=torch.int32, device=dev), flags=torch.empty((nq,), dtype=torch.int32, device=dev))
    gpu.search_batch_device(q.data_ptr(), nq, k, rk, o["nodes"].data_ptr(), o["docs"].data_ptr(), o["scores"].data_ptr(),
                            o["count"].data_ptr(), o["stats"].data_ptr(), o["flags"].data_ptr())
    torch.cuda.synchronize()
    return {kk: v.cpu().numpy() for kk, v in o.items()}


def _check_rows_sorted(r, k):
    sc, nd = r["scores"], r["nodes"]
    assert (r["count"] == k).all()
    assert (np.diff(sc, axis=1) <= 0).all(), "scores must be non-increasing"
    ties = np.diff(sc, axis=1) == 0
    assert (np.diff(nd, axis=1)[ties] > 0).all(), "ties must be ordered by ascending ordinal"
    assert ((r["flags"].astype(np.uint32) & np.uint32(0xC0000000)) == 0).all()


def _pq_oracle_sample(torch, b, pyoracle, base, adj, entry, sim, pq, q_np, k, rk):
    """the CPU oracle's answer for q_np on the full-size PQ index (two passes, see the module docstring)"""
    n, d = base.shape
    host_vec = np.zeros((n, d), dtype=np.float32)   # calloc: pages materialise only where rows are written
    ix = b.IndexData(vectors=host_vec, adj=adj.cpu().numpy(), entry_node=entry, similarity=sim)
    ix.pq_codebooks, ix.pq_centroid, ix.pq_codes = pq["codebooks"], pq["centroid"], pq["codes"].cpu().numpy()
    ix.pq_M, ix.pq_K = ix.pq_codes.shape[1], pq["K"]
    orc = pyoracle.Oracle(b, ix)
    assert orc.desc.vectors == host_vec.ctypes.data, "the oracle must read the array that is filled below"
    first = orc.search_batch(q_np, rk, rk)           # topK = rerankK: the whole rerank set of every query
    need = np.unique(first.nodes[first.nodes >= 0])
    rows = base[torch.from_numpy(need.astype(np.int64)).to(base.device)].cpu().numpy()
    host_vec[need] = rows
    want = orc.search_batch(q_np, k, rk)
    assert (want.stats[:, 0] == first.stats[:, 0]).all() and (want.stats[:, 2] == first.stats[:, 2]).all()
    return want


def _assert_sample_equal(r, want, m, what):
    assert np.array_equal(r["nodes"][:m], want.nodes), f"{what}: neighbour ids differ from the oracle"
    assert np.array_equal(r["scores"][:m].view(np.uint32), want.scores.view(np.uint32)), f"{what}: score bits differ"
    assert np.array_equal(r["stats"][:m], want.stats), f"{what}: visited/reranked/expanded counters differ"
    assert np.array_equal(r["count"][:m], want.count), f"{what}: result counts differ"


def test_c2_full_size_1m_768_dot(pkg, pyoracle):
    torch, bench, gb = _setup()
    b = pkg.binding
    dev = torch.device("cuda", 0)
    n, d, k, rk = 1_000_000, 768, 10, 100
    cen, basis = bench.make_generators(torch, d, dev, 4096, 32)
    base = bench.gen_rows(torch, n, d, 42, 0, cen, basis, 0.15, 0.01, True, dev)
    q = bench.gen_rows(torch, 2048, d, 43, 0, cen, basis, 0.15, 0.01, True, dev)
    adj, entry = gb.build_graph_gpu(torch, base, 1, R=32, L=100, verbose=False)
    desc, keep = b.make_desc_device(n, d, 32, base.data_ptr(), adj.data_ptr(), entry, 1, borrow=True)
    gpu = b.GpuIndex(desc=desc, keepalive=keep, flags=b.DESC_BORROW)
    r1 = _search(torch, gpu, q, k, rk)
    r2 = _search(torch, gpu, q, k, rk)
    _check_rows_sorted(r1, k)
    for key in ("nodes", "scores", "stats"):
        assert np.array_equal(r1[key], r2[key]), f"idempotence: {key}"
    one = gpu.search(q[7].cpu().numpy(), k, rk)
    assert np.array_equal(one.nodes[0], r1["nodes"][7]) and np.array_equal(one.stats[0], r1["stats"][7])
    truth = bench.brute_force_topk(torch, base, q[:256], k, 1).cpu().numpy()
    rec = np.mean([len(set(r1["nodes"][i]) & set(truth[i])) / k for i in range(256)])
    assert rec >= 0.93, rec
    # oracle on a sample of the full-size index (3 GB host copy)
    ix = b.IndexData(vectors=base.cpu().numpy(), adj=adj.cpu().numpy(), entry_node=entry, similarity=1)
    want = pyoracle.Oracle(b, ix).search_batch(q[:512].cpu().numpy(), k, rk)
    _assert_sample_equal(r1, want, 512, "C2")
    gpu.close()


@pytest.fixture(scope="module")
def c3(pkg):
    """the C3 index of bench.py's default run: 10M x 768, default distribution, PQ-32, GPU-built graph"""
    torch, bench, gb = _setup()
    b = pkg.binding
    dev = torch.device("cuda", 0)
    n, d, M = 10_000_000, 768, 32
    base, q = bench.make_pq_data(torch, bench.DISTS[0], n, 8192, d, M, 0, n, False, dev)
    adj, entry = gb.build_graph_gpu(torch, base, 0, R=32, L=100, verbose=False)
    pq = gb.pq_train_encode_gpu(torch, base, M, 0)

    def make(flags):
        desc, keep = b.make_desc_device(n, d, 32, base.data_ptr(), adj.data_ptr(), entry, 0, pq_M=M, pq_K=pq["K"],
                                        pq_codebooks=pq["codebooks"], pq_centroid=pq["centroid"],
                                        pq_codes_ptr=pq["codes"].data_ptr(), borrow=True, extra_flags=flags)
        return b.GpuIndex(desc=desc, keepalive=keep, flags=b.DESC_BORROW)

    fused = make(b.DESC_FUSED_ADC)
    truth = bench.brute_force_topk(torch, base, q[:512], 10, 0).cpu().numpy()
    yield dict(torch=torch, bench=bench, b=b, base=base, q=q, adj=adj, entry=entry, pq=pq, make=make, fused=fused, truth=truth,
               n=n, d=d, M=M)
    fused.close()


C3_RK = 1200  # the rerankK bench.py's sweep selects on the default distribution (recall@10 0.953)


def test_c3_full_size_10m_768_pq32(c3, pyoracle):
    torch, b, base, q, fused = c3["torch"], c3["b"], c3["base"], c3["q"], c3["fused"]
    k, rk = 10, C3_RK
    r1 = _search(torch, fused, q[:4096], k, rk)
    _check_rows_sorted(r1, k)
    assert (r1["stats"][:, 1] == rk).all(), "every query exact-rescored rerankK candidates"
    r2 = _search(torch, fused, q[:4096], k, rk)
    for key in ("nodes", "scores", "stats"):
        assert np.array_equal(r1[key], r2[key]), f"idempotence: {key}"
    rec = np.mean([len(set(r1["nodes"][i]) & set(c3["truth"][i])) / k for i in range(512)])
    rec_lo = np.mean([len(set(x) & set(t)) / k for x, t in zip(_search(torch, fused, q[:512], k, 60)["nodes"], c3["truth"])])
    assert rec >= 0.93 and rec > rec_lo, (rec, rec_lo)
    # exact-rerank scores are true L2 similarities of the returned ids
    ids = torch.from_numpy(r1["nodes"][:64].astype(np.int64)).to(base.device)
    d2 = ((q[:64, None, :].double() - base[ids].double()) ** 2).sum(-1)
    np.testing.assert_allclose(r1["scores"][:64], (1.0 / (1.0 + d2)).cpu().numpy(), rtol=1e-4)
    # ids, score bits and counters of the CPU oracle on a 512-query sample of the FULL-SIZE index
    want = _pq_oracle_sample(torch, b, pyoracle, base, c3["adj"], c3["entry"], 0, c3["pq"], q[:512].cpu().numpy(), k, rk)
    _assert_sample_equal(r1, want, 512, "C3 10M")
    # the specialised kernel, the generic pool kernel on the fused layout, and the plain layout agree bit for bit
    try:
        fused.set_option("no_pqf", 1)
        r3 = _search(torch, fused, q[:1024], k, rk)
    finally:
        fused.set_option("no_pqf", 0)
    plain = c3["make"](0)
    r4 = _search(torch, plain, q[:1024], k, rk)
    plain.close()
    for key in ("nodes", "scores", "stats"):
        assert np.array_equal(r1[key][:1024], r3[key]), f"pqf vs generic: {key}"
        assert np.array_equal(r1[key][:1024], r4[key]), f"fused vs plain layout: {key}"


def test_c5_batch_256_on_the_c3_index(c3, pyoracle):
    """BASELINE.json configs[4]: 256 concurrent queries per step on the C3 index.  Small launches take the engine's
    few-query path; its answers must equal the large-batch launch's and the oracle's (ids, score bits, counters)."""
    torch, b, base, q, fused = c3["torch"], c3["b"], c3["base"], c3["q"], c3["fused"]
    k, rk = 10, C3_RK
    big = _search(torch, fused, q[:4096], k, rk)
    want = _pq_oracle_sample(torch, b, pyoracle, base, c3["adj"], c3["entry"], 0, c3["pq"], q[:256].cpu().numpy(), k, rk)
    for lo in (0, 256, 3840):
        r = _search(torch, fused, q[lo:lo + 256], k, rk)
        _check_rows_sorted(r, k)
        for key in ("nodes", "scores", "stats", "count"):
            assert np.array_equal(r[key], big[key][lo:lo + 256]), f"batch=256 at {lo} vs batch=4096: {key}"
        if lo == 0:
            _assert_sample_equal(r, want, 256, "C5 batch=256")
    for nq in (1, 2, 7, 64):   # and the other small shapes of that path
        r = _search(torch, fused, q[:nq], k, rk)
        for key in ("nodes", "scores", "stats", "count"):
            assert np.array_equal(r[key], big[key][:nq]), f"batch={nq}: {key}"
    one = fused.search(q[5].cpu().numpy(), k, rk)   # host-pointer single-query API
    assert np.array_equal(one.nodes[0], big["nodes"][5]) and np.array_equal(one.stats[0], big["stats"][5])
    assert np.array_equal(one.scores[0].view(np.uint32), big["scores"][5].view(np.uint32))


def test_c4_shard_12m5_1536_pq64(pkg, pyoracle):
    """One doc-range shard of BASELINE.json configs[3] (100M x 1536, PQ-64, 8 shards): 12.5M docs, ordinals mapped to
    the shard's GLOBAL doc ids (shard 3 of 8), two-pass fused blocks (R * lanes-per-node = 128)."""
    torch, bench, gb = _setup()
    b = pkg.binding
    dev = torch.device("cuda", 0)
    n, d, M, k, rk = 12_500_000, 1536, 64, 10, 800
    shard, n_total = 3, 100_000_000
    row_offset = shard * n
    base, q = bench.make_pq_data(torch, bench.DISTS[0], n, 2048, d, M, row_offset, n_total, False, dev)
    adj, entry = gb.build_graph_gpu(torch, base, 0, R=32, L=100, verbose=False)
    pq = gb.pq_train_encode_gpu(torch, base, M, 0)
    ord2doc = torch.arange(n, device=dev, dtype=torch.int32) + row_offset
    desc, keep = b.make_desc_device(n, d, 32, base.data_ptr(), adj.data_ptr(), entry, 0, pq_M=M, pq_K=pq["K"],
                                    pq_codebooks=pq["codebooks"], pq_centroid=pq["centroid"],
                                    pq_codes_ptr=pq["codes"].data_ptr(), ord2doc_ptr=ord2doc.data_ptr(), max_doc=n_total,
                                    borrow=True, extra_flags=b.DESC_FUSED_ADC)
    gpu = b.GpuIndex(desc=desc, keepalive=keep, flags=b.DESC_BORROW)
    r1 = _search(torch, gpu, q, k, rk)
    _check_rows_sorted(r1, k)
    assert (r1["stats"][:, 1] == rk).all()
    assert np.array_equal(r1["docs"], r1["nodes"] + row_offset), "docs are the shard's global doc ids"
    r2 = _search(torch, gpu, q, k, rk)
    for key in ("nodes", "scores", "stats"):
        assert np.array_equal(r1[key], r2[key]), f"idempotence: {key}"
    truth = bench.brute_force_topk(torch, base, q[:512], k, 0).cpu().numpy()
    rec = np.mean([len(set(r1["nodes"][i]) & set(truth[i])) / k for i in range(512)])
    assert rec >= 0.85, rec   # (0.907 measured at rerankK 800, tools/c4_sweep.py)
    # the metric's recall bar on this shard: rerankK 1200 -> 0.9535, 1600 -> 0.9746 (tools/c4_sweep.py, 512 queries); the
    # beam that clears 0.95 with a margin is asserted (PQ-64 on 1536 rotated dims loses about as much as PQ-32 on 768)
    r16 = _search(torch, gpu, q[:512], k, 1600)
    rec16 = np.mean([len(set(r16["nodes"][i]) & set(truth[i])) / k for i in range(512)])
    assert rec16 >= 0.95, rec16
    assert gpu.counter("launches_pqw") > 0, "PQ-64 runs on the four-waves-per-query kernel"
    ids = torch.from_numpy(r1["nodes"][:32].astype(np.int64)).to(dev)
    d2 = ((q[:32, None, :].double() - base[ids].double()) ** 2).sum(-1)
    np.testing.assert_allclose(r1["scores"][:32], (1.0 / (1.0 + d2)).cpu().numpy(), rtol=1e-4)
    want = _pq_oracle_sample(torch, b, pyoracle, base, adj, entry, 0, pq, q[:256].cpu().numpy(), k, rk)
    _assert_sample_equal(r1, want, 256, "C4 shard")
    r256 = _search(torch, gpu, q[:256], k, rk)
    for key in ("nodes", "scores", "stats"):
        assert np.array_equal(r256[key], r1[key][:256]), f"batch=256 vs batch=2048: {key}"
    gpu.close()


def test_mixture_b_recall_is_limited_by_the_codes_not_by_the_graph(pkg):
    """SURVEY 8(d) distribution B (4 096-centre Gaussian mixture, full rank inside a cluster): 32-byte PQ cannot rank it
    (bench.py reports recall 0.10 at rerankK 900 on 10M docs).  The same graph searched with the EXACT provider reaches
    the bar at a small beam, so the loss is the codes', not the traversal's (VERDICT r2 #4b): 1M docs, recall@10 >= 0.95 at
    rerankK 400 without PQ (0.975 measured), < 0.7 with PQ-32 at the same beam (0.48 measured)."""
    torch, bench, gb = _setup()
    b = pkg.binding
    dev = torch.device("cuda", 0)
    n, d, M, k, B = 1_000_000, 768, 32, 10, 1024
    base, q = bench.make_pq_data(torch, "mixtureB", n, B, d, M, 0, n, False, dev)
    adj, entry = gb.build_graph_gpu(torch, base, 0, R=32, L=100, verbose=False)
    truth = bench.brute_force_topk(torch, base, q, k, 0).cpu().numpy()
    desc, keep = b.make_desc_device(n, d, 32, base.data_ptr(), adj.data_ptr(), entry, 0, borrow=True)
    exact = b.GpuIndex(desc=desc, keepalive=keep, flags=b.DESC_BORROW)
    r = _search(torch, exact, q, k, 400)
    rec_exact = np.mean([len(set(r["nodes"][i]) & set(truth[i])) / k for i in range(B)])
    exact.close()
    pq = gb.pq_train_encode_gpu(torch, base, M, 0)
    desc, keep = b.make_desc_device(n, d, 32, base.data_ptr(), adj.data_ptr(), entry, 0, pq_M=M, pq_K=pq["K"], pq_codebooks=pq["codebooks"],
                                    pq_centroid=pq["centroid"], pq_codes_ptr=pq["codes"].data_ptr(), borrow=True, extra_flags=b.DESC_FUSED_ADC)
    fused = b.GpuIndex(desc=desc, keepalive=keep, flags=b.DESC_BORROW)
    r = _search(torch, fused, q, k, 400)
    rec_pq = np.mean([len(set(r["nodes"][i]) & set(truth[i])) / k for i in range(B)])
    fused.close()
    assert rec_exact >= 0.95, rec_exact
    assert rec_pq < 0.7, rec_pq
