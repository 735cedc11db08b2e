"""ABI v2 additions through the C ABI on a real GPU: per-index options, visit-limit early termination, per-query
status, the device-side filter cache, the in-process shard group, scratch accounting."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _same(a, b):
    return (np.array_equal(a.nodes, b.nodes) and np.array_equal(a.scores.view(np.uint32), b.scores.view(np.uint32)) and
            np.array_equal(a.stats, b.stats) and np.array_equal(a.count, b.count))


@pytest.fixture(scope="module")
def small(pkg):
    dg, bl = pkg.datagen, pkg.builder
    base = dg.splitmix_uniform(91, 6000, 64)
    q = dg.splitmix_uniform(92, 64, 64)
    return base, q, bl.build_index_cpu(base, 0, R=32, L=80, pq_M=16), bl.build_index_cpu(base, 0, R=16, L=60)


def test_options_are_per_index(pkg, pyoracle, small):
    """jv_index_set_option changes one handle only; jv_set_option only seeds indexes created afterwards."""
    b = pkg.binding
    base, q, ixq, ix = small
    want = pyoracle.Oracle(b, ixq).search_batch(q, 10, 50)
    a = b.GpuIndex(ixq, flags=b.DESC_FUSED_ADC)
    c = b.GpuIndex(ixq, flags=b.DESC_FUSED_ADC)
    a.set_option("force_big_path", 1)             # handle `a` takes the HBM-scratch rung, `c` is untouched
    ra, fa = a.search_batch_ex(q, 10, 50)[0], a.search_batch_ex(q, 10, 50)[2]
    rc_, fc = c.search_batch_ex(q, 10, 50)[0], c.search_batch_ex(q, 10, 50)[2]
    assert _same(ra, want) and _same(rc_, want)
    assert (fa & b.QFLAG_RETRIED_BIG).all() and not (fc & b.QFLAG_RETRIED_BIG).any()
    try:
        b.set_option("force_big_path", 1)         # default for NEW handles
        d = b.GpuIndex(ixq, flags=b.DESC_FUSED_ADC)
    finally:
        b.set_option("force_big_path", 0)
    assert (d.search_batch_ex(q, 10, 50)[2] & b.QFLAG_RETRIED_BIG).all()
    assert not (c.search_batch_ex(q, 10, 50)[2] & b.QFLAG_RETRIED_BIG).any()
    with pytest.raises(b.JvError):
        a.set_option("no_such_option", 1)
    for g in (a, c, d):
        g.close()


def test_visit_limit_stops_searches_lucene_would_discard(pkg, pyoracle, small):
    """visit_limit = Lucene's KnnCollector.visitLimit(): a search whose visited + expanded stays below it is unchanged;
    one that reaches it stops early with JV_QFLAG_EARLY_TERMINATED and counters that already reach the limit (so the
    collector's earlyTerminated() is true and AbstractKnnVectorQuery runs the exact scan), never with a wrong answer."""
    b = pkg.binding
    base, q, ixq, ix = small
    n = base.shape[0]
    rng = np.random.default_rng(3)
    words = b.accept_words(np.nonzero(rng.random(n) < 0.3)[0], n)
    for data, flags in ((ix, 0), (ixq, 0), (ixq, b.DESC_FUSED_ADC)):
        gpu = b.GpuIndex(data, flags=flags)
        orc = pyoracle.Oracle(b, data)
        for acc in (None, words):
            kw = dict(accept=acc, accept_num_docs=(n if acc is not None else 0))
            want = orc.search_batch(q, 10, 60, **kw)
            work = want.stats[:, 0] + want.stats[:, 2]
            # the generic kernels know `visited` while searching and stop exactly at the limit; the fused-PQ kernels only
            # count expansions while searching (a later, but still safe, stop): give them a limit expansions alone reach
            for limit, must_fire in ((int(np.median(work)), flags == 0 and data is ix), (int(np.median(want.stats[:, 2])), True)):
                res, status, fl, rc = gpu.search_batch_ex(q, 10, 60, visit_limit=limit, **kw)
                assert rc == b.JV_OK and (status == 0).all()
                early = (fl & b.QFLAG_EARLY_TERMINATED) != 0
                if must_fire:
                    assert early.any(), (limit, int(early.sum()))
                # every search Lucene would discard is flagged, whichever kernel ran it (the fused-PQ kernels test the sum
                # once more after they have counted `visited`)
                assert (work[~early] < limit).all(), (limit, work[~early].max())
                # untouched searches: identical to the oracle
                for i in np.nonzero(~early)[0]:
                    assert np.array_equal(res.nodes[i], want.nodes[i]) and np.array_equal(res.stats[i], want.stats[i])
                    assert np.array_equal(res.scores[i].view(np.uint32), want.scores[i].view(np.uint32))
                # stopped searches: no results, the full search would have reached the limit as well, and the reported
                # counters already reach it (so Lucene's collector is earlyTerminated())
                assert (res.count[early] == 0).all() and (res.nodes[early] == -1).all()
                assert (work[early] >= limit).all()
                assert (res.stats[early, 0] + res.stats[early, 2] >= limit).all()
                one, f1 = gpu.search_ex(q[0], 10, 60, visit_limit=limit, **kw)
                assert bool(f1 & b.QFLAG_EARLY_TERMINATED) == bool(early[0])
        gpu.close()


def test_per_query_status_one_bad_query_does_not_fail_its_batch(pkg, pyoracle, small):
    """A query that outgrows even the HBM scratch is reported alone (out_status / the caller's own return code in a
    combined jv_search batch); every other row of the launch stays valid."""
    import threading
    b = pkg.binding
    base, q, ixq, ix = small
    n = base.shape[0]
    gpu = b.GpuIndex(ix)
    orc = pyoracle.Oracle(b, ix)
    gpu.set_option("force_big_path", 1)
    gpu.set_option("no_escalation", 1)      # straight to the HBM-queue form (the LDS-queue rung would hold these queries)
    rk = 40
    for cap in range(200, 1600, 20):   # find a queue size that part of the batch overflows and part does not
        gpu.set_option("big_cand_cap", cap)
        res, status, fl, rc = gpu.search_batch_ex(q, 10, rk)
        bad = status != 0
        if bad.any() and (~bad).any():
            break
    want = orc.search_batch(q, 10, rk)
    assert rc == b.JV_ENOMEM and bad.any() and (~bad).any(), (rc, rk, int(bad.sum()))
    assert (status[bad] == b.JV_ENOMEM).all()
    for i in np.nonzero(~bad)[0]:
        assert np.array_equal(res.nodes[i], want.nodes[i]) and np.array_equal(res.stats[i], want.stats[i])
    with pytest.raises(b.JvError):
        gpu.search_batch(q, 10, rk)          # the plain call still reports the failure
    # combined single-query calls: only the callers of the bad queries see an error
    outcome = {}

    def worker(i):
        try:
            r = gpu.search(q[i], 10, rk)
            outcome[i] = np.array_equal(r.nodes[0], want.nodes[i])
        except b.JvError as e:
            outcome[i] = e.code
    th = [threading.Thread(target=worker, args=(i,)) for i in range(len(q))]
    [t.start() for t in th]
    [t.join() for t in th]
    for i in range(len(q)):
        assert outcome[i] == (b.JV_ENOMEM if bad[i] else True), (i, outcome[i], bool(bad[i]))
    gpu.close()


def test_filter_cache_serves_repeated_filters_from_hbm(pkg, pyoracle, small):
    b = pkg.binding
    base, q, ixq, ix = small
    n = base.shape[0]
    rng = np.random.default_rng(8)
    gpu = b.GpuIndex(ixq, flags=b.DESC_FUSED_ADC)
    orc = pyoracle.Oracle(b, ixq)
    filters = [b.accept_words(np.nonzero(rng.random(n) < f)[0], n) for f in (0.5, 0.2, 0.7)]
    wants = [orc.search_batch(q, 10, 50, accept=w, accept_num_docs=n) for w in filters]
    for rounds in range(3):
        for w, want in zip(filters, wants):
            assert _same(gpu.search_batch(q, 10, 50, accept=w, accept_num_docs=n), want)
            assert _same(gpu.search(q[3], 10, 50, accept=w.copy(), accept_num_docs=n), b.SearchResult(
                want.nodes[3:4], want.docs[3:4], want.scores[3:4], want.count[3:4], want.stats[3:4]))  # same CONTENTS, other buffer
    info = gpu.info()
    assert info.filter_cache_misses == 3 and info.filter_cache_hits == 3 * 3 * 2 - 3, (info.filter_cache_hits, info.filter_cache_misses)
    # a changed bit is a different filter
    w2 = filters[0].copy()
    w2[5] ^= np.uint64(1 << 7)
    assert _same(gpu.search_batch(q, 10, 50, accept=w2, accept_num_docs=n), orc.search_batch(q, 10, 50, accept=w2, accept_num_docs=n))
    assert gpu.info().filter_cache_misses == 4
    # a caller key only FINDS an entry: the same key with other bits must never be answered under the cached filter
    # (a hit is verified byte for byte — the bitset carries deletes and doc-level security)
    for w, want in ((filters[1], wants[1]), (filters[2], wants[2]), (filters[1], wants[1])):
        got, status, _, rc = gpu.search_batch_ex(q, 10, 50, accept=w, accept_num_docs=n, accept_key=777)
        assert rc == b.JV_OK and (status == 0).all() and _same(got, want)
    gpu.set_option("filter_cache", 0)        # off: every call stages its words
    before = gpu.info()
    assert _same(gpu.search_batch(q, 10, 50, accept=filters[1], accept_num_docs=n), wants[1])
    assert gpu.info().filter_cache_misses == before.filter_cache_misses and gpu.info().filter_cache_hits == before.filter_cache_hits
    gpu.close()


def test_shard_group_in_one_process_equals_oracle_merge(pkg, pyoracle):
    """jv_shard_group: doc-range shards (here 3, all on GPU 0) searched concurrently, (doc, score) pairs gathered with one
    peer copy per shard, merged on the first shard's device — equal to the oracle's merge of the oracle's shard answers."""
    b, bl, dg = pkg.binding, pkg.builder, pkg.datagen
    sh = __import__("importlib").import_module("opensearch_jvector_amd.sharding")
    n_total, d, k, rk, G = 9000, 48, 10, 40, 3
    q = dg.splitmix_uniform(43, 80, d)
    shards, gd, gs, stats = [], [], [], 0
    for g in range(G):
        lo, hi = sh.shard_range(n_total, G, g)
        base = dg.splitmix_uniform(42, hi - lo, d, row_offset=lo)
        ix = bl.build_index_cpu(base, 0, R=16, L=60, pq_M=16, ord2doc=np.arange(lo, hi, dtype=np.int32), max_doc=n_total)
        shards.append(b.GpuIndex(ix, flags=b.DESC_FUSED_ADC))
        w = pyoracle.Oracle(b, ix).search_batch(q, k, rk)
        gd.append(w.docs)
        gs.append(w.scores)
        stats = stats + w.stats
    grp = b.ShardGroup(shards)
    for _ in range(2):
        got = grp.search_batch(q, k, rk)
        od, os_ = pyoracle.merge_topk(b, np.concatenate(gd, axis=1), np.concatenate(gs, axis=1), k)
        assert np.array_equal(got.docs, od) and np.array_equal(got.scores.view(np.uint32), os_.view(np.uint32))
        assert np.array_equal(got.stats, stats) and (got.count == k).all()
    grp.close()
    [s.close() for s in shards]


def test_shard_group_merge_at_the_size_of_a_real_step(pkg, pyoracle):
    """The one-process gather + merge at the size of a benchmark step: 65 536 queries, k = 10, two doc-range shards (both on GPU 0:
    the peer-copy path on one device is all a one-GPU box can run — two distinct devices and the RCCL all-gather have NEVER
    been executed by this repo's builder).  The group's answer must equal the oracle's merge of the shards' own batch answers
    (which the parity suites check against the oracle search)."""
    b, bl, dg = pkg.binding, pkg.builder, pkg.datagen
    sh = __import__("importlib").import_module("opensearch_jvector_amd.sharding")
    n_total, d, k, rk, G, nq = 6000, 32, 10, 30, 2, 65536
    q = dg.splitmix_uniform(45, nq, d)
    shards, gd, gs, stats = [], [], [], 0
    for g in range(G):
        lo, hi = sh.shard_range(n_total, G, g)
        base = dg.splitmix_uniform(44, hi - lo, d, row_offset=lo)
        ix = bl.build_index_cpu(base, 0, R=16, L=50, pq_M=16, ord2doc=np.arange(lo, hi, dtype=np.int32), max_doc=n_total)
        shards.append(b.GpuIndex(ix, flags=b.DESC_FUSED_ADC))
        w = shards[-1].search_batch(q, k, rk)
        gd.append(w.docs)
        gs.append(w.scores)
        stats = stats + w.stats
    grp = b.ShardGroup(shards)
    got = grp.search_batch(q, k, rk)
    od, os_ = pyoracle.merge_topk(b, np.concatenate(gd, axis=1), np.concatenate(gs, axis=1), k)
    assert np.array_equal(got.docs, od) and np.array_equal(got.scores.view(np.uint32), os_.view(np.uint32))
    assert np.array_equal(got.stats, stats) and (got.count == k).all()
    grp.close()
    [s.close() for s in shards]


def test_scratch_is_shared_per_device_and_accounted(pkg, small):
    b = pkg.binding
    base, q, ixq, ix = small
    handles = [b.GpuIndex(ixq, flags=b.DESC_FUSED_ADC) for _ in range(6)]
    for h in handles:
        h.search_batch(q, 10, 50)
    sb = [h.info().scratch_bytes for h in handles]
    # every handle reports its own contexts + the ONE shared HBM-scratch of the device: far below 0.7 GB per context
    assert max(sb) < 300 * 2**20 and min(sb) > 0, sb
    [h.close() for h in handles]


def test_sharded_search_with_global_filter_visit_limit_and_status(pkg, pyoracle):
    """jv_search_sharded_batch_ex: the doc filter is a bitset over the GLOBAL doc ids (every leaf search of the reference gets
    its acceptDocs, J/JVectorReader.java:157-163) -> equal to the oracle's merge of the FILTERED shard answers; the visit limit
    applies per shard search and surfaces as the OR of the shards' EARLY flags; per-query status stays JV_OK."""
    b, bl, dg = pkg.binding, pkg.builder, pkg.datagen
    sh = __import__("importlib").import_module("opensearch_jvector_amd.sharding")
    n_total, d, k, rk, G = 9000, 64, 10, 60, 3
    q = dg.splitmix_uniform(43, 64, d)
    rng = np.random.default_rng(17)
    words = b.accept_words(np.nonzero(rng.random(n_total) < 0.4)[0], n_total)
    shards, orcs = [], []
    for g in range(G):
        lo, hi = sh.shard_range(n_total, G, g)
        base = dg.splitmix_uniform(42, hi - lo, d, row_offset=lo)
        ix = bl.build_index_cpu(base, 0, R=32, L=60, pq_M=32, ord2doc=np.arange(lo, hi, dtype=np.int32), max_doc=n_total)
        shards.append(b.GpuIndex(ix, flags=b.DESC_FUSED_ADC))
        orcs.append(pyoracle.Oracle(b, ix))
    grp = b.ShardGroup(shards)
    for acc in (None, words):
        kw = dict(accept=acc, accept_num_docs=(n_total if acc is not None else 0))
        ws = [o.search_batch(q, k, rk, **kw) for o in orcs]
        od, os_ = pyoracle.merge_topk(b, np.concatenate([w.docs for w in ws], axis=1), np.concatenate([w.scores for w in ws], axis=1), k)
        got, status, flags, rc = grp.search_batch_ex(q, k, rk, **kw)
        assert rc == b.JV_OK and (status == 0).all() and (flags == 0).all()
        assert np.array_equal(got.docs, od) and np.array_equal(got.scores.view(np.uint32), os_.view(np.uint32))
        assert np.array_equal(got.stats, sum(w.stats for w in ws))
        if acc is not None:
            ok = set(np.nonzero(np.unpackbits(acc.view(np.uint8), bitorder="little")[:n_total])[0].tolist())
            assert all(int(x) in ok for x in got.docs.reshape(-1) if x >= 0)
        # visit limit: below every shard search's expansion count -> every query is flagged EARLY by some shard
        lim = int(min(w.stats[:, 2].min() for w in ws))
        got2, status2, flags2, rc2 = grp.search_batch_ex(q, k, rk, visit_limit=max(1, lim - 1), **kw)
        assert rc2 == b.JV_OK and (status2 == 0).all() and ((flags2 & b.QFLAG_EARLY_TERMINATED) != 0).all()
    # plain call == ex call without extras
    a1 = grp.search_batch(q, k, rk)
    a2, _, _, _ = grp.search_batch_ex(q, k, rk)
    assert np.array_equal(a1.docs, a2.docs) and np.array_equal(a1.stats, a2.stats)
    grp.close()
    [s.close() for s in shards]


def test_shard_group_on_two_devices(pkg, pyoracle):
    """the gather between DIFFERENT devices (peer copies over xGMI); skipped on one-GPU boxes"""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    b, bl, dg = pkg.binding, pkg.builder, pkg.datagen
    sh = __import__("importlib").import_module("opensearch_jvector_amd.sharding")
    n_total, d, k, rk, G = 8000, 64, 10, 60, 2
    q = dg.splitmix_uniform(43, 96, d)
    shards, ws = [], []
    for g in range(G):
        lo, hi = sh.shard_range(n_total, G, g)
        base = dg.splitmix_uniform(42, hi - lo, d, row_offset=lo)
        ix = bl.build_index_cpu(base, 0, R=32, L=60, pq_M=32, ord2doc=np.arange(lo, hi, dtype=np.int32), max_doc=n_total)
        shards.append(b.GpuIndex(ix, flags=b.DESC_FUSED_ADC, device=g))
        ws.append(pyoracle.Oracle(b, ix).search_batch(q, k, rk))
    assert shards[0].info().device == 0 and shards[1].info().device == 1
    grp = b.ShardGroup(shards)
    od, os_ = pyoracle.merge_topk(b, np.concatenate([w.docs for w in ws], axis=1), np.concatenate([w.scores for w in ws], axis=1), k)
    for gather in (0, 1, 0):   # peer copies, ONE RCCL all-gather of the pair buffers, and back
        grp.set_option("gather", gather)
        for _ in range(2):
            got = grp.search_batch(q, k, rk)
            assert np.array_equal(got.docs, od) and np.array_equal(got.scores.view(np.uint32), os_.view(np.uint32)), gather
    grp.close()
    [s.close() for s in shards]


def test_rccl_gather_executes_with_one_shard(pkg, pyoracle):
    """One-GPU boxes cannot run the shard group's RCCL gather between devices; a group of ONE shard can still take that path:
    librccl.so is opened, `ncclCommInitAll` builds a one-rank communicator, `ncclGroupStart / ncclAllGather / ncclGroupEnd` run on
    the shard's stream and the merge reads the gathered buffer — every RCCL entry point the library resolves has then executed."""
    b, bl, dg = pkg.binding, pkg.builder, pkg.datagen
    n, d, k, rk = 5000, 64, 10, 60
    base = dg.splitmix_uniform(42, n, d)
    q = dg.splitmix_uniform(43, 64, d)
    ix = bl.build_index_cpu(base, 0, R=32, L=60, pq_M=32, ord2doc=np.arange(n, dtype=np.int32) + 7, max_doc=n + 7)
    shard = b.GpuIndex(ix, flags=b.DESC_FUSED_ADC)
    want = pyoracle.Oracle(b, ix).search_batch(q, k, rk)
    grp = b.ShardGroup([shard])
    for gather in (0, 1, 1, 0):
        grp.set_option("gather", gather)
        got = grp.search_batch(q, k, rk)
        assert np.array_equal(got.docs, want.docs) and np.array_equal(got.scores.view(np.uint32), want.scores.view(np.uint32)), gather
    grp.close()
    shard.close()


def test_rccl_gather_is_refused_when_shards_share_a_device(pkg):
    b, bl, dg = pkg.binding, pkg.builder, pkg.datagen
    base = dg.splitmix_uniform(1, 500, 16)
    ix = bl.build_index_cpu(base, 0, R=8, L=20)
    s1, s2 = b.GpuIndex(ix), b.GpuIndex(ix)
    grp = b.ShardGroup([s1, s2])
    with pytest.raises(b.JvError):
        grp.set_option("gather", 1)
    grp.set_option("gather", 0)
    with pytest.raises(b.JvError):
        grp.set_option("no_such_option", 1)
    grp.close()
    s1.close()
    s2.close()


def test_device_batches_on_several_streams_use_their_own_contexts(pkg, pyoracle):
    """jv_search_batch_device from several streams at once (a server that receives the next batch of 256 queries while the previous
    ones are being answered: BASELINE config 5's shape, pipelined): every stream's calls run in a launch context of their own
    (`async_contexts`, default 4; more streams than contexts share, ordered behind the context's last use).  Every batch equals
    the oracle — ids, score bits, counters — whatever ran beside it."""
    import torch
    b, bl, dg = pkg.binding, pkg.builder, pkg.datagen
    n, d, k, rk, B, S, rounds = 20000, 64, 10, 120, 256, 6, 3
    base = dg.splitmix_uniform(42, n, d)
    ix = bl.build_index_cpu(base, 0, R=32, L=60, pq_M=32)
    gpu = b.GpuIndex(ix, flags=b.DESC_FUSED_ADC)
    dev = torch.device("cuda", 0)
    q = dg.splitmix_uniform(43, B * S * rounds, d)
    want = pyoracle.Oracle(b, ix).search_batch(q, k, rk)
    tq = torch.from_numpy(q).to(dev)
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    outs = [[(torch.full((B, k), -7, dtype=torch.int32, device=dev), torch.full((B, k), -7, dtype=torch.int32, device=dev),
              torch.zeros((B, k), dtype=torch.float32, device=dev), torch.zeros((B,), dtype=torch.int32, device=dev),
              torch.zeros((B, 4), dtype=torch.int32, device=dev), torch.zeros((B,), dtype=torch.int32, device=dev)) for _ in range(rounds)] for _ in range(S)]
    torch.cuda.synchronize()
    for r in range(rounds):
        for s_ in range(S):
            o = outs[s_][r]
            lo = (r * S + s_) * B
            gpu.search_batch_device(tq[lo:lo + B].data_ptr(), B, k, rk, *[t.data_ptr() for t in o], stream=streams[s_].cuda_stream)
    torch.cuda.synchronize()
    for r in range(rounds):
        for s_ in range(S):
            o = outs[s_][r]
            lo = (r * S + s_) * B
            assert np.array_equal(o[0].cpu().numpy(), want.nodes[lo:lo + B]), (r, s_)
            assert np.array_equal(o[2].cpu().numpy().view(np.uint32), want.scores[lo:lo + B].view(np.uint32)), (r, s_)
            assert np.array_equal(o[4].cpu().numpy(), want.stats[lo:lo + B]), (r, s_)
            assert (o[5].cpu().numpy() >= 0).all()
    gpu.close()


def test_null_stream_call_is_ordered_behind_the_default_stream(pkg):
    """jv_search_batch_device without a caller stream runs on the library's own non-blocking stream.  A caller that has just
    produced the queries on the legacy default stream (handle 0: torch's current stream unless told otherwise) must still be
    searched with the FINISHED queries: the call waits for what the default stream has in flight.  (Round 4: the graph builder
    handed stream handle 0 over and now and then searched half-written 1 536-d rows.)"""
    import torch
    b, bl, dg = pkg.binding, pkg.builder, pkg.datagen
    n, d, k, rk, B = 4000, 1536, 10, 40, 16384
    base = dg.splitmix_uniform(7, n, d)
    ix = bl.build_index_cpu(base, 0, R=16, L=40)
    gpu = b.GpuIndex(ix)
    dev = torch.device("cuda", 0)
    pool = torch.from_numpy(dg.splitmix_uniform(8, 20000, d)).to(dev)
    o = [torch.empty((B, k), dtype=torch.int32, device=dev), torch.empty((B, k), dtype=torch.int32, device=dev),
         torch.empty((B, k), dtype=torch.float32, device=dev), torch.empty((B,), dtype=torch.int32, device=dev),
         torch.zeros((B, 4), dtype=torch.int32, device=dev), torch.empty((B,), dtype=torch.int32, device=dev)]
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    for it in range(6):
        u = torch.randint(0, pool.shape[0], (B,), generator=g, device=dev)
        q = pool[u].contiguous()            # a 100 MB gather on the default stream, still running when the call below is made
        gpu.search_batch_device(q.data_ptr(), B, k, rk, *[t.data_ptr() for t in o])   # no stream: the library's own
        got = [t.clone() for t in o[:5]]
        torch.cuda.synchronize()
        gpu.search_batch_device(q.data_ptr(), B, k, rk, *[t.data_ptr() for t in o])   # the same call on finished queries
        torch.cuda.synchronize()
        assert torch.equal(got[0], o[0]) and torch.equal(got[2], o[2]) and torch.equal(got[4], o[4]), it
    gpu.close()
