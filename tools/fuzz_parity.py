#!/usr/bin/env python3
"""Randomised differential run: small random indexes (random n, d, R, PQ shape, similarity, tie-heavy or continuous data,
malformed adjacency rows), random search parameters (rerankK 1..1500, filters of random selectivity, rerankFloor, kernel
routing options) through the C ABI against the oracle — ids, score bits and counters must be equal.  Not a pytest file:
`python tools/fuzz_parity.py [seconds] [seed]` on a GPU box; prints the first mismatching configuration and exits 1."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
g.load_package()
b = importlib.import_module("opensearch_jvector_amd.binding")
bl = importlib.import_module("opensearch_jvector_amd.builder")
po = g.load_oracle()

def run(budget, seed0):
    t0 = time.time()
    cases = checks = 0
    while time.time() - t0 < budget:
        seed = seed0 + cases
        rng = np.random.default_rng(seed)
        n = int(rng.integers(150, int(os.environ.get("FUZZ_NMAX", 2500))))   # FUZZ_NMAX=9000: pools of the 4 096-entry launches too
        R = int(rng.choice([8, 16, 32]))
        sim = int(rng.integers(0, 3))
        M = int(rng.choice([2, 4, 16, 32, 32, 64]))   # (32 and 64: the several-waves kernel)
        d = M * int(rng.integers(1, 4)) if M >= 16 else int(rng.choice([4, 8, 24]))
        ties = rng.random() < 0.4
        if ties:
            base = np.zeros((n, d), dtype=np.float32)
            base[:, :4] = rng.integers(0, 3, size=(n, 4))
            if sim == 2:
                base[:, 0] += 1.0          # no zero vectors under cosine
        else:
            base = rng.standard_normal((n, d)).astype(np.float32)
        adj = np.stack([rng.permutation(n)[:R] for _ in range(n)]).astype(np.int32)
        if rng.random() < 0.3:             # malformed rows: holes, self loops, repeated neighbours
            rows = rng.integers(0, n, size=n // 10)
            adj[rows, rng.integers(0, R, size=len(rows))] = -1
            adj[rows[::2], 0] = rows[::2]
            adj[rows[1::2], 1] = adj[rows[1::2], 2]
        cb, cen, codes, K = bl.pq_train_encode_cpu(base, M, sim)
        max_doc = 2 * n
        ord2doc = rng.permutation(max_doc)[:n].astype(np.int32) if rng.random() < 0.5 else None
        ix = b.IndexData(vectors=base, adj=adj, entry_node=int(rng.integers(0, n)), similarity=sim, pq_codebooks=cb, pq_centroid=cen,
                         pq_codes=codes, pq_M=M, pq_K=K, ord2doc=ord2doc, max_doc=(max_doc if ord2doc is not None else n))
        gpu = b.GpuIndex(ix, flags=b.DESC_FUSED_ADC)
        orc = po.Oracle(b, ix)
        opts = {}
        if rng.random() < 0.6:
            opts["lutr_min_queries"] = 0
        if rng.random() < 0.3:
            opts["no_lutr"] = 1
        # round 5: visited counts after the launch — small sets (several hash classes), small arenas (in-kernel fall-back), off
        if rng.random() < 0.3:
            opts["visited_slots"] = int(rng.choice([256, 512, 2048]))
        if rng.random() < 0.25:
            opts["visited_arena_units"] = int(rng.choice([4, 64, 1024]))
        if rng.random() < 0.2:
            opts["visited_after"] = 0
        for k_, v_ in opts.items():
            gpu.set_option(k_, v_)
        nq = int(rng.integers(8, 70))
        q = (rng.integers(0, 3, size=(nq, d)).astype(np.float32) + np.float32(0.5) * (rng.random((nq, d)) < 0.3)) if ties else rng.standard_normal((nq, d)).astype(np.float32)
        if ties:
            q[:, 4:] = 0
            if sim == 2:
                q[:, 0] += 1.0
        for _ in range(4):
            rk = int(rng.choice([1, 3, 10, 40, 100, 192, 200, 400, 900, 1500, 2500, 3900]))
            rk = min(rk, n)
            k = int(min(rk, rng.choice([1, 5, 10, 50])))
            kw = {}
            if rng.random() < 0.6:
                nd = ix.max_doc if ord2doc is not None else n
                frac = float(rng.choice([0.95, 0.7, 0.4, 0.15, 0.03]))
                kw["accept"] = b.accept_words(np.nonzero(rng.random(nd) < frac)[0], nd)
                kw["accept_num_docs"] = nd
            if rng.random() < 0.2:
                kw["rerank_floor"] = float(rng.choice([0.3, 0.6, 100.0]))
            want = orc.search_batch(q, k, rk, **kw)
            got = gpu.search_batch(q, k, rk, **kw)
            checks += 1
            ok = (np.array_equal(got.count, want.count) and np.array_equal(got.nodes, want.nodes) and np.array_equal(got.docs, want.docs) and
                  np.array_equal(got.stats, want.stats) and np.array_equal(got.scores.view(np.uint32), want.scores.view(np.uint32)))
            if not ok:
                bad = [i for i in range(nq) if not (np.array_equal(got.nodes[i], want.nodes[i]) and np.array_equal(got.stats[i], want.stats[i]))]
                print(f"MISMATCH seed={seed} n={n} d={d} R={R} M={M} sim={sim} ties={ties} opts={opts} k={k} rk={rk} "
                      f"filter={'accept' in kw} floor={kw.get('rerank_floor')} bad_queries={bad[:5]}")
                if bad:
                    i = bad[0]
                    print(" got ", got.nodes[i][:10], got.stats[i], "\n want", want.nodes[i][:10], want.stats[i])
                return False
            # Lucene's visit limit on the same call: exactly the searches whose visited + expanded reaches it come back early-terminated
            # (no results), the others unchanged — whichever kernel counted, inside the search launch or after it
            if rng.random() < 0.3 and "rerank_floor" not in kw:
                work = want.stats[:, 0].astype(np.int64) + want.stats[:, 2]
                lim = max(1, int(np.quantile(work, float(rng.choice([0.2, 0.5, 0.9])))))
                got, status, flags, rc = gpu.search_batch_ex(q, k, rk, visit_limit=lim, **kw)
                early = (flags & b.QFLAG_EARLY_TERMINATED) != 0
                checks += 1
                if not (np.array_equal(early, work >= lim) and np.array_equal(got.nodes[~early], want.nodes[~early]) and
                        np.array_equal(got.stats[~early], want.stats[~early]) and (got.count[early] == 0).all() and (got.nodes[early] == -1).all()):
                    print(f"VISIT-LIMIT MISMATCH seed={seed} n={n} d={d} R={R} M={M} sim={sim} ties={ties} opts={opts} k={k} rk={rk} lim={lim} filter={'accept' in kw}")
                    print(" early got ", early.astype(int)[:32], "\n early want", (work >= lim).astype(int)[:32])
                    return False
        # the batched exact scorer on the same index (jv_score_ordinals_batch: bf16 matrix-core pre-filter forced on for these
        # short lists half of the time, canonical re-score) against the oracle's scan + (score desc, doc asc)
        docs_of = ord2doc if ord2doc is not None else np.arange(n, dtype=np.int32)
        for _ in range(2):
            nd = ix.max_doc if ord2doc is not None else n
            kx = int(rng.choice([1, 5, 10, 50, 200]))
            nqx = int(rng.integers(1, nq + 1))
            flags = int(rng.choice([0, b.XB_FORCE_PREFILTER, b.XB_FORCE_PREFILTER, b.XB_NO_PREFILTER]))
            mode = int(rng.integers(0, 3))
            if mode == 0:
                frac = float(rng.choice([0.95, 0.4, 0.1, 0.01]))
                accd = np.nonzero(rng.random(nd) < frac)[0]
                cand = np.nonzero(np.isin(docs_of, accd) & (docs_of >= 0))[0].astype(np.int32)
                got = gpu.score_ordinals_batch(q[:nqx], kx, accept=b.accept_words(accd, nd), accept_num_docs=nd, flags=flags)
            elif mode == 1:
                lst = np.concatenate([rng.integers(0, n, size=int(rng.integers(1, 2 * n))), [-1, n, n + 7]]).astype(np.int32)
                lst = np.unique(lst)   # (duplicates in a caller's list are returned as given: not what Lucene produces)
                cand = lst[(lst >= 0) & (lst < n)]
                cand = cand[docs_of[cand] >= 0]
                got = gpu.score_ordinals_batch(q[:nqx], kx, ordinals=lst, flags=flags)
            else:
                cand = np.nonzero(docs_of >= 0)[0].astype(np.int32)
                got = gpu.score_ordinals_batch(q[:nqx], kx, flags=flags)
            checks += 1
            for i in range(nqx):
                sc = orc.score_ordinals(q[i], cand)
                dd = docs_of[cand]
                order = np.lexsort((dd, -sc.astype(np.float64)))[:kx]
                m = len(order)
                if not (got[3][i] == m and np.array_equal(got[1][i, :m], dd[order]) and np.array_equal(got[0][i, :m], cand[order]) and
                        np.array_equal(got[2][i, :m].view(np.uint32), sc[order].view(np.uint32))):
                    print(f"MISMATCH (exact batch) seed={seed} n={n} d={d} sim={sim} ties={ties} k={kx} nq={nqx} mode={mode} flags={flags} query={i} info={got[4]}")
                    print(" got ", got[1][i, :10], got[2][i, :5], "\n want", dd[order][:10], sc[order][:5])
                    return False
        gpu.close()
        cases += 1
    print(f"fuzz ok: {cases} indexes, {checks} search configurations, {time.time() - t0:.0f}s")
    return True


if __name__ == "__main__":
    ok = run(float(sys.argv[1]) if len(sys.argv) > 1 else 120.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    sys.exit(0 if ok else 1)
