"""Turns a known-answer case (tests/golden/ka_cases.json) into an index + a Lucene-style leaf search,
so the same fixtures pin the oracle (CPU) and the HIP engine (GPU).

The leaf logic mirrors what wraps the path in the reference: Lucene AbstractKnnVectorQuery (filter ->
AcceptDocs, cost <= k -> exact search, approximate search with visitLimit = cost, exact fallback when
the collector early-terminated) around JVectorKnnFloatVectorQuery.approximateSearch
(J/JVectorKnnFloatVectorQuery.java:50-70) and JVectorReader.search (J/JVectorReader.java:129-210)."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

# VectorSimilarityMapper (J/JVectorReader.java:384-432) + the MIP x2 wrap (:220-239)
LUCENE_SIM = {"EUCLIDEAN": (0, 1.0), "DOT_PRODUCT": (1, 1.0), "COSINE": (2, 1.0), "MAXIMUM_INNER_PRODUCT": (1, 2.0)}


def load_cases():
    with open(os.path.join(HERE, "golden", "ka_cases.json")) as f:
        return json.load(f)


def case_index(pkg, case, R=32, L=100):
    docs = sorted(case["docs"], key=lambda x: x["doc"])
    with_vec = [x for x in docs if x["vector"] is not None]
    vectors = np.asarray([x["vector"] for x in with_vec], dtype=np.float32)
    ord2doc = np.asarray([x["doc"] for x in with_vec], dtype=np.int32)
    sim, scale = LUCENE_SIM[case["lucene_similarity"]]
    ix = pkg.builder.build_index_cpu(vectors, sim, R=R, L=L, score_scale=scale, ord2doc=ord2doc, max_doc=len(docs))
    return ix


def leaf_search(pkg, engine, ix, case):
    """engine: object with search_batch(...) and score_ordinals(...). Returns (docs, scores, used_exact)."""
    b = pkg.binding
    k, oqf = case["k"], case["over_query_factor"]
    max_doc = ix.max_doc
    live = np.ones(max_doc, dtype=bool)
    live[case["deleted_docs"]] = False
    accept = None
    cost = None
    if case["filter_docs"] is not None:
        m = np.zeros(max_doc, dtype=bool)
        m[case["filter_docs"]] = True
        m &= live
        accept = m
        cost = int(m.sum())
    elif not live.all():
        accept = live  # Lucene hands liveDocs down as the accept bits
    q = np.asarray(case["query"], dtype=np.float32)

    def exact():
        doc2ord = {int(dd): o for o, dd in enumerate(ix.ord2doc)}
        cand_docs = [dd for dd in np.nonzero(accept if accept is not None else live)[0] if int(dd) in doc2ord]
        ords = np.asarray([doc2ord[int(dd)] for dd in cand_docs], dtype=np.int32)
        sc = engine.score_ordinals(q, ords)
        order = sorted(range(len(ords)), key=lambda i: (-sc[i], cand_docs[i]))[:k]
        return [int(cand_docs[i]) for i in order], [float(sc[i]) for i in order], True

    if cost is not None and cost <= k:
        return exact()
    words = None if accept is None else b.accept_words(np.nonzero(accept)[0], max_doc)
    res = engine.search_batch(q[None, :], k, k * oqf, accept=words, accept_num_docs=max_doc)
    visited = int(res.stats[0][0] + res.stats[0][2])  # visited + expanded (J/JVectorReader.java:204-207)
    if cost is not None and visited >= cost:
        return exact()  # collector.earlyTerminated() -> Lucene's exact fallback
    c = int(res.count[0])
    return [int(x) for x in res.docs[0][:c]], [float(x) for x in res.scores[0][:c]], False
