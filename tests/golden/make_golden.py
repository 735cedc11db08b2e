"""Generates the committed golden fixtures under tests/golden/.  Run in the build container only
(it reads /root/reference); nothing under tests/ reads the reference at test time.

Outputs (data only — inputs and expected outputs):
  ka_cases.json            the analytic known-answer cases of the reference's own tests for the
                           search path (KNNJVectorTests.java et al., cited per case)
  recall_golden.json       vectors produced by importing the reference's Python recall harness
                           (scripts/jvector_index_and_search/jvector_utils/recall_measurement.py)
  reference_data_1000x128.npz   the reference's test data files (src/test/resources/data/
                           test_vectors_1000x128.json, test_queries_100x128.csv) as float32 arrays
"""
import csv
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
T = "src/test/java/org/opensearch/knn/index/codec/jvector/KNNJVectorTests.java"

f32 = np.float32


def l2_score(q, v):
    return float(f32(1) / (f32(1) + f32(np.sum((np.asarray(q, f32) - np.asarray(v, f32)) ** 2))))


def mip_score(q, v):  # Lucene MAXIMUM_INNER_PRODUCT for positive dot: 1 + dot
    return float(f32(1) + f32(np.dot(np.asarray(q, f32), np.asarray(v, f32))))


def cos_score(q, v):
    q, v = np.asarray(q, np.float64), np.asarray(v, np.float64)
    return float((1 + q.dot(v) / np.sqrt(q.dot(q) * v.dot(v))) / 2)


def ka_cases():
    cases = []
    # KA1 testJVectorKnnIndex_simpleCase
    docs = [{"doc": i - 1, "vector": [0.0, 1.0 / i]} for i in range(1, 11)]
    cases.append(dict(name="KA1_l2_simple", cite=f"{T}:73-130", lucene_similarity="EUCLIDEAN", docs=docs,
                      query=[0.0, 0.0], k=3, over_query_factor=5, filter_docs=None, deleted_docs=[],
                      expected_docs=[9, 8, 7],
                      expected_scores=[l2_score([0, 0], [0, 1 / 10]), l2_score([0, 0], [0, 1 / 9]), l2_score([0, 0], [0, 1 / 8])],
                      tol=1e-3))
    # KA2 testJVectorKnnIndex_simpleCase_maxInnerProduct
    docs = [{"doc": i - 1, "vector": [1.0 / i, 0.0]} for i in range(1, 11)]
    cases.append(dict(name="KA2_mip_simple", cite=f"{T}:141-201", lucene_similarity="MAXIMUM_INNER_PRODUCT", docs=docs,
                      query=[1.0, 0.0], k=3, over_query_factor=5, filter_docs=None, deleted_docs=[],
                      expected_docs=[0, 1, 2], expected_scores=[mip_score([1, 0], [1 / i, 0]) for i in (1, 2, 3)], tol=1e-3))
    # KA3 testJVectorKnnIndex_filter_maxInnerProduct (filter_field == "even" <=> i even <=> doc odd)
    cases.append(dict(name="KA3_mip_filter_even", cite=f"{T}:212-273", lucene_similarity="MAXIMUM_INNER_PRODUCT", docs=docs,
                      query=[1.0, 0.0], k=3, over_query_factor=5, filter_docs=[1, 3, 5, 7, 9], deleted_docs=[],
                      expected_docs=[1, 3, 5], expected_scores=[mip_score([1, 0], [1 / i, 0]) for i in (2, 4, 6)], tol=1e-3))
    # KA4 testMissing_fields: odd docs carry no vector
    docs = [{"doc": i, "vector": [0.0, float(i)] if i % 2 == 0 else None} for i in range(10)]
    cases.append(dict(name="KA4_l2_missing_fields", cite=f"{T}:279-336", lucene_similarity="EUCLIDEAN", docs=docs,
                      query=[0.0, 0.0], k=3, over_query_factor=5, filter_docs=None, deleted_docs=[],
                      expected_docs=[0, 2, 4], expected_scores=[l2_score([0, 0], [0, i]) for i in (0, 2, 4)], tol=1e-3))
    # KA5 test_sorted_index: index sort reverses doc order; doc id 9 holds TEST_ID 0
    docs = [{"doc": 9 - i, "vector": [0.0, float(i)], "test_id": i} for i in range(10)]
    cases.append(dict(name="KA5_l2_sorted_index", cite=f"{T}:343-417", lucene_similarity="EUCLIDEAN", docs=docs,
                      query=[0.0, 0.0], k=3, over_query_factor=5, filter_docs=None, deleted_docs=[],
                      expected_docs=[9, 8, 7], expected_test_ids=[0, 1, 2],
                      expected_scores=[l2_score([0, 0], [0, i]) for i in (0, 1, 2)], tol=1e-3))
    # KA6 testLuceneKnnIndex_mergeEnabled_withCompoundFile_cosine
    docs = [{"doc": i - 1, "vector": [1.0 + i, 2.0 * i]} for i in range(1, 11)]
    cases.append(dict(name="KA6_cosine_merged", cite=f"{T}:1218-1271", lucene_similarity="COSINE", docs=docs,
                      query=[1.0, 1.0], k=3, over_query_factor=5, filter_docs=None, deleted_docs=[],
                      expected_docs=[0, 1, 2], expected_scores=[cos_score([1, 1], [1 + i, 2 * i]) for i in (1, 2, 3)], tol=1e-3))
    # KA7 testJVectorKnnIndex_withFilter
    docs = [{"doc": i - 1, "vector": [0.0, 1.0 / i]} for i in range(1, 11)]
    cases.append(dict(name="KA7_l2_filter_even", cite=f"{T}:1301-1352", lucene_similarity="EUCLIDEAN", docs=docs,
                      query=[0.0, 0.0], k=3, over_query_factor=5, filter_docs=[1, 3, 5, 7, 9], deleted_docs=[],
                      expected_docs=[9, 7, 5], expected_scores=[l2_score([0, 0], [0, 1 / i]) for i in (10, 8, 6)], tol=1e-3))
    # KA10 deleted docs never returned (testJVectorKnnIndex_mergeEnabled_withDeletes shape, :1070-1138)
    cases.append(dict(name="KA10_l2_deleted_docs", cite=f"{T}:1070-1138", lucene_similarity="EUCLIDEAN", docs=docs,
                      query=[0.0, 0.0], k=3, over_query_factor=5, filter_docs=None, deleted_docs=[9, 8],
                      expected_docs=[7, 6, 5], expected_scores=[l2_score([0, 0], [0, 1 / i]) for i in (8, 7, 6)], tol=1e-3))
    return cases


def score_mapping_cases():
    """KA11: REST score formulas (src/test/java/org/opensearch/knn/index/engine/JVectorEngineIT.java:421-438,
    CommonTestUtils.java:84-93): L2 1/(1+d^2), cosine (1+cos)/2, innerproduct s<=0 ? 1/(1-s) : s+1."""
    rng = np.random.default_rng(11)
    out = []
    for _ in range(8):
        a = rng.random(6).astype(f32)
        b = rng.random(6).astype(f32)
        a64, b64 = a.astype(np.float64), b.astype(np.float64)
        out.append(dict(a=a.tolist(), b=b.tolist(),
                        l2=float(1 / (1 + np.sum((a64 - b64) ** 2))),
                        cosinesimil=float((1 + a64.dot(b64) / np.sqrt(a64.dot(a64) * b64.dot(b64))) / 2),
                        innerproduct=float(a64.dot(b64) + 1)))
    return out


def recall_golden():
    sys.path.insert(0, os.path.join(REF, "scripts/jvector_index_and_search"))
    from jvector_utils.recall_measurement import GroundTruthTracker, calculate_recall  # the reference's code
    rng = np.random.default_rng(123)
    out = {"cite": "scripts/jvector_index_and_search/jvector_utils/recall_measurement.py:12-108", "ground_truth": [], "recall": []}
    for space in ("l2", "cosine"):
        base = rng.random((60, 5)).astype(f32)
        base[17] = base[3]  # duplicate vectors: first seen wins
        queries = rng.random((4, 5)).astype(f32)
        for k in (1, 5, 100):
            trk = GroundTruthTracker([q for q in queries], k, space)
            for i, v in enumerate(base):
                trk.update(i, v)
            out["ground_truth"].append(dict(space=space, k=k, base=base.tolist(), queries=queries.tolist(),
                                            truth=[[int(x) for x in trk.get_ground_truth(i)] for i in range(len(queries))]))
    for approx, truth in ([[1, 2, 3, 4, 5], [1, 2, 3, 4, 5]], [[1, 2, 3, 4, 9], [1, 2, 3, 4, 5]], [[6, 7, 8], [1, 2, 3]],
                          [[], [1, 2]], [[1, 2], []], [[1, 1, 2], [1, 2, 3, 4]]):
        out["recall"].append(dict(approx=approx, truth=truth, recall=calculate_recall(approx, truth)))
    return out


def reference_data():
    with open(os.path.join(REF, "src/test/resources/data/test_vectors_1000x128.json")) as f:
        txt = f.read().strip()
    rows = []
    for line in txt.splitlines():
        line = line.strip().rstrip(",")
        if not line or line in "[]":
            continue
        obj = json.loads(line)
        vec = obj.get("vector") or obj.get("test_field") or next(v for v in obj.values() if isinstance(v, list))
        rows.append((obj.get("id", len(rows)), vec))
    base = np.asarray([r[1] for r in rows], dtype=f32)
    ids = np.asarray([int(r[0]) for r in rows], dtype=np.int32)
    qs = []
    with open(os.path.join(REF, "src/test/resources/data/test_queries_100x128.csv")) as f:
        for row in csv.reader(f):
            vals = [x for x in row if x.strip() != ""]
            if len(vals) >= 128:
                qs.append([float(x) for x in vals[-128:]])
    queries = np.asarray(qs, dtype=f32)
    return ids, base, queries


if __name__ == "__main__":
    with open(os.path.join(HERE, "ka_cases.json"), "w") as f:
        json.dump(dict(cases=ka_cases(), score_mapping=score_mapping_cases()), f, indent=1)
    with open(os.path.join(HERE, "recall_golden.json"), "w") as f:
        json.dump(recall_golden(), f)
    ids, base, queries = reference_data()
    print("reference data", base.shape, queries.shape, ids[:5], base[0, :4], queries[0, :4])
    np.savez_compressed(os.path.join(HERE, "reference_data_1000x128.npz"), ids=ids, base=base, queries=queries)
