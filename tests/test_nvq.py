"""NVQ-inline vectors: the oracle's dequantiser against golden vectors computed from the reference's
formulas (tests/golden/make_nvq_golden.py), and the HIP path against the oracle on NVQ-only and NVQ + PQ indexes."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def test_oracle_nvq_dequantiser_matches_the_reference_formulas(pkg, pyoracle):
    b = pkg.binding
    g = json.load(open(os.path.join(HERE, "golden", "nvq_golden.json")))
    assert len(g["cases"]) >= 9
    for c in g["cases"]:
        d, M = c["d"], c["M"]
        ix = b.IndexData(vectors=np.zeros((1, d), np.float32), adj=np.full((1, 1), -1, np.int32), entry_node=0)
        ix.nvq_M = M
        ix.nvq_params = np.asarray(c["params"], np.float32).reshape(1, M, 4)
        ix.nvq_bytes = np.asarray(c["codes"], np.uint8).reshape(1, d)
        ix.nvq_global_mean = np.asarray(c["mean"], np.float32)
        got = pyoracle.Oracle(b, ix).nvq_dequantize(0)
        want = np.asarray(c["expected"], np.float32)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (d, M, got, want)


def _nvq_index(pkg, base, sim, pq_M):
    b, bl = pkg.binding, pkg.builder
    ix = bl.build_index_cpu(base, sim, R=16, L=60, pq_M=pq_M)
    params, codes, mean = bl.nvq_encode(base, 2)
    ix.nvq_M, ix.nvq_params, ix.nvq_bytes, ix.nvq_global_mean = 2, params, codes, mean
    return ix


def test_nvq_decode_is_close_to_the_original_vectors(pkg, pyoracle):
    base = pkg.datagen.splitmix_uniform(5, 200, 50) - np.float32(0.5)
    ix = _nvq_index(pkg, base, 0, 0)
    orc = pyoracle.Oracle(pkg.binding, ix)
    err = max(np.abs(orc.nvq_dequantize(i) - base[i]).max() for i in range(0, 200, 7))
    assert err < 0.02, err     # 8 bits over a range of ~1


@pytest.mark.gpu
@pytest.mark.parametrize("sim", [0, 1, 2])
@pytest.mark.parametrize("d", [50, 128])
def test_nvq_gpu_parity(pkg, pyoracle, sim, d):
    """NVQ-only field (exact provider over dequantised vectors, J/JVectorReader.java:357-358) and NVQ + PQ (approximate
    PQ search, rerank against the dequantised records): ids, score bits, counters == oracle; jv_score_ordinals too."""
    b = pkg.binding
    base = pkg.datagen.splitmix_uniform(40 + d, 3000, d) - np.float32(0.4)
    q = pkg.datagen.splitmix_uniform(41 + d, 32, d) - np.float32(0.4)
    for pq_M, flags in ((0, 0), (16, 0), (16, b.DESC_FUSED_ADC)):
        ix = _nvq_index(pkg, base, sim, pq_M)
        orc = pyoracle.Oracle(b, ix)
        gpu = b.GpuIndex(ix, flags=flags)
        for k, rk in ((10, 50), (5, 5)):
            want = orc.search_batch(q, k, rk)
            got = gpu.search_batch(q, k, rk)
            assert np.array_equal(got.nodes, want.nodes) and np.array_equal(got.stats, want.stats), (pq_M, flags, k, rk)
            assert np.array_equal(got.scores.view(np.uint32), want.scores.view(np.uint32)), (pq_M, flags, k, rk)
        ords = np.arange(0, 3000, 11, dtype=np.int32)
        assert np.array_equal(gpu.score_ordinals(q[0], ords).view(np.uint32), orc.score_ordinals(q[0], ords).view(np.uint32))
        # the same field WITHOUT full-precision vectors behind the ABI (vectors = NULL): identical answers
        gpu.close()
    ixn = _nvq_index(pkg, base, sim, 16)
    want = pyoracle.Oracle(b, ixn).search_batch(q, 10, 50)
    desc, keep = b.make_desc(ixn, flags=b.DESC_FUSED_ADC)
    desc.vectors = None
    gpu = b.GpuIndex(desc=desc, keepalive=keep, flags=b.DESC_FUSED_ADC)
    got = gpu.search_batch(q, 10, 50)
    assert np.array_equal(got.nodes, want.nodes) and np.array_equal(got.scores.view(np.uint32), want.scores.view(np.uint32))
    gpu.close()
