"""Parity at BASELINE.json's full sizes, through size-independent properties and an oracle sample.

C2 (1M x 768 dot, fp32) and C3 (10M x 768 L2, PQ-32 + rerank) are built on the GPU exactly as bench.py does.
Checked: every result row is sorted (score desc, ordinal asc), idempotence (same batch twice), batch == single
query, fused layout == plain layout == generic kernel, recall against a GPU brute force, and — where the index fits
a host copy comfortably — ids / score bits / counters equal to the CPU oracle on a query sample."""
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
    assert np.array_equal(r1["nodes"][:512], want.nodes)
    assert np.array_equal(r1["scores"][:512].view(np.uint32), want.scores.view(np.uint32))
    assert np.array_equal(r1["stats"][:512], want.stats)
    gpu.close()


def test_c3_full_size_10m_768_pq32(pkg):
    torch, bench, gb = _setup()
    b = pkg.binding
    dev = torch.device("cuda", 0)
    n, d, M, k, rk = 10_000_000, 768, 32, 10, 160
    zc, Bl, Bg = bench.make_block_generators(torch, d, dev, 4096, M=M, per=2)
    base = bench.gen_rows_block(torch, n, d, 42, 0, zc, Bl, Bg, 0.35, 0.1, 0.005, False, dev)
    q = bench.gen_rows_block(torch, 4096, d, 43, 0, zc, Bl, Bg, 0.35, 0.1, 0.005, False, dev)
    adj, entry = gb.build_graph_gpu(torch, base, 0, R=32, L=100, verbose=False)
    pq = gb.pq_train_encode_gpu(torch, base, M, 0)

    def make(flags):
        desc, keep = b.make_desc_device(n, d, 32, base.data_ptr(), adj.data_ptr(), entry, 0, pq_M=M, pq_K=pq["K"],
                                        pq_codebooks=pq["codebooks"], pq_centroid=pq["centroid"],
                                        pq_codes_ptr=pq["codes"].data_ptr(), borrow=True, extra_flags=flags)
        return b.GpuIndex(desc=desc, keepalive=keep, flags=b.DESC_BORROW)

    fused = make(b.DESC_FUSED_ADC)
    r1 = _search(torch, fused, q, k, rk)
    _check_rows_sorted(r1, k)
    assert (r1["stats"][:, 1] == rk).all(), "every query exact-rescored rerankK candidates"
    r2 = _search(torch, fused, q, k, rk)
    for key in ("nodes", "scores", "stats"):
        assert np.array_equal(r1[key], r2[key]), f"idempotence: {key}"
    truth = bench.brute_force_topk(torch, base, q[:512], k, 0).cpu().numpy()
    rec = np.mean([len(set(r1["nodes"][i]) & set(truth[i])) / k for i in range(512)])
    assert rec >= 0.93, rec
    # exact-rerank scores are true L2 similarities of the returned ids
    ids = torch.from_numpy(r1["nodes"][:64].astype(np.int64)).to(dev)
    d2 = ((q[:64, None, :].double() - base[ids].double()) ** 2).sum(-1)
    np.testing.assert_allclose(r1["scores"][:64], (1.0 / (1.0 + d2)).cpu().numpy(), rtol=1e-4)
    # the specialised kernel, the generic pool kernel on the fused layout, and the plain layout agree bit for bit
    try:
        b.set_option("no_pqf", 1)
        r3 = _search(torch, fused, q[:1024], k, rk)
    finally:
        b.set_option("no_pqf", 0)
    fused.close()
    plain = make(0)
    r4 = _search(torch, plain, q[:1024], k, rk)
    plain.close()
    for key in ("nodes", "scores", "stats"):
        assert np.array_equal(r1[key][:1024], r3[key]), f"pqf vs generic: {key}"
        assert np.array_equal(r1[key][:1024], r4[key]), f"fused vs plain layout: {key}"
