#!/usr/bin/env python3
"""Throughput of the several-waves kernel on row lengths WITHOUT a compiled instance of their own (d = 384, 1 024: the "any d"
instances, NCHT = 0) next to d = 768 (NCHT = 12) on the same kind of data: what the generic rerank costs.
env: N (docs, default 1M), B (queries per launch, 32768), RK (400), DIMS (384,768,1024), MS (32,64)"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__ as g
g.load_package()
b = importlib.import_module("opensearch_jvector_amd.binding")
gb = importlib.import_module("opensearch_jvector_amd.builder_gpu")
import bench

n = int(os.environ.get("N", 1_000_000)); B = int(os.environ.get("B", 32768)); rk = int(os.environ.get("RK", 400))
dev = torch.device("cuda", 0)
for d in [int(x) for x in os.environ.get("DIMS", "384,768,1024").split(",")]:
    for M in [int(x) for x in os.environ.get("MS", "32,64").split(",")]:
        if d % M:
            continue
        base, q = bench.make_pq_data(torch, "rotated", n, B, d, M, 0, n, False, dev)
        adj, entry = gb.build_graph_gpu(torch, base, 0, R=32, L=100, verbose=False)
        pq = gb.pq_train_encode_gpu(torch, base, M, 0)
        desc, keep = b.make_desc_device(n, d, 32, base.data_ptr(), adj.data_ptr(), entry, 0, pq_M=M, pq_K=pq["K"], pq_codebooks=pq["codebooks"],
                                        pq_centroid=pq["centroid"], pq_codes_ptr=pq["codes"].data_ptr(), borrow=True, extra_flags=b.DESC_FUSED_ADC)
        ix = b.GpuIndex(desc=desc, keepalive=keep, flags=b.DESC_BORROW)
        o = [torch.zeros((B, 10), dtype=torch.int32, device=dev), torch.zeros((B, 10), dtype=torch.int32, device=dev), torch.zeros((B, 10), dtype=torch.float32, device=dev),
             torch.zeros((B,), dtype=torch.int32, device=dev), torch.zeros((B, 4), dtype=torch.int32, device=dev), torch.zeros((B,), dtype=torch.int32, device=dev)]
        best = 1e9
        for it in range(3):
            torch.cuda.synchronize(); t = time.time()
            ix.search_batch_device(q.data_ptr(), B, 10, rk, *[x.data_ptr() for x in o])
            torch.cuda.synchronize(); best = min(best, time.time() - t)
        st = o[4].float().mean(0).cpu().numpy()
        bytes_q = st[2] * 32 * (M + 4) + st[1] * 4 * d
        print(f"d={d} PQ-{M} n={n} rerankK={rk}: {B / best:,.0f} QPS, {B * bytes_q / best / 1e9:,.0f} GB/s algorithmic ({B * bytes_q / best / 8e12:.3f} of HBM peak); "
              f"expansions {st[2]:.0f} reranked {st[1]:.0f}; flagged {(o[5] != 0).sum().item()}", flush=True)
        ix.close()
        del base, q, adj, pq
        torch.cuda.empty_cache()
