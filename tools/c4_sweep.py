#!/usr/bin/env python3
"""C4 shard (12.5M x 1536, PQ-64, one of 8 doc-range shards of the 100M corpus): recall@10 and QPS over rerankK, and the
mixtureB question of VERDICT r2 #4(b): is the low recall of PQ-32 on SURVEY 8(d)'s distribution B the graph's or the
codes'?  (exact-provider search on a 1M mixtureB index).  env: WHAT=c4|mixb|both, N (docs), B (queries per launch)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__ as g
g.load_package()
b = importlib.import_module("opensearch_jvector_amd.binding")
gb = importlib.import_module("opensearch_jvector_amd.builder_gpu")
import bench

dev = torch.device("cuda", 0)
what = os.environ.get("WHAT", "both")


def run(ix, q, k, rk, B):
    o = [torch.empty((B, k), dtype=torch.int32, device=dev), torch.empty((B, k), dtype=torch.int32, device=dev),
         torch.empty((B, k), dtype=torch.float32, device=dev), torch.empty((B,), dtype=torch.int32, device=dev),
         torch.empty((B, 4), dtype=torch.int32, device=dev), torch.empty((B,), dtype=torch.int32, device=dev)]
    best = 1e9
    for _ in range(2):
        torch.cuda.synchronize(); t = time.time()
        ix.search_batch_device(q.data_ptr(), B, k, rk, *[x.data_ptr() for x in o])
        torch.cuda.synchronize(); best = min(best, time.time() - t)
    return o[0].cpu().numpy(), o[4].cpu().numpy(), B / best


if what in ("c4", "both"):
    n = int(os.environ.get("N", 12_500_000)); d, M, k = 1536, 64, 10
    B = int(os.environ.get("B", 16384))
    t0 = time.time()
    base, q = bench.make_pq_data(torch, "rotated", n, B, d, M, 3 * n, 8 * n, False, dev)
    adj, entry = gb.build_graph_gpu(torch, base, 0, R=32, L=100, verbose=False)
    pq = gb.pq_train_encode_gpu(torch, base, M, 0)
    desc, keep = b.make_desc_device(n, d, 32, base.data_ptr(), adj.data_ptr(), entry, 0, pq_M=M, pq_K=pq["K"], pq_codebooks=pq["codebooks"],
                                    pq_centroid=pq["centroid"], pq_codes_ptr=pq["codes"].data_ptr(), borrow=True, extra_flags=b.DESC_FUSED_ADC)
    ix = b.GpuIndex(desc=desc, keepalive=keep, flags=b.DESC_BORROW)
    print(f"c4 shard ready in {time.time() - t0:.0f} s", flush=True)
    truth = bench.brute_force_topk(torch, base, q[:512], k, 0).cpu().numpy()
    for rk in [int(x) for x in os.environ.get("RKS", "400,800,1200,1600,1900,2400,3200,3900").split(",")]:
        nodes, stats, qps = run(ix, q, k, rk, B)
        rec = np.mean([len(set(nodes[i]) & set(truth[i])) / k for i in range(512)])
        print(f"c4 rk={rk}: recall@10 {rec:.4f}  {qps:,.0f} QPS  expanded/query {stats[:, 2].mean():.0f}  pqw launches {ix.counter('launches_pqw')}", flush=True)
    ix.close()
    del base, q, adj, pq
    torch.cuda.empty_cache()

if what in ("mixb", "both"):
    n = int(os.environ.get("NMIX", 1_000_000)); d, M, k = 768, 32, 10
    B = 4096
    base, q = bench.make_pq_data(torch, "mixtureB", n, B, d, M, 0, n, False, dev)
    adj, entry = gb.build_graph_gpu(torch, base, 0, R=32, L=100, verbose=False)
    truth = bench.brute_force_topk(torch, base, q[:1024], k, 0).cpu().numpy()
    # exact provider (no PQ): graph quality alone
    desc, keep = b.make_desc_device(n, d, 32, base.data_ptr(), adj.data_ptr(), entry, 0, borrow=True)
    ix = b.GpuIndex(desc=desc, keepalive=keep, flags=b.DESC_BORROW)
    for rk in (50, 100, 200, 400):
        nodes, stats, qps = run(ix, q, k, rk, B)
        rec = np.mean([len(set(nodes[i]) & set(truth[i])) / k for i in range(1024)])
        print(f"mixtureB 1M exact provider rk={rk}: recall@10 {rec:.4f}  {qps:,.0f} QPS", flush=True)
    ix.close()
    pq = gb.pq_train_encode_gpu(torch, base, M, 0)
    desc, keep = b.make_desc_device(n, d, 32, base.data_ptr(), adj.data_ptr(), entry, 0, pq_M=M, pq_K=pq["K"], pq_codebooks=pq["codebooks"],
                                    pq_centroid=pq["centroid"], pq_codes_ptr=pq["codes"].data_ptr(), borrow=True, extra_flags=b.DESC_FUSED_ADC)
    ix = b.GpuIndex(desc=desc, keepalive=keep, flags=b.DESC_BORROW)
    for rk in (400, 1600):
        nodes, stats, qps = run(ix, q, k, rk, B)
        rec = np.mean([len(set(nodes[i]) & set(truth[i])) / k for i in range(1024)])
        print(f"mixtureB 1M PQ-32 rk={rk}: recall@10 {rec:.4f}  {qps:,.0f} QPS", flush=True)
    ix.close()
