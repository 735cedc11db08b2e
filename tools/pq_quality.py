#!/usr/bin/env python3
"""How much recall do the PQ codebooks leave on the table?  Same graph, same queries; the codebooks trained with different
Lloyd iteration counts / sample sizes / seeds; recall@10 at a few rerankK against brute-force ground truth on 2 048 queries.
env: N (docs, default 2M), DIST, RKS, CONFIGS ("iters:max_train:seed[:pp],...": a fourth field pp = k-means++ seeding)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__ as g
g.load_package()
b = importlib.import_module("opensearch_jvector_amd.binding")
gb = importlib.import_module("opensearch_jvector_amd.builder_gpu")
import bench

n = int(os.environ.get("N", 2_000_000)); d = 768; M = 32; NQ = 2048
rks = [int(x) for x in os.environ.get("RKS", "800,1000,1100,1200").split(",")]
cfgs = [tuple(c.split(":")) for c in os.environ.get("CONFIGS", "8:128000:1,8:128000:2,8:128000:1:pp,8:128000:2:pp,25:128000:1,25:128000:1:pp").split(",")]
dev = torch.device("cuda", 0)
base, q = bench.make_pq_data(torch, os.environ.get("DIST", "rotated"), n, NQ, d, M, 0, n, False, dev)
adj, entry = gb.build_graph_gpu(torch, base, 0, R=32, L=100, verbose=False)
gt = bench.brute_force_topk(torch, base, q, 10, 0)
o = [torch.empty((NQ, 10), dtype=torch.int32, device=dev), torch.empty((NQ, 10), dtype=torch.int32, device=dev),
     torch.empty((NQ, 10), dtype=torch.float32, device=dev), torch.empty((NQ,), dtype=torch.int32, device=dev),
     torch.zeros((NQ, 4), dtype=torch.int32, device=dev), torch.empty((NQ,), dtype=torch.int32, device=dev)]
for cfg in cfgs:
    iters, max_train, seed = int(cfg[0]), int(cfg[1]), int(cfg[2])
    seeding = "kmeans++" if len(cfg) > 3 and cfg[3] == "pp" else "random"
    t0 = time.time()
    pq = gb.pq_train_encode_gpu(torch, base, M, 0, iters=iters, max_train=max_train, seed=seed, seeding=seeding)
    torch.cuda.synchronize()
    t_train = time.time() - t0
    # distortion of the codes on the first 200 000 rows
    cb = torch.from_numpy(pq["codebooks"]).to(dev).view(M, 256, d // M)
    cen = torch.from_numpy(pq["centroid"]).to(dev)
    rows = base[:200000] - cen
    codes = pq["codes"][:200000].long()
    rec = torch.stack([cb[m][codes[:, m]] for m in range(M)], 1).reshape(rows.shape[0], d)
    dist = float(((rows - rec) ** 2).sum(1).mean())
    desc, keep = b.make_desc_device(n, d, 32, base.data_ptr(), adj.data_ptr(), entry, 0, pq_M=M, pq_K=pq["K"], pq_codebooks=pq["codebooks"],
                                    pq_centroid=pq["centroid"], pq_codes_ptr=pq["codes"].data_ptr(), borrow=True, extra_flags=b.DESC_FUSED_ADC)
    ix = b.GpuIndex(desc=desc, keepalive=keep, flags=b.DESC_BORROW)
    out = []
    for rk in rks:
        ix.search_batch_device(q.data_ptr(), NQ, 10, rk, *[t_.data_ptr() for t_ in o])
        torch.cuda.synchronize()
        out.append(f"{rk}: {bench.recall_of(o[1], gt):.4f}")
    print(f"iters {iters:3d} sample {max_train:7d} seed {seed} seeding {seeding:8s}: train+encode {t_train:5.1f} s, distortion {dist:.5f}; recall@10 " + "  ".join(out), flush=True)
    ix.close()
    del pq, cb, rec, codes
